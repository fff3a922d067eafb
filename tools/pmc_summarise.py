#!/usr/bin/env python3
"""Dev tool: per-kernel mean of every counter in a rocprofv3 --pmc output directory (counter_collection.csv files).
Kernel names are printed as rocprofv3's kernel trace prints them minus qualifiers / parameter list (= rover_kernel_names())."""
import csv, glob, re, sys, collections


def short_name(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void\s+", "", name)
    depth, out = 0, []
    for ch in name:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()[:90]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short_name(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if not k.startswith(("rover_", "lift_")):
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
