#!/usr/bin/env python3
"""Dev tool: per-kernel mean of every counter in a rocprofv3 --pmc output directory (counter_collection.csv files)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
