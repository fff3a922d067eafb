#!/usr/bin/env python3
"""Dev tool: profiles/issue_counters.json from the rocprofv3 --pmc passes of tools/r04_counters.sh -- per-launch means of the
issue / wait / LDS counters of the step-path kernels (launches 3.. of tools/pmc_run.py), keyed by the kernel name exactly as
rocprofv3 prints it (= rover_kernel_names()).  bench.py turns them into roofline.issue.
Usage: pmc_issue.py <dir with p1 p2 p3> <out.json> [build note]"""
import csv, glob, json, re, sys, collections


def short_name(name):
    """rocprofv3's kernel name without return type, '(anonymous namespace)::' qualifiers and the parameter list."""
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
    return "".join(out).strip()


def stamp():
    """What the counters belong to: the source digest of this checkout's librover_hip.so and the solver / mass settings of the run
    (QB_ITERS / QB_MASS of tools/pmc_run.py, else the cfg defaults) -- bench.py prints a summary only beside the build it describes."""
    import hashlib, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    c = RoverEnvCfg()
    from isaac_rover_orbit_amd import build
    return {"_lib_sha256": build.source_digest() + ":librover_hip.so",   # sources + headers: stable across rebuilds (bench.py: library_digest)
            "_solver_iterations": int(os.environ.get("QB_ITERS") or c.solver_iterations),
            "_mass_model": os.environ.get("QB_MASS") or c.mass_model}


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short_name(r["Kernel_Name"])
        if k.startswith(("rover_step", "rover_scan_step", "lift_step")):
            acc[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
out = {"_how": "rocprofv3 --kernel-trace --pmc (three separate passes, tools/r04_counters.sh) -- python3 tools/pmc_run.py 4096 20; "
               "per-launch means over launches 3..20, N = 4096, 1x MI355X.  SQ_*_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count in units "
               "of 4 shader cycles summed over waves; simds = 4 per CU x 256 CUs.",
       "_build": sys.argv[3] if len(sys.argv) > 3 else "", **stamp()}
for k, d in acc.items():
    out[k] = {"simds": 1024}
    for c, v in d.items():
        v.sort()
        v = [x for _, x in v][2:] or [x for _, x in v]
        out[k][c] = sum(v) / len(v)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
