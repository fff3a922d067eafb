#!/bin/bash
# Dev tool (GPU box): instruction-cache counters of the step kernel (is the 65 KB kernel missing in the 64 KB instruction cache?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_icache}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/p -- python3 $R/tools/pmc_run.py 4096 20 > $O/p.log 2>&1 || { tail -5 $O/p.log; exit 1; }
rocprofv3 --kernel-trace --pmc InstrFetchLatency --output-format csv -d $O/q -- python3 $R/tools/pmc_run.py 4096 20 > $O/q.log 2>&1 || { tail -5 $O/q.log; }
rocprofv3 --kernel-trace --pmc SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB --output-format csv -d $O/r -- python3 $R/tools/pmc_run.py 4096 20 > $O/r.log 2>&1 || { tail -5 $O/r.log; }
python3 - <<PY > $O/icache.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/[pqr]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rover_step_scan_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k, v in sorted(acc.items()):
    v.sort(); w = [x for _, x in v]
    print(k, "first launches", w[:3], "mean of the rest", sum(w[2:]) / max(len(w[2:]), 1))
PY
cat $O/icache.txt; rm -rf $O/p $O/q $O/r
