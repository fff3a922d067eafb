#!/bin/bash
# Dev tool (GPU box), round 5 experiment: where the L2-side fetch of the one-launch kernel comes from -- FETCH_SIZE and us per step with the
# envs of every XCD standing in one strip of the terrain (QB_CLUSTER=1) against the usual random placement
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_cluster}; mkdir -p $O
cd $R
timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
export QB_CLUSTER=1
timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
unset QB_CLUSTER
grep "us per step" $O/quick.txt
cd /tmp && export TMPDIR=/tmp
for c in 0 1; do
  if [ $c = 1 ]; then export QB_CLUSTER=1; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_$c -- python3 $R/tools/pmc_run.py 4096 20 > $O/f_$c.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/t_$c -- python3 $R/tools/pmc_run.py 4096 20 > $O/t_$c.log 2>&1 || exit 1
done
unset QB_CLUSTER
python3 - <<PY > $O/fetch.txt
import csv, glob, collections
for tag in ("0", "1"):
    acc = collections.defaultdict(list)
    for d in ("f_", "t_"):
        for f in glob.glob("$O/%s%s/**/*counter_collection.csv" % (d, tag), recursive=True):
            for r in csv.DictReader(open(f)):
                if "rover_step_scan_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = {}
    for k, v in acc.items():
        v.sort(); v = [x for _, x in v][2:]
        out[k] = sum(v) / len(v)
    print("clustered" if tag == "1" else "random   ", {k: round(v, 1) for k, v in out.items()}, "hit rate", round(out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]), 4))
PY
cat $O/fetch.txt; rm -rf $O/f_0 $O/f_1 $O/t_0 $O/t_1
