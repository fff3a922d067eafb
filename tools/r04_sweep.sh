#!/bin/bash
# Dev tool (GPU box): us per step of a list of tools/build_diag.py variants (TAGS), product build first.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_sweep}; mkdir -p $O
cd $R
timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 > $O/quick.txt 2>&1 || exit 1
for t in ${TAGS}; do
  ABLTAG=$t timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
done
grep "us per step" $O/quick.txt
