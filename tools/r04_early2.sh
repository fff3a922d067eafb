#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_early2}; mkdir -p $O
cd $R
timeout -k 10 120 python3 tools/k1_lite.py > $O/k1_lite.txt 2>&1; echo "lite rc=$?"; cat $O/k1_lite.txt
for t in ${TAGS}; do
  ABLTAG=$t timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
done
grep "us per step" $O/quick.txt
