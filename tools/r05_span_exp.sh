#!/bin/bash
# Dev tool (GPU box), round 5: timing experiment -- us per step with 100 / 70 / 50 % of every window row's chunks copied (builds
# X_SPAN70 / X_SPAN50: WRONG results, timing only), round-4 model and default; the no-wait timeline of the product build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_span}; mkdir -p $O
cd $R
for model in "16 lumped" "32 subtree_weights"; do
  set -- $model; export QB_ITERS=$1 QB_MASS=$2
  timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
  for t in X_SPAN70 X_SPAN50; do
    ABLTAG=$t timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
  done
done
unset QB_ITERS QB_MASS
timeout -k 10 120 python3 tools/k1_lite.py > $O/k1_lite.txt 2>&1; echo "lite rc=$?"
grep "us per step" $O/quick.txt; cat $O/k1_lite.txt
