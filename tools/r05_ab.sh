#!/bin/bash
# Dev tool (GPU box), round 5: us per step of the product build and of the build/abl variants in TAGS, default model and round-4 model
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_ab}; mkdir -p $O
cd $R
for model in "32 subtree_weights" "16 lumped"; do
  set -- $model; export QB_ITERS=$1 QB_MASS=$2
  timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
  for t in ${TAGS}; do
    ABLTAG=$t timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
  done
done
grep "us per step" $O/quick.txt
