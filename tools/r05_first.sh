#!/bin/bash
# Dev tool (GPU box), round 5: the whole -m gpu selection, then us per step of the product build under the model / iteration settings
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_a}; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
for it in 32 16; do for m in subtree_weights lumped; do
  QB_ITERS=$it QB_MASS=$m timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
done; done
grep "us per step" $O/quick.txt
