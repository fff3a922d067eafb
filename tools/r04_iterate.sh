#!/bin/bash
# Dev tool (GPU box): one iteration on the one-launch kernel: the rover path's parity files first, then the two-wave timeline
# (tools/k1_lite.py, needs `python tools/build_diag.py K1LITE`) and us per step of the product build and of the variants in TAGS.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_early}; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_user_terms.py tests/test_gpu_golden.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 120 python3 tools/k1_lite.py > $O/k1_lite.txt 2>&1; echo "lite rc=$?"; cat $O/k1_lite.txt
timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 > $O/quick.txt 2>&1 || exit 1
for t in ${TAGS}; do
  ABLTAG=$t timeout -k 10 120 python3 tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
done
grep "us per step" $O/quick.txt
