#!/bin/bash
# Dev tool (GPU box), round 5: the rover path's parity files on the product build, then us per step of it and of the variants in TAGS
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_abp}; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_golden.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
bash tools/r05_ab.sh $1
