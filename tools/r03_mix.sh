#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03mix; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_lift.py tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python3 tools/lift_time.py 2048 2>&1 | grep product; python3 tools/lift_time.py 8192 2>&1 | grep product
python3 tools/k1_stamps.py > $O/k1_stamps.txt 2>&1; tail -9 $O/k1_stamps.txt
python3 bench.py --no-cpu-baseline 2>/dev/null | cut -c1-200
