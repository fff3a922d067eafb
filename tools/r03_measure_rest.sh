#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
python3 tools/lift_stamps.py 2048 > $O/lift_stamps.txt 2>&1 && \
python3 tools/policy_stamps.py > $O/policy_stamps.txt 2>&1 && \
python3 tools/k1_stamps.py > $O/k1_stamps.txt 2>&1 && \
( python3 tools/lift_time.py 2048; LIFT_LANES=16 python3 tools/lift_time.py 2048; python3 tools/lift_time.py 8192 ) > $O/lift_time.txt 2>&1 && \
python3 tools/n_sweep.py > $O/n_sweep.txt 2>&1
echo "rc=$?"
