#!/bin/bash
# Dev tool (GPU box): FrankaCubeLift-v0 parity tests + step time of both lane mappings + stamps + rocprofv3 kernel stats.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03lift; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_lift.py -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
timeout -k 10 120 python3 tools/lift_time.py 2048 > $O/time.log 2>&1 && LIFT_LANES=16 timeout -k 10 120 python3 tools/lift_time.py 2048 >> $O/time.log 2>&1 && timeout -k 10 120 python3 tools/lift_time.py 8192 >> $O/time.log 2>&1
grep product $O/time.log
timeout -k 10 120 python3 tools/lift_stamps.py 2048 > $O/stamps.txt 2>&1; tail -18 $O/stamps.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/lift_time.py 2048 > $O/stats.log 2>&1
echo "rc=$?"
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-4 {} | cut -c1-70,170- | head -4'
