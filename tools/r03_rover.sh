#!/bin/bash
# Dev tool (GPU box): rover GPU tests + bench + kernel stats.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03rover; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_boundary.py tests/test_gpu_trainer_replay.py -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/stats.log 2>&1
echo "rc=$?"
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-4 {} | cut -c1-60,150- | head -4'
