#!/bin/bash
# Dev tool (GPU box): one build -> measure iteration of round 4.  $1 = tag (output under gpurun_out/$1), $2 = "all" for the whole
# -m gpu suite (default: the parity / config files of the rover path).  Needs `python tools/build_diag.py K1STAMP` first.
R=$GRAFT_REPO_ROOT; TAG=${1:-r04_iter}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
if [ "$2" = "all" ]; then T="tests"; else T="tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_golden.py tests/test_gpu_boundary.py"; fi
timeout -k 10 600 python3 -m pytest $T -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 120 python3 tools/k1_stamps.py > $O/k1_stamps.txt 2>&1; echo "stamps rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/stats_bench.log 2>&1 && \
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/stats
cd $R && timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
grep -A1 "scan: barrier B" $O/k1_stamps.txt | head -1; tail -6 $O/k1_stamps.txt; grep "rover_step_scan\|rover_step_kernel\|rover_scan_step" $O/kernel_stats.csv | cut -c1-60,250-; python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value']/1e6, 'M env-steps/s', d['ms_per_step']*1e3, 'us/step')"
