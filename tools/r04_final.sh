#!/bin/bash
# Dev tool (GPU box): last check of a build -- the whole -m gpu suite, then the bench lines and the rocprofv3 kernel summary kept under profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_final}; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -2 $O/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
python3 bench.py > $O/bench.json 2> $O/bench.err && \
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver.err && \
python3 tools/k1_lite.py > $O/k1_lite.txt 2>&1 && \
cd /tmp && export TMPDIR=/tmp && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extra > $O/stats_bench.log 2>&1 && \
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv && rm -rf $O/stats
echo "rc=$?"
python3 -c "
import json
for f in ('bench','bench_driver_args'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value']/1e6, d['ms_per_step']*1e3, d['roofline']['frac'], d['roofline']['issue']['valu_busy_frac'])
"
grep rover_step_scan $O/kernel_stats.csv | cut -c1-40,200-
