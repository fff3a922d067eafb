#!/bin/bash
# Dev tool (GPU box): the round-2 final measurement set.  Each step writes under gpurun_out/r02f and the chain stops
# at the first failing GPU step.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_fetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_write.log 2>&1 && \
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_tcc.log 2>&1 && \
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/hbm_traffic.json "r02_f build (16 lanes per env with role-split row work, asm solver, Newton rsqrt; triangle scan, 1024-thread scan workgroups, two envs per round)" > $O/traffic.log 2>&1 && \
cp $O/hbm_traffic.json $R/profiles/hbm_traffic.json && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/stats_bench.log 2>&1 && \
cd $R && \
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err && \
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err && \
python3 bench.py --config 4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err && \
python3 bench.py --config 5 > $O/bench_c5.json 2> $O/bench_c5.err && \
python3 tools/n_sweep.py > $O/n_sweep.txt 2>&1
echo "rc=$?"
