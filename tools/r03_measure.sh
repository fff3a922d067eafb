#!/bin/bash
# Dev tool (GPU box): the round-3 final measurement set (needs `python tools/build_diag.py LIFTSTAMP POLSTAMP K1STAMP` first).  Every step writes under gpurun_out/r03f; the chain stops at the first failing GPU step.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O
TAG="r03_f build (one launch per step: step + scan kernel with copy waves, log on demand; 13-body contact report; lift: 8 lanes per env, two pipelined waves, log reduced on demand; policy: reference-architecture kernel)"
cd /tmp && export TMPDIR=/tmp
export ROVER_ALSO_TWO_LAUNCH=1   # pmc_run.py: the two-launch path behind the product run (traffic of both forms)
python3 -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_fetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_write.log 2>&1 && \
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_tcc.log 2>&1 && \
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/hbm_traffic.json "$TAG" > $O/traffic.log 2>&1 && \
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_tcc && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/stats_bench.log 2>&1 && \
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv && rm -rf $O/stats && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pol -- python3 $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --with-policy --profile-steps 20 > $O/stats_pol.log 2>&1 && \
cp $(find $O/stats_pol -name "*kernel_stats.csv" | head -1) $O/policy_kernel_stats.csv && rm -rf $O/stats_pol && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lift -- python3 $R/tools/lift_time.py 2048 > $O/stats_lift.log 2>&1 && \
cp $(find $O/stats_lift -name "*kernel_stats.csv" | head -1) $O/lift_kernel_stats.csv && rm -rf $O/stats_lift && \
cd $R && \
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err && \
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err && \
python3 bench.py --config 4 > $O/bench_c4.json 2> $O/bench_c4.err && \
python3 bench.py --config 5 > $O/bench_c5.json 2> $O/bench_c5.err && \
python3 bench.py --no-cpu-baseline --with-policy > $O/bench_with_policy.json 2> $O/bench_wp.err && \
python3 tools/lift_stamps.py 2048 > $O/lift_stamps.txt 2>&1 && \
LIFT_PIPE=0 python3 tools/lift_stamps.py 2048 > $O/lift_stamps_single_wave.txt 2>&1 && \
python3 tools/policy_stamps.py > $O/policy_stamps.txt 2>&1 && \
python3 tools/k1_stamps.py > $O/k1_stamps.txt 2>&1 && \
( python3 tools/lift_time.py 2048; LIFT_PIPE=0 python3 tools/lift_time.py 2048; LIFT_LANES=16 python3 tools/lift_time.py 2048; python3 tools/lift_time.py 4096; python3 tools/lift_time.py 8192; LIFT_PIPE=1 python3 tools/lift_time.py 8192 ) > $O/lift_time.txt 2>&1 && \
python3 tools/n_sweep.py > $O/n_sweep.txt 2>&1
echo "rc=$?"
