#!/bin/bash
# Dev tool (GPU box): the round-4 measurement set (needs `python tools/build_diag.py K1STAMP POLSTAMP LIFTSTAMP` first).  Every step
# writes under gpurun_out/r04i; the chain stops at the first failing GPU step.  $1: 1 = first half, 2 = second half (two gpurun calls).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O
TAG="r04_i build (one-launch step kernel: reset decided after the physics, final windows to the copy wave, env 0 cast under the manager tail, six barriers; chassis / manager words stored one row per lane through 32-bit offsets; reset log row one value per lane; pipelined packed ray cast, 16-byte row stores, lean LDS-DMA loop; contact report with horizontal components; policy pair kernel layer by layer)"
cd /tmp && export TMPDIR=/tmp
export ROVER_ALSO_TWO_LAUNCH=1   # pmc_run.py: the two-launch path behind the product run (traffic of both forms)
if [ "${1:-1}" = "1" ]; then
python3 -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_fetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_write.log 2>&1 && \
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_tcc.log 2>&1 && \
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/hbm_traffic.json "$TAG" > $O/traffic.log 2>&1 && \
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_tcc && \
unset ROVER_ALSO_TWO_LAUNCH && \
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d $O/p1 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p1.log 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p2.log 2>&1 && \
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p3 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p3.log 2>&1 && \
python3 $R/tools/pmc_summarise.py $O > $O/issue_counters.txt 2>&1 && (cd $R/tools && python3 pmc_issue.py $O $O/issue_counters.json "$TAG") > $O/issue_json.log 2>&1 && \
rm -rf $O/p1 $O/p2 $O/p3 && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extra > $O/stats_bench.log 2>&1 && \
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv && rm -rf $O/stats && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pol -- python3 $R/tools/pair_time.py > $O/pair_time_under_rocprof.txt 2>&1 && \
cp $(find $O/stats_pol -name "*kernel_stats.csv" | head -1) $O/policy_kernel_stats.csv && rm -rf $O/stats_pol && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lift -- python3 $R/tools/lift_time.py 2048 > $O/stats_lift.log 2>&1 && \
cp $(find $O/stats_lift -name "*kernel_stats.csv" | head -1) $O/lift_kernel_stats.csv && rm -rf $O/stats_lift
echo "half 1 rc=$?"
else
cd $R && \
python3 bench.py > $O/bench.json 2> $O/bench.err && \
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver.err && \
python3 bench.py --config 4 > $O/bench_config4.json 2> $O/bench_c4.err && \
python3 bench.py --config 5 > $O/bench_config5.json 2> $O/bench_c5.err && \
python3 tools/pair_time.py > $O/pair_time.txt 2>&1 && \
FUSED=1 python3 tools/k1_stamps.py > $O/k1_stamps.txt 2>&1 && \
python3 tools/k1_lite.py > $O/k1_lite.txt 2>&1 && \
python3 tools/policy_stamps.py pair > $O/policy_stamps_pair.txt 2>&1 && \
python3 tools/policy_stamps.py > $O/policy_stamps.txt 2>&1 && \
python3 tools/lift_stamps.py 2048 > $O/lift_stamps.txt 2>&1 && \
python3 tools/host_path.py > $O/host_path.txt 2>&1 && \
bash tools/r04_reset_cost.sh r04i_reset > /dev/null 2>&1 && cp gpurun_out/r04i_reset/reset_cost.txt $O/reset_cost.txt && \
python3 tools/n_sweep.py > $O/n_sweep.txt 2>&1
echo "half 2 rc=$?"
fi
