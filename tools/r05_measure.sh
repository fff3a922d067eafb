#!/bin/bash
# Dev tool (GPU box): the round-5 measurement set (needs `python tools/build_diag.py K1STAMP K1LITE POLSTAMP LIFTSTAMP` first).  Every step
# writes under gpurun_out/r05m; the chain stops at the first failing GPU step.  $1: 1 = counters / traces, 2 = bench lines / stamps.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O
TAG="r05 final build (hipcc -mllvm -amdgpu-sched-strategy=max-ilp): cfg.mass_model = 1 (bogie-subtree weights), chassis split 3, 32 solver iterations; solver loop 44 instructions per iteration, eight iterations per trip, loop heads 32-byte aligned; bogie sin/cos once per substep, rare paths as not-taken uniform branches; three LDS corner reads per ray; scan phase: each wave of a pair owns a tile (copy wave envs 0, 2; step wave envs 1, 3), barriers L and A only, the reset decision and the final windows through one polled LDS word per env"
cd /tmp && export TMPDIR=/tmp
if [ "${1:-1}" = "1" ]; then
python3 -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 && \
export ROVER_ALSO_TWO_LAUNCH=1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_fetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_write.log 2>&1 && \
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- python3 $R/tools/pmc_run.py 4096 20 > $O/pmc_tcc.log 2>&1 && \
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/hbm_traffic.json "$TAG" > $O/traffic.log 2>&1 && \
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_tcc && \
unset ROVER_ALSO_TWO_LAUNCH && \
for model in "32 subtree_weights default" "16 lumped r04model"; do
  set -- $model; export QB_ITERS=$1 QB_MASS=$2; D=$O/pmc_$3; mkdir -p $D
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d $D/p1 -- python3 $R/tools/pmc_run.py 4096 20 > $D/p1.log 2>&1 && \
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $D/p2 -- python3 $R/tools/pmc_run.py 4096 20 > $D/p2.log 2>&1 && \
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $D/p3 -- python3 $R/tools/pmc_run.py 4096 20 > $D/p3.log 2>&1 || exit 1
  python3 $R/tools/pmc_summarise.py $D > $O/issue_counters_$3.txt 2>&1
  (cd $R/tools && python3 pmc_issue.py $D $O/issue_counters_$3.json "$TAG ($1 iterations, $2)") > $D/issue_json.log 2>&1
  rm -rf $D/p1 $D/p2 $D/p3
done
unset QB_ITERS QB_MASS
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extra > $O/stats_bench.log 2>&1 && \
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv && rm -rf $O/stats && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats16 -- python3 $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extra --solver-iterations 16 --mass-model lumped > $O/stats_bench_r04model.log 2>&1 && \
cp $(find $O/stats16 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_r04model.csv && rm -rf $O/stats16 && \
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
python3 bench.py --no-extra --no-cpu-baseline --solver-iterations 16 --mass-model lumped > $O/bench_r04model.json 2> $O/bench_r04.err && \
python3 bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --solver-iterations 16 > $O/bench_16it_driver_args.json 2> $O/bench_16d.err && \
python3 tools/pair_time.py > $O/pair_time.txt 2>&1 && \
FUSED=1 python3 tools/k1_stamps.py > $O/k1_stamps.txt 2>&1 && \
python3 tools/k1_lite.py > $O/k1_lite.txt 2>&1 && \
python3 tools/policy_stamps.py pair > $O/policy_stamps_pair.txt 2>&1 && \
python3 tools/policy_stamps.py > $O/policy_stamps.txt 2>&1 && \
python3 tools/lift_stamps.py 2048 > $O/lift_stamps.txt 2>&1 && \
python3 tools/host_path.py > $O/host_path.txt 2>&1 && \
python3 tools/n_sweep.py > $O/n_sweep.txt 2>&1
echo "half 2 rc=$?"
fi
ls $O
