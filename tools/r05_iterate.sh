#!/bin/bash
# Dev tool (GPU box), round 5: one iteration on the one-launch kernel -- the rover path's parity files, the issue / LDS counters
# (three --pmc passes) under the round-4 model (16 iterations, lumped mass: comparable with profiles/r04_i_*) and under the product
# default (32 iterations, subtree weights), then us per step of both.   $1 = output tag, TAGS = extra tools/build_diag.py variants
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_it}; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_user_terms.py tests/test_gpu_golden.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
for model in "16 lumped r04model" "32 subtree_weights default"; do
  set -- $model; export QB_ITERS=$1 QB_MASS=$2; D=$O/pmc_$3; mkdir -p $D
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d $D/p1 -- python3 $R/tools/pmc_run.py 4096 20 > $D/p1.log 2>&1 && \
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $D/p2 -- python3 $R/tools/pmc_run.py 4096 20 > $D/p2.log 2>&1 || exit 1
  python3 $R/tools/pmc_summarise.py $D > $O/issue_counters_$3.txt 2>&1
  rm -rf $D
  timeout -k 10 120 python3 $R/tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
done
unset QB_ITERS QB_MASS
for t in ${TAGS}; do
  ABLTAG=$t timeout -k 10 120 python3 $R/tools/quick_bench.py 4096 2000 >> $O/quick.txt 2>&1 || exit 1
done
grep -A16 "rover_step_scan_kernel" $O/issue_counters_r04model.txt | head -20
grep "us per step" $O/quick.txt
