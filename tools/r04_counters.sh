#!/bin/bash
# Dev tool (GPU box): issue / wait / LDS counters of the ONE-LAUNCH step kernel (rover_step_scan_kernel<true>) at N = 4096, three
# separate --pmc passes with the kernel trace only, then the s_memtime stamps of the diagnostic build.  Output: gpurun_out/$TAG/
R=$GRAFT_REPO_ROOT; TAG=${1:-r04_counters}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d $O/p1 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p1.log 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p2.log 2>&1 && \
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/p3 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p3.log 2>&1
echo "pmc rc=$?"
python3 $R/tools/pmc_summarise.py $O > $O/issue_counters.txt 2>&1
(cd $R/tools && python3 pmc_issue.py $O $O/issue_counters.json "${2:-}") > $O/issue_json.log 2>&1
rm -rf $O/p1 $O/p2 $O/p3
cd $R && python3 tools/k1_stamps.py > $O/k1_stamps.txt 2>&1; echo "stamps rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/stats_bench.log 2>&1 && \
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/stats
python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/issue_counters.txt; tail -2 $O/k1_stamps.txt; cat $O/bench.json
