#!/bin/bash
# Dev tool (GPU box): issue / LDS counters of the two step-path kernels (separate --pmc passes, kernel trace only).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/k2pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d $O/p1 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p1.log 2>&1 && \
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p2.log 2>&1 && \
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p3 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p3.log 2>&1
echo rc=$?
python3 $R/tools/pmc_summarise.py $O > $O/summary.txt 2>&1; grep -A 24 "rover_scan_step\|rover_step_kernel_group" $O/summary.txt
