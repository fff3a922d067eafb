#!/bin/bash
# Dev tool (GPU box): issue / LDS counters of the step-path kernels on the FINAL build, for both ray -> thread mappings of the
# scan kernel (lines = product; blocks = 8 x 8 rays per wave).  Separate --pmc passes, kernel trace only.
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for form in 4 3; do
  O=$R/gpurun_out/k2pmc_form$form; mkdir -p $O; export ROVER_SCAN_FORM=$form
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d $O/p1 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p1.log 2>&1 && \
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p2.log 2>&1 && \
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p3 -- python3 $R/tools/pmc_run.py 4096 20 > $O/p3.log 2>&1
  echo "form $form rc=$?"
  python3 $R/tools/pmc_summarise.py $O > $O/summary.txt 2>&1
  rm -rf $O/p1 $O/p2 $O/p3
done
grep -A 22 "rover_scan_step" $R/gpurun_out/k2pmc_form4/summary.txt | head -24; grep -A 22 "rover_scan_step" $R/gpurun_out/k2pmc_form3/summary.txt | grep "BANK_CONFLICT\|LDS_IDX_ACTIVE\|WAIT_ANY"
