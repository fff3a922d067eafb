#!/bin/bash
# Dev tool (GPU box): counters of the scan kernel at N = 65536 behind the group-mapped and behind the lane-mapped step kernel.
# GRBM_GUI_ACTIVE / duration = the clock the kernel ran at.  At most four TCC counters per pass (more: "exceeds the capabilities").
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
N=${1:-65536}
for m in lane group; do
  O=$R/gpurun_out/anomaly_$m; rm -rf $O; mkdir -p $O
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p0 -- python3 $R/tools/anomaly_run.py $N $m > $O/p0.log 2>&1 && \
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $O/p1 -- python3 $R/tools/anomaly_run.py $N $m > $O/p1.log 2>&1 && \
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum --output-format csv -d $O/p2 -- python3 $R/tools/anomaly_run.py $N $m > $O/p2.log 2>&1 && \
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $O/p3 -- python3 $R/tools/anomaly_run.py $N $m > $O/p3.log 2>&1
  echo "$m rc=$?"
  python3 $R/tools/pmc_summarise.py $O > $O/summary.txt 2>&1
  cp $(find $O/p0 -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
  rm -rf $O/p0 $O/p1 $O/p2 $O/p3
done
for m in group lane; do echo "== $m"; grep -A 14 "rover_scan_step\|rover_step_kernel" $R/gpurun_out/anomaly_$m/summary.txt | head -40; python3 -c "
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:4]: print(r[0][:60], r[1], r[3])" $R/gpurun_out/anomaly_$m/kernel_stats.csv; done
