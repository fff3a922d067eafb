#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fusedprof; rm -rf $O; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $O/run.log 2>&1
python3 - $(find $O/p -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:5]: print(r[0][:80], r[1], r[3], r[5], r[6])
PY
python3 - $(find $O/p -name "*kernel_trace.csv" | head -1) <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "rover_step_scan_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
gaps=[int(b["Start_Timestamp"])-int(a["End_Timestamp"]) for a,b in zip(rows[100:300],rows[101:301])]
durs=[int(a["End_Timestamp"])-int(a["Start_Timestamp"]) for a in rows[100:300]]
import statistics as st
print("fused kernel: median duration", st.median(durs), "ns; median gap to the next launch", st.median(gaps), "ns; period", st.median(durs)+st.median(gaps))
PY
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/run.log | head -2; rm -rf $O/p
