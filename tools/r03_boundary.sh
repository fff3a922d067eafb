#!/bin/bash
# Dev tool (GPU box): what a kernel pays for following a dependent kernel (tools/ubench/boundary_ubench.hip).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/boundary; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/boundary_ubench $R/tools/ubench/boundary_ubench.hip 2> $O/build.log && \
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- /tmp/boundary_ubench > $O/run.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/summary.txt
import csv,sys
for r in sorted(list(csv.reader(open(sys.argv[1])))[1:], key=lambda r: r[0]):
    print(r[0][:70], "calls", r[1], "avg ns", r[3], "min", r[5], "max", r[6])
PY
grep follower $O/run.log | tee -a $O/summary.txt; rm -rf $O/prof
