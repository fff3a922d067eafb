#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel stats of the FrankaCubeLift-v0 step at 2048 envs.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03liftprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/lift_time.py 2048 > $O/stats.log 2>&1
echo "rc=$?"; cat $O/stats.log | tail -3
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-200 {} | head -8'
