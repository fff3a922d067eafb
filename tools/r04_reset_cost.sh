#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_reset_cost}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/reset_cost.py 600 $O/resets.txt > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python3 $R/tools/reset_cost.py --join $O | tee $O/reset_cost.txt; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
