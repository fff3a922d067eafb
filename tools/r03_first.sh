#!/bin/bash
# Dev tool (GPU box): round-3 opening check -- GPU tests, a marker + kernel trace, the default bench line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --marker-trace --output-format csv -d $O/marker -- python3 $R/tools/pmc_run.py 4096 6 markers > $O/marker.log 2>&1
echo "marker rc=$?"
cd $R && python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "bench rc=$?"; cat $O/bench_driver.json | cut -c1-600
