#!/bin/bash
# Dev tool (GPU box): what the driver runs at round end -- the whole -m gpu suite, smoke(), the default bench line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03reh; mkdir -p $O
cd $R && ( time timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
( time python3 bench.py --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"; cut -c1-300 $O/bench_driver.json; tail -3 $O/bench_driver.err
