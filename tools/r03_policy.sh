#!/bin/bash
# Dev tool (GPU box): policy parity tests, stamps, the rollout-loop leg of bench.py.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03policy; mkdir -p $O
cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_policy.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python3 bench.py --no-cpu-baseline --with-policy > $O/bench_with_policy.json 2> $O/bench.err; echo "bench rc=$?"
python3 -c "import json; d=json.load(open('$O/bench_with_policy.json')); w=d['with_policy']; print(d['value'], json.dumps(w.get('kernels_us_events')), w.get('ms_per_step'), json.dumps(w.get('pair')), w.get('failed'))"
