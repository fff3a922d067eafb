#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for f in 5 6 5 6; do ROVER_SCAN_FORM=$f python3 bench.py --no-cpu-baseline --with-policy 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('form $f', round(d['value']/1e6,2), {k: round(v,1) for k,v in d['with_policy']['kernels_us_events'].items()}, round(d['with_policy']['ms_per_step']*1e3,1))"; done
