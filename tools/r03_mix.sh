#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03mix; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_trainer_replay.py tests/test_gpu_boundary.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:(round(v['ms']*1e3,2)) for k,v in d['roofline']['kernels'].items()})"
python3 tools/n_sweep.py 4096:group 16384:group 65536:group 65536:lane 131072:group 131072:lane 2>&1 | grep num_envs
