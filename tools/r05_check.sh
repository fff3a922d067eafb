#!/bin/bash
# Dev tool (GPU box), round 5: the whole -m gpu selection, then one default bench.py run (JSON line + stderr kept)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_c}; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.load(open("$O/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "issue", d["roofline"].get("issue"), "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_build"])
print("cpu", d.get("cpu_baseline"))
e=d["extra"]
for k in ("config4","stream_observations","log_every_step","solver_iterations_16","round4_model"):
    v=e.get(k,{}); print(k, v.get("value"), v.get("ms_per_step"), v.get("kernel"), v.get("second_kernel"), v.get("failed"))
r=e["rollout_loop"]
print("rollout", r.get("ms_per_step"), r.get("pair",{}).get("ms_per_step"), r.get("pair_with_device_log",{}).get("ms_per_step"), r.get("trainer_loop",{}).get("ms_per_step"), r.get("trainer_loop_device_log",{}).get("ms_per_step"), r.get("failed"))
print("config5", e["config5"].get("value"), e["config5"].get("ms_per_step"))
PY
