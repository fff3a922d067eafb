#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for t in "" NTSTORE "" NTSTORE; do
python3 - "$t" <<'PY'
import os, sys, json, subprocess
tag=sys.argv[1]
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
if tag:
    from isaac_rover_orbit_amd import _lib
    _lib.LIB_PATH = os.path.join(os.environ["GRAFT_REPO_ROOT"], "build", "abl", f"librover_abl{tag}.so")
sys.argv=["bench.py","--no-cpu-baseline","--with-policy"]
import io, contextlib
buf=io.StringIO()
with contextlib.redirect_stdout(buf):
    import runpy; runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"],"bench.py"), run_name="__main__")
d=json.loads(buf.getvalue().strip().splitlines()[-1])
w=d["with_policy"]
print(tag or "product", round(d["value"]/1e6,2), round(d["ms_per_step"]*1e3,2), {k:round(v["ms"]*1e3,2) for k,v in d["roofline"]["kernels"].items()}, {k:round(v,1) for k,v in w["kernels_us_events"].items()}, round(w["ms_per_step"]*1e3,1), round(w["pair"]["ms_per_step"]*1e3,1))
PY
done
