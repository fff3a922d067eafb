#!/usr/bin/env python3
"""Dev tool (GPU box): no-wait timeline of the one-launch kernel's step wave 0 and copy wave 4 of every workgroup (build K1LITE):
cycles since the step wave's start, medians over the workgroups of one launch."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{os.environ.get('ABLTAG', 'K1LITE')}.so")
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ter = T.make_procedural_terrain((2048, 2048), seed=1234, n_rocks=400); ter.make_spawns(2 * n, seed=41)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
env = RoverEnv(cfg, terrain=ter); env.reset()
stamps = torch.zeros(n // 16, 64, dtype=torch.int64, device="cuda")
fn = env._lib.rover_debug_set_k1_stamps; fn.argtypes = [C.c_void_p]
assert fn(C.c_void_p(stamps.data_ptr())) == 0
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(40, n, 2, device="cuda", generator=g) * 2 - 1
for k in range(40): env.step(acts[k])
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
t0 = s[:, 0:1]
step = {0: "start", 1: "physics done (before A)", 2: "after A2 (reset decided, final windows written)", 11: "ray table requested", 12: "group_store issued",
        13: "force rows stored", 14: "mdp terms", 15: "rewards + reset", 16: "command", 3: "tail done (log, final stores issued)", 4: "ray table / stores retired",
        5: "after B", 6: "env 1 share cast", 7: "after C", 8: "env 2 cast", 9: "after D", 10: "end (env 3 share cast)"}
copy = {0: "before A (link work done)", 1: "after A", 2: "after A2", 3: "windows 0, 1 + ray table requested", 4: "... landed", 5: "env 0 cast",
        6: "... its stores retired", 7: "after B", 8: "window 2 requested (+ env 1 share cast)", 9: "... window 2 landed",
        10: "after C", 11: "window 3 requested", 12: "env 2 share cast", 13: "... window 3 landed", 14: "after D", 15: "end (env 3 share cast)"}
print(env.kernel_names()[0])
for name, off, labels in (("step wave", 0, step), ("copy wave", 32, copy)):
    print(name)
    prev = None
    for i, lab in labels.items():
        if not (s[:, off + i] > 0).all():      # a fine stamp (build K1LITEF only)
            continue
        d = s[:, off + i] - t0[:, 0]
        m = np.median(d)
        print(f"  {lab:50s} {m:8.0f}  (p90 {np.percentile(d, 90):8.0f})" + (f"   +{m - prev:6.0f}" if prev is not None else ""))
        prev = m
