#!/usr/bin/env python3
"""Dev tool (GPU box): A/B kernel timings of alternative builds of librover_hip.so (interleaved rounds, one process)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import _lib, terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = int(os.environ.get("NENV", "4096"))
ter = T.make_procedural_terrain((2048, 2048)); ter.make_spawns(2 * n)
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(64, n, 2, device="cuda", generator=g) * 2 - 1
envs = {}
for tag in sys.argv[1:]:
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "isaac_rover_orbit_amd", "librover_hip.so") if tag == "base" else os.path.join(ROOT, "build", "abl", f"librover_abl{tag}.so")
    cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
    env = RoverEnv(cfg, terrain=ter); env.reset()
    for k in range(20): env.step(acts[k])
    envs[tag] = env
res = {t: [] for t in envs}
for rnd in range(6):
    for tag, env in envs.items():
        a = b = 0.0
        for k in range(50):
            x, y = env.profile_step(acts[k % 64]); a += x; b += y
        res[tag].append((a / 50 * 1e3, b / 50 * 1e3))
for tag, v in res.items():
    v = np.array(v)
    print(f"{tag:10s} step kernel median {np.median(v[:,0]):6.1f} us (min {v[:,0].min():6.1f})   scan kernel median {np.median(v[:,1]):6.1f} us (min {v[:,1].min():6.1f})")
