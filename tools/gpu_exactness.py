#!/usr/bin/env python3
"""Dev diagnostic (GPU box): closed-loop HIP-vs-oracle divergence over a rollout; prints max |diff| per step."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import oracle_config_from, oracle_terrain, small_procedural  # noqa: E402
from isaac_rover_orbit_amd.cfg import RoverEnvCfg  # noqa: E402
from isaac_rover_orbit_amd.envs import RoverEnv  # noqa: E402
from oracle import rover_oracle as ro  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ter = small_procedural()
ter.make_spawns(2 * n)
cfg = RoverEnvCfg()
cfg.scene.num_envs = n
cfg.terrain.kind = "custom"
env = RoverEnv(cfg, terrain=ter)
env.reset()
ocfg, oter = oracle_config_from(ro, env._native_cfg), oracle_terrain(ro, ter)
S = env.get_state().cpu().numpy().copy()
So = ro.new_state(n)
ro.reset_all(ocfg, oter, So)
print("reset: state bit-exact:", np.array_equal(S.view(np.int32), So.view(np.int32)))
rng = np.random.RandomState(0)
worst = 0.0
for k in range(steps):
    a = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    o, r, t, u, info = env.step(torch.from_numpy(a).cuda())
    oo, r_o, t_o, u_o, f_o, l_o = ro.step(ocfg, oter, So, a)
    o = o["policy"].cpu().numpy()
    S = env.get_state().cpu().numpy()
    d_obs = np.abs(np.where(np.isfinite(o), o, 0) - np.where(np.isfinite(oo), oo, 0)).max()
    d_state = np.abs(S - So)[:, :51].max()
    flips = int((t.cpu().numpy().astype(np.uint8) != t_o).sum() + (u.cpu().numpy().astype(np.uint8) != u_o).sum())
    exact = np.array_equal(S.view(np.int32), So.view(np.int32)) and np.array_equal(o.view(np.int32), oo.view(np.int32))
    worst = max(worst, d_obs, d_state)
    if k % 10 == 0 or not exact:
        print(f"step {k}: bit-exact={exact} max|dobs|={d_obs:.3e} max|dstate|={d_state:.3e} flips={flips} "
              f"log_diff={np.abs(env._log.cpu().numpy()[:14] - l_o[:14]).max():.3e}")
print("worst", worst)
