#!/usr/bin/env python3
"""Dev tool (GPU box): env-steps/s at N = 4096 versus the contact solver's iteration count (cfg.solver_iterations)."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = 4096
ter = T.make_procedural_terrain((2048, 2048)); ter.make_spawns(2 * n)
out = []
for iters in (4, 8, 12, 16, 24, 32):
    cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"; cfg.solver_iterations = iters
    env = RoverEnv(cfg, terrain=ter); env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.rand(16, n, 2, device="cuda", generator=g) * 2 - 1
    for k in range(20): env.step(acts[k % 16])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(300): env.step(acts[k % 16])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out.append({"solver_iterations": iters, "env_steps_per_s": n * 300 / dt, "us_per_step": dt / 300 * 1e6})
    print(json.dumps(out[-1]))
    env.close()
