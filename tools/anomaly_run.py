#!/usr/bin/env python3
"""Dev tool (GPU box): reset + a few env steps at a given N and step-kernel mapping (for rocprofv3 passes on the scan kernel)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n, mapping, steps = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 12
ter = T.make_procedural_terrain((2048, 2048)); ter.make_spawns(2 * n)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"; cfg.step_mapping = mapping
if os.environ.get("NO_FORCES"):
    cfg.record_contact_forces = False
env = RoverEnv(cfg, terrain=ter)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(4, n, 2, device="cuda", generator=g) * 2 - 1
for k in range(steps):
    env.step(acts[k % 4])
torch.cuda.synchronize()
env.close()
