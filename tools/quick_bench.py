#!/usr/bin/env python3
"""Dev tool (GPU box): us per env.step() at N envs for one build of the library (ABLTAG = a tools/build_diag.py variant).
usage: [ABLTAG=tag] [QB_ITERS=n] [QB_MASS=lumped|subtree_weights] quick_bench.py [n] [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ABLTAG"):
    from isaac_rover_orbit_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{os.environ['ABLTAG']}.so")
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ter = T.make_procedural_terrain((2048, 2048), seed=1234, n_rocks=400); ter.make_spawns(2 * n, seed=41)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
if os.environ.get("QB_ITERS"): cfg.solver_iterations = int(os.environ["QB_ITERS"])
if os.environ.get("QB_MASS"): cfg.mass_model = os.environ["QB_MASS"]
def cluster_by_xcd(env, n):
    """QB_CLUSTER=1 (experiment): permute the envs' states after the reset so that the envs of the workgroups one XCD runs (workgroup
    b = envs 16 b .. 16 b + 15 runs on XCD b mod 8) stand in one x-strip of the terrain -- what the L2 fetch would be if env -> XCD
    followed the rovers' positions."""
    S = env.get_state()
    order = torch.argsort(S[:, 0]).view(8, -1)                       # eight strips by x, n / 8 envs each
    xcd = (torch.arange(n, device=S.device) // 16) % 8
    slot = torch.zeros(n, dtype=torch.long, device=S.device)
    for k in range(8):
        slot[xcd == k] = order[k]
    env.set_state(S[slot].contiguous())


env = RoverEnv(cfg, terrain=ter)
env.reset()
if os.environ.get("QB_CLUSTER"): cluster_by_xcd(env, n)
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(256, n, 2, device="cuda", generator=g) * 2 - 1
res = []
for rep in range(3):
    for k in range(100): env.step(acts[k % 256])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps): env.step(acts[k % 256])
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / steps * 1e6)
print(os.environ.get("ABLTAG", "product") + (" clustered" if os.environ.get("QB_CLUSTER") else ""), f"iters={cfg.solver_iterations} mass={cfg.mass_model}", env.kernel_names()[0], "us per step:", " ".join(f"{r:.2f}" for r in res), f"-> {n / min(res):.1f} M env-steps/s")
