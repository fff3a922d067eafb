import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = 4096
ter = T.make_procedural_terrain((2048, 2048), seed=1234, n_rocks=400); ter.make_spawns(2 * n, seed=41)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
env = RoverEnv(cfg, terrain=ter); env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(2048, n, 2, device="cuda", generator=g) * 2 - 1
k = 0
for w in range(16):
    resets = torch.zeros((), device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200):
        o, r, te, tr, info = env.step(acts[k % 2048]); k += 1
        resets += (te | tr).sum()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200 * 1e6
    print(f"steps {w*200:5d}..{w*200+199:5d}: {dt:6.2f} us per step (incl. the reset count's own kernels), resets per step {float(resets) / 200:.2f}", flush=True)
# and the same window sizes WITHOUT the extra kernels
for w in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200): env.step(acts[k % 2048]); k += 1
    torch.cuda.synchronize(); print(f"steps {k-200:5d}..{k-1:5d}: {(time.perf_counter() - t0) / 200 * 1e6:6.2f} us per step", flush=True)
