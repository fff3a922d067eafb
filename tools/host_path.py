#!/usr/bin/env python3
"""Dev tool (GPU box): what env.step() costs the HOST per call, against the ~35 us kernel it must stay under (one launch per step:
a host path above the kernel time makes the rollout loop launch-bound).  Round-3 review: one 200-step window of tools/n_sweep.py
read 73 us per step at 4096 envs while the kernel took 44 us -- was that the host path?

  1. Python around the native call (argument checks, buffer rotation, counter mirror, attribute swaps): rover_step stubbed out
  2. the native call itself on an EMPTY queue (ctypes marshalling + hipLaunchKernelGGL): timed per call with a synchronisation
     between calls
  3. the steady loop (no synchronisation inside): us per step over windows of 200 steps, the first windows after construction
     reported one by one (clock ramp / first touch) beside the steady value
"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ter = T.make_procedural_terrain((2048, 2048)); ter.make_spawns(2 * n)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
env = RoverEnv(cfg, terrain=ter)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(16, n, 2, device="cuda", generator=g) * 2 - 1
# 3a. the first windows after construction
wins = []
for w in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(200): env.step(acts[k % 16])
    torch.cuda.synchronize(); wins.append((time.perf_counter() - t0) / 200 * 1e6)
print("us per step, consecutive 200-step windows right after construction + reset:", " ".join(f"{w:.1f}" for w in wins))
# 2. native call on an empty queue
real = env._lib.rover_step
tl = []
for k in range(300):
    torch.cuda.synchronize()
    a = acts[k % 16]
    t0 = time.perf_counter(); env.step(a); tl.append(time.perf_counter() - t0)
tl = sorted(tl)
print(f"env.step() on an empty queue (Python + ctypes + launch): median {tl[len(tl) // 2] * 1e6:.1f} us, p90 {tl[int(len(tl) * 0.9)] * 1e6:.1f} us")
# 1. Python alone: the native entry replaced by a no-op
class _Stub:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, k): return getattr(self._lib, k)
    def rover_step(self, *a): return 0
env._lib = _Stub(real.__self__ if hasattr(real, "__self__") else env._lib)
t0 = time.perf_counter()
for k in range(20000): env.step(acts[k % 16])
py = (time.perf_counter() - t0) / 20000 * 1e6
print(f"Python around the native call (rover_step stubbed): {py:.2f} us per step")
