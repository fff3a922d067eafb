#!/usr/bin/env python3
"""Dev tool (GPU box): us per FrankaCubeLiftEnv.step() at num_envs = 2048; ABLTAG=<variant> loads build/abl/librover_abl<variant>.so."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ABLTAG"):
    from isaac_rover_orbit_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{os.environ['ABLTAG']}.so")
from isaac_rover_orbit_amd.envs import FrankaCubeLiftEnv, LiftEnvCfg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = LiftEnvCfg(); cfg.scene.num_envs = n
env = FrankaCubeLiftEnv(cfg); env.reset()
if os.environ.get("LIFT_LANES"):
    assert env._lib.rover_lift_debug_set_lanes(env._h, int(os.environ["LIFT_LANES"])) == 0
if os.environ.get("LIFT_PIPE"):
    assert env._lib.rover_lift_debug_set_pipeline(env._h, int(os.environ["LIFT_PIPE"])) == 0
g = torch.Generator(device=env.device).manual_seed(0)
acts = torch.rand(64, n, 8, device=env.device, generator=g) * 2 - 1
for k in range(50): env.step(acts[k % 64])
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(500): env.step(acts[k % 64])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
print(f"{os.environ.get('ABLTAG', 'product')} {env.kernel_name()} n={n}: {dt * 1e6:.1f} us per step, {n / dt / 1e6:.2f} M env-steps/s")
