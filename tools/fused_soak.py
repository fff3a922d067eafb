#!/usr/bin/env python3
"""Dev tool (GPU box): long closed run of the one-launch form against the two-launch form (same seeds, same actions): every
K-th step the observations, rewards, flags, log vector and state words must be the same bits."""
import os, sys, time, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ter = T.make_procedural_terrain((2048, 2048), seed=1234, n_rocks=400); ter.make_spawns(2 * n, seed=41)
def make(mode):
    cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"; cfg.step_mapping = "group"; cfg.log_reduction = mode
    env = RoverEnv(cfg, terrain=ter); env.reset(); return env
a, b = make("every_step"), make("on_demand")
ff0 = C.CDLL(a._lib._name).rover_debug_set_fused; ff0.argtypes = [C.c_void_p, C.c_int]
assert ff0(a._h, 0) == 0              # a: two launches per step
if os.environ.get("FUSED_FORM"):      # 1 = copy-wave form, 2 = single-tile form, whatever the batch size
    ff = C.CDLL(b._lib._name).rover_debug_set_fused; ff.argtypes = [C.c_void_p, C.c_int]
    assert ff(b._h, int(os.environ["FUSED_FORM"])) == 0
print(a.kernel_names(), b.kernel_names(), flush=True)
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(64, n, 2, device="cuda", generator=g) * 2 - 1
bad = resets = 0
t0 = time.time()
for k in range(steps):
    ra = a.step(acts[k % 64]); rb = b.step(acts[k % 64])
    if k % 97 == 0 or k == steps - 1:
        same = (torch.equal(ra[0]["policy"].view(torch.int32), rb[0]["policy"].view(torch.int32)) and torch.equal(ra[1], rb[1])
                and torch.equal(ra[2], rb[2]) and torch.equal(ra[3], rb[3]) and torch.equal(a.episode_log_vector, b.episode_log_vector)
                and torch.equal(a.state.view(torch.int32), b.state.view(torch.int32)))
        bad += 0 if same else 1
        resets += int(ra[2].sum() + ra[3].sum())
    if k % 5000 == 0:
        print("step", k, "mismatching checks so far:", bad, f"{time.time() - t0:.0f} s", flush=True)
print(f"{steps} steps x {n} envs: {bad} mismatching checks; resets seen in the checked steps: {resets}")
sys.exit(1 if bad else 0)
