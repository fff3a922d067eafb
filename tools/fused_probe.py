#!/usr/bin/env python3
"""Dev tool (GPU box): one launch per step (rover_step_scan_kernel) against the two-launch path: same bits? how long?"""
import os, sys, time, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ABLTAG"):   # diagnostic build of tools/build_diag.py
    from isaac_rover_orbit_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{os.environ['ABLTAG']}.so")
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ter = T.make_procedural_terrain((2048, 2048), seed=1234, n_rocks=400); ter.make_spawns(2 * n, seed=41)
def make(fused):
    cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"; cfg.step_mapping = "group"
    env = RoverEnv(cfg, terrain=ter)
    fn = C.CDLL(env._lib._name).rover_debug_set_fused; fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(env._h, fused) == 0
    env.reset()
    return env
a, b = make(0), make(int(os.environ.get("FUSED_FORM", "1")))      # 1 = copy-wave form, 2 = single-tile form
print(a.kernel_names(), b.kernel_names())
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(16, n, 2, device="cuda", generator=g) * 2 - 1
S = a.get_state(); S[::37, 51] = torch.tensor([745], dtype=torch.int32).view(torch.float32).item()   # some envs time out soon: in-step resets
a.set_state(S); b.set_state(S)
bad = 0
for k in range(40):
    ra = a.step(acts[k % 16]); rb = b.step(acts[k % 16])
    same = torch.equal(ra[0]["policy"].view(torch.int32), rb[0]["policy"].view(torch.int32)) and torch.equal(ra[1], rb[1]) and torch.equal(a.episode_log_vector, b.episode_log_vector)
    if not same:
        bad += 1
        if bad < 3:
            d = (ra[0]["policy"].view(torch.int32) != rb[0]["policy"].view(torch.int32)).nonzero()
            print("step", k, "obs differ at", d[:5].tolist(), "log", a.episode_log_vector.tolist()[:14], b.episode_log_vector.tolist()[:14])
            envs = torch.unique(d[:, 0]); cols = torch.unique(d[:, 1])
            e0 = int(d[0, 0]); A = ra[0]["policy"][e0].cpu(); B = rb[0]["policy"][e0].cpu()
            for c in d[d[:, 0] == e0][:12, 1].tolist():
                where = (A == B[c]).nonzero().flatten().tolist()[:4]
                print(f"   env {e0} col {c} (ray {c - 4}): two-launch {A[c].item():.6f} one-launch {B[c].item():.6f}; the one-launch value sits in the two-launch row at cols {where}")
            print("   differing elements", d.shape[0], "envs", envs.numel(), "env % 4 histogram", torch.bincount(envs % 4, minlength=4).tolist(),
                  "cols min/max", cols.min().item(), cols.max().item(), "first cols", cols[:12].tolist(), "resets this step", int((ra[2] | ra[3]).sum()))
print("steps with different observations / rewards / log:", bad, "of 40; states equal:", torch.equal(a.get_state().view(torch.int32), b.get_state().view(torch.int32)))
for name, env in (("two launches", a), ("one launch", b)):
    for k in range(20): env.step(acts[k % 16])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(500): env.step(acts[k % 16])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
    x = y = 0.0
    for k in range(20):
        u, v = env.profile_step(acts[k % 16]); x += u; y += v
    print(f"{name}: {dt * 1e6:.1f} us per step = {n / dt / 1e6:.1f} M env-steps/s; events (raw): first kernel {x / 20 * 1e3:.1f} us, second {y / 20 * 1e3:.1f} us")
