#!/usr/bin/env python3
"""Dev tool (GPU box): the wave-private scan kernel (rover_debug_set_scan_form 7) against the product scan kernel: same bits? how long?"""
import os, sys, time, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ter = T.make_procedural_terrain((2048, 2048), seed=1234, n_rocks=400); ter.make_spawns(2 * n, seed=41)
def make(form):
    cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
    env = RoverEnv(cfg, terrain=ter)
    fn = C.CDLL(env._lib._name).rover_debug_set_scan_form; fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(env._h, form) == 0
    env.reset()
    return env
a, b = make(0), make(7)
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(16, n, 2, device="cuda", generator=g) * 2 - 1
bad = 0
for k in range(40):
    oa = a.step(acts[k % 16])[0]["policy"]; ob = b.step(acts[k % 16])[0]["policy"]
    same = torch.equal(oa.view(torch.int32), ob.view(torch.int32))
    if not same:
        bad += 1
        if bad < 3:
            d = (oa.view(torch.int32) != ob.view(torch.int32)).nonzero()
            print("step", k, "differs at", d[:5].tolist(), oa[d[0,0], d[0,1]].item(), ob[d[0,0], d[0,1]].item())
print("steps with different observations:", bad, "of 40; log equal:", torch.equal(a._log, b._log))
for name, env in (("product", a), ("private", b)):
    for k in range(20): env.step(acts[k % 16])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(500): env.step(acts[k % 16])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
    x = y = 0.0
    for k in range(20):
        u, v = env.profile_step(acts[k % 16]); x += u; y += v
    print(f"{name}: {dt * 1e6:.1f} us per step; events (raw): step kernel {x / 20 * 1e3:.1f} us, scan kernel {y / 20 * 1e3:.1f} us")
