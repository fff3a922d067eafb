#!/usr/bin/env python3
"""Dev experiment (GPU box): does a region-sorted workgroup order speed up the scan kernel?"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import _lib, terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = 4096
_lib.LIB_PATH = os.path.join(ROOT, "build", "abl", "librover_ablORDER.so")
_lib.EXPORTS.append("rover_debug_set_scan_slot")
ter = T.make_procedural_terrain((2048, 2048)); ter.make_spawns(2 * n)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
env = RoverEnv(cfg, terrain=ter); env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(64, n, 2, device="cuda", generator=g) * 2 - 1
fn = env._lib.rover_debug_set_scan_slot
fn.argtypes = [C.c_void_p, C.c_void_p]
def timeit(tag):
    ms1 = ms2 = 0
    for k in range(40):
        a, b = env.profile_step(acts[k % 64]); ms1 += a; ms2 += b
    print(f"{tag}: step kernel {ms1/40*1e3:.1f} us, scan kernel {ms2/40*1e3:.1f} us")
timeit("identity order")
for bins in (4, 8, 16):
    pos = env.state[0:2].t().cpu().numpy()
    rx = np.clip((pos[:, 0] / 102.4 * bins).astype(int), 0, bins - 1)
    ry = np.clip((pos[:, 1] / 102.4 * bins).astype(int), 0, bins - 1)
    key = ry * bins + np.where(ry % 2 == 1, bins - 1 - rx, rx)
    order = np.argsort(key, kind="stable")           # slot -> env
    slot = np.empty(n, np.int32); slot[order] = np.arange(n, dtype=np.int32)
    slot_dev = torch.from_numpy(slot).cuda()
    fn(env._h, C.c_void_p(slot_dev.data_ptr()))
    timeit(f"{bins}x{bins} region-sorted order (fresh)")
    timeit(f"{bins}x{bins} region-sorted order (40 steps stale)")
