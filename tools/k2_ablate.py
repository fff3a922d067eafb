#!/usr/bin/env python3
"""Dev tool (GPU box): time rover_height_scan for ablated builds of the scan kernel (RV_K2_ABLATE bits)."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import _lib, terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ter = T.make_procedural_terrain((2048, 2048)); ter.make_spawns(2 * n)
for tag in sys.argv[2:]:
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{tag}.so")
    cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
    env = RoverEnv(cfg, terrain=ter); env.reset()
    scan = torch.empty(n, 961, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(20): env._lib.rover_height_scan(env._h, C.c_void_p(scan.data_ptr()), st)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): env._lib.rover_height_scan(env._h, C.c_void_p(scan.data_ptr()), st)
    e1.record(); torch.cuda.synchronize()
    print(f"ablate={tag}: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per scan launch (N={n})")
    env.close()
