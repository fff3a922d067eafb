#!/usr/bin/env python3
"""Dev tool (GPU box): what the two bracket synchronisations of a timed region cost -- us per env.step() for regions of 20 / 100 / 1000
steps, with the default wait policy and with hipSetDeviceFlags(hipDeviceScheduleSpin).  usage: sync_probe.py [default|spin]"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "spin":
    hip = C.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(C.c_uint(1)))   # hipDeviceScheduleSpin = 0x1
import torch
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = 4096
ter = T.make_procedural_terrain((2048, 2048), seed=1234, n_rocks=400); ter.make_spawns(2 * n, seed=41)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
env = RoverEnv(cfg, terrain=ter); env.reset()
acts = torch.rand(256, n, 2, device="cuda") * 2 - 1
for k in range(400): env.step(acts[k % 256])
for K in (20, 100, 1000):
    res = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K): env.step(acts[k % 256])
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / K * 1e6)
    print(mode, "K", K, "us per step:", " ".join(f"{r:.2f}" for r in res))

# where the fixed cost sits: host-side launch loop, GPU-side span (events), final synchronise
for K in (20,):
    for rep in range(4):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for k in range(K): env.step(acts[k % 256])
        t1 = time.perf_counter(); e1.record()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(mode, f"K {K}: launch loop {1e6 * (t1 - t0):.0f} us, + synchronise {1e6 * (t2 - t1):.0f} us = {1e6 * (t2 - t0):.0f} us; GPU span between the events {1e3 * e0.elapsed_time(e1):.0f} us "
              f"({1e3 * e0.elapsed_time(e1) / K:.2f} per step)")
