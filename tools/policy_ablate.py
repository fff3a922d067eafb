#!/usr/bin/env python3
"""Dev tool (GPU box): GPU time of the policy kernel in diagnostic builds (build/abl/librover_ablPOL*.so)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_policy_weights, synthetic_obs
from isaac_rover_orbit_amd import _lib
tag = sys.argv[1]
if tag != "base":
    _lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{tag}.so")
from isaac_rover_orbit_amd.policy import RoverNet
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ws, bs = random_policy_weights(seed=0, scale=3.0)
net = RoverNet(ws, bs)
obs = torch.from_numpy(synthetic_obs(n)).cuda(); out = torch.empty(n, 2, device="cuda")
for _ in range(20): net.forward(obs, out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(200): net.forward(obs, out)
e1.record(); torch.cuda.synchronize()
print(tag, n, "policy kernel %.1f us per launch (back-to-back)" % (e0.elapsed_time(e1) / 200 * 1e3))
