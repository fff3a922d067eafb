#!/usr/bin/env python3
"""Dev tool (GPU box): the actor + critic pair kernel (rover_policy_forward_pair) back to back on N = 4096 observation rows:
us per launch, beside one network alone.  Run under rocprofv3 --kernel-trace --stats for the profile."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_policy_weights, synthetic_obs
if os.environ.get("POLTAG"):   # a tools/build_diag.py variant of the policy kernels
    from isaac_rover_orbit_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{os.environ['POLTAG']}.so")
from isaac_rover_orbit_amd.policy import RoverNet, forward_pair
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ws, bs = random_policy_weights(seed=0, scale=3.0)
actor = RoverNet(ws, bs)
wc = [w.copy() for w in ws]; bc = [b.copy() for b in bs]
wc[5], bc[5] = wc[5][:1].copy(), bc[5][:1].copy()
critic = RoverNet(wc, bc, final_act="none")
obs = torch.from_numpy(synthetic_obs(n)).cuda()
def timed(fn, reps=500):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
a1 = actor(obs); v1 = critic(obs); a2, v2 = forward_pair(actor, critic, obs)
print("pair == two forwards:", torch.equal(a1.view(torch.int32), a2.view(torch.int32)) and torch.equal(v1.view(torch.int32), v2.view(torch.int32)))
print(f"pair {timed(lambda: forward_pair(actor, critic, obs)):.2f} us   actor alone {timed(lambda: actor(obs)):.2f} us   critic alone {timed(lambda: critic(obs)):.2f} us")
