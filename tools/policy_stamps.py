#!/usr/bin/env python3
"""Dev tool (GPU box): per-phase s_memtime timeline of the policy kernel (diagnostic build POLSTAMP)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import random_policy_weights, synthetic_obs
from isaac_rover_orbit_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{os.environ.get('POLTAG', 'POLSTAMP')}.so")   # python tools/build_diag.py POLSTAMP
from isaac_rover_orbit_amd.policy import RoverNet, forward_pair
n = 4096
PAIR = len(sys.argv) > 1 and sys.argv[1] == "pair"     # the actor + critic pair kernel instead of one network
ws, bs = random_policy_weights(seed=0, scale=3.0)
net = RoverNet(ws, bs)
obs = torch.from_numpy(synthetic_obs(n)).cuda()
stamps = torch.zeros(n // 16, 16, dtype=torch.int64, device="cuda")
fn = net._lib.rover_debug_set_policy_stamps; fn.argtypes = [C.c_void_p]
assert fn(C.c_void_p(stamps.data_ptr())) == 0
if PAIR:
    wc = [w.copy() for w in ws]; bc = [b.copy() for b in bs]
    wc[5], bc[5] = wc[5][:1].copy(), bc[5][:1].copy()
    critic = RoverNet(wc, bc, final_act="none")
for _ in range(5):
    if PAIR: forward_pair(net, critic, obs)
    else: net(obs)
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
names = ["obs tile in LDS", "L1 961->80", "L2 80->60", "L3 64->256", "L4 256->160", "L5 160->128", "L6 128->2"]
for k in range(1, 8):
    d = s[:, k] - s[:, k - 1]
    print(f"{names[k-1]:18s} +{np.median(d):8.0f} cycles (p90 {np.percentile(d, 90):8.0f})")
print("total", np.median(s[:, 7] - s[:, 0]), "start spread", s[:, 0].max() - s[:, 0].min())
