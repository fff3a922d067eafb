#!/usr/bin/env python3
"""Dev tool (GPU box): phase timeline of lift_step_kernel from s_memtime stamps (diagnostic build LIFTSTAMP), 2048 envs."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build", "abl", "librover_ablLIFTSTAMP.so")
from isaac_rover_orbit_amd.envs import FrankaCubeLiftEnv, LiftEnvCfg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = LiftEnvCfg(); cfg.scene.num_envs = n
env = FrankaCubeLiftEnv(cfg); env.reset()
stamps = torch.zeros((n + 7) // 8, 2, 32, dtype=torch.int64, device="cuda")
fn = env._lib.rover_lift_debug_set_stamps; fn.argtypes = [C.c_void_p]
assert fn(C.c_void_p(stamps.data_ptr())) == 0
if os.environ.get("LIFT_PIPE"):
    assert env._lib.rover_lift_debug_set_pipeline(env._h, int(os.environ["LIFT_PIPE"])) == 0
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(16, n, 8, device="cuda", generator=g) * 2 - 1
for k in range(16):
    env.step(acts[k])
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
names = {0: "state loaded (physics half)", 1: "joint sin/cos exchanged"}
for i in range(2):
    names.update({2 + 6 * i: f"sub{i} inverse-dynamics pass", 3 + 6 * i: f"sub{i} exchange + 7x7 solves + integrate",
                  4 + 6 * i: f"sub{i} sin/cos + hand kinematics (+ hand-off, barrier)", 5 + 6 * i: f"sub{i} corner rows + exchange",
                  7 + 6 * i: f"sub{i} pads test + 8 sweeps + integrate"})
names.update({14: "manager words exchanged", 15: "terms + rewards", 16: "reset + command", 17: "stores complete"})
print(env.kernel_name())
pipe = env.kernel_name().endswith("true>")
t0 = s[:, 0, 0]
if not pipe:
    order = [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17]
    prev = None
    for k in order:
        if prev is not None:
            d = s[:, 0, k] - s[:, 0, prev]
            print(f"{names[k]:56s} +{np.median(d):8.0f} (p90 {np.percentile(d, 90):8.0f})")
        prev = k
    print("total first stamp -> last stamp, median:", np.median(s[:, 0, 17] - t0), "s_memtime ticks")
else:
    # two timelines against the arm wave's first stamp: wave 0 = arm (stamps 0..4, 8..10), wave 1 = cube + managers
    for w, order in ((0, [0, 1, 2, 3, 4, 8, 9, 10]), (1, [0, 1, 4, 5, 7, 10, 11, 13, 14, 15, 16, 17])):
        print("wave", w, "(arm)" if w == 0 else "(cube, managers)")
        prev = None
        for k in order:
            at = np.median(s[:, w, k] - t0)
            d = "" if prev is None else f"+{np.median(s[:, w, k] - s[:, w, prev]):8.0f}"
            print(f"   {names[k]:56s} at {at:8.0f} {d}")
            prev = k
    print("total (arm wave's first stamp -> cube wave's last stamp), median:", np.median(s[:, 1, 17] - t0), "s_memtime ticks")
