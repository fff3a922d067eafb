#!/usr/bin/env python3
"""Dev tool (GPU box): does a step in which envs reset take longer?  Runs `steps` env steps at N = 4096 from a fresh reset and
writes the number of resets of every step; run it under `rocprofv3 --kernel-trace` and join with the per-dispatch durations:

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/reset_cost.py 400 OUT/resets.txt
    python3 tools/reset_cost.py --join OUT
"""
import csv, glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--join":
    d = sys.argv[2]
    resets = np.loadtxt(os.path.join(d, "resets.txt"), dtype=np.int64)
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "rover_step_scan" in r["Kernel_Name"] or "rover_step_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows], dtype=np.float64) / 1e3
    n = min(len(dur), len(resets))
    dur, resets = dur[:n], resets[:n]
    print("steps", n, " kernel", rows[0]["Kernel_Name"][:60])
    for lo, hi in ((0, 0), (1, 4), (5, 16), (17, 64), (65, 10 ** 9)):
        m = (resets >= lo) & (resets <= hi)
        if m.any():
            print(f"resets in the step {lo:3d}..{min(hi, 9999):4d}: {m.sum():4d} steps, kernel mean {dur[m].mean():7.2f} us  min {dur[m].min():7.2f}  max {dur[m].max():7.2f}")
    k = max(n // 8, 1)
    for i in range(0, n, k):
        print(f"steps {i:4d}..{min(i + k, n) - 1:4d}: mean resets {resets[i:i + k].mean():7.2f}  kernel mean {dur[i:i + k].mean():7.2f} us")
    sys.exit(0)
import torch
if os.environ.get("ABLTAG"):      # a tools/build_diag.py variant instead of the product library
    from isaac_rover_orbit_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{os.environ['ABLTAG']}.so")
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
steps, out = int(sys.argv[1]), sys.argv[2]
n = 4096
ter = T.make_procedural_terrain((2048, 2048)); ter.make_spawns(2 * n)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
env = RoverEnv(cfg, terrain=ter)
env.reset()
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(64, n, 2, device="cuda", generator=g) * 2 - 1
counts = torch.zeros(steps, dtype=torch.int64, device="cuda")
for k in range(steps):
    _, _, term, trunc, _ = env.step(acts[k % 64])
    counts[k] = (term | trunc).sum()
torch.cuda.synchronize()
np.savetxt(out, counts.cpu().numpy(), fmt="%d")
