#!/usr/bin/env python3
"""Dev tool (GPU box): phase timeline of the group step kernel from s_memtime stamps (diagnostic build)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import _lib, terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = 4096
_lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{os.environ.get('K1TAG', 'K1STAMP')}.so")
_lib.EXPORTS.append("rover_debug_set_k1_stamps")
ter = T.make_procedural_terrain((2048, 2048)); ter.make_spawns(2 * n)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
env = RoverEnv(cfg, terrain=ter)
if os.environ.get("FUSED"):
    fnf = env._lib.rover_debug_set_fused; fnf.argtypes = [C.c_void_p, C.c_int]
    assert fnf(env._h, int(os.environ["FUSED"])) == 0
env.reset()
print(env.kernel_names())
stamps = torch.zeros(n // 16, 32, dtype=torch.int64, device="cuda")
fn = env._lib.rover_debug_set_k1_stamps; fn.argtypes = [C.c_void_p]
assert fn(C.c_void_p(stamps.data_ptr())) == 0
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(8, n, 2, device="cuda", generator=g) * 2 - 1
for k in range(8):
    env.step(acts[k])
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
names = {0: "start", 1: "state loaded + ackermann"}
for i in range(6):
    names[2 + 3 * i] = f"sub{i} start"; names[3 + 3 * i] = f"sub{i} geometry done"; names[4 + 3 * i] = f"sub{i} solver done"
names.update({20: "physics done", 23: "state stored + force gathered", 24: "mdp terms done", 21: "rewards + reset done", 22: "command done", 25: "log + final stores done"})
order = [0, 1] + [k for i in range(6) for k in (2 + 3 * i, 3 + 3 * i, 4 + 3 * i)] + [20, 23, 24, 21, 22, 25]
fused = env.kernel_names()[0].startswith("rover_step_scan_kernel")
if fused:
    names.update({27: "scan: the copy wave's word polled, windows read", 28: "scan: window 1 requested + landed, env 1 cast (tile 1)",
                  29: "scan: window 3 requested + landed", 26: "scan: env 3 cast"})
    order += [27, 28, 29, 26]
keys = order
prev = None
tot = None
for k in keys:
    if prev is not None:
        d = s[:, k] - s[:, prev]
        print(f"{names[k]:28s} +{np.median(d):8.0f} cycles (p90 {np.percentile(d, 90):8.0f})")
    prev = k
tot = s[:, keys[-1]] - s[:, 0]
print("total start->end median", np.median(tot), "cycles")
# the kernel ends with its SLOWEST workgroup: the tail of the distribution (workgroups in which an env reset) is what the launch pays
print("total start->end percentiles  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f   (workgroups %d)" % (np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), len(tot)))
slow = np.argsort(tot)[-4:]
for wg in slow:
    d = [s[wg, keys[i + 1]] - s[wg, keys[i]] for i in range(len(keys) - 1)]
    print("  slow workgroup %4d total %6.0f: " % (wg, tot[wg]) + " ".join("%s=%d" % (names[keys[i + 1]].split()[0][:6] + str(keys[i + 1]), d[i]) for i in range(len(d)) if d[i] > 1.3 * np.median(s[:, keys[i + 1]] - s[:, keys[i]])))
print("last stamp - first stamp over the whole grid: %.0f cycles" % (s[:, keys[-1]].max() - s[:, 0].min()))
