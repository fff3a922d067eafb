#!/usr/bin/env python3
"""Dev tool (GPU box): per-workgroup phase timeline of the scan kernel from s_memtime stamps (diagnostic build)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import _lib, terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = 4096
_lib.LIB_PATH = os.path.join(ROOT, "build", "abl", "librover_ablSTAMP.so")
_lib.EXPORTS.append("rover_debug_scan")
ter = T.make_procedural_terrain((2048, 2048)); ter.make_spawns(2 * n)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
env = RoverEnv(cfg, terrain=ter); env.reset()
scan = torch.empty(n, 961, device="cuda")
stamps = torch.zeros(n, 8, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fn = env._lib.rover_debug_scan
fn.argtypes = [C.c_void_p] * 4
for _ in range(5):
    fn(env._h, C.c_void_p(scan.data_ptr()), C.c_void_p(stamps.data_ptr()), st)
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
t0 = s[:, 0].min()
rel = (s[:, :5] - t0)
order = np.argsort(rel[:, 0])
print("clock ticks (shader cycles @ ~100MHz? s_memtime) -- per-phase medians [start->desc, desc->tile issued, ->barrier passed, ->rays done]:")
d = np.diff(rel, axis=1)
print(np.median(d, axis=0), "p90", np.percentile(d, 90, axis=0))
print("WG start times: min/median/max", rel[:, 0].min(), np.median(rel[:, 0]), rel[:, 0].max(), " end max", rel[:, 4].max())
life = rel[:, 4] - rel[:, 0]
print("WG lifetime median", np.median(life), "p90", np.percentile(life, 90))
# rounds: histogram of start times
h, edges = np.histogram(rel[:, 0], bins=12)
print("start-time histogram:", h.tolist(), "bin width", edges[1] - edges[0])
