#!/usr/bin/env python3
"""Dev tool (GPU box): no-wait timeline of the one-launch kernel's step wave 0 and copy wave 4 of every workgroup (build K1LITE):
cycles since the step wave's start, medians over the workgroups of one launch."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{os.environ.get('ABLTAG', 'K1LITE')}.so")
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ter = T.make_procedural_terrain((2048, 2048), seed=1234, n_rocks=400); ter.make_spawns(2 * n, seed=41)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
env = RoverEnv(cfg, terrain=ter); env.reset()
stamps = torch.zeros(n // 16, 64, dtype=torch.int64, device="cuda")
fn = env._lib.rover_debug_set_k1_stamps; fn.argtypes = [C.c_void_p]
assert fn(C.c_void_p(stamps.data_ptr())) == 0
g = torch.Generator(device="cuda").manual_seed(0)
acts = torch.rand(40, n, 2, device="cuda", generator=g) * 2 - 1
S = env.get_state()      # every 37th env times out in the LAST of the 40 steps: what do the workgroups with a reset cost?
S[::37, 51] = torch.tensor([env.max_episode_length - 40], dtype=torch.int32).view(torch.float32).item()
env.set_state(S)
for k in range(40): out = env.step(acts[k])
torch.cuda.synchronize()
_r = torch.nonzero(out[2] | out[3]).flatten()
reset_wg = torch.unique(_r // 16).cpu().numpy()      # workgroups with an env that reset in the last step
reset_w0 = torch.unique(_r[(_r % 16) < 4] // 16).cpu().numpy()      # ... in their wave 0 (the stamped step wave)
s = stamps.cpu().numpy().astype(np.float64)
t0 = s[:, 0:1]
step = {0: "start", 1: "physics done (before A)", 2: "reset decided, flag word set (no barrier)", 11: "ray table requested", 12: "group_store issued",
        13: "force rows stored", 14: "mdp terms", 15: "rewards + reset", 16: "command", 3: "tail done (log, final stores issued)", 4: "ray table / stores retired",
        5: "copy wave's word polled, windows read", 6: "window 1 requested, landed, env 1 cast (tile 1)", 8: "window 3 requested", 9: "... landed",
        10: "end (env 3 cast)"}
copy = {0: "before A (link work done)", 1: "after A", 2: "windows derived, window 0 requested, the step wave's flag polled", 3: "final windows written, word set",
        4: "... window 0 landed", 5: "env 0 cast (tile 0)", 6: "... its LDS reads returned", 8: "window 2 requested", 9: "... landed", 12: "end (env 2 cast)"}
print(env.kernel_names()[0])
for name, off, labels in (("step wave", 0, step), ("copy wave", 32, copy)):
    print(name)
    prev = None
    for i, lab in labels.items():
        if not (s[:, off + i] > 0).all():      # a fine stamp (build K1LITEF only)
            continue
        d = s[:, off + i] - t0[:, 0]
        m = np.median(d)
        print(f"  {lab:50s} {m:8.0f}  (p90 {np.percentile(d, 90):8.0f})" + (f"   +{m - prev:6.0f}" if prev is not None else ""))
        prev = m
end = np.maximum(s[:, 10], s[:, 32 + 15]) - t0[:, 0]
mask = np.zeros(end.shape[0], bool); mask[reset_wg] = True
print(f"workgroup end (later of the two waves): median {np.median(end):.0f}, p90 {np.percentile(end, 90):.0f}, max {end.max():.0f}; "
      f"the {mask.sum()} workgroups with a reset in this step: {np.sort(end[mask]).astype(int).tolist()}; start spread {s[:, 0].max() - s[:, 0].min():.0f} cycles; "
      f"first start -> last end {(np.maximum(s[:, 10], s[:, 47]).max() - s[:, 0].min()):.0f} cycles")
top = np.argsort(end)[-6:]
for w in top:
    print(f"  slow workgroup {w}: end {end[w]:.0f}  physics done {s[w, 1] - s[w, 0]:.0f}  A2 {s[w, 2] - s[w, 0]:.0f}  tail done {s[w, 3] - s[w, 0]:.0f}  B {s[w, 5] - s[w, 0]:.0f}  reset in it: {bool(mask[w])}")
mask0 = np.zeros(end.shape[0], bool); mask0[reset_w0] = True
if mask0.any() and (~mask).any():      # segment by segment: workgroups with a reset against the others (step wave; fine stamps if the build has them)
    keys = [k for k in step if (s[:, k] > 0).all()]
    prev = None
    print(f"step wave 0, median cycles per segment: no reset in the workgroup / a reset in wave 0 ({mask0.sum()} workgroups)")
    for k in keys:
        if prev is not None:
            d = s[:, k] - s[:, prev]
            print(f"  {step[k]:50s} {np.median(d[~mask]):8.0f} {np.median(d[mask0]):8.0f}")
        prev = k
