#!/usr/bin/env python3
"""Dev tool (GPU box): env-steps/s and kernel times versus num_envs and step-kernel mapping (1 GPU)."""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
if os.environ.get("ABLTAG"):
    from isaac_rover_orbit_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{os.environ['ABLTAG']}.so")
ter = T.make_procedural_terrain((2048, 2048))
out = []
# mapping: auto = the product's choice (mapping and launch form),
# group2 = the group mapping as two launches (log reduced behind every step), groupf / group1 = one launch forced beyond that
# batch size: the copy-wave form / the single-tile form (no copy waves, two workgroups per CU)
CASES = [(1024, "auto"), (1024, "group1"), (2048, "auto"), (2048, "group2"), (4096, "auto"), (4096, "group1"), (4096, "group2"),
         (4096, "lane"), (8192, "auto"), (8192, "groupf"), (8192, "group2"), (16384, "auto"), (16384, "group2"), (32768, "auto"),
         (32768, "group2"), (32768, "lane"), (65536, "auto"), (65536, "group2"), (65536, "lane"), (131072, "auto"), (131072, "group2"),
         (131072, "lane")]
if len(sys.argv) > 1:   # e.g. 1024:group 2048:group:wave  (third field: scan kernel of the step path, auto | generic | epi1)
    CASES = [tuple(a.split(":")) for a in sys.argv[1:]]
import ctypes as C
FORMS = {"auto": 0, "generic": 1, "epi1": 2}
for case in CASES:
    n, mapping, form = int(case[0]), case[1], (case[2] if len(case) > 2 else "auto")
    ter.make_spawns(2 * n)
    cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"; cfg.step_mapping = "group" if mapping.startswith("group") else mapping   # "auto" = the product's own choice
    if mapping == "group2":
        cfg.log_reduction = "every_step"
    env = RoverEnv(cfg, terrain=ter)
    if mapping in ("groupf", "group1", "group2"):      # forced: one launch, copy-wave form / single-tile form; two launches
        ff = C.CDLL(env._lib._name).rover_debug_set_fused; ff.argtypes = [C.c_void_p, C.c_int]
        assert ff(env._h, {"groupf": 1, "group1": 2, "group2": 0}[mapping]) == 0
    fn = C.CDLL(env._lib._name).rover_debug_set_scan_form
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(env._h, FORMS[form]) == 0
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.rand(16, n, 2, device="cuda", generator=g) * 2 - 1
    # 300 untimed steps (the first windows after constructing an env run slower: clock ramp, first touch -- tools/host_path.py),
    # then the BEST of three 200-step windows: one window alone once read 73 us per step beside a 44 us kernel (round 3)
    for k in range(300): env.step(acts[k % 16])
    steps, dt = 200, 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(steps): env.step(acts[k % 16])
        torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
    a = b = 0.0
    for k in range(20):
        x, y = env.profile_step(acts[k % 16]); a += x; b += y
    r = {"num_envs": n, "mapping": mapping, "kernels": list(env.kernel_names()), "scan_form": form, "env_steps_per_s": n * steps / dt, "us_per_step": dt / steps * 1e6,
         "step_kernel_us": a / 20 * 1e3, "scan_kernel_us": b / 20 * 1e3}
    print(json.dumps(r)); out.append(r)
    env.close(); del env; torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "n_sweep.json"), "w"), indent=1)
