#!/usr/bin/env python3
"""Dev tool (GPU box): a short, fixed workload for rocprofv3 --pmc passes: reset + 20 env steps at N=4096."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("ABLTAG"):   # diagnostic build of tools/build_diag.py
    from isaac_rover_orbit_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "build", "abl", f"librover_abl{os.environ['ABLTAG']}.so")
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ter = T.make_procedural_terrain((2048, 2048)); ter.make_spawns(2 * n)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
if os.environ.get("QB_ITERS"): cfg.solver_iterations = int(os.environ["QB_ITERS"])      # e.g. the round-4 model: 16 / lumped
if os.environ.get("QB_MASS"): cfg.mass_model = os.environ["QB_MASS"]
cfg.roctx_markers = len(sys.argv) > 3 and sys.argv[3] == "markers"     # rocprofv3 --marker-trace: one range per env step
env = RoverEnv(cfg, terrain=ter)
if os.environ.get("ROVER_SCAN_FORM"):   # 3 = 8 x 8 ray blocks per wave, 4 = lines (rover_debug_set_scan_form)
    import ctypes as C
    fn = C.CDLL(env._lib._name).rover_debug_set_scan_form; fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(env._h, int(os.environ["ROVER_SCAN_FORM"])) == 0
def cluster_by_xcd(env, n):
    """QB_CLUSTER=1 (experiment): permute the envs' states after the reset so that the envs of the workgroups one XCD runs (workgroup
    b = envs 16 b .. 16 b + 15 runs on XCD b mod 8) stand in one x-strip of the terrain -- what the L2 fetch would be if env -> XCD
    followed the rovers' positions."""
    S = env.get_state()
    order = torch.argsort(S[:, 0]).view(8, -1)                       # eight strips by x, n / 8 envs each
    xcd = (torch.arange(n, device=S.device) // 16) % 8
    slot = torch.zeros(n, dtype=torch.long, device=S.device)
    for k in range(8):
        slot[xcd == k] = order[k]
    env.set_state(S[slot].contiguous())


def run(env):
    env.reset()
    if os.environ.get("QB_CLUSTER"): cluster_by_xcd(env, n)
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.rand(min(steps, 64), n, 2, device="cuda", generator=g) * 2 - 1
    for k in range(steps):
        env.step(acts[k % acts.shape[0]])
    torch.cuda.synchronize()
    env.close()
run(env)
if os.environ.get("ROVER_ALSO_TWO_LAUNCH"):   # the same workload on the two-launch path (the traffic of the step kernel alone)
    cfg2 = RoverEnvCfg(); cfg2.scene.num_envs = n; cfg2.terrain.kind = "custom"; cfg2.log_reduction = "every_step"
    cfg2.solver_iterations, cfg2.mass_model = cfg.solver_iterations, cfg.mass_model
    env2 = RoverEnv(cfg2, terrain=ter)
    import ctypes as C
    ff = C.CDLL(env2._lib._name).rover_debug_set_fused; ff.argtypes = [C.c_void_p, C.c_int]
    assert ff(env2._h, 0) == 0
    run(env2)
