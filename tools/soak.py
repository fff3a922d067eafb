#!/usr/bin/env python3
"""Dev tool (GPU box): long random-action soak of the hot path -- N envs x many steps with in-step resets -- checking that
no observation / reward / state word ever becomes NaN (the scan kernel reports a ray outside its staged window as NaN)
and that -inf only appears where rays leave the map.  Prints one JSON line."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import terrain as T
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs import RoverEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
ter = T.make_procedural_terrain((2048, 2048), seed=1234); ter.make_spawns(2 * n)
cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
env = RoverEnv(cfg, terrain=ter); obs, _ = env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
nan_obs = torch.zeros((), device="cuda"); inf_obs = torch.zeros((), device="cuda"); bad_rew = torch.zeros((), device="cuda")
resets = torch.zeros((), device="cuda"); counts = torch.zeros(4, device="cuda")
t0 = time.perf_counter()
for k in range(steps):
    a = torch.rand(n, 2, device="cuda", generator=g) * 2 - 1
    if k % 500 < 250:
        a[:, 0] = a[:, 0].abs()          # phases of mostly-forward driving so that rovers travel (far / collision / success occur)
    obs, rew, term, trunc, info = env.step(a)
    if k % 10 == 0:
        o = obs["policy"]
        nan_obs += torch.isnan(o).sum(); inf_obs += torch.isinf(o).sum(); bad_rew += (~torch.isfinite(rew)).sum()
    lv = env.episode_log_vector
    resets += lv[13]; counts += torch.where(lv[13] > 0, lv[7:11], torch.zeros_like(lv[7:11]))
torch.cuda.synchronize()
st = env.get_state()      # (N, 72): env-major copy of the SoA state
print(json.dumps({"n": n, "steps": steps, "seconds": time.perf_counter() - t0, "nan_obs": nan_obs.item(), "inf_obs": inf_obs.item(),
                  "nonfinite_rewards": bad_rew.item(), "nonfinite_state_words": (~torch.isfinite(st)).sum().item(),
                  "resets": resets.item(), "terminations_timeout_success_far_collision": counts.tolist(),
                  "max_abs_xy": st[:, 0:2].abs().max().item(), "max_speed": st[:, 7:10].norm(dim=1).max().item(),
                  "max_ang_speed": st[:, 10:13].norm(dim=1).max().item(), "z_range": [st[:, 2].min().item(), st[:, 2].max().item()]}))
