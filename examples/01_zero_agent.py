#!/usr/bin/env python3
"""Zero-action agent on the HIP-backed AAURoverEnv-v0 (counterpart of the reference's examples/01_zero_agent.py:36-52).

    python examples/01_zero_agent.py --num_envs 4096 --steps 500
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_rover_orbit_amd import terrain as T  # noqa: E402
from isaac_rover_orbit_amd.cfg import RoverEnvCfg  # noqa: E402
from isaac_rover_orbit_amd.envs import RoverEnv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--random", action="store_true", help="uniform random actions instead of zeros (02_random_agent)")
    args = ap.parse_args()
    terrain = T.make_procedural_terrain((2048, 2048), seed=1234)
    terrain.make_spawns(2 * args.num_envs)
    cfg = RoverEnvCfg()
    cfg.scene.num_envs = args.num_envs
    cfg.terrain.kind = "custom"
    env = RoverEnv(cfg, terrain=terrain)
    print("observation space:", env.observation_space, "action space:", env.action_space)
    obs, _ = env.reset()
    actions = torch.zeros(env.action_space.shape, device=env.unwrapped.device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if args.random:
            actions = torch.rand_like(actions) * 2 - 1
        obs, rew, terminated, truncated, info = env.step(actions)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.steps} steps x {args.num_envs} envs: {args.steps * args.num_envs / dt / 1e6:.1f} M env-steps/s; "
          f"mean reward {rew.mean().item():+.4f}; episode log: "
          + ", ".join(f"{k.split('/')[-1]}={v.item():.3f}" for k, v in info.get("episode", {}).items()))
    env.close()


if __name__ == "__main__":
    main()
