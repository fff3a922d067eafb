#!/usr/bin/env python3
"""Closed-loop evaluation of a trained actor (counterpart of the reference's examples/03_inference_pretrained/eval.py:
146-155): the skrl checkpoint's policy is packed once and evaluated by the fused MFMA kernel between env steps.

    python examples/03_eval_policy.py --checkpoint <.../policies/best_agent.pt> --num_envs 1024 --steps 750
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_rover_orbit_amd import terrain as T  # noqa: E402
from isaac_rover_orbit_amd.cfg import RoverEnvCfg  # noqa: E402
from isaac_rover_orbit_amd.envs import RoverEnv  # noqa: E402
from isaac_rover_orbit_amd.policy import RoverNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--num_envs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=750)
    args = ap.parse_args()
    terrain = T.make_procedural_terrain((2048, 2048), seed=1234)
    terrain.make_spawns(2 * args.num_envs)
    cfg = RoverEnvCfg()
    cfg.scene.num_envs = args.num_envs
    cfg.terrain.kind = "custom"
    env = RoverEnv(cfg, terrain=terrain)
    actor = RoverNet.from_checkpoint(args.checkpoint, role="policy")
    obs, _ = env.reset()
    counts = torch.zeros(4)                              # time_out, is_success, far_from_target, collision
    episodes = 0.0
    for _ in range(args.steps):
        torch.nan_to_num_(obs["policy"], neginf=0.0)    # rays that leave the map report -inf (ORBIT RayCaster semantics)
        obs, rew, terminated, truncated, info = env.step(actor.act(obs))
        log = env.episode_log_vector.cpu()               # one small device -> host copy per step (this is an example)
        if log[13] > 0:
            episodes += float(log[13])
            counts += log[7:11]
    print(f"episodes finished: {episodes:.0f}; terminations (time_out, success, far, collision): {counts.tolist()}")
    env.close()


if __name__ == "__main__":
    main()
