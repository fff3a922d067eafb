#!/usr/bin/env python3
"""Minimal PPO on the HIP-backed AAURoverEnv-v0: the reference's actor / critic architecture (learning/skrl/models.py) and
hyper-parameters (learning/skrl/rover_ppo.yaml: rollouts 60, 4 epochs, 60 mini-batches, gamma 0.99, lambda 0.95,
lr 1e-4, clip 0.2, grad-norm 0.5, KL-adaptive learning rate, kl_threshold 0.008).  Rollouts run entirely on the fused
kernels (policy mean and value through ``RoverNet``, re-packed after every update; env.step = two HIP kernels); only
the PPO update itself uses torch autograd.  A stand-in for the reference's skrl trainer (examples/02_train/train.py),
which needs packages that are not part of this repository.

    python examples/04_train_ppo.py --num_envs 4096 --iterations 100
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isaac_rover_orbit_amd import terrain as T  # noqa: E402
from isaac_rover_orbit_amd.cfg import RoverEnvCfg  # noqa: E402
from isaac_rover_orbit_amd.envs import RoverEnv  # noqa: E402
from isaac_rover_orbit_amd.policy import RoverNet  # noqa: E402


class Net(nn.Module):
    """models.py:39-103 / 106-163: encoder 961 -> 80 -> 60 on obs[:, 3:-1], MLP (4 + 60) -> 256 -> 160 -> 128 -> out."""

    def __init__(self, out_dim, final_tanh):
        super().__init__()
        act = nn.LeakyReLU
        self.dense_encoder = nn.Module()
        self.dense_encoder.encoder_layers = nn.ModuleList([nn.Linear(961, 80), act(), nn.Linear(80, 60), act()])
        self.mlp = nn.ModuleList([nn.Linear(64, 256), act(), nn.Linear(256, 160), act(), nn.Linear(160, 128), act(),
                                  nn.Linear(128, out_dim)] + ([nn.Tanh()] if final_tanh else []))
        if final_tanh:
            self.log_std_parameter = nn.Parameter(torch.zeros(out_dim))

    def forward(self, s):
        e = s[:, 3:-1]
        for layer in self.dense_encoder.encoder_layers:
            e = layer(e)
        x = torch.cat([s[:, 0:4], e], 1)
        for layer in self.mlp:
            x = layer(x)
        return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--rollouts", type=int, default=60)
    ap.add_argument("--out", default=None, help="write the per-iteration statistics as JSON lines")
    ap.add_argument("--save", default=None, help="write a skrl-style checkpoint {'policy': state_dict, 'value': state_dict}")
    args = ap.parse_args()
    torch.manual_seed(42)
    dev = torch.device("cuda")
    n, Tn = args.num_envs, args.rollouts
    terrain = T.make_procedural_terrain((2048, 2048), seed=1234)
    terrain.make_spawns(2 * n)
    cfg = RoverEnvCfg(); cfg.scene.num_envs = n; cfg.terrain.kind = "custom"
    env = RoverEnv(cfg, terrain=terrain)
    policy, value = Net(2, True).to(dev), Net(1, False).to(dev)
    opt = torch.optim.Adam(list(policy.parameters()) + list(value.parameters()), lr=1e-4)
    gamma, lam, clip, vclip, kl_thr = 0.99, 0.95, 0.2, 0.2, 0.008

    obs_buf = torch.empty(Tn, n, 965, device=dev)
    act_buf = torch.empty(Tn, n, 2, device=dev)
    logp_buf, val_buf, rew_buf = (torch.empty(Tn, n, device=dev) for _ in range(3))
    done_buf = torch.empty(Tn, n, device=dev)
    obs, _ = env.reset()
    o = torch.nan_to_num(obs["policy"], neginf=0.0)
    out = open(args.out, "w") if args.out else None
    for it in range(args.iterations):
        t0 = time.perf_counter()
        # ---- rollout on the fused kernels
        actor = RoverNet.from_state_dict(policy.state_dict(), final_act="tanh")
        critic = RoverNet.from_state_dict(value.state_dict(), final_act="none")
        log_std = policy.log_std_parameter.detach().clamp(-20.0, 2.0)
        std = log_std.exp()
        ep_count = torch.zeros((), device=dev); ep_stats = torch.zeros(4, device=dev)
        for t in range(Tn):
            mean = actor(o)
            a = mean + std * torch.randn_like(mean)
            logp_buf[t] = (-0.5 * ((a - mean) / std) ** 2 - log_std - 0.9189385332).sum(1)
            val_buf[t] = critic(o).squeeze(1)
            obs_buf[t], act_buf[t] = o, a
            obs, rew, term, trunc, info = env.step(a.clamp(-1.0, 1.0))      # clip_actions (models.py:66)
            o = torch.nan_to_num(obs["policy"], neginf=0.0)
            rew_buf[t], done_buf[t] = rew, (term | trunc).float()
            lv = env.episode_log_vector
            ep_count += lv[13]; ep_stats += torch.where(lv[13] > 0, lv[7:11], torch.zeros_like(lv[7:11]))
        torch.cuda.synchronize(); t_roll = time.perf_counter() - t0
        # ---- GAE (skrl PPO: bootstraps through time-outs like the reference's config)
        with torch.no_grad():
            last_v = critic(o).squeeze(1)
            adv = torch.zeros_like(rew_buf); gae = torch.zeros(n, device=dev)
            for t in reversed(range(Tn)):
                nv = last_v if t == Tn - 1 else val_buf[t + 1]
                nd = 1.0 - done_buf[t]
                delta = rew_buf[t] + gamma * nv * nd - val_buf[t]
                gae = delta + gamma * lam * nd * gae
                adv[t] = gae
            ret = adv + val_buf
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        # ---- PPO update (torch autograd)
        B = Tn * n
        fo, fa, flp, fv, fr, fadv = (x.reshape(B, *x.shape[2:]) for x in (obs_buf, act_buf, logp_buf, val_buf, ret, adv))
        kl_mean = 0.0
        for epoch in range(4):
            perm = torch.randperm(B, device=dev)
            kls = []
            for mb in perm.chunk(60):
                mean = policy(fo[mb])
                ls = policy.log_std_parameter.clamp(-20.0, 2.0)
                lp = (-0.5 * ((fa[mb] - mean) / ls.exp()) ** 2 - ls - 0.9189385332).sum(1)
                ratio = (lp - flp[mb]).exp()
                with torch.no_grad():
                    kls.append(((ratio - 1) - (lp - flp[mb])).mean())
                pl = -torch.min(ratio * fadv[mb], ratio.clamp(1 - clip, 1 + clip) * fadv[mb]).mean()
                v = value(fo[mb]).squeeze(1)
                v = fv[mb] + (v - fv[mb]).clamp(-vclip, vclip)
                vl = ((fr[mb] - v) ** 2).mean()
                opt.zero_grad(set_to_none=True)
                (pl + vl).backward()
                nn.utils.clip_grad_norm_(list(policy.parameters()) + list(value.parameters()), 0.5)
                opt.step()
            kl_mean = torch.stack(kls).mean().item()
            lr = opt.param_groups[0]["lr"]                     # KLAdaptiveRL
            if kl_mean > 2 * kl_thr: lr = max(lr / 1.5, 1e-6)
            elif kl_mean < 0.5 * kl_thr: lr = min(lr * 1.5, 1e-2)
            for g in opt.param_groups: g["lr"] = lr
        torch.cuda.synchronize()
        st = {"iteration": it, "mean_step_reward": rew_buf.mean().item(), "episodes": ep_count.item(),
              "time_out": ep_stats[0].item(), "success": ep_stats[1].item(), "far": ep_stats[2].item(),
              "collision": ep_stats[3].item(), "kl": kl_mean, "lr": opt.param_groups[0]["lr"],
              "rollout_s": t_roll, "rollout_env_steps_per_s": Tn * n / t_roll, "iteration_s": time.perf_counter() - t0}
        print(json.dumps(st), flush=True)
        if out:
            out.write(json.dumps(st) + "\n"); out.flush()
    if args.save:
        torch.save({"policy": policy.state_dict(), "value": value.state_dict()}, args.save)
    env.close()


if __name__ == "__main__":
    main()
