#!/usr/bin/env python3
"""Baseline (2) of SURVEY 8d (GPU box): the manager-level arithmetic of one env step written as UNFUSED torch-ROCm ops in
the reference's launch pattern -- process_actions + 6 x Ackermann apply (ackermann_actions.py:226-322), command update
(terrain_importer.py:97-101), the three observation terms + concat (observations.py:15-45), seven reward terms
(rewards.py:14-137), four termination terms (terminations.py:14-64) -- on resident tensors at N = 4096.  The physics, the
ray-caster and the contact report (PhysX / Warp in the reference) are NOT included: this is a LOWER bound of what the
reference-style path costs per step on this GPU, to put next to the two fused kernels of this repository.
Restated here from the survey's description of those files (timing only, not a parity artefact).  Prints one JSON line."""
import json, math, sys, time
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
R = 961
wheel_radius, d_fr, d_mw, wl = 0.1, 0.77, 0.894, 0.849
L_ep = 750.0


def ackermann(lin, ang):
    direction = torch.where(lin >= 0, 1.0, -1.0)
    turn = torch.where(ang >= 0, 1.0, -1.0)
    la, aa = lin.abs(), ang.abs()
    radius = torch.where(aa > 0, la / torch.where(aa > 0, aa, torch.ones_like(aa)), torch.full_like(aa, float("inf")))
    radius = torch.clamp(radius, min=0.8 * d_mw)
    r_ml, r_mr = radius - d_mw / 2 * turn, radius + d_mw / 2 * turn
    r_f = torch.sqrt((radius - d_fr / 2 * turn) ** 2 + wl ** 2 / 4)
    r_r = torch.sqrt((radius + d_fr / 2 * turn) ** 2 + wl ** 2 / 4)
    point = radius < d_mw
    v_lin = torch.where(aa > 0, aa, la)
    vel = [torch.where(point, -(la + 1) * turn, r * v_lin * direction) / wheel_radius for r in (r_ml, r_f, r_f, r_r, r_mr, r_r)]
    steer = torch.atan2(torch.full_like(radius, wl), radius - d_fr / 2) * turn
    steer = torch.where(point, torch.full_like(steer, math.pi / 4) * turn, steer)
    return torch.stack([steer, -steer, -steer, steer], 1), torch.stack(vel, 1)


def step(state):
    raw, prev, pos, quat, target, heading_w_cmd, hits_z, forces, t = state
    processed = raw * 1.0 + (-0.0135)
    for _ in range(6):                                       # rover_env.py:64-72: apply_actions every substep
        joint_pos, joint_vel = ackermann(processed[:, 0], processed[:, 1])
    # command update
    yaw = torch.atan2(2 * (quat[:, 0] * quat[:, 3] + quat[:, 1] * quat[:, 2]), 1 - 2 * (quat[:, 2] ** 2 + quat[:, 3] ** 2))
    d = target - pos
    c, s = torch.cos(yaw), torch.sin(yaw)
    cmd_b = torch.stack([c * d[:, 0] + s * d[:, 1], -s * d[:, 0] + c * d[:, 1], d[:, 2]], 1)
    heading_b = torch.remainder(heading_w_cmd - yaw + math.pi, 2 * math.pi) - math.pi
    # observations
    dist = torch.norm(cmd_b[:, :2], dim=1)
    angle = torch.atan2(cmd_b[:, 1], cmd_b[:, 0])
    scan = pos[:, 2:3] - hits_z - 0.26878
    obs = torch.cat([raw, (dist * 0.11).unsqueeze(1), (angle / math.pi).unsqueeze(1), scan], 1)
    # rewards
    r_dist = (1.0 / (1.0 + 0.11 * dist * dist)) / L_ep
    r_reach = torch.where(dist < 0.18, (L_ep - t) / L_ep, torch.zeros_like(dist))
    dlin, dang = (raw[:, 0] - prev[:, 0]).abs(), (raw[:, 1] - prev[:, 1]).abs()
    osc = (torch.where(dlin * 3 > 0.05, (dlin * 3) ** 2, torch.zeros_like(dlin)) ** 2 +
           torch.where(dang * 3 > 0.05, (dang * 3) ** 2, torch.zeros_like(dang)) ** 2) / L_ep
    r_angle = torch.where(angle.abs() > 2.0, angle.abs() / L_ep, torch.zeros_like(angle))
    r_back = torch.where(raw[:, 0] < 0, torch.full_like(dist, 1.0 / L_ep), torch.zeros_like(dist))
    fnorm = torch.norm(forces.view(n, -1, 3), dim=1)
    coll = (fnorm.sum(-1) > 1).float()
    r_far = (dist > 11.0).float()
    reward = (5 * r_dist + 5 * r_reach - 0.1 * osc - 1.5 * r_angle - 0.5 * r_back - 2 * coll - 2 * r_far) * 0.2
    # terminations
    time_out = t >= L_ep
    done = time_out | (dist < 0.18) | (dist > 11.0) | (coll > 0)
    return obs, reward, done, time_out, joint_pos, joint_vel, heading_b


state = (torch.rand(n, 2, device=dev, generator=g) * 2 - 1, torch.rand(n, 2, device=dev, generator=g) * 2 - 1,
         torch.rand(n, 3, device=dev, generator=g) * 50, torch.nn.functional.normalize(torch.randn(n, 4, device=dev, generator=g), dim=1),
         torch.rand(n, 3, device=dev, generator=g) * 50, torch.rand(n, device=dev, generator=g) * 6 - 3,
         torch.rand(n, R, device=dev, generator=g), torch.randn(n, 13, 1, 3, device=dev, generator=g), torch.full((n,), 100.0, device=dev))
with torch.no_grad():
    for _ in range(10): step(state)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 100
    for _ in range(reps): step(state)
    torch.cuda.synchronize()
us = (time.perf_counter() - t0) / reps * 1e6
print(json.dumps({"n": n, "torch_unfused_manager_arithmetic_us_per_step": us, "env_steps_per_s_upper_bound": n / us * 1e6,
                  "note": "no physics / ray-cast / contact report; reference-style launch pattern (6 x ackermann per step)"}))
