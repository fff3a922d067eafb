#!/usr/bin/env python3
"""Closed-loop sanity signal (the shipped checkpoint is one of the three frozen numeric artefacts of the
reference).  Runs the reference's pretrained PPO policy ``robots/aau_rover/policies/best_agent.pt`` -- trained in Isaac
Sim / PhysX -- against THIS repository's model of the environment (CPU oracle; the HIP path is bit-identical to it) and
compares with random / zero policies and with deliberately wrong observation conventions.

Build-container only (needs /root/reference); nothing of the checkpoint is copied into the repository.

    python tools/policy_closed_loop.py [num_envs] [steps]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import terrain as T  # noqa: E402
from oracle import rover_oracle as ro  # noqa: E402

CKPT = "/root/reference/rover_envs/envs/navigation/robots/aau_rover/policies/best_agent.pt"


def load_policy(path=CKPT):
    """Deterministic (mean) action of GaussianNeuralNetwork, rover_envs/envs/navigation/learning/skrl/models.py:39-102:
    encoder 961 -> 80 -> 60 on obs[:, 3:-1], MLP on cat(obs[:, 0:4], enc) 64 -> 256 -> 160 -> 128 -> 2, leaky_relu, tanh."""
    W = {k: v.float() for k, v in torch.load(path, map_location="cpu", weights_only=False)["policy"].items()}
    act = torch.nn.functional.leaky_relu

    def policy(obs: np.ndarray) -> np.ndarray:
        s = torch.from_numpy(np.where(np.isfinite(obs), obs, 0).astype(np.float32))
        e = s[:, 3:-1]
        e = act(e @ W["dense_encoder.encoder_layers.0.weight"].T + W["dense_encoder.encoder_layers.0.bias"])
        e = act(e @ W["dense_encoder.encoder_layers.2.weight"].T + W["dense_encoder.encoder_layers.2.bias"])
        x = torch.cat([s[:, 0:4], e], 1)
        for i in (0, 2, 4):
            x = act(x @ W[f"mlp.{i}.weight"].T + W[f"mlp.{i}.bias"])
        return torch.tanh(x @ W["mlp.6.weight"].T + W["mlp.6.bias"]).numpy().astype(np.float32)

    return policy


def variant(name, o):
    o = o.copy()
    sc = o[:, 4:].reshape(-1, 31, 31)          # [row i (y offset), column j (x offset)], x fastest
    if name == "scan transposed (y fastest)":
        sc = sc.transpose(0, 2, 1)
    elif name == "scan flipped in x":
        sc = sc[:, :, ::-1]
    elif name == "scan flipped in y":
        sc = sc[:, ::-1, :]
    elif name == "heading sign flipped":
        o[:, 3] = -o[:, 3]
    elif name == "scan sign flipped":
        sc = -sc
    elif name == "scan zeroed":
        sc = sc * 0
    o[:, 4:] = np.ascontiguousarray(sc).reshape(-1, 961)
    return o


def run(n, steps, actor, t, seed=3):
    cfg = ro.default_config(seed_lo=seed)
    S = ro.new_state(n)
    obs = ro.reset_all(cfg, t, S)
    terms, log = np.zeros(4), np.zeros(16, np.float32)
    for _ in range(steps):
        obs, rew, te, tr, f, log = ro.step(cfg, t, S, actor(obs), log=log)
        if log[13] > 0:
            terms += log[7:11]
    return terms     # time_limit, success, far, collision


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    ter = T.make_procedural_terrain((2048, 2048))
    ter.make_spawns(2 * n)
    t = ro.TerrainData(ter.height, ter.obstacle, ter.safe_rock_mask, 0.05, ter.min_x, ter.min_y, ter.spawn_locations)
    pol = load_policy()
    rng = np.random.RandomState(0)
    rows = [("pretrained policy, observations as implemented", lambda o: pol(o)),
            ("random actions U(-1,1)", lambda o: rng.uniform(-1, 1, (n, 2)).astype(np.float32)),
            ("zero actions", lambda o: np.zeros((n, 2), np.float32))]
    for v in ("scan transposed (y fastest)", "scan flipped in x", "scan flipped in y", "heading sign flipped",
              "scan sign flipped", "scan zeroed"):
        rows.append((f"pretrained policy, {v}", (lambda vv: (lambda o: pol(variant(vv, np.where(np.isfinite(o), o, 0)))))(v)))
    print(f"{n} envs x {steps} steps, procedural terrain seed 1234; episodes ended by cause")
    for name, actor in rows:
        tl, su, far, col = run(n, steps, actor, t)
        tot = max(tl + su + far + col, 1)
        print(f"  {name:52s} success {su:4.0f}  far {far:4.0f}  collision {col:4.0f}  time_limit {tl:3.0f}  success rate {su / tot:.2f}")


if __name__ == "__main__":
    main()
