#!/usr/bin/env python3
"""Dynamics-fidelity study, oracle only (round-3 review item 5; build container: needs /root/reference for the checkpoint).

Variants of the rover model, each scored by (i) the one frozen artefact of the reference that exercises dynamics -- the
Isaac-Sim-trained policy ``best_agent.pt`` driven closed loop (``tools/policy_closed_loop.py``) -- and (ii) analytic checks on a
flat map: rest height, straight-line speed, steady response to (lin, ang) = (1, 1): speed, crab angle, yaw rate.

    python tools/dynamics_study.py [num_envs] [steps]  > profiles/r05_dynamics_study.txt
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from isaac_rover_orbit_amd import terrain as T  # noqa: E402
from oracle import rover_oracle as ro  # noqa: E402
import policy_closed_loop as pcl  # noqa: E402

VARIANTS = [  # name, study-variant bits (rvo_set_model_variant), solver iterations, friction, cfg.mass_model
    ("PRODUCT (round 5): subtree weights on the bogie coordinates (cfg.mass_model = 1), split-mass Jacobi, 32 iterations", 0, 32, 0.75, 1),
    ("lumped mass (cfg.mass_model = 0), 16 iterations [round 4's product apart from the chassis split factor: profiles/r05_dynamics_study_split6.txt]", 0, 16, 0.75, 0),
    ("lumped mass, 32 iterations", 0, 32, 0.75, 0),
    ("subtree weights, 16 iterations", 0, 16, 0.75, 1),
    ("subtree weights, 64 iterations", 0, 64, 0.75, 1),
    ("subtree weights + wheel contact on the triangle surface the rays see, 32 iterations", 1, 32, 0.75, 1),
    ("subtree weights, 32 iterations, friction 1.0", 0, 32, 1.0, 1),
    ("(d) sequential PGS on the DIAGONAL mass matrix (no coupling), lumped, 32 iterations", 4, 32, 0.75, 0),
    ("(b) coupled 9x9 mass matrix + sequential PGS (double precision, study only), 32 iterations", 2, 32, 0.75, 0),
    ("(b) coupled, 32 iterations, friction 1.0", 2, 32, 1.0, 0),
]


def flat_checks(iters, mu, mass_model):
    ter = T.make_flat_terrain((512, 512))
    ter.spawn_locations = np.array([[12.8, 12.8, 0.0]] * 4, np.float32)
    t = ro.TerrainData(ter.height, ter.obstacle, ter.safe_rock_mask, 0.05, ter.min_x, ter.min_y, ter.spawn_locations)
    cfg = ro.default_config(seed_lo=1)
    cfg.solver_iterations, cfg.friction_mu, cfg.far_threshold, cfg.success_threshold = iters, mu, 1e9, -1.0
    cfg.mass_model, cfg.rew_far_threshold, cfg.rew_success_threshold = mass_model, 1e9, -1.0
    out = {}
    for name, act in (("straight", (1.0135, 0.0135)), ("turn11", (1.0, 1.0))):
        S = ro.new_state(1)
        ro.reset_all(cfg, t, S)
        S[0, 3:7] = (1, 0, 0, 0)
        a = np.array([act], np.float32)
        hist = []
        for k in range(60):
            ro.step(cfg, t, S, a)
            q = S[0, 3:7]
            yaw = np.arctan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] ** 2 + q[3] ** 2))
            hist.append((S[0, 0], S[0, 1], S[0, 2], yaw, S[0, 7], S[0, 8]))
        h = np.array(hist)
        v = np.hypot(h[-1, 4], h[-1, 5])
        out[name] = {"speed": float(v), "z": float(h[-1, 2]),
                     "crab_deg": float(np.degrees((np.arctan2(h[-1, 5], h[-1, 4]) - h[-1, 3] + np.pi) % (2 * np.pi) - np.pi)),
                     "yaw_rate": float((np.unwrap(h[:, 3])[-1] - np.unwrap(h[:, 3])[-11]) / 2.0)}
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    ter = T.make_procedural_terrain((2048, 2048))
    ter.make_spawns(2 * n)
    t = ro.TerrainData(ter.height, ter.obstacle, ter.safe_rock_mask, 0.05, ter.min_x, ter.min_y, ter.spawn_locations)
    pol = pcl.load_policy()
    print(f"dynamics study: {n} envs x {steps} steps of best_agent.pt (deterministic mean action) on the procedural terrain, per model variant;")
    print("flat-map checks: rest height z (body origin; the reference's constant is 0.26878 above the contact plane), straight-line speed at lin = 1 (m/s, rim speed cap "
          "6 rad/s x 0.1018 m = 0.61), steady response to (lin, ang) = (1, 1): speed, crab angle (velocity vs heading), yaw rate\n")
    for name, bits, iters, mu, mass_model in VARIANTS:
        ro.set_model_variant(bits)
        t0 = time.time()
        cfg_terms = None
        # closed loop
        cfg = ro.default_config(seed_lo=3)
        cfg.solver_iterations, cfg.friction_mu, cfg.mass_model = iters, mu, mass_model
        S = ro.new_state(n)
        obs = ro.reset_all(cfg, t, S)
        terms, log = np.zeros(4), np.zeros(16, np.float32)
        for _ in range(steps):
            obs, rew, te, tr, f, log = ro.step(cfg, t, S, pol(obs), log=log)
            if log[13] > 0:
                terms += log[7:11]
        tl, su, far, col = terms
        tot = max(terms.sum(), 1)
        fc = flat_checks(iters, mu, mass_model)
        print(f"{name}")
        print(f"    policy closed loop: success {su:4.0f}  far {far:4.0f}  collision {col:4.0f}  time_limit {tl:3.0f}  success rate {su / tot:.2f}")
        print(f"    flat map: rest z {fc['straight']['z']:.4f}  straight speed {fc['straight']['speed']:.3f} (crab {fc['straight']['crab_deg']:+.1f} deg)   "
              f"(1, 1): speed {fc['turn11']['speed']:.3f}  crab {fc['turn11']['crab_deg']:+.1f} deg  yaw rate {fc['turn11']['yaw_rate']:+.3f} rad/s   [{time.time() - t0:.0f} s]")
        sys.stdout.flush()
    ro.set_model_variant(0)


if __name__ == "__main__":
    main()
