"""GPU tests that compare the HIP path DIRECTLY with the fixtures generated from the reference's own torch code
(tools/gen_golden.py), through the C ABI -- no oracle in between:

  * rover_mdp_terms        vs tests/golden/mdp_terms.npz   (observations.py:15-32, rewards.py:14-137, terminations.py:14-64)
  * rover_reset_with_draws vs tests/golden/reset.npz       (randomizations.py:12-39, terrain_importer.py:74-95, 134-175)

Tolerances: the reference evaluates sqrt / atan2 / cos / sin / norm with torch (libm-grade kernels), the HIP path with its
explicit fp32 polynomial sequences: <= 2e-6 abs + 2e-6 rel on O(1) values; every comparison / flag / index is exact.
The oracle-vs-fixture half of the chain (tests/test_oracle_golden.py) is collected here too so that one `-m gpu` run holds
the whole fixture -> oracle -> HIP chain.
"""
import numpy as np
import pytest
import torch

import test_oracle_golden as _og
from helpers import assert_close, check_reset_against_fixture, flat, reset_fixture_case

pytestmark = pytest.mark.gpu

# ---- the fixture -> oracle half, re-collected under the gpu marker (same functions, same assertions)
for _name in dir(_og):
    if _name.startswith("test_"):
        globals()[f"test_oracle_{_name[5:]}"] = getattr(_og, _name)


def _small_env(n=64, **over):
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    from isaac_rover_orbit_amd.envs import RoverEnv
    ter = flat()
    cfg = RoverEnvCfg()
    cfg.scene.num_envs = n
    cfg.terrain.kind = "custom"
    for k, v in over.items():
        setattr(cfg, k, v)
    if ter.spawn_locations is None or ter.spawn_locations.shape[0] != 2 * n:
        ter.make_spawns(2 * n)
    return RoverEnv(cfg, terrain=ter)


def test_mdp_terms_match_reference_fixture(golden_dir):
    """The 13 term functions of the step kernel's tail vs the reference's outputs (256 rows incl. every threshold)."""
    m = np.load(f"{golden_dir}/mdp_terms.npz")
    env = _small_env()
    od, oa, rew, term = env.mdp_terms(m["cmd"], m["action"], m["prev_action"], m["episode_length_buf"], m["force_matrix_w"])
    od, oa, rew, term = od.cpu().numpy(), oa.cpu().numpy(), rew.cpu().numpy(), term.cpu().numpy()
    assert_close(od, m["obs_distance"][:, 0], 2e-6, 2e-6, "distance_to_target_euclidean")
    assert_close(oa, m["obs_angle"][:, 0], 2e-6, 2e-6, "angle_to_target_observation")
    names = ["rew_distance_to_target", "rew_reached_target", "rew_oscillation", "rew_angle_to_target",
             "rew_heading_soft_contraint", "rew_collision", "rew_far_from_target"]
    for i, nm in enumerate(names):
        assert_close(rew[:, i], m[nm], 1e-9, 2e-6, nm)
    assert (term[:, 0] == (m["episode_length_buf"] >= 750)).all(), "time_out"
    assert (term[:, 1] == m["term_is_success"]).all(), "is_success"
    assert (term[:, 2] == m["term_far_from_target"]).all(), "far_from_target"
    assert (term[:, 3] == m["term_collision"]).all(), "collision_with_obstacles"
    env.close()


def test_mdp_terms_entry_is_bit_exact_vs_oracle(oracle, golden_dir):
    m = np.load(f"{golden_dir}/mdp_terms.npz")
    env = _small_env()
    od, oa, rew, term = env.mdp_terms(m["cmd"], m["action"], m["prev_action"], m["episode_length_buf"], m["force_matrix_w"])
    ood, ooa, orew, oterm = oracle.mdp_terms(oracle.default_config(), m["cmd"], m["action"], m["prev_action"],
                                             m["episode_length_buf"], m["force_matrix_w"])
    assert_close(od.cpu().numpy(), ood, 0, 0, "distance")
    assert_close(oa.cpu().numpy(), ooa, 0, 0, "angle")
    assert_close(rew.cpu().numpy(), orew, 0, 0, "rewards")
    assert (term.cpu().numpy() == oterm.astype(bool)).all()
    env.close()


def _reset_fixture_env(g):
    from isaac_rover_orbit_amd import _lib
    from isaac_rover_orbit_amd import terrain as T
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    from isaac_rover_orbit_amd.envs import RoverEnv
    n = int(g["num_envs"])
    H, W = g["heightmap"].shape
    ter = T.Terrain(ground=g["heightmap"].astype(np.float32), obstacle=np.zeros((H, W), np.float32),
                    rock_mask=g["safe_mask"].astype(np.uint8), safe_rock_mask=g["safe_mask"].astype(np.uint8),
                    resolution=float(g["resolution"]), min_x=float(g["min_xy"][0]), min_y=float(g["min_xy"][1]))
    ter.spawn_locations = g["spawn_table"].astype(np.float32)
    cfg = RoverEnvCfg()
    cfg.scene.num_envs = n
    cfg.terrain.kind = "custom"
    cfg.commands.max_target_tries = int(g["max_tries"])
    env = RoverEnv(cfg, terrain=ter)
    assert env._native_cfg.max_target_tries == int(g["max_tries"])
    return env, _lib


def test_reset_matches_reference_fixture(golden_dir):
    """reset_root_state_rover + sample_new_targets + heading draw with the reference's recorded torch draws injected:
    pose, env origin, target and heading command of the HIP reset kernel vs the reference's outputs (two batches: all envs,
    then a scattered subset -- the others must not change)."""
    g = np.load(f"{golden_dir}/reset.npz")
    env, words = _reset_fixture_env(g)
    env.state[words.ENV_ORIGIN:words.ENV_ORIGIN + 2] = 100.0      # rover_env.py:24-25, overwritten by the reset
    for batch in (0, 1):
        mask, row, yaw_u, theta_u, heading_u, expect = reset_fixture_case(g, batch)
        before = env.get_state().cpu().numpy()
        obs, _ = env.reset_with_draws(mask, row, yaw_u, theta_u, heading_u)
        S = env.get_state().cpu().numpy()
        check_reset_against_fixture(S, expect, words)
        assert (S[mask == 0] == before[mask == 0]).all(), "envs outside env_ids must not change"
        assert not np.isnan(S).any(), "a NaN means more theta draws were consumed than the reference made"
        ids = expect["ids"]
        assert (S[ids][:, words.EP_LEN].view(np.int32) == 0).all()
        assert (S[ids][:, words.ACTION:words.ACTION + 4] == 0).all(), "action manager reset"
        o = obs["policy"].cpu().numpy()
        assert np.isfinite(o[:, :4]).all()
    env.close()


def test_reset_with_draws_is_bit_exact_vs_oracle(oracle, golden_dir):
    g = np.load(f"{golden_dir}/reset.npz")
    env, words = _reset_fixture_env(g)
    from helpers import oracle_config_from, oracle_terrain
    ocfg = oracle_config_from(oracle, env._native_cfg)
    oter = oracle_terrain(oracle, env.terrain_data)
    So = env.get_state().cpu().numpy().astype(np.float32).copy()
    for batch in (0, 1):
        mask, row, yaw_u, theta_u, heading_u, _ = reset_fixture_case(g, batch)
        obs, _ = env.reset_with_draws(mask, row, yaw_u, theta_u, heading_u)
        oo = oracle.reset_with_draws(ocfg, oter, So, mask, row, yaw_u, theta_u, heading_u)
        assert_close(env.get_state().cpu().numpy(), So, 0, 0, "state after injected reset")
        assert_close(obs["policy"].cpu().numpy(), oo, 0, 0, "observation after injected reset")
    env.close()


def test_seed_rekeys_the_resets():
    """env.seed(s) / reset(seed=s) (gymnasium contract) change the Philox key of the resets that follow."""
    a, b = _small_env(64), _small_env(64)
    a.reset()
    b.reset()
    assert torch.equal(a.get_state(), b.get_state())
    b.reset(seed=7)
    a.reset()
    assert not torch.equal(a.get_state()[:, :3], b.get_state()[:, :3]), "a new seed must change the spawn draws"
    a.seed(7)
    a.set_state(torch.zeros_like(a.get_state()))
    b.set_state(torch.zeros_like(b.get_state()))
    a.reset()
    b.reset()
    assert torch.equal(a.get_state(), b.get_state()), "same seed, same counters => same draws"
    c = _small_env(64, seed=7)
    c.set_call_counter(a.call_counter - 1)          # the per-batch spawn permutation is keyed by (seed, call counter)
    c.reset()
    assert torch.equal(c.get_state()[:, :7], a.get_state()[:, :7]), "seed() == cfg.seed"
    for e in (a, b, c):
        e.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_env_on_second_device_while_first_is_current(oracle):
    """The C ABI launches on the handle's device whatever the thread's current device is."""
    from helpers import small_procedural
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    from isaac_rover_orbit_amd.envs import RoverEnv
    ter = small_procedural()
    ter.make_spawns(2 * 128)
    envs = []
    for dev in ("cuda:0", "cuda:1"):
        cfg = RoverEnvCfg()
        cfg.scene.num_envs = 128
        cfg.terrain.kind = "custom"
        cfg.sim.device = dev
        envs.append(RoverEnv(cfg, terrain=ter))
    torch.cuda.set_device(0)
    rng = np.random.RandomState(0)
    for e in envs:
        e.reset()
    for _ in range(4):
        a = rng.uniform(-1, 1, (128, 2)).astype(np.float32)
        outs = [e.step(torch.from_numpy(a).to(e.device)) for e in envs]
        assert torch.equal(outs[0][0]["policy"].cpu(), outs[1][0]["policy"].cpu())
        assert torch.equal(outs[0][1].cpu(), outs[1][1].cpu())
    for e in envs:
        e.close()
