"""The CPU oracle against the golden vectors produced by the reference's own torch / numpy code
(tools/gen_golden.py) and the known answers of SURVEY.md Appendix D."""
import numpy as np

from helpers import assert_close


def test_ackermann_golden(oracle, golden_dir):
    g = np.load(f"{golden_dir}/ackermann.npz")
    cfg = oracle.default_config()
    p, s, w = oracle.ackermann(cfg, g["raw"])
    assert_close(p, g["processed"], 0, 0, "processed")
    assert_close(w, g["wheel"], 0, 0, "wheel speeds (bit exact)")
    assert_close(s, g["steer"], 2e-6, 2e-6, "steer angles (atan2 rounding)")


def test_ackermann_known_answers(oracle):
    """steer [FL,RL,RR,FR], wheels [ML,FL,RL,RR,MR,FR]."""
    cfg = oracle.default_config()
    raw = np.array([[0, 0], [1, 0], [1, 1], [-1, 0.5], [0.2, -1], [0.0135, 0.0135], [0.5, 0.0135]], np.float32)
    p, s, w = oracle.ackermann(cfg, raw)
    assert_close(s[0], [-0.943894] * 4, 2e-6, 0, "zero action steer")
    assert_close(w[0], [-0.074655, -0.083025, -0.083025, -0.186975, -0.195345, -0.186975], 2e-6, 0)
    assert_close(w[1], [9.804655, 9.813025, 9.813025, 9.916975, 9.925345, 9.916975], 2e-6, 1e-6)
    assert_close(s[2], [0.943894] * 4, 2e-6, 0)
    assert_close(w[2], [5.455345, 6.066975, 6.066975, 13.663025, 14.274655, 13.663025], 2e-6, 1e-6)
    assert_close(s[3], [0.463589] * 4, 2e-6, 0)
    q = np.pi / 4
    assert_close(s[4], [-q, q, -q, q], 1e-7, 0, "point turn steer signs")
    assert_close(w[4], [11.865] * 3 + [-11.865] * 3, 2e-6, 1e-6, "point turn wheels")
    assert (s[5] == 0).all() and (w[5] == 0).all()
    assert (s[6] == 0).all()
    assert_close(w[6], [4.865] * 6, 2e-6, 1e-6)


def test_mdp_terms_golden(oracle, golden_dir):
    m = np.load(f"{golden_dir}/mdp_terms.npz")
    cfg = oracle.default_config()
    od, oa, rew, term = oracle.mdp_terms(cfg, m["cmd"], m["action"], m["prev_action"], m["episode_length_buf"],
                                         m["force_matrix_w"])
    assert_close(od, m["obs_distance"][:, 0], 2e-6, 2e-6, "distance_to_target_euclidean")
    assert_close(oa, m["obs_angle"][:, 0], 2e-6, 2e-6, "angle_to_target_observation")
    names = ["rew_distance_to_target", "rew_reached_target", "rew_oscillation", "rew_angle_to_target",
             "rew_heading_soft_contraint", "rew_collision", "rew_far_from_target"]
    for i, nm in enumerate(names):
        assert_close(rew[:, i], m[nm], 1e-9, 2e-6, nm)
    assert (term[:, 0] == (m["episode_length_buf"] >= 750)).all()
    assert (term[:, 1].astype(bool) == m["term_is_success"]).all()
    assert (term[:, 2].astype(bool) == m["term_far_from_target"]).all()
    assert (term[:, 3].astype(bool) == m["term_collision"]).all()
    # the reference ignores the `threshold` parameter of the collision terms (B-8)
    assert (m["rew_collision_thr100"] == m["rew_collision"]).all()


def test_height_scan_term_golden(oracle, golden_dir):
    m = np.load(f"{golden_dir}/mdp_terms.npz")
    hs = oracle.height_scan_term(oracle.default_config(), m["pos_w"][:, 2], m["ray_hits_z"])
    assert_close(hs, m["obs_height_scan"], 0, 0, "height_scan_rover")
    assert np.isinf(hs).sum() == 1 and hs[np.isinf(hs)][0] < 0          # a miss (+inf hit) gives -inf
    assert abs(hs[0, 0] - 9.93122) < 1e-5                                # App. D


def test_heightmap_lookups_golden(oracle, golden_dir):
    h = np.load(f"{golden_dir}/heightmap.npz")
    b = h["wavy_bounds"]
    t = oracle.TerrainData(h["wavy_heightmap"], safe_mask=h["wavy_mask"], min_x=np.float32(b[0]), min_y=np.float32(b[1]))
    assert_close(oracle.get_height_at(t, h["wavy_query_xy"]), h["wavy_query_height"], 0, 0, "get_height_at")
    assert (oracle.target_invalid(t, h["wavy_query_xy"]) == h["wavy_query_invalid"]).all()


def test_philox_known_answers(oracle):
    """Random123 kat_vectors for philox4x32-10."""
    assert oracle.philox(0, 0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert oracle.philox(f, f, f, f, f, f) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_update_command_against_float64(oracle):
    """ORBIT's yaw_quat + quat_rotate_inverse + wrap_to_pi restated (App. C): check against a float64 formula."""
    rng = np.random.RandomState(0)
    n = 4000
    yaw = rng.uniform(-np.pi, np.pi, n)
    roll, pitch = rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n)
    cr, sr, cp, sp, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
    quat = np.stack([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                     cr * cp * sy - sr * sp * cy], 1).astype(np.float32)
    pos = rng.uniform(20, 80, (n, 3)).astype(np.float32)
    tgt = (pos + rng.uniform(-12, 12, (n, 3))).astype(np.float32)
    hc = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    cb, hb = oracle.update_command(pos, quat, tgt, hc)
    q = quat.astype(np.float64)
    y = np.arctan2(2 * (q[:, 0] * q[:, 3] + q[:, 1] * q[:, 2]), 1 - 2 * (q[:, 2] ** 2 + q[:, 3] ** 2))
    d = tgt.astype(np.float64) - pos.astype(np.float64)
    ex = np.stack([np.cos(y) * d[:, 0] + np.sin(y) * d[:, 1], -np.sin(y) * d[:, 0] + np.cos(y) * d[:, 1], d[:, 2]], 1)
    assert_close(cb, ex, 2e-5, 2e-6, "pos_command_b")
    hw = np.arctan2(2 * (q[:, 0] * q[:, 3] + q[:, 1] * q[:, 2]), 1 - 2 * (q[:, 2] ** 2 + q[:, 3] ** 2))
    e = (hc.astype(np.float64) - hw + np.pi) % (2 * np.pi) - np.pi
    err = np.abs(((hb - e) + np.pi) % (2 * np.pi) - np.pi)
    assert err.max() < 5e-6
    assert (hb <= np.pi + 1e-6).all() and (hb > -np.pi - 1e-6).all()


def _reset_fixture_oracle_setup(oracle, g):
    cfg = oracle.default_config(max_target_tries=int(g["max_tries"]), scan_nx=3, scan_ny=3, scan_size_x=0.2, scan_size_y=0.2)
    ter = oracle.TerrainData(g["heightmap"], None, g["safe_mask"], float(g["resolution"]), float(g["min_xy"][0]),
                             float(g["min_xy"][1]), g["spawn_table"])
    return cfg, ter


def test_reset_golden(oracle, golden_dir):
    """reset_root_state_rover + sample_new_targets + the heading draw, with the reference's recorded torch draws injected
    (randomizations.py:12-39, terrain_importer.py:74-95, 134-175): the oracle reproduces pose / origin / target / heading."""
    from helpers import check_reset_against_fixture, reset_fixture_case
    g = np.load(f"{golden_dir}/reset.npz")
    cfg, ter = _reset_fixture_oracle_setup(oracle, g)
    n = int(g["num_envs"])
    S = oracle.new_state(n)
    S[:, oracle.ENV_ORIGIN:oracle.ENV_ORIGIN + 2] = 100.0        # rover_env.py:24-25, overwritten by the reset
    for batch in (0, 1):
        mask, row, yaw_u, theta_u, heading_u, expect = reset_fixture_case(g, batch)
        before = S.copy()
        oracle.reset_with_draws(cfg, ter, S, mask, row, yaw_u, theta_u, heading_u)
        check_reset_against_fixture(S, expect, oracle)
        untouched = mask == 0
        assert (S[untouched] == before[untouched]).all(), "envs outside env_ids must not change"
        # number of theta draws consumed = the reference's rejection rounds (safe rock mask, terrain_utils.py:202-223)
        tries = g[f"b{batch}_tries"]
        assert (np.isfinite(g[f"b{batch}_theta_u"]).sum(1) == tries).all()
    assert not np.isnan(S).any()
