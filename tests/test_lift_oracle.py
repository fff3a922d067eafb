"""FrankaCubeLift-v0, CPU side: the oracle's independent restatement of the reference's term functions against
the fixture generated from the reference's own torch code, and plausibility of the reduced arm / cube / gripper model (no
physics oracle exists: PhysX; parity of that layer is unpinned)."""
import numpy as np
import pytest

from helpers import assert_close


@pytest.fixture(scope="module")
def lo():
    from oracle import lift_oracle
    lift_oracle.build()
    lift_oracle.lib()
    return lift_oracle


def test_lift_terms_golden(lo, golden_dir):
    """rewards.py:20-67 (object_is_lifted, object_ee_distance, object_goal_distance x 2 stds) and observations.py:19-31
    (object_position_in_robot_root_frame): 192 rows incl. the 0.06 m threshold from both sides and arbitrary root poses."""
    g = np.load(f"{golden_dir}/lift_terms.npz")
    lifted, reach, goal, fine, pos_b = lo.terms(g["object_pos_w"], g["ee_pos_w"], g["robot_root_state_w"], g["command"])
    assert_close(lifted, g["rew_object_is_lifted"], 0, 0, "object_is_lifted")
    assert_close(reach, g["rew_object_ee_distance"], 3e-7, 2e-6, "object_ee_distance (tanh kernel)")
    assert_close(goal, g["rew_object_goal_distance_03"], 3e-7, 2e-6, "object_goal_distance std 0.3")
    assert_close(fine, g["rew_object_goal_distance_005"], 3e-7, 2e-6, "object_goal_distance std 0.05")
    assert_close(pos_b, g["obs_object_position_in_robot_root_frame"], 3e-7, 2e-6, "object_position_in_robot_root_frame")
    assert lifted[:3].tolist() == [0.0, 1.0, 0.0]                 # z = 0.06 (not >), 0.0600001, 0.0599999


def test_tanh_and_model_kinematics(lo):
    x = np.linspace(-9, 9, 4001).astype(np.float32)
    t = np.array([lo.lib().lfo_tanhf(float(v)) for v in x], np.float32)
    assert np.abs(t - np.tanh(x.astype(np.float64))).max() < 2e-7
    cfg = lo.default_config()
    tcp, R, v, w = lo.hand_pose(cfg, lo.Q_DEFAULT)
    assert abs(tcp[1]) < 1e-6 and 0.40 < tcp[0] < 0.52 and 0.30 < tcp[2] < 0.48      # home pose: in front of the base, above the table
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and abs(np.linalg.det(R) - 1) < 1e-5
    # hand twist = numerical derivative of the pose
    rng = np.random.RandomState(0)
    q = lo.Q_DEFAULT.copy()
    q[:7] += rng.uniform(-0.3, 0.3, 7)
    qd = np.zeros(9, np.float32)
    qd[:7] = rng.uniform(-1, 1, 7)
    tcp0, R0, v, w = lo.hand_pose(cfg, q, qd)
    eps = 1e-3
    tcp1, R1, _, _ = lo.hand_pose(cfg, q + eps * qd)
    assert np.allclose((tcp1 - tcp0) / eps, v, atol=3e-3)
    dR = (R1 - R0) / eps @ R0.T
    assert np.allclose([dR[2, 1], dR[0, 2], dR[1, 0]], w, atol=5e-3)


def test_arm_dynamics_are_consistent(lo):
    """Mass matrix symmetric positive definite; gravity torque = gradient of the potential energy (checked through the
    torque-free equilibrium droop: stiffness 80 N m / rad holds the home pose within tau_g / 80)."""
    rng = np.random.RandomState(1)
    for _ in range(5):
        q = (lo.Q_DEFAULT[:7] + rng.uniform(-0.5, 0.5, 7)).astype(np.float32)
        M = lo.mass_matrix(q)
        assert np.abs(M - M.T).max() < 1e-5 and np.linalg.eigvalsh((M + M.T) / 2).min() > 1e-3
    cfg = lo.default_config()
    S = lo.new_state(2)
    lo.reset(cfg, S)
    a = np.zeros((2, 8), np.float32)
    for _ in range(150):
        lo.step(cfg, S, a)
    droop = S[0, :7] - lo.Q_DEFAULT[:7]
    tau_g = lo.gravity_torque(S[0, :7])
    assert np.abs(S[0, 9:16]).max() < 0.02, "arm at rest"
    assert np.allclose(80.0 * (-droop), tau_g, atol=0.6), (droop, tau_g)      # PD torque balances gravity
    assert np.abs(droop).max() < 0.35


def test_cube_rests_slides_and_is_grasped(lo):
    cfg = lo.default_config()
    S = lo.new_state(1)
    lo.reset(cfg, S)
    a = np.zeros((1, 8), np.float32)
    for _ in range(60):
        lo.step(cfg, S, a)
    assert abs(S[0, lo.OBJ_POS + 2] - 0.02) < 5e-4 and np.abs(S[0, lo.OBJ_LIN:lo.OBJ_LIN + 6]).max() < 1e-2   # at rest on the table
    # Coulomb friction: a horizontal push decays at mu g
    S[0, lo.OBJ_LIN] = 0.6
    x0 = S[0, lo.OBJ_POS]
    for _ in range(25):
        lo.step(cfg, S, a)
    travelled = S[0, lo.OBJ_POS] - x0
    assert abs(travelled - 0.6 ** 2 / (2 * cfg.mu_table * 9.81)) < 0.012 and abs(S[0, lo.OBJ_LIN]) < 1e-2
    # grasp: put the cube between the open fingers at the tool centre point, close, it stays in the hand
    S2 = lo.new_state(1)
    lo.reset(cfg, S2)
    for _ in range(100):
        lo.step(cfg, S2, a)
    tcp, R, _, _ = lo.hand_pose(cfg, S2[0, :9])
    S2[0, lo.OBJ_POS:lo.OBJ_POS + 3] = tcp
    S2[0, lo.OBJ_LIN:lo.OBJ_LIN + 6] = 0
    S2[0, 7:9] = 0.0205                           # pads just off the faces (closing takes 0.1 s at 0.2 m/s: the cube would drop)
    close = a.copy()
    close[0, 7] = -1.0
    for _ in range(100):
        lo.step(cfg, S2, close)
    tcp2, _, _, _ = lo.hand_pose(cfg, S2[0, :9])
    assert np.linalg.norm(S2[0, lo.OBJ_POS:lo.OBJ_POS + 3] - tcp2) < 0.012, "the cube is held between the pads"
    assert S2[0, 7] < 0.03 and S2[0, 8] < 0.03 and S2[0, 7] > 0.012        # fingers stopped by the cube (half width 0.02)
    assert S2[0, lo.OBJ_POS + 2] > 0.2
    # without closing it falls to the table
    S3 = S2.copy()
    S3[0, 7:9] = 0.04
    for _ in range(100):
        lo.step(cfg, S3, a)
    assert S3[0, lo.OBJ_POS + 2] < 0.05


def test_lift_mdp_ordering_and_resets(lo):
    cfg = lo.default_config()
    n = 64
    S = lo.new_state(n)
    obs = lo.reset(cfg, S)
    assert obs.shape == (n, 36) and np.abs(obs[:, :18]).max() == 0          # default pose, zero velocities
    assert np.allclose(obs[:, 18:21], S[:, lo.OBJ_POS:lo.OBJ_POS + 3]) and (obs[:, 24] == 1).all()
    assert (S[:, lo.OBJ_POS] >= 0.4 - 1e-6).all() and (S[:, lo.OBJ_POS] <= 0.6 + 1e-6).all()
    assert (np.abs(S[:, lo.OBJ_POS + 1]) <= 0.25 + 1e-6).all() and (S[:, lo.CMD] >= 0.3).all() and (S[:, lo.CMD] <= 0.7).all()
    rng = np.random.RandomState(0)
    log = np.zeros(16, np.float32)
    trunc_seen = 0
    for k in range(252):
        a = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
        obs, r, term, trunc, log = lo.step(cfg, S, a, log=log)
        assert np.isfinite(obs).all() and np.isfinite(r).all()
        trunc_seen += int(trunc.sum())
        if k < 249:
            assert trunc.sum() == 0
            assert np.array_equal(obs[~term.astype(bool), 28:36], a[~term.astype(bool)])       # last_action of non-reset envs
    assert trunc_seen >= n - 8                                    # 5 s episodes: everybody times out at step 250
    assert log[8] >= 0 and np.isfinite(log[:8]).all()
    ep = S[:, lo.EP_LEN].view(np.int32)
    assert ep.max() <= 2
