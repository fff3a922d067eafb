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


def _panda_lagrangian(q, params):
    """Independent float64 restatement of the arm's rigid-body dynamics from the SAME published parameters (modified-DH
    table, link masses / centres of mass / principal inertias -- data, restated here) by the Lagrangian route: link frames
    by homogeneous transforms, geometric Jacobians of every centre of mass, M = sum m Jv^T Jv + Jw^T (R I R^T) Jw and the
    potential energy.  Nothing of lift_model.h's recursive Newton-Euler code is shared."""
    A, D, alpha, m, Cm, Ii = params
    T = np.eye(4)
    Rs, os_, zs = [], [], []
    for i in range(7):
        ca, sa, ct, st = np.cos(alpha[i]), np.sin(alpha[i]), np.cos(q[i]), np.sin(q[i])
        Ti = np.array([[ct, -st, 0.0, A[i]],
                       [st * ca, ct * ca, -sa, -sa * D[i]],
                       [st * sa, ct * sa, ca, ca * D[i]],
                       [0.0, 0.0, 0.0, 1.0]])
        T = T @ Ti
        Rs.append(T[:3, :3].copy()); os_.append(T[:3, 3].copy()); zs.append(T[:3, 2].copy())
    M = np.zeros((7, 7))
    U = 0.0
    for i in range(7):
        c = os_[i] + Rs[i] @ Cm[i]
        Jv, Jw = np.zeros((3, 7)), np.zeros((3, 7))
        for j in range(i + 1):
            Jv[:, j] = np.cross(zs[j], c - os_[j])
            Jw[:, j] = zs[j]
        Iw = Rs[i] @ np.diag(Ii[i]) @ Rs[i].T
        M += m[i] * Jv.T @ Jv + Jw.T @ Iw @ Jw
        U += m[i] * 9.81 * c[2]
    return M, U


def test_arm_dynamics_match_an_independent_lagrangian_restatement(lo):
    """lm_rne (recursive Newton-Euler, float32) against a float64 Lagrangian restatement: mass matrix, gravity torque =
    dU/dq, and the velocity terms c(q, qd) = Mdot qd - 1/2 d(qd^T M qd)/dq, at random configurations."""
    params = (np.array([0.0, 0.0, 0.0, 0.0825, -0.0825, 0.0, 0.088]), np.array([0.333, 0.0, 0.316, 0.0, 0.384, 0.0, 0.0]),
              np.array([0.0, -np.pi / 2, np.pi / 2, np.pi / 2, -np.pi / 2, np.pi / 2, np.pi / 2]),
              np.array([4.97, 0.647, 3.228, 3.588, 1.226, 1.667, 1.495]),
              np.array([[0.0039, 0.0021, -0.0476], [-0.0031, -0.0287, 0.0035], [0.0275, 0.0392, -0.0665],
                        [-0.0532, 0.1044, 0.0275], [-0.0118, 0.0411, -0.0384], [0.0601, -0.0141, -0.0105],
                        [0.0054, -0.0021, 0.1050]]),
              np.array([[0.70, 0.71, 0.0091], [0.0080, 0.0281, 0.0260], [0.0372, 0.0362, 0.0108], [0.0259, 0.0196, 0.0283],
                        [0.0355, 0.0295, 0.0086], [0.0020, 0.0043, 0.0054], [0.0260, 0.0240, 0.0060]]))
    rng = np.random.RandomState(5)
    lo_q = np.array([-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973])
    hi_q = np.array([2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973])
    eps = 1e-5
    for _ in range(8):
        q = rng.uniform(lo_q, hi_q)
        qd = rng.uniform(-1.5, 1.5, 7)
        M, _ = _panda_lagrangian(q, params)
        assert np.abs(lo.mass_matrix(q) - M).max() < 2e-5 * max(1.0, np.abs(M).max())
        # gravity torque = gradient of the potential energy (central differences in float64)
        g = np.array([(_panda_lagrangian(q + eps * e, params)[1] - _panda_lagrangian(q - eps * e, params)[1]) / (2 * eps)
                      for e in np.eye(7)])
        assert np.abs(lo.gravity_torque(q) - g).max() < 2e-4
        # Coriolis / centrifugal torque from the derivatives of M
        dM = [(_panda_lagrangian(q + eps * e, params)[0] - _panda_lagrangian(q - eps * e, params)[0]) / (2 * eps)
              for e in np.eye(7)]
        Mdot = sum(dM[k] * qd[k] for k in range(7))
        c = Mdot @ qd - 0.5 * np.array([qd @ dM[k] @ qd for k in range(7)])
        c_model = lo.inverse_dynamics(q, qd, np.zeros(7), gravity=0.0)
        assert np.abs(c_model - c).max() < 5e-4, (c_model, c)
        # and the full inverse dynamics is linear in the acceleration: tau(qdd) - tau(0) = M qdd
        qdd = rng.uniform(-2, 2, 7)
        tau = lo.inverse_dynamics(q, qd, qdd) - lo.inverse_dynamics(q, qd, np.zeros(7))
        assert np.abs(tau - M @ qdd).max() < 2e-4
