"""Plausibility of the reduced rover model (no physics oracle exists for PhysX: SURVEY section 4, item 3).
Runs on the CPU oracle; the GPU tests show the HIP kernels reproduce the oracle bit for bit."""
import numpy as np
import pytest

from helpers import flat, oracle_terrain, small_procedural

H = 1.0 / 30.0


def settle(ro, cfg, t, S, n=45):
    z4, z6 = np.zeros((S.shape[0], 4), np.float32), np.zeros((S.shape[0], 6), np.float32)
    ro.physics_step(cfg, t, S, z4, z6, n)
    return S


def fresh(ro, n=1, xy=(25.6, 25.6), z=0.5):
    S = ro.new_state(n)
    S[:, ro.POS:ro.POS + 3] = [xy[0], xy[1], z]
    return S


def link_table_wheel_loads(golden_dir):
    """Static wheel loads (N) of the articulated rover on flat ground from the USD link table alone (tests/golden/rover_model.json:
    per-link masses and centres of mass, bogie pivots, wheel centres) -- no solver involved.  Frictionless statics are determinate:
    vertical force, pitch and roll about the total centre of mass, and one moment balance per bogie about its pivot axis (the
    subtree's weight at ITS centre of mass against the two wheels' loads).  Order FL, FR, CL, CR, RL, RR."""
    import json
    import os
    m = json.load(open(os.path.join(golden_dir, "rover_model.json")))
    g = 9.81
    links = m["links"]
    mass = sum(l["mass"] for l in links.values())
    com = sum(l["mass"] * np.array(l["com"]) for l in links.values()) / mass
    names = ["FL", "FR", "CL", "CR", "RL", "RR"]
    wc = np.array([m["wheel_centres"][k] for k in names])
    A, b = [], []
    A.append(np.ones(6)); b.append(mass * g)                                    # vertical force
    A.append(wc[:, 0] - com[0]); b.append(0.0)                                  # pitch about the centre of mass
    A.append(wc[:, 1] - com[1]); b.append(0.0)                                  # roll
    for bogie, sub in m["subtrees"].items():
        piv, ax = np.array(m["bogies"][bogie]["pivot"]), np.array(m["bogies"][bogie]["axis"])
        ms = sum(links[l]["mass"] for l in sub["links"])
        cs = sum(links[l]["mass"] * np.array(links[l]["com"]) for l in sub["links"]) / ms
        row = np.zeros(6)
        for i, k in enumerate(names):
            if m["wheel_bogie"][k] == bogie:
                row[i] = np.dot(ax, np.cross(wc[i] - piv, [0.0, 0.0, 1.0]))     # moment of a unit load at the wheel about the axis
        A.append(row); b.append(np.dot(ax, np.cross(cs - piv, [0.0, 0.0, ms * g])))
    return np.linalg.solve(np.array(A), np.array(b)), mass


def test_static_rest_height_and_load_sharing(oracle, golden_dir):
    """The static wheel-load split is what the reference asset's link table says: 23 of the 25 kg sit in the
    three bogie subtrees, the front ones' centres of mass ahead of their pivots -- the front wheels carry MORE than the centre
    wheels (centre : front = 0.74 : 1).  Rounds 1-4 lumped the mass at the chassis (cfg.mass_model = 0: 1.96 : 1 by lever arms alone)."""
    ro = oracle
    expect, mass = link_table_wheel_loads(golden_dir)
    assert abs(mass - 25.0) < 1e-9 and 0.70 < expect[2] / expect[0] < 0.78
    t = oracle_terrain(ro, flat(), 2)
    # frictionless: the normal loads are statically determinate -> the solver must land on the link table's figures
    S = settle(ro, ro.default_config(friction_mu=0.0), t, fresh(ro), 300)
    lam = S[0, ro.LAMBDA_N:ro.LAMBDA_N + 6] / H
    assert np.abs(lam - expect).max() < 0.25, (lam, expect)                      # N, of 30 .. 45 N per wheel
    assert abs(lam[2] / lam[0] - expect[2] / expect[0]) < 0.005
    # the lumped model of rounds 1-4, for the record: the lever arms of a massless bogie
    S0 = settle(ro, ro.default_config(friction_mu=0.0, mass_model=0), t, fresh(ro), 300)
    lam0 = S0[0, ro.LAMBDA_N:ro.LAMBDA_N + 6] / H
    assert 1.9 < lam0[2] / lam0[0] < 2.0
    # default configuration (friction 0.75, braked wheels): friction pre-stress between the axles shifts the split a little
    cfg = ro.default_config()
    assert cfg.mass_model == 1 and cfg.solver_iterations == 32                   # aau_rover_simple.py:33
    S = settle(ro, cfg, t, fresh(ro))
    # body origin rests 0.26878 m above the contact plane (observations.py:45) => flat-ground height scan == 0
    assert abs(S[0, ro.POS + 2] - 0.26878) < 2e-4
    lam = S[0, ro.LAMBDA_N:ro.LAMBDA_N + 6] / H
    assert abs(lam.sum() - 25.0 * 9.81) < 0.5                                 # wheels carry the weight
    assert abs(lam[0] - lam[1]) < 1.0 and abs(lam[2] - lam[3]) < 1.0 and abs(lam[4] - lam[5]) < 1.0   # left/right symmetric
    assert abs(lam[2] / lam[0] - expect[2] / expect[0]) < 0.12
    assert np.abs(S[0, ro.BOGIE_Q:ro.BOGIE_Q + 3]).max() < 1e-3 and np.abs(S[0, ro.LINVEL:ro.ANGVEL + 3]).max() < 1e-3
    scan = ro.height_scan(cfg, t, S)
    assert np.abs(scan).max() < 3e-4


def test_straight_line_speed_is_omega_r(oracle):
    ro = oracle
    cfg, t = ro.default_config(), oracle_terrain(ro, flat(), 2)
    S = settle(ro, cfg, t, fresh(ro))
    w = np.full((1, 6), 4.0, np.float32)
    ro.physics_step(cfg, t, S, np.zeros((1, 4), np.float32), w, 60)
    x0 = S[0, 0]
    ro.physics_step(cfg, t, S, np.zeros((1, 4), np.float32), w, 90)
    v = (S[0, 0] - x0) / (90 * H)
    omega = S[0, ro.WHEEL_QD:ro.WHEEL_QD + 6].mean()
    assert abs(omega - 4.0) < 0.2                                               # stiff velocity drive (kd 4000)
    assert abs(v - omega * 0.10179) < 0.01 and abs(S[0, 1] - 25.6) < 0.02       # no slip, no drift
    # joint velocity limit 6 rad/s (aau_rover_simple.py:52): a 10 rad/s target saturates
    ro.physics_step(cfg, t, S, np.zeros((1, 4), np.float32), np.full((1, 6), 10.0, np.float32), 60)
    assert S[0, ro.WHEEL_QD:ro.WHEEL_QD + 6].max() <= 6.0 + 1e-6
    assert np.linalg.norm(S[0, ro.LINVEL:ro.LINVEL + 2]) < 6.0 * 0.10179 + 0.02


def test_point_turn_spins_in_place(oracle):
    ro = oracle
    cfg, t = ro.default_config(), oracle_terrain(ro, flat(), 2)
    S = settle(ro, cfg, t, fresh(ro))
    q = np.pi / 4
    steer = np.array([[-q, q, q, -q]], np.float32)                   # model order FL, FR, RL, RR (X pattern)
    wheel = np.array([[-3, 3, -3, 3, -3, 3]], np.float32)            # left backwards, right forwards => CCW
    ro.physics_step(cfg, t, S, steer, wheel, 150)
    assert S[0, ro.ANGVEL + 2] > 0.2                                  # yaw rate positive (counter-clockwise)
    assert np.linalg.norm(S[0, 0:2] - 25.6) < 0.15                    # stays in place
    assert np.allclose(S[0, ro.STEER_Q:ro.STEER_Q + 4], steer[0], atol=2e-2)


@pytest.mark.parametrize("R", [1.5, 3.0, -2.0])
def test_steady_state_turn_radius(oracle, R):
    """on flat ground a kinematically consistent Ackermann command (every wheel tangent to its circle
    about a common centre on the centre-axle line, rim speeds proportional to the circle radii) is followed without slip:
    the measured radius v / yaw_rate equals the commanded one.  (AckermannAction2 itself does NOT command this -- it gives all
    four steer joints the SAME angle, quirk B-5; that kinematics is pinned by the fixture, not by this plausibility test.)"""
    ro = oracle
    cfg, t = ro.default_config(), oracle_terrain(ro, flat(), 2)
    S = settle(ro, cfg, t, fresh(ro))
    mc = ro.model_constants()
    wheels = mc[7:25].reshape(6, 3)                 # FL, FR, CL, CR, RL, RR in the body frame (x forward, y left)
    r_contact = float(mc[46])              # wheel contact radius 0.10179
    xc = float(wheels[2, 0])                        # centre axle
    v_cmd = 0.35
    yaw_rate = v_cmd / R
    steer = np.zeros((1, 4), np.float32)
    wheel = np.zeros((1, 6), np.float32)
    for k, si in ((0, 0), (1, 1), (4, 2), (5, 3)):  # steerable wheels -> steer joints FL, FR, RL, RR
        steer[0, si] = np.arctan2(wheels[k, 0] - xc, R - wheels[k, 1]) if R > 0 else np.arctan2(-(wheels[k, 0] - xc), -(R - wheels[k, 1]))
    for k in range(6):
        rho = np.hypot(wheels[k, 0] - xc, R - wheels[k, 1])
        wheel[0, k] = abs(yaw_rate) * rho / r_contact
    ro.physics_step(cfg, t, S, steer, wheel, 150)                 # reach the steady state
    p0, q0 = S[0, 0:2].copy(), S[0, ro.QUAT:ro.QUAT + 4].copy()
    n = 120
    ro.physics_step(cfg, t, S, steer, wheel, n)
    yaw = lambda q: 2.0 * np.arctan2(q[3], q[0])                  # noqa: E731  (flat ground: pure yaw)
    dyaw = (yaw(S[0, ro.QUAT:ro.QUAT + 4]) - yaw(q0) + np.pi) % (2 * np.pi) - np.pi
    speed = np.linalg.norm(S[0, ro.LINVEL:ro.LINVEL + 2])
    rate = dyaw / (n * H)
    assert np.sign(rate) == np.sign(R)
    assert abs(speed / abs(rate) - abs(R)) < 0.08 * abs(R), (speed, rate, R)       # radius within 8 %
    assert abs(speed - v_cmd) < 0.05 * v_cmd + 0.01                                # body-origin speed = commanded
    chord = np.linalg.norm(S[0, 0:2] - p0)
    assert abs(chord - 2 * abs(R) * abs(np.sin(dyaw / 2))) < 0.05 * chord + 0.01   # the path is that circle's chord
    assert abs(S[0, ro.POS + 2] - 0.26878) < 2e-3                                  # stays on the ground


def test_steering_joint_is_first_order_lag(oracle):
    """kp 8000 / kd 1000 => time constant ~ kd / kp = 0.125 s, rate limit 6 rad/s (aau_rover_simple.py:43-49)."""
    ro = oracle
    cfg, t = ro.default_config(), oracle_terrain(ro, flat(), 2)
    S = settle(ro, cfg, t, fresh(ro))
    steer = np.full((1, 4), 0.5, np.float32)
    ro.physics_step(cfg, t, S, steer, np.zeros((1, 6), np.float32), 4)     # 0.133 s ~ one time constant
    q = S[0, ro.STEER_Q]
    assert 0.25 < q < 0.42
    ro.physics_step(cfg, t, S, steer, np.zeros((1, 6), np.float32), 40)
    assert abs(S[0, ro.STEER_Q] - 0.5) < 2e-3
    assert np.abs(S[0, ro.STEER_QD:ro.STEER_QD + 4]).max() <= 6.0


def test_energy_does_not_grow_on_flat_ground(oracle):
    ro = oracle
    cfg, t = ro.default_config(), oracle_terrain(ro, flat(), 2)
    S = settle(ro, cfg, t, fresh(ro))
    S[0, ro.LINVEL:ro.LINVEL + 2] = [0.5, 0.2]
    S[0, ro.ANGVEL + 2] = 0.4
    S[0, ro.WHEEL_QD:ro.WHEEL_QD + 6] = 0.0
    z4, z6 = np.zeros((1, 4), np.float32), np.zeros((1, 6), np.float32)
    ke0 = 0.5 * 25 * (0.5 ** 2 + 0.2 ** 2) + 0.5 * 6.15 * 0.4 ** 2      # 4.1 J
    ke_prev = np.inf
    for _ in range(30):
        ro.physics_step(cfg, t, S, z4, z6, 3)
        v, w = S[0, ro.LINVEL:ro.LINVEL + 3], S[0, ro.ANGVEL:ro.ANGVEL + 3]
        ke = 0.5 * 25 * (v @ v) + 0.5 * 6.15 * w[2] ** 2
        # braking pitches the weighted bogies by ~0.8 deg (cfg.mass_model = 1); what they give back while they settle is the
        # only rebound: < 0.05 % of the initial kinetic energy
        assert ke <= ke_prev + 5e-4 * ke0
        ke_prev = ke
    assert ke_prev < 1e-3                                              # braked wheels stop the rover


def test_slope_parking_and_friction_limit(oracle):
    """With braked wheels the rover holds on a slope below atan(mu) and slides on a steeper one."""
    ro = oracle
    from isaac_rover_orbit_amd import terrain as T
    for slope, holds in ((0.3, True), (1.2, False)):
        Hh = W = 600
        x = np.arange(W, dtype=np.float32) * 0.05
        g = np.tile(slope * x, (Hh, 1)).astype(np.float32)
        zero = np.zeros((Hh, W), np.uint8)
        ter = T.Terrain(ground=g, obstacle=np.zeros_like(g), rock_mask=zero, safe_rock_mask=zero)
        ter.spawn_locations = np.array([[15.0, 15.0, 15.0 * slope]], np.float32)
        cfg, t = ro.default_config(), oracle_terrain(ro, ter)
        S = fresh(ro, 1, (15.0, 15.0), 15.0 * slope + 0.45)
        settle(ro, cfg, t, S, 30)
        x0 = S[0, 0]
        settle(ro, cfg, t, S, 60)
        moved = abs(S[0, 0] - x0)
        assert (moved < 0.02) if holds else (moved > 0.3), (slope, moved)
        assert np.isfinite(S).all()


def test_bogies_conform_and_respect_limits(oracle):
    ro = oracle
    ter = small_procedural()
    cfg, t = ro.default_config(), oracle_terrain(ro, ter, 512)
    S = ro.new_state(256)
    ro.reset_all(cfg, t, S)
    rng = np.random.RandomState(0)
    lim = 0.17453292519943295
    for _ in range(40):
        a = rng.uniform(-1, 1, (256, 2)).astype(np.float32)
        ro.step(cfg, t, S, a)
        assert np.isfinite(S).all()
        assert np.abs(S[:, ro.BOGIE_Q:ro.BOGIE_Q + 3]).max() <= lim + 1e-6
        q = S[:, ro.QUAT:ro.QUAT + 4]
        assert np.allclose((q ** 2).sum(1), 1.0, atol=1e-5)
        assert np.linalg.norm(S[:, ro.LINVEL:ro.LINVEL + 3], axis=1).max() <= 1.5 + 1e-5
        tilt = 1 - 2 * (q[:, 1] ** 2 + q[:, 2] ** 2)                     # cos of the tilt angle
        assert tilt.min() > 0.8
    assert np.abs(S[:, ro.BOGIE_Q:ro.BOGIE_Q + 3]).max() > 0.01           # the suspension is actually used


def test_collision_only_from_obstacle_layer(oracle):
    """Contact forces are reported only for wheels standing on the obstacle layer (rover_env_cfg.py:72-75)."""
    ro = oracle
    from isaac_rover_orbit_amd import terrain as T
    Hh = W = 600
    g = np.zeros((Hh, W), np.float32)
    ob = np.zeros((Hh, W), np.float32)
    ob[290:312, 305:318] = 0.05                                           # a 5 cm slab under the front-left wheel only
    zero = np.zeros((Hh, W), np.uint8)
    ter = T.Terrain(ground=g, obstacle=ob, rock_mask=zero, safe_rock_mask=zero)
    ter.spawn_locations = np.array([[15.0, 15.0, 0.0]], np.float32)
    cfg, t = ro.default_config(), oracle_terrain(ro, ter)
    S = fresh(ro, 1, (15.17, 14.65), 0.45)                                # FL wheel at (15.61, 15.04) -> on the slab
    f = None
    for _ in range(40):
        f = ro.physics_step(cfg, t, S, np.zeros((1, 4), np.float32), np.zeros((1, 6), np.float32), 1)
    rows = np.nonzero(np.abs(f[0]).sum(1) > 0)[0]
    assert list(rows) == [9]                                              # FL_Drive only
    assert f[0, 9, 2] > 5.0                                               # carries load => "collision" (> 1 N)
    od, oa, rew, term = ro.mdp_terms(cfg, np.zeros((1, 3), np.float32) + 5, np.zeros((1, 2), np.float32),
                                     np.zeros((1, 2), np.float32), np.zeros(1, np.int32), f)
    assert term[0, 3] == 1 and rew[0, 5] == 1.0


def belly_rock_terrain(height):
    """A flat map with ONE box of the obstacle layer, 0.7 m x 0.5 m, placed so that a rover at (15, 15) heading +x straddles it:
    x in [14.5, 15.2], |y - 15| <= 0.25 -- between the wheel tracks (inner wheel faces at |y| ~ 0.33), under the rear bogie beam."""
    from isaac_rover_orbit_amd import terrain as T
    Hh = W = 600
    g = np.zeros((Hh, W), np.float32)
    ob = np.zeros((Hh, W), np.float32)
    ob[295:306, 290:305] = height
    zero = np.zeros((Hh, W), np.uint8)
    ter = T.Terrain(ground=g, obstacle=ob, rock_mask=zero, safe_rock_mask=zero)
    ter.spawn_locations = np.array([[15.0, 15.0, 0.0]], np.float32)
    return ter


def test_link_bodies_report_contact_with_a_rock_between_the_wheels(oracle):
    """Round-2 review: only a wheel standing on the obstacle layer produced force, so a rover straddling a 0.5 m rock never
    ended on `collision`.  The 13 sensor bodies (rover_env_cfg.py:72-75) are 3 bogies + 4 steer links + 6 wheels: a rock under
    the belly that no wheel touches reaches the rear bogie beam (0.286 m above the ground) and terminates the episode; a 0.2 m
    rock in the same place passes underneath."""
    ro = oracle
    for height, expect in ((0.5, True), (0.2, False)):
        cfg, t = ro.default_config(), oracle_terrain(ro, belly_rock_terrain(height))
        S = fresh(ro, 1, (15.0, 15.0), 0.26878)
        f = ro.physics_step(cfg, t, S, np.zeros((1, 4), np.float32), np.zeros((1, 6), np.float32), 1)
        assert np.abs(f[0, 7:]).max() == 0.0                              # no wheel stands on the obstacle layer
        assert np.abs(f[0, :, :2]).max() == 0.0                           # the rock's top is flat: the link forces are vertical here
        rows = np.nonzero(f[0, :, 2] > 0)[0]
        od, oa, rew, term = ro.mdp_terms(cfg, np.zeros((1, 3), np.float32) + 5, np.zeros((1, 2), np.float32),
                                         np.zeros((1, 2), np.float32), np.zeros(1, np.int32), f)
        if expect:
            assert list(rows) == [2] and f[0, 2, 2] > 1000.0              # R_Boogie, ~0.21 m inside the rock at 2e4 N/m x 2 points
            assert term[0, 3] == 1 and rew[0, 5] == 1.0                   # collision_with_obstacles / collision_penalty
        else:
            assert len(rows) == 0 and term[0, 3] == 0
    # through the whole step: the episode ends on `collision` in the first step and the env is reset
    cfg, t = ro.default_config(), oracle_terrain(ro, belly_rock_terrain(0.5))
    S = fresh(ro, 1, (15.0, 15.0), 0.26878)
    S[:, ro.TARGET_W:ro.TARGET_W + 3] = [20.0, 15.0, 0.0]
    obs, rew, term, trunc, force, log = ro.step(cfg, t, S, np.zeros((1, 2), np.float32))
    assert term[0] == 1 and trunc[0] == 0 and force[0, 2, 2] > 1000.0 and log[10] == 1.0
    # a steer fork: a 0.25 m post just inboard of the front-left wheel
    from isaac_rover_orbit_amd import terrain as T
    ter = belly_rock_terrain(0.0)
    ob = ter.obstacle.copy()
    ob[305:307, 308:310] = 0.25                                           # x in [15.40, 15.50], y in [15.25, 15.35]: fork at (15.44, 15.3125), wheel at y = 15.3925
    ter2 = T.Terrain(ground=ter.ground, obstacle=ob, rock_mask=ter.rock_mask, safe_rock_mask=ter.safe_rock_mask)
    ter2.spawn_locations = ter.spawn_locations
    cfg, t = ro.default_config(), oracle_terrain(ro, ter2)
    S = fresh(ro, 1, (15.0, 15.0), 0.26878)
    f = ro.physics_step(cfg, t, S, np.zeros((1, 4), np.float32), np.zeros((1, 6), np.float32), 1)
    assert f[0, 3, 2] > 1.0 and np.abs(f[0, 7:]).max() == 0.0             # FL_Steer touches, the wheel beside it does not
    assert f[0, 3, 0] == 0.0 and f[0, 3, 1] > 0.0                         # ... on the post's edge in y (rows 306 / 307): the force leans away from it, +y
    # a FRONTAL hit: the fork meets the rising face of a 2 m post (one column at x = 15.40, the fork at x = 15.44 on the slope
    # down to the next column) -- the force points along the surface's normal, away from the face: x row > 0, and much larger
    # than the z row on a face that steep (rewards.py:119-124 norms per xyz component: a graze and a frontal hit now differ)
    ob = ter.obstacle.copy()
    ob[305:307, 308:309] = 2.0
    ter3 = T.Terrain(ground=ter.ground, obstacle=ob, rock_mask=ter.rock_mask, safe_rock_mask=ter.safe_rock_mask)
    ter3.spawn_locations = ter.spawn_locations
    cfg, t = ro.default_config(), oracle_terrain(ro, ter3)
    S = fresh(ro, 1, (15.0, 15.0), 0.26878)
    f = ro.physics_step(cfg, t, S, np.zeros((1, 4), np.float32), np.zeros((1, 6), np.float32), 1)
    assert f[0, 3, 2] > 1.0 and f[0, 3, 0] > 5.0 * f[0, 3, 2]
    od, oa, rew, term = ro.mdp_terms(cfg, np.zeros((1, 3), np.float32) + 5, np.zeros((1, 2), np.float32), np.zeros((1, 2), np.float32),
                                     np.zeros(1, np.int32), f)
    assert term[0, 3] == 1


def test_oracle_is_deterministic_and_shard_invariant(oracle):
    ro = oracle
    ter = small_procedural()
    t = oracle_terrain(ro, ter, 256)
    rng = np.random.RandomState(4)
    acts = rng.uniform(-1, 1, (10, 128, 2)).astype(np.float32)

    def run(lo, hi):
        cfg = ro.default_config(seed_lo=77)          # a fresh config = call counter 0 (it keys the per-batch spawn rows)
        S = ro.new_state(hi - lo)
        ro.reset_all(cfg, t, S, env_id_offset=lo)
        outs = []
        for a in acts:
            outs.append(ro.step(cfg, t, S, a[lo:hi], env_id_offset=lo)[0])
        return S, np.stack(outs)

    S_all, o_all = run(0, 128)
    S_all2, o_all2 = run(0, 128)
    assert np.array_equal(S_all, S_all2) and np.array_equal(o_all, o_all2)
    S_lo, o_lo = run(0, 50)
    S_hi, o_hi = run(50, 128)
    assert np.array_equal(np.concatenate([S_lo, S_hi]), S_all)
    assert np.array_equal(np.concatenate([o_lo, o_hi], 1), o_all)


def test_step_ordering_quirks(oracle):
    """B-13 (stale command in reward/termination), B-14 (counter before terminations), reset-time action zeroing."""
    ro = oracle
    cfg, t = ro.default_config(), oracle_terrain(ro, flat(), 8)
    S = ro.new_state(4)
    ro.reset_all(cfg, t, S)
    Si = S.view(np.int32)
    # env 0: the STALE command says "success" although the true target is 9 m away -> terminates anyway
    S[0, ro.CMD_B:ro.CMD_B + 2] = 0.01
    # env 1: counter at 749 -> incremented to 750 before the time_out test -> truncated, not terminated
    Si[1, ro.EP_LEN] = 749
    a = np.full((4, 2), 0.3, np.float32)
    obs, rew, term, trunc, force, log = ro.step(cfg, t, S, a)
    assert term[0] == 1 and trunc[0] == 0 and trunc[1] == 1 and term[1] == 0 and term[2] == 0 and trunc[2] == 0
    # reached_target reward on env 0 uses the incremented counter: 5 * (750 - 1)/750 * 0.2 (+ the other terms)
    assert rew[0] > 0.9
    assert (obs[:2, :2] == 0).all() and (obs[2:, :2] == a[2:]).all()        # reset envs see a zeroed last action
    assert (Si[:2, ro.EP_LEN] == 0).all() and (Si[2:, ro.EP_LEN] == 1).all()
    assert log[13] == 2 and log[7] == 1 and log[8] == 1                      # one time_limit, one is_success
    # fresh command for the observation: distance obs ~ 9 m * 0.11 after the reset
    assert abs(obs[0, 2] - 0.99) < 2e-3


def test_spawn_rows_are_distinct_inside_a_reset_batch(oracle):
    """randomizations.py:22 takes a randperm PREFIX: envs that reset together never share a spawn table row.  spawn_draw = 1
    (default) reproduces that with an affine bijection of the global env ids, redrawn per call from (seed, call counter);
    spawn_draw = 0 draws one row per env independently (rows repeat)."""
    ro = oracle
    n = 300
    table = np.stack([np.arange(2 * n, dtype=np.float32) + 20.0, np.full(2 * n, 25.0, np.float32), np.zeros(2 * n, np.float32)], 1)
    ter = flat()
    t = ro.TerrainData(ter.height, ter.obstacle, ter.safe_rock_mask, ter.resolution, ter.min_x, ter.min_y, table)
    seen = []
    for draw, expect_distinct in ((1, True), (0, False)):
        cfg = ro.default_config(spawn_draw=draw, seed_lo=3)
        S = ro.new_state(n)
        ro.reset_all(cfg, t, S)
        rows = (S[:, 0] - 20.0).astype(int)
        assert (len(set(rows.tolist())) == n) == expect_distinct
        assert rows.min() >= 0 and rows.max() < 2 * n
        S2 = ro.new_state(n)
        ro.reset_all(cfg, t, S2)                       # next call: a different permutation (independent draws are keyed
        assert np.array_equal(S2[:, 0], S[:, 0]) != expect_distinct    # by the env's own reset count instead)
        seen.append(rows)
    # shards of one batch draw from the same bijection: no collision across GPUs either
    cfg = ro.default_config(seed_lo=3)
    lo, hi = ro.new_state(100), ro.new_state(200)
    ro.reset_all(cfg, t, lo, env_id_offset=0)
    cfg.counter_lo = 0
    ro.reset_all(cfg, t, hi, env_id_offset=100)
    assert np.array_equal(np.concatenate([lo[:, 0], hi[:, 0]]) - 20.0, seen[0].astype(np.float32))
