"""SURVEY section 5, sanitizer row (CPU only -- never on the GPU box's card): the three C oracles built with AddressSanitizer +
UndefinedBehaviorSanitizer replay a BASELINE config-1 rollout, the reference fixture batches (Ackermann, MDP terms, reset with
recorded draws, heightmap look-ups), a contact-rich procedural rollout with in-step resets, the FrankaCubeLift-v0 oracle and the
policy oracle -- any out-of-bounds access, misaligned load, signed overflow or shift error aborts the child process."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import flat, oracle_terrain, small_procedural, random_policy_weights
    from oracle import rover_oracle as ro, lift_oracle as lo, policy_oracle as po
    assert "_san" in ro.lib()._name and "_san" in lo.lib()._name and "_san" in po.lib()._name
    G = os.path.join(ROOT, "tests", "golden")
    # ---- BASELINE config 1: N = 1, flat terrain, 64 random-action steps
    ter = flat(1024); ter.spawn_locations = np.array([[25.6, 25.6, 0.0]], np.float32)
    cfg, t = ro.default_config(), oracle_terrain(ro, ter)
    S = ro.new_state(1); ro.reset_all(cfg, t, S)
    rng = np.random.RandomState(0)
    for k in range(64):
        obs, rew, term, trunc, force, log = ro.step(cfg, t, S, rng.uniform(-1, 1, (1, 2)).astype(np.float32))
    assert np.isfinite(obs).all() and np.isfinite(S).all()
    # ---- reference fixtures through the unit entries
    g = np.load(os.path.join(G, "ackermann.npz")); ro.ackermann(cfg, g["raw"])
    m = np.load(os.path.join(G, "mdp_terms.npz"))
    ro.mdp_terms(cfg, m["cmd"], m["action"], m["prev_action"], m["episode_length_buf"], m["force_matrix_w"])
    ro.height_scan_term(cfg, m["pos_w"][:, 2], m["ray_hits_z"])
    r = np.load(os.path.join(G, "reset.npz"))
    n = int(r["num_envs"])
    from isaac_rover_orbit_amd import terrain as T
    H, W = r["heightmap"].shape
    rt = T.Terrain(ground=r["heightmap"].astype(np.float32), obstacle=np.zeros((H, W), np.float32), rock_mask=r["safe_mask"].astype(np.uint8),
                   safe_rock_mask=r["safe_mask"].astype(np.uint8), resolution=float(r["resolution"]), min_x=float(r["min_xy"][0]),
                   min_y=float(r["min_xy"][1]))
    rt.spawn_locations = r["spawn_table"].astype(np.float32)
    c3 = ro.default_config(max_target_tries=int(r["max_tries"]))
    t3 = oracle_terrain(ro, rt)
    S3 = ro.new_state(n)
    from helpers import reset_fixture_case, check_reset_against_fixture
    for b in (0, 1):
        mask, row, yaw, th, hd, expect = reset_fixture_case(r, b)
        ro.reset_with_draws(c3, t3, S3, mask, row, yaw, th, hd)
        check_reset_against_fixture(S3, expect, ro)
    hm = np.load(os.path.join(G, "heightmap.npz"))
    ht = ro.TerrainData(hm["wavy_heightmap"].astype(np.float32), None, hm["wavy_mask"].astype(np.uint8), 0.05, float(hm["wavy_bounds"][0]),
                        float(hm["wavy_bounds"][1]), np.zeros((1, 3), np.float32))
    assert np.array_equal(ro.get_height_at(ht, hm["wavy_query_xy"]), hm["wavy_query_height"])
    ro.target_invalid(ht, hm["wavy_query_xy"])
    # ---- contact-rich procedural rollout (rocks, in-step resets, rays leaving the map for envs at the border)
    ter2 = small_procedural()
    cfg2, t2 = ro.default_config(seed_lo=3), oracle_terrain(ro, ter2, 2 * 96)
    S2 = ro.new_state(96); ro.reset_all(cfg2, t2, S2)
    S2[:8, 0:2] = [[0.3, 0.3], [50.9, 50.9], [0.2, 25.0], [25.0, 51.0], [51.0, 0.2], [0.0, 0.0], [51.15, 51.15], [10.0, 0.1]]
    S2[8:16, 51] = np.array([748], np.int32).view(np.float32)
    for k in range(40):
        obs, rew, term, trunc, force, log = ro.step(cfg2, t2, S2, rng.uniform(-1, 1, (96, 2)).astype(np.float32))
    ro.height_scan(cfg2, t2, S2)
    # ---- FrankaCubeLift-v0 oracle: 60 steps with grasp attempts + the pinned term functions
    lc = lo.default_config(); LS = lo.new_state(32); lo.reset(lc, LS)
    for k in range(260):
        a = rng.uniform(-1, 1, (32, 8)).astype(np.float32)
        if k % 40 < 20: a[:, 7] = -1
        lo.step(lc, LS, a)
    lt = np.load(os.path.join(G, "lift_terms.npz"))
    lo.terms(lt["object_pos_w"], lt["ee_pos_w"], lt["robot_root_state_w"], lt["command"])
    lo.mass_matrix(lo.Q_DEFAULT[:7]); lo.gravity_torque(lo.Q_DEFAULT[:7]); lo.model_constants()
    # ---- policy oracle: actor + critic on 37 rows (not a multiple of the 16-row tile)
    for out_dim in (2, 1):
        ws, bs = random_policy_weights(seed=out_dim, out_dim=out_dim)
        d = po.default_desc(out_dim)
        x = rng.uniform(-1, 1, (37, 965)).astype(np.float32)
        y = po.forward(d, ws, bs, x)
        assert np.isfinite(y).all()
    print("sanitized oracles: ok")
''')


def test_oracles_under_asan_and_ubsan(tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("gcc has no libasan here")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "sanitized"], stdout=subprocess.DEVNULL)
    script = tmp_path / "child.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + CHILD)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", ROVER_ORACLE_DIR=os.path.join(ROOT, "oracle", "_san"),
               OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-4000:])
    assert "sanitized oracles: ok" in out.stdout
    assert "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-4000:]
