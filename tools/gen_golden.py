#!/usr/bin/env python3
"""Generate the golden vectors under ``tests/golden`` by EVALUATING the reference's own torch/numpy
arithmetic (imported from /root/reference under the sys.modules stubs of ``reference_stubs.py``).

Run in the build container only (the reference never travels to the GPU box):

    python tools/gen_golden.py

Outputs (small .npz files, data only -- inputs and the reference's outputs):

* ``ackermann.npz``   -- AckermannAction2.process_actions/ackermann     (ackermann_actions.py:226-322)
* ``mdp_terms.npz``   -- 3 obs + 7 reward + 3 termination term functions  (observations.py, rewards.py,
                          terminations.py)
* ``heightmap.npz``   -- HeightmapManager.mesh_to_heightmap / get_height_at (terrain_utils.py:23-84),
                          TerrainManager.check_if_target_is_valid (:202-223),
                          TerrainManager.random_rover_spawns(seed=41) (:330-385)

* ``reset.npz``       -- reset_root_state_rover (randomizations.py:12-39), RoverTerrainImporter.sample_new_targets /
                          generate_random_targets (terrain_importer.py:134-175) and the heading draw of
                          TerrainBasedPositionCommand._resample_command (:74-95), with every torch draw RECORDED
                          (spawn index, yaw uniform, the theta uniforms incl. rejected ones, heading uniform) so that
                          the outcome can be replayed by injecting the draws (torch's RNG stream itself cannot be)

* ``lift_terms.npz``  -- FrankaCubeLift-v0: the reference's own reward / observation terms
                          (manipulation/mdp/rewards.py:20-67, observations.py:19-31) on 192 rows incl. every threshold; the
                          two ORBIT frame-transform helpers they call (``utils.math.combine_frame_transforms`` /
                          ``subtract_frame_transforms``, third-party, absent) are supplied as a restatement of their
                          documented quaternion algebra -- that part is restated, not pinned

* ``policy_forward.npz`` -- SURVEY 8f-3: ``GaussianNeuralNetwork.compute`` / ``DeterministicNeuralNetwork.compute`` of the
                          reference's OWN model classes (rover_envs/envs/navigation/learning/skrl/models.py:24-36, 89-102, 151-163;
                          skrl's base classes by tests/doubles/skrl) with seeded random weights generated here and loaded via
                          ``load_state_dict``, on 256 synthetic rows + 64 rows an env produced.  Stored: the generating seeds
                          with sha256 digests of the regenerated weights / synthetic rows, the env rows, the outputs.

Functions that live in third-party code absent from the container (ORBIT math utils, PhysX, Warp
ray-caster, cv2 morphology) cannot be evaluated and are NOT covered here; see DESIGN.md "parity".
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import reference_stubs  # noqa: E402

reference_stubs.install()

from rover_envs.mdp.actions import ackermann_actions as ref_aa  # noqa: E402
import rover_envs.envs.navigation.mdp as ref_mdp  # noqa: E402
from rover_envs.envs.navigation.utils.terrains import terrain_utils as ref_tu  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(1)


# --------------------------------------------------------------------------------------- ackermann
def gen_ackermann():
    cls = ref_aa.AckermannAction2
    term = cls.__new__(cls)
    # AAU rover cfg: rover_envs/envs/navigation/robots/aau_rover/env_cfg.py:21-31
    term._wheel_radius = 0.1
    term._rear_and_front_wheel_distance = 0.77
    term._middle_wheel_distance = 0.894
    term._wheelbase_length = 0.849
    cls.device = "cpu"
    term._scale = torch.tensor((1.0, 1.0)).unsqueeze(0)
    term._offset = torch.tensor(-0.0135).unsqueeze(0)

    g = torch.Generator().manual_seed(1234)
    raw = torch.rand(512, 2, generator=g) * 2 - 1
    edge = torch.tensor([
        [0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [-1.0, 0.5], [0.2, -1.0], [0.0135, 0.0135], [0.5, 0.0135],
        [0.0135, 0.7], [0.0135, -0.7], [-1.0, -1.0], [1.0, -1.0], [-0.3, 0.0135], [0.9, 0.95],
        [0.7287, 1.0135], [0.9075, 1.0135], [0.65, 0.9], [-0.65, 0.9], [1e-4, 1e-4], [0.0136, 0.0134],
    ])
    # larger-than-unit actions: step() does not clip (SURVEY 8b last row)
    wide = (torch.rand(64, 2, generator=g) * 2 - 1) * 3.0
    raw = torch.cat([edge, raw, wide], 0).to(torch.float32)

    term._raw_actions = torch.zeros_like(raw)
    term.process_actions(raw)
    processed = term._processed_actions.clone()
    steer, wheel = term.ackermann(processed[:, 0], processed[:, 1])
    np.savez(os.path.join(OUT, "ackermann.npz"),
             raw=raw.numpy(), processed=processed.numpy(), steer=steer.numpy(), wheel=wheel.numpy(),
             cfg=np.array([0.1, 0.77, 0.894, 0.849, 1.0, 1.0, -0.0135], dtype=np.float64))
    print("ackermann:", raw.shape, steer.shape, wheel.shape)


# --------------------------------------------------------------------------------------- mdp terms
class _Env:
    pass


def _make_env(cmd, action, prev_action, ep_len, force, pos_w, hits, max_len=750):
    env = _Env()
    env.num_envs = cmd.shape[0]
    env.device = "cpu"
    env.max_episode_length = max_len
    env.episode_length_buf = ep_len
    env.command_manager = types.SimpleNamespace(get_command=lambda name: cmd)
    env.action_manager = types.SimpleNamespace(action=action, prev_action=prev_action)
    contact = types.SimpleNamespace(data=types.SimpleNamespace(force_matrix_w=force))
    scanner = types.SimpleNamespace(data=types.SimpleNamespace(pos_w=pos_w, ray_hits_w=hits))
    env.scene = types.SimpleNamespace(sensors={"contact_sensor": contact, "height_scanner": scanner})
    return env


NSCAN = 16


def gen_mdp_terms():
    g = torch.Generator().manual_seed(4321)
    n = 256
    cmd = (torch.rand(n, 3, generator=g) * 2 - 1) * torch.tensor([12.5, 12.5, 0.5])
    # rows that sit on / around every threshold used by the terms
    cmd[:12] = torch.tensor([
        [9.0, 0.0, 0.1], [0.1, 0.1, 0.0], [-3.0, -0.2, 0.0], [0.0, 12.0, 0.0], [0.18, 0.0, 0.0],
        [0.17999, 0.0, 0.0], [11.0, 0.0, 0.0], [11.0001, 0.0, 0.0], [0.0, 0.0, 0.0], [-1.0, 0.0, 0.0],
        [-1.0, -1e-8, 0.0], [-0.5, 1.0926, 0.0],
    ])
    action = torch.rand(n, 2, generator=g) * 2 - 1
    prev_action = torch.rand(n, 2, generator=g) * 2 - 1
    action[:4] = torch.tensor([[0.5, 0.2], [-0.3, 0.1], [1.0, -1.0], [0.0, 0.0]])
    prev_action[:4] = torch.tensor([[0.4, 0.1], [0.3, 0.1], [0.0, 0.0], [0.0, 0.0]])
    prev_action[4] = action[4] - 0.05 / 3        # exactly at the oscillation threshold
    prev_action[5] = action[5] - 0.017
    ep_len = torch.randint(0, 751, (n,), generator=g)
    ep_len[:4] = torch.tensor([0, 10, 700, 749])
    force = torch.zeros(n, 13, 1, 3)
    hit = torch.rand(n, generator=g) < 0.3
    force[hit] = (torch.rand(int(hit.sum()), 13, 1, 3, generator=g) * 2 - 1) * \
        (torch.rand(int(hit.sum()), 1, 1, 1, generator=g) * 3.0)
    force[:4] = 0.0
    force[2, 3, 0] = torch.tensor([0.0, 2.0, 0.0])
    force[6, 9, 0] = torch.tensor([0.6, 0.3, 0.1])        # sum of per-axis norms == 1.0 -> not > 1
    force[7, 9, 0] = torch.tensor([0.6, 0.3, 0.1001])
    pos_w = torch.rand(n, 3, generator=g) * torch.tensor([100.0, 100.0, 2.0])
    hits = torch.rand(n, 961, 3, generator=g) * torch.tensor([100.0, 100.0, 1.5])
    hits[3, 17, 2] = float("inf")                             # a ray miss (ORBIT: +inf)
    pos_w[0, 2] = 10.5
    hits[0, :, 2] = 0.3

    env = _make_env(cmd, action, prev_action, ep_len, force, pos_w, hits)
    sensor = types.SimpleNamespace(name="contact_sensor")
    scanner = types.SimpleNamespace(name="height_scanner")
    out = {
        "obs_angle": ref_mdp.angle_to_target_observation(env, "target_pose"),
        "obs_distance": ref_mdp.distance_to_target_euclidean(env, "target_pose"),
        "obs_height_scan": ref_mdp.height_scan_rover(env, scanner),
        "rew_distance_to_target": ref_mdp.distance_to_target_reward(env, "target_pose"),
        "rew_reached_target": ref_mdp.reached_target(env, "target_pose", 0.18),
        "rew_oscillation": ref_mdp.oscillation_penalty(env),
        "rew_angle_to_target": ref_mdp.angle_to_target_penalty(env, "target_pose"),
        "rew_heading_soft_contraint": ref_mdp.heading_soft_contraint(env, types.SimpleNamespace(name="robot")),
        "rew_collision": ref_mdp.collision_penalty(env, sensor, 1.0),
        "rew_far_from_target": ref_mdp.far_from_target_reward(env, "target_pose", 11.0),
        "term_is_success": ref_mdp.is_success(env, "target_pose", 0.18),
        "term_far_from_target": ref_mdp.far_from_target(env, "target_pose", 11.0),
        "term_collision": ref_mdp.collision_with_obstacles(env, sensor, 1.0),
    }
    arrays = {k: v.numpy() for k, v in out.items()}
    arrays["obs_height_scan"] = arrays["obs_height_scan"][:NSCAN]      # keep the fixture small
    # threshold params are IGNORED by the reference's collision terms (rewards.py:123, terminations.py:62)
    arrays["rew_collision_thr100"] = ref_mdp.collision_penalty(env, sensor, 100.0).numpy()
    np.savez_compressed(os.path.join(OUT, "mdp_terms.npz"),
                        cmd=cmd.numpy(), action=action.numpy(), prev_action=prev_action.numpy(),
                        episode_length_buf=ep_len.numpy().astype(np.int64),
                        force_matrix_w=force.numpy(), pos_w=pos_w[:NSCAN].numpy(),
                        ray_hits_z=hits[:NSCAN, :, 2].numpy(), max_episode_length=np.int64(750), **arrays)
    print("mdp_terms:", {k: tuple(v.shape) for k, v in arrays.items()})


# --------------------------------------------------------------------------------------- heightmap
def _grid_mesh(nx, ny, size_x, size_y, zfun, x0=0.0, y0=0.0):
    xs = np.linspace(x0, x0 + size_x, nx)
    ys = np.linspace(y0, y0 + size_y, ny)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    Z = zfun(X, Y)
    verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    faces = []
    for j in range(ny - 1):
        for i in range(nx - 1):
            a = j * nx + i
            faces.append([a, a + 1, a + nx])
            faces.append([a + 1, a + nx + 1, a + nx])
    return verts, np.asarray(faces, dtype=np.uint32)


def gen_heightmap():
    out = {}
    # (1) SURVEY App. D known answer: 4 verts / 2 faces -> (41, 41) all 2.0
    v = np.array([[0, 0, 0], [4, 0, 0], [0, 4, 1], [4, 4, 2]], dtype=np.float32)
    f = np.array([[0, 1, 2], [1, 3, 2]], dtype=np.uint32)
    hm = ref_tu.HeightmapManager(0.05, v, f)
    out["quad_vertices"], out["quad_faces"] = v, f
    out["quad_heightmap"] = hm.heightmap
    out["quad_bounds"] = np.array([hm.min_x, hm.min_y, hm.max_x, hm.max_y], dtype=np.float64)

    # (2) a 7 m x 6 m wavy sheet, 0.25 m triangles, non-zero mesh origin
    verts, faces = _grid_mesh(29, 25, 7.0, 6.0,
                              lambda X, Y: 0.3 * np.sin(1.3 * X) * np.cos(0.9 * Y) + 0.05 * X, x0=-0.5, y0=0.25)
    hm = ref_tu.HeightmapManager(0.05, verts, faces)
    out["wavy_vertices"], out["wavy_faces"] = verts, faces
    out["wavy_heightmap"] = hm.heightmap
    out["wavy_bounds"] = np.array([hm.min_x, hm.min_y, hm.max_x, hm.max_y], dtype=np.float64)

    # get_height_at / check_if_target_is_valid need the tensors the reference only creates on CUDA
    # (terrain_utils.py:19-21) -> inject CPU tensors by hand.
    hm.heightmap_tensor = torch.from_numpy(hm.heightmap)
    hm.offset_tensor = torch.tensor([hm.min_x, hm.min_y])
    g = torch.Generator().manual_seed(99)
    pos = torch.rand(300, 2, generator=g) * torch.tensor([9.0, 8.0]) - 1.5     # includes out-of-range
    out["wavy_query_xy"] = pos.numpy()
    out["wavy_query_height"] = hm.get_height_at(pos.clone()).numpy()

    tm = ref_tu.TerrainManager.__new__(ref_tu.TerrainManager)
    tm._heightmap_manager = hm
    tm.resolution_in_m = 0.05
    rng = np.random.RandomState(7)
    mask = (rng.rand(*hm.heightmap.shape) < 0.35).astype(np.uint8)
    tm.rock_mask_tensor = torch.from_numpy(mask).unsqueeze(-1)
    env_ids = torch.arange(300)
    bad_ids, bad_len = tm.check_if_target_is_valid(env_ids, pos.clone(), device="cpu")
    out["wavy_mask"] = mask
    invalid = np.zeros(300, dtype=np.uint8)
    invalid[bad_ids.numpy()] = 1
    out["wavy_query_invalid"] = invalid
    assert bad_len == int(invalid.sum())

    # (3) random_rover_spawns(seed=41): needs a map larger than 2 x 20 m border -> 1000 x 900 cells
    rng = np.random.RandomState(3)
    H, W = 1000, 900
    big_h = rng.rand(H, W).astype(np.float32)
    big_mask = (rng.rand(H, W) < 0.4).astype(np.uint8)
    hm2 = types.SimpleNamespace(min_x=1.0, min_y=1.5)
    tm2 = ref_tu.TerrainManager.__new__(ref_tu.TerrainManager)
    tm2._heightmap_manager = hm2
    tm2.resolution_in_m = 0.05
    spawns = tm2.random_rover_spawns(rock_mask=big_mask, heightmap=big_h, n_spawns=64, seed=41)
    out["spawn_seed_hm"] = np.int64(3)      # heightmap = RandomState(3).rand(H, W); mask = next rand(H, W) < 0.4
    out["spawn_shape"] = np.array([H, W], dtype=np.int64)
    out["spawn_min_xy"] = np.array([1.0, 1.5])
    out["spawn_locations"] = spawns
    np.savez_compressed(os.path.join(OUT, "heightmap.npz"), **out)
    print("heightmap:", {k: getattr(v, "shape", ()) for k, v in out.items()})


# --------------------------------------------------------------------------------------- reset / command sampling
class _DrawRecorder:
    """Wraps torch.rand / torch.randperm / Tensor.uniform_ while the reference's reset code runs: the calls are served
    by a seeded CPU generator and every returned tensor is logged.  ``uniform_(lo, hi)`` is served as
    ``rand * (hi - lo) + lo`` (torch's own definition of U(lo, hi)) so that its underlying uniform is known."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.rand_log, self.perm_log, self.uniform_log = [], [], []

    def __enter__(self):
        self._rand, self._randperm, self._uniform = torch.rand, torch.randperm, torch.Tensor.uniform_
        rec = self

        def rand(*size, device=None, **kw):
            out = rec._rand(*size, generator=rec.g)
            rec.rand_log.append(out.clone())
            return out

        def randperm(n, device=None, **kw):
            out = rec._randperm(n, generator=rec.g)
            rec.perm_log.append(out.clone())
            return out

        def uniform_(self_t, lo=0.0, hi=1.0):
            u = rec._rand(self_t.shape, generator=rec.g)
            rec.uniform_log.append(u.clone())
            self_t.copy_(u * (hi - lo) + lo)
            return self_t

        torch.rand, torch.randperm, torch.Tensor.uniform_ = rand, randperm, uniform_
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randperm, torch.Tensor.uniform_ = self._rand, self._randperm, self._uniform


def gen_reset():
    import math
    from rover_envs.envs.navigation.mdp import randomizations as ref_rand
    from rover_envs.envs.navigation.utils.terrains import terrain_importer as ref_ti

    N, MAX_TRIES = 96, 32
    # ---- shared terrain data: a 30 m x 25 m heightmap whose look-up offsets are non-zero (exercises B-1), a blobby
    #      safe-rock mask (~35 % blocked => the rejection loop of sample_new_targets runs several rounds)
    rng = np.random.RandomState(17)
    H, W = 500, 600
    yy, xx = np.mgrid[0:H, 0:W]
    heightmap = 0.4 * np.sin(xx * 0.021) * np.cos(yy * 0.017) + 0.02 * rng.standard_normal((H, W))
    heightmap = (np.round(heightmap * 64.0) / 64.0).astype(np.float32)   # 1/64 m steps: keeps the fixture small
    blobs = np.zeros((H, W), np.float32)
    for _ in range(260):
        cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(6, 30)
        blobs = np.maximum(blobs, ((xx - cx) ** 2 + (yy - cy) ** 2 < r * r).astype(np.float32))
    mask = blobs.astype(np.uint8)
    hm = ref_tu.HeightmapManager.__new__(ref_tu.HeightmapManager)
    hm.resolution_in_m = 0.05
    hm.heightmap = heightmap
    hm.min_x, hm.min_y = 1.25, 2.5
    hm.heightmap_tensor = torch.from_numpy(heightmap)
    hm.offset_tensor = torch.tensor([hm.min_x, hm.min_y])
    tm = ref_tu.TerrainManager.__new__(ref_tu.TerrainManager)
    tm._heightmap_manager = hm
    tm.resolution_in_m = 0.05
    tm.rock_mask_tensor = torch.from_numpy(mask).unsqueeze(-1)
    # spawn table: 2 N rows inside the map (terrain_utils.py:123-124: n_spawns = 2 * num_envs)
    spawns = np.stack([rng.uniform(9.5, 20.0, 2 * N), rng.uniform(9.5, 15.0, 2 * N), rng.uniform(-0.3, 0.3, 2 * N)], 1)
    tm.spawn_locations = torch.from_numpy(spawns.astype(np.float32))

    importer = ref_ti.RoverTerrainImporter.__new__(ref_ti.RoverTerrainImporter)
    importer._cfg = types.SimpleNamespace(num_envs=N)
    importer._terrainManager = tm
    importer.target_distance = 9.0                            # terrain_importer.py:132
    importer.device = "cpu"
    importer.env_origins = torch.zeros(N, 3)
    importer.env_origins[:, 0:2] += 100.0                      # rover_env.py:24-25 (overwritten by the first reset)

    written = {}
    asset = types.SimpleNamespace(write_root_pose_to_sim=lambda pose, env_ids: written.update(pose=pose.clone(),
                                                                                               env_ids=env_ids.clone()))
    robot = types.SimpleNamespace(data=types.SimpleNamespace(default_root_state=torch.zeros(N, 13)))
    scene = {"robot": asset}
    env = types.SimpleNamespace(device="cpu", num_envs=N,
                                scene=types.SimpleNamespace(terrain=importer, __getitem__=None))
    env.scene = type("Scene", (), {"terrain": importer, "__getitem__": lambda self, k: scene[k]})()

    cmd = ref_ti.TerrainBasedPositionCommand.__new__(ref_ti.TerrainBasedPositionCommand)
    cmd.terrain, cmd.robot = importer, robot
    cmd.device, cmd.num_envs = "cpu", N
    cmd.cfg = types.SimpleNamespace(simple_heading=False, ranges=types.SimpleNamespace(heading=(-math.pi, math.pi)))
    cmd.pos_command_w = torch.zeros(N, 3)
    cmd.heading_command_w = torch.zeros(N)

    out = {"heightmap": heightmap, "safe_mask": mask, "min_xy": np.array([hm.min_x, hm.min_y], np.float64),
           "resolution": np.float64(0.05), "spawn_table": spawns.astype(np.float32), "num_envs": np.int64(N),
           "max_tries": np.int64(MAX_TRIES), "z_offset": np.float64(0.5), "target_distance": np.float64(9.0),
           "heading_range": np.array([-math.pi, math.pi], np.float64)}
    # two batches: the initial reset of every env, then a partial in-step reset of a scattered subset
    batches = [torch.arange(N), torch.tensor(sorted(rng.choice(N, 23, replace=False).tolist()))]
    for b, env_ids in enumerate(batches):
        k = len(env_ids)
        with _DrawRecorder(1000 + b) as rec:
            ref_rand.reset_root_state_rover(env, env_ids, types.SimpleNamespace(name="robot"), z_offset=0.5)
            n_rand_reset = len(rec.rand_log)
            ids_iter = []
            orig_check = tm.check_if_target_is_valid

            def spy(e_ids, pos, device="cuda:0", _orig=orig_check, _log=ids_iter):
                _log.append(e_ids.clone())
                return _orig(e_ids, pos, device=device)

            tm.check_if_target_is_valid = spy
            cmd._resample_command(env_ids)
            tm.check_if_target_is_valid = orig_check
        assert len(rec.perm_log) == 1 and n_rand_reset == 1 and len(rec.uniform_log) == 1
        theta_u = np.full((k, MAX_TRIES), np.nan, np.float32)
        tries = np.zeros(k, np.int64)
        pos_of = {int(e): i for i, e in enumerate(env_ids.tolist())}
        for ids, u in zip(ids_iter, rec.rand_log[1:]):
            for e, val in zip(ids.tolist(), u.tolist()):
                i = pos_of[int(e)]
                theta_u[i, tries[i]] = val
                tries[i] += 1
        assert tries.max() <= MAX_TRIES and tries.min() >= 1
        pre = f"b{b}_"
        out[pre + "env_ids"] = env_ids.numpy().astype(np.int64)
        out[pre + "spawn_index"] = rec.perm_log[0][:k].numpy().astype(np.int64)          # randomizations.py:22
        out[pre + "yaw_u"] = rec.rand_log[0].numpy()                                      # :30
        out[pre + "theta_u"] = theta_u                                                    # terrain_importer.py:169
        out[pre + "tries"] = tries
        out[pre + "heading_u"] = rec.uniform_log[0].numpy()                               # :93-95
        out[pre + "root_pose"] = written["pose"].numpy()                                  # (k, 7) pos + quat (w, x, y, z)
        out[pre + "env_origins"] = importer.env_origins[env_ids].numpy()                  # randomizations.py:37
        out[pre + "pos_command_w"] = cmd.pos_command_w[env_ids].numpy()
        out[pre + "heading_command_w"] = cmd.heading_command_w[env_ids].numpy()
        print(f"reset batch {b}: {k} envs, target tries max {tries.max()} mean {tries.mean():.2f}")
    np.savez_compressed(os.path.join(OUT, "reset.npz"), **out)


# --------------------------------------------------------------------------------------- manipulation (lift) terms
def _quat_apply(q, v):
    """ORBIT utils.math.quat_apply (w, x, y, z): v + 2 w (q_v x v) + 2 q_v x (q_v x v)."""
    w, xyz = q[:, 0:1], q[:, 1:4]
    t = torch.cross(xyz, v, dim=-1) * 2
    return v + w * t + torch.cross(xyz, t, dim=-1)


def _combine_frame_transforms(t01, q01, t12=None, q12=None):
    t02 = t01 + _quat_apply(q01, t12) if t12 is not None else t01
    return t02, q01            # q12 is never passed by the reference's terms (identity)


def _subtract_frame_transforms(t01, q01, t02=None, q02=None):
    q10 = torch.cat([q01[:, 0:1], -q01[:, 1:4]], dim=-1)
    t12 = _quat_apply(q10, t02 - t01) if t02 is not None else _quat_apply(q10, -t01)
    return t12, q10


def gen_lift_terms():
    class SceneEntityCfg:       # default arguments of the reference's term functions are built at import time
        def __init__(self, name, **kw):
            self.name = name
    sys.modules["omni.isaac.orbit.managers"].SceneEntityCfg = SceneEntityCfg
    m = sys.modules["omni.isaac.orbit.utils.math"]
    m.combine_frame_transforms, m.subtract_frame_transforms = _combine_frame_transforms, _subtract_frame_transforms
    # the package __init__ builds the whole ORBIT cfg tree (needs real configclasses); the term functions live in two
    # self-contained modules, loaded here by path
    import importlib.util
    lift_mdp = types.SimpleNamespace()
    for fn in ("rewards", "observations"):
        spec = importlib.util.spec_from_file_location(
            f"_ref_lift_{fn}", os.path.join(reference_stubs.REFERENCE_ROOT, "rover_envs", "envs", "manipulation", "mdp", fn + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        for k, v in vars(mod).items():
            if callable(v) and getattr(v, "__module__", "") == mod.__name__:
                setattr(lift_mdp, k, v)

    g = torch.Generator().manual_seed(2024)
    n = 192
    obj = torch.rand(n, 3, generator=g) * torch.tensor([0.6, 0.8, 0.5]) + torch.tensor([0.2, -0.4, -0.1])
    obj[:8, 2] = torch.tensor([0.06, 0.0600001, 0.0599999, 0.055, -0.05, -0.0500001, 0.2, 0.02])    # around every threshold
    ee = obj + (torch.rand(n, 3, generator=g) - 0.5) * torch.tensor([0.6, 0.6, 0.6])
    ee[8:12] = obj[8:12]                                                                            # distance 0
    root = torch.zeros(n, 13)
    root[:, 0:3] = (torch.rand(n, 3, generator=g) - 0.5) * 0.2
    ang = torch.rand(n, generator=g) * 6.2831853
    axis = torch.nn.functional.normalize(torch.rand(n, 3, generator=g) - 0.5, dim=-1)
    root[:, 3] = torch.cos(ang / 2)
    root[:, 4:7] = axis * torch.sin(ang / 2).unsqueeze(-1)
    root[:64, 0:3] = 0.0
    root[:64, 3:7] = torch.tensor([1.0, 0.0, 0.0, 0.0])          # the task's robot root: origin, identity
    cmd = torch.zeros(n, 7)
    cmd[:, 0:3] = torch.rand(n, 3, generator=g) * torch.tensor([0.4, 0.4, 0.5]) + torch.tensor([0.3, 0.3, 0.0])
    cmd[:, 3] = 1.0
    env = types.SimpleNamespace(
        scene={"object": types.SimpleNamespace(data=types.SimpleNamespace(root_pos_w=obj)),
               "ee_frame": types.SimpleNamespace(data=types.SimpleNamespace(target_pos_w=ee.unsqueeze(1))),
               "robot": types.SimpleNamespace(data=types.SimpleNamespace(root_state_w=root))},
        command_manager=types.SimpleNamespace(get_command=lambda name: cmd))
    out = {
        "object_pos_w": obj.numpy(), "ee_pos_w": ee.numpy(), "robot_root_state_w": root.numpy(), "command": cmd.numpy(),
        "rew_object_is_lifted": lift_mdp.object_is_lifted(env, minimal_height=0.06).numpy(),
        "rew_object_ee_distance": lift_mdp.object_ee_distance(env, std=0.1).numpy(),
        "rew_object_goal_distance_03": lift_mdp.object_goal_distance(env, std=0.3, minimal_height=0.06, command_name="object_pose").numpy(),
        "rew_object_goal_distance_005": lift_mdp.object_goal_distance(env, std=0.05, minimal_height=0.06, command_name="object_pose").numpy(),
        "obs_object_position_in_robot_root_frame": lift_mdp.object_position_in_robot_root_frame(env).numpy(),
    }
    np.savez_compressed(os.path.join(OUT, "lift_terms.npz"), **out)
    print("lift_terms:", {k: tuple(v.shape) for k, v in out.items()})


# --------------------------------------------------------------------------------------- policy forward (f-3)
def _test_paths():
    tests = os.path.dirname(OUT)
    for p in (os.path.join(tests, "doubles"), tests, os.path.dirname(tests)):
        if p not in sys.path:
            sys.path.insert(0, p)


def _load_reference_models():
    """rover_envs/envs/navigation/learning/skrl/models.py by path, with tests/doubles/skrl standing in for skrl 1.1.0's three
    base classes (Model, GaussianMixin, DeterministicMixin -- none of them takes part in ``compute``)."""
    import importlib.util
    _test_paths()
    spec = importlib.util.spec_from_file_location(
        "_ref_skrl_models", os.path.join(reference_stubs.REFERENCE_ROOT, "rover_envs", "envs", "navigation", "learning", "skrl", "models.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_networks(models, ws_bs_by_role):
    """The reference's actor / critic built as get_models.py:36-62 builds them, with the given weights loaded."""
    space = types.SimpleNamespace(shape=(965,))
    aspace = types.SimpleNamespace(shape=(2,))
    kw = dict(observation_space=space, action_space=aspace, device="cpu", mlp_input_size=4, mlp_layers=[256, 160, 128],
              mlp_activation="leaky_relu", encoder_input_size=961, encoder_layers=[80, 60], encoder_activation="leaky_relu")
    nets = {"policy": models.GaussianNeuralNetwork(**kw), "value": models.DeterministicNeuralNetwork(**kw)}
    for role, (ws, bs) in ws_bs_by_role.items():
        sd = nets[role].state_dict()
        names = [f"dense_encoder.encoder_layers.{i}" for i in (0, 2)] + [f"mlp.{i}" for i in (0, 2, 4, 6)]
        for nm, w, b in zip(names, ws, bs):
            assert tuple(sd[nm + ".weight"].shape) == w.shape
            sd[nm + ".weight"] = torch.from_numpy(w.copy())
            sd[nm + ".bias"] = torch.from_numpy(b.copy())
        nets[role].load_state_dict(sd)
        nets[role].eval()
    return nets


def _env_rows(n_rows=64):
    """Observation rows as the env produces them (CPU oracle == HIP path bit for bit): 16 envs x 4 points of a random-action rollout
    on the small procedural test terrain."""
    _test_paths()
    from helpers import oracle_config_from, oracle_terrain, small_procedural
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    from oracle import rover_oracle as ro
    n = 16
    ter = small_procedural()
    ter.make_spawns(2 * n)
    cfg = RoverEnvCfg()
    cfg.scene.num_envs = n
    ocfg = oracle_config_from(ro, cfg.to_native())
    oter = oracle_terrain(ro, ter)
    S = ro.new_state(n)
    rows = [ro.reset_all(ocfg, oter, S)]
    rng = np.random.RandomState(11)
    for t in range(1, 16):
        o = ro.step(ocfg, oter, S, rng.uniform(-1, 1, (n, 2)).astype(np.float32))[0]
        if t % 5 == 0:
            rows.append(o)
    rows = np.concatenate(rows, 0)[:n_rows].astype(np.float32)
    assert rows.shape == (n_rows, 965) and np.isfinite(rows).all()
    return rows


def gen_policy_forward():
    import hashlib
    _test_paths()
    from helpers import random_policy_weights, synthetic_obs
    models = _load_reference_models()
    seeds = {"policy": 101, "value": 202}
    scale = 3.0                      # 3 x the torch default init range: O(1) outputs, the tanh head is exercised
    wb = {"policy": random_policy_weights(seed=seeds["policy"], out_dim=2, scale=scale),
          "value": random_policy_weights(seed=seeds["value"], out_dim=1, scale=scale)}
    nets = reference_networks(models, wb)
    syn_seed = 77
    syn = synthetic_obs(256, seed=syn_seed)
    env_rows = _env_rows(64)
    obs = torch.from_numpy(np.concatenate([syn, env_rows], 0))
    with torch.no_grad():
        mean, log_std, _ = nets["policy"].compute({"states": obs}, role="policy")       # models.py:89-102 (tanh head :86)
        value, _ = nets["value"].compute({"states": obs}, role="value")                   # models.py:151-163
    assert mean.shape == (320, 2) and value.shape == (320, 1) and float(log_std.abs().max()) == 0.0

    def digest(arrs):
        h = hashlib.sha256()
        for a in arrs:
            h.update(np.ascontiguousarray(a, dtype=np.float32).tobytes())
        return h.hexdigest()
    out = {"weight_seed_policy": np.int64(seeds["policy"]), "weight_seed_value": np.int64(seeds["value"]),
           "weight_scale": np.float64(scale), "synthetic_seed": np.int64(syn_seed), "synthetic_rows": np.int64(256),
           "weights_sha256_policy": np.array(digest(wb["policy"][0] + wb["policy"][1])),
           "weights_sha256_value": np.array(digest(wb["value"][0] + wb["value"][1])),
           "synthetic_sha256": np.array(digest([syn])),
           "env_rows": env_rows, "policy_mean": mean.numpy(), "value": value.numpy()}
    np.savez_compressed(os.path.join(OUT, "policy_forward.npz"), **out)
    print("policy_forward:", mean.shape, value.shape, "|mean| max", float(mean.abs().max()), "|value| max", float(value.abs().max()))


if __name__ == "__main__":
    gen_policy_forward()
    gen_lift_terms()
    gen_reset()
    gen_ackermann()
    gen_mdp_terms()
    gen_heightmap()
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))
