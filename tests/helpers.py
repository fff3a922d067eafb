"""Shared helpers of the test-suite: small terrains and oracle <-> env plumbing."""
from __future__ import annotations

import functools

import numpy as np

from isaac_rover_orbit_amd import terrain as T


@functools.lru_cache(maxsize=4)
def small_procedural(size=1024, seed=7, sigma_z=0.15, n_rocks=120):
    """A 51.2 m procedural terrain (spawn band 20 m .. 31.2 m) -- generated once per session."""
    return T.make_procedural_terrain((size, size), seed=seed, sigma_z=sigma_z, n_rocks=n_rocks)


@functools.lru_cache(maxsize=2)
def flat(size=1024):
    return T.make_flat_terrain((size, size))


def oracle_terrain(ro, ter, n_spawns=None, spawn_seed=41):
    if n_spawns is not None and (ter.spawn_locations is None or ter.spawn_locations.shape[0] != n_spawns):
        ter.make_spawns(n_spawns, seed=spawn_seed)
    return ro.TerrainData(ter.height, ter.obstacle, ter.safe_rock_mask, ter.resolution, ter.min_x, ter.min_y,
                          ter.spawn_locations, lookup=getattr(ter, "lookup_height", None))


def oracle_config_from(ro, native):
    """Copy a ``_lib.RoverConfig`` (product side) field by field into the oracle's Config struct."""
    cfg = ro.Config()
    for name, _ in ro.Config._fields_:
        v = getattr(native, name)
        if name == "rew_weight":
            for i in range(7):
                cfg.rew_weight[i] = v[i]
        else:
            setattr(cfg, name, v)
    return cfg


def assert_close(a, b, atol, rtol, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    fin = np.isfinite(a) & np.isfinite(b)
    assert (np.isfinite(a) == np.isfinite(b)).all(), f"{what}: non-finite pattern differs"
    assert (a[~fin] == b[~fin]).all(), f"{what}: infinities differ"
    err = np.abs(a[fin] - b[fin])
    tol = atol + rtol * np.abs(b[fin])
    if not (err <= tol).all():
        i = int(np.argmax(err - tol))
        raise AssertionError(f"{what}: max |err| {err.max():.3e} (worst excess at flat idx {i}: {a[fin][i]} vs {b[fin][i]})")


# ---------------------------------------------------------------------------------------------- policy networks
def random_policy_weights(seed=0, out_dim=2, scale=1.0):
    """Random-init weights of the reference architecture (get_models.py:36-62), torch Linear default init ranges."""
    rng = np.random.RandomState(seed)
    K, N = [961, 80, 64, 256, 160, 128], [80, 60, 256, 160, 128, out_dim]
    ws = [(rng.uniform(-1, 1, (n, k)) / np.sqrt(k) * scale).astype(np.float32) for k, n in zip(K, N)]
    bs = [(rng.uniform(-1, 1, (n,)) / np.sqrt(k) * scale).astype(np.float32) for k, n in zip(K, N)]
    return ws, bs


def torch_policy_reference(ws, bs, obs, final_tanh=True, device="cpu"):
    """Plain torch fp32 restatement of GaussianNeuralNetwork.compute / DeterministicNeuralNetwork.compute
    (rover_envs/envs/navigation/learning/skrl/models.py:89-103, 151-163)."""
    import torch
    act = torch.nn.functional.leaky_relu
    W = [torch.as_tensor(w, device=device) for w in ws]
    B = [torch.as_tensor(b, device=device) for b in bs]
    s = torch.as_tensor(obs, device=device)
    e = s[:, 3:-1]
    e = act(torch.nn.functional.linear(e, W[0], B[0]))
    e = act(torch.nn.functional.linear(e, W[1], B[1]))
    x = torch.cat([s[:, 0:4], e], 1)
    for i in (2, 3, 4):
        x = act(torch.nn.functional.linear(x, W[i], B[i]))
    x = torch.nn.functional.linear(x, W[5], B[5])
    return (torch.tanh(x) if final_tanh else x).cpu().numpy()


def synthetic_obs(n, seed=0):
    """Observation rows shaped like the env's: actions in [-1, 1], distance / heading terms, height scan around -0.27."""
    rng = np.random.RandomState(seed)
    obs = np.empty((n, 965), dtype=np.float32)
    obs[:, 0:2] = rng.uniform(-1, 1, (n, 2))
    obs[:, 2] = rng.uniform(0, 1.2, n)
    obs[:, 3] = rng.uniform(-1, 1, n)
    obs[:, 4:] = -0.27 + 0.15 * rng.standard_normal((n, 961))
    return obs


def policy_forward_fixture(golden_dir):
    """tests/golden/policy_forward.npz (tools/gen_golden.py::gen_policy_forward): outputs of the REFERENCE's own
    GaussianNeuralNetwork.compute / DeterministicNeuralNetwork.compute (learning/skrl/models.py:89-102, 151-163).  The weights
    and the synthetic rows are regenerated from the recorded seeds and checked against the recorded digests.
    Returns {"policy": (ws, bs), "value": (ws, bs)}, obs (320, 965), expected mean (320, 2), expected value (320, 1)."""
    import hashlib
    import os
    g = np.load(os.path.join(golden_dir, "policy_forward.npz"))

    def digest(arrs):
        h = hashlib.sha256()
        for a in arrs:
            h.update(np.ascontiguousarray(a, dtype=np.float32).tobytes())
        return h.hexdigest()
    scale = float(g["weight_scale"])
    nets = {"policy": random_policy_weights(seed=int(g["weight_seed_policy"]), out_dim=2, scale=scale),
            "value": random_policy_weights(seed=int(g["weight_seed_value"]), out_dim=1, scale=scale)}
    for role, (ws, bs) in nets.items():
        assert digest(ws + bs) == str(g[f"weights_sha256_{role}"]), f"{role}: regenerated weights differ from the fixture's"
    syn = synthetic_obs(int(g["synthetic_rows"]), seed=int(g["synthetic_seed"]))
    assert digest([syn]) == str(g["synthetic_sha256"]), "regenerated synthetic rows differ from the fixture's"
    obs = np.concatenate([syn, g["env_rows"]], 0).astype(np.float32)
    return nets, obs, g["policy_mean"], g["value"]


# ---------------------------------------------------------------------------------------------- reset fixture
def reset_fixture_case(g, batch):
    """Per-env arrays (scattered from the batch's env_ids) of tests/golden/reset.npz: what ``reset_with_draws`` consumes
    and what the reference produced (tools/gen_golden.py::gen_reset)."""
    n, mt = int(g["num_envs"]), int(g["max_tries"])
    pre = f"b{batch}_"
    ids = g[pre + "env_ids"]
    mask = np.zeros(n, np.uint8)
    mask[ids] = 1
    spawn_row = np.zeros(n, np.int32)
    spawn_row[ids] = g[pre + "spawn_index"]
    yaw_u = np.zeros(n, np.float32)
    yaw_u[ids] = g[pre + "yaw_u"]
    theta_u = np.full((n, mt), np.nan, np.float32)
    theta_u[ids] = g[pre + "theta_u"]
    heading_u = np.zeros(n, np.float32)
    heading_u[ids] = g[pre + "heading_u"]
    expect = {"ids": ids, "root_pose": g[pre + "root_pose"], "env_origins": g[pre + "env_origins"],
              "pos_command_w": g[pre + "pos_command_w"], "heading_command_w": g[pre + "heading_command_w"]}
    return mask, spawn_row, yaw_u, theta_u, heading_u, expect


def check_reset_against_fixture(state_aos, expect, words):
    """state_aos: (n, 72) state words after the injected reset; ``words`` = module with the word offsets."""
    ids = expect["ids"]
    S = state_aos[ids]
    # spawn row + z_offset: float adds only => exact; yaw -> quaternion uses cos/sin (torch vs rv_*: <= 2 ulp)
    assert_close(S[:, words.POS:words.POS + 3], expect["root_pose"][:, 0:3], 0, 0, "root position (randomizations.py:22-27)")
    assert_close(S[:, words.QUAT:words.QUAT + 4], expect["root_pose"][:, 3:7], 2.5e-7, 0, "root quaternion (:30-33)")
    assert_close(S[:, words.ENV_ORIGIN:words.ENV_ORIGIN + 3], expect["env_origins"], 0, 0, "env_origins (:37)")
    # target = origin + 9 (cos, sin)(theta): |err| <= 9 * 2 ulp(1) + ulp(30); z = heightmap cell (exact)
    assert_close(S[:, words.TARGET_W:words.TARGET_W + 2], expect["pos_command_w"][:, 0:2], 4e-6, 0,
                 "target xy (terrain_importer.py:168-173)")
    assert_close(S[:, words.TARGET_W + 2], expect["pos_command_w"][:, 2], 0, 0, "target z (get_height_at, terrain_utils.py:62-84)")
    assert_close(S[:, words.HEADING_CMD_W], expect["heading_command_w"], 2.4e-7, 0, "heading command (:93-95)")


def gfx950_code_objects(lib_path):
    """The gfx950 code objects embedded in a HIP shared library: the .hip_fatbin section is a sequence of clang offload bundles
    (one per translation unit): magic, u64 entry count, entries {u64 offset, u64 size, u64 id length, id}."""
    import struct
    data = open(lib_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], data.find(magic)
    while pos >= 0:
        n = struct.unpack_from("<Q", data, pos + len(magic))[0]
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", data, q)
            ident = data[q + 24:q + 24 + idlen].decode()
            q += 24 + idlen
            if "gfx950" in ident and size > 0:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(magic, pos + len(magic))
    return out
