"""ctypes binding of the CPU oracle (oracle/rover_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product package ``isaac_rover_orbit_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ROVER_ORACLE_DIR: load the library from another directory (the sanitizer builds of `make -C oracle sanitized`)
_LIB_PATH = os.path.join(os.environ.get("ROVER_ORACLE_DIR") or _HERE, "librover_oracle.so")

NUM_REW, NUM_TERM, NUM_BODIES, LOG_WORDS = 7, 4, 13, 16

# state word offsets (rover_oracle.h)
POS, QUAT, LINVEL, ANGVEL = 0, 3, 7, 10
BOGIE_Q, STEER_Q, WHEEL_Q, BOGIE_QD, STEER_QD, WHEEL_QD = 13, 16, 20, 26, 29, 33
TARGET_W, HEADING_CMD_W, ENV_ORIGIN, ACTION, PREV_ACTION, TIME_LEFT, EP_LEN = 39, 42, 43, 46, 48, 50, 51
CMD_B, HEADING_CMD_B, EP_SUM, METRIC_POS, METRIC_HEAD, LAMBDA_N, RESET_COUNT = 52, 55, 56, 63, 64, 65, 71
STATE_WORDS = 72


class Config(C.Structure):
    _fields_ = [
        ("scale_lin", C.c_float), ("scale_ang", C.c_float), ("offset_lin", C.c_float), ("offset_ang", C.c_float),
        ("wheel_radius", C.c_float), ("d_fr", C.c_float), ("d_mw", C.c_float), ("wheelbase", C.c_float),
        ("sim_dt", C.c_float), ("decimation", C.c_int32), ("max_episode_length", C.c_int32),
        ("max_episode_length_s", C.c_float),
        ("success_threshold", C.c_float), ("far_threshold", C.c_float), ("target_distance", C.c_float),
        ("heading_lo", C.c_float), ("heading_hi", C.c_float), ("resample_time", C.c_float),
        ("rew_weight", C.c_float * NUM_REW),
        ("obs_scale_distance", C.c_float), ("obs_scale_heading", C.c_float),
        ("scan_resolution", C.c_float), ("scan_size_x", C.c_float), ("scan_size_y", C.c_float),
        ("scan_height_offset", C.c_float), ("scan_nx", C.c_int32), ("scan_ny", C.c_int32),
        ("reset_z_offset", C.c_float), ("reset_mode", C.c_int32), ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
        ("friction_mu", C.c_float), ("solver_iterations", C.c_int32), ("max_target_tries", C.c_int32),
        ("step_mapping", C.c_int32), ("spawn_draw", C.c_int32), ("counter_lo", C.c_uint32), ("counter_hi", C.c_uint32),
        ("scan_surface", C.c_int32), ("mass_model", C.c_int32),
        ("rew_success_threshold", C.c_float), ("rew_far_threshold", C.c_float),
    ]


class Terrain(C.Structure):
    _fields_ = [
        ("height", C.c_void_p), ("obstacle", C.c_void_p), ("safe_mask", C.c_void_p),
        ("H", C.c_int32), ("W", C.c_int32), ("resolution", C.c_float), ("min_x", C.c_float), ("min_y", C.c_float),
        ("spawns", C.c_void_p), ("n_spawns", C.c_int32), ("lookup", C.c_void_p),
    ]


def build(force: bool = False) -> str:
    if os.environ.get("ROVER_ORACLE_DIR"):
        return _LIB_PATH
    src = os.path.join(_HERE, "rover_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librover_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.rvo_state_words.restype = C.c_int
        _lib.rvo_num_threads.restype = C.c_int
        _lib.rvo_model_constants.restype = C.c_int
        assert _lib.rvo_state_words() == STATE_WORDS
        if os.environ.get("RVO_MODEL_VARIANT"):   # dynamics study only (tools/dynamics_study.py): the physics checks under a variant
            _lib.rvo_set_model_variant(int(os.environ["RVO_MODEL_VARIANT"]))
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def default_config(**overrides) -> Config:
    cfg = Config()
    lib().rvo_default_config(C.byref(cfg))
    for k, v in overrides.items():
        if k == "rew_weight":
            for i, w in enumerate(v):
                cfg.rew_weight[i] = w
        else:
            setattr(cfg, k, v)
    return cfg


class TerrainData:
    """Keeps the numpy arrays alive behind the C struct."""

    def __init__(self, height, obstacle=None, safe_mask=None, resolution=0.05, min_x=0.0, min_y=0.0, spawns=None,
                 lookup=None):
        self.height = _f32(height)
        self.lookup = _f32(lookup) if lookup is not None else None
        H, W = self.height.shape
        self.obstacle = _f32(obstacle) if obstacle is not None else np.zeros((H, W), np.float32)
        self.safe_mask = (np.ascontiguousarray(safe_mask, dtype=np.uint8) if safe_mask is not None
                          else np.zeros((H, W), np.uint8))
        self.spawns = _f32(spawns) if spawns is not None else np.zeros((1, 3), np.float32)
        self.c = Terrain(_p(self.height), _p(self.obstacle), _p(self.safe_mask), H, W, resolution, min_x, min_y,
                         _p(self.spawns), self.spawns.shape[0], _p(self.lookup))


def model_constants() -> np.ndarray:
    n = lib().rvo_model_constants(None, 0)
    out = np.zeros(n, np.float32)
    lib().rvo_model_constants(_p(out), n)
    return out


def ackermann(cfg: Config, raw):
    raw = _f32(raw)
    n = raw.shape[0]
    processed, steer, wheel = np.zeros((n, 2), np.float32), np.zeros((n, 4), np.float32), np.zeros((n, 6), np.float32)
    lib().rvo_ackermann(C.byref(cfg), n, _p(raw), _p(processed), _p(steer), _p(wheel))
    return processed, steer, wheel


def mdp_terms(cfg: Config, cmd_b, action, prev_action, ep_len, force):
    cmd_b, action, prev_action = _f32(cmd_b), _f32(action), _f32(prev_action)
    force = _f32(force).reshape(cmd_b.shape[0], NUM_BODIES * 3)
    ep_len = np.ascontiguousarray(ep_len, dtype=np.int32)
    n = cmd_b.shape[0]
    od, oa = np.zeros(n, np.float32), np.zeros(n, np.float32)
    rew, term = np.zeros((n, NUM_REW), np.float32), np.zeros((n, NUM_TERM), np.uint8)
    lib().rvo_mdp_terms(C.byref(cfg), n, _p(cmd_b), _p(action), _p(prev_action), _p(ep_len), _p(force), _p(od), _p(oa),
                        _p(rew), _p(term))
    return od, oa, rew, term


def height_scan_term(cfg: Config, pos_z, hit_z):
    pos_z, hit_z = _f32(pos_z), _f32(hit_z)
    out = np.zeros_like(hit_z)
    lib().rvo_height_scan_term(C.byref(cfg), hit_z.shape[0], hit_z.shape[1], _p(pos_z), _p(hit_z), _p(out))
    return out


def get_height_at(t: TerrainData, xy):
    xy = _f32(xy)
    out = np.zeros(xy.shape[0], np.float32)
    lib().rvo_get_height_at(C.byref(t.c), xy.shape[0], _p(xy), _p(out))
    return out


def target_invalid(t: TerrainData, xy):
    xy = _f32(xy)
    out = np.zeros(xy.shape[0], np.uint8)
    lib().rvo_target_invalid(C.byref(t.c), xy.shape[0], _p(xy), _p(out))
    return out


def update_command(pos, quat, target_w, heading_cmd_w):
    pos, quat, target_w, heading_cmd_w = _f32(pos), _f32(quat), _f32(target_w), _f32(heading_cmd_w)
    n = pos.shape[0]
    cmd_b, hb = np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    lib().rvo_update_command(n, _p(pos), _p(quat), _p(target_w), _p(heading_cmd_w), _p(cmd_b), _p(hb))
    return cmd_b, hb


def philox(c0, c1, c2, c3, k0, k1):
    out = (C.c_uint32 * 4)()
    lib().rvo_philox4x32(C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3), C.c_uint32(k0), C.c_uint32(k1), out)
    return [int(x) for x in out]


def terrain_sample(t: TerrainData, xy):
    xy = _f32(xy)
    n = xy.shape[0]
    h, gx, gy, ob = (np.zeros(n, np.float32) for _ in range(4))
    lib().rvo_terrain_sample(C.byref(t.c), n, _p(xy), _p(h), _p(gx), _p(gy), _p(ob))
    return h, gx, gy, ob


def height_scan(cfg: Config, t: TerrainData, state):
    state = _f32(state)
    n = state.shape[0]
    out = np.zeros((n, cfg.scan_nx * cfg.scan_ny), np.float32)
    lib().rvo_height_scan(C.byref(cfg), C.byref(t.c), n, _p(state), _p(out))
    return out


def physics_step(cfg: Config, t: TerrainData, state, steer_target, wheel_target, substeps=1, want_force=True):
    assert state.dtype == np.float32 and state.flags.c_contiguous
    n = state.shape[0]
    st, wt = _f32(steer_target), _f32(wheel_target)
    force = np.zeros((n, NUM_BODIES, 3), np.float32) if want_force else None
    lib().rvo_physics_step(C.byref(cfg), C.byref(t.c), n, _p(state), _p(st), _p(wt), substeps, _p(force))
    return force


def new_state(n: int) -> np.ndarray:
    s = np.zeros((n, STATE_WORDS), np.float32)
    s[:, QUAT] = 1.0
    return s


def _bump_counter(cfg: Config):
    """The call counter (number of reset / step calls so far) lives in the config struct on the oracle side: the C functions
    read it, this binding advances it after every reset_all / step, like the HIP library does inside its handle."""
    c = ((cfg.counter_hi << 32) | cfg.counter_lo) + 1
    cfg.counter_lo, cfg.counter_hi = c & 0xFFFFFFFF, (c >> 32) & 0xFFFFFFFF


def reset_all(cfg: Config, t: TerrainData, state, env_id_offset=0):
    n = state.shape[0]
    obs = np.zeros((n, 4 + cfg.scan_nx * cfg.scan_ny), np.float32)
    lib().rvo_reset_all(C.byref(cfg), C.byref(t.c), n, env_id_offset, _p(state), _p(obs))
    _bump_counter(cfg)
    return obs


def reset_with_draws(cfg: Config, t: TerrainData, state, mask, spawn_row, yaw_u, theta_u, heading_u, env_id_offset=0):
    """``_reset_idx`` of the masked envs with the reference's recorded torch draws injected (tests/golden/reset.npz)."""
    assert state.dtype == np.float32 and state.flags.c_contiguous
    n = state.shape[0]
    mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    spawn_row = np.ascontiguousarray(spawn_row, dtype=np.int32)
    yaw_u, theta_u, heading_u = _f32(yaw_u), _f32(theta_u), _f32(heading_u)
    assert theta_u.shape == (n, cfg.max_target_tries)
    obs = np.zeros((n, 4 + cfg.scan_nx * cfg.scan_ny), np.float32)
    lib().rvo_reset_with_draws(C.byref(cfg), C.byref(t.c), n, env_id_offset, _p(state), _p(mask), _p(spawn_row), _p(yaw_u),
                               _p(theta_u), _p(heading_u), _p(obs))
    return obs


def step(cfg: Config, t: TerrainData, state, action, env_id_offset=0, log=None):
    assert state.dtype == np.float32 and state.flags.c_contiguous
    action = _f32(action)
    n = state.shape[0]
    obs = np.zeros((n, 4 + cfg.scan_nx * cfg.scan_ny), np.float32)
    reward = np.zeros(n, np.float32)
    terminated, truncated = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    force = np.zeros((n, NUM_BODIES, 3), np.float32)
    if log is None:
        log = np.zeros(LOG_WORDS, np.float32)
    lib().rvo_step(C.byref(cfg), C.byref(t.c), n, env_id_offset, _p(state), _p(action), _p(obs), _p(reward),
                   _p(terminated), _p(truncated), _p(force), _p(log))
    _bump_counter(cfg)
    return obs, reward, terminated, truncated, force, log


def set_num_threads(n: int):
    """Threads of the OpenMP loops (0 = the OpenMP default = all host cores)."""
    lib().rvo_set_num_threads(int(n))


def set_model_variant(v: int):
    """Study variants of the dynamics model (rover_oracle.c; DESIGN.md section 4, docs/history.md section 5): bit 0 = wheel contact on the triangle surface,
    bit 1 = coupled 9 x 9 mass matrix + sequential PGS.  0 = the model the HIP path implements (the only value tests use)."""
    lib().rvo_set_model_variant(int(v))


def max_threads() -> int:
    """The OpenMP default thread count (what ``set_num_threads(0)`` selects)."""
    lib().rvo_set_num_threads(0)
    return int(lib().rvo_num_threads())


def int_view(state):
    return state.view(np.int32)
