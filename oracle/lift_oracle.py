"""ctypes binding of the FrankaCubeLift-v0 CPU checker (oracle/lift_oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ROVER_ORACLE_DIR: load the library from another directory (the sanitizer builds of `make -C oracle sanitized`)
_LIB_PATH = os.path.join(os.environ.get("ROVER_ORACLE_DIR") or _HERE, "liblift_oracle.so")
NUM_REW, NUM_TERM, OBS, ACT, LOG_WORDS, STATE_WORDS = 6, 2, 36, 8, 16, 64
Q, QD, OBJ_POS, OBJ_QUAT, OBJ_LIN, OBJ_ANG, CMD, TIME_LEFT, EP_LEN, ACTION, PREV_ACTION, EP_SUM, RESET_COUNT = \
    0, 9, 18, 21, 25, 28, 31, 38, 39, 40, 48, 56, 62
Q_DEFAULT = np.array([0.0, -0.569, 0.0, -2.810, 0.0, 3.037, 0.741, 0.04, 0.04], np.float32)


class Config(C.Structure):
    """Mirror of ``struct lfo_config`` (oracle/lift_oracle.c) = ``struct lift_config`` of the C ABI (include/rover_lift.h)."""
    _fields_ = [
        ("sim_dt", C.c_float), ("decimation", C.c_int32), ("max_episode_length", C.c_int32), ("max_episode_length_s", C.c_float),
        ("action_scale", C.c_float), ("finger_open", C.c_float), ("finger_close", C.c_float),
        ("rew_weight", C.c_float * NUM_REW),
        ("reach_std", C.c_float), ("goal_std", C.c_float), ("goal_fine_std", C.c_float), ("minimal_height", C.c_float),
        ("drop_height", C.c_float), ("cmd_lo", C.c_float * 3), ("cmd_hi", C.c_float * 3), ("cmd_resample_time", C.c_float),
        ("obj_init", C.c_float * 3), ("obj_range_lo", C.c_float * 3), ("obj_range_hi", C.c_float * 3), ("ee_offset_z", C.c_float),
        ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32), ("solver_iterations", C.c_int32), ("mu_table", C.c_float),
        ("mu_pad", C.c_float),
    ]


def build(force: bool = False) -> str:
    if os.environ.get("ROVER_ORACLE_DIR"):
        return _LIB_PATH
    srcs = [os.path.join(_HERE, "lift_oracle.c")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblift_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.lfo_tanhf.restype = C.c_float
        _lib.lfo_tanhf.argtypes = [C.c_float]
        assert _lib.lfo_state_words() == STATE_WORDS and _lib.lfo_config_bytes() == C.sizeof(Config)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def default_config(**over) -> Config:
    c = Config()
    lib().lfo_default_config(C.byref(c))
    for k, v in over.items():
        if isinstance(v, (list, tuple, np.ndarray)):
            for i, x in enumerate(v):
                getattr(c, k)[i] = x
        else:
            setattr(c, k, v)
    return c


def terms(obj_pos, ee_pos, root_state, cmd, reach_std=0.1, goal_std=0.3, goal_fine_std=0.05, minimal_height=0.06):
    obj_pos, ee_pos, root_state, cmd = _f32(obj_pos), _f32(ee_pos), _f32(root_state), _f32(cmd)
    n = obj_pos.shape[0]
    lifted, reach, goal, fine = (np.zeros(n, np.float32) for _ in range(4))
    pos_b = np.zeros((n, 3), np.float32)
    lib().lfo_terms(n, _p(obj_pos), _p(ee_pos), _p(root_state), _p(cmd), C.c_float(reach_std), C.c_float(goal_std),
                    C.c_float(goal_fine_std), C.c_float(minimal_height), _p(lifted), _p(reach), _p(goal), _p(fine), _p(pos_b))
    return lifted, reach, goal, fine, pos_b


def new_state(n: int) -> np.ndarray:
    return np.zeros((n, STATE_WORDS), np.float32)


def reset(cfg: Config, state, env_id_offset=0):
    n = state.shape[0]
    obs = np.zeros((n, OBS), np.float32)
    lib().lfo_reset(C.byref(cfg), n, env_id_offset, _p(state), _p(obs))
    return obs


def step(cfg: Config, state, action, env_id_offset=0, log=None):
    assert state.dtype == np.float32 and state.flags.c_contiguous
    action = _f32(action)
    n = state.shape[0]
    obs, reward = np.zeros((n, OBS), np.float32), np.zeros(n, np.float32)
    term, trunc = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    if log is None:
        log = np.zeros(LOG_WORDS, np.float32)
    lib().lfo_step(C.byref(cfg), n, env_id_offset, _p(state), _p(action), _p(obs), _p(reward), _p(term), _p(trunc), _p(log))
    return obs, reward, term, trunc, log


def hand_pose(cfg: Config, q9, qd9=None):
    q9 = _f32(q9)
    qd9 = _f32(qd9) if qd9 is not None else np.zeros(9, np.float32)
    tcp, R, v, w = np.zeros(3, np.float32), np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().lfo_hand_pose(C.byref(cfg), _p(q9), _p(qd9), _p(tcp), _p(R), _p(v), _p(w))
    return tcp, R.reshape(3, 3), v, w


def gravity_torque(q7):
    tau = np.zeros(7, np.float32)
    lib().lfo_gravity_torque(_p(_f32(q7)), _p(tau))
    return tau


def inverse_dynamics(q7, qd7, qdd7, gravity=9.81):
    """tau = M(q) qdd + c(q, qd) + g(q): the model's recursive Newton-Euler pass"""
    q, qd, qdd = _f32(q7), _f32(qd7), _f32(qdd7)
    tau = np.zeros(7, np.float32)
    lib().lfo_inverse_dynamics(_p(q), _p(qd), _p(qdd), C.c_float(gravity), _p(tau))
    return tau


def max_threads() -> int:
    return int(lib().lfo_max_threads())


def model_constants() -> np.ndarray:
    n = lib().lfo_model_constants(None, 0)
    out = np.zeros(n, np.float32)
    lib().lfo_model_constants(_p(out), n)
    return out


def arm_substep(h, target7, q7, qd7):
    """One arm substep on copies of (q, qd); returns the new (q, qd)."""
    q, qd = _f32(q7).copy(), _f32(qd7).copy()
    lib().lfo_arm_substep(C.c_float(h), _p(_f32(target7)), _p(q), _p(qd))
    return q, qd


def mass_matrix(q7):
    M = np.zeros(49, np.float32)
    lib().lfo_mass_matrix(_p(_f32(q7)), _p(M))
    return M.reshape(7, 7)
