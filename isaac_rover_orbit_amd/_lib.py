"""ctypes loader of the C-ABI HIP library ``librover_hip.so`` (declared in ``include/rover_hip.h``).

The product path has NO CPU fallback: if the library is missing or a call fails, a ``RoverHipError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "librover_hip.so")

NUM_REW, NUM_TERM, NUM_BODIES, LOG_WORDS, STATE_WORDS = 7, 4, 13, 16, 72

# state word offsets (include/rover_hip.h)
POS, QUAT, LINVEL, ANGVEL = 0, 3, 7, 10
BOGIE_Q, STEER_Q, WHEEL_Q, BOGIE_QD, STEER_QD, WHEEL_QD = 13, 16, 20, 26, 29, 33
TARGET_W, HEADING_CMD_W, ENV_ORIGIN, ACTION, PREV_ACTION, TIME_LEFT, EP_LEN = 39, 42, 43, 46, 48, 50, 51
CMD_B, HEADING_CMD_B, EP_SUM, METRIC_POS, METRIC_HEAD, LAMBDA_N, RESET_COUNT = 52, 55, 56, 63, 64, 65, 71

EXPORTS = [
    "rover_default_config", "rover_create", "rover_destroy", "rover_set_terrain", "rover_set_terrain_q16", "rover_workspace_bytes", "rover_bind",
    "rover_reset", "rover_reset_with_draws", "rover_set_seed", "rover_get_counter", "rover_set_counter", "rover_step", "rover_profile_step",
    "rover_profile_event_overhead", "rover_mdp_terms", "rover_ackermann", "rover_height_scan", "rover_physics", "rover_model_constants",
    "rover_state_words", "rover_config_bytes", "rover_last_error", "rover_version", "rover_set_markers", "rover_kernel_names",
    "rover_set_log_deferred", "rover_flush_log", "rover_step_begin", "rover_step_finish", "rover_set_obs_streaming",
    "rover_terrain_rasterize", "rover_terrain_surface", "rover_terrain_rock_mask", "rover_terrain_scratch_bytes",   # rover_terrain.h
    "rover_set_terrain_lookup",
    "rover_policy_default_desc", "rover_policy_packed_floats", "rover_policy_pack", "rover_policy_forward",  # rover_policy.h
    "rover_policy_forward_pair",
    "rover_lift_default_config", "rover_lift_config_bytes", "rover_lift_state_words", "rover_lift_create", "rover_lift_destroy",
    "rover_lift_workspace_bytes", "rover_lift_bind", "rover_lift_reset", "rover_lift_step", "rover_lift_terms",   # rover_lift.h
    "rover_lift_model_constants", "rover_lift_set_seed", "rover_lift_profile_step", "rover_lift_kernel_name",
    "rover_lift_set_log_deferred", "rover_lift_flush_log",
]
POLICY_MAX_LAYERS = 8
ACT_NONE, ACT_LEAKY_RELU, ACT_TANH = 0, 1, 2


class PolicyLayer(C.Structure):
    """Mirror of ``struct rover_policy_layer``."""
    _fields_ = [("K", C.c_int32), ("N", C.c_int32), ("act", C.c_int32), ("split_k", C.c_int32),
                ("w_off", C.c_uint32), ("b_off", C.c_uint32)]


class PolicyDesc(C.Structure):
    """Mirror of ``struct rover_policy_desc``."""
    _fields_ = [("obs_dim", C.c_int32), ("prop_dim", C.c_int32), ("enc_offset", C.c_int32), ("enc_dim", C.c_int32),
                ("n_enc", C.c_int32), ("n_mlp", C.c_int32), ("leaky_slope", C.c_float),
                ("layers", PolicyLayer * POLICY_MAX_LAYERS)]



class RoverHipError(RuntimeError):
    pass


LIFT_NUM_REW, LIFT_OBS, LIFT_ACT, LIFT_STATE_WORDS, LIFT_LOG_WORDS = 6, 36, 8, 64, 16
# lift state word offsets (include/rover_lift.h)
LIFT_Q, LIFT_QD, LIFT_OBJ_POS, LIFT_OBJ_QUAT, LIFT_OBJ_LIN, LIFT_OBJ_ANG, LIFT_CMD = 0, 9, 18, 21, 25, 28, 31
LIFT_TIME_LEFT, LIFT_EP_LEN, LIFT_ACTION, LIFT_PREV_ACTION, LIFT_EP_SUM, LIFT_RESET_COUNT = 38, 39, 40, 48, 56, 62


class LiftConfig(C.Structure):
    """Mirror of ``struct lift_config`` (include/rover_lift.h)."""
    _fields_ = [
        ("sim_dt", C.c_float), ("decimation", C.c_int32), ("max_episode_length", C.c_int32), ("max_episode_length_s", C.c_float),
        ("action_scale", C.c_float), ("finger_open", C.c_float), ("finger_close", C.c_float),
        ("rew_weight", C.c_float * LIFT_NUM_REW),
        ("reach_std", C.c_float), ("goal_std", C.c_float), ("goal_fine_std", C.c_float), ("minimal_height", C.c_float),
        ("drop_height", C.c_float), ("cmd_lo", C.c_float * 3), ("cmd_hi", C.c_float * 3), ("cmd_resample_time", C.c_float),
        ("obj_init", C.c_float * 3), ("obj_range_lo", C.c_float * 3), ("obj_range_hi", C.c_float * 3), ("ee_offset_z", C.c_float),
        ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32), ("solver_iterations", C.c_int32), ("mu_table", C.c_float),
        ("mu_pad", C.c_float),
    ]


class RoverConfig(C.Structure):
    """Mirror of ``struct rover_config``."""
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


_lib = None


def _hip_runtimes() -> set:
    out = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    out.add(line.split()[-1])
    except OSError:
        pass
    return out


def load():
    """Load ``librover_hip.so`` (built in-tree by ``__graft_entry__.build()`` / ``isaac_rover_orbit_amd.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RoverHipError(
            f"{LIB_PATH} not found: build it with `python -m isaac_rover_orbit_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the rover hot path.")
    # torch-ROCm bundles its own libamdhip64.so.7; device pointers and streams are only meaningful inside ONE HIP runtime,
    # so torch must be loaded first: the dynamic linker then binds librover_hip.so to the runtime torch already mapped.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    runtimes = _hip_runtimes()
    if len(runtimes) > 1:
        raise RoverHipError(f"two HIP runtimes are mapped into this process ({sorted(runtimes)}): import torch before "
                            "loading librover_hip.so")
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.rover_default_config.argtypes = [C.POINTER(RoverConfig)]
    lib.rover_create.argtypes = [C.POINTER(RoverConfig), i32, i32, i32, C.POINTER(vp)]
    lib.rover_destroy.argtypes = [vp]
    lib.rover_set_terrain.argtypes = [vp, vp, vp, vp, i32, i32, f32, f32, f32, vp, i32]
    lib.rover_set_terrain_q16.argtypes = [vp, vp, f32]
    lib.rover_workspace_bytes.argtypes = [vp]
    lib.rover_workspace_bytes.restype = C.c_size_t
    lib.rover_bind.argtypes = [vp, vp, vp, C.c_size_t]
    lib.rover_reset.argtypes = [vp, vp, vp]
    lib.rover_reset_with_draws.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.rover_set_seed.argtypes = [vp, C.c_uint32, C.c_uint32]
    lib.rover_get_counter.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.rover_set_counter.argtypes = [vp, C.c_uint64]
    lib.rover_profile_event_overhead.argtypes = [vp, vp, i32, C.POINTER(C.c_float)]
    lib.rover_set_markers.argtypes = [vp, i32]
    lib.rover_set_log_deferred.argtypes = [vp, i32]
    lib.rover_flush_log.argtypes = [vp, vp, vp]
    lib.rover_set_obs_streaming.argtypes = [vp, i32]
    lib.rover_step_begin.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.rover_step_finish.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.rover_kernel_names.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t]
    lib.rover_mdp_terms.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.rover_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.rover_profile_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.rover_ackermann.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    lib.rover_height_scan.argtypes = [vp, vp, vp]
    lib.rover_physics.argtypes = [vp, vp, vp, i32, vp, vp]
    lib.rover_model_constants.argtypes = [vp, i32]
    lib.rover_terrain_rasterize.argtypes = [vp, vp, i32, vp, i32, i32, vp]
    lib.rover_terrain_surface.argtypes = [vp, vp, i32, vp, i32, i32, C.c_double, C.c_double, C.c_double, vp]
    lib.rover_set_terrain_lookup.argtypes = [vp, vp]
    lib.rover_terrain_rock_mask.argtypes = [vp, i32, i32, C.c_double, vp, vp, vp, vp]
    lib.rover_terrain_scratch_bytes.argtypes = [i32, i32]
    lib.rover_terrain_scratch_bytes.restype = C.c_size_t
    lib.rover_policy_default_desc.argtypes = [C.POINTER(PolicyDesc), i32, i32]
    lib.rover_policy_packed_floats.argtypes = [C.POINTER(PolicyDesc)]
    lib.rover_policy_packed_floats.restype = C.c_size_t
    lib.rover_policy_pack.argtypes = [C.POINTER(PolicyDesc), C.POINTER(vp), C.POINTER(vp), vp]
    lib.rover_policy_forward.argtypes = [C.POINTER(PolicyDesc), vp, i32, vp, i32, vp, vp]
    lib.rover_policy_forward_pair.argtypes = [C.POINTER(PolicyDesc), vp, C.POINTER(PolicyDesc), vp, i32, vp, i32, vp, vp, vp]
    lib.rover_lift_default_config.argtypes = [C.POINTER(LiftConfig)]
    lib.rover_lift_config_bytes.restype = C.c_size_t
    lib.rover_lift_create.argtypes = [C.POINTER(LiftConfig), i32, i32, i32, C.POINTER(vp)]
    lib.rover_lift_destroy.argtypes = [vp]
    lib.rover_lift_workspace_bytes.argtypes = [vp]
    lib.rover_lift_workspace_bytes.restype = C.c_size_t
    lib.rover_lift_bind.argtypes = [vp, vp, vp, C.c_size_t]
    lib.rover_lift_reset.argtypes = [vp, vp, vp]
    lib.rover_lift_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.rover_lift_model_constants.argtypes = [vp, i32]
    lib.rover_lift_set_seed.argtypes = [vp, C.c_uint32, C.c_uint32]
    lib.rover_lift_profile_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.rover_lift_set_log_deferred.argtypes = [vp, C.c_int32]
    lib.rover_lift_flush_log.argtypes = [vp, vp, vp]
    lib.rover_lift_kernel_name.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.rover_lift_debug_set_lanes.argtypes = [vp, i32]
    lib.rover_lift_terms.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.rover_last_error.restype = C.c_char_p
    lib.rover_version.restype = C.c_char_p
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int:
            fn.restype = C.c_int
    lib.rover_config_bytes.restype = C.c_size_t
    if lib.rover_config_bytes() != C.sizeof(RoverConfig):
        raise RoverHipError("struct rover_config of librover_hip.so does not match the Python mirror")
    if lib.rover_lift_config_bytes() != C.sizeof(LiftConfig) or lib.rover_lift_state_words() != LIFT_STATE_WORDS:
        raise RoverHipError("struct lift_config / lift state layout of librover_hip.so does not match the Python mirror")
    if lib.rover_state_words() != STATE_WORDS:
        raise RoverHipError("librover_hip.so state layout does not match the Python binding")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().rover_last_error().decode("utf-8", "replace")
        raise RoverHipError(f"{what} failed (code {rc}): {msg}")


def default_config() -> RoverConfig:
    cfg = RoverConfig()
    check(load().rover_default_config(C.byref(cfg)), "rover_default_config")
    return cfg
