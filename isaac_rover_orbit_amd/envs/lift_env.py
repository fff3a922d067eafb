"""``FrankaCubeLiftEnv`` -- the gymnasium / ORBIT ``RLTaskEnv``-shaped boundary of the MI355X-native ``FrankaCubeLift-v0``
path (BASELINE config 5).

Reference: gym id ``FrankaCubeLift-v0`` = ORBIT ``RLTaskEnv`` on ``FrankaCubeLiftEnvCfg``
(``rover_envs/envs/manipulation/config/franka/__init__.py:6-14``, ``joint_pos_env_cfg.py:25-82``,
``manipulation_env_cfg.py:93-235``, ``mdp/rewards.py``, ``mdp/observations.py``).  ``step()`` is two HIP launches through the
C ABI of ``include/rover_lift.h``; the model that stands in for PhysX (7-DOF arm dynamics, gripper, cube / table / finger
contact) is specified in docs/history.md section 9 (f-4) and implemented in ``csrc/lift_kernels.hip`` (eight lanes per env) -- parity of
that layer is unpinned.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from .. import _lib
from ._logdict import LogDict
from .rover_env import RLTaskEnv, _ptr, _spaces

REWARD_ORDER = ["reaching_object", "lifting_object", "object_goal_tracking", "object_goal_tracking_fine_grained", "action_rate",
                "joint_vel"]                                                   # manipulation_env_cfg.py:118-144
TERMINATION_ORDER = ["time_out", "object_dropping"]                            # :147-157
OBS_TERMS = [("joint_pos", 9), ("joint_vel", 9), ("object_pose", 3), ("target_object_position", 7), ("actions", 8)]   # :104-110
LOG_KEYS = [f"Episode Reward/{k}" for k in REWARD_ORDER] + [f"Episode Termination/{k}" for k in TERMINATION_ORDER]


@dataclass
class _Scene:
    num_envs: int = 4096           # manipulation_env_cfg.py:197
    env_spacing: float = 2.5


@dataclass
class _Sim:
    dt: float = 0.01               # :232
    device: str = "cuda:0"


@dataclass
class LiftEnvCfg:
    """``FrankaCubeLiftEnvCfg`` in the shape the kernels consume (values = the reference cfg files cited above)."""
    scene: _Scene = field(default_factory=_Scene)
    sim: _Sim = field(default_factory=_Sim)
    decimation: int = 2
    episode_length_s: float = 5.0
    reward_weights: dict = field(default_factory=lambda: dict(zip(REWARD_ORDER, (1.0, 15.0, 16.0, 5.0, 1e-3, 1e-4))))
    reach_std: float = 0.1
    goal_std: float = 0.3
    goal_fine_std: float = 0.05
    minimal_height: float = 0.06
    drop_height: float = -0.05
    command_ranges: dict = field(default_factory=lambda: {"pos_x": (0.3, 0.7), "pos_y": (0.3, 0.7), "pos_z": (0.0, 0.0)})
    command_resampling_time: float = 5.0
    object_init_pos: tuple = (0.5, 0.0, 0.055)
    object_pose_range: dict = field(default_factory=lambda: {"x": (-0.1, 0.1), "y": (-0.25, 0.25), "z": (0.0, 0.0)})
    action_scale: float = 0.5
    finger_open: float = 0.04
    finger_close: float = 0.0
    ee_offset_z: float = 0.1034
    seed: int = 0
    solver_iterations: int = 8
    # extras["log"]: "on_demand" reduces the episodic sums of the envs that reset when the dictionary is READ (same numbers, no
    # second kernel launch per step); "every_step" runs the reduction behind every step like the C entry's default
    log_reduction: str = "on_demand"
    env_id_offset: int = 0

    @property
    def max_episode_length(self) -> int:
        return math.ceil(self.episode_length_s / (self.sim.dt * self.decimation))

    def to_native(self) -> "_lib.LiftConfig":
        c = _lib.LiftConfig()
        _lib.check(_lib.load().rover_lift_default_config(C.byref(c)), "rover_lift_default_config")
        c.sim_dt, c.decimation = self.sim.dt, self.decimation
        c.max_episode_length, c.max_episode_length_s = self.max_episode_length, self.episode_length_s
        c.action_scale, c.finger_open, c.finger_close = self.action_scale, self.finger_open, self.finger_close
        for i, k in enumerate(REWARD_ORDER):
            c.rew_weight[i] = self.reward_weights[k]
        c.reach_std, c.goal_std, c.goal_fine_std = self.reach_std, self.goal_std, self.goal_fine_std
        c.minimal_height, c.drop_height = self.minimal_height, self.drop_height
        for i, k in enumerate(("pos_x", "pos_y", "pos_z")):
            c.cmd_lo[i], c.cmd_hi[i] = self.command_ranges[k]
        c.cmd_resample_time = self.command_resampling_time
        for i, k in enumerate(("x", "y", "z")):
            c.obj_init[i] = self.object_init_pos[i]
            c.obj_range_lo[i], c.obj_range_hi[i] = self.object_pose_range[k]
        c.ee_offset_z = self.ee_offset_z
        c.seed_lo, c.seed_hi = self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF
        c.solver_iterations = self.solver_iterations
        return c


class _Managers:
    pass


class FrankaCubeLiftEnv(RLTaskEnv):
    """MI355X-native ``FrankaCubeLift-v0``."""

    def __init__(self, cfg: LiftEnvCfg | None = None, render_mode=None, **kwargs):
        self.cfg = cfg if cfg is not None else LiftEnvCfg()
        if not torch.cuda.is_available():
            raise _lib.RoverHipError("FrankaCubeLiftEnv needs a ROCm GPU: the path is HIP-only (no CPU fallback)")
        self.device = torch.device(self.cfg.sim.device)
        self.num_envs = n = int(self.cfg.scene.num_envs)
        self._lib = _lib.load()
        self._native_cfg = self.cfg.to_native()
        self.max_episode_length = self.cfg.max_episode_length
        self.step_dt = self.cfg.sim.dt * self.cfg.decimation
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(self.device):
            h = C.c_void_p()
            _lib.check(self._lib.rover_lift_create(C.byref(self._native_cfg), n, int(self.cfg.env_id_offset), dev_index, C.byref(h)),
                       "rover_lift_create")
            self._h = h
            self.state = torch.zeros(_lib.LIFT_STATE_WORDS, n, dtype=torch.float32, device=self.device)
            ws = int(self._lib.rover_lift_workspace_bytes(h))
            self._workspace = torch.zeros(max(ws, 4) // 4, dtype=torch.float32, device=self.device)
            _lib.check(self._lib.rover_lift_bind(h, _ptr(self.state), _ptr(self._workspace), ws), "rover_lift_bind")
            self._obs = [torch.zeros(n, _lib.LIFT_OBS, device=self.device) for _ in range(2)]
            self._rew = [torch.zeros(n, device=self.device) for _ in range(2)]
            self._term = [torch.zeros(n, dtype=torch.uint8, device=self.device) for _ in range(2)]
            self._trunc = [torch.zeros(n, dtype=torch.uint8, device=self.device) for _ in range(2)]
            self._log = torch.zeros(_lib.LIFT_LOG_WORDS, device=self.device)
        self._cur = 0
        self._log_pending = False
        self._log_deferred = self.cfg.log_reduction == "on_demand"
        if self.cfg.log_reduction not in ("on_demand", "every_step"):
            raise ValueError("log_reduction must be 'on_demand' or 'every_step'")
        _lib.check(self._lib.rover_lift_set_log_deferred(self._h, int(self._log_deferred)), "rover_lift_set_log_deferred")
        self._log_dict = LogDict(self, {k: self._log[i] for i, k in enumerate(LOG_KEYS)})
        self.extras = {"log": self._log_dict, "episode": self._log_dict}
        self.episode_length_buf = self.state[_lib.LIFT_EP_LEN].view(torch.int32)
        self.obs_buf = {"policy": self._obs[0]}
        self.observation_manager = _Managers()
        self.observation_manager.group_obs_dim = {"policy": (_lib.LIFT_OBS,)}
        self.observation_manager.group_obs_term_dim = {"policy": [(d,) for _, d in OBS_TERMS]}
        self.observation_manager.active_terms = {"policy": [k for k, _ in OBS_TERMS]}
        self.action_manager = _Managers()
        self.action_manager.action_term_dim = [7, 1]
        self.action_manager.total_action_dim = _lib.LIFT_ACT
        self.command_manager = _Managers()
        self.command_manager.get_command = lambda name: self.state[_lib.LIFT_CMD:_lib.LIFT_CMD + 7].t()
        self.single_observation_space = _spaces.Dict({"policy": _spaces.Box(-np.inf, np.inf, (_lib.LIFT_OBS,), np.float32)})
        self.single_action_space = _spaces.Box(-np.inf, np.inf, (_lib.LIFT_ACT,), np.float32)
        self.observation_space = _spaces.Dict({"policy": _spaces.Box(-np.inf, np.inf, (n, _lib.LIFT_OBS), np.float32)})
        self.action_space = _spaces.Box(-np.inf, np.inf, (n, _lib.LIFT_ACT), np.float32)
        self.common_step_counter = 0
        self._closed = False

    @property
    def unwrapped(self):
        return self

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # views used by user code / tests
    @property
    def joint_pos(self):
        return self.state[_lib.LIFT_Q:_lib.LIFT_Q + 9].t()

    @property
    def object_pos_w(self):
        return self.state[_lib.LIFT_OBJ_POS:_lib.LIFT_OBJ_POS + 3].t()

    def seed(self, seed: int = -1) -> int:
        """gymnasium / ORBIT ``env.seed``: re-keys the counter-based RNG of every reset that follows (``rover_lift_set_seed``).
        A negative seed keeps the current key."""
        seed = int(seed)
        if seed >= 0:
            self.cfg.seed = seed
            lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
            self._native_cfg.seed_lo, self._native_cfg.seed_hi = lo, hi
            _lib.check(self._lib.rover_lift_set_seed(self._h, lo, hi), "rover_lift_set_seed")
        return int(self.cfg.seed)

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.seed(seed)
        obs = self._obs[self._cur]
        _lib.check(self._lib.rover_lift_reset(self._h, _ptr(obs), self._stream()), "rover_lift_reset")
        self.obs_buf = {"policy": obs}
        return self.obs_buf, self.extras

    def step(self, action: torch.Tensor):
        if action.dtype != torch.float32 or not action.is_contiguous() or action.device != self.device:
            action = action.to(device=self.device, dtype=torch.float32).contiguous()
        if action.shape != (self.num_envs, _lib.LIFT_ACT):
            raise ValueError(f"action must have shape ({self.num_envs}, {_lib.LIFT_ACT}), got {tuple(action.shape)}")
        self._cur ^= 1
        k = self._cur
        _lib.check(self._lib.rover_lift_step(self._h, _ptr(action), _ptr(self._obs[k]), _ptr(self._rew[k]), _ptr(self._term[k]),
                                             _ptr(self._trunc[k]), _ptr(self._log), self._stream()), "rover_lift_step")
        self.common_step_counter += 1
        self._log_pending = self._log_deferred
        self.obs_buf = {"policy": self._obs[k]}
        return self.obs_buf, self._rew[k], self._term[k].view(torch.bool), self._trunc[k].view(torch.bool), self.extras

    def flush_log(self):
        """Bring ``extras["log"]`` up to date (``rover_lift_flush_log``).  Called by the log dictionary itself on any read; only
        needed by code that kept a reference to one of its tensors from an earlier step."""
        if self._log_pending:
            self._log_pending = False
            _lib.check(self._lib.rover_lift_flush_log(self._h, _ptr(self._log), self._stream()), "rover_lift_flush_log")

    def profile_step(self, action: torch.Tensor):
        """``step`` with HIP-event timing: returns ``(ms_step_kernel, ms_log_kernel)``, event overhead included.  Syncs."""
        action = action.to(device=self.device, dtype=torch.float32).contiguous()
        if action.shape != (self.num_envs, _lib.LIFT_ACT):
            raise ValueError(f"action must have shape ({self.num_envs}, {_lib.LIFT_ACT}), got {tuple(action.shape)}")
        self._cur ^= 1
        k = self._cur
        a, b = C.c_float(0.0), C.c_float(0.0)
        _lib.check(self._lib.rover_lift_profile_step(self._h, _ptr(action), _ptr(self._obs[k]), _ptr(self._rew[k]), _ptr(self._term[k]),
                                                     _ptr(self._trunc[k]), _ptr(self._log), self._stream(), C.byref(a), C.byref(b)),
                   "rover_lift_profile_step")
        self.common_step_counter += 1
        self._log_pending = self._log_deferred
        self.obs_buf = {"policy": self._obs[k]}
        return a.value, b.value

    def kernel_name(self) -> str:
        """Name of the step kernel as rocprofv3's kernel trace prints it (``rover_lift_kernel_name``)."""
        buf = C.create_string_buffer(64)
        _lib.check(self._lib.rover_lift_kernel_name(self._h, buf, 64), "rover_lift_kernel_name")
        return buf.value.decode()

    # checkpoint / resume: the state words + the RNG key are the whole environment (counter-based Philox)
    def state_dict(self) -> dict:
        self.flush_log()
        return {"state": self.get_state().cpu(), "obs": self.obs_buf["policy"].detach().cpu().clone(),
                "log": self._log.detach().cpu().clone(), "num_envs": self.num_envs,
                "common_step_counter": int(self.common_step_counter),
                "seed_lo": int(self._native_cfg.seed_lo), "seed_hi": int(self._native_cfg.seed_hi)}

    def load_state_dict(self, sd: dict):
        if int(sd["num_envs"]) != self.num_envs or tuple(sd["state"].shape) != (self.num_envs, _lib.LIFT_STATE_WORDS):
            raise ValueError("checkpoint was taken from an env of a different size")
        self.set_state(sd["state"])
        self.flush_log()                              # nothing of the old trajectory may land in the restored vector later
        self._log.copy_(sd["log"].to(self.device))
        self._obs[self._cur].copy_(sd["obs"].to(self.device))
        self.common_step_counter = int(sd.get("common_step_counter", 0))
        self.seed(int(sd["seed_lo"]) | (int(sd["seed_hi"]) << 32))
        self.obs_buf = {"policy": self._obs[self._cur]}
        return self.obs_buf

    def terms(self, obj_pos, ee_pos, root_state, cmd):
        """The reference's own term functions on caller rows (``rover_lift_terms``): lifted, reach, goal, goal_fine, obj_pos_b."""
        dev = self.device
        t = [torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev) for a in (obj_pos, ee_pos, root_state, cmd)]
        n = int(t[0].shape[0])
        outs = [torch.empty(n, device=dev) for _ in range(4)] + [torch.empty(n, 3, device=dev)]
        _lib.check(self._lib.rover_lift_terms(self._h, n, *[_ptr(x) for x in t], *[_ptr(x) for x in outs], self._stream()),
                   "rover_lift_terms")
        return outs

    def get_state(self) -> torch.Tensor:
        return self.state.t().contiguous()

    def set_state(self, s: torch.Tensor):
        self.state.copy_(s.to(self.device, torch.float32).t())

    def close(self):
        if not self._closed and getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self._lib.rover_lift_destroy(self._h)
            self._h, self._closed = None, True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
