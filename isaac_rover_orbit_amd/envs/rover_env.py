"""``RoverEnv`` -- the gymnasium / ORBIT ``RLTaskEnv``-shaped boundary of the MI355X-native rover hot path.

Reproduces the object protocol the reference's consumers use:

* ``rover_envs/envs/navigation/entrypoints/rover_env.py:12-102``  ``RoverEnv.__init__ / step / _reset_idx``
* ``rover_envs/utils/skrl_utils.py:38-41,114-142``                 skrl "isaac-orbit" wrapper + trainer loop
* ``examples/02_train/train.py:123-134``                           ``observation_manager.group_obs_dim`` etc.
* the mdp term signatures (``env.command_manager.get_command``, ``env.action_manager.action``,
  ``env.scene.sensors[...]``, ``env.episode_length_buf`` ...)

``step()`` is one or two HIP kernel launches through the C ABI (``include/rover_hip.h``); nothing is computed in Python or
torch on the hot path and there is no host synchronisation.  All tensors live on the env's GPU.

User-written reward / termination terms (a cfg table entry whose ``func`` is a callable, ``cfg.py``) switch ``step()`` to the
SLOW path: ``rover_step_begin`` (physics + built-in terms) -> the user's ``func(env, **params)`` evaluated in torch on the
facades below, with ORBIT's manager semantics (reward += func * weight * dt with its own episodic sum and
``Episode Reward/<name>`` key; terminations OR-ed into the reset mask BEFORE the reset) -> ``rover_step_finish`` (reset of the
masked envs, command update, observation rows).  Still no host synchronisation.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..cfg import OBS_ORDER, REWARD_ORDER, TERMINATION_ORDER, RoverEnvCfg
from ..terrain import Terrain, make_flat_terrain, make_procedural_terrain
from ._logdict import LogDict

try:  # gymnasium is what the reference uses (robots/aau_rover/__init__.py:3); optional here
    import gymnasium as gym
    from gymnasium import spaces as _spaces
    _GymEnv = gym.Env
except Exception:  # pragma: no cover - exercised only where gymnasium is absent
    gym = None
    _GymEnv = object

    class _Box:
        """Minimal stand-in for ``gymnasium.spaces.Box`` (shape / low / high / dtype only)."""

        def __init__(self, low, high, shape, dtype=np.float32):
            self.low = np.full(shape, low, dtype=dtype)
            self.high = np.full(shape, high, dtype=dtype)
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class _Dict(dict):
        pass

    class _spaces:  # noqa: N801
        Box = _Box
        Dict = _Dict


def _ptr(t: torch.Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


class _ActionManager:
    """``env.action_manager`` facade (ORBIT ActionManager): zero-copy views into the SoA state tensor."""

    def __init__(self, env):
        self._env = env
        self.action_term_dim = [2]                      # train.py:132
        self.total_action_dim = 2
        self.active_terms = ["actions"]

    @property
    def action(self) -> torch.Tensor:
        return self._env.state[_lib.ACTION:_lib.ACTION + 2].t()

    @property
    def prev_action(self) -> torch.Tensor:
        return self._env.state[_lib.PREV_ACTION:_lib.PREV_ACTION + 2].t()


class _ObservationManager:
    def __init__(self, rays: int):
        self.group_obs_dim = {"policy": (4 + rays,)}                       # train.py:131
        self.group_obs_term_dim = {"policy": [(2,), (1,), (1,), (rays,)]}  # get_models.py:39
        self.active_terms = {"policy": list(OBS_ORDER)}


class _CommandManager:
    """``env.command_manager`` facade: ``get_command("target_pose")`` = ``pos_command_b`` (terrain_importer.py:65-68)."""

    def __init__(self, env):
        self._env = env
        self.active_terms = ["target_pose"]

    def get_command(self, name: str) -> torch.Tensor:
        if name != "target_pose":
            raise KeyError(name)
        return self._env.state[_lib.CMD_B:_lib.CMD_B + 3].t()

    @property
    def pos_command_w(self):
        return self._env.state[_lib.TARGET_W:_lib.TARGET_W + 3].t()

    @property
    def heading_command_w(self):
        return self._env.state[_lib.HEADING_CMD_W]

    @property
    def heading_command_b(self):
        return self._env.state[_lib.HEADING_CMD_B]

    @property
    def metrics(self):
        return {"error_pos": self._env.state[_lib.METRIC_POS], "error_heading": self._env.state[_lib.METRIC_HEAD]}


class _SensorData:
    pass


class _ContactSensor:
    """``scene.sensors["contact_sensor"]``: ``data.force_matrix_w`` (N, 13, 1, 3) view of the SoA force buffer."""
    body_names = ["FL_Boogie", "FR_Boogie", "R_Boogie", "FL_Steer", "FR_Steer", "RL_Steer", "RR_Steer", "CL_Drive",
                  "CR_Drive", "FL_Drive", "FR_Drive", "RL_Drive", "RR_Drive"]

    def __init__(self, env):
        self._env = env

    @property
    def data(self):
        env = self._env
        if env._force is None:
            raise RuntimeError("contact forces are not recorded (cfg.record_contact_forces=False)")
        d = _SensorData()
        n = env.num_envs
        d.net_forces_w = env._force.view(_lib.NUM_BODIES, 3, n).permute(2, 0, 1)
        d.force_matrix_w = d.net_forces_w.unsqueeze(2)
        return d


class _RayCaster:
    """``scene.sensors["height_scanner"]``: ``data.pos_w`` and a lazily rebuilt ``data.ray_hits_w`` (slow path)."""

    def __init__(self, env):
        self._env = env

    @property
    def data(self):
        env = self._env
        d = _SensorData()
        pos = env.state[_lib.POS:_lib.POS + 3].t()
        d.pos_w = pos
        # between the two halves of a slow step the observation rows are still the previous step's: user terms get a scan of the
        # CURRENT pose (the reference's RayCaster has updated by the time rewards are computed, rover_env_cfg.py:275-276)
        # ... and with noise / clip / scale on the height_scan term the row no longer holds the raw scan at all
        scan = env._fresh_scan() if (env._in_user_terms or env._obs_post_scan) else env.obs_buf["policy"][:, 4:]
        hit_z = pos[:, 2:3] - scan - env.cfg.height_scanner.height_offset
        q = env.state[_lib.QUAT:_lib.QUAT + 4].t()
        yaw = torch.atan2(2 * (q[:, 0] * q[:, 3] + q[:, 1] * q[:, 2]), 1 - 2 * (q[:, 2] ** 2 + q[:, 3] ** 2))
        nx, ny = env.cfg.height_scanner.grid
        hs = env.cfg.height_scanner
        xs = torch.arange(nx, device=pos.device, dtype=torch.float64) * hs.resolution - 0.5 * hs.size[0]
        ys = torch.arange(ny, device=pos.device, dtype=torch.float64) * hs.resolution - 0.5 * hs.size[1]
        gx, gy = torch.meshgrid(xs.float(), ys.float(), indexing="xy")
        ox, oy = gx.reshape(1, -1), gy.reshape(1, -1)
        c, s = torch.cos(yaw).unsqueeze(1), torch.sin(yaw).unsqueeze(1)
        d.ray_hits_w = torch.stack([pos[:, 0:1] + c * ox - s * oy, pos[:, 1:2] + s * ox + c * oy, hit_z], dim=-1)
        return d


class _TerrainFacade:
    """``scene.terrain``: ``env_origins`` + ``get_spawn_locations()`` (terrain_importer.py:177-185)."""

    def __init__(self, env):
        self._env = env

    @property
    def env_origins(self):
        return self._env.state[_lib.ENV_ORIGIN:_lib.ENV_ORIGIN + 3].t()

    def get_spawn_locations(self):
        return self._env._spawns_dev


class _RobotData:
    """Zero-copy views of the SoA state; the body-frame velocities are rotated on access (ORBIT's ``quat_rotate_inverse``)."""

    def __init__(self, st):
        self.root_pos_w = st[_lib.POS:_lib.POS + 3].t()
        self.root_quat_w = st[_lib.QUAT:_lib.QUAT + 4].t()
        self.root_lin_vel_w = st[_lib.LINVEL:_lib.LINVEL + 3].t()
        self.root_ang_vel_w = st[_lib.ANGVEL:_lib.ANGVEL + 3].t()
        self.joint_pos = st[_lib.BOGIE_Q:_lib.BOGIE_Q + 13].t()
        self.joint_vel = st[_lib.BOGIE_QD:_lib.BOGIE_QD + 13].t()

    def _to_body(self, v):
        q = self.root_quat_w
        w, u = q[:, 0:1], q[:, 1:4]
        return v * (2.0 * w * w - 1.0) - 2.0 * w * torch.cross(u, v, dim=-1) + 2.0 * u * (u * v).sum(-1, keepdim=True)

    @property
    def root_lin_vel_b(self):
        return self._to_body(self.root_lin_vel_w)

    @property
    def root_ang_vel_b(self):
        return self._to_body(self.root_ang_vel_w)


class _Robot:
    """``scene["robot"]`` (ORBIT Articulation): ``data.root_pos_w / root_quat_w / root_lin_vel_w / root_ang_vel_w / root_lin_vel_b /
    root_ang_vel_b / joint_pos / joint_vel`` (joint order: 3 bogies, 4 steer joints, 6 wheels)."""
    joint_names = ["FL_Boogie_Revolute", "FR_Boogie_Revolute", "R_Boogie_Revolute", "FL_Steer_Revolute", "FR_Steer_Revolute",
                   "RL_Steer_Revolute", "RR_Steer_Revolute", "FL_Drive_Continuous", "FR_Drive_Continuous", "CL_Drive_Continuous",
                   "CR_Drive_Continuous", "RL_Drive_Continuous", "RR_Drive_Continuous"]

    def __init__(self, env):
        self._env = env

    @property
    def data(self):
        return _RobotData(self._env.state)


class _Scene:
    def __init__(self, env):
        self.terrain = _TerrainFacade(env)
        self.sensors = {"contact_sensor": _ContactSensor(env), "height_scanner": _RayCaster(env)}
        self.articulations = {"robot": _Robot(env)}
        self.num_envs = env.num_envs

    def __getitem__(self, name):
        if name in self.sensors:
            return self.sensors[name]
        if name in self.articulations:
            return self.articulations[name]
        raise KeyError(name)


class RLTaskEnv(_GymEnv):
    """Name the skrl shim checks with ``isinstance(env.unwrapped, RLTaskEnv)`` (skrl_utils.py:38)."""
    metadata = {"render_modes": [None]}


LOG_KEYS = ([f"Episode Reward/{k}" for k in REWARD_ORDER] + [f"Episode Termination/{k}" for k in TERMINATION_ORDER]
            + ["Metrics/target_pose/error_pos", "Metrics/target_pose/error_heading"])


class RoverEnv(RLTaskEnv):
    """MI355X-native ``AAURoverEnv-v0``.  ``RoverEnv(cfg)`` / ``gym.make("AAURoverEnv-v0", cfg=cfg)``."""

    def __init__(self, cfg: RoverEnvCfg | None = None, terrain: Terrain | None = None, render_mode=None, **kwargs):
        if cfg is not None and not isinstance(cfg, RoverEnvCfg):
            # a reference-style cfg (the reference's AAURoverEnvCfg on ORBIT / compat configclasses): translate it
            from ..compat.convert import from_reference_cfg, is_reference_cfg
            if not is_reference_cfg(cfg):
                raise TypeError(f"unsupported cfg type {type(cfg)}")
            cfg = from_reference_cfg(cfg)
        self.cfg = cfg if cfg is not None else RoverEnvCfg()
        self.render_mode = render_mode
        self.cfg.validate()
        if not torch.cuda.is_available():
            raise _lib.RoverHipError("RoverEnv needs a ROCm GPU: the hot path is HIP-only (no CPU fallback)")
        self.device = torch.device(self.cfg.sim.device)
        if self.device.type != "cuda":
            raise _lib.RoverHipError(f"cfg.sim.device must be a cuda (ROCm) device, got {self.device}")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.num_envs = int(self.cfg.scene.num_envs)
        self._lib = _lib.load()
        self._native_cfg = self.cfg.to_native()
        nx, ny = self.cfg.height_scanner.grid
        self.num_rays = nx * ny
        self.obs_dim = 4 + self.num_rays
        self.max_episode_length = self.cfg.max_episode_length
        self.max_episode_length_s = self.cfg.episode_length_s
        self.physics_dt = self.cfg.sim.dt
        self.step_dt = self.cfg.sim.dt * self.cfg.decimation
        self.common_step_counter = 0

        # ---- shared terrain data (replicated per GPU)
        tc = self.cfg.terrain
        if terrain is None:
            if tc.kind == "custom":
                terrain = tc.terrain
            elif tc.kind == "flat":
                terrain = make_flat_terrain(tc.shape)
            elif tc.kind == "procedural":
                terrain = make_procedural_terrain(tc.shape, seed=tc.seed, sigma_z=tc.sigma_z, n_rocks=tc.n_rocks)
            else:
                raise ValueError(f"unknown terrain kind {tc.kind}")
        self.terrain_data = terrain
        n_global = self.cfg.global_num_envs or self.num_envs
        if terrain.spawn_locations is None:
            terrain.make_spawns(2 * n_global, seed=tc.spawn_seed)   # terrain_utils.py:123-124: n_spawns = 2 * num_envs
        if self.cfg.spawn_draw == "distinct" and len(terrain.spawn_locations) < n_global:
            # the affine row bijection (a * gid + b) mod n_spawns only separates envs while gid < n_spawns
            # (randomizations.py:22 draws a randperm prefix, which needs len(table) >= number of envs as well)
            raise ValueError(f"spawn_draw='distinct' needs a spawn table with at least global_num_envs = {n_global} rows "
                             f"(got {len(terrain.spawn_locations)}); use spawn_draw='independent' or a larger table")
        with torch.cuda.device(self.device):
            dev = self.device
            self._height_dev = torch.from_numpy(terrain.height).to(dev)
            self._obstacle_dev = torch.from_numpy(terrain.obstacle).to(dev)
            self._mask_dev = torch.from_numpy(np.ascontiguousarray(terrain.safe_rock_mask, dtype=np.uint8)).to(dev)
            self._spawns_dev = torch.from_numpy(np.ascontiguousarray(terrain.spawn_locations, dtype=np.float32)).to(dev)

            # ---- native handle + caller-owned buffers
            h = C.c_void_p()
            _lib.check(self._lib.rover_create(C.byref(self._native_cfg), self.num_envs, int(self.cfg.env_id_offset),
                                              self._dev_index, C.byref(h)), "rover_create")
            self._h = h
            H, W = terrain.shape
            _lib.check(self._lib.rover_set_terrain(h, _ptr(self._height_dev), _ptr(self._obstacle_dev), _ptr(self._mask_dev),
                                                   H, W, float(terrain.resolution), float(terrain.min_x),
                                                   float(terrain.min_y), _ptr(self._spawns_dev),
                                                   int(self._spawns_dev.shape[0])), "rover_set_terrain")
            self._lookup_dev = None
            if terrain.lookup_height is not None:   # mesh-ingested terrain: the reference's look-up heightmap next to the surface
                self._lookup_dev = torch.from_numpy(terrain.lookup_height).to(dev)
                _lib.check(self._lib.rover_set_terrain_lookup(h, _ptr(self._lookup_dev)), "rover_set_terrain_lookup")
            self._height_q_dev = None
            q16 = terrain.height_q16() if getattr(self.cfg, "use_int16_terrain", True) else None
            if q16 is not None:     # exact 16-bit copy for the ray-caster kernel (half the staged bytes, same results)
                self._height_q_dev = torch.from_numpy(q16[0]).to(dev)
                _lib.check(self._lib.rover_set_terrain_q16(h, _ptr(self._height_q_dev), q16[1]), "rover_set_terrain_q16")
            n = self.num_envs
            self.state = torch.zeros(_lib.STATE_WORDS, n, dtype=torch.float32, device=dev)
            self.state[_lib.QUAT] = 1.0
            ws = int(self._lib.rover_workspace_bytes(h))
            self._workspace = torch.zeros(max(ws, 4) // 4, dtype=torch.float32, device=dev)
            _lib.check(self._lib.rover_bind(h, _ptr(self.state), _ptr(self._workspace), ws), "rover_bind")
            # two rotating output sets: the tensors returned by step k stay valid during step k + 1 (skrl keeps
            # `states` while it asks for `next_states`, skrl_utils.py:121-135)
            self._nbuf = 2
            self._obs = [torch.zeros(n, self.obs_dim, dtype=torch.float32, device=dev) for _ in range(self._nbuf)]
            self._rew = [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(self._nbuf)]
            self._term = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(self._nbuf)]
            self._trunc = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(self._nbuf)]
            self._force = (torch.zeros(_lib.NUM_BODIES * 3, n, dtype=torch.float32, device=dev)
                           if self.cfg.record_contact_forces else None)
            self._log = torch.zeros(_lib.LOG_WORDS, dtype=torch.float32, device=dev)
        self._cur = 0
        self.episode_length_buf = self.state[_lib.EP_LEN].view(torch.int32)
        self.obs_buf = {"policy": self._obs[0]}
        self.reward_buf = self._rew[0]
        self.reset_terminated = self._term[0].view(torch.bool)
        self.reset_time_outs = self._trunc[0].view(torch.bool)
        # extras["log"]: 0-d views into the device log vector, refreshed by the kernels (no host sync).  "on_demand" (default): the
        # reduction behind it runs when the dictionary is READ -- the same numbers, and rover_step may then be ONE kernel launch
        self._log_pending = False
        self._log_deferred = getattr(self.cfg, "log_reduction", "on_demand") == "on_demand"
        if getattr(self.cfg, "log_reduction", "on_demand") not in ("on_demand", "every_step"):
            raise ValueError("log_reduction must be 'on_demand' or 'every_step'")
        _lib.check(self._lib.rover_set_log_deferred(self._h, int(self._log_deferred)), "rover_set_log_deferred")
        _lib.check(self._lib.rover_set_obs_streaming(self._h, int(bool(getattr(self.cfg, "stream_observations", False)))), "rover_set_obs_streaming")
        log_items = {k: self._log[i] for i, k in enumerate(LOG_KEYS)}
        self._log_slots = {k: i for i, k in enumerate(LOG_KEYS)}       # key -> word of the host mirror (set_log_values)
        # ---- user-written terms (cfg.py): evaluated in torch between the two halves of the step
        self._user_rewards = list(self.cfg.custom_terms(self.cfg.rewards, REWARD_ORDER).items())
        self._user_terminations = list(self.cfg.custom_terms(self.cfg.terminations, TERMINATION_ORDER).items())
        self._slow_path = bool(self._user_rewards or self._user_terminations)
        # ORBIT's observation post-processing (noise -> clip -> scale, ObservationManager.compute_group) for the terms that ask for
        # it (cfg.observation_post): the kernels write those terms raw, the columns are finished in torch behind every step / reset
        cols = {"actions": slice(0, 2), "distance": slice(2, 3), "heading": slice(3, 4), "height_scan": slice(4, 4 + self.num_rays)}
        self._obs_post = [(cols[name], t) for name, t in self.cfg.observation_post().items()]
        self._obs_post_scan = any(sl.start == 4 for sl, _ in self._obs_post)
        self._obs_gen = torch.Generator(device=self.device).manual_seed(int(self.cfg.seed))
        self._in_user_terms = False
        self._scan_cache = None
        if self._slow_path:
            with torch.cuda.device(self.device):
                if self._force is None:    # the second half's built-in collision term reads the force rows
                    self._force = torch.zeros(_lib.NUM_BODIES * 3, n, dtype=torch.float32, device=dev)
                self._user_sums = {name: torch.zeros(n, dtype=torch.float32, device=dev) for name, _ in self._user_rewards}
                self._user_log = torch.zeros(len(self._user_rewards) + len(self._user_terminations), dtype=torch.float32, device=dev)
                self._reset_mask = torch.zeros(n, dtype=torch.uint8, device=dev)
            for i, (name, _) in enumerate(self._user_rewards):
                log_items[f"Episode Reward/{name}"] = self._user_log[i]
                self._log_slots[f"Episode Reward/{name}"] = self._log.numel() + i
            for i, (name, _) in enumerate(self._user_terminations):
                log_items[f"Episode Termination/{name}"] = self._user_log[len(self._user_rewards) + i]
                self._log_slots[f"Episode Termination/{name}"] = self._log.numel() + len(self._user_rewards) + i
        self._log_items_device = dict(log_items)
        self._log_host = None
        self._log_host_stale = True
        self._log_dict = LogDict(self, log_items)
        if getattr(self.cfg, "log_values", "device") != "device":
            self.set_log_values(self.cfg.log_values)
        self.extras = {"log": self._log_dict, "episode": self._log_dict}   # rover_env.py:39

        # ---- manager / scene facades + spaces
        self.action_manager = _ActionManager(self)
        self.observation_manager = _ObservationManager(self.num_rays)
        self.command_manager = _CommandManager(self)
        self.scene = _Scene(self)
        # ORBIT exposes batched spaces (01_zero_agent.py:44-52 builds zeros(env.action_space.shape))
        self.single_observation_space = _spaces.Dict({"policy": _spaces.Box(-np.inf, np.inf, (self.obs_dim,), np.float32)})
        self.single_action_space = _spaces.Box(-np.inf, np.inf, (2,), np.float32)
        self.observation_space = _spaces.Dict({"policy": _spaces.Box(-np.inf, np.inf, (n, self.obs_dim), np.float32)})
        self.action_space = _spaces.Box(-np.inf, np.inf, (n, 2), np.float32)
        self._step_args = None
        self._closed = False
        if getattr(self.cfg, "roctx_markers", False):
            self.set_markers(True)

    # ------------------------------------------------------------------------------------------------------------
    @property
    def unwrapped(self):
        return self

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def call_counter(self) -> int:
        """Number of reset() / step() launches so far: keys the per-batch spawn permutation (cfg.spawn_draw="distinct")."""
        c = C.c_uint64(0)
        _lib.check(self._lib.rover_get_counter(self._h, C.byref(c)), "rover_get_counter")
        return int(c.value)

    def set_call_counter(self, counter: int):
        """Restore the call counter (checkpoint resume / replaying a rollout from a given point)."""
        _lib.check(self._lib.rover_set_counter(self._h, int(counter)), "rover_set_counter")
        self._sync_counter()

    def _bump_counter(self):
        # mirror of the handle's call counter in the config struct (one launch = one count): a config derived from
        # `_native_cfg` at any time (tests: the oracle's) continues in step with the handle
        cfg = self._native_cfg
        lo = cfg.counter_lo + 1
        if lo > 0xFFFFFFFF:
            lo, cfg.counter_hi = 0, cfg.counter_hi + 1
        cfg.counter_lo = lo

    def _sync_counter(self):
        # the struct mirror tracks the handle, so that a config derived from it (tests: the oracle's) continues in step
        c = self.call_counter
        self._native_cfg.counter_lo, self._native_cfg.counter_hi = c & 0xFFFFFFFF, (c >> 32) & 0xFFFFFFFF

    def seed(self, seed: int = -1) -> int:
        """gymnasium / ORBIT ``env.seed``: re-keys the counter-based RNG of every reset that follows (``rover_set_seed``).
        A negative seed keeps the current key (ORBIT draws a random one there; determinism is preferred here)."""
        seed = int(seed)
        if seed >= 0:
            self.cfg.seed = seed
            lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
            self._native_cfg.seed_lo, self._native_cfg.seed_hi = lo, hi
            _lib.check(self._lib.rover_set_seed(self._h, lo, hi), "rover_set_seed")
            self._obs_gen.manual_seed(seed)      # the observation-noise draws of plain (non-ORBIT) noise objects
        return int(self.cfg.seed)

    def reset(self, seed: int | None = None, options=None):
        """ORBIT ``RLTaskEnv.reset``: (re-seed,) reset every env, return the first observation."""
        if seed is not None:
            self.seed(seed)
        self.flush_log()     # a pending on-demand reduction belongs to the step before this reset (the reset advances the launch tag)
        if self._slow_path:  # RewardManager.reset of every env: the user terms' episodic sums start over
            for sums in self._user_sums.values():
                sums.zero_()
        obs = self._obs[self._cur]
        _lib.check(self._lib.rover_reset(self._h, _ptr(obs), self._stream()), "rover_reset")
        self._bump_counter()
        if self._obs_post:
            self._post_observations(obs)
        self.obs_buf = {"policy": obs}
        return self.obs_buf, self.extras

    def reset_with_draws(self, mask, spawn_row, yaw_u, theta_u, heading_u):
        """``_reset_idx`` of the masked envs with the reference's recorded torch draws injected in place of the Philox
        draws (``rover_reset_with_draws``; parity protocol for reset outcomes, tests/golden/reset.npz)."""
        dev, n = self.device, self.num_envs
        rows = np.asarray(spawn_row)
        if rows.size and (rows.min() < 0 or rows.max() >= int(self._spawns_dev.shape[0])):
            raise ValueError(f"spawn_row must index the spawn table (0 <= row < {int(self._spawns_dev.shape[0])})")
        mask_d = None if mask is None else torch.as_tensor(np.ascontiguousarray(mask, dtype=np.uint8), device=dev)
        row_d = torch.as_tensor(np.ascontiguousarray(spawn_row, dtype=np.int32), device=dev)
        yaw_d = torch.as_tensor(np.ascontiguousarray(yaw_u, dtype=np.float32), device=dev)
        th_d = torch.as_tensor(np.ascontiguousarray(theta_u, dtype=np.float32), device=dev)
        hd_d = torch.as_tensor(np.ascontiguousarray(heading_u, dtype=np.float32), device=dev)
        if row_d.shape != (n,) or yaw_d.shape != (n,) or hd_d.shape != (n,) or th_d.shape != (n, self._native_cfg.max_target_tries):
            raise ValueError("draw arrays must have one entry (theta: max_target_tries entries) per env")
        self.flush_log()
        obs = self._obs[self._cur]
        _lib.check(self._lib.rover_reset_with_draws(self._h, _ptr(mask_d), _ptr(row_d), _ptr(yaw_d), _ptr(th_d), _ptr(hd_d),
                                                    _ptr(obs), self._stream()), "rover_reset_with_draws")
        torch.cuda.current_stream(dev).synchronize()      # the draw tensors are temporaries
        if self._obs_post:
            self._post_observations(obs)
        self.obs_buf = {"policy": obs}
        return self.obs_buf, self.extras

    def _check_action(self, action: torch.Tensor) -> torch.Tensor:
        if action.dtype != torch.float32 or not action.is_contiguous() or action.device != self.device:
            action = action.to(device=self.device, dtype=torch.float32).contiguous()
        if action.shape != (self.num_envs, 2):
            raise ValueError(f"action must have shape ({self.num_envs}, 2), got {tuple(action.shape)}")
        return action

    def step(self, action: torch.Tensor):
        """``RoverEnv.step`` (rover_env.py:42-102): two asynchronous kernel launches, no host sync."""
        action = self._check_action(action)
        if self._slow_path:
            return self._step_with_user_terms(action)
        self._cur = (self._cur + 1) % self._nbuf
        k = self._cur
        rc = self._lib.rover_step(self._h, C.c_void_p(action.data_ptr()), self._obs_ptr[k], self._rew_ptr[k],
                                  self._term_ptr[k], self._trunc_ptr[k], self._force_ptr, self._log_ptr,
                                  self._stream())
        if rc != 0:
            _lib.check(rc, "rover_step")
        self._bump_counter()
        self.common_step_counter += 1
        self._log_pending = self._log_deferred
        self._log_host_stale = True
        self.obs_buf = self._obs_dicts[k]
        if self._obs_post:
            self._post_observations(self.obs_buf["policy"])
        self.reward_buf = self._rew[k]
        self.reset_terminated = self._term_b[k]
        self.reset_time_outs = self._trunc_b[k]
        return self.obs_buf, self.reward_buf, self.reset_terminated, self.reset_time_outs, self.extras

    def _fresh_scan(self) -> torch.Tensor:
        """Height scan of the pose the physics left (slow path, once per step, only if a user term asks for the ray hits)."""
        if self._scan_cache is None:
            self._scan_cache = self.height_scan()
        return self._scan_cache

    def _step_with_user_terms(self, action: torch.Tensor):
        """The slow path: ``RoverEnv.step`` of the reference (rover_env.py:62-99) with the user's terms in ORBIT's order.

        rover_step_begin  :62-86   action, 6 physics steps, counters, built-in terminations + rewards (+ episodic sums)
        torch             :82-86   user terminations (OR-ed into terminated / time-outs), user rewards (+= func * weight * dt, own
                                   episodic sums); both see the state the physics left, the STALE command (B-13) and the incremented
                                   episode counter (B-14) -- what ORBIT's managers hand to a term function at that point
        torch             :27-39   the user terms' share of _reset_idx's log: mean episodic sum of the envs about to reset /
                                   max_episode_length_s, termination counts; values persist while no env resets
        rover_step_finish :89-99   built-in share of the log, reset of the masked envs, command update, observation rows"""
        self._cur = (self._cur + 1) % self._nbuf
        k = self._cur
        st = self._stream()
        _lib.check(self._lib.rover_step_begin(self._h, C.c_void_p(action.data_ptr()), self._rew_ptr[k], self._term_ptr[k],
                                              self._trunc_ptr[k], self._force_ptr, st), "rover_step_begin")
        self._bump_counter()
        self.common_step_counter += 1
        self.reward_buf = self._rew[k]
        self.reset_terminated = self._term_b[k]
        self.reset_time_outs = self._trunc_b[k]
        self._in_user_terms, self._scan_cache = True, None
        try:
            dones = []
            for name, t in self._user_terminations:
                v = t.func(self, **t.params).to(torch.bool).reshape(self.num_envs)
                dones.append(v)
                (self.reset_time_outs if t.time_out else self.reset_terminated).logical_or_(v)
            for name, t in self._user_rewards:
                if t.weight == 0.0:      # ORBIT's RewardManager skips zero-weight terms
                    continue
                val = t.func(self, **t.params).to(torch.float32).reshape(self.num_envs) * (float(t.weight) * self.step_dt)
                self.reward_buf += val
                self._user_sums[name] += val
        finally:
            self._in_user_terms, self._scan_cache = False, None
        mask_b = torch.logical_or(self.reset_terminated, self.reset_time_outs)
        self._reset_mask.copy_(mask_b)
        cnt = mask_b.sum()
        any_reset = cnt > 0
        nr = len(self._user_rewards)
        for i, (name, _) in enumerate(self._user_rewards):
            s = self._user_sums[name]
            new = (s * mask_b).sum() / cnt.clamp(min=1) / self.max_episode_length_s
            self._user_log[i] = torch.where(any_reset, new, self._user_log[i])
            s.masked_fill_(mask_b, 0.0)
        for i, v in enumerate(dones):
            self._user_log[nr + i] = torch.where(any_reset, torch.logical_and(v, mask_b).sum().to(torch.float32), self._user_log[nr + i])
        _lib.check(self._lib.rover_step_finish(self._h, _ptr(self._reset_mask), self._obs_ptr[k], self._force_ptr, self._log_ptr, st),
                   "rover_step_finish")
        self._log_pending = False        # the second half reduces the built-in log eagerly
        self._log_host_stale = True
        self.obs_buf = self._obs_dicts[k]
        if self._obs_post:
            self._post_observations(self.obs_buf["policy"])
        return self.obs_buf, self.reward_buf, self.reset_terminated, self.reset_time_outs, self.extras

    def _post_observations(self, obs: torch.Tensor):
        """noise, clip, scale of the observation terms in ``cfg.observation_post()`` on the finished row, in ORBIT's order."""
        self._scan_cache = None          # the raw scan of the new pose, should a consumer of ray_hits_w ask for it
        for sl, t in self._obs_post:
            x = obs[:, sl]
            nz = t.noise
            if nz is not None:
                if callable(getattr(nz, "func", None)):          # ORBIT NoiseCfg: noise.func(data, cfg) (torch's global generator)
                    x = nz.func(x, nz)
                elif hasattr(nz, "n_min"):                       # plain objects: this env's own generator (seeded by cfg.seed)
                    x = x + torch.rand(x.shape, device=x.device, generator=self._obs_gen) * (nz.n_max - nz.n_min) + nz.n_min
                elif hasattr(nz, "std"):
                    x = x + getattr(nz, "mean", 0.0) + nz.std * torch.randn(x.shape, device=x.device, generator=self._obs_gen)
                elif hasattr(nz, "bias"):
                    x = x + nz.bias
                else:
                    raise TypeError(f"unsupported observation noise {nz!r}")
            if t.clip is not None:
                x = x.clip(min=float(t.clip[0]), max=float(t.clip[1]))
            if float(t.scale) != 1.0:
                x = x * float(t.scale)
            obs[:, sl] = x

    def flush_log(self):
        """Bring ``extras["log"]`` / ``episode_log_vector`` up to date (``rover_flush_log``).  The log dictionary calls it on
        every read; only code that kept one of its tensors from an earlier step needs to call it itself."""
        if self._log_pending:
            self._log_pending = False
            _lib.check(self._lib.rover_flush_log(self._h, self._log_ptr, self._stream()), "rover_flush_log")
        if self._log_host is not None and self._log_host_stale:      # log_values = "host": ONE copy + ONE synchronisation per step
            self._log_host_stale = False
            nb = self._log.numel()
            self._log_host[:nb].copy_(self._log, non_blocking=True)
            if self._log_host.numel() > nb:
                self._log_host[nb:].copy_(self._user_log, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()

    def set_log_values(self, where: str = "device"):
        """Where the 0-d tensors of ``extras["log"]`` / ``extras["episode"]`` live.  ``"device"`` (ORBIT's way): views of the device
        log vector -- a consumer that calls ``.item()`` on each of them, as the reference's trainer does after every step
        (skrl_utils.py:139-142), synchronises once per entry.  ``"host"``: views of a pinned host mirror that the first read after a
        step refreshes with one copy and one synchronisation; the ``.item()`` calls are then free.  Same numbers either way."""
        if where not in ("device", "host"):
            raise ValueError("log_values must be 'device' or 'host'")
        self.flush_log()
        if where == "device":
            self._log_host = None
            self._log_dict._rebind(self._log_items_device)
            return
        nb = self._log.numel()
        nu = self._user_log.numel() if self._slow_path else 0
        self._log_host = torch.zeros(nb + nu, dtype=torch.float32).pin_memory()
        self._log_host_stale = True
        items = {key: self._log_host[self._log_slots[key]] for key in self._log_items_device}
        self._log_dict._rebind(items)

    def profile_step(self, action: torch.Tensor):
        """``step`` (same validation and bookkeeping) with HIP-event timing of the two kernels; returns
        ``(ms_step_kernel, ms_scan_kernel)``, event overhead included (see ``profile_event_overhead``).  Syncs."""
        action = self._check_action(action)
        self._cur = (self._cur + 1) % self._nbuf
        k = self._cur
        a, b = C.c_float(0.0), C.c_float(0.0)
        _lib.check(self._lib.rover_profile_step(self._h, C.c_void_p(action.data_ptr()), self._obs_ptr[k], self._rew_ptr[k],
                                                self._term_ptr[k], self._trunc_ptr[k], self._force_ptr, self._log_ptr,
                                                self._stream(), C.byref(a), C.byref(b)), "rover_profile_step")
        self._bump_counter()
        self.common_step_counter += 1
        self._log_pending = self._log_deferred
        self._log_host_stale = True
        self.obs_buf = self._obs_dicts[k]
        if self._obs_post:
            self._post_observations(self.obs_buf["policy"])
        self.reward_buf = self._rew[k]
        self.reset_terminated = self._term_b[k]
        self.reset_time_outs = self._trunc_b[k]
        return a.value, b.value

    def profile_event_overhead(self, reps: int = 100) -> float:
        """Milliseconds an empty HIP-event pair reports on the launch stream: subtract from ``profile_step`` figures."""
        ms = C.c_float(0.0)
        _lib.check(self._lib.rover_profile_event_overhead(self._h, self._stream(), int(reps), C.byref(ms)),
                   "rover_profile_event_overhead")
        return ms.value

    def kernel_names(self) -> tuple[str, str]:
        """Names of the two kernels ``step()`` launches, as rocprofv3's kernel trace prints them (``rover_kernel_names``)."""
        a, b = C.create_string_buffer(128), C.create_string_buffer(128)
        _lib.check(self._lib.rover_kernel_names(self._h, a, b, 128), "rover_kernel_names")
        return a.value.decode(), b.value.decode()

    def set_markers(self, enabled: bool = True):
        """roctx ranges around the launches of every ``step()`` (``rocprofv3 --marker-trace``); ``cfg.roctx_markers``."""
        _lib.check(self._lib.rover_set_markers(self._h, int(bool(enabled))), "rover_set_markers")

    def __getattr__(self, name):
        # lazily built pointer caches (kept out of __init__ so that tensors can be swapped in tests)
        if name in ("_obs_ptr", "_rew_ptr", "_term_ptr", "_trunc_ptr", "_force_ptr", "_log_ptr", "_obs_dicts", "_term_b",
                    "_trunc_b"):
            self._obs_ptr = [_ptr(t) for t in self._obs]
            self._rew_ptr = [_ptr(t) for t in self._rew]
            self._term_ptr = [_ptr(t) for t in self._term]
            self._trunc_ptr = [_ptr(t) for t in self._trunc]
            self._force_ptr = _ptr(self._force)
            self._log_ptr = _ptr(self._log)
            self._obs_dicts = [{"policy": t} for t in self._obs]
            self._term_b = [t.view(torch.bool) for t in self._term]
            self._trunc_b = [t.view(torch.bool) for t in self._trunc]
            return self.__dict__[name]
        raise AttributeError(name)

    # ---- unit entry points used by the parity tests ------------------------------------------------------------
    def ackermann(self, raw: torch.Tensor):
        raw = raw.to(self.device, torch.float32).contiguous()
        n = raw.shape[0]
        processed = torch.empty(n, 2, device=self.device)
        steer = torch.empty(n, 4, device=self.device)
        wheel = torch.empty(n, 6, device=self.device)
        _lib.check(self._lib.rover_ackermann(self._h, n, _ptr(raw), _ptr(processed), _ptr(steer), _ptr(wheel),
                                             self._stream()), "rover_ackermann")
        return processed, steer, wheel

    def mdp_terms(self, cmd_b, action, prev_action, ep_len, force):
        """The step kernel's term functions on caller rows (``rover_mdp_terms``): returns ``(obs_distance, obs_angle,
        rew (n, 7) unweighted, term (n, 4) bool [time_out, is_success, far_from_target, collision])``."""
        dev = self.device
        cmd_b = torch.as_tensor(cmd_b, dtype=torch.float32, device=dev).contiguous()
        n = int(cmd_b.shape[0])
        action = torch.as_tensor(action, dtype=torch.float32, device=dev).contiguous()
        prev_action = torch.as_tensor(prev_action, dtype=torch.float32, device=dev).contiguous()
        ep_len = torch.as_tensor(ep_len, device=dev).to(torch.int32).contiguous()
        force = torch.as_tensor(force, dtype=torch.float32, device=dev).reshape(n, _lib.NUM_BODIES * 3).contiguous()
        od, oa = torch.empty(n, device=dev), torch.empty(n, device=dev)
        rew = torch.empty(n, _lib.NUM_REW, device=dev)
        term = torch.empty(n, _lib.NUM_TERM, dtype=torch.uint8, device=dev)
        _lib.check(self._lib.rover_mdp_terms(self._h, n, _ptr(cmd_b), _ptr(action), _ptr(prev_action), _ptr(ep_len), _ptr(force),
                                             _ptr(od), _ptr(oa), _ptr(rew), _ptr(term), self._stream()), "rover_mdp_terms")
        return od, oa, rew, term.bool()

    def height_scan(self) -> torch.Tensor:
        scan = torch.empty(self.num_envs, self.num_rays, device=self.device)
        _lib.check(self._lib.rover_height_scan(self._h, _ptr(scan), self._stream()), "rover_height_scan")
        return scan

    def physics(self, steer_target: torch.Tensor, wheel_target: torch.Tensor, substeps: int = 1):
        st = steer_target.to(self.device, torch.float32).contiguous()
        wt = wheel_target.to(self.device, torch.float32).contiguous()
        force = torch.zeros(_lib.NUM_BODIES * 3, self.num_envs, device=self.device)
        _lib.check(self._lib.rover_physics(self._h, _ptr(st), _ptr(wt), int(substeps), _ptr(force), self._stream()),
                   "rover_physics")
        return force.view(_lib.NUM_BODIES, 3, self.num_envs).permute(2, 0, 1)

    @property
    def episode_log_vector(self) -> torch.Tensor:
        """The raw 16-float device vector behind ``extras["log"]``: [0:7] mean episodic reward sums per term, [7:11]
        number of envs that ended this step by (time_out, is_success, far_from_target, collision), [11:13] mean metrics,
        [13] number of envs that were reset in this step.  Entries other than [13] keep their last value while no env
        resets (ORBIT only refreshes ``extras["log"]`` on resets).  Reading it runs the pending reduction first."""
        self.flush_log()
        return self._log

    # ---- state access (env state is never checkpointed in the reference; here it is just a tensor) -------------
    def get_state(self) -> torch.Tensor:
        """(num_envs, 72) copy of the per-env state words (AoS, same word order as include/rover_hip.h)."""
        self._sync_counter()
        return self.state.t().contiguous()

    def set_state(self, state_aos: torch.Tensor):
        self.state.copy_(state_aos.to(self.device, torch.float32).t())

    # ---- checkpoint / resume: the state words ARE the environment (the RNG is a counter-based Philox keyed by the seed,
    # the global env id and the per-env reset counter, which is a state word), so resuming is exact
    def state_dict(self) -> dict:
        """Everything needed to continue bit-for-bit: state words, the last observation, the episodic log vector."""
        self.flush_log()
        return {"state": self.get_state().cpu(), "obs": self.obs_buf["policy"].detach().cpu().clone(),
                "log": self._log.detach().cpu().clone(), "num_envs": self.num_envs,
                "common_step_counter": int(self.common_step_counter), "call_counter": self.call_counter,
                # the Philox key: an env re-seeded with seed() / reset(seed=) must be restored with ITS key
                "seed_lo": int(self._native_cfg.seed_lo), "seed_hi": int(self._native_cfg.seed_hi)}

    def load_state_dict(self, sd: dict):
        """Restore a ``state_dict()`` of an env of the same size and configuration; returns the observation dict."""
        if int(sd["num_envs"]) != self.num_envs or tuple(sd["state"].shape) != (self.num_envs, _lib.STATE_WORDS):
            raise ValueError("checkpoint was taken from an env of a different size")
        self.set_state(sd["state"])
        self.flush_log()                                       # nothing of the old trajectory may land in the restored vector later
        self._log.copy_(sd["log"].to(self._log.device))
        self._obs[self._cur].copy_(sd["obs"].to(self.device))
        self.common_step_counter = int(sd.get("common_step_counter", 0))
        if "seed_lo" in sd:
            self.seed(int(sd["seed_lo"]) | (int(sd["seed_hi"]) << 32))
        _lib.check(self._lib.rover_set_counter(self._h, int(sd.get("call_counter", 0))), "rover_set_counter")
        self._sync_counter()                                   # the struct mirror follows the handle
        self.obs_buf = self._obs_dicts[self._cur]
        return self.obs_buf

    def render(self):
        return None

    def close(self):
        if not self._closed and getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self._lib.rover_destroy(self._h)
            self._h = None
            self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
