"""``OracleRoverEnv`` -- the RoverEnv surface on CPU tensors, backed by the CPU oracle.  TEST INFRASTRUCTURE.

Lets the reference's own trainer stack (``examples/02_train/train.py`` -> ``SkrlVecEnvWrapper`` ->
``SkrlSequentialLogTrainer.train``, rover_envs/utils/skrl_utils.py:15-41, 96-148) run UNCHANGED in the GPU-less build
container, and records its env-facing call sequence with every action and every returned tensor.  The recorded transcript
(tests/golden/trainer_transcript.npz, written by tools/gen_trainer_transcript.py) is replayed on the MI355X against the HIP
env (tests/test_gpu_trainer_replay.py): the oracle and the HIP path agree bit for bit, so the replay must reproduce it.
"""
from __future__ import annotations

import numpy as np
import torch

from isaac_rover_orbit_amd import _lib
from isaac_rover_orbit_amd.cfg import RoverEnvCfg
from isaac_rover_orbit_amd.envs.rover_env import (LOG_KEYS, RLTaskEnv, _ActionManager, _CommandManager, _ObservationManager,
                                                  _spaces)
from isaac_rover_orbit_amd.terrain import make_flat_terrain, make_procedural_terrain


class OracleRoverEnv(RLTaskEnv):
    def __init__(self, cfg=None, terrain=None, render_mode=None, **kwargs):
        from oracle import rover_oracle as ro
        from helpers import oracle_config_from, oracle_terrain
        if cfg is not None and not isinstance(cfg, RoverEnvCfg):
            from isaac_rover_orbit_amd.compat.convert import from_reference_cfg
            cfg = from_reference_cfg(cfg)
        self.cfg = cfg if cfg is not None else RoverEnvCfg()
        self.ctor_kwargs = dict(kwargs)
        self.ro = ro
        self.device = torch.device("cpu")
        self.num_envs = n = int(self.cfg.scene.num_envs)
        self._native_cfg = self.cfg.to_native()
        self.ocfg = oracle_config_from(ro, self._native_cfg)
        tc = self.cfg.terrain
        if terrain is None:
            terrain = (tc.terrain if tc.kind == "custom" else make_flat_terrain(tc.shape) if tc.kind == "flat" else
                       make_procedural_terrain(tc.shape, seed=tc.seed, sigma_z=tc.sigma_z, n_rocks=tc.n_rocks))
        self.terrain_data = terrain
        if terrain.spawn_locations is None:
            terrain.make_spawns(2 * (self.cfg.global_num_envs or n), seed=tc.spawn_seed)
        self.oter = oracle_terrain(ro, terrain)
        nx, ny = self.cfg.height_scanner.grid
        self.num_rays, self.obs_dim = nx * ny, 4 + nx * ny
        self.max_episode_length = self.cfg.max_episode_length
        self.S = ro.new_state(n)
        self.state = torch.from_numpy(self.S).t()            # (72, n) view: the manager facades index words first
        self._log_np = np.zeros(_lib.LOG_WORDS, np.float32)
        self._log = torch.from_numpy(self._log_np)
        self._log_dict = {k: self._log[i] for i, k in enumerate(LOG_KEYS)}
        self.extras = {"log": self._log_dict, "episode": self._log_dict}
        self.action_manager = _ActionManager(self)
        self.observation_manager = _ObservationManager(self.num_rays)
        self.command_manager = _CommandManager(self)
        self.single_observation_space = _spaces.Dict({"policy": _spaces.Box(-np.inf, np.inf, (self.obs_dim,), np.float32)})
        self.single_action_space = _spaces.Box(-np.inf, np.inf, (2,), np.float32)
        self.observation_space = _spaces.Dict({"policy": _spaces.Box(-np.inf, np.inf, (n, self.obs_dim), np.float32)})
        self.action_space = _spaces.Box(-np.inf, np.inf, (n, 2), np.float32)
        self.common_step_counter = 0
        # ---- the recording
        self.calls = []
        self.rec = {"actions": [], "obs_head": [], "obs_rowsum": [], "reward": [], "terminated": [], "truncated": [], "log": []}
        self.reset_obs = None

    @property
    def unwrapped(self):
        return self

    def _record_obs(self, obs):
        return obs[:, :8].copy(), obs.astype(np.float64).sum(axis=1)

    def reset(self, seed=None, options=None):
        self.calls.append("reset")
        obs = self.ro.reset_all(self.ocfg, self.oter, self.S, env_id_offset=int(self.cfg.env_id_offset))
        self.reset_obs = self._record_obs(obs)
        self.obs_buf = {"policy": torch.from_numpy(obs)}
        return self.obs_buf, self.extras

    def step(self, action):
        self.calls.append("step")
        a = action.detach().cpu().numpy().astype(np.float32)
        assert a.shape == (self.num_envs, 2)
        obs, rew, term, trunc, _force, _ = self.ro.step(self.ocfg, self.oter, self.S, a, env_id_offset=int(self.cfg.env_id_offset),
                                                       log=self._log_np)
        self.common_step_counter += 1
        head, rowsum = self._record_obs(obs)
        for k, v in (("actions", a), ("obs_head", head), ("obs_rowsum", rowsum), ("reward", rew), ("terminated", term),
                     ("truncated", trunc), ("log", self._log_np.copy())):
            self.rec[k].append(v)
        self.obs_buf = {"policy": torch.from_numpy(obs)}
        return (self.obs_buf, torch.from_numpy(rew), torch.from_numpy(term.astype(bool)), torch.from_numpy(trunc.astype(bool)),
                self.extras)

    def close(self):
        self.calls.append("close")

    def transcript(self) -> dict:
        out = {k: np.stack(v) for k, v in self.rec.items()}
        out["reset_obs_head"], out["reset_obs_rowsum"] = self.reset_obs
        out["calls"] = np.array(self.calls)
        out["num_envs"] = np.int64(self.num_envs)
        out["seed"] = np.int64(self.cfg.seed)
        out["terrain"] = np.array([self.cfg.terrain.kind, str(tuple(self.cfg.terrain.shape)), str(self.cfg.terrain.seed),
                                   str(self.cfg.terrain.sigma_z), str(self.cfg.terrain.n_rocks)])
        return out
