"""Declarative configuration of ``AAURoverEnv-v0`` in the *shape* of the reference's ORBIT ``@configclass`` tables
(term name -> func / weight / params), so that user cfg edits carry over:

* ``RoverEnvCfg``          <- ``rover_envs/envs/navigation/rover_env_cfg.py:228-278``
* ``AckermannActionCfg``   <- ``rover_envs/mdp/actions/actions_cfg.py:9-51``
* ``AAURoverEnvCfg``       <- ``rover_envs/envs/navigation/robots/aau_rover/env_cfg.py:10-31``

The built-in term *functions* are fused into the HIP kernels; the cfg carries their names (strings), weights, scales and
thresholds, which are compiled into the kernel parameter block (``_lib.RoverConfig``).

**User-written terms** (the normal way to use the reference: ``rover_env_cfg.py:126-183`` are tables of arbitrary ``func=``):
a reward or termination entry whose ``func`` is a *callable* -- signature ``func(env, **params) -> (num_envs,) tensor``, as in
``mdp/rewards.py:14-137`` / ``mdp/terminations.py:14-64`` -- is kept and evaluated in torch by ``RoverEnv.step`` on the env's
manager / scene facades between the two halves of the step (``rover_step_begin`` / ``rover_step_finish``: the SLOW path, three
launches + the torch ops of the terms).  The built-in entries must stay in the table (a built-in reward is switched off with
``weight=0``); with the stock table the env takes the one-launch fast path.  The observation TERMS are fixed (they define the row
layout the policy checkpoint expects); their ORBIT post-processing -- ``noise`` (``Unoise(n_min, n_max)``, imported for that by
``rover_env_cfg.py:23``), ``clip``, and a ``scale`` on the terms the kernels do not scale -- is applied in torch on the finished
row, in ORBIT's order (noise, clip, scale), for the terms that ask for it (``RoverEnvCfg.observation_post``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any

from . import _lib


@dataclass
class TermCfg:
    func: Any            # the name of a built-in term function (str) or a user callable func(env, **params)
    weight: float = 1.0
    scale: float = 1.0
    params: dict = field(default_factory=dict)
    time_out: bool = False
    noise: Any = None    # observation terms: ORBIT NoiseCfg (``noise.func(data, noise)``) or an object with n_min / n_max, mean / std, bias
    clip: Any = None     # observation terms: (min, max)


@dataclass
class SceneCfg:
    num_envs: int = 256          # rover_env_cfg.py:233-234
    env_spacing: float = 4.0
    replicate_physics: bool = False


@dataclass
class SimCfg:
    dt: float = 1.0 / 30.0       # rover_env_cfg.py:269
    device: str = "cuda:0"
    use_gpu_pipeline: bool = True


@dataclass
class RayCasterCfg:
    """``height_scanner`` (rover_env_cfg.py:78-86)."""
    resolution: float = 0.1
    size: tuple = (3.0, 3.0)
    offset_z: float = 10.0
    attach_yaw_only: bool = True
    max_distance: float = 100.0
    height_offset: float = 0.26878   # observations.py:45
    # what the vertical rays hit: "triangles" = the triangle mesh of the heightfield (each cell split along its
    # (i, j) - (i+1, j+1) diagonal; what ORBIT's Warp mesh ray-cast of that terrain returns), "bilinear" = smooth patch
    surface: str = "triangles"

    @property
    def grid(self):
        # ORBIT patterns.grid_pattern: arange(-size/2, size/2 + 1e-9, resolution)
        nx = int(math.floor((self.size[0] + 1e-9) / self.resolution)) + 1
        ny = int(math.floor((self.size[1] + 1e-9) / self.resolution)) + 1
        return nx, ny


@dataclass
class AckermannActionCfg:
    """actions_cfg.py:9-51 with the AAU values of aau_rover/env_cfg.py:21-31."""
    asset_name: str = "robot"
    scale: tuple = (1.0, 1.0)
    offset: Any = -0.0135            # scalar => broadcast to both channels (reference quirk B-4)
    wheelbase_length: float = 0.849
    middle_wheel_distance: float = 0.894
    rear_and_front_wheel_distance: float = 0.77
    wheel_radius: float = 0.1
    min_steering_radius: float = 0.8
    steering_joint_names: tuple = (".*Steer_Revolute",)
    drive_joint_names: tuple = (".*Drive_Continuous",)


@dataclass
class CommandCfg:
    """CommandsCfg.target_pose (rover_env_cfg.py:187-200) + RoverTerrainImporter.target_distance (terrain_importer.py:132)."""
    resampling_time_range: tuple = (150.0, 150.0)
    heading_range: tuple = (-math.pi, math.pi)
    simple_heading: bool = False
    target_distance: float = 9.0
    max_target_tries: int = 32


@dataclass
class TerrainCfg:
    """Synthetic stand-in for the missing terrain USDs: ``kind`` in {"procedural", "flat", "custom"}."""
    kind: str = "procedural"
    shape: tuple = (2048, 2048)
    seed: int = 1234
    sigma_z: float = 0.15
    n_rocks: int = 400
    terrain: Any = None              # an isaac_rover_orbit_amd.terrain.Terrain for kind == "custom"
    spawn_seed: int = 41


def _default_observations():
    return {
        "actions": TermCfg("last_action"),
        "distance": TermCfg("distance_to_target_euclidean", scale=0.11, params={"command_name": "target_pose"}),
        "heading": TermCfg("angle_to_target_observation", scale=1 / math.pi, params={"command_name": "target_pose"}),
        "height_scan": TermCfg("height_scan_rover", scale=1, params={"sensor_cfg": "height_scanner"}),
    }


def _default_rewards():
    return {
        "distance_to_target": TermCfg("distance_to_target_reward", weight=5.0, params={"command_name": "target_pose"}),
        "reached_target": TermCfg("reached_target", weight=5.0, params={"command_name": "target_pose", "threshold": 0.18}),
        "oscillation": TermCfg("oscillation_penalty", weight=-0.1),
        "angle_to_target": TermCfg("angle_to_target_penalty", weight=-1.5, params={"command_name": "target_pose"}),
        "heading_soft_contraint": TermCfg("heading_soft_contraint", weight=-0.5, params={"asset_cfg": "robot"}),
        "collision": TermCfg("collision_penalty", weight=-2.0, params={"sensor_cfg": "contact_sensor", "threshold": 1.0}),
        "far_from_target": TermCfg("far_from_target_reward", weight=-2.0, params={"command_name": "target_pose", "threshold": 11.0}),
    }


def _default_terminations():
    return {
        "time_limit": TermCfg("time_out", time_out=True),
        "is_success": TermCfg("is_success", params={"command_name": "target_pose", "threshold": 0.18}),
        "far_from_target": TermCfg("far_from_target", params={"command_name": "target_pose", "threshold": 11.0}),
        "collision": TermCfg("collision_with_obstacles", params={"sensor_cfg": "contact_sensor", "threshold": 1.0}),
    }


REWARD_ORDER = ["distance_to_target", "reached_target", "oscillation", "angle_to_target", "heading_soft_contraint",
                "collision", "far_from_target"]
REWARD_FUNCS = ["distance_to_target_reward", "reached_target", "oscillation_penalty", "angle_to_target_penalty",
                "heading_soft_contraint", "collision_penalty", "far_from_target_reward"]
TERMINATION_ORDER = ["time_limit", "is_success", "far_from_target", "collision"]
TERMINATION_FUNCS = ["time_out", "is_success", "far_from_target", "collision_with_obstacles"]
OBS_ORDER = ["actions", "distance", "heading", "height_scan"]
OBS_FUNCS = ["last_action", "distance_to_target_euclidean", "angle_to_target_observation", "height_scan_rover"]


@dataclass
class RoverEnvCfg:
    scene: SceneCfg = field(default_factory=SceneCfg)
    sim: SimCfg = field(default_factory=SimCfg)
    decimation: int = 6                      # rover_env_cfg.py:270
    episode_length_s: float = 150.0          # rover_env_cfg.py:271
    height_scanner: RayCasterCfg = field(default_factory=RayCasterCfg)
    actions: AckermannActionCfg = field(default_factory=AckermannActionCfg)
    commands: CommandCfg = field(default_factory=CommandCfg)
    observations: dict = field(default_factory=_default_observations)
    rewards: dict = field(default_factory=_default_rewards)
    terminations: dict = field(default_factory=_default_terminations)
    terrain: TerrainCfg = field(default_factory=TerrainCfg)
    reset_z_offset: float = 0.5              # randomizations.py:12
    reset_velocities: str = "reference"      # "reference" (root pose only, B-17) | "zero"
    # spawn row of a reset (randomizations.py:22 draws a randperm prefix): "distinct" = no two envs reset in the same call
    # share a row (like the reference), "independent" = one uniform draw per env (rows may repeat)
    spawn_draw: str = "distinct"
    seed: int = 0
    friction: float = 0.75
    solver_iterations: int = 32             # Jacobi sweeps of the contact solver = the reference's solver_position_iteration_count
                                            # (aau_rover_simple.py:33); rounds 1-4 ran 16
    # where the rover's 25 kg act: "subtree_weights" = the weight of each bogie subtree (beam + steer links + wheels: 7 + 7 + 9 kg
    # of the USD link table) at its own centre of mass -- the static wheel loads of the articulated rover; "lumped" = rounds 1-4
    mass_model: str = "subtree_weights"
    step_mapping: str = "auto"              # "auto" | "lane" (one env per lane) | "group" (sixteen lanes per env)
    record_contact_forces: bool = True       # materialise contact_sensor.data.force_matrix_w every step
    use_int16_terrain: bool = True           # stage the exact int16 copy of the heightfield in the scan kernel when it exists
    roctx_markers: bool = False              # roctx ranges around the two launches of every step (rocprofv3 --marker-trace)
    # extras["log"]: "on_demand" = the reduction of the episodic sums runs when the dictionary is read (same numbers; lets a step be
    # ONE kernel launch where the fused step + scan kernel applies); "every_step" = behind every step, like the C entry's default
    log_reduction: str = "on_demand"
    # observation rows written with streaming (non-temporal) stores by the one-launch kernels: ~2 % faster step when nothing on the
    # device reads the rows next; off by default because a policy kernel behind the step finds plainly stored rows in L2
    stream_observations: bool = False
    log_values: str = "device"              # extras["log"] entries: 0-d device tensors (ORBIT) | "host": views of a pinned mirror, ONE
                                            # synchronisation per step for a trainer that .item()s every entry (skrl_utils.py:139-142)
    # multi-GPU sharding: this process simulates global env ids [env_id_offset, env_id_offset + num_envs)
    env_id_offset: int = 0
    global_num_envs: int | None = None

    # ------------------------------------------------------------------------------------------------------------
    def custom_terms(self, table: dict, order: list) -> dict:
        """The user-written entries of a term table: everything beside the built-in names, ``func`` a callable."""
        return {k: t for k, t in table.items() if k not in order}

    @property
    def has_custom_terms(self) -> bool:
        return bool(self.custom_terms(self.rewards, REWARD_ORDER) or self.custom_terms(self.terminations, TERMINATION_ORDER))

    def validate(self):
        for table, order, funcs, what in ((self.rewards, REWARD_ORDER, REWARD_FUNCS, "reward"),
                                          (self.terminations, TERMINATION_ORDER, TERMINATION_FUNCS, "termination"),
                                          (self.observations, OBS_ORDER, OBS_FUNCS, "observation")):
            builtin = [k for k in table if k in order]
            if builtin != order:
                raise ValueError(f"the built-in {what} terms {order} must all be present, in this order (fused in the HIP kernels; "
                                 f"switch a reward off with weight=0); got {list(table)}")
            for name, fn in zip(order, funcs):
                if table[name].func != fn:
                    raise ValueError(f"{what} term '{name}' must use func '{fn}' (got '{table[name].func}')")
            for name, t in self.custom_terms(table, order).items():
                if what == "observation":
                    raise ValueError(f"observation term '{name}': the observation row is fixed (4 + rays columns, the layout the policy "
                                     "checkpoint expects); user-written terms are supported for rewards and terminations")
                if not callable(t.func):
                    raise ValueError(f"{what} term '{name}': func must be a callable func(env, **params) (got {t.func!r}); the built-in "
                                     f"{what} terms are {order}")
        if self.reset_velocities not in ("reference", "zero"):
            raise ValueError("reset_velocities must be 'reference' or 'zero'")
        if self.spawn_draw not in ("distinct", "independent"):
            raise ValueError("spawn_draw must be 'distinct' or 'independent'")
        if self.log_values not in ("device", "host"):
            raise ValueError("log_values must be 'device' or 'host'")
        if self.mass_model not in ("lumped", "subtree_weights"):
            raise ValueError("mass_model must be 'subtree_weights' or 'lumped'")
        if self.commands.simple_heading:
            raise ValueError("simple_heading=True is not supported (the reference cfg uses False, rover_env_cfg.py:195)")

    def observation_post(self) -> dict:
        """Observation terms whose ORBIT post-processing (ObservationManager.compute_group: noise, then clip, then scale) is not
        what the kernels compute: any ``noise`` / ``clip``, or a ``scale`` != 1 on ``actions`` / ``height_scan`` (the kernels scale
        ``distance`` and ``heading`` only).  For these the kernels write the raw term (scale 1) and ``RoverEnv`` finishes the columns
        in torch (three small ops per term and step: not the one-launch fast path any more, but the same step kernel)."""
        out = {}
        for name in OBS_ORDER:
            t = self.observations[name]
            if t.noise is not None or t.clip is not None or (name in ("actions", "height_scan") and float(t.scale) != 1.0):
                if t.clip is not None and len(tuple(t.clip)) != 2:
                    raise ValueError(f"observation term '{name}': clip must be (min, max)")
                out[name] = t
        return out

    @property
    def max_episode_length(self) -> int:
        # ORBIT: ceil(episode_length_s / (sim.dt * decimation))
        return math.ceil(self.episode_length_s / (self.sim.dt * self.decimation))

    def to_native(self) -> "_lib.RoverConfig":
        self.validate()
        c = _lib.default_config()
        a = self.actions
        off = a.offset if isinstance(a.offset, (tuple, list)) else (a.offset, a.offset)
        c.scale_lin, c.scale_ang = float(a.scale[0]), float(a.scale[1])
        c.offset_lin, c.offset_ang = float(off[0]), float(off[1])
        c.wheel_radius, c.d_fr = a.wheel_radius, a.rear_and_front_wheel_distance
        c.d_mw, c.wheelbase = a.middle_wheel_distance, a.wheelbase_length
        c.sim_dt, c.decimation = self.sim.dt, self.decimation
        c.max_episode_length = self.max_episode_length
        c.max_episode_length_s = self.episode_length_s
        # four independent table entries in the reference (rover_env_cfg.py:136, 162 rewards; :173, 177 terminations)
        c.success_threshold = self.terminations["is_success"].params["threshold"]
        c.far_threshold = self.terminations["far_from_target"].params["threshold"]
        c.rew_success_threshold = self.rewards["reached_target"].params["threshold"]
        c.rew_far_threshold = self.rewards["far_from_target"].params["threshold"]
        c.target_distance = self.commands.target_distance
        c.heading_lo, c.heading_hi = self.commands.heading_range
        if self.commands.resampling_time_range[0] != self.commands.resampling_time_range[1]:
            raise ValueError("resampling_time_range must be a single value (rover_env_cfg.py:196)")
        c.resample_time = self.commands.resampling_time_range[0]
        for i, name in enumerate(REWARD_ORDER):
            c.rew_weight[i] = self.rewards[name].weight
        post = self.observation_post()         # terms finished in torch get their raw value from the kernels
        c.obs_scale_distance = 1.0 if "distance" in post else self.observations["distance"].scale
        c.obs_scale_heading = 1.0 if "heading" in post else self.observations["heading"].scale
        hs = self.height_scanner
        c.scan_resolution, c.scan_size_x, c.scan_size_y = hs.resolution, hs.size[0], hs.size[1]
        c.scan_nx, c.scan_ny = hs.grid
        c.scan_height_offset = hs.height_offset
        c.scan_surface = {"triangles": 0, "bilinear": 1}[hs.surface]
        c.reset_z_offset = self.reset_z_offset
        c.reset_mode = 0 if self.reset_velocities == "reference" else 1
        c.spawn_draw = {"independent": 0, "distinct": 1}[self.spawn_draw]
        c.seed_lo, c.seed_hi = self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF
        c.friction_mu = self.friction
        c.solver_iterations = self.solver_iterations
        c.mass_model = {"lumped": 0, "subtree_weights": 1}[self.mass_model]
        c.max_target_tries = self.commands.max_target_tries
        c.step_mapping = {"auto": 0, "lane": 1, "group": 2}[self.step_mapping]
        return c


@dataclass
class AAURoverEnvCfg(RoverEnvCfg):
    """``AAURoverEnv-v0`` (robots/aau_rover/env_cfg.py:10-31): the defaults above already are the AAU values."""
