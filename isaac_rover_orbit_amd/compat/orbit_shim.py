"""Import-compatibility namespace for the subset of ``omni.isaac.orbit`` (ORBIT, Isaac Sim) that the reference's
``AAURoverEnv-v0`` cfg files, learning glue and examples import.

Nothing here simulates anything: the classes are *declarative* stand-ins (configclass tables, marker base classes) so
that ``rover_envs/envs/navigation/rover_env_cfg.py``, ``robots/aau_rover/env_cfg.py``, ``assets/robots/aau_rover_simple.py``,
``mdp/*.py`` ... import unchanged and produce cfg objects that ``compat.convert.from_reference_cfg`` turns into the
kernel parameter block.  The environment class behind ``RLTaskEnv`` is the HIP-backed ``isaac_rover_orbit_amd.envs.RoverEnv``.
"""
from __future__ import annotations

import copy
import dataclasses
import math
import pickle
import sys
import types

MISSING = dataclasses.MISSING


# ------------------------------------------------------------------------------------------------- configclass
def _is_field(name, value, annotations):
    if name.startswith("__") or name in ("_open",):
        return False
    if name in annotations:
        return True
    if isinstance(value, (types.FunctionType, classmethod, staticmethod, property, type)):
        return False
    return True


def _collect_fields(cls):
    fields = {}
    for klass in reversed(cls.__mro__):
        if klass is object:
            continue
        ann = klass.__dict__.get("__annotations__", {})
        for name in ann:
            if not name.startswith("__"):
                fields.setdefault(name, MISSING)
        for name, value in klass.__dict__.items():
            if _is_field(name, value, ann):
                fields[name] = value
    return fields


def configclass(cls=None, **_):
    """ORBIT's ``@configclass``: dataclass-like, mutable defaults are deep-copied per instance, ``replace`` / ``copy`` /
    ``to_dict`` helpers, un-annotated class attributes are fields too."""

    def wrap(c):
        fields = _collect_fields(c)
        c.__cfg_fields__ = fields
        user_post_init = c.__dict__.get("__post_init__")

        def __init__(self, **kwargs):
            flds = type(self).__cfg_fields__
            for name, default in flds.items():
                if name in kwargs:
                    setattr(self, name, kwargs.pop(name))
                else:
                    setattr(self, name, default if default is MISSING else copy.deepcopy(default))
            if kwargs:
                if getattr(type(self), "_open", False):
                    for k, v in kwargs.items():
                        setattr(self, k, v)
                else:
                    raise TypeError(f"{type(self).__name__}() got unexpected field(s) {sorted(kwargs)}")
            post = getattr(self, "__post_init__", None)
            if post is not None:
                post()

        def replace(self, **kw):
            new = copy.deepcopy(self)
            for k, v in kw.items():
                setattr(new, k, v)
            return new

        def to_dict(self):
            out = {}
            for k, v in vars(self).items():
                out[k] = v.to_dict() if hasattr(v, "to_dict") else v
            return out

        def __repr__(self):
            inner = ", ".join(f"{k}={v!r}" for k, v in vars(self).items())
            return f"{type(self).__name__}({inner})"

        c.__init__ = __init__
        c.replace = replace
        c.copy = lambda self: copy.deepcopy(self)
        c.to_dict = to_dict
        c.__repr__ = __repr__
        if user_post_init is None and not any("__post_init__" in k.__dict__ for k in c.__mro__[1:]):
            c.__post_init__ = lambda self: None
        return c

    return wrap if cls is None else wrap(cls)


@configclass
class _OpenCfg:
    """A cfg that accepts any keyword (spawners, PhysX, materials ... whose fields are irrelevant off Isaac Sim)."""
    _open = True


# ------------------------------------------------------------------------------------------------- cfg classes
class SceneEntityCfg:
    """``SceneEntityCfg("contact_sensor")`` / ``SceneEntityCfg(name="robot")`` -- positional name as in ORBIT."""

    def __init__(self, name=MISSING, joint_names=None, joint_ids=slice(None), body_names=None, body_ids=slice(None)):
        self.name, self.joint_names, self.joint_ids = name, joint_names, joint_ids
        self.body_names, self.body_ids = body_names, body_ids

    def replace(self, **kw):
        new = copy.deepcopy(self)
        for k, v in kw.items():
            setattr(new, k, v)
        return new

    def to_dict(self):
        return dict(vars(self))

    def __repr__(self):
        return f"SceneEntityCfg(name={self.name!r})"


@configclass
class ManagerTermBaseCfg:
    func = MISSING
    params: dict = {}


@configclass
class ObservationTermCfg(ManagerTermBaseCfg):
    noise = None
    clip = None
    scale = None


@configclass
class ObservationGroupCfg:
    concatenate_terms: bool = True
    enable_corruption: bool = False


@configclass
class RewardTermCfg(ManagerTermBaseCfg):
    weight: float = MISSING


@configclass
class TerminationTermCfg(ManagerTermBaseCfg):
    time_out: bool = False


@configclass
class RandomizationTermCfg(ManagerTermBaseCfg):
    mode: str = MISSING
    interval_range_s = None


@configclass
class CurriculumTermCfg(ManagerTermBaseCfg):
    pass


@configclass
class ActionTermCfg:
    class_type = MISSING
    asset_name: str = MISSING


@configclass
class CommandTermCfg:
    class_type = MISSING
    resampling_time_range = MISSING
    debug_vis: bool = False


@configclass
class InteractiveSceneCfg:
    num_envs: int = MISSING
    env_spacing: float = MISSING
    lazy_sensor_update: bool = True
    replicate_physics: bool = True


@configclass
class AssetBaseCfg:
    @configclass
    class InitialStateCfg:
        pos = (0.0, 0.0, 0.0)
        rot = (1.0, 0.0, 0.0, 0.0)

    class_type = None
    prim_path: str = MISSING
    spawn = None
    init_state = InitialStateCfg()
    collision_group = 0
    debug_vis: bool = False


@configclass
class ArticulationCfg(AssetBaseCfg):
    @configclass
    class InitialStateCfg(AssetBaseCfg.InitialStateCfg):
        lin_vel = (0.0, 0.0, 0.0)
        ang_vel = (0.0, 0.0, 0.0)
        joint_pos: dict = {".*": 0.0}
        joint_vel: dict = {".*": 0.0}

    init_state = InitialStateCfg()
    soft_joint_pos_limit_factor: float = 1.0
    actuators: dict = MISSING


@configclass
class RigidObjectCfg(AssetBaseCfg):
    pass


@configclass
class ImplicitActuatorCfg:
    class_type = None
    joint_names_expr = MISSING
    effort_limit = None
    velocity_limit = None
    stiffness = MISSING
    damping = MISSING


@configclass
class SensorBaseCfg:
    class_type = None
    prim_path: str = MISSING
    update_period: float = 0.0
    history_length: int = 0
    debug_vis: bool = False


@configclass
class ContactSensorCfg(SensorBaseCfg):
    track_pose: bool = False
    track_air_time: bool = False
    force_threshold: float = 1.0
    filter_prim_paths_expr: list = []


@configclass
class GridPatternCfg:
    resolution: float = MISSING
    size = MISSING
    direction = (0.0, 0.0, -1.0)
    ordering: str = "xy"


@configclass
class RayCasterCfg(SensorBaseCfg):
    @configclass
    class OffsetCfg:
        pos = (0.0, 0.0, 0.0)
        rot = (1.0, 0.0, 0.0, 0.0)

    mesh_prim_paths: list = MISSING
    offset = OffsetCfg()
    attach_yaw_only: bool = MISSING
    pattern_cfg = MISSING
    max_distance: float = 1e6
    drift_range = (0.0, 0.0)


@configclass
class TerrainImporterCfg:
    class_type = None
    collision_group: int = -1
    prim_path: str = MISSING
    num_envs: int = 1
    terrain_type: str = "generator"
    terrain_generator = None
    usd_path = None
    env_spacing = None
    visual_material = None
    physics_material = None
    max_init_terrain_level = None
    debug_vis: bool = False


@configclass
class ViewerCfg:
    eye = (7.5, 7.5, 7.5)
    lookat = (0.0, 0.0, 0.0)
    cam_prim_path: str = "/OmniverseKit_Persp"
    resolution = (1280, 720)


class PhysxCfg(_OpenCfg):
    pass


@configclass
class SimulationCfg:
    physics_prim_path: str = "/physicsScene"
    dt: float = 1.0 / 60.0
    substeps: int = 1
    gravity = (0.0, 0.0, -9.81)
    enable_scene_query_support: bool = False
    use_fabric: bool = True
    disable_contact_processing: bool = False
    use_gpu_pipeline: bool = True
    device: str = "cuda:0"
    physx = PhysxCfg()
    physics_material = None


@configclass
class BaseEnvCfg:
    viewer = ViewerCfg()
    sim = SimulationCfg()
    ui_window_class_type = None
    decimation: int = MISSING
    scene = MISSING
    observations = MISSING
    actions = MISSING
    randomization = None


@configclass
class RLTaskEnvCfg(BaseEnvCfg):
    is_finite_horizon: bool = False
    episode_length_s: float = MISSING
    rewards = MISSING
    terminations = MISSING
    curriculum = None
    commands = None


def additive_uniform_noise(data, cfg):       # omni.isaac.orbit.utils.noise (not in /root/reference): data + U(n_min, n_max)
    import torch
    return data + torch.rand_like(data) * (cfg.n_max - cfg.n_min) + cfg.n_min


def additive_gaussian_noise(data, cfg):
    import torch
    return data + cfg.mean + cfg.std * torch.randn_like(data)


def constant_bias_noise(data, cfg):
    return data + cfg.bias


@configclass
class AdditiveUniformNoiseCfg:
    func = staticmethod(additive_uniform_noise)
    n_min: float = -1.0
    n_max: float = 1.0


@configclass
class AdditiveGaussianNoiseCfg:
    func = staticmethod(additive_gaussian_noise)
    mean: float = 0.0
    std: float = 1.0


@configclass
class ConstantBiasNoiseCfg:
    func = staticmethod(constant_bias_noise)
    bias: float = 0.0


# ------------------------------------------------------------------------------------------------- marker classes
class ActionTerm:
    """Base class name only: the Ackermann term is fused into the step kernel."""

    def __init__(self, cfg=None, env=None):
        self.cfg, self._env = cfg, env


class CommandTerm:
    def __init__(self, cfg=None, env=None):
        self.cfg, self._env = cfg, env
        self.metrics = {}


class _Unavailable:
    """Stands for a simulator-side class (Articulation, RayCaster ...) that has no counterpart off Isaac Sim."""

    def __init__(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is an Isaac Sim / ORBIT runtime class; the MI355X path replaces it "
                           "with fused HIP kernels (see DESIGN.md)")


def _unavailable(name):
    return type(name, (_Unavailable,), {})


# mdp term functions referenced by NAME in the cfg tables (fused in the kernels)
def last_action(env):
    return env.action_manager.action


def time_out(env):
    return env.episode_length_buf >= env.max_episode_length


# ------------------------------------------------------------------------------------------------- utils.math
def yaw_quat(quat):
    import torch
    qw, qx, qy, qz = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    yaw = torch.atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz))
    out = torch.zeros_like(quat)
    out[:, 3] = torch.sin(yaw / 2)
    out[:, 0] = torch.cos(yaw / 2)
    return out / out.norm(dim=-1, keepdim=True)


def quat_rotate_inverse(q, v):
    import torch
    q_w, q_vec = q[:, 0], q[:, 1:]
    a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * torch.bmm(q_vec.view(-1, 1, 3), v.view(-1, 3, 1)).squeeze(-1) * 2.0
    return a - b + c


def wrap_to_pi(angles):
    angles = angles.clone()
    angles %= 2 * math.pi
    angles -= 2 * math.pi * (angles > math.pi)
    return angles


def print_dict(val, nesting: int = 0, start: bool = True):
    if isinstance(val, dict):
        if not start:
            print("")
        for k in val:
            print(" " * nesting + f"{k}: ", end="")
            print_dict(val[k], nesting + 4, start=False)
    else:
        print(val)


def dump_pickle(filename, data):
    import os
    os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
    with open(filename if filename.endswith(".pkl") else filename + ".pkl", "wb") as f:
        pickle.dump(data.to_dict() if hasattr(data, "to_dict") else data, f)


def dump_yaml(filename, data, sort_keys: bool = False):
    import os

    import yaml
    os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)

    def plain(x):
        if hasattr(x, "to_dict"):
            x = x.to_dict()
        if isinstance(x, dict):
            return {str(k): plain(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [plain(v) for v in x]
        if isinstance(x, (int, float, str, bool)) or x is None:
            return x
        return repr(x)

    with open(filename if filename.endswith(".yaml") else filename + ".yaml", "w") as f:
        yaml.safe_dump(plain(data), f, sort_keys=sort_keys)


# ------------------------------------------------------------------------------------------------- app launcher
class _App:
    def is_running(self):
        return True

    def close(self):
        pass


class AppLauncher:
    """``omni.isaac.orbit.app.AppLauncher``: there is no Omniverse Kit to boot; keeps the CLI contract of train.py."""

    def __init__(self, launcher_args=None, **kwargs):
        self.app = _App()

    @staticmethod
    def add_app_launcher_args(parser):
        g = parser.add_argument_group("app_launcher arguments (ignored: no Omniverse Kit on the MI355X path)")
        for flag in ("--headless", "--livestream", "--offscreen_render", "--verbose", "--experience"):
            try:
                if flag in ("--livestream",):
                    g.add_argument(flag, type=int, default=-1)
                elif flag == "--experience":
                    g.add_argument(flag, type=str, default="")
                else:
                    g.add_argument(flag, action="store_true", default=False)
            except Exception:  # the script already defines the flag (train.py adds --headless itself? no; be lenient)
                pass


# ------------------------------------------------------------------------------------------------- module tree
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    return m


def build_modules() -> dict:
    """name -> module for everything the census of SURVEY 8b lists."""
    from ..envs.rover_env import RLTaskEnv

    class BaseEnv(RLTaskEnv):
        pass

    spawn_names = ["UsdFileCfg", "DomeLightCfg", "SphereLightCfg", "DistantLightCfg", "CollisionPropertiesCfg",
                   "RigidBodyPropertiesCfg", "ArticulationRootPropertiesCfg", "RigidBodyMaterialCfg", "PreviewSurfaceCfg",
                   "MassPropertiesCfg", "GroundPlaneCfg", "SphereCfg", "CuboidCfg", "MdlFileCfg", "SpawnerCfg"]
    sim_attrs = {n: type(n, (_OpenCfg,), {}) for n in spawn_names}
    sim_attrs.update(SimulationCfg=SimulationCfg, PhysxCfg=PhysxCfg, SimulationContext=_unavailable("SimulationContext"))
    sim = _module("omni.isaac.orbit.sim", **sim_attrs)

    utils_math = _module("omni.isaac.orbit.utils.math", yaw_quat=yaw_quat, quat_rotate_inverse=quat_rotate_inverse,
                         wrap_to_pi=wrap_to_pi)
    utils_cc = _module("omni.isaac.orbit.utils.configclass", configclass=configclass)
    utils = _module("omni.isaac.orbit.utils", configclass=configclass, math=utils_math,
                    noise=_module("omni.isaac.orbit.utils.noise", AdditiveUniformNoiseCfg=AdditiveUniformNoiseCfg,
                                  AdditiveGaussianNoiseCfg=AdditiveGaussianNoiseCfg, ConstantBiasNoiseCfg=ConstantBiasNoiseCfg),
                    dict=_module("omni.isaac.orbit.utils.dict", print_dict=print_dict),
                    io=_module("omni.isaac.orbit.utils.io", dump_pickle=dump_pickle, dump_yaml=dump_yaml))
    utils.configclass = configclass   # `from omni.isaac.orbit.utils import configclass` must give the decorator

    commands_cfg = _module("omni.isaac.orbit.envs.mdp.commands.commands_cfg", CommandTermCfg=CommandTermCfg)
    commands = _module("omni.isaac.orbit.envs.mdp.commands", commands_cfg=commands_cfg)
    mdp = _module("omni.isaac.orbit.envs.mdp", last_action=last_action, time_out=time_out, commands=commands,
                  __all__=["last_action", "time_out"])
    base_env = _module("omni.isaac.orbit.envs.base_env", BaseEnv=BaseEnv, VecEnvObs=dict)
    rl_task_env = _module("omni.isaac.orbit.envs.rl_task_env", RLTaskEnv=RLTaskEnv, VecEnvStepReturn=tuple)
    envs = _module("omni.isaac.orbit.envs", RLTaskEnv=RLTaskEnv, RLTaskEnvCfg=RLTaskEnvCfg, BaseEnv=BaseEnv,
                   BaseEnvCfg=BaseEnvCfg, ViewerCfg=ViewerCfg, mdp=mdp, base_env=base_env, rl_task_env=rl_task_env)

    action_manager = _module("omni.isaac.orbit.managers.action_manager", ActionTerm=ActionTerm, ActionTermCfg=ActionTermCfg)
    managers = _module("omni.isaac.orbit.managers", ActionTermCfg=ActionTermCfg, ActionTerm=ActionTerm,
                       CurriculumTermCfg=CurriculumTermCfg, ObservationGroupCfg=ObservationGroupCfg,
                       ObservationTermCfg=ObservationTermCfg, RandomizationTermCfg=RandomizationTermCfg,
                       RewardTermCfg=RewardTermCfg, SceneEntityCfg=SceneEntityCfg, TerminationTermCfg=TerminationTermCfg,
                       CommandTerm=CommandTerm, CommandTermCfg=CommandTermCfg, action_manager=action_manager)

    Articulation, RigidObject = _unavailable("Articulation"), _unavailable("RigidObject")
    articulation = _module("omni.isaac.orbit.assets.articulation", Articulation=Articulation, ArticulationCfg=ArticulationCfg)
    assets = _module("omni.isaac.orbit.assets", ArticulationCfg=ArticulationCfg, AssetBaseCfg=AssetBaseCfg,
                     Articulation=Articulation, RigidObject=RigidObject, RigidObjectCfg=RigidObjectCfg,
                     articulation=articulation)
    patterns = _module("omni.isaac.orbit.sensors.patterns", GridPatternCfg=GridPatternCfg)
    sensors = _module("omni.isaac.orbit.sensors", ContactSensorCfg=ContactSensorCfg, RayCasterCfg=RayCasterCfg,
                      patterns=patterns, ContactSensor=_unavailable("ContactSensor"), RayCaster=_unavailable("RayCaster"))
    TerrainImporter = type("TerrainImporter", (), {"__init__": lambda self, cfg=None: setattr(self, "cfg", cfg)})
    terrains = _module("omni.isaac.orbit.terrains", TerrainImporter=TerrainImporter, TerrainImporterCfg=TerrainImporterCfg)
    markers_cfg = _module("omni.isaac.orbit.markers.config", CUBOID_MARKER_CFG=_OpenCfg(markers={"cuboid": _OpenCfg(scale=(1, 1, 1))}))
    markers = _module("omni.isaac.orbit.markers", VisualizationMarkers=_unavailable("VisualizationMarkers"), config=markers_cfg)
    scene = _module("omni.isaac.orbit.scene", InteractiveSceneCfg=InteractiveSceneCfg, InteractiveScene=_unavailable("InteractiveScene"))
    actuators = _module("omni.isaac.orbit.actuators", ImplicitActuatorCfg=ImplicitActuatorCfg)
    app = _module("omni.isaac.orbit.app", AppLauncher=AppLauncher)

    orbit = _module("omni.isaac.orbit", app=app, envs=envs, managers=managers, scene=scene, sensors=sensors, sim=sim,
                    assets=assets, actuators=actuators, terrains=terrains, markers=markers, utils=utils)

    def parse_env_cfg(task_name: str, use_gpu=None, num_envs=None, use_fabric=None):
        """``omni.isaac.orbit_tasks.utils.parse_env_cfg``: instantiate the task's cfg entry point, apply the CLI overrides."""
        from . import gym_api
        spec = gym_api().spec(task_name)
        entry = spec.kwargs["env_cfg_entry_point"]
        if isinstance(entry, str):
            mod, attr = entry.split(":")
            import importlib
            entry = getattr(importlib.import_module(mod), attr)
        cfg = entry() if callable(entry) else entry
        if use_gpu is not None and not use_gpu:
            raise RuntimeError("the MI355X rover path has no CPU pipeline (--cpu is not supported)")
        if num_envs is not None:
            cfg.scene.num_envs = num_envs
        return cfg

    orbit_tasks_utils = _module("omni.isaac.orbit_tasks.utils", parse_env_cfg=parse_env_cfg)
    orbit_tasks = _module("omni.isaac.orbit_tasks", utils=orbit_tasks_utils)

    stage = _module("omni.isaac.core.utils.stage", get_current_stage=lambda: None)
    prims_mod = _module("omni.isaac.core.utils.prims")
    core_utils = _module("omni.isaac.core.utils", stage=stage, prims=prims_mod)
    core = _module("omni.isaac.core", utils=core_utils,
                   materials=_module("omni.isaac.core.materials", PhysicsMaterial=_unavailable("PhysicsMaterial")),
                   prims=_module("omni.isaac.core.prims", XFormPrim=_unavailable("XFormPrim")))
    isaac = _module("omni.isaac", orbit=orbit, orbit_tasks=orbit_tasks, core=core)
    omni = _module("omni", isaac=isaac)

    carb = _module("carb", log_info=lambda *a: None, log_warn=lambda *a: print("[carb warn]", *a, file=sys.stderr),
                   log_error=lambda *a: print("[carb error]", *a, file=sys.stderr))
    pxr = _module("pxr", **{n: _module(f"pxr.{n}") for n in ("Usd", "UsdGeom", "Sdf", "PhysxSchema", "UsdPhysics", "Gf", "Vt")})

    mods = {}

    def walk(m):
        if m.__name__ in mods:
            return
        mods[m.__name__] = m
        for v in list(m.__dict__.values()):
            if isinstance(v, types.ModuleType) and v.__name__.startswith(m.__name__ + "."):
                walk(v)

    for root in (omni, carb, pxr):
        walk(root)
    # `from omni.isaac.orbit.utils import configclass` must give the decorator while
    # `from omni.isaac.orbit.utils.configclass import configclass` must find a module: register the latter by name only
    mods[utils_cc.__name__] = utils_cc
    return mods
