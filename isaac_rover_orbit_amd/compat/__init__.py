"""Drop-in compatibility layer: lets the reference's training scripts and cfg files run against
the MI355X-native environment without Isaac Sim / ORBIT.

    import isaac_rover_orbit_amd.compat as compat
    compat.install()                      # before importing rover_envs.* / examples/02_train/train.py
    import rover_envs.envs.navigation.robots           # the reference's own registration module, unchanged
    env = gym.make("AAURoverEnv-v0", cfg=parse_env_cfg("AAURoverEnv-v0", num_envs=4096))   # -> HIP-backed RoverEnv

``install()`` (1) registers the ``omni.isaac.orbit`` / ``omni.isaac.orbit_tasks`` / ``omni.isaac.core`` / ``carb`` / ``pxr``
import namespace of ``orbit_shim`` unless the real packages are importable, (2) provides a minimal gymnasium-style
registry when ``gymnasium`` is not installed, and (3) redirects the entry point of the rover task ids to
``isaac_rover_orbit_amd.envs:RoverEnv`` -- the reference's ``RoverEnv.step`` is exactly the hot path being replaced.
"""
from __future__ import annotations

import importlib
import importlib.util
import sys
import types

ROVER_TASK_IDS = ("AAURoverEnv-v0",)
_installed = False


# ------------------------------------------------------------------------------------------------- gym registry
class _Spec:
    def __init__(self, id, entry_point, kwargs):
        self.id, self.entry_point, self.kwargs = id, entry_point, dict(kwargs or {})


class _MiniGym(types.ModuleType):
    """The three calls the reference makes on ``gymnasium``: register / make / spec (+ spaces.Box)."""

    def __init__(self, name="gymnasium"):
        super().__init__(name)
        self.registry = {}
        from ..envs.rover_env import _spaces
        self.spaces = types.ModuleType(name + ".spaces")
        self.spaces.Box = _spaces.Box
        self.spaces.Dict = _spaces.Dict
        self.spaces.__path__ = []
        self.spaces.box = types.ModuleType(name + ".spaces.box")
        self.spaces.box.Box = _spaces.Box
        self.Env = object
        self.__path__ = []

    def register(self, id, entry_point=None, disable_env_checker=True, kwargs=None, **_):
        self.registry[id] = _Spec(id, entry_point, kwargs)

    def spec(self, id):
        return self.registry[id]

    def make(self, id, **kwargs):
        spec = self.registry[id]
        entry = spec.entry_point
        if isinstance(entry, str):
            mod, attr = entry.split(":")
            entry = getattr(importlib.import_module(mod), attr)
        kw = {k: v for k, v in spec.kwargs.items() if k not in ("env_cfg_entry_point", "best_model_path", "get_agent_fn")}
        kw.update(kwargs)
        return entry(**kw)


_gym = None


def gym_api():
    """gymnasium if installed, else the minimal registry above (shared singleton, also aliased as ``gym``)."""
    global _gym
    if _gym is None:
        try:
            import gymnasium
            _gym = gymnasium
        except ImportError:
            _gym = _MiniGym()
            sys.modules.setdefault("gymnasium", _gym)
            sys.modules.setdefault("gymnasium.spaces", _gym.spaces)
            sys.modules.setdefault("gymnasium.spaces.box", _gym.spaces.box)
    return _gym


def _redirect_entry_points(gym):
    """Route the rover task ids to the HIP-backed env whenever they get registered (now or later)."""
    target = "isaac_rover_orbit_amd.envs:RoverEnv"
    orig_register = gym.register

    def register(id, entry_point=None, **kw):
        if id in ROVER_TASK_IDS:
            entry_point = target
        return orig_register(id, entry_point=entry_point, **kw)

    if not getattr(gym.register, "_rover_redirect", False):
        register._rover_redirect = True
        gym.register = register
    reg = getattr(gym, "registry", None)
    if isinstance(reg, dict):
        for tid in ROVER_TASK_IDS:
            if tid in reg and hasattr(reg[tid], "entry_point"):
                try:
                    reg[tid].entry_point = target
                except Exception:
                    pass


def install(force: bool = False) -> dict:
    """Install the compatibility namespace; returns {module name: module} of what was registered."""
    global _installed
    from . import orbit_shim
    registered = {}
    have_real_orbit = False
    if not force:
        try:
            have_real_orbit = importlib.util.find_spec("omni.isaac.orbit") is not None
        except (ImportError, ValueError):
            have_real_orbit = False
    if not have_real_orbit:
        for name, mod in orbit_shim.build_modules().items():
            if force or name not in sys.modules:
                sys.modules[name] = mod
                registered[name] = mod
    # pymeshlab is only touched by TerrainManager.get_mesh (USD mesh read-out, terrain_utils.py:129-155), which this
    # path replaces by isaac_rover_orbit_amd.terrain; let the import of terrain_utils succeed without it
    try:
        have_pymeshlab = importlib.util.find_spec("pymeshlab") is not None
    except (ImportError, ValueError):
        have_pymeshlab = False
    if not have_pymeshlab and "pymeshlab" not in sys.modules:
        stub = types.ModuleType("pymeshlab")

        def _no_pymeshlab(*a, **k):
            raise RuntimeError("pymeshlab is not installed; use isaac_rover_orbit_amd.terrain.terrain_from_mesh instead")
        stub.Mesh = stub.MeshSet = _no_pymeshlab
        sys.modules["pymeshlab"] = stub
        registered["pymeshlab"] = stub
    gym = gym_api()
    if "gym" not in sys.modules:            # the reference's learning glue still imports the legacy name (`from gym.spaces import Box`)
        sys.modules["gym"] = gym
        sys.modules["gym.spaces"] = gym.spaces
        if hasattr(gym.spaces, "box"):
            sys.modules["gym.spaces.box"] = gym.spaces.box
    _redirect_entry_points(gym)
    _installed = True
    return registered


def register_default_tasks():
    """Register ``AAURoverEnv-v0`` with this package's own cfg (for users who do not have the reference checked out)."""
    from ..cfg import AAURoverEnvCfg
    gym = gym_api()
    gym.register(id="AAURoverEnv-v0", entry_point="isaac_rover_orbit_amd.envs:RoverEnv", disable_env_checker=True,
                 kwargs={"env_cfg_entry_point": AAURoverEnvCfg})
    from ..envs.lift_env import LiftEnvCfg
    # manipulation/config/franka/__init__.py:6-14 (entry point there: ORBIT's generic RLTaskEnv on FrankaCubeLiftEnvCfg)
    gym.register(id="FrankaCubeLift-v0", entry_point="isaac_rover_orbit_amd.envs:FrankaCubeLiftEnv", disable_env_checker=True,
                 kwargs={"env_cfg_entry_point": LiftEnvCfg})
