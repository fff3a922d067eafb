"""Dev-only helper: import the *arithmetic* of the reference (pure torch/numpy code) in a
container that has none of its third-party runtime (Isaac Sim / ORBIT / carb / pxr / pymeshlab).

Used only by ``tools/gen_golden.py`` to produce the fixtures under ``tests/golden``; the reference
source itself never enters this repository and never travels to the GPU box.

What is stubbed: every ``omni.*`` / ``carb`` / ``pxr`` / ``pymeshlab`` module the reference imports at
top level.  The stubs manufacture empty, subclassable types on attribute
access, so ``class AckermannAction2(ActionTerm)`` still executes; none of the stubbed names is ever
*called* by the functions we evaluate.
"""
from __future__ import annotations

import sys
import types

REFERENCE_ROOT = "/root/reference"


class _Stub(types.ModuleType):
    __all__: list = []
    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        t = type(name, (), {})
        setattr(self, name, t)
        return t


_STUBBED = [
    "carb", "pymeshlab", "pxr", "cv2",
    "omni", "omni.isaac", "omni.isaac.core", "omni.isaac.core.utils", "omni.isaac.core.utils.prims",
    "omni.isaac.core.utils.stage", "omni.isaac.core.materials", "omni.isaac.core.prims",
    "omni.isaac.orbit", "omni.isaac.orbit.assets", "omni.isaac.orbit.assets.articulation",
    "omni.isaac.orbit.envs", "omni.isaac.orbit.envs.mdp", "omni.isaac.orbit.managers",
    "omni.isaac.orbit.managers.action_manager", "omni.isaac.orbit.utils", "omni.isaac.orbit.utils.math",
    "omni.isaac.orbit.sensors", "omni.isaac.orbit.markers", "omni.isaac.orbit.markers.config",
    "omni.isaac.orbit.terrains", "omni.isaac.orbit.sim", "omni.isaac.orbit.scene",
    "omni.isaac.orbit.actuators", "omni.isaac.orbit.utils.noise", "omni.isaac.orbit.envs.base_env",
    "omni.isaac.orbit.envs.rl_task_env", "omni.isaac.orbit.utils.configclass",
]


def install() -> None:
    for name in _STUBBED:
        if name not in sys.modules:
            sys.modules[name] = _Stub(name)
    for name in _STUBBED:
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[name])
    sys.modules["omni.isaac.orbit.utils"].configclass = lambda c: c
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
