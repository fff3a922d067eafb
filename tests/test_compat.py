"""Drop-in layer: the reference's OWN registration module and cfg classes import unchanged on top of
``isaac_rover_orbit_amd.compat`` and drive the kernel parameter block.  Needs the reference checkout (build container
only); skrl is absent there, so an on-demand test double stands in for it."""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import math

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "rover_envs")), reason="reference checkout not present")


class _Stub(types.ModuleType):
    __all__ = []
    __path__ = []

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        t = type(n, (), {})
        setattr(self, n, t)
        return t


class _FakePackages(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def __init__(self, roots):
        self.roots = roots

    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in self.roots:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, m):
        pass


@pytest.fixture(scope="module")
def ref_env():
    import isaac_rover_orbit_amd.compat as compat
    saved_path, saved_meta = list(sys.path), list(sys.meta_path)
    before = set(sys.modules)
    compat.install()
    try:
        import skrl  # noqa: F401
    except ImportError:
        sys.meta_path.insert(0, _FakePackages({"skrl"}))
    sys.path.insert(0, REF)
    import rover_envs.envs.navigation.robots  # noqa: F401  (the reference's gym.register calls)
    yield compat
    sys.path[:], sys.meta_path[:] = saved_path, saved_meta
    for name in set(sys.modules) - before:
        if name.split(".")[0] in ("rover_envs", "skrl", "omni", "carb", "pxr", "pymeshlab", "gym"):
            sys.modules.pop(name, None)


def test_reference_registration_and_cfg_drive_the_kernels(ref_env):
    compat = ref_env
    from omni.isaac.orbit_tasks.utils import parse_env_cfg          # the name train.py imports (train.py:104)
    from isaac_rover_orbit_amd import _lib
    from isaac_rover_orbit_amd.compat.convert import from_reference_cfg
    gym = compat.gym_api()
    spec = gym.spec("AAURoverEnv-v0")
    assert spec.entry_point == "isaac_rover_orbit_amd.envs:RoverEnv"          # hot path redirected to the HIP env
    assert spec.kwargs["best_model_path"].endswith("policies/best_agent.pt")   # eval.py:148 still finds its checkpoint
    cfg = parse_env_cfg("AAURoverEnv-v0", use_gpu=True, num_envs=4096)
    assert type(cfg).__module__ == "rover_envs.envs.navigation.robots.aau_rover.env_cfg"   # the reference's own class
    assert cfg.scene.num_envs == 4096 and cfg.decimation == 6 and cfg.episode_length_s == 150
    converted = from_reference_cfg(cfg)
    # the gym-redirect / compat path serves the reference's trainer, which .item()s every log entry after every step
    # (skrl_utils.py:139-142): extras["log"] lives in the pinned host mirror there; the native cfg keeps ORBIT's device tensors
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    assert converted.log_values == "host" and RoverEnvCfg().log_values == "device"
    native = converted.to_native()
    default = _lib.default_config()
    for name, _ in _lib.RoverConfig._fields_:
        a, b = getattr(native, name), getattr(default, name)
        assert (list(a) == list(b)) if name == "rew_weight" else (a == b), name   # our defaults ARE the reference cfg
    # edits made the reference way reach the parameter block
    cfg.rewards.collision.weight = -7.0
    cfg.actions.actions.offset = 0.0
    cfg.scene.height_scanner.pattern_cfg.resolution = 0.2
    cfg.terminations.is_success.params["threshold"] = 0.25
    cfg.rewards.reached_target.params["threshold"] = 0.25
    n2 = from_reference_cfg(cfg).to_native()
    assert n2.rew_weight[5] == -7.0 and n2.offset_lin == 0.0 and n2.scan_nx == 16 and abs(n2.success_threshold - 0.25) < 1e-7
    with pytest.raises(RuntimeError):
        parse_env_cfg("AAURoverEnv-v0", use_gpu=False)                         # no CPU pipeline on this path


def test_termination_omitted_but_reward_kept(ref_env):
    """a cfg that drops the ``is_success`` / ``far_from_target`` TERMINATION but keeps the reward of the same name
    (independent entries in the reference: rover_env_cfg.py:136 / 162 vs :173 / 177) must keep that reward as configured."""
    from omni.isaac.orbit_tasks.utils import parse_env_cfg
    from isaac_rover_orbit_amd.compat.convert import from_reference_cfg
    cfg = parse_env_cfg("AAURoverEnv-v0", num_envs=8)
    del cfg.terminations.is_success
    del cfg.terminations.far_from_target
    n = from_reference_cfg(cfg).to_native()
    assert n.success_threshold == -1.0 and n.far_threshold == float("inf")         # the terminations can never fire
    assert abs(n.rew_success_threshold - 0.18) < 1e-7 and abs(n.rew_far_threshold - 11.0) < 1e-7   # the rewards are untouched
    assert n.rew_weight[1] == 5.0 and n.rew_weight[6] == -2.0


def test_reference_learning_glue_imports(ref_env):
    """The consumers of the env (examples/02_train/train.py:104-112) import on top of the shim."""
    import rover_envs.learning.train  # noqa: F401
    import rover_envs.utils.config as rcfg
    import rover_envs.utils.skrl_utils as su
    from omni.isaac.orbit.envs import RLTaskEnv
    from isaac_rover_orbit_amd.envs import RoverEnv
    assert issubclass(RoverEnv, RLTaskEnv)              # skrl_utils.py:38 isinstance check
    assert hasattr(su, "SkrlVecEnvWrapper") and hasattr(su, "SkrlSequentialLogTrainer") and hasattr(rcfg, "parse_skrl_cfg")
    exp = rcfg.parse_skrl_cfg("AAURoverEnv-v0_PPO")    # rover_ppo.yaml
    assert exp["agent"]["rollouts"] == 60


def test_user_written_term_in_reference_cfg_is_kept(ref_env):
    """The reference's term tables hold arbitrary ``func=`` (rover_env_cfg.py:126-183): a callable under a new key is a user-written
    term and survives the conversion (RoverEnv evaluates it on its slow path); what is NOT a callable, or a foreign function under a
    built-in key, still raises."""
    from omni.isaac.orbit.managers import RewardTermCfg, TerminationTermCfg
    from omni.isaac.orbit_tasks.utils import parse_env_cfg
    from isaac_rover_orbit_amd.cfg import REWARD_ORDER, TERMINATION_ORDER
    from isaac_rover_orbit_amd.compat.convert import from_reference_cfg
    cfg = parse_env_cfg("AAURoverEnv-v0", num_envs=8)
    bonus = lambda env, scale: scale          # noqa: E731
    cfg.rewards.my_bonus = RewardTermCfg(func=bonus, weight=1.5, params={"scale": 2.0})
    cfg.terminations.my_stop = TerminationTermCfg(func=lambda env: None, time_out=True)
    out = from_reference_cfg(cfg)
    assert list(out.rewards) == REWARD_ORDER + ["my_bonus"] and out.rewards["my_bonus"].func is bonus
    assert out.rewards["my_bonus"].weight == 1.5 and out.rewards["my_bonus"].params == {"scale": 2.0}
    assert list(out.terminations) == TERMINATION_ORDER + ["my_stop"] and out.terminations["my_stop"].time_out and out.has_custom_terms
    assert not from_reference_cfg(parse_env_cfg("AAURoverEnv-v0", num_envs=8)).has_custom_terms       # the stock table: fast path
    cfg = parse_env_cfg("AAURoverEnv-v0", num_envs=8)
    cfg.rewards.my_bonus = RewardTermCfg(func="not a callable", weight=1.0)
    with pytest.raises(ValueError):
        from_reference_cfg(cfg)
    cfg = parse_env_cfg("AAURoverEnv-v0", num_envs=8)
    cfg.rewards.oscillation = RewardTermCfg(func=bonus, weight=1.0)           # a built-in key with a foreign function
    with pytest.raises(ValueError):
        from_reference_cfg(cfg)


def test_observation_noise_and_clip_of_a_reference_cfg_are_kept(ref_env):
    """``Unoise`` is imported by the reference's cfg for exactly this (rover_env_cfg.py:23): ``noise=`` / ``clip=`` on an ObsTerm
    survive the conversion (RoverEnv applies them in torch, ORBIT's order); the term is then written raw by the kernels (scale 1);
    without ``enable_corruption`` the group's noise is dropped as ORBIT's ObservationManager drops it."""
    import torch
    from omni.isaac.orbit.utils.noise import AdditiveUniformNoiseCfg as Unoise
    from omni.isaac.orbit_tasks.utils import parse_env_cfg
    from isaac_rover_orbit_amd.compat.convert import from_reference_cfg
    cfg = parse_env_cfg("AAURoverEnv-v0", num_envs=8)
    assert not from_reference_cfg(cfg).observation_post()                      # the stock table: nothing to finish in torch
    cfg.observations.policy.height_scan.noise = Unoise(n_min=-0.02, n_max=0.02)
    cfg.observations.policy.height_scan.clip = (-1.0, 1.0)
    cfg.observations.policy.distance.clip = (0.0, 50.0)
    out = from_reference_cfg(cfg)
    post = out.observation_post()
    assert list(post) == ["distance", "height_scan"] and post["height_scan"].clip == (-1.0, 1.0)
    nz = post["height_scan"].noise
    x = torch.zeros(1000)
    y = nz.func(x, nz)                                                          # ORBIT's call convention
    assert float(y.min()) >= -0.02 and float(y.max()) <= 0.02 and float(y.std()) > 0.005
    n = out.to_native()
    assert n.obs_scale_distance == 1.0 and n.obs_scale_heading == pytest.approx(1 / math.pi)   # distance is finished in torch: raw from the kernel
    cfg.observations.policy.enable_corruption = False
    post = from_reference_cfg(cfg).observation_post()
    assert post["height_scan"].noise is None and post["height_scan"].clip == (-1.0, 1.0)


def test_configclass_semantics():
    from isaac_rover_orbit_amd.compat.orbit_shim import MISSING, configclass

    @configclass
    class A:
        x: int = 1
        items: list = [1, 2]
        name = "a"
        req: float = MISSING

    @configclass
    class B(A):
        y = 5

        def __post_init__(self):
            self.y = self.y * 2

    a1, a2 = A(), A(x=3)
    a1.items.append(9)
    assert a2.items == [1, 2] and a2.x == 3 and a1.name == "a" and a1.req is MISSING      # mutable defaults are per instance
    b = B(req=1.5)
    assert b.y == 10 and b.replace(x=7).x == 7 and b.x == 1 and b.to_dict()["items"] == [1, 2]
    with pytest.raises(TypeError):
        A(bogus=1)
