"""skrl.envs.wrappers.torch: Wrapper + the "isaac-orbit" wrapper (skrl 1.1.0 semantics): the first reset()
resets the env, later calls return the cached observation; step() returns obs["policy"] and reward / terminated / truncated
viewed as (N, 1)."""
import torch


class Wrapper:
    def __init__(self, env):
        self._env = env
        try:
            self._unwrapped = env.unwrapped
        except AttributeError:
            self._unwrapped = env
        dev = getattr(self._unwrapped, "device", "cpu")
        self.device = torch.device(dev)

    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        if hasattr(self._env, key):
            return getattr(self._env, key)
        if hasattr(self._unwrapped, key):
            return getattr(self._unwrapped, key)
        raise AttributeError(f"Wrapped environment ({type(self._unwrapped).__name__}) does not have attribute '{key}'")

    @property
    def num_envs(self):
        return self._unwrapped.num_envs if hasattr(self._unwrapped, "num_envs") else 1

    @property
    def num_agents(self):
        return 1

    @property
    def state_space(self):
        return getattr(self._unwrapped, "single_observation_space", self._env.observation_space)

    @property
    def observation_space(self):
        return self._env.observation_space

    @property
    def action_space(self):
        return self._env.action_space

    def close(self):
        self._env.close()


class IsaacOrbitWrapper(Wrapper):
    def __init__(self, env):
        super().__init__(env)
        self._reset_once = True
        self._obs_dict = None
        self._info = {}

    @property
    def observation_space(self):
        return self._unwrapped.single_observation_space["policy"]

    @property
    def action_space(self):
        return self._unwrapped.single_action_space

    def step(self, actions):
        self._obs_dict, reward, terminated, truncated, self._info = self._env.step(actions)
        return self._obs_dict["policy"], reward.view(-1, 1), terminated.view(-1, 1), truncated.view(-1, 1), self._info

    def reset(self):
        if self._reset_once:
            self._obs_dict, self._info = self._env.reset()
            self._reset_once = False
        return self._obs_dict["policy"], self._info

    def render(self, *args, **kwargs):
        return None


def wrap_env(env, wrapper="auto", verbose=True):
    if wrapper in ("isaac-orbit", "auto"):
        return IsaacOrbitWrapper(env)
    raise ValueError(f"the skrl test double only knows the 'isaac-orbit' wrapper, got {wrapper!r}")
