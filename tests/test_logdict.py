"""``extras["log"]`` reduced on demand (``envs/_logdict.py``): every read accessor lets the env run its pending reduction first,
key iteration does not, and copies / pickles are plain dicts that do not drag the env along.  Host only."""
import copy
import pickle

import torch

from isaac_rover_orbit_amd.envs._logdict import LogDict


class _FakeEnv:
    def __init__(self):
        self.flushes = 0
        self.vec = torch.zeros(3)

    def flush_log(self):
        self.flushes += 1
        self.vec += 1.0           # what a reduction does: refresh the vector the dict's tensors are views of


def test_reads_flush_and_key_iteration_does_not():
    env = _FakeEnv()
    d = LogDict(env, {"Episode Reward/a": env.vec[0], "Episode Termination/b": env.vec[1]})
    assert list(d) == ["Episode Reward/a", "Episode Termination/b"] and list(d.keys()) == list(d) and len(d) == 2 and env.flushes == 0
    assert float(d["Episode Reward/a"]) == 1.0 and env.flushes == 1                  # the view shows the refreshed value
    assert [float(v) for v in d.values()] == [2.0, 2.0] and env.flushes == 2
    assert {k: float(v) for k, v in d.items()} == {"Episode Reward/a": 3.0, "Episode Termination/b": 3.0}
    assert float(d.get("Episode Termination/b")) == 4.0 and d.get("missing") is None
    assert isinstance(d, dict)


def test_copies_are_plain_dicts_without_the_env():
    env = _FakeEnv()
    d = LogDict(env, {"a": env.vec[0]})
    for c in (copy.copy(d), d.copy(), copy.deepcopy(d), pickle.loads(pickle.dumps(d))):
        assert type(c) is dict and set(c) == {"a"}
    deep = copy.deepcopy(d)
    before = float(deep["a"])
    env.flush_log()
    assert float(deep["a"]) == before                                                # a snapshot, not a view
