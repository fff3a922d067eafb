"""``extras["log"]`` reduced on demand (``envs/_logdict.py``): every read accessor lets the env run its pending reduction first,
key iteration and merges too, and copies / pickles are plain dicts that do not drag the env along.  Host only."""
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


def test_reads_flush():
    env = _FakeEnv()
    d = LogDict(env, {"Episode Reward/a": env.vec[0], "Episode Termination/b": env.vec[1]})
    assert len(d) == 2 and env.flushes == 0
    assert float(d["Episode Reward/a"]) == 1.0 and env.flushes == 1                  # the view shows the refreshed value
    assert [float(v) for v in d.values()] == [2.0, 2.0] and env.flushes == 2
    assert {k: float(v) for k, v in d.items()} == {"Episode Reward/a": 3.0, "Episode Termination/b": 3.0}
    assert float(d.get("Episode Termination/b")) == 4.0 and d.get("missing") is None
    assert isinstance(d, dict)
    # key iteration flushes too (the env's flush is a no-op when nothing is pending): that is what takes dict(d) & co. off
    # CPython's PyDict_Merge fast path, which would copy the stored tensors without any accessor
    n = env.flushes
    assert list(d) == ["Episode Reward/a", "Episode Termination/b"] and list(d.keys()) == list(d) and env.flushes > n


def test_merges_and_conversions_see_flushed_values():
    """``dict(d)``, ``{**d}``, ``other.update(d)``, ``d | x`` / ``x | d`` (the advisor's list): every one of them must run the
    pending reduction before the values are taken."""
    for make in (lambda d: dict(d), lambda d: {**d}, lambda d: (lambda o: (o.update(d), o)[1])({}), lambda d: d | {}, lambda d: {} | d):
        env = _FakeEnv()
        d = LogDict(env, {"a": env.vec[0], "b": env.vec[1]})
        c = make(d)
        assert env.flushes >= 1, "no flush before the values were copied"
        assert type(c) is dict and set(c) == {"a", "b"}
        assert float(c["a"]) == float(env.vec[0]) >= 1.0                              # the refreshed vector, not the stale zeros


def test_copies_are_plain_dicts_without_the_env():
    env = _FakeEnv()
    d = LogDict(env, {"a": env.vec[0]})
    for c in (copy.copy(d), d.copy(), copy.deepcopy(d), pickle.loads(pickle.dumps(d))):
        assert type(c) is dict and set(c) == {"a"}
    deep = copy.deepcopy(d)
    before = float(deep["a"])
    env.flush_log()
    assert float(deep["a"]) == before                                                # a snapshot, not a view
