"""``extras["log"]`` reduced on demand: a dict of 0-d device tensors whose read accessors first let the env run its pending
log reduction (``flush_log``), so that the values are the ones a per-step reduction would have left there."""
from __future__ import annotations


class LogDict(dict):
    """A ``dict`` (ORBIT's ``extras["log"]`` holds 0-d tensors under "Episode Reward/<term>" ... keys) bound to an env with a
    ``flush_log()`` method.  ``__iter__`` / ``keys`` are overridden as well: CPython's ``dict(d)``, ``{**d}``, ``other.update(d)``
    and ``d | x`` copy a dict subclass through the C fast path (stored values, no ``__getitem__``) unless the subclass overrides
    ``__iter__`` -- with the override they go through ``keys()`` + ``__getitem__`` and see flushed values."""

    def __init__(self, env, items):
        super().__init__(items)
        self._env = env

    def _rebind(self, items):
        """Swap the stored tensors (``RoverEnv.set_log_values``: device views <-> views of the pinned host mirror)."""
        super().clear()
        super().update(items)

    def __getitem__(self, k):
        self._env.flush_log()
        return super().__getitem__(k)

    def get(self, k, default=None):
        self._env.flush_log()
        return super().get(k, default)

    def __iter__(self):
        self._env.flush_log()
        return super().__iter__()

    def keys(self):
        self._env.flush_log()
        return super().keys()

    def __or__(self, other):
        self._env.flush_log()
        return dict(super().items()) | dict(other)

    def __ror__(self, other):
        self._env.flush_log()
        return dict(other) | dict(super().items())

    def items(self):
        self._env.flush_log()
        return super().items()

    def values(self):
        self._env.flush_log()
        return super().values()

    def copy(self):
        self._env.flush_log()
        return dict(super().items())

    # copies and pickles are plain dicts of the current values: they must not drag the env (a native handle) along
    def __copy__(self):
        return self.copy()

    def __deepcopy__(self, memo):
        self._env.flush_log()
        return {k: v.clone() for k, v in super().items()}

    def __reduce__(self):
        self._env.flush_log()
        return (dict, (dict(super().items()),))
