"""On-disk episode traces in the reference's recorder layout (``rover_envs/utils/recorder/data_recorder/
base.py:7-87``, ``hdf_recorder.py:10-88``).

Per-env buffers collect ``(observation, action, reward, terminated [, extras...])`` rows; when an env's episode ends its rows
are appended to the current file.  File layout = ``HDF5DataRecorder``'s: datasets ``observations (rows, obs_dim)``,
``actions (rows, act_dim)``, ``rewards (rows, 1) float32``, ``terminated (rows, 1) bool`` (+ one dataset per extra), attribute
``number_of_steps``; a new file ``<base>_<index>`` is started when ``max_rows`` would be exceeded.  With ``h5py`` installed the
files are HDF5 (``.h5``, readable by the reference's tooling); without it (this environment) the same datasets go into a
``.npz`` with ``number_of_steps`` as a 0-d array.  Pure host-side I/O: nothing here is on the step() hot path.
"""
from __future__ import annotations

import os

import numpy as np

try:  # optional, as in the reference
    import h5py
except Exception:  # pragma: no cover
    h5py = None


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


class EpisodeRecorder:
    def __init__(self, base_filename: str, num_envs: int, obs_dim: int, act_dim: int, extras: dict | None = None,
                 max_rows: int = 500_000, backend: str = "auto"):
        if base_filename.endswith((".h5", ".hdf5", ".npz")):
            raise ValueError("base filename must not carry an extension (hdf_recorder.py:21)")
        if backend == "auto":
            backend = "h5" if h5py is not None else "npz"
        if backend == "h5" and h5py is None:
            raise RuntimeError("h5py is not installed; use backend='npz'")
        self.base_filename, self.num_envs, self.max_rows, self.backend = base_filename, int(num_envs), int(max_rows), backend
        self.obs_dim, self.act_dim = int(obs_dim), int(act_dim)
        self.extras = dict(extras or {})                     # name -> {"shape": tuple, "dtype": dtype}
        self.keys = ["observations", "actions", "rewards", "terminated"] + list(self.extras)
        self.buffers = [self._empty() for _ in range(self.num_envs)]
        self.current_row, self.file_index, self.files = 0, 0, []
        self._new_file()

    def _empty(self):
        return {k: [] for k in self.keys}

    def _spec(self):
        spec = {"observations": ((self.obs_dim,), np.float32), "actions": ((self.act_dim,), np.float32),
                "rewards": ((1,), np.float32), "terminated": ((1,), np.bool_)}
        for k, p in self.extras.items():
            spec[k] = (tuple(p["shape"]), np.dtype(p["dtype"]))
        return spec

    def _new_file(self):
        self.file_name = f"{self.base_filename}_{self.file_index}.{'h5' if self.backend == 'h5' else 'npz'}"
        self.file_index += 1
        self.files.append(self.file_name)
        self.current_row = 0
        if self.backend == "h5":
            with h5py.File(self.file_name, "w") as f:
                for k, (shape, dt) in self._spec().items():
                    f.create_dataset(k, (self.max_rows, *shape), dtype=dt, maxshape=(self.max_rows, *shape))
                f.attrs["number_of_steps"] = 0
        else:
            self._mem = {k: [] for k in self.keys}

    def append_to_buffer(self, obs, action, reward, done, info=None):
        """One env step of every env (base.py:37-58): rows are buffered per env and written when that env is done."""
        obs, action = _np(obs).reshape(self.num_envs, -1), _np(action).reshape(self.num_envs, -1)
        reward, done = _np(reward).reshape(self.num_envs, -1), _np(done).reshape(self.num_envs, -1).astype(bool)
        ext = {k: _np(info[k]) for k in self.extras} if self.extras else {}
        for e in range(self.num_envs):
            b = self.buffers[e]
            b["observations"].append(obs[e]); b["actions"].append(action[e])
            b["rewards"].append(reward[e, :1]); b["terminated"].append(done[e, :1])
            for k in self.extras:
                b[k].append(ext[k][e])
            if done[e].any():
                self.write_to_disk(e)

    def write_to_disk(self, env_id: int):
        b = self.buffers[env_id]
        n = len(b["observations"])
        if n == 0:
            return
        if n > self.max_rows:
            raise ValueError(f"an episode of {n} rows does not fit a file of max_rows = {self.max_rows}")
        if self.current_row + n > self.max_rows:
            self._close_file()
            self._new_file()
        chunk = {k: np.asarray(v) for k, v in b.items()}
        if self.backend == "h5":
            with h5py.File(self.file_name, "a") as f:
                for k, v in chunk.items():
                    f[k][self.current_row:self.current_row + n] = v
                f.attrs["number_of_steps"] += n
        else:
            for k, v in chunk.items():
                self._mem[k].append(v)
        self.current_row += n
        self.buffers[env_id] = self._empty()

    def flush(self):
        for e in range(self.num_envs):
            self.write_to_disk(e)

    def _close_file(self):
        if self.backend == "h5":
            with h5py.File(self.file_name, "a") as f:
                for k in self.keys:
                    f[k].resize(f.attrs["number_of_steps"], axis=0)
        else:
            spec = self._spec()
            out = {k: (np.concatenate(v, 0) if v else np.zeros((0, *spec[k][0]))).astype(spec[k][1]) for k, v in self._mem.items()}
            out["number_of_steps"] = np.int64(self.current_row)
            np.savez_compressed(self.file_name, **out)

    def close(self):
        self.flush()
        self._close_file()
        return list(self.files)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def load_trace(path: str) -> dict:
    """``{dataset: array, "number_of_steps": int}`` of one trace file (``.h5`` or ``.npz``)."""
    if path.endswith(".npz"):
        d = dict(np.load(path))
        d["number_of_steps"] = int(d["number_of_steps"])
        return d
    if h5py is None:
        raise RuntimeError("h5py is needed to read " + os.path.basename(path))
    with h5py.File(path, "r") as f:
        n = int(f.attrs["number_of_steps"])
        d = {k: f[k][:n] for k in f.keys()}
    d["number_of_steps"] = n
    return d
