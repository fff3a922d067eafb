"""Episode traces in the reference recorder's layout (hdf_recorder.py:32-51 datasets + number_of_steps; .npz stand-in for HDF5)."""
import numpy as np
import pytest


def test_recorder_layout_and_rollover(tmp_path):
    from isaac_rover_orbit_amd.trace import EpisodeRecorder, load_trace
    n, od, ad = 5, 7, 2
    rng = np.random.RandomState(0)
    rec = EpisodeRecorder(str(tmp_path / "run"), n, od, ad, extras={"pos": {"shape": (3,), "dtype": np.float32}}, max_rows=40,
                          backend="npz")
    lengths = {e: 0 for e in range(n)}
    written, step_rows = [], []
    for t in range(30):
        obs, act = rng.rand(n, od).astype(np.float32), rng.rand(n, ad).astype(np.float32)
        rew, done = rng.rand(n).astype(np.float32), (rng.rand(n) < 0.15)
        rec.append_to_buffer(obs, act, rew, done, {"pos": rng.rand(n, 3).astype(np.float32)})
        for e in range(n):
            lengths[e] += 1
            step_rows.append((e, obs[e], rew[e]))
            if done[e]:
                written.append(lengths[e])
                lengths[e] = 0
    files = rec.close()
    total = sum(load_trace(f)["number_of_steps"] for f in files)
    assert total == 30 * n and len(files) >= 3                      # 150 rows, at most 40 per file
    for f in files:
        d = load_trace(f)
        k = d["number_of_steps"]
        assert 0 < k <= 40
        assert d["observations"].shape == (k, od) and d["actions"].shape == (k, ad) and d["pos"].shape == (k, 3)
        assert d["rewards"].shape == (k, 1) and d["rewards"].dtype == np.float32
        assert d["terminated"].shape == (k, 1) and d["terminated"].dtype == np.bool_
    # episodes are contiguous blocks that end with terminated = True (except the final flush)
    d0 = load_trace(files[0])
    ends = np.nonzero(d0["terminated"][:, 0])[0]
    assert len(ends) >= 1 and ends[0] + 1 == written[0]
    with pytest.raises(ValueError):
        EpisodeRecorder(str(tmp_path / "x.h5"), 1, 1, 1)


def test_recorder_rejects_an_episode_longer_than_a_file(tmp_path):
    from isaac_rover_orbit_amd.trace import EpisodeRecorder
    rec = EpisodeRecorder(str(tmp_path / "tiny"), 1, 2, 1, max_rows=5, backend="npz")
    for t in range(6):
        rec.append_to_buffer(np.zeros((1, 2), np.float32), np.zeros((1, 1), np.float32), np.zeros(1, np.float32), np.array([False]))
    with pytest.raises(ValueError, match="max_rows"):
        rec.append_to_buffer(np.zeros((1, 2), np.float32), np.zeros((1, 1), np.float32), np.zeros(1, np.float32), np.array([True]))


def test_recorder_hdf5_branch(tmp_path):
    """The reference's on-disk format proper (hdf_recorder.py:32-51); runs where h5py is installed."""
    h5py = pytest.importorskip("h5py")
    from isaac_rover_orbit_amd.trace import EpisodeRecorder, load_trace
    rec = EpisodeRecorder(str(tmp_path / "run"), 2, 3, 1, max_rows=16, backend="h5")
    for t in range(10):
        rec.append_to_buffer(np.full((2, 3), t, np.float32), np.zeros((2, 1), np.float32), np.ones(2, np.float32),
                             np.array([t % 4 == 3, t == 9]))
    files = rec.close()
    assert all(f.endswith(".h5") for f in files)
    with h5py.File(files[0], "r") as f:
        assert set(f.keys()) >= {"observations", "actions", "rewards", "terminated"} and f.attrs["number_of_steps"] > 0
    assert sum(load_trace(f)["number_of_steps"] for f in files) == 20
