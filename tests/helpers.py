"""Shared helpers of the test-suite: small terrains and oracle <-> env plumbing."""
from __future__ import annotations

import functools

import numpy as np

from isaac_rover_orbit_amd import terrain as T


@functools.lru_cache(maxsize=4)
def small_procedural(size=1024, seed=7, sigma_z=0.15, n_rocks=120):
    """A 51.2 m procedural terrain (spawn band 20 m .. 31.2 m) -- generated once per session."""
    return T.make_procedural_terrain((size, size), seed=seed, sigma_z=sigma_z, n_rocks=n_rocks)


@functools.lru_cache(maxsize=2)
def flat(size=1024):
    return T.make_flat_terrain((size, size))


def oracle_terrain(ro, ter, n_spawns=None, spawn_seed=41):
    if n_spawns is not None and (ter.spawn_locations is None or ter.spawn_locations.shape[0] != n_spawns):
        ter.make_spawns(n_spawns, seed=spawn_seed)
    return ro.TerrainData(ter.height, ter.obstacle, ter.safe_rock_mask, ter.resolution, ter.min_x, ter.min_y,
                          ter.spawn_locations)


def oracle_config_from(ro, native):
    """Copy a ``_lib.RoverConfig`` (product side) field by field into the oracle's Config struct."""
    cfg = ro.Config()
    for name, _ in ro.Config._fields_:
        v = getattr(native, name)
        if name == "rew_weight":
            for i in range(7):
                cfg.rew_weight[i] = v[i]
        else:
            setattr(cfg, name, v)
    return cfg


def assert_close(a, b, atol, rtol, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    fin = np.isfinite(a) & np.isfinite(b)
    assert (np.isfinite(a) == np.isfinite(b)).all(), f"{what}: non-finite pattern differs"
    assert (a[~fin] == b[~fin]).all(), f"{what}: infinities differ"
    err = np.abs(a[fin] - b[fin])
    tol = atol + rtol * np.abs(b[fin])
    if not (err <= tol).all():
        i = int(np.argmax(err - tol))
        raise AssertionError(f"{what}: max |err| {err.max():.3e} (worst excess at flat idx {i}: {a[fin][i]} vs {b[fin][i]})")
