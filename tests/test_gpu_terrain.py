"""GPU parity of the device-side terrain ingestion against the host restatement in terrain.py (numpy /
scipy, itself pinned by the reference's heightmap golden vector in tests/test_terrain.py).  Byte masks and the max-splat
heightmap are integer / order-independent work: the bar is BIT-EXACT."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grid_mesh(nx, ny, size_x, size_y, seed, bumps=6):
    """A regular triangle mesh of a bumpy surface (the shape the reference's terrain USDs have)."""
    rng = np.random.RandomState(seed)
    x = np.linspace(0.0, size_x, nx)
    y = np.linspace(0.0, size_y, ny)
    X, Y = np.meshgrid(x, y)
    Z = 0.1 * np.sin(X * 0.7) * np.cos(Y * 0.5)
    for _ in range(bumps):
        cx, cy, r, h = rng.uniform(3, size_x - 3), rng.uniform(3, size_y - 3), rng.uniform(0.4, 0.9), rng.uniform(0.3, 0.7)
        Z += h * np.exp(-((X - cx) ** 2 + (Y - cy) ** 2) / (2 * (r / 2) ** 2))
    verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float32)
    idx = np.arange(nx * ny).reshape(ny, nx)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    faces = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)])
    return verts, faces


def test_mesh_to_heightmap_matches_host():
    from isaac_rover_orbit_amd import terrain as T, terrain_hip as TH
    verts, faces = _grid_mesh(301, 241, 30.0, 24.0, seed=3)
    ref, *bounds = T.mesh_to_heightmap(verts, faces)
    hm, *bounds_d = TH.mesh_to_heightmap(verts, faces)
    assert tuple(hm.shape) == ref.shape and bounds == bounds_d
    assert np.array_equal(hm.cpu().numpy(), ref)
    assert (ref > -99).mean() > 0.5            # the mesh really covers the grid (the [j, i] quirk leaves a strip at -99)
    # sparse triangles, exact zeros of both signs (-0.0 must still replace the -99 initial value)
    verts[:, 2] = np.where(np.abs(verts[:, 2]) < 0.02, np.float32(-0.0), verts[:, 2])
    sub = faces[::7]
    ref, *_ = T.mesh_to_heightmap(verts, sub)
    hm, *_ = TH.mesh_to_heightmap(verts, sub)
    assert np.array_equal(hm.cpu().numpy(), ref) and (ref == -99).any() and (ref == 0).any()


def test_mesh_surface_matches_host_and_feeds_terrain_from_mesh():
    """rover_terrain_surface (the mesh's height at the grid nodes) is bit-identical to terrain.mesh_surface_heights, also
    through terrain_from_mesh(backend="hip"), which keeps the reference's bounding-box heightmap as the look-up layer."""
    from isaac_rover_orbit_amd import terrain as T, terrain_hip as TH
    verts, faces = _grid_mesh(141, 117, 14.0, 11.6, seed=9)
    verts[:, 0] += 0.013
    verts[:, 1] -= 0.021                                    # vertices off the 0.05 m grid
    hm, x0, y0, _, _ = T.mesh_to_heightmap(verts, faces)
    ref = T.mesh_surface_heights(verts, faces, hm.shape, float(x0), float(y0))
    dev = TH.mesh_surface_heights(verts, faces, hm.shape, float(x0), float(y0)).cpu().numpy()
    assert np.array_equal(dev, ref)
    assert ((ref > -99) & (ref <= hm + 1e-6)).mean() > 0.5
    a = T.terrain_from_mesh(verts, faces)
    b = T.terrain_from_mesh(verts, faces, backend="hip")
    assert np.array_equal(a.height, b.height) and np.array_equal(a.lookup_height, b.lookup_height)
    assert np.array_equal(a.safe_rock_mask, b.safe_rock_mask)


@pytest.mark.parametrize("name", ["quad", "wavy"])
def test_mesh_to_heightmap_golden(golden_dir, name):
    """The reference's own mesh_to_heightmap outputs (tools/gen_golden.py) through the device rasteriser."""
    from isaac_rover_orbit_amd import terrain_hip as TH
    g = np.load(f"{golden_dir}/heightmap.npz")
    hm, min_x, min_y, max_x, max_y = TH.mesh_to_heightmap(g[f"{name}_vertices"], g[f"{name}_faces"])
    assert np.array_equal(hm.cpu().numpy(), g[f"{name}_heightmap"])
    assert np.array_equal(np.array([min_x, min_y, max_x, max_y], dtype=np.float64), g[f"{name}_bounds"])


@pytest.mark.parametrize("shape,seed", [((1024, 1024), 5), ((600, 811), 11)])
def test_rock_masks_match_host(shape, seed):
    from isaac_rover_orbit_amd import terrain as T, terrain_hip as TH
    ter = T.make_procedural_terrain(shape, seed=seed, n_rocks=60 if shape[0] > 700 else 30)
    assert ter.rock_mask.sum() > 0 and ter.safe_rock_mask.sum() > ter.rock_mask.sum()
    rock, safe = TH.find_rocks_in_heightmap(ter.height)
    assert np.array_equal(rock.cpu().numpy(), ter.rock_mask)
    assert np.array_equal(safe.cpu().numpy(), ter.safe_rock_mask)


def test_rock_masks_with_enclosed_holes_and_border_rocks():
    """A ring-shaped ridge (its inside is a hole that binary_fill_holes must fill), a spiral wall (many propagation
    rounds) and a rock cut by the map border (cv2 border handling)."""
    from isaac_rover_orbit_amd import terrain as T, terrain_hip as TH
    H = W = 512
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    h = np.zeros((H, W), np.float32)
    r = np.hypot(xx - 150, yy - 150)
    h += 0.8 * np.exp(-((r - 40) / 4.0) ** 2)                      # ring, radius 40 cells
    t = np.arctan2(yy - 350, xx - 350)
    rs = np.hypot(xx - 350, yy - 350)
    h += 0.8 * np.exp(-(((rs - 12 * (t + np.pi) - 20) % 75.4) / 4.0) ** 2) * (rs < 110)   # spiral wall
    h += 0.9 * np.exp(-((xx - 2) ** 2 + (yy - 256) ** 2) / 200.0)   # rock on the left border
    rock_ref, safe_ref = T.find_rocks_in_heightmap(h)
    assert rock_ref[150, 150] == 1                                  # the inside of the ring was filled
    rock, safe = TH.find_rocks_in_heightmap(h)
    assert np.array_equal(rock.cpu().numpy(), rock_ref)
    assert np.array_equal(safe.cpu().numpy(), safe_ref)


def test_terrain_backend_hip_equals_numpy_and_feeds_the_env():
    from isaac_rover_orbit_amd import terrain as T
    a = T.make_procedural_terrain((1024, 1024), seed=7, n_rocks=50)
    b = T.make_procedural_terrain((1024, 1024), seed=7, n_rocks=50, backend="hip")
    assert np.array_equal(a.rock_mask, b.rock_mask) and np.array_equal(a.safe_rock_mask, b.safe_rock_mask)
    assert np.array_equal(a.make_spawns(64), b.make_spawns(64))
    verts, faces = _grid_mesh(201, 201, 40.0, 40.0, seed=1, bumps=10)
    ta, tb = T.terrain_from_mesh(verts, faces), T.terrain_from_mesh(verts, faces, backend="hip")
    assert np.array_equal(ta.height, tb.height) and np.array_equal(ta.safe_rock_mask, tb.safe_rock_mask)


def test_argument_errors_are_reported():
    from isaac_rover_orbit_amd import _lib
    lib = _lib.load()
    assert lib.rover_terrain_rock_mask(None, 10, 10, 0.3, None, None, None, None) != 0
    assert b"NULL" in lib.rover_last_error()
    assert lib.rover_terrain_scratch_bytes(0, 5) == 0


def test_mesh_surface_with_huge_triangles():
    """A ground plane made of two triangles that cover the whole 1500 x 1300 node grid (one lane used to walk all of its
    nodes) plus a small pyramid on top: still bit-identical to the host restatement, and fast (a wave per triangle)."""
    import time
    from isaac_rover_orbit_amd import terrain as T, terrain_hip as TH
    X, Y = 75.0, 65.0
    verts = np.array([[0, 0, 0.1], [X, 0, 0.3], [0, Y, -0.2], [X, Y, 0.0],
                      [30.0, 30.0, 0.0], [31.0, 30.0, 0.0], [30.5, 31.0, 0.0], [30.5, 30.4, 0.8]], np.float64)
    faces = np.array([[0, 1, 2], [1, 3, 2], [4, 5, 7], [5, 6, 7], [6, 4, 7]], np.int64)
    shape = (1301, 1501)
    ref = T.mesh_surface_heights(verts, faces, shape, 0.0, 0.0)
    TH.mesh_surface_heights(verts, faces, shape, 0.0, 0.0)          # warm-up (module load)
    t0 = time.perf_counter()
    dev = TH.mesh_surface_heights(verts, faces, shape, 0.0, 0.0).cpu().numpy()
    dt = time.perf_counter() - t0
    assert np.array_equal(dev, ref) and (ref > -99).all() and ref.max() > 0.7
    assert dt < 0.5, f"surface kernel took {dt:.3f} s for two grid-sized triangles"
