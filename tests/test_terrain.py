"""Init-time terrain producers against the reference's outputs (tests/golden/heightmap.npz)."""
import os

import numpy as np

from isaac_rover_orbit_amd import terrain as T


def test_mesh_to_heightmap_golden(golden_dir):
    h = np.load(f"{golden_dir}/heightmap.npz")
    hm, mnx, mny, mxx, mxy = T.mesh_to_heightmap(h["quad_vertices"], h["quad_faces"])
    assert hm.shape == (41, 41) and (hm == 2.0).all() and mnx == 1.0 and mny == 1.0
    assert np.array_equal(hm, h["quad_heightmap"])
    hm, mnx, mny, mxx, mxy = T.mesh_to_heightmap(h["wavy_vertices"], h["wavy_faces"])
    assert np.array_equal(hm, h["wavy_heightmap"])
    assert np.allclose([mnx, mny, mxx, mxy], h["wavy_bounds"])


def test_spawn_table_golden(golden_dir):
    h = np.load(f"{golden_dir}/heightmap.npz")
    rng = np.random.RandomState(int(h["spawn_seed_hm"]))
    H, W = h["spawn_shape"]
    hm = rng.rand(H, W).astype(np.float32)
    mask = (rng.rand(H, W) < 0.4).astype(np.uint8)
    sp = T.random_rover_spawns(mask, hm, 64, float(h["spawn_min_xy"][0]), float(h["spawn_min_xy"][1]))
    assert np.array_equal(sp, h["spawn_locations"])


def test_lookup_quirk_golden(golden_dir):
    """Terrain.get_height_at / target_invalid reproduce the reference's `xy / res + min` indexing (B-1)."""
    h = np.load(f"{golden_dir}/heightmap.npz")
    b = h["wavy_bounds"]
    z = np.zeros_like(h["wavy_heightmap"])
    t = T.Terrain(ground=h["wavy_heightmap"], obstacle=z, min_x=float(b[0]), min_y=float(b[1]),
                  rock_mask=h["wavy_mask"], safe_rock_mask=h["wavy_mask"])
    assert np.array_equal(t.get_height_at(h["wavy_query_xy"]), h["wavy_query_height"])
    assert np.array_equal(t.target_invalid(h["wavy_query_xy"]).astype(np.uint8), h["wavy_query_invalid"])


def test_rock_mask_pipeline():
    """Sobel -> close -> fill -> open -> dilate 11 -> dilate 42 (terrain_utils.py:265-311), cv2 anchor convention."""
    hm = np.zeros((200, 220), np.float32)
    hm[90:110, 100:120] = 0.5                       # a 1 m x 1 m, 0.5 m tall block
    hm[30, 30] = 1.0                                # a single-cell spike: removed by the 7x7 opening
    rock, safe = T.find_rocks_in_heightmap(hm, 0.3)
    assert rock[100, 110] == 1 and rock[30, 30] == 0
    ys, xs = np.nonzero(rock)
    # gradient ring (1 cell around the block) closed + filled, then dilated by 11 (5 cells each side)
    assert ys.min() == 90 - 1 - 5 and ys.max() == 109 + 1 + 5 and xs.min() == 100 - 1 - 5 and xs.max() == 119 + 1 + 5
    ys2, xs2 = np.nonzero(safe)
    # even kernel 42: cv2 anchor = 21 -> offsets [-21, +20]: the mask grows 20 cells toward -y/-x and 21 toward +y/+x
    assert ys2.min() == ys.min() - 20 and ys2.max() == ys.max() + 21
    assert xs2.min() == xs.min() - 20 and xs2.max() == xs.max() + 21
    assert set(np.unique(rock)) <= {0, 1} and rock.dtype == np.uint8


def test_procedural_terrain_contract():
    t = T.make_procedural_terrain((1024, 1024), seed=3, n_rocks=60)
    assert t.height.dtype == np.float32 and t.height.shape == (1024, 1024)
    assert np.array_equal(t.height, t.ground + t.obstacle)
    assert (t.obstacle >= 0).all() and t.obstacle.max() > 0.15
    assert abs(float(t.ground.std()) - 0.15) < 1e-3
    # every steep rock is inside the rock mask; every rock cell is inside the safe mask
    assert ((t.obstacle > 1e-3) & (t.safe_rock_mask == 0)).sum() == 0
    sp = t.make_spawns(128)
    assert sp.shape == (128, 3) and sp[:, :2].min() >= 20.0 and sp[:, :2].max() < 51.2 - 20.0 + 1e-3
    cx = np.rint((sp[:, 0] - t.min_x) / 0.05).astype(int)
    cy = np.rint((sp[:, 1] - t.min_y) / 0.05).astype(int)
    assert (t.safe_rock_mask[cy, cx] == 0).all()
    assert np.allclose(sp[:, 2], t.height[cy, cx])
    t2 = T.make_procedural_terrain((1024, 1024), seed=3, n_rocks=60)
    assert np.array_equal(t.height, t2.height)                      # deterministic


def test_terrain_from_mesh_roundtrip(golden_dir):
    """A mesh-ingested terrain carries the reference's bounding-box heightmap for the look-ups (golden fixture) and the
    mesh surface itself for the wheels / ray-caster; surface="heightmap" reproduces the one-layer round-1 behaviour."""
    h = np.load(f"{golden_dir}/heightmap.npz")
    t = T.terrain_from_mesh(h["wavy_vertices"], h["wavy_faces"])
    assert np.array_equal(t.lookup_height, h["wavy_heightmap"]) and (t.obstacle == 0).all()
    assert np.array_equal(t.get_height_at(h["wavy_query_xy"]), h["wavy_query_height"])        # reference look-ups unchanged
    assert (t.height <= t.lookup_height + 1e-6).all(), "bounding-box max can only over-estimate the mesh"
    assert (t.lookup_height - t.height).max() > 0.01
    one = T.terrain_from_mesh(h["wavy_vertices"], h["wavy_faces"], surface="heightmap")
    assert np.array_equal(one.height, h["wavy_heightmap"]) and one.lookup_height is None


def test_mesh_surface_is_the_exact_vertical_ray_hit(golden_dir):
    """mesh_surface_heights == first hit of a vertical ray through every grid node (the reference's RayCaster semantics,
    rover_env_cfg.py:78-86), checked against the exact float64 mesh ray-cast of oracle/mesh_raycast.py."""
    from oracle import mesh_raycast as mr
    h = np.load(f"{golden_dir}/heightmap.npz")
    V, F = h["wavy_vertices"], h["wavy_faces"]
    t = T.terrain_from_mesh(V, F)
    H, W = t.shape
    X, Y = np.meshgrid(t.min_x + 0.05 * np.arange(W), t.min_y + 0.05 * np.arange(H), indexing="xy")
    z = mr.vertical_ray_hits(V, F, np.stack([X.ravel(), Y.ravel()], 1)).reshape(H, W)
    hit = np.isfinite(z)
    assert hit.mean() > 0.99
    assert np.abs(t.height[hit] - z[hit]).max() < 1e-6
    # and between the nodes the triangle surface of the heightfield stays close to the mesh (0.25 m triangles, smooth sheet)
    rng = np.random.RandomState(0)
    q = np.stack([rng.uniform(t.min_x + 0.1, t.min_x + 0.05 * (W - 3), 4000), rng.uniform(t.min_y + 0.1, t.min_y + 0.05 * (H - 3), 4000)], 1)
    zq = mr.vertical_ray_hits(V, F, q)
    hv, hf = mr.heightfield_mesh(t.height, 0.05, t.min_x, t.min_y)
    zs = mr.vertical_ray_hits(hv, hf, q)
    zl = mr.vertical_ray_hits(*mr.heightfield_mesh(t.lookup_height, 0.05, t.min_x, t.min_y), q)
    assert np.abs(zs - zq).max() < 2e-3, "node-sampled surface follows the mesh"
    assert np.abs(zl - zq).max() > 0.05, "the bounding-box heightmap does not (it is a look-up table, not a surface)"


def test_exact_int16_copy_and_mesh_quantisation():
    """height_q16 finds the finest power-of-two quantum that represents the terrain exactly (or none); terrain_from_mesh
    can snap an ingested mesh to such a quantum (deviation <= half a quantum)."""
    from isaac_rover_orbit_amd import terrain as T
    ter = T.make_procedural_terrain((256, 256), seed=3, n_rocks=4)
    q, scale = ter.height_q16()
    assert scale == 2.0 ** -13 and q.dtype == np.int16 and np.array_equal(q.astype(np.float32) * np.float32(scale), ter.height)
    assert T.make_procedural_terrain((256, 256), seed=3, n_rocks=4, quantize=False).height_q16() is None
    tall = T.Terrain(ground=T.quantize_heights(ter.ground * 40.0, 2.0 ** -10), obstacle=np.zeros_like(ter.ground),
                     rock_mask=np.zeros(ter.shape, np.uint8), safe_rock_mask=np.zeros(ter.shape, np.uint8))
    assert np.abs(tall.height).max() > 16.0 and tall.height_q16()[1] == 2.0 ** -10
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "heightmap.npz"))
    raw = T.terrain_from_mesh(g["wavy_vertices"], g["wavy_faces"])
    snap = T.terrain_from_mesh(g["wavy_vertices"], g["wavy_faces"], quantize=2.0 ** -12)
    cov = raw.height > -99
    assert np.abs(snap.height[cov] - raw.height[cov]).max() <= 2.0 ** -13 and snap.height_q16() is not None
