"""Shared terrain data of the rover environment (init-time, host side).

The reference builds, once per process, a 0.05 m heightmap of the hidden merged terrain mesh, a rock mask, a
"safe" (dilated) rock mask and a seeded spawn table (``rover_envs/envs/navigation/utils/terrains/terrain_utils.py``:
``HeightmapManager`` :14-87, ``TerrainManager.find_rocks_in_heightmap`` :265-311, ``random_rover_spawns`` :330-385).
Its three terrain USDs (ground / obstacles / merged, ``assets/terrains/debug/debug_terrains.py:17-52``) are not
shipped, so the data is synthesised here with the same three-layer contract:

* ``ground``    -- collision ground only                      (``terrain_only.usd``)
* ``obstacle``  -- height of the rocks above the ground, 0 elsewhere (``rocks_merged.usd``; contact-sensor filter)
* ``height``    -- ground + obstacle, the surface the wheels touch and the ray-caster sees (``terrain_merged.usd``)

Everything in this module is numpy / scipy and runs once at start-up; the per-step consumers are the HIP kernels.
``backend="hip"`` hands the two heavy producers (mesh rasterisation, rock masks) to the device kernels of
``terrain_hip`` instead (bit-identical results).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

Q16_SCALE = 2.0 ** -13     # height quantum (0.122 mm) of the procedural terrains: exact int16 copy for |h| < 4 m
Q16_SCALES = tuple(2.0 ** -e for e in (13, 12, 11, 10, 9, 8))   # quanta tried for an exact int16 copy (+-4 m ... +-128 m)
RESOLUTION = 0.05          # terrain_utils.py:108
GRADIENT_THRESHOLD = 0.3   # terrain_utils.py:109
SPAWN_SEED = 41            # terrain_utils.py:124
SPAWN_BORDER_M = 20.0      # terrain_utils.py:334


# ----------------------------------------------------------------------------------------------- mesh -> heightmap
def mesh_bounding_boxes(vertices: np.ndarray, faces: np.ndarray, resolution: float = RESOLUTION):
    """Grid extent and per-triangle cell bounding boxes of ``mesh_to_heightmap`` (terrain_utils.py:26-49), shared by the
    host loop below and the device rasteriser (``terrain_hip.mesh_to_heightmap``).

    Returns ``(shape, (min_x, min_y, max_x, max_y), bbox (F, 4) int32 {min_i, max_i, min_j, max_j}, zmax (F,) fp32)``.
    """
    vertices = np.asarray(vertices)
    border = 1.0
    min_x, min_y, _ = np.min(vertices, axis=0) + border
    max_x, max_y, _ = np.max(vertices, axis=0) - border
    grid_x = (max_x - min_x) / resolution
    grid_y = (max_y - min_y) / resolution
    # NB the reference allocates (int(grid_x + 1), int(grid_y + 1)) but indexes it [j (y), i (x)] (:35, :55)
    shape = (int(grid_x + 1), int(grid_y + 1))
    cell_x = (max_x - min_x) / grid_x
    cell_y = (max_y - min_y) / grid_y
    tri = vertices[np.asarray(faces).astype(np.int64)]                      # (F, 3, 3)
    lo, hi, zmax = tri.min(axis=1), tri.max(axis=1), tri[:, :, 2].max(axis=1)
    # python int() truncates toward zero, like the reference
    min_i = np.trunc((lo[:, 0] - min_x) / cell_x).astype(np.int64)
    max_i = np.minimum(np.trunc((hi[:, 0] - min_x) / cell_x).astype(np.int64), shape[1] - 1)
    min_j = np.trunc((lo[:, 1] - min_y) / cell_y).astype(np.int64)
    max_j = np.minimum(np.trunc((hi[:, 1] - min_y) / cell_y).astype(np.int64), shape[0] - 1)
    if tri.shape[0] and (min_i.min() < -shape[1] or min_j.min() < -shape[0]):
        raise IndexError("triangle bounding box outside the heightmap (numpy index would be out of bounds)")
    bbox = np.stack([min_i, max_i, min_j, max_j], axis=1).astype(np.int32)
    return shape, (min_x, min_y, max_x, max_y), bbox, zmax.astype(np.float32)


def mesh_to_heightmap(vertices: np.ndarray, faces: np.ndarray, resolution: float = RESOLUTION):
    """``HeightmapManager.mesh_to_heightmap`` (terrain_utils.py:23-57).

    1 m border trimmed, cell value = max vertex z over every triangle whose *bounding box* covers the cell
    (no true rasterisation), cells never covered stay at -99.  Returns ``(heightmap[y, x], min_x, min_y, max_x, max_y)``.
    """
    shape, (min_x, min_y, max_x, max_y), bbox, zmax = mesh_bounding_boxes(vertices, faces, resolution)
    hm = np.ones(shape, dtype=np.float32) * -99
    for f in range(bbox.shape[0]):
        i0, i1, j0, j1 = (int(v) for v in bbox[f])
        if i1 < i0 or j1 < j0:
            continue
        # python's range(min, max + 1) with a negative min wraps through negative indices in the reference
        ii = np.arange(i0, i1 + 1)
        jj = np.arange(j0, j1 + 1)
        sub = hm[np.ix_(jj, ii)]
        hm[np.ix_(jj, ii)] = np.maximum(sub, zmax[f])
    return hm, min_x, min_y, max_x, max_y


# ----------------------------------------------------------------------------------------------- rock masks
def mesh_node_boxes(vertices: np.ndarray, faces: np.ndarray, shape, min_x: float, min_y: float, resolution: float = RESOLUTION):
    """Per-triangle corners (F, 9) fp64 and the grid NODES inside each triangle's xy bounding box (F, 4) int32
    ``{min_i, max_i, min_j, max_j}`` (clamped to the grid; empty boxes have max < min) -- shared by the host loop below and the
    device kernel (``terrain_hip.mesh_surface_heights``)."""
    tri = np.asarray(vertices, dtype=np.float64)[np.asarray(faces).astype(np.int64)]       # (F, 3, 3)
    lo, hi = tri[:, :, :2].min(axis=1), tri[:, :, :2].max(axis=1)
    H, W = shape
    min_i = np.maximum(np.ceil((lo[:, 0] - min_x) / resolution - 1e-9), 0).astype(np.int64)
    max_i = np.minimum(np.floor((hi[:, 0] - min_x) / resolution + 1e-9), W - 1).astype(np.int64)
    min_j = np.maximum(np.ceil((lo[:, 1] - min_y) / resolution - 1e-9), 0).astype(np.int64)
    max_j = np.minimum(np.floor((hi[:, 1] - min_y) / resolution + 1e-9), H - 1).astype(np.int64)
    box = np.stack([min_i, max_i, min_j, max_j], axis=1).astype(np.int32)
    return np.ascontiguousarray(tri.reshape(-1, 9)), box


def mesh_surface_heights(vertices: np.ndarray, faces: np.ndarray, shape, min_x: float, min_y: float,
                         resolution: float = RESOLUTION) -> np.ndarray:
    """Height of the mesh at every grid node: z of the first hit of a vertical ray from above through
    ``(min_x + i * res, min_y + j * res)`` = max over the covering triangles of the triangle's plane there; -99 where no
    triangle covers the node.  This is the surface the reference's RayCaster (mesh ray-cast, rover_env_cfg.py:78-86) and
    PhysX see; ``mesh_to_heightmap`` above (bounding-box max) is only what the reference uses for target / spawn look-ups.
    Host loop over triangles (fine up to ~1e5 faces); ``terrain_hip.mesh_surface_heights`` is the device version (identical
    arithmetic, bit-identical result)."""
    tri, box = mesh_node_boxes(vertices, faces, shape, min_x, min_y, resolution)
    H, W = shape
    out = np.full((H, W), -99.0, dtype=np.float32)
    for f in range(tri.shape[0]):
        i0, i1, j0, j1 = (int(v) for v in box[f])
        if i1 < i0 or j1 < j0:
            continue
        ax, ay, az, bx, by, bz, cx, cy, cz = (float(v) for v in tri[f])
        den = (by - cy) * (ax - cx) + (cx - bx) * (ay - cy)
        if den == 0.0:
            continue
        px = (min_x + resolution * np.arange(i0, i1 + 1, dtype=np.float64))[None, :]
        py = (min_y + resolution * np.arange(j0, j1 + 1, dtype=np.float64))[:, None]
        w0 = ((by - cy) * (px - cx) + (cx - bx) * (py - cy)) / den
        w1 = ((cy - ay) * (px - cx) + (ax - cx) * (py - cy)) / den
        w2 = 1.0 - w0 - w1
        inside = (w0 >= -1e-9) & (w1 >= -1e-9) & (w2 >= -1e-9)
        if inside.any():
            z = (w0 * az + w1 * bz + w2 * cz).astype(np.float32)
            blk = out[j0:j1 + 1, i0:i1 + 1]
            np.maximum(blk, np.where(inside, z, np.float32(-99.0)), out=blk)
    return out


def _dilate(mask: np.ndarray, k: int) -> np.ndarray:
    """cv2.dilate with a k x k ones kernel, default anchor (k // 2) and default (ignored) border:
    dst(y, x) = max src(y + dy, x + dx), dy, dx in [-(k // 2), k - 1 - k // 2]."""
    from scipy import ndimage
    return ndimage.maximum_filter(mask, size=k, mode="constant", cval=0)


def _erode(mask: np.ndarray, k: int) -> np.ndarray:
    from scipy import ndimage
    return ndimage.minimum_filter(mask, size=k, mode="constant", cval=1)


def find_rocks_in_heightmap(heightmap: np.ndarray, threshold: float = GRADIENT_THRESHOLD):
    """``TerrainManager.find_rocks_in_heightmap`` (terrain_utils.py:265-311) without OpenCV.

    Sobel (wrap) magnitude > threshold -> close 3x3 -> fill holes -> open 7x7 -> dilate 11x11 (rock mask)
    -> dilate 42x42 (safe rock mask).  cv2 is absent from the build image; the morphology is restated with
    scipy.ndimage rank filters using cv2's anchor convention for the even 42 kernel (pinned by construction only).
    """
    from scipy import ndimage
    from scipy.signal import convolve2d

    sobel_x = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]])
    sobel_y = np.array([[-1, -2, -1], [0, 0, 0], [1, 2, 1]])
    grad_x = convolve2d(heightmap, sobel_x, mode="same", boundary="wrap")
    grad_y = convolve2d(heightmap, sobel_y, mode="same", boundary="wrap")
    mag = np.sqrt(grad_x ** 2 + grad_y ** 2)
    rock = np.zeros(heightmap.shape, dtype=np.uint8)
    rock[mag > threshold] = 1
    rock = _erode(_dilate(rock, 3), 3)                          # MORPH_CLOSE 3x3
    rock = ndimage.binary_fill_holes(rock).astype(np.uint8)
    rock = _dilate(_erode(rock, 7), 7)                          # MORPH_OPEN 7x7
    rock = _dilate(rock, 11)
    safe = _dilate(rock, 42)
    return rock.astype(np.uint8), safe.astype(np.uint8)


# ----------------------------------------------------------------------------------------------- spawn table
def random_rover_spawns(rock_mask: np.ndarray, heightmap: np.ndarray, n_spawns: int, min_x: float, min_y: float,
                        resolution: float = RESOLUTION, border_offset: float = SPAWN_BORDER_M, seed=SPAWN_SEED):
    """``TerrainManager.random_rover_spawns`` (terrain_utils.py:330-385): legacy numpy stream
    (``np.random.seed(seed)`` + two ``randint`` per try), rejection on the (safe) rock mask."""
    rng = np.random.RandomState(seed) if seed is not None else np.random
    height, width = rock_mask.shape
    min_xy = int(border_offset / resolution)
    max_xy = int(min(height, width) - min_xy)
    assert max_xy < width and max_xy < height
    assert max_xy > min_xy, "terrain smaller than twice the spawn border"
    if np.all(rock_mask[min_xy:max_xy, min_xy:max_xy] != 0):   # the reference would spin forever in the loop below
        raise ValueError("no rock-free cell inside the spawn border: cannot build a spawn table")
    out = np.zeros((n_spawns, 3), dtype=np.float32)
    for i in range(n_spawns):
        while True:
            x = rng.randint(min_xy, max_xy)
            y = rng.randint(min_xy, max_xy)
            if rock_mask[y, x] == 0:
                out[i] = (x, y, heightmap[y, x])
                break
    out[:, 0] = out[:, 0] * resolution + min_x
    out[:, 1] = out[:, 1] * resolution + min_y
    return out


# ----------------------------------------------------------------------------------------------- procedural terrain
def _value_noise(rng: np.random.RandomState, shape, cell: int) -> np.ndarray:
    from scipy import ndimage
    gh, gw = shape[0] // cell + 3, shape[1] // cell + 3
    lattice = rng.standard_normal((gh, gw)).astype(np.float32)
    up = ndimage.zoom(lattice, cell, order=3, mode="nearest", prefilter=True)
    return up[cell:cell + shape[0], cell:cell + shape[1]].astype(np.float32)


def fbm_heightfield(shape=(2048, 2048), seed=1234, sigma_z=0.15, base_cell=256, octaves=5, persistence=0.5):
    """Fractional-Brownian value noise: bicubic-upsampled Gaussian lattices, octave o has lattice
    spacing base_cell / 2**o cells (12.8 m at the default 0.05 m cell) and amplitude persistence**o."""
    rng = np.random.RandomState(seed)
    out = np.zeros(shape, dtype=np.float32)
    amp = 1.0
    for o in range(octaves):
        out += amp * _value_noise(rng, shape, max(base_cell >> o, 4))
        amp *= persistence
    out -= out.mean()
    out *= sigma_z / max(float(out.std()), 1e-9)
    return out.astype(np.float32)


def gaussian_rocks(shape, n_rocks=400, seed=1234, h_range=(0.25, 0.6), aspect_range=(1.0, 1.4), resolution=RESOLUTION,
                   border_m=5.0):
    """Obstacle layer: compactly supported Gaussian bumps of height h and footprint radius r = h * aspect
    (0.25 .. 0.84 m), sigma = r / 2.  The ranges keep every rock steeper than the reference's rock detector threshold
    (Sobel magnitude 0.3 <=> slope 0.75, terrain_utils.py:109,283) and wider than its 7x7 opening, so the obstacle layer and the rock mask
    describe the same set of stones, as the reference's rocks_merged.usd / gradient mask pair does."""
    rng = np.random.RandomState(seed + 1)
    H, W = shape
    out = np.zeros(shape, dtype=np.float32)
    b = border_m / resolution
    for _ in range(n_rocks):
        cx, cy = rng.uniform(b, W - b), rng.uniform(b, H - b)
        h = rng.uniform(*h_range)
        r = h * rng.uniform(*aspect_range) / resolution
        x0, x1 = int(max(cx - r, 0)), int(min(cx + r + 1, W))
        y0, y1 = int(max(cy - r, 0)), int(min(cy + r + 1, H))
        yy, xx = np.mgrid[y0:y1, x0:x1]
        d2 = (xx - cx) ** 2 + (yy - cy) ** 2
        sig2 = (0.5 * r) ** 2
        edge = np.exp(-0.5 * r * r / sig2)
        bump = h * (np.exp(-0.5 * d2 / sig2) - edge) / (1.0 - edge)
        bump[d2 > r * r] = 0.0
        out[y0:y1, x0:x1] = np.maximum(out[y0:y1, x0:x1], bump.astype(np.float32))
    return out


def quantize_heights(h: np.ndarray, scale: float = Q16_SCALE) -> np.ndarray:
    """Round heights to integer multiples of ``scale`` (exactly representable in fp32 AND in int16 for |h| < 4 m)."""
    return (np.rint(np.asarray(h, dtype=np.float64) / scale) * scale).astype(np.float32)


@dataclass
class Terrain:
    """Host copy of the shared, read-only terrain data every env (and every GPU rank) uses."""
    ground: np.ndarray
    obstacle: np.ndarray
    min_x: float = 0.0
    min_y: float = 0.0
    resolution: float = RESOLUTION
    rock_mask: np.ndarray | None = None
    safe_rock_mask: np.ndarray | None = None
    spawn_locations: np.ndarray | None = None
    backend: str = "numpy"   # who derives the rock masks: "numpy" (host, as the reference) or "hip" (device kernels)
    # heightmap of HeightmapManager.get_height_at (target z look-ups).  None = `height` (procedural terrains); terrains
    # ingested from a mesh carry the reference's bounding-box heightmap here next to the exact mesh surface in `height`
    lookup_height: np.ndarray | None = None
    height: np.ndarray = field(init=False)

    def __post_init__(self):
        self.ground = np.ascontiguousarray(self.ground, dtype=np.float32)
        self.obstacle = np.ascontiguousarray(self.obstacle, dtype=np.float32)
        assert self.ground.shape == self.obstacle.shape and self.ground.ndim == 2
        self.height = (self.ground + self.obstacle).astype(np.float32)
        if self.lookup_height is not None:
            self.lookup_height = np.ascontiguousarray(self.lookup_height, dtype=np.float32)
            assert self.lookup_height.shape == self.ground.shape
        if self.backend not in ("numpy", "hip"):
            raise ValueError(f"unknown terrain backend {self.backend!r}")
        if self.rock_mask is None or self.safe_rock_mask is None:
            # the reference derives the masks from ITS heightmap (terrain_utils.py:118-121)
            src = self.lookup_height if self.lookup_height is not None else self.height
            if self.backend == "hip":   # explicit choice, no fallback: raises without the HIP library / a GPU
                from . import terrain_hip
                rock, safe = terrain_hip.find_rocks_in_heightmap(src, GRADIENT_THRESHOLD)
                self.rock_mask, self.safe_rock_mask = rock.cpu().numpy(), safe.cpu().numpy()
            else:
                self.rock_mask, self.safe_rock_mask = find_rocks_in_heightmap(src, GRADIENT_THRESHOLD)

    @property
    def shape(self):
        return self.height.shape

    def height_q16(self, scales=None):
        """``(q, scale)`` with an int16 array q such that ``height == q * scale`` EXACTLY for a power-of-two ``scale``
        (2^-13 m ... 2^-8 m, i.e. a range of +-4 m ... +-128 m), or None when the terrain is not representable that way
        (arbitrary ingested meshes that were not snapped with ``quantize_heights``).  The ray-caster kernel stages this copy
        (half the bytes) when it exists; results are bit-identical either way."""
        h64 = self.height.astype(np.float64)
        for scale in (scales or Q16_SCALES):
            q = h64 / scale
            if np.abs(q).max() <= 32767 and np.all(q == np.rint(q)):
                q16 = q.astype(np.int16)
                assert np.array_equal(q16.astype(np.float32) * np.float32(scale), self.height)
                return q16, float(scale)
        return None

    def make_spawns(self, n_spawns: int, seed=SPAWN_SEED, border_offset=SPAWN_BORDER_M) -> np.ndarray:
        """Spawn table of the reference: ``n_spawns = 2 * num_envs`` (terrain_utils.py:123-124)."""
        # the reference reads the spawn heights off ITS heightmap (terrain_utils.py:122-124)
        hm = self.lookup_height if self.lookup_height is not None else self.height
        self.spawn_locations = random_rover_spawns(self.safe_rock_mask, hm, n_spawns, self.min_x, self.min_y,
                                                   self.resolution, border_offset, seed)
        return self.spawn_locations

    # host-side restatement of the two look-ups used by the command sampler (kept for user code / tests)
    def cell_of(self, xy: np.ndarray):
        """Index quirk of the reference (terrain_utils.py:75, 211-212): ``long(xy / res + [min_x, min_y])``."""
        xy = np.asarray(xy, dtype=np.float32)
        sx = (xy[:, 0] / np.float32(self.resolution) + np.float32(self.min_x)).astype(np.int64)
        sy = (xy[:, 1] / np.float32(self.resolution) + np.float32(self.min_y)).astype(np.int64)
        return np.clip(sx, 0, self.shape[1] - 1), np.clip(sy, 0, self.shape[0] - 1)

    def get_height_at(self, xy: np.ndarray) -> np.ndarray:
        cx, cy = self.cell_of(xy)
        return (self.lookup_height if self.lookup_height is not None else self.height)[cy, cx]

    def target_invalid(self, xy: np.ndarray) -> np.ndarray:
        cx, cy = self.cell_of(xy)
        return self.safe_rock_mask[cy, cx] == 1


def make_flat_terrain(shape=(2048, 2048), z=0.0) -> Terrain:
    """flat ground, no rocks."""
    g = np.full(shape, z, dtype=np.float32)
    zero = np.zeros(shape, dtype=np.uint8)
    return Terrain(ground=g, obstacle=np.zeros(shape, np.float32), rock_mask=zero, safe_rock_mask=zero.copy())


def make_procedural_terrain(shape=(2048, 2048), seed=1234, sigma_z=0.15, n_rocks=400, h_range=(0.25, 0.6),
                            aspect_range=(1.0, 1.4), quantize: bool = True, backend: str = "numpy") -> Terrain:
    """SURVEY 8d config 2 (and config 4 with sigma_z = 0.4): fBm ground + Gaussian rocks.  With ``quantize`` both layers
    are rounded to 2^-13 m (0.12 mm, far below the 0.05 m cell), which makes the terrain exactly representable in int16."""
    scale = min(shape) / 2048.0
    ground = fbm_heightfield(shape, seed=seed, sigma_z=sigma_z, base_cell=max(int(256 * scale), 8))
    rocks = gaussian_rocks(shape, n_rocks=n_rocks, seed=seed, h_range=h_range, aspect_range=aspect_range,
                           border_m=min(5.0, 0.1 * min(shape) * RESOLUTION))
    if quantize:
        ground, rocks = quantize_heights(ground), quantize_heights(rocks)
    return Terrain(ground=ground, obstacle=rocks, backend=backend)


def terrain_from_mesh(vertices, faces, ground_vertices=None, ground_faces=None, backend: str = "numpy",
                      quantize: float | None = None, surface: str = "mesh") -> Terrain:
    """Ingest triangle meshes the way ``TerrainManager.__init__`` does (terrain_utils.py:92-127) -- and keep what the
    reference's *simulation* sees next to what its *look-ups* see:

    * ``lookup_height`` = ``mesh_to_heightmap`` of the merged (hidden) mesh: the reference's bounding-box heightmap, used
      exactly where the reference uses it (rock masks, spawn heights, target z);
    * ``height`` (ground + obstacle) = the mesh surface itself sampled at the grid nodes (``surface="mesh"``, default): what
      the reference's ray-caster hits and its wheels stand on (rover_env_cfg.py:78-86, debug_terrains.py:17-52).  With
      ``surface="heightmap"`` the bounding-box heightmap doubles as the surface (the round-1 behaviour; over-estimates every
      slope by up to one triangle's height range).

    An optional ground-only mesh of the same extent (``terrain_only.usd``) gives the ground layer, and the obstacle layer is
    the positive difference of the two surfaces.  Without it the terrain has no obstacles.  ``quantize`` (a power of two, e.g.
    ``2 ** -12``) snaps the surface heights to that quantum -- at most half a quantum off the mesh -- which gives the terrain an
    exact int16 copy for the ray-caster kernel.
    """
    if surface not in ("mesh", "heightmap"):
        raise ValueError("surface must be 'mesh' or 'heightmap'")
    if backend == "hip":
        from . import terrain_hip

        def to_hm(v, f):
            h, x0, y0, x1, y1 = terrain_hip.mesh_to_heightmap(v, f)
            return h.cpu().numpy(), x0, y0, x1, y1

        def to_surface(v, f, shape, x0, y0):
            return terrain_hip.mesh_surface_heights(v, f, shape, x0, y0).cpu().numpy()
    else:
        to_hm, to_surface = mesh_to_heightmap, mesh_surface_heights

    def layer(v, f):
        hm, x0, y0, _, _ = to_hm(v, f)
        if surface == "heightmap":
            return hm, hm, x0, y0
        surf = to_surface(v, f, hm.shape, float(x0), float(y0))
        surf = np.where(surf <= -98.0, hm, surf).astype(np.float32)   # nodes no triangle covers: the look-up value
        return hm, surf, x0, y0

    hm, surf, min_x, min_y = layer(vertices, faces)
    if quantize is not None:
        surf = quantize_heights(surf, quantize)
    lookup = None if surface == "heightmap" else hm
    if ground_vertices is None:
        return Terrain(ground=surf, obstacle=np.zeros_like(surf), min_x=float(min_x), min_y=float(min_y), backend=backend,
                       lookup_height=lookup)
    _, gsurf, gx, gy = layer(ground_vertices, ground_faces)
    if quantize is not None:
        gsurf = quantize_heights(gsurf, quantize)
    if gsurf.shape != surf.shape or abs(gx - min_x) > 1e-6 or abs(gy - min_y) > 1e-6:
        raise ValueError("ground mesh and merged mesh must cover the same extent")
    obstacle = np.maximum(surf - gsurf, 0.0).astype(np.float32)
    obstacle[obstacle < 1e-3] = 0.0
    return Terrain(ground=surf - obstacle, obstacle=obstacle, min_x=float(min_x), min_y=float(min_y), backend=backend,
                   lookup_height=lookup)
