"""Exact vertical ray-cast against a triangle mesh (numpy, float64).  TEST INFRASTRUCTURE ONLY.

What the reference's height scanner does (``RayCasterCfg`` with ``mesh_prim_paths=["/World/terrain/hidden_terrain"]``,
rays from +10 m straight down, ``rover_envs/envs/navigation/rover_env_cfg.py:78-86``; ORBIT evaluates it with Warp's
``mesh_query_ray``): the first hit from above = the highest point of the mesh on the vertical line through (x, y).
Used by tests / tools to measure how far the heightfield surfaces of the product path are from the source mesh.
"""
from __future__ import annotations

import numpy as np


def vertical_ray_hits(vertices: np.ndarray, faces: np.ndarray, xy: np.ndarray, chunk: int = 4096) -> np.ndarray:
    """z of the first hit of a downward vertical ray through each ``xy`` row; ``-inf`` where the ray misses the mesh."""
    V = np.asarray(vertices, dtype=np.float64)
    F = np.asarray(faces).astype(np.int64)
    Q = np.asarray(xy, dtype=np.float64)
    out = np.full(Q.shape[0], -np.inf)
    tri = V[F]                                             # (T, 3, 3)
    lo, hi = tri[:, :, :2].min(axis=1), tri[:, :, :2].max(axis=1)
    # bin the queries on a uniform grid of about the mean triangle size
    size = max(float(np.mean(hi - lo)), 1e-6) * 2.0
    q0 = Q.min(axis=0)
    qc = np.floor((Q - q0) / size).astype(np.int64)
    ncx = int(qc[:, 0].max()) + 1
    key = qc[:, 1] * ncx + qc[:, 0]
    order = np.argsort(key, kind="stable")
    skey = key[order]
    ncy = int(qc[:, 1].max()) + 1
    starts = np.searchsorted(skey, np.arange(ncx * ncy + 1))
    cl = np.clip(np.floor((lo - q0) / size).astype(np.int64), 0, [ncx - 1, ncy - 1])
    ch = np.clip(np.floor((hi - q0) / size).astype(np.int64), 0, [ncx - 1, ncy - 1])
    for t in range(tri.shape[0]):
        a, b, c = tri[t]
        den = (b[1] - c[1]) * (a[0] - c[0]) + (c[0] - b[0]) * (a[1] - c[1])
        if den == 0.0:
            continue
        idx = []
        for cy in range(cl[t, 1], ch[t, 1] + 1):
            k0, k1 = cy * ncx + cl[t, 0], cy * ncx + ch[t, 0]
            idx.append(order[starts[k0]:starts[k1 + 1]])
        idx = np.concatenate(idx) if idx else np.empty(0, np.int64)
        if idx.size == 0:
            continue
        p = Q[idx]
        w0 = ((b[1] - c[1]) * (p[:, 0] - c[0]) + (c[0] - b[0]) * (p[:, 1] - c[1])) / den
        w1 = ((c[1] - a[1]) * (p[:, 0] - c[0]) + (a[0] - c[0]) * (p[:, 1] - c[1])) / den
        w2 = 1.0 - w0 - w1
        eps = -1e-12
        inside = (w0 >= eps) & (w1 >= eps) & (w2 >= eps)
        if inside.any():
            z = w0[inside] * a[2] + w1[inside] * b[2] + w2[inside] * c[2]
            ii = idx[inside]
            np.maximum.at(out, ii, z)
    return out


def heightfield_mesh(height: np.ndarray, resolution: float, min_x: float = 0.0, min_y: float = 0.0):
    """The triangle mesh of a heightfield: vertices on the cell nodes, every cell split along its (i, j) - (i+1, j+1)
    diagonal (ORBIT's heightfield-to-mesh convention; the surface ``cfg.height_scanner.surface = "triangles"`` scans)."""
    H, W = height.shape
    xs = min_x + resolution * np.arange(W, dtype=np.float64)
    ys = min_y + resolution * np.arange(H, dtype=np.float64)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    verts = np.stack([X.ravel(), Y.ravel(), height.astype(np.float64).ravel()], 1)
    i, j = np.meshgrid(np.arange(H - 1), np.arange(W - 1), indexing="ij")
    n00 = (i * W + j).ravel()
    n01, n10, n11 = n00 + 1, n00 + W, n00 + W + 1
    faces = np.concatenate([np.stack([n00, n01, n11], 1), np.stack([n00, n11, n10], 1)], 0)
    return verts, faces
