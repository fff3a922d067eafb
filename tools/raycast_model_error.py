#!/usr/bin/env python3
"""How far are the height-scanner surfaces from a ray-cast of the terrain's triangle mesh?  

The reference ray-casts the hidden merged terrain MESH (rover_env_cfg.py:78-86, Warp mesh_query_ray).  Here the terrain is
a 0.05 m heightfield; its mesh is the grid triangulated along the (i, j) - (i+1, j+1) diagonals.  This tool scans random
poses on the BASELINE config 2 / config 4 terrains with the CPU oracle (== the HIP path, bit for bit) and reports, against
the exact float64 mesh ray-cast (oracle/mesh_raycast.py):
    * surface = "triangles" (default since round 2): the triangle planes themselves  -> fp32 rounding only
    * surface = "bilinear"  (round 1):               bilinear patch                  -> up to twist / 4 per cell
and, for a mesh whose vertices do not sit on the grid (terrain_from_mesh), the node-sampled surface vs the bounding-box
heightmap the reference only uses for look-ups.  Runs on the CPU in ~1 min:   python tools/raycast_model_error.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import terrain as T          # noqa: E402
from oracle import mesh_raycast as mr                   # noqa: E402
from oracle import rover_oracle as ro                   # noqa: E402


def scan_stats(ter, cfg_kw, n_pose, seed, lo, hi):
    rng = np.random.RandomState(seed)
    t = ro.TerrainData(ter.height, ter.obstacle, ter.safe_rock_mask, ter.resolution, ter.min_x, ter.min_y)
    S = ro.new_state(n_pose)
    S[:, 0] = rng.uniform(lo, hi, n_pose)
    S[:, 1] = rng.uniform(lo, hi, n_pose)
    S[:, 2] = 0.5
    yaw = rng.uniform(0, 2 * np.pi, n_pose)
    S[:, 3], S[:, 6] = np.cos(yaw / 2), np.sin(yaw / 2)
    out = {}
    cfg = ro.default_config(scan_height_offset=0.0, **cfg_kw)
    nx, ny = cfg.scan_nx, cfg.scan_ny
    ox = (-0.5 * cfg.scan_size_x + cfg.scan_resolution * np.arange(nx))[None, None, :]
    oy = (-0.5 * cfg.scan_size_y + cfg.scan_resolution * np.arange(ny))[None, :, None]
    c, s = np.cos(yaw)[:, None, None], np.sin(yaw)[:, None, None]
    X = (S[:, 0].astype(np.float64)[:, None, None] + c * ox - s * oy).reshape(-1)
    Y = (S[:, 1].astype(np.float64)[:, None, None] + s * ox + c * oy).reshape(-1)
    # exact mesh hits only need the cells the rays touch: build the mesh of the covered window
    j0, j1 = int((X.min() - ter.min_x) / 0.05) - 1, int((X.max() - ter.min_x) / 0.05) + 3
    i0, i1 = int((Y.min() - ter.min_y) / 0.05) - 1, int((Y.max() - ter.min_y) / 0.05) + 3
    verts, faces = mr.heightfield_mesh(ter.height[i0:i1, j0:j1], 0.05, ter.min_x + 0.05 * j0, ter.min_y + 0.05 * i0)
    z = mr.vertical_ray_hits(verts, faces, np.stack([X, Y], 1)).reshape(n_pose, ny * nx)
    twist = np.abs(ter.height[:-1, :-1] + ter.height[1:, 1:] - ter.height[:-1, 1:] - ter.height[1:, :-1])
    out["max_cell_twist_over_4_m"] = float(twist.max() / 4)
    for name, surf in (("triangles", 0), ("bilinear", 1)):
        cfg.scan_surface = surf
        obs = ro.height_scan(cfg, t, S)                      # = pos.z - hit.z  (offset 0)
        hit = 0.5 - obs.astype(np.float64)
        err = np.abs(hit - z)
        # relative to the observation term itself (height_scan_rover = pos.z - hit.z - 0.26878, typically 0.2 ... 0.5 m)
        term = np.abs(0.5 - z - 0.26878)
        rel = err / np.maximum(term, 1e-3)
        out[name] = {"max_abs_m": float(err.max()), "p99_abs_m": float(np.percentile(err, 99)), "mean_abs_m": float(err.mean()),
                     "p99_rel": float(np.percentile(rel, 99)), "share_of_rays_above_1e-3_rel": float((rel > 1e-3).mean())}
    return out


def mesh_case():
    """A 24 m x 20 m wavy sheet with boulders, 0.18 m triangles whose vertices do NOT sit on the 0.05 m grid."""
    rng = np.random.RandomState(5)
    nx, ny = 134, 112
    xs, ys = np.linspace(-0.3, 23.7, nx), np.linspace(0.2, 20.2, ny)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    Z = 0.25 * np.sin(0.7 * X) * np.cos(0.5 * Y) + 0.03 * X
    for _ in range(25):
        cx, cy, r, hgt = rng.uniform(3, 21), rng.uniform(3, 17), rng.uniform(0.3, 0.8), rng.uniform(0.2, 0.5)
        Z += hgt * np.exp(-((X - cx) ** 2 + (Y - cy) ** 2) / (2 * (r / 2) ** 2))
    V = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    a = (np.arange(ny - 1)[:, None] * nx + np.arange(nx - 1)[None, :]).ravel()
    F = np.concatenate([np.stack([a, a + 1, a + nx], 1), np.stack([a + 1, a + nx + 1, a + nx], 1)], 0).astype(np.uint32)
    ter = T.terrain_from_mesh(V, F)
    H, W = ter.shape
    q = np.stack([rng.uniform(ter.min_x + 0.2, ter.min_x + 0.05 * (W - 5), 200000),
                  rng.uniform(ter.min_y + 0.2, ter.min_y + 0.05 * (H - 5), 200000)], 1)
    z = mr.vertical_ray_hits(V, F, q)
    res = {"mesh": f"{F.shape[0]} triangles, 0.18 m edges, heights {Z.min():.2f} .. {Z.max():.2f} m", "rays": int(np.isfinite(z).sum())}
    for name, layer in (("node-sampled mesh surface + triangle scan (default)", ter.height),
                        ("reference bounding-box heightmap as the surface (round 1)", ter.lookup_height)):
        zs = mr.vertical_ray_hits(*mr.heightfield_mesh(layer, 0.05, ter.min_x, ter.min_y), q)
        ok = np.isfinite(zs) & np.isfinite(z)
        e = np.abs(zs[ok] - z[ok])
        res[name] = {"max_abs_m": float(e.max()), "p99_abs_m": float(np.percentile(e, 99)), "mean_abs_m": float(e.mean())}
    return res


if __name__ == "__main__":
    ro.build()
    out = {}
    out["config 2 (sigma_z 0.15 m, 31x31 rays @0.1 m)"] = scan_stats(
        T.make_procedural_terrain((2048, 2048), seed=1234, sigma_z=0.15, n_rocks=400), {}, 512, 0, 22.0, 80.0)
    out["config 4 (sigma_z 0.4 m, 32x32 rays @0.05 m)"] = scan_stats(
        T.make_procedural_terrain((2048, 2048), seed=1234, sigma_z=0.4, n_rocks=400),
        dict(scan_nx=32, scan_ny=32, scan_resolution=0.05, scan_size_x=1.55, scan_size_y=1.55), 512, 1, 22.0, 80.0)
    out["mesh ingestion"] = mesh_case()
    print(json.dumps(out, indent=1))
    json.dump(out, open(os.path.join(ROOT, "profiles", "r02_raycast_model_error.json"), "w"), indent=1)
