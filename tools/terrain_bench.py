#!/usr/bin/env python3
"""Dev tool (GPU box): wall time of the terrain ingestion producers, host (numpy / scipy) vs device (HIP), on the
bench-sized 2048 x 2048 map and on a triangle mesh of the same extent.  Prints one JSON object."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaac_rover_orbit_amd import terrain as T, terrain_hip as TH


def timed(fn, reps=1):
    fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


res = {}
ter = T.make_procedural_terrain((2048, 2048), seed=1234, sigma_z=0.15, n_rocks=400)
h_dev = torch.from_numpy(ter.height).cuda()
t_host, (rock, safe) = timed(lambda: T.find_rocks_in_heightmap(ter.height))
t_dev, (rock_d, safe_d) = timed(lambda: TH.find_rocks_in_heightmap(h_dev), reps=5)
assert np.array_equal(rock_d.cpu().numpy(), rock) and np.array_equal(safe_d.cpu().numpy(), safe)
res["rock_mask_2048x2048"] = {"host_s": t_host, "device_s": t_dev, "speedup": t_host / t_dev, "identical": True}

# triangle mesh with one vertex per 0.1 m over 104 x 104 m (the reference's terrain USDs are of this kind)
n = 1041
x = np.linspace(0.0, 104.0, n, dtype=np.float32)
X, Y = np.meshgrid(x, x)
Z = (0.3 * np.sin(X * 0.3) * np.cos(Y * 0.2)).astype(np.float32)
verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
idx = np.arange(n * n).reshape(n, n)
a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
faces = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)])
sub = faces[:: max(1, len(faces) // 100000)]      # the host loop is O(minutes) on all 2.2 M triangles: time a 100 k sample
t0 = time.perf_counter(); ref_sub, *_ = T.mesh_to_heightmap(verts, sub); t_host_sub = time.perf_counter() - t0
t_dev_sub, (hm_sub, *_) = timed(lambda: TH.mesh_to_heightmap(verts, sub), reps=3)
assert np.array_equal(hm_sub.cpu().numpy(), ref_sub)
t_dev_all, (hm_all, *_) = timed(lambda: TH.mesh_to_heightmap(verts, faces), reps=3)
res["mesh_to_heightmap"] = {"triangles_sample": int(len(sub)), "host_s_sample": t_host_sub, "device_s_sample": t_dev_sub,
                            "host_s_all_extrapolated": t_host_sub * len(faces) / len(sub), "triangles_all": int(len(faces)),
                            "device_s_all": t_dev_all, "grid": list(hm_all.shape), "identical_on_sample": True}
print(json.dumps(res))
