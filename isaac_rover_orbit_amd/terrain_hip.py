"""Device-side terrain ingestion: the two heavy init-time producers of ``terrain.py`` as HIP kernels.

``mesh_to_heightmap``        -- ``HeightmapManager.mesh_to_heightmap`` (terrain_utils.py:23-57)
``find_rocks_in_heightmap``  -- ``TerrainManager.find_rocks_in_heightmap`` (terrain_utils.py:265-311)

Both go through the C ABI of ``include/rover_terrain.h`` and return device tensors; results are bit-identical to the
numpy / scipy restatement in ``terrain.py`` (``tests/test_gpu_terrain.py``).  There is no CPU fallback here: without the
HIP library or a GPU these functions raise.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .terrain import GRADIENT_THRESHOLD, RESOLUTION, mesh_bounding_boxes, mesh_node_boxes


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _device(device) -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.RoverHipError("terrain_hip needs a ROCm GPU (no CPU fallback)")
    return torch.device(device)


def mesh_to_heightmap(vertices: np.ndarray, faces: np.ndarray, resolution: float = RESOLUTION, device="cuda"):
    """Returns ``(heightmap (H, W) fp32 device tensor, min_x, min_y, max_x, max_y)``.  The per-triangle cell boxes are
    computed on the host exactly as the reference does (vectorised, O(F)); the O(F x box) max-splat runs on the GPU."""
    dev = _device(device)
    lib = _lib.load()
    shape, (min_x, min_y, max_x, max_y), bbox, zmax = mesh_bounding_boxes(vertices, faces, resolution)
    with torch.cuda.device(dev):
        bbox_d = torch.from_numpy(np.ascontiguousarray(bbox)).to(dev)
        zmax_d = torch.from_numpy(np.ascontiguousarray(zmax)).to(dev)
        hm = torch.empty(shape, dtype=torch.float32, device=dev)
        _lib.check(lib.rover_terrain_rasterize(bbox_d.data_ptr(), zmax_d.data_ptr(), int(bbox.shape[0]), hm.data_ptr(),
                                               int(shape[0]), int(shape[1]), _stream(dev)), "rover_terrain_rasterize")
        torch.cuda.current_stream(dev).synchronize()   # bbox_d / zmax_d may be freed after this
    return hm, min_x, min_y, max_x, max_y


def mesh_surface_heights(vertices: np.ndarray, faces: np.ndarray, shape, min_x: float, min_y: float,
                         resolution: float = RESOLUTION, device="cuda") -> torch.Tensor:
    """``(H, W)`` fp32 device tensor: the mesh's height at every grid node (``rover_terrain_surface``; bit-identical to
    ``terrain.mesh_surface_heights``).  The per-triangle node boxes are computed on the host (vectorised, O(F))."""
    dev = _device(device)
    lib = _lib.load()
    tri, box = mesh_node_boxes(vertices, faces, shape, float(min_x), float(min_y), resolution)
    with torch.cuda.device(dev):
        tri_d = torch.from_numpy(tri).to(dev)
        box_d = torch.from_numpy(np.ascontiguousarray(box)).to(dev)
        hm = torch.empty(tuple(shape), dtype=torch.float32, device=dev)
        _lib.check(lib.rover_terrain_surface(tri_d.data_ptr(), box_d.data_ptr(), int(tri.shape[0]), hm.data_ptr(), int(shape[0]),
                                             int(shape[1]), float(min_x), float(min_y), float(resolution), _stream(dev)),
                   "rover_terrain_surface")
        torch.cuda.current_stream(dev).synchronize()
    return hm


def find_rocks_in_heightmap(heightmap, threshold: float = GRADIENT_THRESHOLD, device="cuda"):
    """Returns ``(rock_mask, safe_rock_mask)``, two ``(H, W)`` uint8 device tensors."""
    dev = _device(heightmap.device if isinstance(heightmap, torch.Tensor) and heightmap.is_cuda else device)
    lib = _lib.load()
    h = heightmap if isinstance(heightmap, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(heightmap, dtype=np.float32))
    h = h.to(device=dev, dtype=torch.float32).contiguous()
    H, W = int(h.shape[0]), int(h.shape[1])
    with torch.cuda.device(dev):
        rock = torch.empty((H, W), dtype=torch.uint8, device=dev)
        safe = torch.empty((H, W), dtype=torch.uint8, device=dev)
        scratch = torch.empty(int(lib.rover_terrain_scratch_bytes(H, W)), dtype=torch.uint8, device=dev)
        _lib.check(lib.rover_terrain_rock_mask(h.data_ptr(), H, W, float(threshold), rock.data_ptr(), safe.data_ptr(),
                                               scratch.data_ptr(), _stream(dev)), "rover_terrain_rock_mask")
    return rock, safe
