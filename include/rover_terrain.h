/*
 * rover_terrain.h -- C ABI of the device-side terrain ingestion (librover_hip.so), SURVEY 8(a12) / 8(f-2).
 *
 * Replaces the init-time numpy / OpenCV producers of the shared terrain data in
 *     rover_envs/envs/navigation/utils/terrains/terrain_utils.py
 *         HeightmapManager.mesh_to_heightmap          :23-57    -> rover_terrain_rasterize
 *         TerrainManager.find_rocks_in_heightmap      :265-311  -> rover_terrain_rock_mask
 *     and adds rover_terrain_surface: the mesh's true height at the grid nodes (what the reference's RayCaster / PhysX see)
 * (random_rover_spawns :330-385 draws from numpy's legacy MT19937 stream and stays on the host.)
 *
 * Conventions as in rover_hip.h: plain C, caller-owned DEVICE buffers, int return codes, rover_last_error() for the
 * text.  Results are bit-identical to the host restatement in isaac_rover_orbit_amd/terrain.py (tests/test_gpu_terrain.py).
 */
#ifndef ROVER_TERRAIN_H
#define ROVER_TERRAIN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mesh_to_heightmap, terrain_utils.py:36-55: heightmap[j, i] = max(heightmap[j, i], max vertex z of the triangle) for
 * every cell of the triangle's BOUNDING BOX (no true rasterisation), cells never covered stay at -99.
 *   bbox   (n_faces, 4) int32  {min_i, max_i, min_j, max_j} in cells, computed by the caller exactly as the reference
 *          does (python int() truncation, max clamped to the grid: :42-49); negative indices wrap like numpy's do
 *   zmax   (n_faces,) fp32     max vertex z per triangle (:51)
 *   height (H, W) fp32         output, row = y (j), column = x (i); overwritten (initialised to -99 by this call)
 * Asynchronous on `stream`. */
int rover_terrain_rasterize(const int32_t *bbox, const float *zmax, int32_t n_faces, float *height, int32_t H, int32_t W,
                            void *stream);

/* The surface the wheels touch and the height scanner's rays hit, sampled from the SOURCE MESH: height[j, i] = z of the
 * first hit of a vertical ray from above through node (min_x + i * res, min_y + j * res) = max over the triangles whose
 * xy-projection covers the node of the triangle's plane there; nodes no triangle covers stay at -99.  (The reference
 * ray-casts the hidden terrain mesh itself, rover_env_cfg.py:78-86; its bounding-box heightmap above over-estimates every
 * slope and is kept for what the reference uses it for: target / spawn look-ups.)
 *   tri      (n_faces, 9) fp64   triangle corners (ax, ay, az, bx, by, bz, cx, cy, cz)
 *   node_box (n_faces, 4) int32  {min_i, max_i, min_j, max_j}: nodes inside the triangle's bounding box, clamped to the grid
 * Same arithmetic as isaac_rover_orbit_amd/terrain.py::mesh_surface_heights (bit-identical).  Asynchronous on `stream`. */
int rover_terrain_surface(const double *tri, const int32_t *node_box, int32_t n_faces, float *height, int32_t H, int32_t W,
                          double min_x, double min_y, double resolution, void *stream);

/* find_rocks_in_heightmap, terrain_utils.py:265-311: Sobel (wrap) gradient magnitude > threshold -> MORPH_CLOSE 3x3 ->
 * fill holes -> MORPH_OPEN 7x7 -> dilate 11x11 (`rock`) -> dilate 42x42 (`safe`); cv2 anchor convention for the even
 * kernel (window offsets [-(k / 2), k - 1 - k / 2]).
 *   height (H, W) fp32 in; rock, safe (H, W) uint8 out (0 / 1);
 *   scratch: rover_terrain_scratch_bytes(H, W) bytes of device memory.
 * The fill-holes propagation iterates to convergence, so this call SYNCHRONISES `stream` (init-time code, like the
 * reference's). */
int rover_terrain_rock_mask(const float *height, int32_t H, int32_t W, double threshold, uint8_t *rock, uint8_t *safe,
                            void *scratch, void *stream);
size_t rover_terrain_scratch_bytes(int32_t H, int32_t W);

#ifdef __cplusplus
}
#endif
#endif /* ROVER_TERRAIN_H */
