// terrain_kernels.hip -- device-side terrain ingestion (gfx950): mesh bounding-box rasterisation into the 0.05 m
// heightmap and the Sobel / morphology rock masks.  Init-time code (runs once per terrain), bit-identical to the host
// restatement in isaac_rover_orbit_amd/terrain.py.  Reference: rover_envs/envs/navigation/utils/terrains/
// terrain_utils.py:23-57 (mesh_to_heightmap), :265-311 (find_rocks_in_heightmap).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/rover_hip.h"
#include "../../include/rover_terrain.h"
#include "rover_internal.hpp"

namespace {

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t _e = (expr);                                                                                        \
        if (_e != hipSuccess) return rover_internal_fail(ROVER_ERR_HIP, #expr ": %s", hipGetErrorString(_e));         \
    } while (0)

__global__ void fill_f32_kernel(float *dst, size_t n, float v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}

// float max through integer atomics: floats with a clear sign bit order like ints, those with the sign bit set (-0.0
// included, which must still beat -99) like reversed uints
__device__ __forceinline__ void atomic_max_f32(float *addr, float v)
{
    if (__float_as_int(v) >= 0) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

// one thread per triangle: max-splat its z over the cells of its bounding box (terrain_utils.py:51-55)
__global__ void rasterize_kernel(const int4 *__restrict__ bbox, const float *__restrict__ zmax, int n_faces,
                                 float *__restrict__ hm, int H, int W)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_faces) return;
    const int4 b = bbox[f];  // min_i, max_i, min_j, max_j
    if (b.y < b.x || b.w < b.z) return;
    const float z = zmax[f];
    for (int j = b.z; j <= b.w; ++j) {
        const int jj = j < 0 ? j + H : j;  // numpy negative-index wrap
        if (jj < 0 || jj >= H) continue;
        for (int i = b.x; i <= b.y; ++i) {
            const int ii = i < 0 ? i + W : i;
            if (ii < 0 || ii >= W) continue;
            atomic_max_f32(hm + (size_t)jj * W + ii, z);
        }
    }
}

// one WAVE per triangle: the height of its PLANE at every grid node its xy-projection covers (first hit of a vertical
// ray from above = max over the covering triangles); the 64 lanes walk the triangle's node box in row-major order, so a
// small triangle costs one trip and a ground plane made of two huge triangles is spread over 64 lanes x coalesced rows
// instead of one lane walking millions of nodes (round-2 advisor finding).  fp64 barycentric weights in exactly the
// operation order of terrain.mesh_surface_heights (the host restatement), rounded to fp32 once; float max through integer
// atomics => independent of the visiting order.
__global__ __launch_bounds__(256) void surface_kernel(const double *__restrict__ tri /* n_faces x 9 */, const int4 *__restrict__ nbox,
                                                      int n_faces, float *__restrict__ hm, int H, int W, double min_x, double min_y,
                                                      double res)
{
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= n_faces) return;
    const double *t = tri + (size_t)f * 9;
    const double ax = t[0], ay = t[1], az = t[2], bx = t[3], by = t[4], bz = t[5], cx = t[6], cy = t[7], cz = t[8];
    const double den = (by - cy) * (ax - cx) + (cx - bx) * (ay - cy);
    if (den == 0.0) return;
    const int4 b = nbox[f];  // min_i, max_i, min_j, max_j (nodes, already clamped to the grid by the host)
    if (b.y < b.x || b.w < b.z) return;
    const long long bw = (long long)b.y - b.x + 1, cells = bw * ((long long)b.w - b.z + 1);
    for (long long c = lane; c < cells; c += 64) {
        const int j = b.z + (int)(c / bw), i = b.x + (int)(c % bw);
        const double py = min_y + res * (double)j, px = min_x + res * (double)i;
        const double w0 = ((by - cy) * (px - cx) + (cx - bx) * (py - cy)) / den;
        const double w1 = ((cy - ay) * (px - cx) + (ax - cx) * (py - cy)) / den;
        const double w2 = 1.0 - w0 - w1;
        if (w0 >= -1e-9 && w1 >= -1e-9 && w2 >= -1e-9)
            atomic_max_f32(hm + (size_t)j * W + i, (float)(w0 * az + w1 * bz + w2 * cz));
    }
}

// Sobel with wrap-around borders (scipy convolve2d boundary="wrap"), fp64 like numpy: the 6-term sums of fp32 heights
// times {1, 2} are exact in fp64, so only the two squares, their sum and the square root round -- in numpy's order.
__global__ void sobel_threshold_kernel(const float *__restrict__ h, int H, int W, double thr, uint8_t *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int xm = x == 0 ? W - 1 : x - 1, xp = x == W - 1 ? 0 : x + 1;
    const int ym = y == 0 ? H - 1 : y - 1, yp = y == H - 1 ? 0 : y + 1;
    const double a = h[(size_t)ym * W + xm], b = h[(size_t)ym * W + x], c = h[(size_t)ym * W + xp];
    const double d = h[(size_t)y * W + xm], f = h[(size_t)y * W + xp];
    const double g = h[(size_t)yp * W + xm], hh = h[(size_t)yp * W + x], k = h[(size_t)yp * W + xp];
    const double gx = (c - a) + 2.0 * (f - d) + (k - g);
    const double gy = (g - a) + 2.0 * (hh - b) + (k - c);
    const double mag = sqrt(gx * gx + gy * gy);
    out[(size_t)y * W + x] = mag > thr ? 1 : 0;
}

// k x k rank filters of a 0/1 image as two 1-D passes.  Window offsets [-(k / 2), k - 1 - k / 2] (cv2 anchor k / 2 =
// scipy origin 0); outside the image: 0 for the max filter (dilate), 1 for the min filter (erode) -- i.e. ignored.
template <bool IS_MAX, bool ALONG_X>
__global__ void rank1d_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int H, int W, int k)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int lo = -(k / 2), hi = k - 1 - k / 2;
    uint8_t r = IS_MAX ? 0 : 1;
    for (int o = lo; o <= hi; ++o) {
        const int xx = ALONG_X ? x + o : x, yy = ALONG_X ? y : y + o;
        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
        const uint8_t v = in[(size_t)yy * W + xx];
        r = IS_MAX ? (r | v) : (r & v);
    }
    out[(size_t)y * W + x] = r;
}

// scipy.ndimage.binary_fill_holes: the background reachable from the image border through 4-connected background
// paths stays background, everything else becomes foreground.  `reach` = background reached so far.
__global__ void holes_init_kernel(const uint8_t *__restrict__ mask, uint8_t *__restrict__ reach, int H, int W)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const bool border = x == 0 || y == 0 || x == W - 1 || y == H - 1;
    reach[(size_t)y * W + x] = (border && mask[(size_t)y * W + x] == 0) ? 1 : 0;
}
// one thread per row (ALONG_X) or per column: a forward and a backward sweep carry reachability along the line
template <bool ALONG_X>
__global__ void holes_sweep_kernel(const uint8_t *__restrict__ mask, uint8_t *__restrict__ reach, int H, int W, int *changed)
{
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_lines = ALONG_X ? H : W, len = ALONG_X ? W : H;
    if (line >= n_lines) return;
    const size_t base = ALONG_X ? (size_t)line * W : (size_t)line, step = ALONG_X ? 1 : (size_t)W;
    bool any = false;
    uint8_t carry = 0;
    for (int t = 0; t < len; ++t) {
        const size_t p = base + (size_t)t * step;
        const uint8_t bg = mask[p] == 0, r = reach[p];
        const uint8_t nr = r | (carry & bg);
        if (nr != r) { reach[p] = nr; any = true; }
        carry = nr;
    }
    carry = 0;
    for (int t = len - 1; t >= 0; --t) {
        const size_t p = base + (size_t)t * step;
        const uint8_t bg = mask[p] == 0, r = reach[p];
        const uint8_t nr = r | (carry & bg);
        if (nr != r) { reach[p] = nr; any = true; }
        carry = nr;
    }
    if (any) atomicOr(changed, 1);
}
__global__ void holes_finish_kernel(const uint8_t *__restrict__ reach, uint8_t *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = reach[i] ? 0 : 1;
}

template <bool IS_MAX>
void rank2d(const uint8_t *in, uint8_t *tmp, uint8_t *out, int H, int W, int k, hipStream_t st)
{
    const dim3 grid((W + 255) / 256, H), block(256);
    hipLaunchKernelGGL((rank1d_kernel<IS_MAX, true>), grid, block, 0, st, in, tmp, H, W, k);
    hipLaunchKernelGGL((rank1d_kernel<IS_MAX, false>), grid, block, 0, st, tmp, out, H, W, k);
}

}  // namespace

extern "C" {

int rover_terrain_rasterize(const int32_t *bbox, const float *zmax, int32_t n_faces, float *height, int32_t H, int32_t W,
                            void *stream)
{
    if (!height || H < 1 || W < 1 || n_faces < 0) return rover_internal_fail(ROVER_ERR_INVALID, "bad heightmap shape");
    if (n_faces > 0 && (!bbox || !zmax)) return rover_internal_fail(ROVER_ERR_INVALID, "bbox / zmax is NULL");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, height, n, -99.0f);
    if (n_faces > 0)
        hipLaunchKernelGGL(rasterize_kernel, dim3((n_faces + 63) / 64), dim3(64), 0, st, reinterpret_cast<const int4 *>(bbox),
                           zmax, n_faces, height, H, W);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

size_t rover_terrain_scratch_bytes(int32_t H, int32_t W)
{
    if (H < 1 || W < 1) return 0;
    return 3 * (((size_t)H * W + 255) & ~(size_t)255) + 256;  // three byte images + the convergence flag
}

int rover_terrain_surface(const double *tri, const int32_t *node_box, int32_t n_faces, float *height, int32_t H, int32_t W,
                          double min_x, double min_y, double resolution, void *stream)
{
    if (!tri || !node_box || !height || H <= 0 || W <= 0 || n_faces < 0 || !(resolution > 0.0))
        return rover_internal_fail(ROVER_ERR_INVALID, "rover_terrain_surface: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, height, n, -99.0f);
    if (n_faces > 0)
        hipLaunchKernelGGL(surface_kernel, dim3((n_faces + 3) / 4), dim3(256), 0, st, tri,
                           reinterpret_cast<const int4 *>(node_box), n_faces, height, H, W, min_x, min_y, resolution);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

int rover_terrain_rock_mask(const float *height, int32_t H, int32_t W, double threshold, uint8_t *rock, uint8_t *safe,
                            void *scratch, void *stream)
{
    if (!height || !rock || !safe || !scratch) return rover_internal_fail(ROVER_ERR_INVALID, "NULL argument");
    if (H < 3 || W < 3) return rover_internal_fail(ROVER_ERR_INVALID, "heightmap smaller than the Sobel kernel");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t n = (size_t)H * W, pad = (n + 255) & ~(size_t)255;
    uint8_t *a = static_cast<uint8_t *>(scratch), *b = a + pad, *c = b + pad;
    int *flag = reinterpret_cast<int *>(c + pad);
    const dim3 grid((W + 255) / 256, H), block(256);
    hipLaunchKernelGGL(sobel_threshold_kernel, grid, block, 0, st, height, H, W, threshold, a);
    rank2d<true>(a, b, c, H, W, 3, st);    // MORPH_CLOSE 3x3 = dilate, erode   (terrain_utils.py:293-294)
    rank2d<false>(c, b, a, H, W, 3, st);
    // fill holes (:301): reach -> b, result -> c
    hipLaunchKernelGGL(holes_init_kernel, grid, block, 0, st, a, b, H, W);
    for (int round = 0; round < H + W; ++round) {
        HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), st));
        hipLaunchKernelGGL((holes_sweep_kernel<true>), dim3((H + 63) / 64), dim3(64), 0, st, a, b, H, W, flag);
        hipLaunchKernelGGL((holes_sweep_kernel<false>), dim3((W + 63) / 64), dim3(64), 0, st, a, b, H, W, flag);
        int changed = 0;
        HIP_TRY(hipMemcpyAsync(&changed, flag, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (!changed) break;
    }
    hipLaunchKernelGGL(holes_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b, c, n);
    rank2d<false>(c, b, a, H, W, 7, st);   // MORPH_OPEN 7x7 = erode, dilate    (:303-304)
    rank2d<true>(a, b, c, H, W, 7, st);
    rank2d<true>(c, b, rock, H, W, 11, st);   // dilate 11x11                    (:305-306)
    rank2d<true>(rock, b, safe, H, W, 42, st);  // dilate 42x42 -> safe mask      (:308-309)
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return ROVER_OK;
}

}  // extern "C"
