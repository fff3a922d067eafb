// rover_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) + C ABI of the AAURoverEnv-v0 hot path.
//
// Two launches per env.step():
//   K1  rover_step_kernel[_group]  one env per LANE (large N) or per 16-LANE GROUP (wheel slot x channel role per lane; default),
//                             SoA state (state[word * N + env]).  process_action -> Ackermann -> 6 x {implicit-PD
//                             steer joints, 6-wheel contact solve against the bilinear heightfield, wheel motors,
//                             integration} -> counters -> terminations -> rewards -> in-lane reset (Philox) ->
//                             command update; writes the observation head and a 32-byte scan descriptor per env.
//                             Wave-level butterfly reductions produce the per-wave partials of extras["log"].
//   K2  rover_scan_step_kernel<Q16, TRI, 1024, EPI>  PERSISTENT 1024-thread workgroups (two per CU), each walking PAIRS of
//                             envs (EPI = 2): the yaw-rotated 3 x 3 m terrain windows of both envs are copied global -> LDS
//                             asynchronously (global_load_lds_dwordx4, dense int16 or fp32 tiles, ONE buffer per env), one
//                             wait + barrier, then every thread casts its vertical ray in both envs against the TRIANGLE
//                             of the heightfield cell it falls in (four LDS corner reads, two fma) and the 965-float
//                             observation rows are written with coalesced stores; one extra workgroup reduces the log
//                             partials in a fixed order (deterministic).  rover_scan_obs_kernel<MODE, Q16, TRI> is the
//                             generic 512-thread form (reset path, unit entries, odd map widths, > 1024 rays).
// There is no matrix-shaped work on this path (gather / integrate / scatter) => no MFMA; the bound is HBM/latency.
//
// Reference behaviour being replaced (file:line in /root/reference): RoverEnv.step entrypoints/rover_env.py:42-102;
// AckermannAction2 mdp/actions/ackermann_actions.py:226-322; mdp terms envs/navigation/mdp/{observations,rewards,
// terminations,randomizations}.py; TerrainBasedPositionCommand utils/terrains/terrain_importer.py:74-175;
// HeightmapManager.get_height_at / check_if_target_is_valid utils/terrains/terrain_utils.py:62-84,202-223.
// Physics / ray-caster / contact sensor are third-party in the reference (PhysX, Warp); the reduced rover model used
// here is specified in DESIGN.md.  All arithmetic is fp32 with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <vector>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <type_traits>

#include <dlfcn.h>

#include "../../include/rover_hip.h"
#include "../../include/rover_debug.h"
#include "rover_model.hpp"
#include "rover_internal.hpp"

namespace {

// quantities that depend only on the physics time step (hoisted reciprocals: one division each, then products); evaluated
// ONCE on the host (same IEEE fp32 operations, -ffp-contract=off) and handed to the kernels in the parameter block
struct StepConsts {
    float h, inv_h, inv_m, inv_I[3];
    float steer_hkp, steer_den_inv, steer_i_over_h, steer_dv_max;
    float wheel_den_inv, wheel_h_over_i, lt_motor;
    float bogie_keep[3], b_winv[3];
    float bogie_gq[3];   // cfg.mass_model = 1: -(m_j g) h / I_j, the bogie-rate change per unit of (R (ax x arm_j)).z
};
__host__ __device__ inline void make_step_consts(float h, StepConsts &k)
{
    constexpr float INERTIA_B[3] = RV_INERTIA_B_INIT;
    constexpr float BOGIE_INERTIA[3] = RV_BOGIE_INERTIA_INIT;
    constexpr float SUBTREE_MASS[3] = RV_SUBTREE_MASS_INIT;
    k.h = h;
    k.inv_h = 1.0f / h;
    k.inv_m = 1.0f / RV_M_TOTAL;
    for (int i = 0; i < 3; ++i) k.inv_I[i] = 1.0f / INERTIA_B[i];
    k.steer_hkp = h * RV_STEER_KP;
    k.steer_den_inv = 1.0f / (RV_STEER_INERTIA + h * RV_STEER_KD + h * h * RV_STEER_KP);
    k.steer_i_over_h = RV_STEER_INERTIA / h;
    k.steer_dv_max = RV_STEER_EFFORT * h / RV_STEER_INERTIA;
    k.wheel_den_inv = 1.0f / (RV_WHEEL_INERTIA + h * RV_WHEEL_KD + h * h * RV_WHEEL_KP);
    k.wheel_h_over_i = h / RV_WHEEL_INERTIA;
    k.lt_motor = RV_WHEEL_EFFORT * h / RV_WHEEL_CONTACT_RADIUS;
    for (int j = 0; j < 3; ++j) {
        k.bogie_keep[j] = 1.0f / (1.0f + h * RV_BOGIE_DAMPING / BOGIE_INERTIA[j]);
        k.b_winv[j] = 1.0f / BOGIE_INERTIA[j];
        k.bogie_gq[j] = (-(SUBTREE_MASS[j] * RV_GRAVITY) * h) * k.b_winv[j];
    }
}

struct RvParams {
    rover_config cfg;
    StepConsts K;
    const float *height;
    const float *lookup;     // heightmap of HeightmapManager.get_height_at (target z); == height unless set separately
    const float *obstacle;
    const uint8_t *safe_mask;
    const float *spawns;
    int H, W, n_spawns;
    float res, min_x, min_y;
    int n, env_id_offset;
    uint32_t spawn_a, spawn_b;  // spawn_draw = 1: row = (spawn_a * global id + spawn_b) mod n_spawns for this launch
    int rays, obs_w;
    int tile_dim;    // LDS tile rows (cells) for the scan kernel
    int tile_pitch;  // cells per LDS tile row (multiple of the cells per 16-byte chunk)
    // optional exact 16-bit copy of `height` (height == height_q * q_scale for every cell): halves the bytes the scan
    // kernel stages; chunk_cells = cells per 16-byte chunk of the array the scan kernel reads (4: fp32, 8: int16)
    const int16_t *height_q;
    float q_scale;
    int chunk_cells;
    float *scan_desc;  // [n][8] per-env scan descriptor written by the step kernel: px, py, pz, cos(yaw), sin(yaw),
                       // i_lo, j_lo, (th | interior << 15 | tw4 << 16) of the terrain window (ints as raw bits)
    // host-precomputed uniforms of the scan kernel (a per-thread IEEE division costs ~10 VALU instructions)
    float inv_res, x_max, y_max, inv_nx;
    int cpr_log;  // log2 of the chunk slots per staged row (smallest power of two >= tile_pitch / chunk_cells)
    int wq, pq;   // 16-byte chunks per heightfield row / per LDS tile row
    int tile_bufs; // LDS tile buffers of the scan kernel (2 when they fit beside full occupancy, else 1)
    // extras["log"]: a wave of the step kernel writes its partial row only when one of its envs reset, tagged (word 15) with
    // this launch's step_tag, and counts itself in *log_counter; the scan kernel reduces the rows carrying the tag -- or, in
    // the common case of a step without resets, reads the counter and does nothing
    uint32_t step_tag;
    unsigned *log_counter;
    // step-form scan kernel, two envs per round: 1 = pairs are dealt to workgroups so that the eight pairs of a 16-row block of
    // observations (rows 16 b .. 16 b + 15) are produced on ONE XCD, the XCD (b mod 8) on which workgroup b of a kernel with
    // one workgroup per 16 rows -- the policy / value forward pass -- will run: its read then hits that XCD's L2
    int xcd_rows;
    int nt_obs;   // 1 = the one-launch kernels store the observation rows with streaming (non-temporal) stores (rover_set_obs_streaming)
};

// ------------------------------------------------------------------------------------------------ small helpers
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
// The same clamp as ONE v_med3_f32 (instead of two compares, two selects and their wait states).  Bit-identical to clampf
// for every finite x when lo < hi and lo is not -0 / hi is not +0 with x the other zero: all uses below have lo < 0 < hi or
// lo = +0 with x never -0 (x is a difference a - b, which is +0 when a == b).
__device__ __forceinline__ float clamp_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(const float *a, const float *b, float *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float sign_torch(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
__device__ __forceinline__ void quat_to_mat(const float *q, float R[3][3])
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.0f - 2.0f * (y * y + z * z);
    R[0][1] = 2.0f * (x * y - w * z);
    R[0][2] = 2.0f * (x * z + w * y);
    R[1][0] = 2.0f * (x * y + w * z);
    R[1][1] = 1.0f - 2.0f * (x * x + z * z);
    R[1][2] = 2.0f * (y * z - w * x);
    R[2][0] = 2.0f * (x * z - w * y);
    R[2][1] = 2.0f * (y * z + w * x);
    R[2][2] = 1.0f - 2.0f * (x * x + y * y);
}
__device__ __forceinline__ void mat_vec(const float R[3][3], const float *v, float *o)
{
    o[0] = R[0][0] * v[0] + R[0][1] * v[1] + R[0][2] * v[2];
    o[1] = R[1][0] * v[0] + R[1][1] * v[1] + R[1][2] * v[2];
    o[2] = R[2][0] * v[0] + R[2][1] * v[1] + R[2][2] * v[2];
}

__device__ __forceinline__ void mat_tvec(const float R[3][3], const float *v, float *o)  // o = R^T v
{
    o[0] = R[0][0] * v[0] + R[1][0] * v[1] + R[2][0] * v[2];
    o[1] = R[0][1] * v[0] + R[1][1] * v[1] + R[2][1] * v[2];
    o[2] = R[0][2] * v[0] + R[1][2] * v[1] + R[2][2] * v[2];
}
__device__ __forceinline__ float wdot3x(const float *a, const float *b, const float *w) { return a[0] * b[0] * w[0] + a[1] * b[1] * w[1] + a[2] * b[2] * w[2]; }
__device__ __forceinline__ float wdot3(const float *a, const float *w) { return a[0] * a[0] * w[0] + a[1] * a[1] * w[1] + a[2] * a[2] * w[2]; }


/* ---- portable fp32 trigonometry ------------------------------------------------------------------------
 * sin/cos/atan2 are evaluated by the SAME explicit fp32 operation sequence here and in the HIP kernels (Cody-Waite
 * reduction + Cephes single-precision minimax polynomials, ~1 ulp), so that the CPU oracle and the GPU agree bit for
 * bit instead of differing by the libm-vs-device-library rounding of sinf/cosf/atan2f.                       */
__device__ __forceinline__ void rv_sincosf(float x, float *s, float *c)
{
    const float k = floorf(x * 0.63661977236758134f + 0.5f); /* nearest multiple of pi/2 */
    float r = x - k * 1.5703125f;
    r = r - k * 4.837512969970703125e-4f;
    r = r - k * 7.54978995489188e-8f;
    const int q = ((int)k) & 3;
    const float z = r * r;
    const float sp = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * (-1.9515295891e-4f)));
    const float cp = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    const float ss = (q & 1) ? cp : sp;
    const float cc = (q & 1) ? sp : cp;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}
// rv_sincosf for |x| < pi/4: the reduction finds k = 0 and leaves r = x, the quadrant selects pick (sp, cp) -- so the two
// polynomials alone give bit-identical results (bogie angles are clamped to +-10 deg)
__device__ __forceinline__ void rv_sincosf_small(float x, float *s, float *c)
{
    const float z = x * x;
    *s = x + x * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * (-1.9515295891e-4f)));
    *c = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
}
__device__ __forceinline__ float rv_sinf(float x) { float s, c; rv_sincosf(x, &s, &c); return s; }
__device__ __forceinline__ float rv_cosf(float x) { float s, c; rv_sincosf(x, &s, &c); return c; }
__device__ __forceinline__ float rv_atan2f(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    float a;
    if (ax == 0.0f) {
        a = (ay == 0.0f) ? 0.0f : 1.5707963267948966f;
    } else {
        // Cephes' range reduction -- t -> -(1 / t) above tan(3 pi / 8), t -> (t - 1) / (t + 1) above tan(pi / 8) -- as ONE division of
        // selected operands: (-1) / t == -(1 / t) and t / 1 == t exactly, so this is the oracle's branchy form bit for bit; written
        // as branches hipcc if-converts it into three IEEE divisions (~10 instructions each) and two selects
        const float t0 = ay / ax;
        const bool big = t0 > 2.414213562373095f, mid = t0 > 0.4142135623730950f;
        const float y0 = big ? 1.5707963267948966f : (mid ? 0.7853981633974483f : 0.0f);
        const float num = big ? -1.0f : (mid ? t0 - 1.0f : t0), den = big ? t0 : (mid ? t0 + 1.0f : 1.0f);
        const float t = num / den;
        const float z = t * t;
        a = y0 + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * t + t);
    }
    if (x < 0.0f) a = RV_PI_F - a;
    return (y < 0.0f) ? -a : a;
}

// Philox4x32-10, counter = (global env id, reset count, draw block, stream), key = seed
__host__ __device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                                    uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }

// ------------------------------------------------------------------------------------------------ (a2) Ackermann
// AckermannAction2.process_actions / ackermann, ackermann_actions.py:226-322 (quirks B-4, B-5 kept)
// rim speeds (before the division by the wheel radius) and steering angles; ackermann_one() below adds the six divisions, the
// 16-lanes-per-env step kernel divides only the speed of the lane's own wheel
__device__ __forceinline__ void ackermann_core(const rover_config &c, const float *raw, float *processed, float *steer,
                                               float *rim /* ML, FL, RL, RR, MR, FR */)
{
    processed[0] = raw[0] * c.scale_lin + c.offset_lin;
    processed[1] = raw[1] * c.scale_ang + c.offset_ang;
    float lin = processed[0], ang = processed[1];
    const float d_fr = c.d_fr, d_mw = c.d_mw, wl = c.wheelbase;
    float direction = sign_torch(lin);
    const float turn = sign_torch(ang);
    if (direction == 0.0f) direction = direction + 1.0f;
    lin = fabsf(lin);
    ang = fabsf(ang);
    const bool not_zero = (ang != 0.0f) || (lin != 0.0f);
    const float min_radius = d_mw * 0.8f;
    float R = not_zero ? lin / ang : INFINITY;
    if (R < min_radius) R = min_radius;
    const float r_ML = R - (d_mw / 2.0f), r_MR = R + (d_mw / 2.0f);
    const float r_FL = R - (d_fr / 2.0f), r_FR = R + (d_fr / 2.0f);
    const float r_RL = R - (d_fr / 2.0f), r_RR = R + (d_fr / 2.0f);
    const bool point = R < d_mw;
    const float pt = (lin + 1.0f) * turn;
    const bool az = ang == 0.0f;
    const float v_FL = point ? -pt : (az ? lin : (r_FL * ang)) * direction;
    const float v_FR = point ? pt : (az ? lin : (r_FR * ang)) * direction;
    const float v_RL = point ? -pt : (az ? lin : (r_RL * ang)) * direction;
    const float v_RR = point ? pt : (az ? lin : (r_RR * ang)) * direction;
    const float v_ML = point ? -pt : (az ? lin : (r_ML * ang)) * direction;
    const float v_MR = point ? pt : (az ? lin : (r_MR * ang)) * direction;
    const float th = rv_atan2f(wl, r_FL) * turn;
    const float q = RV_PI_F / 4.0f;
    rim[0] = v_ML; rim[1] = v_FL; rim[2] = v_RL; rim[3] = v_RR; rim[4] = v_MR; rim[5] = v_FR;
    steer[0] = point ? -q : th;  // FL
    steer[1] = point ? q : th;   // RL
    steer[2] = point ? -q : th;  // RR
    steer[3] = point ? q : th;   // FR
}
__device__ __forceinline__ void ackermann_one(const rover_config &c, const float *raw, float *processed, float *steer,
                                              float *wheel)
{
    float rim[6];
    ackermann_core(c, raw, processed, steer, rim);
#pragma unroll
    for (int i = 0; i < 6; ++i) wheel[i] = rim[i] / c.wheel_radius;
}

// ------------------------------------------------------------------------------------------------ terrain look-ups
// index quirk of get_height_at / check_if_target_is_valid: long(xy / res + [min_x, min_y]) (terrain_utils.py:75,211)
__device__ __forceinline__ void quirk_cell(const RvParams &p, float x, float y, int &cx, int &cy)
{
    const float sx = x / p.res + p.min_x;
    const float sy = y / p.res + p.min_y;
    long long ix = (long long)sx, iy = (long long)sy;
    if (ix < 0) ix = 0;
    if (ix > p.W - 1) ix = p.W - 1;
    if (iy < 0) iy = 0;
    if (iy > p.H - 1) iy = p.H - 1;
    cx = (int)ix;
    cy = (int)iy;
}

// bilinear patch of the 0.05 m grid under (x, y): height, gradient and obstacle-layer height; 4 + 4 gathers
template <bool WANT_OBST>
__device__ __forceinline__ void terrain_sample(const RvParams &p, float x, float y, float &h, float &gx, float &gy,
                                               float &obst)
{
    const float inv_res = p.inv_res;   // = 1.0f / p.res, computed once on the host (same IEEE division)
    float u = (x - p.min_x) * inv_res;
    float v = (y - p.min_y) * inv_res;
    u = clamp_med3(u, 0.0f, (float)(p.W - 1));
    v = clamp_med3(v, 0.0f, (float)(p.H - 1));
    int j0 = (int)u, i0 = (int)v;
    if (j0 > p.W - 2) j0 = p.W - 2;
    if (i0 > p.H - 2) i0 = p.H - 2;
    const float fx = u - (float)j0, fy = v - (float)i0;
    const size_t base = (size_t)i0 * p.W + j0;
    const float *q = p.height + base;
    const float h00 = q[0], h01 = q[1], h10 = q[p.W], h11 = q[p.W + 1];
    const float dx0 = h01 - h00, dx1 = h11 - h10, dy0 = h10 - h00, dy1 = h11 - h01;
    const float hx0 = h00 + fx * dx0;
    const float hx1 = h10 + fx * dx1;
    h = hx0 + fy * (hx1 - hx0);
    gx = (dx0 + fy * (dx1 - dx0)) * inv_res;
    gy = (dy0 + fx * (dy1 - dy0)) * inv_res;
    if (WANT_OBST) {
        const float *o = p.obstacle + base;
        const float o00 = o[0], o01 = o[1], o10 = o[p.W], o11 = o[p.W + 1];
        const float o0 = o00 + fx * (o01 - o00);
        const float o1 = o10 + fx * (o11 - o10);
        obst = o0 + fy * (o1 - o0);
    }
}

// ------------------------------------------------------------------------------------------------ (a6) command
__device__ __forceinline__ float wrap_to_pi(float a)
{
    // fmodf(a, 2 pi) for |a| < 4 pi (here |a| <= 2 pi: a difference of two headings in [-pi, pi]): the quotient is 0 or +-1 and
    // a -+ 2 pi is exact (Sterbenz), so this IS fmodf's result -- sign of a zero result included -- without its long division
    float r = a;
    if (fabsf(a) >= RV_TWO_PI_F) r = copysignf(a - copysignf(RV_TWO_PI_F, a), a);
    if (r != 0.0f && r < 0.0f) r += RV_TWO_PI_F;
    if (r > RV_PI_F) r -= RV_TWO_PI_F;
    return r;
}
__device__ __forceinline__ float heading_of(const float *q)
{
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float fx = 1.0f - 2.0f * (y * y + z * z);
    const float fy = 2.0f * (w * z + x * y);
    return rv_atan2f(fy, fx);
}
// TerrainBasedPositionCommand._update_command, terrain_importer.py:97-101 (ORBIT yaw_quat + quat_rotate_inverse)
__device__ __forceinline__ void update_command_one(const float *pos, const float *quat, const float *target_w,
                                                   float heading_cmd_w, float *cmd_b, float *heading_b, const float *heading_w = nullptr)
{
    const float qw = quat[0], qx = quat[1], qy = quat[2], qz = quat[3];
    const float yaw = rv_atan2f(2.0f * (qw * qz + qx * qy), 1.0f - 2.0f * (qy * qy + qz * qz));
    float yw = rv_cosf(yaw / 2.0f), yz = rv_sinf(yaw / 2.0f);
    const float nrm = sqrtf(yw * yw + yz * yz);
    yw = yw / nrm;
    yz = yz / nrm;
    const float v[3] = {target_w[0] - pos[0], target_w[1] - pos[1], target_w[2] - pos[2]};
    const float s = 2.0f * yw * yw - 1.0f;
    const float qv[3] = {0.0f, 0.0f, yz};
    float cr[3];
    cross3(qv, v, cr);
    const float dt = qv[0] * v[0] + qv[1] * v[1] + qv[2] * v[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) cmd_b[i] = v[i] * s - cr[i] * yw * 2.0f + qv[i] * dt * 2.0f;
    // heading_of(quat) is the SAME expression as `yaw` above (same operands, same operations): one atan2, not two -- hipcc does
    // not merge the two evaluations, each sits behind its own `x == 0` branch
    *heading_b = wrap_to_pi(heading_cmd_w - (heading_w ? *heading_w : yaw));
}

// ------------------------------------------------------------------------------------------------ (a9-a11) mdp terms
__device__ __forceinline__ float collision_measure(const float *F /* 13 x 3 */)
{
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = 0.0f;
#pragma unroll
        for (int b = 0; b < ROVER_NUM_BODIES; ++b) s += F[b * 3 + c] * F[b * 3 + c];
        acc += sqrtf(s);
    }
    return acc;
}

// rewards.py:14-137, terminations.py:14-64, ORBIT mdp.time_out; rew = unweighted term values
// no_force: the caller knows that every entry of F is +0 (wave-uniform), so the collision measure is exactly 0
// terminations.py:14-64 + ORBIT mdp.time_out alone (the same expressions as in mdp_terms_one below): the one-launch kernel
// decides the reset BEFORE the rest of the manager tail, so that the copy wave can stage the windows of the final pose
__device__ __forceinline__ void mdp_terminations(const rover_config &c, const float *cmd_b, int ep_len, bool coll, bool *term)
{
    const float d = sqrtf(cmd_b[0] * cmd_b[0] + cmd_b[1] * cmd_b[1]);
    term[0] = ep_len >= c.max_episode_length;
    term[1] = d < c.success_threshold;
    term[2] = d > c.far_threshold;
    term[3] = coll;
}
// coll_known: < 0 = evaluate the collision measure from F; 0 / 1 = the caller has (same function, same F)
__device__ __forceinline__ void mdp_terms_one(const rover_config &c, const float *cmd_b, const float *action,
                                              const float *prev_action, int ep_len, const float *F, float *rew,
                                              bool *term, bool no_force = false, int coll_known = -1)
{
    const float L = (float)c.max_episode_length;
    const float d = sqrtf(cmd_b[0] * cmd_b[0] + cmd_b[1] * cmd_b[1]);
    const float angle = rv_atan2f(cmd_b[1], cmd_b[0]);
    rew[0] = (1.0f / (1.0f + (0.11f * d * d))) / L;
    rew[1] = (d < c.rew_success_threshold) ? (float)(c.max_episode_length - ep_len) / L : 0.0f;   // the REWARD entry's threshold (rover_env_cfg.py:136)
    {
        const float linear_diff = action[1] - prev_action[1];
        const float angular_diff = action[0] - prev_action[0];
        float ap = (angular_diff * 3.0f > 0.05f) ? (angular_diff * 3.0f) * (angular_diff * 3.0f) : 0.0f;
        float lp = (linear_diff * 3.0f > 0.05f) ? (linear_diff * 3.0f) * (linear_diff * 3.0f) : 0.0f;
        ap = ap * ap;
        lp = lp * lp;
        rew[2] = (ap + lp) / L;
    }
    rew[3] = (fabsf(angle) > 2.0f) ? fabsf(angle) / L : 0.0f;
    rew[4] = (action[0] < 0.0f) ? (float)(1.0 / (double)c.max_episode_length) : 0.0f;
    const bool coll = coll_known >= 0 ? coll_known != 0 : (no_force ? false : collision_measure(F) > 1.0f);  // hard-coded 1, `threshold` ignored (B-8)
    rew[5] = coll ? 1.0f : 0.0f;
    rew[6] = (d > c.rew_far_threshold) ? 1.0f : 0.0f;   // rover_env_cfg.py:162
    term[0] = ep_len >= c.max_episode_length;
    term[1] = d < c.success_threshold;
    term[2] = d > c.far_threshold;
    term[3] = coll;
}

// ------------------------------------------------------------------------------------------------ (a3, a5) dynamics
// Reduced rover model; mirrors oracle/rover_oracle.c physics_substep operation for operation.
// The per-wheel pieces below are shared by the two mappings of the step kernel:
//   "lane"  : one env per lane, the six wheels looped inside the lane            (throughput mapping, large N)
//   "group" : eight lanes per env -- slots [FL, CL, FR, CR, RL, RR, -, -] --, cross-wheel sums by xor butterflies
//             (latency mapping, small N: 8x more waves, ~6x shorter dependent instruction stream per env)
// Both evaluate cross-wheel sums in the tree ((s0+s1)+(s2+s3)) + ((s4+s5)+(0+0)), so they agree bit for bit.
struct Contact {
    float n[3], t[3], s[3];
    float jn_a[3], jt_a[3], js_a[3];  // angular Jacobians R^T (r x dir), in the BODY frame (diagonal inertia)
    float jn_b, jt_b, js_b;
    float mn, mt, ms;
    float a_nt, a_ns, a_ts;  // split-mass coupling between the wheel's own rows
    float bias;
    float ln, lt, ls;
    float obst;
};

#define RV_SPLIT_C 3.0f  // mass splitting: a contact solves against 1/3 of the chassis (six share it; round 5: 3 instead of the textbook
                         // 6 -- the angular / bogie terms dominate a row's split inverse mass, see oracle/rover_oracle.c SPLIT_C)
#define RV_SPLIT_B 2.0f  // contacts sharing one bogie
#define RV_TREE8(a) ((((a)[0] + (a)[1]) + ((a)[2] + (a)[3])) + (((a)[4] + (a)[5]) + ((a)[6] + (a)[7])))

// 1 / sqrt(x) for a positive normal x by Newton's iteration from the classic exponent-halving first guess: 12 instructions
// where the correctly rounded sqrtf + division take ~30 (three of them per physics substep).  Relative error <= ~2 ulp
// (3.4e-2 -> 1.8e-3 -> 4.7e-6 -> 3e-11 before rounding).  The oracle runs the identical sequence, so the two still agree
// bit for bit; used for the unit normal, the tangent and the quaternion normalisation of the rover model only.
__device__ __forceinline__ float rv_rsqrtf(float x)
{
    float y = __uint_as_float(0x5f3759dfu - (__float_as_uint(x) >> 1));
    const float hx = 0.5f * x;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t = hx * y;
        y = y * fmaf(-t, y, 1.5f);
    }
    return y;
}
// fused-multiply-add forms of the small vector helpers (physics only; the oracle uses the identical sequences)
__device__ __forceinline__ float dot3f(const float *a, const float *b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
__device__ __forceinline__ void cross3f(const float *a, const float *b, float *o)
{
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
__device__ __forceinline__ void mat_vecf(const float R[3][3], const float *v, float *o)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = fmaf(R[i][2], v[2], fmaf(R[i][1], v[1], R[i][0] * v[0]));
}
__device__ __forceinline__ void mat_tvecf(const float R[3][3], const float *v, float *o)  // o = R^T v
{
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = fmaf(R[2][i], v[2], fmaf(R[1][i], v[1], R[0][i] * v[0]));
}

// implicit-PD steering joint (kp 8000, kd 1000, effort 12, rate 6; aau_rover_simple.py:43-49)
__device__ __forceinline__ void steer_joint(const StepConsts &k, float target, float &q, float &qd)
{
    const float q0 = q, qd0 = qd;
    const float e = target - q0;
    float v = fmaf(k.steer_hkp, e, RV_STEER_INERTIA * qd0) * k.steer_den_inv;
    const float tau = (v - qd0) * k.steer_i_over_h;
    if (tau > RV_STEER_EFFORT) v = qd0 + k.steer_dv_max;
    if (tau < -RV_STEER_EFFORT) v = qd0 - k.steer_dv_max;
    v = clamp_med3(v, -RV_STEER_VLIM, RV_STEER_VLIM);
    float x = fmaf(k.h, v, q0);
    if (x > RV_STEER_QLIM) { x = RV_STEER_QLIM; v = 0.0f; }
    if (x < -RV_STEER_QLIM) { x = -RV_STEER_QLIM; v = 0.0f; }
    q = x;
    qd = v;
}

// implicit-PD wheel motor about q* = 0 with velocity target (kp 100, kd 4000, effort 12, rate 6; :50-56)
__device__ __forceinline__ void wheel_motor(const StepConsts &k, float target, float lt, float &q, float &qd)
{
    const float q0 = q, qd0 = qd;
    const float tgt = clamp_med3(target, -RV_WHEEL_VLIM, RV_WHEEL_VLIM);
    const float tau_ext = -RV_WHEEL_CONTACT_RADIUS * lt * k.inv_h;
    const float drive = fmaf(RV_WHEEL_KD, tgt, tau_ext - RV_WHEEL_KP * q0);
    float v = fmaf(k.h, drive, RV_WHEEL_INERTIA * qd0) * k.wheel_den_inv;
    const float tau = fmaf(RV_WHEEL_KD, tgt - v, RV_WHEEL_KP * (0.0f - fmaf(k.h, v, q0)));
    if (tau > RV_WHEEL_EFFORT) v = fmaf(RV_WHEEL_EFFORT + tau_ext, k.wheel_h_over_i, qd0);
    if (tau < -RV_WHEEL_EFFORT) v = fmaf(-RV_WHEEL_EFFORT + tau_ext, k.wheel_h_over_i, qd0);
    v = clamp_med3(v, -RV_WHEEL_VLIM, RV_WHEEL_VLIM);
    float x = fmaf(k.h, v, q0);
    if (x > RV_TWO_PI_F) x -= RV_TWO_PI_F;  // PhysX revolute joints report a wrapped position
    if (x < -RV_TWO_PI_F) x += RV_TWO_PI_F;
    q = x;
    qd = v;
}

// constants of a wheel's bogie arm (functions of the slot only): d0 = wheel centre - pivot, ax x d0, ax . d0
struct ArmConsts {
    float d0[3], axd[3], ad;
};
__device__ __forceinline__ ArmConsts make_arm(const float *wb, const float *P, const float *ax)
{
    ArmConsts a;
#pragma unroll
    for (int i = 0; i < 3; ++i) a.d0[i] = wb[i] - P[i];
    cross3f(ax, a.d0, a.axd);
    a.ad = dot3f(ax, a.d0);
    return a;
}
// cfg.mass_model = 1: the weight of a bogie's SUBTREE (beam + steer links + wheels) acts at its own centre of mass c_j: generalised
// gravity force on the bogie coordinate Q_j = -m_j g z . R (ax x arm_j), arm_j = c_j - P rotated by the bogie angle, with
// ax x arm = (ax (ax . d0) - d0) sin q + (ax x d0) cos q for d0 = c_j(q = 0) - P.  `sub` = make_arm(c_j(0), P, ax); gq = K.bogie_gq[j].
// Returns the bogie rate after the gravity impulse of one substep (oracle/rover_oracle.c physics_substep, same operations).
struct SubConsts {
    float e1[3], e2[3];   // ax (ax . d0) - d0 and ax x d0: the coefficients of sin q and cos q in ax x arm
};
__device__ __forceinline__ SubConsts make_sub(const float *cj, const float *P, const float *ax)
{
    const ArmConsts a = make_arm(cj, P, ax);
    SubConsts s;
#pragma unroll
    for (int i = 0; i < 3; ++i) { s.e1[i] = fmaf(ax[i], a.ad, -a.d0[i]); s.e2[i] = a.axd[i]; }
    return s;
}
__device__ __forceinline__ float bogie_gravity(const SubConsts &sub, const float R[3][3], float sb, float cb, float gq, float bd)
{
    float vb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) vb[i] = fmaf(sub.e1[i], sb, sub.e2[i] * cb);
    const float vwz = fmaf(R[2][2], vb[2], fmaf(R[2][1], vb[1], R[2][0] * vb[0]));
    return fmaf(gq, vwz, bd);
}
__device__ __forceinline__ void bogie_sincos(float bq, float *sb, float *cb)
{
    // the small-angle form for every lane, always right for states the integrator produced (|bq| <= 10 deg); the general form only
    // in a wave that holds some other angle (a state written by the caller) -- a wave-uniform branch that is not taken, instead of
    // a per-lane if / else whose unused side is jumped over in every substep (a taken branch costs one wave per SIMD ~15 cycles)
    rv_sincosf_small(bq, sb, cb);
    const bool wide = !(fabsf(bq) < 0.75f);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(wide) != 0ull, 0)) {
        float s2, c2;
        rv_sincosf(bq, &s2, &c2);
        if (wide) { *sb = s2; *cb = c2; }
    }
}
// One contact row of a wheel: angular Jacobian R^T (r x dir) in the body frame, bogie Jacobian dir . (ax x rp) with the
// unilateral lock of a bogie that sits on a stop (normal row: only against the stop; friction rows: fully), split effective mass
struct RowQ {
    float ja[3], jb, m;
};
__device__ __forceinline__ RowQ row_quantities(const StepConsts &k, const float R[3][3], const float *r, const float *axrp,
                                               const float *dir, bool friction_row, bool at_hi, bool at_lo, float b_winv)
{
    RowQ o;
    float x[3];
    cross3f(r, dir, x);
    mat_tvecf(R, x, o.ja);
    float jb = dot3f(dir, axrp);
    const bool lock = friction_row ? (at_hi || at_lo) : ((at_hi && jb > 0.0f) || (at_lo && jb < 0.0f));
    if (lock) jb = 0.0f;
    o.jb = jb;
    // mass splitting: every contact sees 1/3 of the chassis (RV_SPLIT_C) and 1/2 of its bogie
    o.m = 1.0f / (RV_SPLIT_C * k.inv_m + RV_SPLIT_C * wdot3(o.ja, k.inv_I) + RV_SPLIT_B * (jb * jb * b_winv));
    return o;
}
// split-mass coupling between two rows of the same wheel (symmetric in its arguments, bit for bit)
__device__ __forceinline__ float row_coupling(const StepConsts &k, const RowQ &a, const RowQ &b, float b_winv)
{
    return RV_SPLIT_C * wdot3x(a.ja, b.ja, k.inv_I) + RV_SPLIT_B * (a.jb * b.jb * b_winv);
}
__device__ __forceinline__ float dpp_ror8(float x);
// Contact report of a LINK body (bogie / steer link; articulation.py:13-27 adds the obstacle mesh as report pair of every
// sensor body): vertical penalty force at one sample point p0 (body frame, zero bogie angle) that rides on the bogie with
// pivot P / axis ax at angle bq -- non-zero when the point is below the terrain surface where the obstacle layer is present.
// Report only (collision_with_obstacles ends the episode in the same step): the dynamics do not see it.
// Split in two so that the group mapping can issue the point's eight gathers BEFORE the wheel's geometry (their latency
// then overlaps the wheel's own terrain sample instead of following it): link_point_fetch -> LinkSample, link_point_eval.
struct LinkSample {
    float h00, h01, h10, h11, o00, o01, o10, o11, fx, fy, z;
};
__device__ __forceinline__ LinkSample link_point_fetch(const RvParams &p, const float R[3][3], const float *pos, const float *P,
                                                       const float *ax, float bq, const float *p0, const float *bogie_sc = nullptr)
{
    float sb, cb;
    if (bogie_sc) { sb = bogie_sc[0]; cb = bogie_sc[1]; }   // (sin, cos)(bq) the caller already holds
    else bogie_sincos(bq, &sb, &cb);
    const float d0[3] = {p0[0] - P[0], p0[1] - P[1], p0[2] - P[2]};
    float axd[3], pt_b[3], tmp[3];
    cross3f(ax, d0, axd);
    const float ad = dot3f(ax, d0);
#pragma unroll
    for (int i = 0; i < 3; ++i) pt_b[i] = P[i] + fmaf(ax[i], ad * (1.0f - cb), fmaf(axd[i], sb, d0[i] * cb));
    mat_vecf(R, pt_b, tmp);
    // same cell arithmetic as terrain_sample
    float u = ((pos[0] + tmp[0]) - p.min_x) * p.inv_res;
    float v = ((pos[1] + tmp[1]) - p.min_y) * p.inv_res;
    u = clamp_med3(u, 0.0f, (float)(p.W - 1));
    v = clamp_med3(v, 0.0f, (float)(p.H - 1));
    int j0 = (int)u, i0 = (int)v;
    if (j0 > p.W - 2) j0 = p.W - 2;
    if (i0 > p.H - 2) i0 = p.H - 2;
    const size_t base = (size_t)i0 * p.W + j0;
    const float *q = p.height + base, *o = p.obstacle + base;
    LinkSample s;
    s.fx = u - (float)j0; s.fy = v - (float)i0; s.z = pos[2] + tmp[2];
    s.h00 = q[0]; s.h01 = q[1]; s.h10 = q[p.W]; s.h11 = q[p.W + 1];
    s.o00 = o[0]; s.o01 = o[1]; s.o10 = o[p.W]; s.o11 = o[p.W + 1];
    return s;
}
// force[3]: k x penetration along the obstacle surface's UNIT normal (-gx, -gy, 1) / sqrt(1 + gx^2 + gy^2): horizontal components
// from the slope of the surface patch under the point
__device__ __forceinline__ void link_point_eval(const LinkSample &s, float inv_res, float *force)
{
    const float dx0 = s.h01 - s.h00, dx1 = s.h11 - s.h10, dy0 = s.h10 - s.h00, dy1 = s.h11 - s.h01;
    const float hx0 = s.h00 + s.fx * dx0, hx1 = s.h10 + s.fx * dx1;
    const float hgt = hx0 + s.fy * (hx1 - hx0);
    const float gx = (dx0 + s.fy * (dx1 - dx0)) * inv_res;
    const float gy = (dy0 + s.fx * (dy1 - dy0)) * inv_res;
    const float o0 = s.o00 + s.fx * (s.o01 - s.o00), o1 = s.o10 + s.fx * (s.o11 - s.o10);
    const float obst = o0 + s.fy * (o1 - o0);
    const float pen = hgt - s.z;
    // k x (vertical) penetration along the UNIT normal: the magnitude does not grow with the slope of the face
    const float fk = (obst > RV_OBSTACLE_EPS && pen > 0.0f) ? RV_LINK_STIFFNESS * pen : 0.0f;
    const float fz = fk * rv_rsqrtf(fmaf(gx, gx, fmaf(gy, gy, 1.0f)));
    force[0] = -gx * fz;
    force[1] = -gy * fz;
    force[2] = fz;
}
// the obstacle-layer height of a sample (terrain_sample<true>'s `obst`: same cell, same weights, same operations)
__device__ __forceinline__ float link_sample_obstacle(const LinkSample &s)
{
    const float o0 = s.o00 + s.fx * (s.o01 - s.o00), o1 = s.o10 + s.fx * (s.o11 - s.o10);
    return o0 + s.fy * (o1 - o0);
}
__device__ __forceinline__ void link_point_force(const RvParams &p, const float R[3][3], const float *pos, const float *P,
                                                 const float *ax, float bq, const float *p0, float *force)
{
    link_point_eval(link_point_fetch(p, R, pos, P, ax, bq, p0), p.inv_res, force);
}
// one component of the rows of the seven link bodies from the twelve point forces lf[slot][role]: fixed summation order
__device__ __forceinline__ void link_body_forces(const float lf[6][2], float *Fb /* 7 */)
{
    Fb[0] = (lf[0][1] + lf[1][0]) + lf[1][1];   // FL_Boogie
    Fb[1] = (lf[2][1] + lf[3][0]) + lf[3][1];   // FR_Boogie
    Fb[2] = lf[4][1] + lf[5][1];                // R_Boogie
    Fb[3] = lf[0][0]; Fb[4] = lf[2][0]; Fb[5] = lf[4][0]; Fb[6] = lf[5][0];   // FL, FR, RL, RR steer
}
// contact geometry, Jacobians, split effective masses and bias of ONE wheel.  GROUP_ROLE: the 16-lanes-per-env mapping, where
// the two role lanes of a wheel slot share the row work: role A derives the normal row, role B the longitudinal one, both
// the lateral one, and one row_ror:8 exchange hands each lane the row it did not compute (same arithmetic per quantity).
template <bool WANT_OBST, bool GROUP_ROLE = false>
__device__ __forceinline__ void wheel_geometry(const RvParams &p, const StepConsts &k, const float R[3][3], const float *pos,
                                               const float *com_w, const ArmConsts &arm, const float *P, const float *ax,
                                               float b_winv, float bq, bool at_hi, bool at_lo, bool steerable, float steer_q,
                                               Contact &ct, bool role_b = false, const float *bogie_sc = nullptr)
{
    const float *d0 = arm.d0, *axd = arm.axd;
    const float ad = arm.ad;
    float sb, cb;
    // bogie_sc: (sin, cos)(bq) the caller already holds (the group mapping evaluates them once per substep for the subtree's
    // weight: hipcc does not merge two evaluations that each sit behind their own |bq| < 0.75 branch -- 33 instructions per substep)
    if (bogie_sc) { sb = bogie_sc[0]; cb = bogie_sc[1]; }
    else bogie_sincos(bq, &sb, &cb);   // the small-angle form always, for states the integrator produced (|bq| <= 10 deg)
    float cen_b[3], cen_w[3], piv_w[3], ax_w[3], tmp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) cen_b[i] = P[i] + fmaf(ax[i], ad * (1.0f - cb), fmaf(axd[i], sb, d0[i] * cb));
    mat_vecf(R, cen_b, tmp);
#pragma unroll
    for (int i = 0; i < 3; ++i) cen_w[i] = pos[i] + tmp[i];
    mat_vecf(R, P, tmp);
#pragma unroll
    for (int i = 0; i < 3; ++i) piv_w[i] = pos[i] + tmp[i];
    mat_vecf(R, ax, ax_w);
    float hgt, gx, gy;
    ct.obst = 0.0f;
    terrain_sample<WANT_OBST>(p, cen_w[0], cen_w[1], hgt, gx, gy, ct.obst);
    const float inv = rv_rsqrtf(fmaf(gx, gx, fmaf(gy, gy, 1.0f)));
    ct.n[0] = -gx * inv; ct.n[1] = -gy * inv; ct.n[2] = inv;
    const float gap = fmaf(cen_w[2] - hgt, ct.n[2], -RV_WHEEL_CONTACT_RADIUS);
    float cp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) cp[i] = fmaf(-RV_WHEEL_CONTACT_RADIUS, ct.n[i], cen_w[i]);
    float fwd_b[3] = {1.0f, 0.0f, 0.0f}, fwd[3];
    if (steerable) rv_sincosf(steer_q, &fwd_b[1], &fwd_b[0]);
    mat_vecf(R, fwd_b, fwd);
    const float fn = dot3f(fwd, ct.n);
#pragma unroll
    for (int i = 0; i < 3; ++i) ct.t[i] = fmaf(-fn, ct.n[i], fwd[i]);
    const float tl = dot3f(ct.t, ct.t);
    const float tinv = rv_rsqrtf(fmaxf(tl, 1.0e-12f));  // fmaxf = tl > eps ? tl : eps for finite tl
#pragma unroll
    for (int i = 0; i < 3; ++i) ct.t[i] *= tinv;
    cross3f(ct.n, ct.t, ct.s);
    float r[3], rp[3], x[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { r[i] = cp[i] - com_w[i]; rp[i] = cp[i] - piv_w[i]; }
    cross3f(ax_w, rp, x);   // bogie rows: ax . (rp x d) = d . (ax x rp) -- one cross product shared by the three directions
    {   // (one select: as an if / else hipcc built two exec-mask regions around two instructions each)
        const float sep = -gap * k.inv_h;                            // speculative contact while separated
        const float push = RV_BAUMGARTE * (-gap) * k.inv_h;
        ct.bias = gap > 0.0f ? sep : fminf(push, RV_MAX_DEPENETRATION_VEL);   // fminf = push < cap ? push : cap for finite push, one v_min_f32
    }
    if (!GROUP_ROLE) {
        const RowQ qn = row_quantities(k, R, r, x, ct.n, false, at_hi, at_lo, b_winv);
        const RowQ qt = row_quantities(k, R, r, x, ct.t, true, at_hi, at_lo, b_winv);
        const RowQ qs = row_quantities(k, R, r, x, ct.s, true, at_hi, at_lo, b_winv);
#pragma unroll
        for (int i = 0; i < 3; ++i) { ct.jn_a[i] = qn.ja[i]; ct.jt_a[i] = qt.ja[i]; ct.js_a[i] = qs.ja[i]; }
        ct.jn_b = qn.jb; ct.jt_b = qt.jb; ct.js_b = qs.jb;
        ct.mn = qn.m; ct.mt = qt.m; ct.ms = qs.m;
        ct.a_nt = row_coupling(k, qn, qt, b_winv);
        ct.a_ns = row_coupling(k, qn, qs, b_winv);
        ct.a_ts = row_coupling(k, qt, qs, b_winv);
    } else {
        // this lane's row (n for role A, t for role B), the lateral row, and the other role's row by DPP
        const float dx[3] = {role_b ? ct.t[0] : ct.n[0], role_b ? ct.t[1] : ct.n[1], role_b ? ct.t[2] : ct.n[2]};
        const RowQ qx = row_quantities(k, R, r, x, dx, role_b, at_hi, at_lo, b_winv);
        const RowQ qs = row_quantities(k, R, r, x, ct.s, true, at_hi, at_lo, b_winv);
        const float a_xs = row_coupling(k, qx, qs, b_winv);
        RowQ qy;
#pragma unroll
        for (int i = 0; i < 3; ++i) qy.ja[i] = dpp_ror8(qx.ja[i]);
        qy.jb = dpp_ror8(qx.jb);
        qy.m = dpp_ror8(qx.m);
        const float a_ys = dpp_ror8(a_xs);
#pragma unroll
        for (int i = 0; i < 3; ++i) { ct.jn_a[i] = role_b ? qy.ja[i] : qx.ja[i]; ct.jt_a[i] = role_b ? qx.ja[i] : qy.ja[i]; ct.js_a[i] = qs.ja[i]; }
        ct.jn_b = role_b ? qy.jb : qx.jb; ct.jt_b = role_b ? qx.jb : qy.jb; ct.js_b = qs.jb;
        ct.mn = role_b ? qy.m : qx.m; ct.mt = role_b ? qx.m : qy.m; ct.ms = qs.m;
        ct.a_nt = row_coupling(k, qx, qy, b_winv);
        ct.a_ns = role_b ? a_ys : a_xs;
        ct.a_ts = role_b ? a_xs : a_ys;
    }
}

// ---- solver arithmetic (operation for operation the one of oracle/rover_oracle.c, see the comment there) ---------------
// The generalised velocity is held as four CHANNEL PAIRS {linear, angular}: c0 = (v.x, w.x), c1 = (v.y, w.y),
// c2 = (v.z, w.z), c3 = (bogie rate, 0).  A wheel has two ROLES: A owns c0, c1; B owns c2, c3.  Every pair is one register
// pair worked on by packed fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32).  A role's channels in order:
// ch0 = V[0].x, ch1 = V[0].y, ch2 = V[1].x, ch3 = V[1].y.  Row velocities (round 5, 44 instead of 48 instructions per iteration):
//   normal row     pn = p.x + p.y,  p = fma2(Jn[1], V[1], fma2(Jn[0], V[0], Cn)),   Cn = (-bias, 0) in role A, (0, 0) in role B
//   t and s rows   (pt, ps) = ONE packed chain over ch0 .. ch3: fma2(Jts[c], (ch_c, ch_c), .) starting from Cts = (-rim speed, 0) / (0, 0)
//   un' = pn_A + pn_B = J_n . V - bias,  ut' = pt_A + pt_B = J_t . V - rim speed,  us = ps_A + ps_B
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
struct RoleRows {
    f2 Jn[2];                // normal row, by channel pair
    f2 Jts[4];               // (J_t, J_s) of channel ch0 .. ch3
    f2 Cn, Cts;              // constants of the chains
    f2 Mn[2], Mt[2], Ms[2];  // Jacobian pairs times the inverse mass pair of the channel
};
__device__ __forceinline__ void role_partials(const RoleRows &r, const f2 *V, float &pn, float &pt, float &ps)
{
    const f2 p = fma2(r.Jn[1], V[1], fma2(r.Jn[0], V[0], r.Cn));
    pn = p.x + p.y;
    f2 ts = fma2(r.Jts[0], (f2){V[0].x, V[0].x}, r.Cts);
    ts = fma2(r.Jts[1], (f2){V[0].y, V[0].y}, ts);
    ts = fma2(r.Jts[2], (f2){V[1].x, V[1].x}, ts);
    ts = fma2(r.Jts[3], (f2){V[1].y, V[1].y}, ts);
    pt = ts.x;
    ps = ts.y;
}
__device__ __forceinline__ void role_outputs(const RoleRows &r, float dn, float dt, float ds, f2 *o)
{
    const f2 dn2 = {dn, dn}, dt2 = {dt, dt}, ds2 = {ds, ds};
#pragma unroll
    for (int k = 0; k < 2; ++k) o[k] = fma2(r.Ms[k], ds2, fma2(r.Mt[k], dt2, r.Mn[k] * dn2));
}
// inverse mass pairs of the channels: A = {(1/m, 1/Ixx), (1/m, 1/Iyy)}, B = {(1/m, 1/Izz), (1/I_bogie, 0)}
// (written out row by row: arrays of pointers into `ct` would keep the struct in memory -- hipcc then promotes it to LDS and
// fetches the workgroup size from the dispatch packet in host memory, a 15 us stall)
__device__ __forceinline__ void make_role_row(const float *dir, const float *ja, float jb, bool role_b, const f2 &minv0,
                                              const f2 &minv1, f2 *J, f2 *M)
{
    const float j0x = role_b ? dir[2] : dir[0], j0y = role_b ? ja[2] : ja[0];
    const float j1x = role_b ? jb : dir[1], j1y = role_b ? 0.0f : ja[1];
    J[0] = (f2){j0x, j0y};
    J[1] = (f2){j1x, j1y};
    M[0] = J[0] * minv0;
    M[1] = J[1] * minv1;
}
// cw = the rim speed the (stiff) motor prescribes (role A carries it, and the row's bias, in its chain constants)
__device__ __forceinline__ void make_role(const Contact &ct, bool role_b, const f2 &minv0, const f2 &minv1, float cw, RoleRows &r)
{
    f2 Jt[2], Js[2];
    make_role_row(ct.n, ct.jn_a, ct.jn_b, role_b, minv0, minv1, r.Jn, r.Mn);
    make_role_row(ct.t, ct.jt_a, ct.jt_b, role_b, minv0, minv1, Jt, r.Mt);
    make_role_row(ct.s, ct.js_a, ct.js_b, role_b, minv0, minv1, Js, r.Ms);
    r.Jts[0] = (f2){Jt[0].x, Js[0].x};
    r.Jts[1] = (f2){Jt[0].y, Js[0].y};
    r.Jts[2] = (f2){Jt[1].x, Js[1].x};
    r.Jts[3] = (f2){Jt[1].y, Js[1].y};
    r.Cn = (f2){role_b ? 0.0f : -ct.bias, 0.0f};
    r.Cts = (f2){role_b ? 0.0f : -cw, 0.0f};
}
// max / min / symmetric clamp as single instructions (v_max_f32 / v_min_f32 / v_med3_f32: total order with -0 < +0, which
// the oracle's max_ord / min_ord / med3_sym restate)
__device__ __forceinline__ float max_zero_ord(float a) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ float min_ord_s(float uniform, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(b)); return r; }
__device__ __forceinline__ float med3_sym(float x, float lim) { return __builtin_amdgcn_fmed3f(x, -lim, lim); }
// Gauss-Seidel over the wheel's own rows through the split-mass coupling terms
__device__ __forceinline__ void impulse_update(const StepConsts &k, Contact &ct, float mu, float un, float ut, float us,
                                               float &dn, float &dt, float &ds)
{
    const float ln = max_zero_ord(fmaf(-un, ct.mn, ct.ln));   // un = J_n . V - bias (the bias rides in the row chain)
    dn = ln - ct.ln;
    ct.ln = ln;
    const float lim = mu * ln;
    const float lmax = min_ord_s(k.lt_motor, lim);
    const float lt = med3_sym(fmaf(-fmaf(ct.a_nt, dn, ut), ct.mt, ct.lt), lmax);
    dt = lt - ct.lt;
    ct.lt = lt;
    const float ls = med3_sym(fmaf(-fmaf(ct.a_ts, dt, fmaf(ct.a_ns, dn, us)), ct.ms, ct.ls), lim);
    ds = ls - ct.ls;
    ct.ls = ls;
}

// chassis integration shared by both mappings: velocity cap, symplectic Euler, quaternion update
__device__ __forceinline__ void chassis_integrate(float h, const float R[3][3], float *v, float *wb, float *com_w,
                                                  float *pos, float *quat, float *linvel, float *angvel,
                                                  float (*R_next)[3] = nullptr, float *com_off_next = nullptr)
{
    constexpr float COM_B[3] = RV_COM_B_INIT;
    // speed cap.  sqrt is monotonic, so |v|^2 <= cap^2 implies sqrt(|v|^2) <= cap: the square root and the division are only
    // evaluated by waves in which some lane may exceed the cap (a wave-uniform branch; same results as testing every lane)
    const float v2 = dot3f(v, v);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(v2 > RV_MAX_LINEAR_VEL * RV_MAX_LINEAR_VEL) != 0ull, 0)) {
        const float sp = sqrtf(v2);
        if (sp > RV_MAX_LINEAR_VEL) {
            const float sc = RV_MAX_LINEAR_VEL / sp;
            v[0] *= sc; v[1] *= sc; v[2] *= sc;
        }
    }
    float w[3];
    mat_vecf(R, wb, w);  // angular velocity back to the world frame
#pragma unroll
    for (int i = 0; i < 3; ++i) { com_w[i] = fmaf(h, v[i], com_w[i]); linvel[i] = v[i]; angvel[i] = w[i]; }
    {
        const float qw = quat[0], qx = quat[1], qy = quat[2], qz = quat[3];
        const float hh = 0.5f * h;
        const float nw = qw + hh * (-w[0] * qx - w[1] * qy - w[2] * qz);
        const float nx = qx + hh * (w[0] * qw + w[1] * qz - w[2] * qy);
        const float ny = qy + hh * (w[1] * qw + w[2] * qx - w[0] * qz);
        const float nz = qz + hh * (w[2] * qw + w[0] * qy - w[1] * qx);
        const float inv = rv_rsqrtf(nw * nw + nx * nx + ny * ny + nz * nz);
        quat[0] = nw * inv; quat[1] = nx * inv; quat[2] = ny * inv; quat[3] = nz * inv;
    }
    float R2[3][3], com_off[3];
    quat_to_mat(quat, R2);
    mat_vecf(R2, COM_B, com_off);
#pragma unroll
    for (int i = 0; i < 3; ++i) pos[i] = com_w[i] - com_off[i];
    if (R_next) {  // the next substep starts from exactly these values (same quaternion): hand them over instead of re-forming
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            com_off_next[i] = com_off[i];
#pragma unroll
            for (int j = 0; j < 3; ++j) R_next[i][j] = R2[i][j];
        }
    }
}

__device__ __forceinline__ void bogie_integrate(float h, float bq0, float bdv, float &q_out, float &qd_out)
{
    // (selects, not branches: hipcc turned the nested ifs into four exec-mask regions per substep)
    const float q = fmaf(h, bdv, bq0);
    const bool hi = q > RV_BOGIE_QLIM, lo = q < -RV_BOGIE_QLIM;
    q_out = hi ? RV_BOGIE_QLIM : (lo ? -RV_BOGIE_QLIM : q);
    qd_out = ((hi && bdv > 0.0f) || (lo && bdv < 0.0f)) ? 0.0f : bdv;
}

// ---- "lane" mapping: one physics substep of one env, S = the env's state words in registers.  Same arithmetic as the
// group mapping; the prescaled Jacobians M = J * minv are re-formed where they are used (identical roundings) instead of
// being kept in registers for six wheels.
template <bool RECORD_FORCE>
__device__ __forceinline__ void physics_substep(const RvParams &p, const StepConsts &K, float *S, const float *steer_t,
                                                const float *wheel_t, float *F /* 39, only if RECORD_FORCE */)
{
    constexpr float COM_B[3] = RV_COM_B_INIT;
    constexpr float WHEEL_B[6][3] = RV_WHEEL_B_INIT;
    constexpr int WHEEL_STEER[6] = RV_WHEEL_STEER_INIT;
    constexpr int WHEEL_BODY[6] = RV_WHEEL_BODY_INIT;
    constexpr int SLOT_WHEEL[6] = RV_SLOT_WHEEL_INIT;
    constexpr float BOGIE_PIVOT[3][3] = RV_BOGIE_PIVOT_INIT;
    constexpr float BOGIE_AXIS[3][3] = RV_BOGIE_AXIS_INIT;

    const float h = K.h;
    const float mu = p.cfg.friction_mu;
    // ---- 1. steering joints
#pragma unroll
    for (int s = 0; s < 4; ++s) steer_joint(K, steer_t[s], S[ROVER_STEER_Q + s], S[ROVER_STEER_QD + s]);
    // ---- 2. chassis frame, gravity
    float R[3][3];
    quat_to_mat(S + ROVER_QUAT, R);
    float com_off[3], com_w[3];
    mat_vecf(R, COM_B, com_off);
#pragma unroll
    for (int i = 0; i < 3; ++i) com_w[i] = S[ROVER_POS + i] + com_off[i];
    float v[3] = {S[ROVER_LINVEL], S[ROVER_LINVEL + 1], fmaf(-RV_GRAVITY, h, S[ROVER_LINVEL + 2])};
    float w[3];
    mat_tvecf(R, S + ROVER_ANGVEL, w);
    float bd[3], bq[3];
    bool at_hi[3], at_lo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        bq[j] = S[ROVER_BOGIE_Q + j];
        bd[j] = S[ROVER_BOGIE_QD + j] * K.bogie_keep[j];
        at_hi[j] = bq[j] >= RV_BOGIE_QLIM - 1.0e-5f;
        at_lo[j] = bq[j] <= -RV_BOGIE_QLIM + 1.0e-5f;
    }
    float bsc[3][2];   // (sin, cos) of the three bogie angles: once per substep, for the subtrees' weights and the wheels' geometry
    {   // (cfg.mass_model = 0: the same operations with a zero coefficient, as in the group mapping and the oracle)
        constexpr float SUBTREE_COM[3][3] = RV_SUBTREE_COM_INIT;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float P[3] = {BOGIE_PIVOT[j][0], BOGIE_PIVOT[j][1], BOGIE_PIVOT[j][2]};
            const float ax[3] = {BOGIE_AXIS[j][0], BOGIE_AXIS[j][1], BOGIE_AXIS[j][2]};
            const float cj[3] = {SUBTREE_COM[j][0], SUBTREE_COM[j][1], SUBTREE_COM[j][2]};
            bogie_sincos(bq[j], &bsc[j][0], &bsc[j][1]);
            bd[j] = bogie_gravity(make_sub(cj, P, ax), R, bsc[j][0], bsc[j][1], p.cfg.mass_model == 1 ? K.bogie_gq[j] : 0.0f, bd[j]);
        }
    }
    // ---- 3. contact geometry (slot order), warm start
    Contact C[6];
    const f2 minvA0 = {K.inv_m, K.inv_I[0]}, minvA1 = {K.inv_m, K.inv_I[1]}, minvB0 = {K.inv_m, K.inv_I[2]};
    f2 VA[2] = {{v[0], w[0]}, {v[1], w[1]}}, VB0 = {v[2], w[2]};
    float bdy[3] = {0.0f, 0.0f, 0.0f};
    float oa[4][8], ob0[2][8], ob1[2][8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        oa[0][q] = oa[1][q] = oa[2][q] = oa[3][q] = 0.0f;
        ob0[0][q] = ob0[1][q] = ob1[0][q] = ob1[1][q] = 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int k = SLOT_WHEEL[s], j = s >> 1, si = WHEEL_STEER[k];
        const float wb[3] = {WHEEL_B[k][0], WHEEL_B[k][1], WHEEL_B[k][2]};
        const float P[3] = {BOGIE_PIVOT[j][0], BOGIE_PIVOT[j][1], BOGIE_PIVOT[j][2]};
        const float ax[3] = {BOGIE_AXIS[j][0], BOGIE_AXIS[j][1], BOGIE_AXIS[j][2]};
        const ArmConsts arm = make_arm(wb, P, ax);
        wheel_geometry<RECORD_FORCE>(p, K, R, S + ROVER_POS, com_w, arm, P, ax, K.b_winv[j], bq[j], at_hi[j], at_lo[j], si >= 0,
                                     S[ROVER_STEER_Q + (si >= 0 ? si : 0)], C[s], false, bsc[j]);
        C[s].ln = RV_WARM_START * S[ROVER_LAMBDA_N + k];
        C[s].lt = 0.0f;
        C[s].ls = 0.0f;
        const f2 minvB1 = {K.b_winv[j], 0.0f};
        RoleRows ra, rb;
        make_role(C[s], false, minvA0, minvA1, 0.0f, ra);
        make_role(C[s], true, minvB0, minvB1, 0.0f, rb);
        oa[0][s] = ra.Mn[0].x * C[s].ln; oa[1][s] = ra.Mn[0].y * C[s].ln;
        oa[2][s] = ra.Mn[1].x * C[s].ln; oa[3][s] = ra.Mn[1].y * C[s].ln;
        ob0[0][s] = rb.Mn[0].x * C[s].ln; ob0[1][s] = rb.Mn[0].y * C[s].ln;
        ob1[0][s] = rb.Mn[1].x * C[s].ln; ob1[1][s] = rb.Mn[1].y * C[s].ln;
    }
    // ---- 4. wheel-parallel projected Jacobi with mass splitting (iteration -1 = the warm-start contributions)
    for (int it = -1; it < p.cfg.solver_iterations; ++it) {
        if (it >= 0) {
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int k = SLOT_WHEEL[s], j = s >> 1;
                const f2 minvB1 = {K.b_winv[j], 0.0f};
                RoleRows ra, rb;
                const float cw = RV_WHEEL_CONTACT_RADIUS * S[ROVER_WHEEL_QD + k];   // rim speed prescribed by the (stiff) motor
                make_role(C[s], false, minvA0, minvA1, cw, ra);
                make_role(C[s], true, minvB0, minvB1, cw, rb);
                const f2 VB[2] = {VB0, {bd[j], bdy[j]}};
                float pna, pta, psa, pnb, ptb, psb;
                role_partials(ra, VA, pna, pta, psa);
                role_partials(rb, VB, pnb, ptb, psb);
                const float un = pna + pnb, ut = pta + ptb, us = psa + psb;
                float dn, dt, ds;
                impulse_update(K, C[s], mu, un, ut, us, dn, dt, ds);
                f2 o[2];
                role_outputs(ra, dn, dt, ds, o);
                oa[0][s] = o[0].x; oa[1][s] = o[0].y; oa[2][s] = o[1].x; oa[3][s] = o[1].y;
                role_outputs(rb, dn, dt, ds, o);
                ob0[0][s] = o[0].x; ob0[1][s] = o[0].y; ob1[0][s] = o[1].x; ob1[1][s] = o[1].y;
            }
        }
        VA[0].x += RV_TREE8(oa[0]); VA[0].y += RV_TREE8(oa[1]);
        VA[1].x += RV_TREE8(oa[2]); VA[1].y += RV_TREE8(oa[3]);
        VB0.x += RV_TREE8(ob0[0]); VB0.y += RV_TREE8(ob0[1]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            bd[j] += ob1[0][2 * j] + ob1[0][2 * j + 1];
            bdy[j] += ob1[1][2 * j] + ob1[1][2 * j + 1];
        }
    }
    v[0] = VA[0].x; v[1] = VA[1].x; v[2] = VB0.x;
    w[0] = VA[0].y; w[1] = VA[1].y; w[2] = VB0.y;
    // ---- 5. wheel motors, 6. obstacle contact report
    if (RECORD_FORCE) {
#pragma unroll
        for (int i = 0; i < ROVER_NUM_BODIES * 3; ++i) F[i] = 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int k = SLOT_WHEEL[s];
        wheel_motor(K, wheel_t[k], C[s].lt, S[ROVER_WHEEL_Q + k], S[ROVER_WHEEL_QD + k]);
        S[ROVER_LAMBDA_N + k] = C[s].ln;
        if (RECORD_FORCE) {
            if (C[s].obst > RV_OBSTACLE_EPS) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    F[WHEEL_BODY[k] * 3 + i] = fmaf(C[s].s[i], C[s].ls, fmaf(C[s].t[i], C[s].lt, C[s].n[i] * C[s].ln)) * K.inv_h;
            }
        }
    }
    if (RECORD_FORCE) {   // link bodies (bogies, steer links): pose of the substep's start, like the wheel rows
        constexpr float LINK_POINT[6][2][3] = RV_LINK_POINT_INIT;
        float lf3[3][6][2], Fb[7];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int j = s >> 1;
            const float P[3] = {BOGIE_PIVOT[j][0], BOGIE_PIVOT[j][1], BOGIE_PIVOT[j][2]};
            const float ax[3] = {BOGIE_AXIS[j][0], BOGIE_AXIS[j][1], BOGIE_AXIS[j][2]};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float p0[3] = {LINK_POINT[s][r][0], LINK_POINT[s][r][1], LINK_POINT[s][r][2]};
                float f3[3];
                link_point_force(p, R, S + ROVER_POS, P, ax, bq[j], p0, f3);
                lf3[0][s][r] = f3[0]; lf3[1][s][r] = f3[1]; lf3[2][s][r] = f3[2];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            link_body_forces(lf3[i], Fb);
#pragma unroll
            for (int b = 0; b < 7; ++b) F[b * 3 + i] = Fb[b];
        }
    }
    // ---- 7. integrate
    chassis_integrate(h, R, v, w, com_w, S + ROVER_POS, S + ROVER_QUAT, S + ROVER_LINVEL, S + ROVER_ANGVEL);
#pragma unroll
    for (int j = 0; j < 3; ++j) bogie_integrate(h, bq[j], bd[j], S[ROVER_BOGIE_Q + j], S[ROVER_BOGIE_QD + j]);
}

// ---- "group" mapping: SIXTEEN lanes per env = 8 wheel slots [FL, CL, FR, CR, RL, RR, -, -] x 2 roles.  Lane l of a
// 16-lane row: slot = l & 7, role = l >> 3 (A: lanes 0-7 own the channel pairs (v.x, w.x), (v.y, w.y); B: lanes 8-15 own
// (v.z, w.z), (bogie rate, 0)).  Chassis state and the wheel's contact frame are replicated in the two lanes of a slot.
struct GroupLane {
    // chassis (identical in the 16 lanes of a group)
    float pos[3], quat[4], linvel[3], angvel[3];
    float R[3][3], com_off[3];  // rotation matrix of `quat` and R * COM_B, carried from substep to substep
    // this lane's bogie / steer joint / wheel
    float bq, bqd, sq, sqd, wq, wqd, lam;
    float steer_t, wheel_t;
    // constants of this lane's slot
    float P[3], ax[3], b_winv, bogie_keep;
    float bogie_gq;   // K.bogie_gq of the lane's bogie (cfg.mass_model = 1)
    SubConsts sub;    // arm of the bogie subtree's centre of mass about the pivot (cfg.mass_model = 1)
    float lp[3];   // this lane's link-body sample point
    ArmConsts arm;
    f2 minv0, minv1;  // inverse mass pairs of the lane's two channels (negated in idle slot 7, see physics_substep_group)
    bool steerable, wheel_active, role_b;
};

// cross-lane moves as DPP operands (no LDS traffic): quad_perm [1,0,3,2] / [2,3,0,1] = xor 1 / 2, row_half_mirror =
// lane i <-> 7 - i inside each group of 8, row_ror:8 = lane i <-> i ^ 8 inside each row of 16 (the other role of the slot)
__device__ __forceinline__ float dpp_ror8(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xF, 0xF, true)); }
// u_r = s_r(own role) + s_r(other role) for the three rows: 3 fused v_add_f32_dpp.  The block starts with the register
// written longest ago; s_nop 1 covers the VALU-write -> DPP-read hazard (2 wait states) of the first instruction.
__device__ __forceinline__ void cross_role_sum3(float &a, float &b, float &c)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(a), "+v"(b), "+v"(c));
}
// The role's four contribution registers summed over the 8 wheel slots of its half-row, ((s0+s1)+(s2+s3))+((s4+s5)+(s6+s7)),
// as 3 x 4 fused v_add_f32_dpp.  Register 2 of role B is the BOGIE channel, which is shared by the two wheels of one bogie
// only: levels 2 and 3 are masked off for lanes 8-15 (bank_mask 0x3) on registers 2 and 3.
__device__ __forceinline__ void slot_sum4(float &a0, float &a1, float &a2, float &a3)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
        "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x3 bound_ctrl:1"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}

// Phase stamps of the diagnostic builds (tools/build_diag.py); nothing in the product build.
#if defined(RV_K1_STAMP) || defined(RV_K1_LITE)
#include "rover_diag.inc"
#endif
#ifndef K1_STAMP
#define K1_STAMP(k) do { } while (0)
#define K1_STAMP_NOWAIT(k) do { } while (0)
#endif
#ifndef K1_LITE
#define K1_LITE(k) do { } while (0)
#endif
#ifndef K1_LITE_F
#define K1_LITE_F(k) do { } while (0)
#endif
// the projected-Jacobi iterations of one lane (wheel slot x role): see the arithmetic contract above RoleRows
__device__ __forceinline__ void solver_iteration_generic(const StepConsts &K, Contact &ct, const RoleRows &rr, f2 *V, float mu)
{
    float un, ut, us;
    role_partials(rr, V, un, ut, us);
    cross_role_sum3(un, ut, us);
    float dn, dt, ds;
    impulse_update(K, ct, mu, un, ut, us, dn, dt, ds);
    f2 o[2];
    role_outputs(rr, dn, dt, ds, o);
    float a0 = o[0].x, a1 = o[0].y, a2 = o[1].x, a3 = o[1].y;
    slot_sum4(a0, a1, a2, a3);
    V[0] += (f2){a0, a1};
    V[1] += (f2){a2, a3};
}

// The same iteration, hand-scheduled (two iterations per loop trip).  What the measurements on gfx950 say for ONE wave per
// SIMD (tools/ubench/bank_probe.hip): an instruction costs ~4.5 cycles of issue if it is 4 bytes (VOP1/VOP2), ~5.5 if it is
// 8 bytes (VOP3, packed fp32, DPP), whatever its dependences; s_nop 1 costs ~8; a register written by a packed op stalls a
// consumer in the next two slots.  So the loop is written for instruction COUNT: no moves (the accumulated impulses
// ping-pong between two registers over the two unrolled halves), no s_nop (the 2 wait states a DPP read needs after a VALU
// write are filled with independent work), VOP2 encodings wherever the operation allows, rows interleaved behind the
// packed ops.  Temporaries that are used both as 32-bit and as (even-aligned) 64-bit operands live in v[238:255].
#define RV_DPP_FULL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define RV_DPP_LANES_A " row_mask:0xf bank_mask:0x3 bound_ctrl:1\n\t"
// One iteration = 44 instructions (round 4: 48): the row constants ride in the chains, the t and s rows share one packed chain with
// the channel broadcast by op_sel, the normal row finishes first (its cross-role add needs two wait states after the VALU write of
// its operand: the last two links of the (t, s) chain fill them).
#define RV_BC_LO " op_sel_hi:[1,0,1]\n\t"                  /* src1.lo for both halves */
#define RV_BC_HI " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"   /* src1.hi for both halves */
#define RV_SOLVER_HALF(LN_IN, LN_OUT, LT_IN, LT_OUT, LS_IN, LS_OUT, FILLER)                                            \
    "v_pk_fma_f32 v[240:241], %[jn0], %[v0], %[cn]\n\t"                                                              \
    "v_pk_fma_f32 v[242:243], %[jts0], %[v0], %[cts]" RV_BC_LO                                                       \
    "v_pk_fma_f32 v[240:241], %[jn1], %[v1], v[240:241]\n\t"                                                         \
    "v_pk_fma_f32 v[242:243], %[jts1], %[v0], v[242:243]" RV_BC_HI                                                   \
    "v_add_f32 v240, v240, v241\n\t"                     /* own share of un'                                       */ \
    "v_pk_fma_f32 v[242:243], %[jts2], %[v1], v[242:243]" RV_BC_LO                                                   \
    "v_pk_fma_f32 v[242:243], %[jts3], %[v1], v[242:243]" RV_BC_HI                                                   \
    "v_add_f32_dpp v240, v240, v240 row_ror:8" RV_DPP_FULL   /* un' = J_n . V - bias                                  */ \
    "v_fma_f32 v245, -v240, %[mn], " LN_IN "\n\t"                                                                    \
    "v_add_f32_dpp v242, v242, v242 row_ror:8" RV_DPP_FULL   /* ut' = J_t . V - rim speed                             */ \
    "v_add_f32_dpp v243, v243, v243 row_ror:8" RV_DPP_FULL   /* us                                                    */ \
    "v_max_f32 " LN_OUT ", 0, v245\n\t"                    /* ln                                                    */ \
    "v_sub_f32 v246, " LN_OUT ", " LN_IN "\n\t"            /* dn                                                    */ \
    "v_mul_f32 v245, %[mu], " LN_OUT "\n\t"                /* lim = mu * ln                                         */ \
    "v_pk_mul_f32 v[252:253], %[mn0], v[246:247] op_sel_hi:[1,0]\n\t"                                                \
    "v_fmac_f32 v242, %[ant], v246\n\t"                    /* ut' + a_nt * dn                                       */ \
    "v_pk_mul_f32 v[254:255], %[mn1], v[246:247] op_sel_hi:[1,0]\n\t"                                                \
    "v_min_f32 v247, %[ltm], v245\n\t"                     /* lmax = min(lim, lt_motor)                             */ \
    "v_fmac_f32 v243, %[ans], v246\n\t"                    /* us + a_ns * dn                                        */ \
    "v_fma_f32 v244, -v242, %[mt], " LT_IN "\n\t"                                                                    \
    "v_med3_f32 " LT_OUT ", v244, -v247, v247\n\t"         /* lt                                                    */ \
    "v_sub_f32 v248, " LT_OUT ", " LT_IN "\n\t"            /* dt                                                    */ \
    "v_fmac_f32 v243, %[ats], v248\n\t"                    /* ... + a_ts * dt                                       */ \
    "v_pk_fma_f32 v[252:253], %[mt0], v[248:249], v[252:253] op_sel_hi:[1,0,1]\n\t"                                  \
    "v_fma_f32 v244, -v243, %[ms], " LS_IN "\n\t"                                                                    \
    "v_pk_fma_f32 v[254:255], %[mt1], v[248:249], v[254:255] op_sel_hi:[1,0,1]\n\t"                                  \
    "v_med3_f32 " LS_OUT ", v244, -v245, v245\n\t"         /* ls                                                    */ \
    "v_sub_f32 v250, " LS_OUT ", " LS_IN "\n\t"            /* ds                                                    */ \
    "v_pk_fma_f32 v[252:253], %[ms0], v[250:251], v[252:253] op_sel_hi:[1,0,1]\n\t"                                  \
    "v_pk_fma_f32 v[254:255], %[ms1], v[250:251], v[254:255] op_sel_hi:[1,0,1]\n\t"                                  \
    FILLER                                                                                                             \
    "v_add_f32_dpp v252, v252, v252 quad_perm:[1,0,3,2]" RV_DPP_FULL                                                   \
    "v_add_f32_dpp v253, v253, v253 quad_perm:[1,0,3,2]" RV_DPP_FULL                                                   \
    "v_add_f32_dpp v254, v254, v254 quad_perm:[1,0,3,2]" RV_DPP_FULL                                                   \
    "v_add_f32_dpp v255, v255, v255 quad_perm:[1,0,3,2]" RV_DPP_FULL                                                   \
    "v_add_f32_dpp v252, v252, v252 quad_perm:[2,3,0,1]" RV_DPP_FULL                                                   \
    "v_add_f32_dpp v253, v253, v253 quad_perm:[2,3,0,1]" RV_DPP_FULL                                                   \
    "v_add_f32_dpp v254, v254, v254 quad_perm:[2,3,0,1]" RV_DPP_LANES_A                                                \
    "v_add_f32_dpp v255, v255, v255 quad_perm:[2,3,0,1]" RV_DPP_LANES_A                                                \
    "v_add_f32_dpp v252, v252, v252 row_half_mirror" RV_DPP_FULL                                                       \
    "v_add_f32_dpp v253, v253, v253 row_half_mirror" RV_DPP_FULL                                                       \
    "v_add_f32_dpp v254, v254, v254 row_half_mirror" RV_DPP_LANES_A                                                    \
    "v_add_f32_dpp v255, v255, v255 row_half_mirror" RV_DPP_LANES_A                                                    \
    "v_pk_add_f32 %[v0], %[v0], v[252:253]\n\t"                                                                      \
    "v_pk_add_f32 %[v1], %[v1], v[254:255]\n\t"

// UNROLL = iterations per loop trip (2, 4 or 8): the impulses ping-pong between two register sets, so a trip is an even number of halves.
#define RV_SOLVER_ASM(TRIPS, BODY)                                                                                                \
    asm volatile(                                                                                                                 \
        ".p2align 5\n\t"                                                                                                        \
        "1:\n\t" BODY "s_cbranch_scc1 1b\n\t"                                                                                  \
        : [v0] "+v"(v0), [v1] "+v"(v1), [ln] "+v"(ln), [lt] "+v"(lt), [ls] "+v"(ls), [cnt] "+s"(TRIPS)                            \
        : [jn0] "v"(rr.Jn[0]), [jn1] "v"(rr.Jn[1]), [jts0] "v"(rr.Jts[0]), [jts1] "v"(rr.Jts[1]), [jts2] "v"(rr.Jts[2]),          \
          [jts3] "v"(rr.Jts[3]), [cn] "v"(rr.Cn), [cts] "v"(rr.Cts), [mn0] "v"(rr.Mn[0]), [mn1] "v"(rr.Mn[1]), [mt0] "v"(rr.Mt[0]), \
          [mt1] "v"(rr.Mt[1]), [ms0] "v"(rr.Ms[0]), [ms1] "v"(rr.Ms[1]), [mn] "v"(ct.mn), [mt] "v"(ct.mt), [ms] "v"(ct.ms),        \
          [ant] "v"(ct.a_nt), [ans] "v"(ct.a_ns), [ats] "v"(ct.a_ts), [mu] "s"(mu), [ltm] "s"(K.lt_motor)                         \
        : "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251",         \
          "v252", "v253", "v254", "v255", "scc")
#define RV_HALF_A(FILLER) RV_SOLVER_HALF("%[ln]", "v249", "%[lt]", "v251", "%[ls]", "v238", FILLER)
#define RV_HALF_B(FILLER) RV_SOLVER_HALF("v249", "%[ln]", "v251", "%[lt]", "v238", "%[ls]", FILLER)
#ifndef RV_SOLVER_UNROLL
#define RV_SOLVER_UNROLL 8   // measured, us per step at 32 iterations: 2 -> 40.65, 4 -> 40.33 (39.66 on the later build), 8 -> 39.50
#endif
__device__ __forceinline__ void solver_iterations_group(const StepConsts &K, Contact &ct, const RoleRows &rr, f2 *V, float mu,
                                                        int iterations)
{
    // (the loop head on a 32-byte boundary: with one wave per SIMD nothing hides an instruction fetch that straddles a line --
    // two builds that differed in ONE float constant measured 40.4 and 41.1 us per step because the loops had moved from
    // offsets 0 / 32 to 24 / 60 of their 64-byte lines; the padding is at most seven s_nop per substep)
    f2 v0 = V[0], v1 = V[1];
    float ln = ct.ln, lt = ct.lt, ls = ct.ls;
    int rest = iterations;
    if (RV_SOLVER_UNROLL == 8) {
        int octs = rest >> 3;
        rest &= 7;
        if (octs > 0)
            RV_SOLVER_ASM(octs, RV_HALF_A("s_sub_u32 %[cnt], %[cnt], 1\n\t") RV_HALF_B("s_nop 0\n\t") RV_HALF_A("s_nop 0\n\t") RV_HALF_B("s_nop 0\n\t")
                                    RV_HALF_A("s_nop 0\n\t") RV_HALF_B("s_nop 0\n\t") RV_HALF_A("s_nop 0\n\t") RV_HALF_B("s_cmp_lg_u32 %[cnt], 0\n\t"));
    }
    if (RV_SOLVER_UNROLL >= 4) {
        int quads = rest >> 2;
        rest &= 3;
        if (quads > 0)   // four iterations per trip: the taken branch at the end of a trip is a fetch bubble nothing hides
            RV_SOLVER_ASM(quads, RV_HALF_A("s_sub_u32 %[cnt], %[cnt], 1\n\t") RV_HALF_B("s_nop 0\n\t") RV_HALF_A("s_nop 0\n\t")
                                     RV_HALF_B("s_cmp_lg_u32 %[cnt], 0\n\t"));
    }
    int pairs = rest >> 1;
    if (pairs > 0) RV_SOLVER_ASM(pairs, RV_HALF_A("s_sub_u32 %[cnt], %[cnt], 1\n\t") RV_HALF_B("s_cmp_lg_u32 %[cnt], 0\n\t"));
    V[0] = v0;
    V[1] = v1;
    ct.ln = ln;
    ct.lt = lt;
    ct.ls = ls;
    if (iterations & 1) solver_iteration_generic(K, ct, rr, V, mu);
}

// LINK_ELSEWHERE: the lane's link-body sample point AND the obstacle-layer height under its wheel are evaluated by the lane's twin
// in the copy wave (one-launch kernel): no obstacle gathers here, Fw[0..2] hold the wheel's force unconditionally (the caller
// zeroes them where the twin found no obstacle layer -- `on ? x : 0` either way), Fw[3]
// is left alone here.
template <bool RECORD_FORCE, bool LINK_ELSEWHERE = false>
__device__ __forceinline__ void physics_substep_group(const RvParams &p, const StepConsts &K, GroupLane &g,
                                                      float *Fw /* [0..2]: this wheel's force, [3..5]: this lane's link-point force */,
                                                      int sidx = 0)
{
    K1_STAMP(2 + 3 * sidx);
    constexpr float COM_B[3] = RV_COM_B_INIT;
    const float h = K.h;
    const float mu = p.cfg.friction_mu;
    if (g.steerable) steer_joint(K, g.steer_t, g.sq, g.sqd);
    float R[3][3], com_w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        com_w[i] = g.pos[i] + g.com_off[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = g.R[i][j];
    }
    float v[3] = {g.linvel[0], g.linvel[1], fmaf(-RV_GRAVITY, h, g.linvel[2])};
    float w[3];
    mat_tvecf(R, g.angvel, w);
    const float bq = g.bq;
    float bd = g.bqd * g.bogie_keep;
    float bsc[2];   // (sin, cos) of the bogie angle: once per substep, for the subtree's weight and for the wheel's geometry
    bogie_sincos(bq, &bsc[0], &bsc[1]);
    // the subtree's weight on the bogie coordinate.  NO branch on cfg.mass_model: the lumped model runs the same ten instructions
    // with a zero coefficient (g.bogie_gq; the oracle does the same) -- a wave-uniform branch here cost 1.35 us per step (the
    // kernel sits at 256 VGPRs / ~100 SGPRs: the branch's extra live scalar state turned into scalar re-loads inside the substeps)
    bd = bogie_gravity(g.sub, R, bsc[0], bsc[1], g.bogie_gq, bd);
    const bool at_hi = bq >= RV_BOGIE_QLIM - 1.0e-5f, at_lo = bq <= -RV_BOGIE_QLIM + 1.0e-5f;
    Contact ct;
    LinkSample ls;
    if (RECORD_FORCE && !LINK_ELSEWHERE) ls = link_point_fetch(p, R, g.pos, g.P, g.ax, bq, g.lp, bsc);      // pose of the substep's start, like the wheel rows
    wheel_geometry<RECORD_FORCE && !LINK_ELSEWHERE, true>(p, K, R, g.pos, com_w, g.arm, g.P, g.ax, g.b_winv, bq, at_hi, at_lo, g.steerable, g.sq, ct, g.role_b, bsc);
    if (RECORD_FORCE && !LINK_ELSEWHERE) link_point_eval(ls, p.inv_res, Fw + 3);
    K1_STAMP(3 + 3 * sidx);
    ct.ln = RV_WARM_START * g.lam;
    ct.lt = 0.0f;
    ct.ls = 0.0f;
    // Idle slots 6, 7 shadow wheel slot 5 (same state, same contact frame, hence the same impulses) and need no zeroing: slot 7
    // applies its contributions with NEGATED inverse masses (g.idle_sign = -1), so the first level of every cross-slot sum
    // forms s6 + s7 = c + (-c) = +0 exactly -- the exact zero the oracle's tree has in those two places.
    // this lane's role: two channel pairs of the Jacobians (+ prescaled copies) and of the shared velocities
    const bool rb = g.role_b;
    RoleRows rr;
    {
        const float cw = RV_WHEEL_CONTACT_RADIUS * g.wqd;  // rim speed prescribed by the (stiff) motor
        make_role(ct, rb, g.minv0, g.minv1, cw, rr);
    }
    f2 V[2];
    V[0] = rb ? (f2){v[2], w[2]} : (f2){v[0], w[0]};
    V[1] = rb ? (f2){bd, 0.0f} : (f2){v[1], w[1]};
    {
        // warm start: contribution of the cached normal impulse alone
        const f2 l2 = {ct.ln, ct.ln};
        const f2 o0 = rr.Mn[0] * l2, o1 = rr.Mn[1] * l2;
        float a0 = o0.x, a1 = o0.y, a2 = o1.x, a3 = o1.y;
        slot_sum4(a0, a1, a2, a3);
        V[0] += (f2){a0, a1};
        V[1] += (f2){a2, a3};
    }
    solver_iterations_group(K, ct, rr, V, mu, p.cfg.solver_iterations);
    {
        // both roles need the full velocity again: fetch the other role's two pairs
        const float p0x = dpp_ror8(V[0].x), p0y = dpp_ror8(V[0].y), p1x = dpp_ror8(V[1].x), p1y = dpp_ror8(V[1].y);
        v[0] = rb ? p0x : V[0].x; w[0] = rb ? p0y : V[0].y;
        v[1] = rb ? p1x : V[1].x; w[1] = rb ? p1y : V[1].y;
        v[2] = rb ? V[0].x : p0x; w[2] = rb ? V[0].y : p0y;
        bd = rb ? V[1].x : p1x;
    }
    K1_STAMP(4 + 3 * sidx);
    wheel_motor(K, g.wheel_t, ct.lt, g.wq, g.wqd);
    g.lam = ct.ln;
    if (RECORD_FORCE) {
        const bool on = LINK_ELSEWHERE ? true : ct.obst > RV_OBSTACLE_EPS;
#pragma unroll
        for (int i = 0; i < 3; ++i) Fw[i] = on ? fmaf(ct.s[i], ct.ls, fmaf(ct.t[i], ct.lt, ct.n[i] * ct.ln)) * K.inv_h : 0.0f;
    }
    chassis_integrate(h, R, v, w, com_w, g.pos, g.quat, g.linvel, g.angvel, g.R, g.com_off);
    bogie_integrate(h, bq, bd, g.bq, g.bqd);
}

// ------------------------------------------------------------------------------------------------ (a7, a8) reset
// Injected draws (rover_reset_with_draws): the uniforms the reference drew from torch's generator (recorded in
// tests/golden/reset.npz) take the place of the Philox draws -- same code path otherwise.
struct ResetDraws {
    int32_t spawn_row;     // randomizations.py:22   randperm(len(table))[:k]
    float yaw_u;           // :30                    rand(k)
    const float *theta_u;  // terrain_importer.py:169 rand(len(env_ids)) of every rejection round, in order
    float heading_u;       // :93-95                 uniform_(lo, hi) = u * (hi - lo) + lo
};

// target on the 9 m circle around (ox, oy) with rejection on the safe rock mask (terrain_importer.py:134-175)
__device__ __forceinline__ void sample_target(const RvParams &p, float ox, float oy, uint32_t gid, uint32_t count, const float *inj_theta,
                                              float &tx_out, float &ty_out, float &tz_out)
{
    const rover_config &c = p.cfg;
    float tx = 0.0f, ty = 0.0f;
    int tries = 0;
    bool done = false;
    uint32_t r[4] = {0u, 0u, 0u, 0u};
    while (!done) {
        float u;
        if (inj_theta) {
            u = inj_theta[tries];
        } else {
            if ((tries & 3) == 0) philox4x32(gid, count, 1u + (uint32_t)(tries >> 2), 0u, c.seed_lo, c.seed_hi, r);
            const int w = tries & 3;
            u = u01(w == 0 ? r[0] : (w == 1 ? r[1] : (w == 2 ? r[2] : r[3])));
        }
        const float theta = u * 2.0f * RV_PI_F;
        tx = rv_cosf(theta) * c.target_distance + ox;
        ty = rv_sinf(theta) * c.target_distance + oy;
        int cx, cy;
        quirk_cell(p, tx, ty, cx, cy);
        ++tries;
        if (p.safe_mask[(size_t)cy * p.W + cx] != 1 || tries >= c.max_target_tries) done = true;
    }
    int cx, cy;
    quirk_cell(p, tx, ty, cx, cy);
    tx_out = tx;
    ty_out = ty;
    tz_out = p.lookup[(size_t)cy * p.W + cx] + 0.0f;
}
// heading ~ U(lo, hi) (terrain_importer.py:93-95); TerrainBasedPositionCommand._resample_command
__device__ __forceinline__ void resample_command(const RvParams &p, float *S, uint32_t gid, uint32_t count, float heading_u,
                                                 const float *inj_theta = nullptr)
{
    const rover_config &c = p.cfg;
    sample_target(p, S[ROVER_ENV_ORIGIN + 0], S[ROVER_ENV_ORIGIN + 1], gid, count, inj_theta, S[ROVER_TARGET_W + 0], S[ROVER_TARGET_W + 1],
                  S[ROVER_TARGET_W + 2]);
    S[ROVER_HEADING_CMD_W] = heading_u * (c.heading_hi - c.heading_lo) + c.heading_lo;
    S[ROVER_TIME_LEFT] = c.resample_time;
}

// reset_root_state_rover (randomizations.py:12-39) + ORBIT manager resets (RLTaskEnv._reset_idx), in two halves: everything
// that is DRAWN or looked up (spawn row, yaw, target with its rejection loop, heading) depends on (global id, reset count) and
// the terrain only -- not on the step under way -- so the one-launch kernel's copy wave evaluates it during the physics, for
// every env, and the step wave's reset is the assignments of reset_apply (no Philox, no dependent loads on its path).
struct ResetOutcome {
    float px, py, pz, qw, qz, tx, ty, tz, heading_cmd;
};
__device__ __forceinline__ void reset_draw(const RvParams &p, uint32_t gid, uint32_t count, const ResetDraws *inj, ResetOutcome &o)
{
    const rover_config &c = p.cfg;
    uint32_t r[4];
    philox4x32(gid, count, 0u, 0u, c.seed_lo, c.seed_hi, r);
    uint32_t row;
    if (inj) row = (uint32_t)inj->spawn_row;
    else if (c.spawn_draw == 1)   // distinct rows inside one reset batch (randomizations.py:22: a randperm prefix)
        row = (uint32_t)(((uint64_t)p.spawn_a * (uint64_t)(gid % (uint32_t)p.n_spawns) + (uint64_t)p.spawn_b) % (uint64_t)p.n_spawns);
    else row = r[0] % (uint32_t)p.n_spawns;
    o.px = p.spawns[3 * row + 0];
    o.py = p.spawns[3 * row + 1];
    o.pz = p.spawns[3 * row + 2] + c.reset_z_offset;
    const float angle = (inj ? inj->yaw_u : u01(r[1])) * 2.0f * RV_PI_F;
    o.qw = rv_cosf(angle / 2.0f);
    o.qz = rv_sinf(angle / 2.0f);
    sample_target(p, o.px, o.py, gid, count, inj ? inj->theta_u : nullptr, o.tx, o.ty, o.tz);
    const float heading_u = inj ? inj->heading_u : u01(r[2]);
    o.heading_cmd = heading_u * (c.heading_hi - c.heading_lo) + c.heading_lo;
}
__device__ __forceinline__ void reset_apply(const RvParams &p, float *S, const ResetOutcome &o)
{
    const rover_config &c = p.cfg;
    const uint32_t count = __float_as_uint(S[ROVER_RESET_COUNT]);
    S[ROVER_POS + 0] = o.px; S[ROVER_POS + 1] = o.py; S[ROVER_POS + 2] = o.pz;
    S[ROVER_QUAT + 0] = o.qw; S[ROVER_QUAT + 1] = 0.0f; S[ROVER_QUAT + 2] = 0.0f;
    S[ROVER_QUAT + 3] = o.qz;
    S[ROVER_ENV_ORIGIN + 0] = o.px; S[ROVER_ENV_ORIGIN + 1] = o.py; S[ROVER_ENV_ORIGIN + 2] = o.pz;
    if (c.reset_mode == 1) {
#pragma unroll
        for (int i = ROVER_LINVEL; i < ROVER_TARGET_W; ++i) S[i] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) S[ROVER_LAMBDA_N + k] = 0.0f;
    S[ROVER_ACTION] = S[ROVER_ACTION + 1] = 0.0f;
    S[ROVER_PREV_ACTION] = S[ROVER_PREV_ACTION + 1] = 0.0f;
#pragma unroll
    for (int i = 0; i < ROVER_NUM_REW; ++i) S[ROVER_EP_SUM + i] = 0.0f;
    S[ROVER_METRIC_POS] = 0.0f;
    S[ROVER_METRIC_HEAD] = 0.0f;
    S[ROVER_TARGET_W + 0] = o.tx; S[ROVER_TARGET_W + 1] = o.ty; S[ROVER_TARGET_W + 2] = o.tz;
    S[ROVER_HEADING_CMD_W] = o.heading_cmd;
    S[ROVER_TIME_LEFT] = c.resample_time;
    S[ROVER_EP_LEN] = __int_as_float(0);
    S[ROVER_RESET_COUNT] = __uint_as_float(count + 1u);
}
__device__ __forceinline__ void reset_one(const RvParams &p, float *S, uint32_t gid, const ResetDraws *inj = nullptr)
{
    ResetOutcome o;
    reset_draw(p, gid, __float_as_uint(S[ROVER_RESET_COUNT]), inj, o);
    reset_apply(p, S, o);
}

// CommandTerm.compute: metrics -> timer -> (resample) -> update (terrain_importer.py:97-106)
__device__ __forceinline__ void command_compute(const RvParams &p, float *S, uint32_t gid, float step_dt)
{
    const float heading_w = heading_of(S + ROVER_QUAT);   // the pose does not change inside this function
    {
        const float dx = S[ROVER_TARGET_W] - S[ROVER_POS], dy = S[ROVER_TARGET_W + 1] - S[ROVER_POS + 1],
                    dz = S[ROVER_TARGET_W + 2] - S[ROVER_POS + 2];
        S[ROVER_METRIC_POS] = sqrtf(dx * dx + dy * dy + dz * dz);
        S[ROVER_METRIC_HEAD] = fabsf(wrap_to_pi(S[ROVER_HEADING_CMD_W] - heading_w));
    }
    S[ROVER_TIME_LEFT] -= step_dt;
    if (S[ROVER_TIME_LEFT] <= 0.0f) {
        const uint32_t count = __float_as_uint(S[ROVER_RESET_COUNT]);
        uint32_t r[4];
        philox4x32(gid, count, 0u, 1u, p.cfg.seed_lo, p.cfg.seed_hi, r);
        resample_command(p, S, gid, count ^ 0x80000000u, u01(r[2]));
    }
    update_command_one(S + ROVER_POS, S + ROVER_QUAT, S + ROVER_TARGET_W, S[ROVER_HEADING_CMD_W], S + ROVER_CMD_B,
                       S + ROVER_HEADING_CMD_B, &heading_w);
}

// observation head [last_action(2), distance * 0.11, heading / pi] (ObservationCfg, rover_env_cfg.py:97-123)
__device__ __forceinline__ void write_obs_head(const RvParams &p, const float *S, float *__restrict__ obs, int e)
{
    const float cbx = S[ROVER_CMD_B], cby = S[ROVER_CMD_B + 1];
    float *o = obs + (size_t)e * p.obs_w;
    o[0] = S[ROVER_ACTION];
    o[1] = S[ROVER_ACTION + 1];
    o[2] = sqrtf(cbx * cbx + cby * cby) * p.cfg.obs_scale_distance;
    o[3] = rv_atan2f(cby, cbx) * p.cfg.obs_scale_heading;
}

// terrain window of the yaw-rotated ray pattern: rows [i_lo, i_lo + th), tw4 16-byte chunks per row starting at cell
// j_lo (a multiple of the cells per chunk: 4 for the fp32 heightfield, 8 for its exact int16 copy)
struct ScanWindow {
    float px, py, pz, cy, sy;
    int i_lo, j_lo, th, tw4;
    int interior;  // every ray is >= 2 cells inside the map and the window is not truncated: no per-ray bounds work
};
__device__ __forceinline__ ScanWindow scan_window(const RvParams &p, const float *pos, const float *quat)
{
    const rover_config &c = p.cfg;
    ScanWindow w;
    const float qw = quat[0], qx = quat[1], qy = quat[2], qz = quat[3];
    // yaw-only attachment (rover_env_cfg.py:81): (cos, sin)(yaw) straight from the quaternion
    const float a = 1.0f - 2.0f * (qy * qy + qz * qz);
    const float b = 2.0f * (qw * qz + qx * qy);
    const float inv = 1.0f / sqrtf(a * a + b * b);
    w.px = pos[0]; w.py = pos[1]; w.pz = pos[2];
    w.cy = a * inv;
    w.sy = b * inv;
    const float inv_res = p.inv_res;
    // window covered by the rotated pattern (+ slack), clamped to the map; the left edge is aligned down to a whole
    // 16-byte chunk so that every row can be staged with 16-byte loads
    const float hx = 0.5f * c.scan_size_x, hy = 0.5f * c.scan_size_y;
    const float ex = fabsf(w.cy) * hx + fabsf(w.sy) * hy, ey = fabsf(w.sy) * hx + fabsf(w.cy) * hy;
    int j_lo = (int)floorf((w.px - ex - p.min_x) * inv_res) - 1;
    int i_lo = (int)floorf((w.py - ey - p.min_y) * inv_res) - 1;
    int j_hi = (int)floorf((w.px + ex - p.min_x) * inv_res) + 2;
    int i_hi = (int)floorf((w.py + ey - p.min_y) * inv_res) + 2;
    j_lo = max(0, min(j_lo, p.W - 1)); j_hi = max(0, min(j_hi, p.W - 1));
    i_lo = max(0, min(i_lo, p.H - 1)); i_hi = max(0, min(i_hi, p.H - 1));
    const int cc = p.chunk_cells, sh = (cc == 8) ? 3 : 2;
    j_lo &= ~(cc - 1);
    w.i_lo = i_lo;
    w.j_lo = j_lo;
    w.th = min(i_hi - i_lo + 1, p.tile_dim);
    w.tw4 = min(min((j_hi - j_lo + cc) >> sh, p.tile_pitch >> sh), (p.W - j_lo + cc - 1) >> sh);   // 16-byte chunks per row
    const float m = 2.0f * p.res;
    w.interior = (w.px - ex > p.min_x + m) && (w.px + ex < p.x_max - m) && (w.py - ey > p.min_y + m) &&
                 (w.py + ey < p.y_max - m) && (i_hi - i_lo + 1 <= p.tile_dim) && (((j_hi - j_lo + cc) >> sh) <= (p.tile_pitch >> sh));
    return w;
}
__device__ __forceinline__ void write_scan_desc(const RvParams &p, const float *pos, const float *quat, int e)
{
    const ScanWindow w = scan_window(p, pos, quat);
    float4 *d = reinterpret_cast<float4 *>(p.scan_desc + (size_t)e * 8);
    d[0] = make_float4(w.px, w.py, w.pz, w.cy);
    d[1] = make_float4(w.sy, __int_as_float(w.i_lo), __int_as_float(w.j_lo), __int_as_float(w.th | (w.interior << 15) | (w.tw4 << 16)));
}

__device__ __forceinline__ float wave_sum(float x)
{
    // 64-lane butterfly: every lane ends with the same (order-deterministic) sum
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

// ================================================================================================ K1: step kernel
// PHASE 0: the whole env step.  PHASE 1 / 2: the step in two halves with the caller in between (the product runs phase 1 in the
// sixteen-lanes-per-env mapping, step_group_body<0, true>: no scratch; phase 1 here is its one-env-per-lane twin) -- the SLOW PATH for user-written
// reward / termination terms (rover_env_cfg.py:126-183 are tables of arbitrary `func=`; ORBIT's managers evaluate them on the
// state the physics left, BEFORE _reset_idx).  1 = action, physics, counters, the built-in terms and their episodic sums: state,
// reward, flags and forces are stored, nothing is reset.  2 = the rest of rover_env.py:89-99 for the reset mask the caller
// hands back (built-in OR user terminations): log contributions, reset, command update, observation head.  The built-in terms
// are re-evaluated in phase 2 for the log's termination counts -- a pure function of words phase 1 stored (stale command,
// actions, episode counter) and of the force rows.
template <int PHASE>
__device__ __forceinline__ void step_lane_body(const RvParams &p, float *__restrict__ state, const float *__restrict__ action,
                                               float *__restrict__ obs, float *__restrict__ reward, uint8_t *__restrict__ terminated,
                                               uint8_t *__restrict__ truncated, float *__restrict__ force, float *__restrict__ log_partial,
                                               const uint8_t *__restrict__ reset_mask)
{
    const int e_raw = blockIdx.x * 64 + threadIdx.x;
    const bool active = e_raw < p.n;
    const int e = active ? e_raw : p.n - 1;  // idle tail lanes shadow the last env and store nothing
    const int N = p.n;
    const rover_config &c = p.cfg;

    // physics words first (pose, velocities, joints, actions, contact cache); manager words after the physics so that
    // they do not occupy registers across the solver
    float S[ROVER_STATE_WORDS];
    float F[ROVER_NUM_BODIES * 3];
    if (PHASE == 2) {
#pragma unroll
        for (int i = 0; i < ROVER_STATE_WORDS; ++i) S[i] = state[(size_t)i * N + e];
#pragma unroll
        for (int i = 0; i < ROVER_NUM_BODIES * 3; ++i) F[i] = force[(size_t)i * N + e];
    } else {
#pragma unroll
        for (int i = 0; i < ROVER_TARGET_W; ++i) S[i] = state[(size_t)i * N + e];
#pragma unroll
        for (int i = ROVER_ACTION; i < ROVER_ACTION + 2; ++i) S[i] = state[(size_t)i * N + e];
#pragma unroll
        for (int i = ROVER_LAMBDA_N; i < ROVER_LAMBDA_N + 6; ++i) S[i] = state[(size_t)i * N + e];

        // rover_env.py:62 ActionManager.process_action
        S[ROVER_PREV_ACTION] = S[ROVER_ACTION];
        S[ROVER_PREV_ACTION + 1] = S[ROVER_ACTION + 1];
        const float2 a = reinterpret_cast<const float2 *>(action)[e];
        S[ROVER_ACTION] = a.x;
        S[ROVER_ACTION + 1] = a.y;
        float steer_m[4], wheel_m[6];
        {
            float processed[2], steer[4], wheel[6];
            ackermann_one(c, S + ROVER_ACTION, processed, steer, wheel);
            steer_m[0] = steer[0]; steer_m[1] = steer[3]; steer_m[2] = steer[1]; steer_m[3] = steer[2];  // FL, FR, RL, RR
            wheel_m[0] = wheel[1]; wheel_m[1] = wheel[5]; wheel_m[2] = wheel[0];                            // FL, FR, CL,
            wheel_m[3] = wheel[4]; wheel_m[4] = wheel[2]; wheel_m[5] = wheel[3];                            // CR, RL, RR
        }

        // rover_env.py:64-72 decimation loop; the contact report is the one of the last physics step
        const StepConsts &K = p.K;
        for (int s = 0; s < c.decimation - 1; ++s) physics_substep<false>(p, K, S, steer_m, wheel_m, nullptr);
        if (c.decimation > 0) {
            physics_substep<true>(p, K, S, steer_m, wheel_m, F);
        } else {
#pragma unroll
            for (int i = 0; i < ROVER_NUM_BODIES * 3; ++i) F[i] = 0.0f;
        }

#pragma unroll
        for (int i = ROVER_TARGET_W; i < ROVER_ACTION; ++i) S[i] = state[(size_t)i * N + e];
#pragma unroll
        for (int i = ROVER_TIME_LEFT; i < ROVER_LAMBDA_N; ++i) S[i] = state[(size_t)i * N + e];
        S[ROVER_RESET_COUNT] = state[(size_t)ROVER_RESET_COUNT * N + e];
        // :79
        S[ROVER_EP_LEN] = __int_as_float(__float_as_int(S[ROVER_EP_LEN]) + 1);
    }
    const int ep_len = __float_as_int(S[ROVER_EP_LEN]);
    // :82-86 terminations + rewards on the command of the PREVIOUS step (B-13)
    float rew[ROVER_NUM_REW];
    bool term[ROVER_NUM_TERM];
    mdp_terms_one(c, S + ROVER_CMD_B, S + ROVER_ACTION, S + ROVER_PREV_ACTION, ep_len, F, rew, term);
    const bool time_out = term[0];
    const bool term_any = term[1] | term[2] | term[3];
    const float step_dt = c.sim_dt * (float)c.decimation;
    float total = 0.0f;
    if (PHASE != 2) {
#pragma unroll
        for (int i = 0; i < ROVER_NUM_REW; ++i) {
            if (c.rew_weight[i] != 0.0f) {
                const float val = rew[i] * c.rew_weight[i] * step_dt;
                total += val;
                S[ROVER_EP_SUM + i] += val;
            }
        }
    }
    if (PHASE == 1) {   // everything up to the reset decision is on record; the caller's terms come next
        if (active) {
#pragma unroll
            for (int i = 0; i < ROVER_STATE_WORDS; ++i) state[(size_t)i * N + e] = S[i];
            reward[e] = total;
            terminated[e] = term_any ? 1 : 0;
            truncated[e] = time_out ? 1 : 0;
#pragma unroll
            for (int i = 0; i < ROVER_NUM_BODIES * 3; ++i) force[(size_t)i * N + e] = F[i];
        }
        return;
    }
    // :89-91 reset, with the episodic log contributions captured first
    const bool do_reset = PHASE == 2 ? reset_mask[e] != 0 : (term_any | time_out);
    float lg[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) lg[i] = 0.0f;
    if (do_reset && active) {
#pragma unroll
        for (int i = 0; i < ROVER_NUM_REW; ++i) lg[i] = S[ROVER_EP_SUM + i];
#pragma unroll
        for (int i = 0; i < ROVER_NUM_TERM; ++i) lg[7 + i] = term[i] ? 1.0f : 0.0f;
        lg[11] = S[ROVER_METRIC_POS];
        lg[12] = S[ROVER_METRIC_HEAD];
        lg[13] = 1.0f;
    }
    const uint32_t gid = (uint32_t)(p.env_id_offset + e);
    if (do_reset) reset_one(p, S, gid);
    // :93 command update
    command_compute(p, S, gid, step_dt);

    // wavefront reductions of the log partials (one row per wave) -- only in waves where some env reset
    if (__ballot(do_reset && active) != 0ull) {
#pragma unroll
        for (int i = 0; i < 14; ++i) lg[i] = wave_sum(lg[i]);
        if (threadIdx.x < 16) {
            float vsel = threadIdx.x == 15 ? __uint_as_float(p.step_tag) : 0.0f;
#pragma unroll
            for (int i = 0; i < 14; ++i) vsel = (threadIdx.x == i) ? lg[i] : vsel;
            log_partial[(size_t)blockIdx.x * ROVER_LOG_WORDS + threadIdx.x] = vsel;
        }
        if (threadIdx.x == 0) {
            p.log_counter[0] = 1u;           // "rows to reduce": a flag (tested against zero, returned to zero by the reduction)
            p.log_counter[1] = p.step_tag;   // the latest launch with resets
        }
    }

    if (active) {
        write_obs_head(p, S, obs, e);
        write_scan_desc(p, S + ROVER_POS, S + ROVER_QUAT, e);
#pragma unroll
        for (int i = 0; i < ROVER_STATE_WORDS; ++i) state[(size_t)i * N + e] = S[i];
        if (PHASE == 0) {
            reward[e] = total;
            terminated[e] = term_any ? 1 : 0;
            truncated[e] = time_out ? 1 : 0;
            if (force) {
#pragma unroll
                for (int i = 0; i < ROVER_NUM_BODIES * 3; ++i) force[(size_t)i * N + e] = F[i];
            }
        }
    }
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void rover_step_kernel(RvParams p, float *__restrict__ state,
                                                        const float *__restrict__ action, float *__restrict__ obs,
                                                        float *__restrict__ reward, uint8_t *__restrict__ terminated,
                                                        uint8_t *__restrict__ truncated, float *__restrict__ force,
                                                        float *__restrict__ log_partial)
{
    step_lane_body<0>(p, state, action, obs, reward, terminated, truncated, force, log_partial, nullptr);
}
__global__ __launch_bounds__(64) void rover_step_finish_kernel(RvParams p, float *__restrict__ state, float *__restrict__ obs,
                                                               float *__restrict__ force, float *__restrict__ log_partial,
                                                               const uint8_t *__restrict__ reset_mask)
{
    step_lane_body<2>(p, state, nullptr, obs, nullptr, nullptr, nullptr, force, log_partial, reset_mask);
}

// ---- per-slot constants of the "group" mapping: one 64-byte row per wheel slot, fetched with ONE level of loads
struct SlotConst {
    float wb[3], P[3], ax[3];
    int32_t k, j, si, body;   // wheel, bogie, steer joint (-1: none), contact-sensor body row
    float lp[2][3];           // link-body sample point of the role-A / role-B lane (RV_LINK_POINT_INIT)
    float sc[3];              // centre of mass of the bogie's subtree at q = 0 (RV_SUBTREE_COM_INIT)
    int32_t pad[2];
};
#define RV_SLOT_ROW(k_, j_, si_, body_, s_) \
    {{RV_WB_##k_}, {RV_BP_##j_}, {RV_BA_##j_}, k_, j_, si_, body_, RV_LP_##s_, {RV_SC_##j_}, {0, 0}}
#define RV_SC_0 0.27019f, 0.38237f, -0.06738f
#define RV_SC_1 0.27019f, -0.38237f, -0.06738f
#define RV_SC_2 -0.40167f, 0.0f, -0.07031f
#define RV_LP_0 {{0.44f, 0.3125f, -0.10f}, {0.29675f, 0.3075f, 0.0175f}}
#define RV_LP_1 {{0.08025f, 0.3055f, -0.0685f}, {0.007f, 0.3085f, -0.12f}}
#define RV_LP_2 {{0.44f, -0.3125f, -0.10f}, {0.29675f, -0.3075f, 0.0175f}}
#define RV_LP_3 {{0.08025f, -0.3055f, -0.0685f}, {0.007f, -0.3085f, -0.12f}}
#define RV_LP_4 {{-0.44f, 0.3125f, -0.10f}, {-0.3825f, 0.19625f, 0.0175f}}
#define RV_LP_5 {{-0.44f, -0.3125f, -0.10f}, {-0.3825f, -0.19625f, 0.0175f}}
#define RV_WB_0 0.44f, 0.3925f, -0.16699f
#define RV_WB_1 0.44f, -0.3925f, -0.16699f
#define RV_WB_2 0.007f, 0.3885f, -0.16699f
#define RV_WB_3 0.007f, -0.3885f, -0.16699f
#define RV_WB_4 -0.44f, 0.3925f, -0.16699f
#define RV_WB_5 -0.44f, -0.3925f, -0.16699f
#define RV_BP_0 0.1535f, 0.2225f, 0.03f
#define RV_BP_1 0.1535f, -0.2225f, 0.03f
#define RV_BP_2 -0.325f, 0.0f, 0.03f
#define RV_BA_0 0.0f, 1.0f, 0.0f
#define RV_BA_1 0.0f, -1.0f, 0.0f
#define RV_BA_2 1.0f, 0.0f, 0.0f
// slots [FL, CL, FR, CR, RL, RR, -, -] = wheels [0, 2, 1, 3, 4, 5]; the idle slots 6, 7 shadow slot 5
__device__ const SlotConst d_SLOT[8] = {
    RV_SLOT_ROW(0, 0, 0, 9, 0), RV_SLOT_ROW(2, 0, -1, 7, 1), RV_SLOT_ROW(1, 1, 1, 10, 2), RV_SLOT_ROW(3, 1, -1, 8, 3),
    RV_SLOT_ROW(4, 2, 2, 11, 4), RV_SLOT_ROW(5, 2, 3, 12, 5), RV_SLOT_ROW(5, 2, 3, 12, 5), RV_SLOT_ROW(5, 2, 3, 12, 5)};
static_assert(sizeof(SlotConst) == 96, "SlotConst row");
static float d_SLOT_host_sc(int j, int i)
{
    const float sc[3][3] = {{RV_SC_0}, {RV_SC_1}, {RV_SC_2}};
    return sc[j][i];
}
// host copy of the sample-point columns of d_SLOT (consistency check against RV_LINK_POINT_INIT in rover_model_constants)
static float d_SLOT_host_lp(int slot, int role, int i)
{
    const float lp[6][2][3] = {RV_LP_0, RV_LP_1, RV_LP_2, RV_LP_3, RV_LP_4, RV_LP_5};
    return lp[slot][role][i];
}

struct GroupIds {
    int slot, k, j, si, body;
    bool wheel_active, role_b;
    bool owner;  // the lane that stores this wheel's words (role A of an active slot)
};
__device__ __forceinline__ GroupIds group_ids(int lane, const SlotConst &sc)
{
    GroupIds id;
    id.slot = lane & 7;
    id.role_b = (lane & 8) != 0;
    id.wheel_active = id.slot < 6;
    id.owner = id.wheel_active && !id.role_b;
    id.k = sc.k;
    id.j = sc.j;
    id.si = sc.si;
    id.body = sc.body;
    return id;
}
__device__ __forceinline__ void group_load(const float *__restrict__ state, int N, int e, const GroupIds &id, const SlotConst &sc,
                                           const StepConsts &K, GroupLane &g, int mass_model)
{
#pragma unroll
    for (int i = 0; i < 3; ++i) g.pos[i] = state[(size_t)(ROVER_POS + i) * N + e];
#pragma unroll
    for (int i = 0; i < 4; ++i) g.quat[i] = state[(size_t)(ROVER_QUAT + i) * N + e];
#pragma unroll
    for (int i = 0; i < 3; ++i) g.linvel[i] = state[(size_t)(ROVER_LINVEL + i) * N + e];
#pragma unroll
    for (int i = 0; i < 3; ++i) g.angvel[i] = state[(size_t)(ROVER_ANGVEL + i) * N + e];
    g.bq = state[(size_t)(ROVER_BOGIE_Q + id.j) * N + e];
    g.bqd = state[(size_t)(ROVER_BOGIE_QD + id.j) * N + e];
    const int si = id.si >= 0 ? id.si : 0;
    g.sq = state[(size_t)(ROVER_STEER_Q + si) * N + e];
    g.sqd = state[(size_t)(ROVER_STEER_QD + si) * N + e];
    g.wq = state[(size_t)(ROVER_WHEEL_Q + id.k) * N + e];
    g.wqd = state[(size_t)(ROVER_WHEEL_QD + id.k) * N + e];
    g.lam = state[(size_t)(ROVER_LAMBDA_N + id.k) * N + e];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        g.P[i] = sc.P[i];
        g.ax[i] = sc.ax[i];
    }
    g.arm = make_arm(sc.wb, sc.P, sc.ax);
#pragma unroll
    for (int i = 0; i < 3; ++i) g.lp[i] = id.role_b ? sc.lp[1][i] : sc.lp[0][i];
    {
        // per-lane pick of the bogie constants (same values as K.b_winv[j] / K.bogie_keep[j])
        float bw = K.b_winv[0], bk = K.bogie_keep[0];
#pragma unroll
        for (int j = 1; j < 3; ++j) { bw = (id.j == j) ? K.b_winv[j] : bw; bk = (id.j == j) ? K.bogie_keep[j] : bk; }
        g.b_winv = bw;
        g.bogie_keep = bk;
        float gq = K.bogie_gq[0];
#pragma unroll
        for (int j = 1; j < 3; ++j) gq = (id.j == j) ? K.bogie_gq[j] : gq;
        g.bogie_gq = mass_model == 1 ? gq : 0.0f;
        g.sub = make_sub(sc.sc, sc.P, sc.ax);
    }
    g.steerable = id.si >= 0;
    g.wheel_active = id.wheel_active;
    g.role_b = id.role_b;
    {
        constexpr float COM_B[3] = RV_COM_B_INIT;
        quat_to_mat(g.quat, g.R);
        mat_vecf(g.R, COM_B, g.com_off);
    }
    {
        const float sg = id.slot == 7 ? -1.0f : 1.0f;
        const float m = sg * K.inv_m;
        g.minv0 = (f2){m, sg * (id.role_b ? K.inv_I[2] : K.inv_I[0])};
        g.minv1 = (f2){id.role_b ? sg * g.b_winv : m, id.role_b ? 0.0f : sg * K.inv_I[1]};
    }
}
// physical state back to HBM: chassis by slot 0 / role A, bogie j by slot 2j, steer / wheel words by their wheel's role-A lane
// Words that every lane of an env's sixteen holds (the replicated chassis, the manager words): lane r of the group stores word
// W0 + r -- ONE store instruction for up to sixteen SoA rows instead of one per row from a single lane.  A store instruction
// costs the CU's address path ~50 cycles however few lanes it has (measured in the one-launch kernel: 53 of them removed =
// -1.3 us, the copy waves' window requests queue behind them), a 16-way select ~15 VALU instructions.
// (The select chain is a template recursion: written as a loop, hipcc recognises `x = v[r]` in it, turns it into a run-time index
// and sends the whole array to scratch -- or, promoted, to LDS.)
// word `row` of env e in an SoA array of rows of N floats, addressed by a 32-bit BYTE offset (rows x N x 4 < 4 GiB: checked at
// bind time): the store becomes `global_store_dword voffset, vdata, s[base]` -- no 64-bit multiply-add per lane
__device__ __forceinline__ float *soa_word(float *__restrict__ base, int row, int N, int e)
{
    const unsigned byte_off = ((unsigned)row * (unsigned)N + (unsigned)e) * 4u;
    return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off);
}
template <int I, int CNT, int OFF, int LEN>
__device__ __forceinline__ float pick_by_lane(int r, const float (&v)[LEN], float x)
{
    if constexpr (I < CNT) return pick_by_lane<I + 1, CNT, OFF, LEN>(r, v, (r == I) ? v[OFF + I] : x);
    else return x;
}
template <int W0, int CNT, int OFF, int LEN>
__device__ __forceinline__ void store_rows_by_lane(float *__restrict__ state, int N, int e, int r, const float (&v)[LEN])
{
    static_assert(CNT >= 1 && CNT <= 16 && OFF + CNT <= LEN, "one row per lane of the group");
    const float x = pick_by_lane<1, CNT, OFF, LEN>(r, v, v[OFF]);
    if (r < CNT) *soa_word(state, W0 + r, N, e) = x;
}
__device__ __forceinline__ void group_store(float *__restrict__ state, int N, int e, const GroupIds &id, const GroupLane &g, int lane)
{
    {
        const float ch[13] = {g.pos[0], g.pos[1], g.pos[2], g.quat[0], g.quat[1], g.quat[2], g.quat[3],
                              g.linvel[0], g.linvel[1], g.linvel[2], g.angvel[0], g.angvel[1], g.angvel[2]};
        static_assert(ROVER_POS == 0 && ROVER_QUAT == 3 && ROVER_LINVEL == 7 && ROVER_ANGVEL == 10, "chassis words 0..12");
        store_rows_by_lane<ROVER_POS, 13, 0>(state, N, e, lane & 15, ch);
    }
    if (id.owner) {
        if ((id.slot & 1) == 0) {
            *soa_word(state, ROVER_BOGIE_Q + id.j, N, e) = g.bq;
            *soa_word(state, ROVER_BOGIE_QD + id.j, N, e) = g.bqd;
        }
        if (id.si >= 0) {
            *soa_word(state, ROVER_STEER_Q + id.si, N, e) = g.sq;
            *soa_word(state, ROVER_STEER_QD + id.si, N, e) = g.sqd;
        }
        *soa_word(state, ROVER_WHEEL_Q + id.k, N, e) = g.wq;
        *soa_word(state, ROVER_WHEEL_QD + id.k, N, e) = g.wqd;
        *soa_word(state, ROVER_LAMBDA_N + id.k, N, e) = g.lam;
    }
}

#define RV_K1G_THREADS 256
#define RV_K1G_ENVS (RV_K1G_THREADS / 16)
// surface height from four staged cells (unscaled): the plane of the cell's triangle the ray falls in (cells split along
// the (i, j) - (i+1, j+1) diagonal: what a ray-cast of the terrain's triangle mesh returns) or the bilinear patch
template <bool TRI, typename cell_t>
__device__ __forceinline__ float patch_height(const cell_t *q, int pitch, float fx, float fy)
{
    // Four separate 16-bit LDS reads for the int16 tile.  With unaligned access enabled hipcc merges the two cells of a row into
    // ONE ds_read_b32 at a 2-byte-aligned address, which costs more LDS time than the two reads it replaces (scan kernel 22.5
    // vs 18.6 us at num_envs = 4096): rover_scan_step_kernel is compiled with target("no-unaligned-access-mode").  (A
    // `volatile` read also prevents the merge, but turns the reads into serialised FLAT loads.)
    const float h00 = (float)q[0], h01 = (float)q[1], h10 = (float)q[pitch], h11 = (float)q[pitch + 1];
    if (TRI) {
        const bool lower = fx >= fy;                       // lower triangle: corners 00, 01, 11; upper: 00, 10, 11
        const float pm = lower ? h01 : h10;
        const float d1 = pm - h00, d2 = h11 - pm;
        const float a = lower ? d1 : d2, b = lower ? d2 : d1;
        return fmaf(fy, b, fmaf(fx, a, h00));
    }
    const float dx0 = h01 - h00, dx1 = h11 - h10;
    const float hx0 = h00 + fx * dx0;
    const float hx1 = h10 + fx * dx1;
    return hx0 + fy * (hx1 - hx0);
}
// ------------------------------------------------------------------------------------------------ scan, wave-private form
// The same rays from the same staged cells as rover_scan_step_kernel<true, TRI, ...>, cast by ONE wave for up to four envs
// from two wave-private LDS tiles: no workgroup barrier, K1's launch shape (one wave per SIMD).  What hides the LDS latency is
// the wave's own instruction-level parallelism (sixteen rays per lane, 256 VGPRs to unroll into).  As a kernel of its own
// (measurement hook, rover_debug_set_scan_form(sim, 7)) and as the last phase of the fused step kernel.
struct PrivateWindows {      // wave-uniform: the windows of the wave's four envs
    float px[4], py[4], pz[4], cy[4], sy[4];
    int i_lo[4], j_lo[4], pk[4];
};
template <bool TRI, bool FAST>
__device__ __forceinline__ float private_ray(const RvParams &p, const int16_t *tile, int pitch, int th, float px, float py, float pz,
                                             float cy, float sy, int i_lo, int j_lo, float rx, float ry)
{
    const float x = px + (cy * rx - sy * ry);
    const float y = py + (sy * rx + cy * ry);
    float hgt;
    if (FAST) {
        const float u = (x - p.min_x) * p.inv_res;
        const float v = (y - p.min_y) * p.inv_res;
        const int j0 = (int)u, i0 = (int)v;
        const float fx = u - (float)j0, fy = v - (float)i0;
        hgt = patch_height<TRI>(tile + (__umul24(i0 - i_lo, pitch) + (j0 - j_lo)), pitch, fx, fy);
    } else if (x < p.min_x || x > p.x_max || y < p.min_y || y > p.y_max) {
        return pz - INFINITY - p.cfg.scan_height_offset;  // ray leaves the terrain: ORBIT RayCaster reports +inf
    } else {
        float u = (x - p.min_x) * p.inv_res;
        float v = (y - p.min_y) * p.inv_res;
        u = clampf(u, 0.0f, (float)(p.W - 1));
        v = clampf(v, 0.0f, (float)(p.H - 1));
        int j0 = (int)u, i0 = (int)v;
        if (j0 > p.W - 2) j0 = p.W - 2;
        if (i0 > p.H - 2) i0 = p.H - 2;
        const float fx = u - (float)j0, fy = v - (float)i0;
        const int jl = j0 - j_lo, il = i0 - i_lo;
        const int tw = min(pitch, p.W - j_lo);
        const bool in_tile = jl >= 0 && il >= 0 && jl + 1 < tw && il + 1 < th;
        const int jc = max(0, min(jl, tw - 2)), ic = max(0, min(il, th - 2));
        hgt = patch_height<TRI>(tile + ic * pitch + jc, pitch, fx, fy);
        if (!in_tile) return __int_as_float(0x7fc00000);  // a ray outside the staged window is a bug: NaN
    }
    hgt *= p.q_scale;
    return pz - hgt - p.cfg.scan_height_offset;  // observations.py:45
}
// Window copy by ONE wave, whole rows per instruction: with tw4 chunks per row a global_load_lds moves rpi = 64 / tw4 rows
// (lanes rpi * tw4 .. 63 idle); lane -> (row in the group, chunk in the row) is formed once per tile.  LDS chunk index of
// (row r, chunk c) = r * tw4 + c, as in the scan kernels.
// The loop is written out: the execution mask is set ONCE (whole groups of rows: lanes with lr < rpi; the last, partial group:
// lanes whose row exists), an iteration is the load, one v_add of the lanes' 32-bit byte offset, one s_add of M0 and the
// compare-and-branch on M0 -- five instructions.  (hipcc's loop around the builtin took fifteen per load -- a mask and a branch
// around every load, a 64-bit address add, M0 through a move and a nop -- and one wave per SIMD issues them one by one: 2.5 k
// cycles per window, which round 3 read as the cost of the LDS-DMA instruction itself.)
#ifdef RV_ROW_SPANS
// Round 5, MEASURED AND NOT THE PRODUCT FORM (tools/build_diag.py SPANS; DESIGN.md section 6: bit-exact, fewer bytes, +1.0 us per
// step -- the copy wave's instruction issue is what the staging phase costs, not its bytes): PER-ROW SPANS.  The staged box is the
// bounding box of the yaw-rotated ray pattern (1.64 x its area averaged over the yaw): a chunk (eight cells of one row) is requested only where a ray can need it.  The LDS layout stays the dense tile the cast
// addresses -- a lane that is switched off simply does not write its 16 bytes -- so nothing changes for the rays.  Which rows a
// CHUNK COLUMN needs is an interval (the pattern's rectangle is convex): every lane derives, once per window, the iterations
// [first, first + span] of the copy loop in which its (row-in-group, chunk) is wanted, and an iteration is the old five instructions
// plus a subtract, an unsigned compare, two exec moves and a counter -- ten instructions for five, and ~70 per window to set up.
//   rectangle: centre (uc, vc) in window cells, half extents (Hx, Hy), rotation (c, s);  a ray at (u, v) reads columns floor(u),
//   floor(u) + 1 and rows floor(v), floor(v) + 1.  For the strip du in [da, db] of a chunk column (one cell + half a cell of slack
//   each side) the rows' extent is bounded by  max_i min(l_i(da), l_i(db)) .. min_i max(h_i(da), h_i(db))  over the two edge pairs
//   l_1/h_1 = -(c / s) du -/+ Hx / |s|,  l_2/h_2 = (s / c) du -/+ Hy / |c|  (a pair whose divisor is ~0 is dropped): an OUTER bound,
//   whatever the rounding (v_rcp is enough); one more row of slack each side.  Windows that touch the map's edge (the cast clamps
//   there) are copied whole.
__device__ __forceinline__ void private_issue(const RvParams &p, const PrivateWindows &w, int j, int16_t *tile, int lane)
{
    const int th = w.pk[j] & 0x7FFF, tw4 = max(w.pk[j] >> 16, 1);
    const int rpi = max(64 / tw4, 1);                                  // wave-uniform (tw4 <= 64: checked by the host)
    const float inv_tw4 = 1.0f / (float)tw4;
    const int lr = (int)(((float)lane + 0.5f) * inv_tw4);               // lane / tw4, exact
    const int lc = lane - (int)__umul24(lr, tw4);
    unsigned voff = (unsigned)(__umul24(lr, p.wq) + lc) * 16u;        // the lane's byte offset from the window's first chunk
    const int16_t *base = p.height_q + ((size_t)w.i_lo[j] * p.W + w.j_lo[j]);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) int16_t *)tile;
    const int groups = __builtin_amdgcn_readfirstlane((th + rpi - 1) / rpi);   // copy-loop iterations: rpi rows each, the last one partial
    const unsigned dstep = (unsigned)(rpi * tw4) * 16u, vstep = (unsigned)__umul24(rpi, p.wq) * 16u;
    const float inv_rpi = 1.0f / (float)rpi;
    // rows this lane's chunk column needs (window rows), then the loop iterations k with row k * rpi + lr among them
    int r_first = 0, r_last = th - 1;
    if ((w.pk[j] >> 15) & 1) {   // interior window (wave-uniform)
        const float uc = (w.px[j] - p.min_x) * p.inv_res - (float)w.j_lo[j], vc = (w.py[j] - p.min_y) * p.inv_res - (float)w.i_lo[j];
        const float Hx = 0.5f * p.cfg.scan_size_x * p.inv_res, Hy = 0.5f * p.cfg.scan_size_y * p.inv_res;
        const float c = w.cy[j], sn = w.sy[j];
        const bool use1 = fabsf(sn) > 1.0e-3f, use2 = fabsf(c) > 1.0e-3f;
        const float is = __builtin_amdgcn_rcpf(use1 ? sn : 1.0f), ic = __builtin_amdgcn_rcpf(use2 ? c : 1.0f);
        const float m1 = use1 ? -c * is : 0.0f, w1 = use1 ? Hx * fabsf(is) : 1.0e9f;
        const float m2 = use2 ? sn * ic : 0.0f, w2 = use2 ? Hy * fabsf(ic) : 1.0e9f;
        const float da = (float)(8 * lc) - 1.5f - uc, db = (float)(8 * lc + 7) + 1.5f - uc;
        const float pa = m1 * da, pb = m1 * db, qa = m2 * da, qb = m2 * db;
        const float lo = fmaxf(fminf(pa, pb) - w1, fminf(qa, qb) - w2), hi = fminf(fmaxf(pa, pb) + w1, fmaxf(qa, qb) + w2);
        // (slack: the slopes carry v_rcp's ulp and |du| <= ~110 cells: 1e-4 cells; one whole row each side on top of the +1 of floor(v) + 1)
        r_first = max((int)floorf(vc + lo) - 1, 0);
        r_last = min((int)floorf(vc + hi) + 2, th - 1);
    }
    // k * rpi + lr in [r_first, r_last]:  k >= (r_first - lr) / rpi (floor: one group early at most),  k <= floor((r_last - lr) / rpi)
    const int kf = max((int)floorf(((float)(r_first - lr) + 0.5f) * inv_rpi), 0);
    const int kl = (int)floorf(((float)(r_last - lr) + 0.5f) * inv_rpi);
    const bool never = lr >= rpi || kl < kf || r_last < r_first;
    const unsigned first = never ? 0x7FFFFFFFu : (unsigned)kf, span = never ? 0u : (unsigned)(kl - kf);
    unsigned long long saved;
    unsigned t;
    unsigned m0_saved;
    unsigned k = 0u;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_mov_b32 %[m0s], m0\n\t"
        "s_mov_b32 m0, %[lds0]\n\t"
        "s_cmp_lt_u32 %[k], %[groups]\n\t"
        "s_cbranch_scc0 2f\n"
        "1:\n\t"
        "v_sub_u32 %[t], %[k], %[first]\n\t"
        "v_cmp_le_u32 vcc, %[t], %[span]\n\t"
        "s_and_b64 exec, %[saved], vcc\n\t"
        "global_load_lds_dwordx4 %[voff], %[base]\n\t"
        "s_mov_b64 exec, %[saved]\n\t"
        "v_add_u32 %[voff], %[vstep], %[voff]\n\t"
        "s_add_u32 m0, m0, %[dstep]\n\t"
        "s_add_u32 %[k], %[k], 1\n\t"
        "s_cmp_lt_u32 %[k], %[groups]\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        "s_mov_b32 m0, %[m0s]"
        : [saved] "=&s"(saved), [m0s] "=&s"(m0_saved), [voff] "+v"(voff), [k] "+s"(k), [t] "=&v"(t)
        : [lds0] "s"(lds0), [groups] "s"((unsigned)groups), [base] "s"(base), [vstep] "s"(vstep), [dstep] "s"(dstep),
          [first] "v"(first), [span] "v"(span)
        : "memory", "scc", "vcc");
}
#else
__device__ __forceinline__ void private_issue(const RvParams &p, const PrivateWindows &w, int j, int16_t *tile, int lane)
{
    const int th = w.pk[j] & 0x7FFF, tw4 = max(w.pk[j] >> 16, 1);
    // Quotients of small integers by v_rcp_f32 (a <= tile_dim <= 1024, b <= 64: the exact quotient's fractional part is 0 or >= 1 / 64,
    // the product's error < 1e-4, so floor(a * rcp(b) + 1e-3) IS a / b): four instructions where hipcc's integer division takes ~25 --
    // three quotients per window, on the copy wave's chain between barrier A and barrier B
    const float inv_tw4 = __builtin_amdgcn_rcpf((float)tw4);
    const int rpi = __builtin_amdgcn_readfirstlane(max((int)(64.0f * inv_tw4 + 1.0e-3f), 1));   // 64 / tw4: wave-uniform (tw4 <= 64: checked by the host)
    const int lr = (int)(((float)lane + 0.5f) * inv_tw4);               // lane / tw4 (fractional part >= 0.5 / 64)
    const int lc = lane - (int)__umul24(lr, tw4);
    unsigned voff = (unsigned)(__umul24(lr, p.wq) + lc) * 16u;        // the lane's byte offset from the window's first chunk
    const int16_t *base = p.height_q + ((size_t)w.i_lo[j] * p.W + w.j_lo[j]);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) int16_t *)tile;
    const int n_full = __builtin_amdgcn_readfirstlane((int)((float)th * __builtin_amdgcn_rcpf((float)rpi) + 1.0e-3f));   // th / rpi: whole groups of rpi rows
    const int n_last = th - n_full * rpi;                                                   // rows of the last group
    const unsigned dstep = (unsigned)__builtin_amdgcn_readfirstlane((rpi * tw4) * 16), vstep = (unsigned)__umul24(rpi, p.wq) * 16u;
    const unsigned lds_end = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)n_full * dstep));
    const unsigned long long m_full = __builtin_amdgcn_ballot_w64(lr < rpi), m_last = __builtin_amdgcn_ballot_w64(lr < n_last);
    unsigned long long saved;
    unsigned m0_saved;
    asm volatile(
        "s_mov_b64 %[saved], exec\n\t"
        "s_mov_b32 %[m0s], m0\n\t"
        "s_mov_b32 m0, %[lds0]\n\t"
        "s_mov_b64 exec, %[mfull]\n\t"
        "s_cmp_lt_u32 m0, %[lend]\n\t"
        "s_cbranch_scc0 2f\n"
        "1:\n\t"
        "global_load_lds_dwordx4 %[voff], %[base]\n\t"
        "v_add_u32 %[voff], %[vstep], %[voff]\n\t"
        "s_add_u32 m0, m0, %[dstep]\n\t"
        "s_cmp_lt_u32 m0, %[lend]\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        "s_mov_b64 exec, %[mlast]\n\t"
        "s_cbranch_execz 3f\n\t"
        "global_load_lds_dwordx4 %[voff], %[base]\n"
        "3:\n\t"
        "s_mov_b32 m0, %[m0s]\n\t"
        "s_mov_b64 exec, %[saved]"
        : [saved] "=&s"(saved), [m0s] "=&s"(m0_saved), [voff] "+v"(voff)
        : [lds0] "s"(lds0), [lend] "s"(lds_end), [mfull] "s"(m_full), [mlast] "s"(m_last), [base] "s"(base), [vstep] "s"(vstep),
          [dstep] "s"(dstep)
        : "memory", "scc");
}
#endif
// (Measured and not kept: the same copy through registers -- 16-byte global loads issued before the rays of the env that still
// occupies the tile, LDS writes afterwards.  A single wave issues one global_load_lds_dwordx4 every ~145 cycles, 2.6 k cycles per
// window during which nothing else of the wave issues; the register route was slower still: 53.0 us per step against 48.3 us.
// The four waves of a CU move 4 x 4 windows x 18.9 KB = 302 KB per step through one L1 either way.)
// the windows of the wave's four envs from the per-lane window of each env's row (lane 16 j speaks for env j)
__device__ __forceinline__ void private_windows(const ScanWindow &sw, PrivateWindows &w)
{
    const int pk = sw.th | (sw.interior << 15) | (sw.tw4 << 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        auto bcast = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16 * j)); };
        w.px[j] = bcast(sw.px); w.py[j] = bcast(sw.py); w.pz[j] = bcast(sw.pz); w.cy[j] = bcast(sw.cy); w.sy[j] = bcast(sw.sy);
        w.i_lo[j] = __builtin_amdgcn_readlane(sw.i_lo, 16 * j); w.j_lo[j] = __builtin_amdgcn_readlane(sw.j_lo, 16 * j);
        w.pk[j] = __builtin_amdgcn_readlane(pk, 16 * j);
    }
}
// the same windows, declared wave-uniform to the compiler (values merged from two wave-uniform paths lose that property in its
// divergence analysis, and private_issue's scalar operands need it)
__device__ __forceinline__ void uniform_windows(PrivateWindows &w)
{
    auto uf = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        w.px[j] = uf(w.px[j]); w.py[j] = uf(w.py[j]); w.pz[j] = uf(w.pz[j]); w.cy[j] = uf(w.cy[j]); w.sy[j] = uf(w.sy[j]);
        w.i_lo[j] = __builtin_amdgcn_readfirstlane(w.i_lo[j]); w.j_lo[j] = __builtin_amdgcn_readfirstlane(w.j_lo[j]);
        w.pk[j] = __builtin_amdgcn_readfirstlane(w.pk[j]);
    }
}
#define RV_FUSED_ATTR __attribute__((target("no-unaligned-access-mode")))
constexpr int PRIVATE_ROUNDS = 16, PRIVATE_GROUP = 4;   // a pipeline group is a quad of rounds (one 16-byte store per lane)
// Ray -> (round m, lane): ray = 256 (m >> 2) + 4 lane + (m & 3).  A lane's four rays of a QUAD of rounds are neighbours in the
// observation row, so a quad ends in ONE 16-byte store per lane (row pointer in SGPRs, the lane's byte offset in one VGPR shared
// by all quads, the quad in the immediate offset); a 4-byte store per round cost the wave ~45 cycles each.  Rows are only 4-byte
// aligned (965 floats): the 16-byte stores straddle cache lines, which the one-wave-per-SIMD scan does not notice.
// (a.y, a.y) * b as one v_pk_mul_f32 (op_sel picks the high half of `a` for both lanes: no move into the low half first)
__device__ __forceinline__ f2 pk_mul_hi(f2 a, f2 b)
{
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "s"(b));
    return r;
}
// The four cells of a ray as inline asm: hipcc moves its own LDS loads next to their first use (past scheduling barriers), which
// undoes the software pipeline below.  The loads are asynchronous -- the registers they name are defined only after the
// lds_cell_wait that lists them (its "+v" ties make every later use depend on the wait).
__device__ __forceinline__ void lds_cell4_issue(unsigned row0, unsigned row1, int &h00, int &h01, int &h10, int &h11)
{
    // (Measured, round 4: the two cells of a row as ONE 2-byte-aligned ds_read_b32, split in the epilogue: 45.1 us per step against
    // 35.6 -- a misaligned 32-bit LDS read costs far more than the two 16-bit reads it replaces, as in round 2's scan kernel.)
    asm volatile("ds_read_i16 %0, %4\n\t"
                 "ds_read_i16 %1, %4 offset:2\n\t"
                 "ds_read_i16 %2, %5\n\t"
                 "ds_read_i16 %3, %5 offset:2"
                 : "=&v"(h00), "=&v"(h01), "=&v"(h10), "=&v"(h11) : "v"(row0), "v"(row1));
}
// The TRIANGLE surface needs three of the four corners: 00, 11 and ONE of 01 / 10 -- fx >= fy (lower triangle) picks 01, and is known
// before the reads are issued: the select moves from the value to the ADDRESS (`mid` = lower ? a0 : a0 + 2 pitch - 2, read at offset 2;
// `r1m` = a0 + 2 pitch - 2, corner 11 at offset 4): three LDS instructions per ray instead of four, the same three operands into the
// same arithmetic.  (The conflict cycles of a 64-lane 16-bit read are a property of the instruction, docs/history.md section 3.5: the lever is the COUNT.)
__device__ __forceinline__ void lds_cell3_issue(unsigned a0, unsigned mid, unsigned r1m, int &h00, int &hm, int &h11)
{
    asm volatile("ds_read_i16 %0, %3\n\t"
                 "ds_read_i16 %1, %4 offset:2\n\t"
                 "ds_read_i16 %2, %5 offset:4"
                 : "=&v"(h00), "=&v"(hm), "=&v"(h11) : "v"(a0), "v"(mid), "v"(r1m));
}
template <int CNT>
__device__ __forceinline__ void lds_cell3_wait(int (&h00)[4], int (&hm)[4], int (&h11)[4])
{
    asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(h00[0]), "+v"(hm[0]), "+v"(h11[0]), "+v"(h00[1]), "+v"(hm[1]), "+v"(h11[1]),
                 "+v"(h00[2]), "+v"(hm[2]), "+v"(h11[2]), "+v"(h00[3]), "+v"(hm[3]), "+v"(h11[3]) : "n"(CNT));
}
template <int CNT>
__device__ __forceinline__ void lds_cell_wait(int (&h00)[4], int (&h01)[4], int (&h10)[4], int (&h11)[4])
{
    // (an asm statement takes at most 30 operands: the wait carries half of the group, an empty statement behind it the rest)
    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(h00[0]), "+v"(h01[0]), "+v"(h10[0]), "+v"(h11[0]), "+v"(h00[1]), "+v"(h01[1]), "+v"(h10[1]), "+v"(h11[1]) : "n"(CNT));
    asm volatile("" : "+v"(h00[2]), "+v"(h01[2]), "+v"(h10[2]), "+v"(h11[2]), "+v"(h00[3]), "+v"(h01[3]), "+v"(h10[3]), "+v"(h11[3]));
}
__device__ __forceinline__ int private_ray_index(int m, int lane) { return ((m >> 2) << 8) + (lane << 2) + (m & 3); }
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v4f_u4 __attribute__((ext_vector_type(4), aligned(4)));   // a row is 4-byte aligned
// (The stores are plain C++ on purpose: written as inline asm they were invisible to hipcc's hazard recogniser, and a 16-byte
// store reads its data registers for several cycles after issue -- the next group's first packed multiply overwrote the second
// dword of lanes 12 .. 15 of every row.)
template <int Q>
__device__ __forceinline__ void private_store4(float *row /* wave-uniform */, unsigned lane_bytes16, v4f_t v, bool nt)
{
    // (address space 1: `row` was rebuilt from two readfirstlanes, and a generic pointer would make this a FLAT store, which
    // also counts in lgkmcnt -- the counter the pipeline's LDS reads are waited on)
    typedef __attribute__((address_space(1))) v4f_u4 *global_v4;
    typedef __attribute__((address_space(1))) char *global_bytes;
    // streaming (non-temporal) stores on request (rover_set_obs_streaming): the 15.8 MB of observation rows leave every CU within
    // the same few microseconds and nothing of this kernel reads them back -- 34.9 us per step against 35.6 with plain stores
    // (tools/quick_bench.py, round 4).  Not the default: the next reader of the rows is usually the policy kernel, whose tile copy
    // finds plainly stored rows in L2 (pair kernel 31.0 us behind plain stores, 33.7 us behind streamed ones).
    if (nt) __builtin_nontemporal_store(v, (global_v4)((global_bytes)row + 1024 * Q + lane_bytes16));
    else *(global_v4)((global_bytes)row + 1024 * Q + lane_bytes16) = v;
}
// quad `quad` of a row: quads 0 .. 2 of a dense pattern are full; in the last one lanes below n3 / 4 hold four rays, lane n3 / 4
// the n3 % 4 that remain (n3 = rays - 768: 193 .. 256)
__device__ __forceinline__ void private_store_quad(float *row, int lane, int rays, int quad, v4f_t o4, bool nt)
{
    const unsigned lane_bytes16 = (unsigned)lane * 16u;
    switch (quad) {   // the quad is the store's immediate offset
    case 0: private_store4<0>(row, lane_bytes16, o4, nt); break;
    case 1: private_store4<1>(row, lane_bytes16, o4, nt); break;
    case 2: private_store4<2>(row, lane_bytes16, o4, nt); break;
    default: {
        const int n3 = rays - 64 * (PRIVATE_ROUNDS - 4);
        if (lane < (n3 >> 2)) {
            private_store4<3>(row, lane_bytes16, o4, nt);
        } else if (lane == (n3 >> 2)) {
            __attribute__((address_space(1))) float *tail = (__attribute__((address_space(1))) float *)row + 256 * 3 + 4 * lane;
            if ((n3 & 3) > 0) tail[0] = o4.x;
            if ((n3 & 3) > 1) tail[1] = o4.y;
            if ((n3 & 3) > 2) tail[2] = o4.z;
        }
        break;
    }
    }
}
// the wave-uniform observation row of env e (rebuilt from two readfirstlanes: the stores want the pointer in SGPRs)
__device__ __forceinline__ float *private_row(float *out, int e, int row_stride, int col0)
{
    const unsigned long long a = (unsigned long long)(out + (size_t)e * row_stride + col0);
    return (float *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                     (unsigned)__builtin_amdgcn_readfirstlane((int)a));
}
// A lane without a ray in round m repeats ray 0 (the table says so) and stores nothing: no branches inside an env's rounds --
// one basic block.  Rounds [M0, M1) of the env's sixteen (the step wave and its copy wave share an env's rays).
// The pipelined path is for interior windows of DENSE patterns (961 .. 1024 rays: 31 x 31 and 32 x 32, every round but the last
// full -- its stores need no lane mask, and a mask costs scalar instructions that one wave per SIMD issues no faster than vector
// ones); anything else takes the rolled loop below.
//
// The interior path is a software pipeline over groups of PRIVATE_GROUP rounds, held in place with scheduling barriers (one
// wave per SIMD: nothing but the wave's own instruction stream hides the LDS latency, and left to itself the scheduler waits
// for a pair of rays right after issuing their reads):   coordinates(g + 1) | reads(g + 1) | triangle + store(g).
// Same arithmetic per ray as private_ray<TRI, true> -- written for instruction count (45 -> 26 VALU per ray):
//   * rotation, translation and cell coordinates as packed pairs: (cy, sy) rx, (-sy, cy) ry, their sum, + (px, py),
//     + (-min_x, -min_y), * inv_res -- six v_pk_*_f32, each the IEEE operation of the scalar form (a - b = a + (-b),
//     (-s) r = -(s r));
//   * fx = u - (float)(int)u is v_fract_f32(u) for 0 <= u < 2^23 (the difference is exact, and below 1);
//   * cell address = ((i0 * pitch + j0) << 1) + (tile - 2 (i_lo * pitch + j_lo)): one v_mad_u32_u24, one v_lshl_add_u32;
//   * pz - hgt * q_scale as ONE fma: q_scale is a power of two (rover_set_terrain_q16 checks), so the product is exact and the
//     fused form rounds the same real number once, like the subtraction did.
template <bool TRI, int M0 = 0, int M1 = PRIVATE_ROUNDS>
__device__ __forceinline__ void private_cast(const RvParams &p, const PrivateWindows &w, int j, const int16_t *tile, int lane, int e_base,
                                             float *__restrict__ out, int row_stride, int col0, const f2 (&oxy)[PRIVATE_ROUNDS],
                                             const float2 *__restrict__ ray_xy)
{
    if constexpr (M1 <= M0) {   // an empty share
        return;
    } else {
    constexpr int CC = 8, G = PRIVATE_GROUP;
    static_assert(G == 4 && M0 % 4 == 0 && M1 % 4 == 0, "a share is whole quads of rounds");
    const int th = w.pk[j] & 0x7FFF;
    const int pitch = (w.pk[j] >> 16) * CC;
    float *row = private_row(out, e_base + j, row_stride, col0);   // wave-uniform by construction (e_base is the wave's first env)
    if (M0 * 64 >= p.rays) return;
    if (__builtin_expect(((w.pk[j] >> 15) & 1) && p.rays > 64 * (PRIVATE_ROUNDS - 1), 1)) {   // (the common case falls through)
        typedef const __attribute__((address_space(3))) int16_t *lds_cell_ptr;
        const f2 A = {w.cy[j], w.sy[j]}, BN = {-w.sy[j], w.cy[j]}, P = {w.px[j], w.py[j]}, NMIN = {-p.min_x, -p.min_y};
        const f2 IR = {p.inv_res, p.inv_res};
        const unsigned tile_lds = (unsigned)(size_t)(lds_cell_ptr)tile;
        const unsigned base = tile_lds - 2u * (unsigned)(w.i_lo[j] * pitch + w.j_lo[j]);   // wraps like the per-ray subtraction did
        // uniforms of the epilogue pinned in VGPRs: under SGPR pressure hipcc re-loads them from the kernel arguments in the middle
        // of the pipeline, and a scalar load's s_waitcnt lgkmcnt(0) also waits for every LDS read in flight
        float nqs = -p.q_scale, pz = w.pz[j], hoff = p.cfg.scan_height_offset;
        asm volatile("" : "+v"(nqs), "+v"(pz), "+v"(hoff));
        float fx[2][G], fy[2][G];
        int h00[2][G], h01[2][G], h10[2][G], h11[2][G];   // TRI: h01 holds the ONE middle corner (01 or 10), h10 is unused
        const unsigned row1m = 2u * (unsigned)pitch - 2u;   // byte distance from cell (i, j) to cell (i + 1, j) minus one cell
        constexpr int NG = (M1 - M0 + G - 1) / G;
#pragma unroll
        for (int g = 0; g <= NG; ++g) {
            const int b = g & 1;
            if (g < NG) {
                unsigned a0[G];
                f2 t0[G], t1[G];
                // stage-major over the group's rays: no packed op reads the result of the one before it
#define RV_CAST_STAGE(...)                                                                                         \
    _Pragma("unroll") for (int q = 0; q < G; ++q) { const int m = M0 + g * G + q; if (m < M1) { __VA_ARGS__ } }
                RV_CAST_STAGE(t0[q] = A * (f2){oxy[m].x, oxy[m].x};)
                RV_CAST_STAGE(t1[q] = pk_mul_hi(oxy[m], BN);)
                RV_CAST_STAGE(t0[q] = t0[q] + t1[q];)
                RV_CAST_STAGE(t0[q] = P + t0[q];)
                RV_CAST_STAGE(t0[q] = t0[q] + NMIN;)
                RV_CAST_STAGE(t0[q] = t0[q] * IR;)
                RV_CAST_STAGE(t1[q] = (f2){__builtin_amdgcn_fractf(t0[q].x), __builtin_amdgcn_fractf(t0[q].y)};
                              fx[b][q] = t1[q].x; fy[b][q] = t1[q].y;)
                RV_CAST_STAGE(a0[q] = __umul24((int)t0[q].y, pitch) + (unsigned)(int)t0[q].x;)
                RV_CAST_STAGE(a0[q] = (a0[q] << 1) + base;)
                unsigned r1m[G], mid[G];
                if (TRI) {
                    RV_CAST_STAGE(r1m[q] = a0[q] + row1m;)
                    RV_CAST_STAGE(mid[q] = fx[b][q] >= fy[b][q] ? a0[q] : r1m[q];)
                }
#undef RV_CAST_STAGE
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const int m = M0 + g * G + q;
                    if (m < M1) {
                        if (TRI) lds_cell3_issue(a0[q], mid[q], r1m[q], h00[b][q], h01[b][q], h11[b][q]);
                        else lds_cell4_issue(a0[q], a0[q] + 2u * (unsigned)pitch, h00[b][q], h01[b][q], h10[b][q], h11[b][q]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g > 0) {
                const int pb = b ^ 1;
                float ov[G];
                // the previous group's sixteen reads have returned when at most the fifteen youngest LDS operations are outstanding
                // (LDS returns in order; the group issued just above is sixteen operations); after the last group: all of them
                // (triangle surface: three reads per ray, the group above is twelve operations)
                if (TRI) {
                    if (g < NG) lds_cell3_wait<12>(h00[pb], h01[pb], h11[pb]);
                    else lds_cell3_wait<0>(h00[pb], h01[pb], h11[pb]);
                } else {
                    if (g < NG) lds_cell_wait<15>(h00[pb], h01[pb], h10[pb], h11[pb]);
                    else lds_cell_wait<0>(h00[pb], h01[pb], h10[pb], h11[pb]);
                }
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const int m = M0 + (g - 1) * G + q;
                    if (m < M1) {
                        float hgt;
                        const float f00 = (float)h00[pb][q], f11 = (float)h11[pb][q];
                        if (TRI) {
                            const bool lower = fx[pb][q] >= fy[pb][q];
                            const float pm = (float)h01[pb][q];   // the middle corner the address select fetched: 01 (lower) or 10
                            const float d1 = pm - f00, d2 = f11 - pm;
                            const float ta = lower ? d1 : d2, tb = lower ? d2 : d1;
                            hgt = fmaf(fy[pb][q], tb, fmaf(fx[pb][q], ta, f00));
                        } else {
                            const float f01 = (float)h01[pb][q], f10 = (float)h10[pb][q];
                            const float dx0 = f01 - f00, dx1 = f11 - f10;
                            const float hx0 = f00 + fx[pb][q] * dx0;
                            const float hx1 = f10 + fx[pb][q] * dx1;
                            hgt = hx0 + fy[pb][q] * (hx1 - hx0);
                        }
                        ov[q] = fmaf(nqs, hgt, pz) - hoff;   // observations.py:45
                    }
                }
                {
                    const v4f_t o4 = {ov[0], ov[1], ov[2], ov[3]};
                    private_store_quad(row, lane, p.rays, (M0 >> 2) + g - 1, o4, p.nt_obs != 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
#pragma unroll 1
        for (int m = M0; m < M1; ++m) {
            if (private_ray_index(m, 0) < p.rays) {
                // (the table registers indexed by a loop counter would leave the registers: the slow path re-reads the table)
                const float2 xy = ray_xy[m * 64 + lane];
                const float o = private_ray<TRI, false>(p, tile, pitch, th, w.px[j], w.py[j], w.pz[j], w.cy[j], w.sy[j], w.i_lo[j],
                                                        w.j_lo[j], xy.x, xy.y);   // (also right for an interior window: the bounds tests pass)
                const int r = private_ray_index(m, lane);
                if (r < p.rays) row[r] = o;
            }
        }
    }
    }
}
// ray_xy: 1024 x (x, y) pattern offsets, entry m * 64 + lane = ray private_ray_index(m, lane) (rays past the pattern repeat ray 0), built by the host
template <bool TRI>
__device__ __forceinline__ void scan_private_wave(const RvParams &p, int16_t *tile0, int16_t *tile1, int lane, int n_env, int e_base,
                                                  const PrivateWindows &w, float *__restrict__ out, int row_stride, int col0,
                                                  const float2 *__restrict__ ray_xy)
{
    constexpr int ROUNDS = PRIVATE_ROUNDS;
    auto issue = [&](int j, int16_t *tile) { private_issue(p, w, j, tile, lane); };
    f2 oxy[ROUNDS];
    auto cast = [&](int j, const int16_t *tile) { private_cast<TRI>(p, w, j, tile, lane, e_base, out, row_stride, col0, oxy, ray_xy); };
    if (n_env <= 0) return;
    issue(0, tile0);
    if (n_env > 1) issue(1, tile1);
#pragma unroll
    for (int m = 0; m < ROUNDS; ++m) {
        const float2 v = ray_xy[m * 64 + lane];
        oxy[m] = (f2){v.x, v.y};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    K1_STAMP(27);
    cast(0, tile0);
    K1_STAMP(28);
    if (n_env > 2) {   // tile 0 is dead (its LDS reads have returned: their values were consumed); env 2's window lands under env 1's rays
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        issue(2, tile0);
        K1_STAMP_NOWAIT(19);
    }
    if (n_env > 1) cast(1, tile1);
    K1_STAMP(29);
    if (n_env > 2) {   // env 2's window was issued before env 1's rays: it has landed; env 3's goes out now and lands under env 2's rays
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (n_env > 3) issue(3, tile1);
        K1_STAMP(30);
        cast(2, tile0);
    }
    K1_STAMP(31);
    if (n_env > 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        cast(3, tile1);
    }
}

// ---- the fused step kernel's scan phase WITHOUT copy waves, ONE tile per wave: for batches of more than one round of workgroups
// (N > 16 x CUs).  Copy waves would put four 256-VGPR waves on a SIMD (two workgroups per CU); here a workgroup is the four step
// waves with 4 x 18.9 KB of LDS, two workgroups share a CU as on the two-launch path, and what fills one wave's copy stalls
// (2.5 k cycles per window) and LDS latencies is the other workgroup's wave on the same SIMD.
template <bool TRI>
__device__ __forceinline__ void scan_single_tile_wave(const RvParams &p, int16_t *tile, int lane, int n_env, int e_base,
                                                      const PrivateWindows &w, float *__restrict__ out, int row_stride, int col0,
                                                      const float2 *__restrict__ ray_xy)
{
    if (n_env <= 0) return;
    private_issue(p, w, 0, tile, lane);
    f2 oxy[PRIVATE_ROUNDS];
#pragma unroll
    for (int m = 0; m < PRIVATE_ROUNDS; ++m) {
        const float2 v = ray_xy[m * 64 + lane];
        oxy[m] = (f2){v.x, v.y};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < n_env) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            private_cast<TRI>(p, w, j, tile, lane, e_base, out, row_stride, col0, oxy, ray_xy);
            if (j + 1 < n_env) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile's last reads have returned
                private_issue(p, w, j + 1, tile, lane);
            }
        }
    }
}

// ---- the fused step kernel's scan phase with COPY WAVES.  One wave issues a global_load_lds_dwordx4 every ~145 cycles and
// nothing else meanwhile (2.3 k cycles per window: the CU's memory pipeline delivers ~30 B/clk to its requesting waves), so the four
// step waves of a workgroup are paired with four copy waves (waves 4..7, one per SIMD, asleep at a barrier during the physics):
// copy wave k + 4 serves step wave k.  During the physics it draws what a reset of each env WOULD draw (reset_draw) and, during
// the LAST substep, evaluates the link-body sample points and the wheels' obstacle look-ups of the contact report.  The reset is
// decided by the step wave right after the physics (mdp_terminations: the other termination terms are functions of words loaded
// before the physics).  Windows travel through LDS (win[wave][env][8 words]).
// Round 5, the product form (RV_OWN_TILES = 1): TWO workgroup barriers, executed by all eight waves on every path, and ONE polled
// LDS word per env (word 3 of the env's position slot in the hand-over area):
//   L   pose + bogie angles of the last substep's start written | copy: link-point forces, obstacle heights | step: the last substep
//   A   pose the physics left + link forces written (word = 0)  | step: contact report, collision flag, RESET DECISION -> word = 1 | 2,
//       manager tail                                            | copy: windows of the four poses, window 0 requested at once into
//       tile 0; polls the word; an env that resets: window from the spawn pose it drew (window 0 restaged); final windows -> `win`,
//       word = 3
//   then each wave OWNS A TILE and runs at its own pace, no barrier:
//       copy wave, tile 0: wait, rays of env 0, request window 2, wait, rays of env 2
//       step wave, tile 1: polls the word for 3, reads `win`, requests window 1, wait, rays of env 1, request window 3, wait, rays of env 3
// The barrier form (RV_OWN_TILES = 0; rounds 4 - 5: the copy wave requests all four windows, the waves meet at barriers B, C, D and
// split envs 1 - 3 by RV_SHARE_*) is kept as a build variant (tools/build_diag.py BARRIERS): 38.90 -> 37.77 us per step for the own-tiles form.
// A single wave's cast is bound by its instruction issue (~5 cycles per instruction); all eight waves of a CU casting at once are bound
// by the LDS array (three 16-bit reads per ray-round x 8 waves; profiles/r05_exec_probe.txt).
__device__ __forceinline__ void windows_to_lds(float *win, const ScanWindow &sw, int lane)
{
    if ((lane & 15) == 0) {
        float4 *d = reinterpret_cast<float4 *>(win + (lane >> 4) * 8);
        d[0] = make_float4(sw.px, sw.py, sw.pz, sw.cy);
        d[1] = make_float4(sw.sy, __int_as_float(sw.i_lo), __int_as_float(sw.j_lo), __int_as_float(sw.th | (sw.interior << 15) | (sw.tw4 << 16)));
    }
}
__device__ __forceinline__ void windows_from_lds(const float *win, PrivateWindows &w)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 d0 = reinterpret_cast<const float4 *>(win + j * 8)[0], d1 = reinterpret_cast<const float4 *>(win + j * 8)[1];
        auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
        w.px[j] = uni(d0.x); w.py[j] = uni(d0.y); w.pz[j] = uni(d0.z); w.cy[j] = uni(d0.w); w.sy[j] = uni(d1.x);
        w.i_lo[j] = __builtin_amdgcn_readfirstlane(__float_as_int(d1.y)); w.j_lo[j] = __builtin_amdgcn_readfirstlane(__float_as_int(d1.z));
        w.pk[j] = __builtin_amdgcn_readfirstlane(__float_as_int(d1.w));
    }
}
// LDS of the copy-wave form behind the eight tiles: windows [4 waves][64] floats (1 KB; [4 envs][8] used), then per step wave
// RV_HAND floats of hand-over: [0, 48) rotation matrix + position of its four envs at the START of the last substep,
// [64, 128) the bogie angle of every lane, [128, 192) the lane's link-point force (z; x and y: [320, 384), [384, 448)) and
// [192, 256) the obstacle-layer height under the lane's wheel, written by its twin in the copy wave, [256, 304) the reset outcomes
// of the four envs (12 floats each: reset_draw evaluated by the copy wave during the physics).
constexpr int RV_HAND = 448;
__device__ __forceinline__ float *fused_win(float *lds, const RvParams &p, int wv)
{
    return reinterpret_cast<float *>(reinterpret_cast<int16_t *>(lds) + (size_t)8 * p.tile_dim * p.tile_pitch) + wv * 64;
}
__device__ __forceinline__ float *fused_link(float *lds, const RvParams &p, int wv) { return fused_win(lds, p, 0) + 256 + wv * RV_HAND; }
// Rounds (whole quads) of an env's sixteen cast by the STEP wave; the copy wave takes the rest.  Env 0 is the copy wave's alone
// (cast under the manager tail, so that tile 0 is free for window 2 before barrier B); beside env 2 it stages window 3.
#ifndef RV_OWN_TILES
#define RV_OWN_TILES 1   // 1: each wave of a pair stages and casts two envs through its own tile, no barrier behind A; 0: the barrier form (B, C, D)
#endif
#ifndef RV_SHARE_1
#define RV_SHARE_1 16
#endif
#ifndef RV_SHARE_2
#define RV_SHARE_2 16
#endif
#ifndef RV_SHARE_3
#define RV_SHARE_3 8
#endif
constexpr int SHARE_1 = RV_SHARE_1, SHARE_2 = RV_SHARE_2, SHARE_3 = RV_SHARE_3;
constexpr int SHARE_MAX = SHARE_1 > SHARE_2 ? (SHARE_1 > SHARE_3 ? SHARE_1 : SHARE_3) : (SHARE_2 > SHARE_3 ? SHARE_2 : SHARE_3);
template <bool TRI>
__device__ __forceinline__ void scan_copy_wave(const RvParams &p, const float *__restrict__ state, float *lds, int partner, int lane,
                                               float *__restrict__ obs, const float2 *__restrict__ ray_xy)
{
    const int tile_cells = p.tile_dim * p.tile_pitch;
    int16_t *tile0 = reinterpret_cast<int16_t *>(lds) + (size_t)(2 * partner) * tile_cells, *tile1 = tile0 + tile_cells;
    const float *win = fused_win(lds, p, partner);
    const int e_base = (int)(blockIdx.x * 4 + partner) * 4;
    const int n_env = max(0, min(4, p.n - e_base));
    PrivateWindows w;
    // the ray table: the same 8 KB for every wave of every step.  Requested NOW -- this wave has registers to spare during the
    // physics -- and not behind barrier A, where its sixteen loads would queue in front of env 0's cast with the window requests
    // (a load instruction costs ~120 cycles of the wave's issue there, whatever it carries)
    f2 oxy[PRIVATE_ROUNDS];
#pragma unroll
    for (int m = 0; m < PRIVATE_ROUNDS; ++m) {
        const float2 v = ray_xy[m * 64 + lane];
        oxy[m] = (f2){v.x, v.y};
    }
    {   // before anything of this step is known: what a reset of each of the partner's envs WOULD draw (spawn row, yaw, target with
        // its rejection loop, heading: functions of (global id, reset count) and the terrain).  The copy wave is asleep during the
        // physics anyway; the step wave's reset becomes a handful of LDS reads instead of Philox rounds and dependent loads.
        const int e = min(e_base + (lane >> 4), p.n - 1);
        const uint32_t count = __float_as_uint(state[(size_t)ROVER_RESET_COUNT * p.n + e]);
        ResetOutcome ro;
        reset_draw(p, (uint32_t)(p.env_id_offset + e), count, nullptr, ro);
        if ((lane & 15) == 0) {
            float4 *d = reinterpret_cast<float4 *>(fused_link(lds, p, partner) + 256 + (lane >> 4) * 12);
            d[0] = make_float4(ro.px, ro.py, ro.pz, ro.qw);
            d[1] = make_float4(ro.qz, ro.tx, ro.ty, ro.tz);
            d[2] = make_float4(ro.heading_cmd, 0.0f, 0.0f, 0.0f);
        }
    }
    {   // L: the step wave has left the pose of the last substep's start; this lane evaluates the link-body sample point of its
        // twin (same slot, same role: same constants, same arithmetic -- link_point_fetch / link_point_eval) and the obstacle
        // layer under the twin's wheel (the patch terrain_sample<true> would gather: same cell, same weights) while the twin solves
        __syncthreads();                                                // L
        const float *lk = fused_link(lds, p, partner);
        const SlotConst sc = d_SLOT[lane & 7];
        const bool role_b = (lane & 8) != 0;
        const float *pe = lk + (lane >> 4) * 12;
        const float R[3][3] = {{pe[0], pe[1], pe[2]}, {pe[3], pe[4], pe[5]}, {pe[6], pe[7], pe[8]}};
        const float pos[3] = {pe[9], pe[10], pe[11]};
        const float P[3] = {sc.P[0], sc.P[1], sc.P[2]}, ax[3] = {sc.ax[0], sc.ax[1], sc.ax[2]};
        float lp[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) lp[i] = role_b ? sc.lp[1][i] : sc.lp[0][i];
        const float bq = lk[64 + lane];
        float bsc[2];
        bogie_sincos(bq, &bsc[0], &bsc[1]);
        const LinkSample ls = link_point_fetch(p, R, pos, P, ax, bq, lp, bsc);
        const float wbp[3] = {sc.wb[0], sc.wb[1], sc.wb[2]};
        const LinkSample ws = link_point_fetch(p, R, pos, P, ax, bq, wbp, bsc);   // the wheel centre rides on the bogie like a link point
        float f3[3];
        link_point_eval(ls, p.inv_res, f3);
        const_cast<float *>(lk)[320 + lane] = f3[0];
        const_cast<float *>(lk)[384 + lane] = f3[1];
        const_cast<float *>(lk)[128 + lane] = f3[2];
        const_cast<float *>(lk)[192 + lane] = link_sample_obstacle(ws);
    }
    K1_LITE(0);
    __syncthreads();                                                    // A: the step wave has left the pose the physics produced
    K1_LITE(1);
    {   // the windows of that pose, derived HERE (this wave's lanes 16 j speak for env j), and the first two requested at once: in all
        // but ~0.1 % of the cases they are the final ones, and they travel while the step wave still gathers the contact report
        float *lk = const_cast<float *>(fused_link(lds, p, partner));
        const float4 d = reinterpret_cast<const float4 *>(lk + 48 + (lane >> 4) * 4)[0];
        const float4 q = reinterpret_cast<const float4 *>(lk + 304 + (lane >> 4) * 4)[0];
        const float pf[3] = {d.x, d.y, d.z}, qf[4] = {q.x, q.y, q.z, q.w};
        const ScanWindow sw = scan_window(p, pf, qf);
        private_windows(sw, w);
        if (n_env > 0) private_issue(p, w, 0, tile0, lane);
        if (!RV_OWN_TILES && n_env > 1) private_issue(p, w, 1, tile1, lane);
        // the step wave's decision (it does not wait for this wave: no barrier): poll the word of this lane's env.  Bounded -- a
        // protocol error must end as a wrong observation the parity tests catch, not as a hung GPU
        float flag = 0.0f;
        for (int spin = 0; spin < (1 << 20); ++spin) {
            // (an LDS-address-space pointer: through a generic `volatile float *` hipcc emits a FLAT load -- it queues behind the
            // window requests just issued and its s_waitcnt vmcnt(0) waits for all of them to land before the word is even seen)
            flag = *(volatile __attribute__((address_space(3))) float *)(__attribute__((address_space(3))) float *)(lk + 48 + (lane >> 4) * 4 + 3);
            if (__builtin_amdgcn_ballot_w64(flag == 0.0f) == 0ull) break;
            __builtin_amdgcn_s_sleep(1);
        }
        K1_LITE(2);
        const bool resets = flag == 2.0f;
        const unsigned long long rmask = __builtin_amdgcn_ballot_w64(resets);
        if (__builtin_expect(rmask != 0ull, 0)) {   // wave-uniform, rare: an env of this wave resets -> the window of the spawn pose this wave drew for it
            float pr[3] = {pf[0], pf[1], pf[2]}, qr[4] = {qf[0], qf[1], qf[2], qf[3]};
            if (resets) {
                const float4 *o = reinterpret_cast<const float4 *>(lk + 256 + (lane >> 4) * 12);
                const float4 o0 = o[0], o1 = o[1];
                pr[0] = o0.x; pr[1] = o0.y; pr[2] = o0.z;
                qr[0] = o0.w; qr[1] = 0.0f; qr[2] = 0.0f; qr[3] = o1.x;
            }
            const ScanWindow swr = scan_window(p, pr, qr);   // (the same function of the same pose for the lanes that do not reset)
            private_windows(swr, w);
            uniform_windows(w);
            windows_to_lds(const_cast<float *>(win), swr, lane);
            // restage what was requested for a pose that is no longer the final one: the earlier copy into the same tile must have
            // landed first (two LDS-DMA streams into one tile would interleave)
            if (n_env > 0 && (rmask & 0x1ull)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); private_issue(p, w, 0, tile0, lane); }
            if (!RV_OWN_TILES && n_env > 1 && (rmask & 0x10000ull)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); private_issue(p, w, 1, tile1, lane); }
        } else {
            windows_to_lds(const_cast<float *>(win), sw, lane);
        }
        if (RV_OWN_TILES) {
            // "the final windows stand in `win`": the reset word once more (0 -> 1 | 2 by the step wave -> 3 here; a wave's LDS operations
            // complete in order).  The step wave polls it behind its tail and then stages and casts envs 1 and 3 from ITS tile
            if ((lane & 15) == 0)
                *(volatile __attribute__((address_space(3))) float *)(__attribute__((address_space(3))) float *)(lk + 48 + (lane >> 4) * 4 + 3) = 3.0f;
        }
    }
    K1_LITE(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    K1_LITE(4);
    if (RV_OWN_TILES) {
        // EACH WAVE OWNS A TILE (round 5): this wave stages and casts envs 0 and 2 through tile 0, the step wave envs 1 and 3 through tile 1,
        // each at its own pace -- no barrier behind A.  (The barrier form below had this wave request all four windows and the two waves
        // meet at B, C, D: the step wave idled ~4 k cycles at B behind its tail while this wave's chain ran.)
        if (n_env > 0) private_cast<TRI, 0, PRIVATE_ROUNDS>(p, w, 0, tile0, lane, e_base, obs, p.obs_w, 4, oxy, ray_xy);
        K1_LITE(5);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wave's reads of tile 0 have returned: the tile is free
        K1_LITE(6);
        K1_LITE(7);
        if (n_env > 2) private_issue(p, w, 2, tile0, lane);
        K1_LITE(8);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        K1_LITE(9);
        K1_LITE(10);
        K1_LITE(11);
        if (n_env > 2) private_cast<TRI, 0, PRIVATE_ROUNDS>(p, w, 2, tile0, lane, e_base, obs, p.obs_w, 4, oxy, ray_xy);
        K1_LITE(12);
        K1_LITE(13);
        K1_LITE(14);
        K1_LITE(15);
        return;
    }
    // env 0 under the step wave's manager tail.  (Measured, us per step at 4096 envs: this order 33.8; window 2 requested before
    // barrier B as well 34.35 -- the step wave waits for the request's issue; window 1 requested only after env 0's cast 34.35.)
    if (n_env > 0) private_cast<TRI, 0, PRIVATE_ROUNDS>(p, w, 0, tile0, lane, e_base, obs, p.obs_w, 4, oxy, ray_xy);
    K1_LITE(5);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    K1_LITE(6);
    __syncthreads();                                                    // B: env 0 is cast, tile 0 is free
    K1_LITE(7);
    if (n_env > 2) private_issue(p, w, 2, tile0, lane);
    if (n_env > 1) private_cast<TRI, SHARE_1, PRIVATE_ROUNDS>(p, w, 1, tile1, lane, e_base, obs, p.obs_w, 4, oxy, ray_xy);
    K1_LITE(8);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    K1_LITE(9);
    __syncthreads();                                                    // C: window 2 has landed, tile 1 is free
    K1_LITE(10);
    if (n_env > 3) private_issue(p, w, 3, tile1, lane);
    K1_LITE(11);
    if (n_env > 2) private_cast<TRI, SHARE_2, PRIVATE_ROUNDS>(p, w, 2, tile0, lane, e_base, obs, p.obs_w, 4, oxy, ray_xy);
    K1_LITE(12);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    K1_LITE(13);
    __syncthreads();                                                    // D: window 3 has landed
    K1_LITE(14);
    if (n_env > 3) private_cast<TRI, SHARE_3, PRIVATE_ROUNDS>(p, w, 3, tile1, lane, e_base, obs, p.obs_w, 4, oxy, ray_xy);
    K1_LITE(15);
}

// ================================================================================================ K1g: step, group mapping
// Sixteen lanes per env: a wave holds 4 envs, a 256-thread workgroup 16.  The waves of a workgroup never talk to each
// other; the workgroup exists for PLACEMENT: its four waves go to the four SIMDs of one CU, so at N = 4096 (256 workgroups
// on 256 CUs) every SIMD issues for exactly one wave -- with single-wave workgroups two waves can land on one SIMD, where
// the younger one starves behind the older one's VALU stream (measured: 61 us instead of 30 us).
// Physics: one wheel slot x role per lane, chassis replicated; MDP tail (terminations, rewards, reset, command):
// replicated in the 16 lanes, stored by lane 0 of the group.
// FUSE: 0 = the step alone (the scan kernel follows as a second launch and reads the 32-byte descriptors); 1 / 2 = the height
// scan of the wave's four envs (bilinear patch / triangle mesh, int16 terrain copy) is the LAST PHASE of the same wave, from two
// wave-private LDS tiles (scan_private_wave): one launch per env step, no descriptor round trip, no second dispatch.
// BEGIN_ONLY: the first half of the two-phase step (rover_step_begin; see step_lane_body): everything up to the reset decision
// is stored -- physical state, manager words, reward, flags, force rows -- and nothing is reset.
template <int FUSE, bool BEGIN_ONLY = false>
__device__ __forceinline__ void step_group_body(const RvParams &p, float *__restrict__ state, const float *__restrict__ action,
                                                float *__restrict__ obs, float *__restrict__ reward, uint8_t *__restrict__ terminated,
                                                uint8_t *__restrict__ truncated, float *__restrict__ force,
                                                float *__restrict__ log_partial, float *lds, const float2 *__restrict__ ray_xy)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (RV_K1G_THREADS / 64) + (threadIdx.x >> 6);
    const int e_raw = wave * 4 + (lane >> 4);
    const bool active = e_raw < p.n;
    const int e = active ? e_raw : p.n - 1;
    const int N = p.n;
    const rover_config &c = p.cfg;
    const SlotConst sc = d_SLOT[lane & 7];
    const GroupIds id = group_ids(lane, sc);
    K1_STAMP(0);
    K1_LITE(0);

    const StepConsts &K = p.K;
    GroupLane g;
    group_load(state, N, e, id, sc, K, g, p.cfg.mass_model);
    // rover_env.py:62 ActionManager.process_action
    float act[2], prev[2];
    prev[0] = state[(size_t)(ROVER_ACTION + 0) * N + e];
    prev[1] = state[(size_t)(ROVER_ACTION + 1) * N + e];
    const float2 a = reinterpret_cast<const float2 *>(action)[e];
    act[0] = a.x;
    act[1] = a.y;
    float S[ROVER_STATE_WORDS];
    {
        float processed[2], steer[4], rim[6];
        ackermann_core(c, act, processed, steer, rim);
        const float steer_m[4] = {steer[0], steer[3], steer[1], steer[2]};              // FL, FR, RL, RR
        const float rim_m[6] = {rim[1], rim[5], rim[0], rim[4], rim[2], rim[3]};        // FL, FR, CL, CR, RL, RR
        float st = steer_m[0], rv = rim_m[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) st = (id.si == i) ? steer_m[i] : st;
#pragma unroll
        for (int i = 1; i < 6; ++i) rv = (id.k == i) ? rim_m[i] : rv;
        g.steer_t = st;
        g.wheel_t = rv / c.wheel_radius;   // one division per lane instead of six (same quotient)
    }
    // rover_env.py:64-72 decimation loop
    float Fw[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};   // [0..2] this wheel's force, [3..5] this lane's link-point force
    f2 oxy[PRIVATE_ROUNDS];   // the scan phase's ray table (one-launch forms)
    K1_STAMP(1);
    for (int s = 0; s < c.decimation - 1; ++s) physics_substep_group<false>(p, K, g, nullptr, s);
    // manager words: requested HERE, in front of the last substep, so that their latency hides under it -- requested at the top of the
    // kernel (rounds 2 - 4) they held 26 VGPRs through all six substeps of a kernel that sits at 256 VGPRs (-0.4 us per step)
#pragma unroll
    for (int i = ROVER_TARGET_W; i < ROVER_ACTION; ++i) S[i] = state[(size_t)i * N + e];
#pragma unroll
    for (int i = ROVER_TIME_LEFT; i < ROVER_LAMBDA_N; ++i) S[i] = state[(size_t)i * N + e];
    S[ROVER_RESET_COUNT] = state[(size_t)ROVER_RESET_COUNT * N + e];
    if constexpr (FUSE == 1 || FUSE == 2) {
        // copy-wave form: the link-body sample points of the contact report (evaluated at the pose of the last substep's START)
        // are the copy wave's work -- it is asleep until now; this wave goes straight into the substep
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        float *lk = fused_link(lds, p, wv);
        if ((lane & 15) == 0) {
            float4 *d = reinterpret_cast<float4 *>(lk + (lane >> 4) * 12);
            d[0] = make_float4(g.R[0][0], g.R[0][1], g.R[0][2], g.R[1][0]);
            d[1] = make_float4(g.R[1][1], g.R[1][2], g.R[2][0], g.R[2][1]);
            d[2] = make_float4(g.R[2][2], g.pos[0], g.pos[1], g.pos[2]);
        }
        lk[64 + lane] = g.bq;
        __syncthreads();                                                // L
        if (c.decimation > 0) physics_substep_group<true, true>(p, K, g, Fw, c.decimation - 1);
    } else {
        if (c.decimation > 0) physics_substep_group<true>(p, K, g, Fw, c.decimation - 1);
    }
    K1_STAMP(20);
    K1_LITE(1);
    if constexpr (FUSE == 1 || FUSE == 2) {
        // the pose the physics left: handed to the copy wave, which derives the scan windows itself and requests the first two right
        // behind barrier A -- before the reset decision exists; an env that then resets (rare) gets its window restaged
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        float *lk = fused_link(lds, p, wv);
        if ((lane & 15) == 0) {
            float4 *d = reinterpret_cast<float4 *>(lk + 48 + (lane >> 4) * 4);      // [48, 64): position (x, y, z, -)
            float4 *q = reinterpret_cast<float4 *>(lk + 304 + (lane >> 4) * 4);     // [304, 320): quaternion
            d[0] = make_float4(g.pos[0], g.pos[1], g.pos[2], 0.0f);
            q[0] = make_float4(g.quat[0], g.quat[1], g.quat[2], g.quat[3]);
        }
        __syncthreads();                                                // A
        if (c.decimation > 0) {
            const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const float *lk = fused_link(lds, p, wv);
            Fw[3] = lk[320 + lane]; Fw[4] = lk[384 + lane]; Fw[5] = lk[128 + lane];
            if (!(lk[192 + lane] > RV_OBSTACLE_EPS)) { Fw[0] = 0.0f; Fw[1] = 0.0f; Fw[2] = 0.0f; }   // the wheel is not on the obstacle layer
        }
    }
    if (!id.wheel_active) { Fw[0] = 0.0f; Fw[1] = 0.0f; Fw[2] = 0.0f; Fw[3] = 0.0f; Fw[4] = 0.0f; Fw[5] = 0.0f; }
    // contact report: gather the six Drive-body forces and the twelve link-point forces of the env (sensor body order) into
    // every lane -- only in waves where some body touches the obstacle layer (otherwise every force is the +0 the array holds)
    float F[ROVER_NUM_BODIES * 3];
#pragma unroll
    for (int i = 0; i < ROVER_NUM_BODIES * 3; ++i) F[i] = 0.0f;
    // (the link points' z component only: penetration is what makes any component of a point force non-zero)
    const bool any_force = __ballot(Fw[0] != 0.0f || Fw[1] != 0.0f || Fw[2] != 0.0f || Fw[5] != 0.0f) != 0ull;
    if (__builtin_expect(any_force, 0)) {   // (rare paths out of line: the common case falls through, a taken branch costs ~15 cycles)
        constexpr int BODY_SLOT[6] = {1, 3, 0, 2, 4, 5};  // bodies 7..12 = CL, CR, FL, FR, RL, RR -> solver slot
        const int base = lane & ~15;
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int i = 0; i < 3; ++i) F[(7 + b) * 3 + i] = __shfl(Fw[i], base + BODY_SLOT[b], 64);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float lf[6][2], Fb[7];
#pragma unroll
            for (int sl = 0; sl < 6; ++sl) {
                lf[sl][0] = __shfl(Fw[3 + i], base + sl, 64);
                lf[sl][1] = __shfl(Fw[3 + i], base + 8 + sl, 64);
            }
            link_body_forces(lf, Fb);
#pragma unroll
            for (int b = 0; b < 7; ++b) F[b * 3 + i] = Fb[b];
        }
    }
    int coll_known = -1;   // one-launch forms: the collision flag, evaluated ahead of the other terms
    if constexpr (FUSE == 1 || FUSE == 2) {
        // The reset is decided NOW -- time-out / success / far are functions of words loaded before the physics (B-13: the terms
        // see the previous step's command), the collision flag of the report just gathered -- so the FINAL pose is known here: the
        // pose the physics left, or the spawn pose the copy wave drew.  Its windows go to the copy wave, which stages them and
        // casts the first env under the rest of the manager tail.
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const bool coll = any_force ? collision_measure(F) > 1.0f : false;
        coll_known = coll ? 1 : 0;
        bool term_e[ROVER_NUM_TERM];
        mdp_terminations(c, S + ROVER_CMD_B, __float_as_int(S[ROVER_EP_LEN]) + 1, coll, term_e);
        // the decision goes to the copy wave, which owns the windows: one word per env (word 3 of the position slot; 0 = not decided
        // yet -- written with the pose before barrier A --, 1 = the pose stands, 2 = the env resets).  NO barrier here: the copy wave is
        // busy issuing its window requests (a wave's global_load_lds issue is blocking) and polls the word when it is done
        if ((lane & 15) == 0)
            *(volatile __attribute__((address_space(3))) float *)(__attribute__((address_space(3))) float *)(fused_link(lds, p, wv) + 48 + (lane >> 4) * 4 + 3) =
                (term_e[0] | term_e[1] | term_e[2] | term_e[3]) ? 2.0f : 1.0f;   // (ds_write: a generic volatile pointer gives a FLAT store)
        K1_LITE(2);
        // the ray table of the scan phase: requested now, so that it arrives under the manager tail
#pragma unroll
        for (int m = 0; m < PRIVATE_ROUNDS; ++m) {
            oxy[m] = (f2){0.0f, 0.0f};
            if (m < SHARE_MAX) {
                const float2 v = ray_xy[m * 64 + lane];
                oxy[m] = (f2){v.x, v.y};
            }
        }
    }
    K1_LITE_F(11);
    if (active) group_store(state, N, e, id, g, lane);
    K1_LITE_F(12);
    if (force && active && id.owner) {
#pragma unroll
        for (int i = 0; i < 3; ++i) *soa_word(force, id.body * 3 + i, N, e) = Fw[i];
    }
    if (force && active && (lane & 15) < 7) {   // the three rows of the seven link bodies
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float fb = F[i];
#pragma unroll
            for (int b = 1; b < 7; ++b) fb = ((lane & 15) == b) ? F[b * 3 + i] : fb;
            *soa_word(force, (lane & 15) * 3 + i, N, e) = fb;
        }
    }
    K1_STAMP(23);
    K1_LITE_F(13);

    // ---- MDP tail on the manager words (replicated in the group; loaded before the physics, see above)
#pragma unroll
    for (int i = 0; i < 3; ++i) S[ROVER_POS + i] = g.pos[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) S[ROVER_QUAT + i] = g.quat[i];
    S[ROVER_ACTION] = act[0]; S[ROVER_ACTION + 1] = act[1];
    S[ROVER_PREV_ACTION] = prev[0]; S[ROVER_PREV_ACTION + 1] = prev[1];
    // words reset_one may clear in reset_mode 1 (stored by their owner lanes only when that happens)
#pragma unroll
    for (int i = ROVER_LINVEL; i < ROVER_TARGET_W; ++i) S[i] = 0.0f;
#pragma unroll
    for (int i = ROVER_LAMBDA_N; i < ROVER_LAMBDA_N + 6; ++i) S[i] = 0.0f;

    const int ep_len = __float_as_int(S[ROVER_EP_LEN]) + 1;
    S[ROVER_EP_LEN] = __int_as_float(ep_len);
    float rew[ROVER_NUM_REW];
    bool term[ROVER_NUM_TERM];
    mdp_terms_one(c, S + ROVER_CMD_B, S + ROVER_ACTION, S + ROVER_PREV_ACTION, ep_len, F, rew, term, !any_force, coll_known);
    K1_STAMP(24);
    K1_LITE_F(14);
    const bool time_out = term[0];
    const bool term_any = term[1] | term[2] | term[3];
    const float step_dt = c.sim_dt * (float)c.decimation;
    float total = 0.0f;
#pragma unroll
    for (int i = 0; i < ROVER_NUM_REW; ++i) {
        if (c.rew_weight[i] != 0.0f) {
            const float val = rew[i] * c.rew_weight[i] * step_dt;
            total += val;
            S[ROVER_EP_SUM + i] += val;
        }
    }
    const bool do_reset = term_any | time_out;
    const bool writer = active && (lane & 15) == 0;
    if constexpr (BEGIN_ONLY) {
        if (active) {
            store_rows_by_lane<ROVER_TARGET_W, 16, ROVER_TARGET_W>(state, N, e, lane & 15, S);
            store_rows_by_lane<ROVER_TARGET_W + 16, 10, ROVER_TARGET_W + 16>(state, N, e, lane & 15, S);
        }
        if (writer) {
            reward[e] = total;
            terminated[e] = term_any ? 1 : 0;
            truncated[e] = time_out ? 1 : 0;
        }
        return;
    }
    // episodic log contributions (before the reset clears the sums): the manager words are replicated in the env's sixteen lanes,
    // so lane r of the group forms word r of its env's contribution itself -- no transposition, ONE value per lane to reduce
    const bool any_reset = __ballot(do_reset && writer) != 0ull;
    float lgx = 0.0f;
    if (__builtin_expect(any_reset, 0) && do_reset && active) {
        const float lgv[14] = {S[ROVER_EP_SUM + 0], S[ROVER_EP_SUM + 1], S[ROVER_EP_SUM + 2], S[ROVER_EP_SUM + 3], S[ROVER_EP_SUM + 4],
                               S[ROVER_EP_SUM + 5], S[ROVER_EP_SUM + 6], term[0] ? 1.0f : 0.0f, term[1] ? 1.0f : 0.0f,
                               term[2] ? 1.0f : 0.0f, term[3] ? 1.0f : 0.0f, S[ROVER_METRIC_POS], S[ROVER_METRIC_HEAD], 1.0f};
        static_assert(ROVER_NUM_REW == 7 && ROVER_NUM_TERM == 4, "log row layout");
        const float x = pick_by_lane<1, 14, 0, 14>(lane & 15, lgv, lgv[0]);
        lgx = (lane & 15) < 14 ? x : 0.0f;
    }
    const uint32_t gid = (uint32_t)(p.env_id_offset + e);
    if (__builtin_expect(any_reset, 0) && do_reset) {   // (any_reset: some env of the wave resets -- wave-uniform, rare)
        if constexpr (FUSE == 1 || FUSE == 2) {   // drawn by the copy wave during the physics (scan_copy_wave)
            const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const float4 *d = reinterpret_cast<const float4 *>(fused_link(lds, p, wv) + 256 + (lane >> 4) * 12);
            const float4 d0 = d[0], d1 = d[1], d2 = d[2];
            const ResetOutcome ro = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x};
            reset_apply(p, S, ro);
        } else {
            reset_one(p, S, gid);
        }
        if (active) {
            // the reset rewrites the root pose (lane 0 stores it below), the contact cache of every wheel and, in
            // reset_mode 1, all velocities / joint words
            if (id.owner) state[(size_t)(ROVER_LAMBDA_N + id.k) * N + e] = 0.0f;
            if (c.reset_mode == 1) {
                if (id.owner) {
                    state[(size_t)(ROVER_WHEEL_Q + id.k) * N + e] = 0.0f;
                    state[(size_t)(ROVER_WHEEL_QD + id.k) * N + e] = 0.0f;
                    if (id.si >= 0) {
                        state[(size_t)(ROVER_STEER_Q + id.si) * N + e] = 0.0f;
                        state[(size_t)(ROVER_STEER_QD + id.si) * N + e] = 0.0f;
                    }
                    if ((id.slot & 1) == 0) {
                        state[(size_t)(ROVER_BOGIE_Q + id.j) * N + e] = 0.0f;
                        state[(size_t)(ROVER_BOGIE_QD + id.j) * N + e] = 0.0f;
                    }
                }
                // chassis velocities: by the lanes that stored them in group_store (lane r owns row r: same lane, same address,
                // program order)
                if ((lane & 15) >= ROVER_LINVEL && (lane & 15) < ROVER_BOGIE_Q) *soa_word(state, lane & 15, N, e) = 0.0f;
            }
        }
    }
    K1_STAMP(21);
    K1_LITE_F(15);
    command_compute(p, S, gid, step_dt);
    K1_STAMP(22);
    K1_LITE_F(16);

    if (__builtin_expect(any_reset, 0)) {   // this wave's row of the log partials, tagged with the step, and the flag for the reduction
        // Sum over the wave's four envs in the order of the 64-lane butterfly this replaces -- (env 0 + env 2) + (env 1 + env 3),
        // then the butterfly's four additions of the other lanes' +0, which only ever turn a -0 into +0: one addition of +0.
        float x = lgx;
        x += __shfl_xor(x, 32, 64);
        x += __shfl_xor(x, 16, 64);
        x = x + 0.0f;
        if (lane < 16) log_partial[(size_t)wave * ROVER_LOG_WORDS + lane] = lane == 15 ? __uint_as_float(p.step_tag) : x;
        if (lane == 0) {
            p.log_counter[0] = 1u;           // "rows to reduce" (the reduction tests it against zero and returns it to zero)
            p.log_counter[1] = p.step_tag;   // the latest launch with resets (every wave of a launch stores the same value)
        }
    }
    if (active) {   // the manager words are replicated in the env's sixteen lanes: lane r stores row W0 + r (store_rows_by_lane)
        const int r = lane & 15;
        {
            float *o = obs + (size_t)e * p.obs_w;   // write_obs_head, one element per lane
            const float cbx = S[ROVER_CMD_B], cby = S[ROVER_CMD_B + 1];
            const float head[4] = {S[ROVER_ACTION], S[ROVER_ACTION + 1], sqrtf(cbx * cbx + cby * cby) * p.cfg.obs_scale_distance,
                                   rv_atan2f(cby, cbx) * p.cfg.obs_scale_heading};
            const float x = pick_by_lane<1, 4, 0, 4>(r, head, head[0]);
            if (r < 4) o[r] = x;
        }
        if (FUSE == 0 && r == 0) write_scan_desc(p, S + ROVER_POS, S + ROVER_QUAT, e);
        if (do_reset) store_rows_by_lane<ROVER_POS, 7, ROVER_POS>(state, N, e, r, S);
        static_assert(ROVER_LAMBDA_N - ROVER_TARGET_W == 26, "manager words 39..64");
        store_rows_by_lane<ROVER_TARGET_W, 16, ROVER_TARGET_W>(state, N, e, r, S);
        store_rows_by_lane<ROVER_TARGET_W + 16, 10, ROVER_TARGET_W + 16>(state, N, e, r, S);
        if (r == 0) {
            state[(size_t)ROVER_RESET_COUNT * N + e] = S[ROVER_RESET_COUNT];
            reward[e] = total;
            terminated[e] = term_any ? 1 : 0;
            truncated[e] = time_out ? 1 : 0;
        }
    }
    K1_STAMP(25);
    if constexpr (FUSE == 3 || FUSE == 4) {   // no copy waves, one tile per wave (more than one round of workgroups)
        const int tile_cells = p.tile_dim * p.tile_pitch;
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        int16_t *tile = reinterpret_cast<int16_t *>(lds) + (size_t)wv * tile_cells;
        PrivateWindows pw;
        private_windows(scan_window(p, S + ROVER_POS, S + ROVER_QUAT), pw);
        scan_single_tile_wave<FUSE == 4>(p, tile, lane, max(0, min(4, p.n - wave * 4)), wave * 4, pw, obs, p.obs_w, 4, ray_xy);
        K1_STAMP(26);
    }
    if constexpr (FUSE == 1 || FUSE == 2) {
        // ---- height scan of the wave's envs 1, 2, 3 (env 0: the copy wave's, under the tail above; see scan_copy_wave for the protocol)
        const int tile_cells = p.tile_dim * p.tile_pitch;
        const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        int16_t *tile0 = reinterpret_cast<int16_t *>(lds) + (size_t)(2 * wv) * tile_cells, *tile1 = tile0 + tile_cells;
        float *win = fused_win(lds, p, wv);
        const int n_scan = max(0, min(4, p.n - wave * 4));
        const int e_base = wave * 4;
        K1_LITE(3);
        if (!RV_OWN_TILES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ray table (own-tiles form: the cast's first use waits for it; the stores need not retire before the window request)
        K1_LITE(4);
        if (RV_OWN_TILES) {
            // this wave owns tile 1: envs 1 and 3, staged and cast here, at this wave's own pace (scan_copy_wave: envs 0 and 2 through tile 0).
            // The windows are the copy wave's: read once its word says they stand (bounded poll: a protocol error ends as a wrong
            // observation the parity tests catch, not as a hung GPU)
            PrivateWindows pw;
            {
                const float *lk = fused_link(lds, p, wv);
                for (int spin = 0; spin < (1 << 20); ++spin) {
                    const float f = *(volatile __attribute__((address_space(3))) float *)(__attribute__((address_space(3))) float *)(const_cast<float *>(lk) + 48 + (lane >> 4) * 4 + 3);
                    if (__builtin_amdgcn_ballot_w64(f != 3.0f) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            windows_from_lds(win, pw);
            K1_LITE(5);
            K1_STAMP(27);
            if (n_scan > 1) {
                private_issue(p, pw, 1, tile1, lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                private_cast<FUSE == 2, 0, PRIVATE_ROUNDS>(p, pw, 1, tile1, lane, e_base, obs, p.obs_w, 4, oxy, ray_xy);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads of tile 1 have returned: the tile is free
            }
            K1_STAMP(28);
            K1_LITE(6);
            K1_LITE(7);
            if (n_scan > 3) {
                private_issue(p, pw, 3, tile1, lane);
                K1_LITE(8);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                K1_STAMP(29);
                K1_LITE(9);
                private_cast<FUSE == 2, 0, PRIVATE_ROUNDS>(p, pw, 3, tile1, lane, e_base, obs, p.obs_w, 4, oxy, ray_xy);
            }
            K1_STAMP(26);
            K1_LITE(10);
            return;
        }
        __syncthreads();                                                // B: the copy wave has cast env 0 and requested window 2
        K1_LITE(5);
        // (barrier form) the windows of the final poses: written by the copy wave (it owns them), read behind barrier B
        PrivateWindows pw;
        windows_from_lds(win, pw);
        K1_STAMP(27);
        if (n_scan > 1) private_cast<FUSE == 2, 0, SHARE_1>(p, pw, 1, tile1, lane, e_base, obs, p.obs_w, 4, oxy, ray_xy);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        K1_STAMP(28);
        K1_LITE(6);
        __syncthreads();                                                // C: window 2 has landed, tile 1 is free
        K1_LITE(7);
        if (n_scan > 2) private_cast<FUSE == 2, 0, SHARE_2>(p, pw, 2, tile0, lane, e_base, obs, p.obs_w, 4, oxy, ray_xy);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        K1_STAMP(29);
        K1_LITE(8);
        __syncthreads();                                                // D: window 3 has landed
        K1_LITE(9);
        if (n_scan > 3) private_cast<FUSE == 2, 0, SHARE_3>(p, pw, 3, tile1, lane, e_base, obs, p.obs_w, 4, oxy, ray_xy);
        K1_STAMP(26);
        K1_LITE(10);
    }
}
// first half of the two-phase step, sixteen lanes per env (spill-free; the one-env-per-lane form of the same phase would carry
// 740 bytes of scratch per lane)
__global__ __launch_bounds__(RV_K1G_THREADS) void rover_step_begin_kernel(RvParams p, float *__restrict__ state,
                                                                          const float *__restrict__ action, float *__restrict__ reward,
                                                                          uint8_t *__restrict__ terminated, uint8_t *__restrict__ truncated,
                                                                          float *__restrict__ force)
{
    step_group_body<0, true>(p, state, action, nullptr, reward, terminated, truncated, force, nullptr, nullptr, nullptr);
}
__global__ __launch_bounds__(RV_K1G_THREADS) void rover_step_kernel_group(RvParams p, float *__restrict__ state,
                                                              const float *__restrict__ action, float *__restrict__ obs,
                                                              float *__restrict__ reward, uint8_t *__restrict__ terminated,
                                                              uint8_t *__restrict__ truncated, float *__restrict__ force,
                                                              float *__restrict__ log_partial)
{
    step_group_body<0>(p, state, action, obs, reward, terminated, truncated, force, log_partial, nullptr, nullptr);
}
// One launch per env step: the group-mapped step with the height scan as its last phase (TRI: triangle-mesh surface).
template <bool TRI>
__global__ __launch_bounds__(2 * RV_K1G_THREADS) RV_FUSED_ATTR void rover_step_scan_kernel(
    RvParams p, float *__restrict__ state, const float *__restrict__ action, float *__restrict__ obs, float *__restrict__ reward,
    uint8_t *__restrict__ terminated, uint8_t *__restrict__ truncated, float *__restrict__ force, float *__restrict__ log_partial,
    const float2 *__restrict__ ray_xy)
{
    extern __shared__ __align__(16) float lds[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wv < RV_K1G_THREADS / 64) {  // waves 0..3: the step (threadIdx.x < 256: the group kernel's own indexing); waves 4..7: their copy waves
        // (s_setprio 3 for the step wave during the physics, so that its copy wave only gets the issue slots it leaves empty: measured,
        // 83.8 k vs 84.0 k stamped cycles -- nothing)
        step_group_body<TRI ? 2 : 1>(p, state, action, obs, reward, terminated, truncated, force, log_partial, lds, ray_xy);
    } else {
        scan_copy_wave<TRI>(p, state, lds, wv - RV_K1G_THREADS / 64, (int)(threadIdx.x & 63), obs, ray_xy);
    }
}

__global__ __launch_bounds__(RV_K1G_THREADS) void rover_physics_kernel_group(RvParams p, float *__restrict__ state,
                                                                             const float *steer_t, const float *wheel_t,
                                                                             int substeps, float *force)
{
    const int lane = threadIdx.x & 63;
    const int e_raw = (blockIdx.x * (RV_K1G_THREADS / 64) + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool active = e_raw < p.n;
    const int e = active ? e_raw : p.n - 1;
    const int N = p.n;
    const SlotConst sc = d_SLOT[lane & 7];
    const GroupIds id = group_ids(lane, sc);
    const StepConsts &K = p.K;
    GroupLane g;
    group_load(state, N, e, id, sc, K, g, p.cfg.mass_model);
    g.steer_t = steer_t[4 * e + (id.si >= 0 ? id.si : 0)];
    g.wheel_t = wheel_t[6 * e + id.k];
    float Fw[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int s = 0; s < substeps - 1; ++s) physics_substep_group<false>(p, K, g, nullptr);
    if (substeps > 0) physics_substep_group<true>(p, K, g, Fw);
    if (!id.wheel_active) { Fw[3] = 0.0f; Fw[4] = 0.0f; Fw[5] = 0.0f; }
    if (active) group_store(state, N, e, id, g, lane);
    if (force && active && id.owner) {
#pragma unroll
        for (int i = 0; i < 3; ++i) force[(size_t)(id.body * 3 + i) * N + e] = Fw[i];
    }
    if (force) {
        const int base = lane & ~15;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float lf[6][2], Fb[7];
#pragma unroll
            for (int sl = 0; sl < 6; ++sl) {
                lf[sl][0] = __shfl(Fw[3 + i], base + sl, 64);
                lf[sl][1] = __shfl(Fw[3 + i], base + 8 + sl, 64);
            }
            link_body_forces(lf, Fb);
            if (active && (lane & 15) < 7) {
                float fb = Fb[0];
#pragma unroll
                for (int b = 1; b < 7; ++b) fb = ((lane & 15) == b) ? Fb[b] : fb;
                force[(size_t)((lane & 15) * 3 + i) * N + e] = fb;
            }
        }
    }
}

// reset of every env (env.reset()): reset_one + _update_command, no physics.  With `mask` / draws (rover_reset_with_draws):
// RLTaskEnv._reset_idx of the masked envs with the supplied uniforms in place of the Philox draws.
__global__ __launch_bounds__(64) void rover_reset_kernel(RvParams p, float *__restrict__ state, const uint8_t *mask,
                                                         const int32_t *spawn_row, const float *yaw_u, const float *theta_u,
                                                         const float *heading_u)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= p.n) return;
    if (mask && !mask[e]) return;
    const int N = p.n;
    float S[ROVER_STATE_WORDS];
#pragma unroll
    for (int i = 0; i < ROVER_STATE_WORDS; ++i) S[i] = state[(size_t)i * N + e];
    if (spawn_row) {
        ResetDraws d;
        d.spawn_row = spawn_row[e];
        d.yaw_u = yaw_u[e];
        d.theta_u = theta_u + (size_t)e * p.cfg.max_target_tries;
        d.heading_u = heading_u[e];
        reset_one(p, S, (uint32_t)(p.env_id_offset + e), &d);
    } else {
        reset_one(p, S, (uint32_t)(p.env_id_offset + e));
    }
    update_command_one(S + ROVER_POS, S + ROVER_QUAT, S + ROVER_TARGET_W, S[ROVER_HEADING_CMD_W], S + ROVER_CMD_B,
                       S + ROVER_HEADING_CMD_B);
#pragma unroll
    for (int i = 0; i < ROVER_STATE_WORDS; ++i) state[(size_t)i * N + e] = S[i];
}

// ================================================================================================ K2: scan + obs
// Persistent workgroups, one env at a time.  out row = out + env * row_stride; scan values start at column col0.
//   MODE 0  pose from the state tensor, scan columns only                                   (rover_height_scan)
//   MODE 1  pose from the state tensor + observation head                                   (rover_reset)
//   MODE 2  pose + terrain window from the 32-byte descriptor the step kernel left (one scalar load), the head was
//           written by the step kernel; the LAST workgroup reduces the log partials          (rover_step)
// extras["log"]: deterministic reduction of the per-wave log partials by the threads of ONE workgroup (workgroup 0 of the
// scan kernel, after its last env): rows carrying this step's tag, GROUPS x 16 words, then a fixed-order sum.  In a step
// without resets -- the common case -- the counter is zero and nothing is read.  (Until round 3 an EXTRA workgroup summed
// all n_waves rows every step; it was dispatched last, i.e. when the first persistent workgroup retired, so its whole
// duration sat on the kernel's tail: ~2 us at 4096 envs, and 150 us at 131072 envs after the group-mapped step kernel,
// whose waves hold 4 envs instead of 64 -- the "N = 131072 anomaly" of the round-2 sweep.)
template <int THREADS>
__device__ __forceinline__ void reduce_log_partials(const RvParams &p, float *lds, int tid, const float *__restrict__ log_partial,
                                                    int n_waves, float *__restrict__ log_out)
{
    const rover_config &c = p.cfg;
    // Rows carry the tag of the launch that wrote them; the reduction sums the rows of the LATEST launch in which an env reset
    // (log_counter[1]).  Run behind every step that is this step's tag whenever the counter is non-zero; run on demand
    // (rover_flush_log) it reproduces what the per-step reduction would hold: entries 0..12 from the latest step with resets,
    // entry 13 = that step's count if it IS the latest step, else 0.
    const unsigned resets = *reinterpret_cast<volatile unsigned *>(p.log_counter);
    if (resets == 0u) {
        if (tid == 0) log_out[13] = 0.0f;
        return;
    }
    const unsigned latest = *reinterpret_cast<volatile unsigned *>(p.log_counter + 1);
    constexpr int GROUPS = THREADS / 16;
    const int word = tid & 15, grp = tid >> 4;
    // rows w = grp, grp + GROUPS, ... summed in that order; sixteen rows' loads are issued together (a loop of dependent
    // load -> add rounds paid one memory latency per row: ~4 us for the 1024 rows of 4096 envs).  A row past the end contributes
    // +0, which leaves the sum as it is (the sum starts at +0 and can never be -0).
    float acc = 0.0f;
    constexpr int U = 16;
    for (int w0 = grp; w0 < n_waves; w0 += GROUPS * U) {
        unsigned tag[U];
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int w = w0 + u * GROUPS;
            const bool in = w < n_waves;
            const size_t row = (size_t)(in ? w : grp) * ROVER_LOG_WORDS;
            tag[u] = in ? __float_as_uint(log_partial[row + 15]) : 0u;   // tags start at 1: 0 never matches
            v[u] = log_partial[row + word];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += (tag[u] == latest) ? v[u] : 0.0f;
    }
    lds[grp * 16 + word] = acc;
    __syncthreads();
    if (tid < 16) {
        float s = 0.0f;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) s += lds[g * 16 + tid];
        lds[THREADS + tid] = s;
    }
    __syncthreads();
    if (tid < 14) {
        const float cnt = lds[THREADS + 13];
        if (tid == 13) {
            log_out[13] = latest == p.step_tag ? cnt : 0.0f;
        } else if (cnt > 0.0f) {
            const float s = lds[THREADS + tid];
            float val;
            if (tid < ROVER_NUM_REW) val = s / cnt / c.max_episode_length_s;  // Episode Reward/<term>
            else if (tid < 11) val = s;                                        // Episode Termination/<term>
            else val = s / cnt;                                                // Metrics/target_pose/*
            log_out[tid] = val;
        }
    }
    if (tid == 0) *p.log_counter = 0u;
}
#ifndef RV_K2_THREADS
#define RV_K2_THREADS 512   // 8 waves share the LDS tiles; <= 40 KiB of LDS per workgroup admits 4 workgroups = 32 waves per CU
#endif
//   TRI     the rays hit the triangle mesh of the heightfield (cfg.scan_surface = 0) instead of the bilinear patch
template <int MODE, bool Q16, bool TRI>
__global__ __launch_bounds__(RV_K2_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void rover_scan_obs_kernel(RvParams p, const float *__restrict__ state,
                                                             float *__restrict__ out, int row_stride, int col0,
                                                             const float *__restrict__ log_partial, int n_waves,
                                                             float *__restrict__ log_out,
                                                             const float *__restrict__ scan_desc)
{
    // scan_desc (= p.scan_desc) is a separate read-only, non-aliased argument so that the per-env descriptor is fetched
    // with scalar loads: they do not queue behind the in-flight tile copies on the vector-memory counter
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x;
    const int N = p.n;
    const rover_config &c = p.cfg;
    // persistent workgroups: the grid is sized to what the chip holds at once (launching one 8-wave workgroup per env
    // costs ~10 us of dispatch alone at N = 4096); each workgroup walks envs blockIdx.x, blockIdx.x + n_wg, ...
    const int n_wg = (int)gridDim.x;
#define RV_STAMP(k) do { } while (0)
    // LDS carve: [0, 64) ray x offsets, [64, 128) ray y offsets, [128, 192) reciprocals 1/t, [192, ...) tile_bufs terrain
    // tiles (16-B aligned) of fp32 heights or, when the terrain has an exact 16-bit copy, of int16 heights (half the bytes)
    using cell_t = typename std::conditional<Q16, int16_t, float>::type;
    constexpr int CC = Q16 ? 8 : 4;  // cells per 16-byte chunk
    typedef float v4f __attribute__((ext_vector_type(4)));
    float *ox_tab = lds, *oy_tab = lds + 64, *inv_tab = lds + 128;
    cell_t *tile_base = reinterpret_cast<cell_t *>(lds + 192);
    const size_t tile_cells = (size_t)p.tile_dim * p.tile_pitch;
    const cell_t *hsrc = Q16 ? reinterpret_cast<const cell_t *>(p.height_q) : reinterpret_cast<const cell_t *>(p.height);
    // ORBIT grid_pattern: arange(-size/2, size/2 + 1e-9, res) evaluated in double, x fastest (App. C)
    if (tid < c.scan_nx) ox_tab[tid] = (float)(-0.5 * (double)c.scan_size_x + (double)c.scan_resolution * (double)tid);
    if (tid >= 64 && tid < 64 + c.scan_ny)
        oy_tab[tid - 64] = (float)(-0.5 * (double)c.scan_size_y + (double)c.scan_resolution * (double)(tid - 64));
    if (tid >= 128 && tid < 192) inv_tab[tid - 128] = 1.0f / (float)max(tid - 128, 1);
    const float inv_res = p.inv_res;
    const bool vec_ok = ((p.W & (CC - 1)) == 0) && ((reinterpret_cast<uintptr_t>(hsrc) & 15) == 0);
    const bool two_bufs = p.tile_bufs == 2;
    const float x_max = p.x_max, y_max = p.y_max;
    const float inv_nx = p.inv_nx;
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);

    auto get_window = [&](int e) -> ScanWindow {
        ScanWindow w;
        if (MODE == 2) {
            const float4 *d = reinterpret_cast<const float4 *>(scan_desc + (size_t)e * 8);
            const float4 d0 = d[0], d1 = d[1];
            w.px = d0.x; w.py = d0.y; w.pz = d0.z; w.cy = d0.w; w.sy = d1.x;
            w.i_lo = __float_as_int(d1.y);
            w.j_lo = __float_as_int(d1.z);
            const int pk = __float_as_int(d1.w);
            w.th = pk & 0x7FFF;
            w.interior = (pk >> 15) & 1;
            w.tw4 = pk >> 16;
        } else {
            float pos[3], quat[4];
#pragma unroll
            for (int i = 0; i < 3; ++i) pos[i] = state[(size_t)(ROVER_POS + i) * N + e];
#pragma unroll
            for (int i = 0; i < 4; ++i) quat[i] = state[(size_t)(ROVER_QUAT + i) * N + e];
            w = scan_window(p, pos, quat);
        }
        return w;
    };
    // Stage the th x tw4 chunk window of env `w` DENSELY (row pitch = tw4 chunks): chunk k of the row-major list goes to
    // LDS chunk k, moved by an asynchronous global->LDS copy (global_load_lds_dwordx4: no VGPR staging, no ds_write; the
    // LDS address is wave base + lane * 16, which the dense list order satisfies).  The copies stay in flight while the
    // workgroup casts the rays of the previous env; `s_waitcnt vmcnt(0)` + barrier retire them before the tile is read.
    auto issue_tile = [&](const ScanWindow &w, cell_t *tile) {
        const int tw4 = w.tw4, th = w.th;
        if (vec_ok) {
            const v4f *src = reinterpret_cast<const v4f *>(hsrc + (size_t)w.i_lo * p.W + w.j_lo);
            v4f *dst = reinterpret_cast<v4f *>(tile);
            const int nchunk = th * tw4;
            const float inv_tw4 = inv_tab[tw4];
            for (int k0 = 0; k0 < nchunk; k0 += RV_K2_THREADS) {
                const int k = k0 + tid;
                if (k < nchunk) {
                    const int r = (int)(((float)k + 0.5f) * inv_tw4);  // k / tw4, exact for k < 2^20
                    const int cq = k - (int)__umul24(r, tw4);
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src + (size_t)(__umul24(r, p.wq) + cq)),
                        (__attribute__((address_space(3))) void *)(dst + (k0 + wave_base)), 16, 0, 0);
                }
            }
        } else {
            const int pitch = tw4 * CC, tw = min(pitch, p.W - w.j_lo);
            for (int r = tid >> 6; r < th; r += RV_K2_THREADS / 64)
                for (int cc = tid & 63; cc < tw; cc += 64)
                    tile[r * pitch + cc] = hsrc[(size_t)(w.i_lo + r) * p.W + (w.j_lo + cc)];
        }
    };

    int e = blockIdx.x;
    if (e >= N) return;
    __syncthreads();  // tables
    ScanWindow w = get_window(e);
    issue_tile(w, tile_base);
    // the window of the env after next is fetched one iteration early (scalar loads), so its latency hides under a ray phase
    ScanWindow wn = (e + n_wg < N) ? get_window(e + n_wg) : w;
    for (int it = 0;; ++it) {
    RV_STAMP(0);
    // with two tile buffers one barrier per env is enough: the buffer filled next was last read two envs ago
    cell_t *tile = tile_base + (two_bufs ? (size_t)(it & 1) * tile_cells : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    RV_STAMP(1);
    const int e_next = e + n_wg;
    const bool more = e_next < N;
    if (more && two_bufs) issue_tile(wn, tile_base + (size_t)((it + 1) & 1) * tile_cells);
    const ScanWindow wnn = (e_next + n_wg < N) ? get_window(e_next + n_wg) : wn;
    RV_STAMP(2);
    const float px = w.px, py = w.py, pz = w.pz, cy = w.cy, sy = w.sy;
    const int i_lo = w.i_lo, j_lo = w.j_lo, th = w.th;
    const int pitch = w.tw4 * CC;  // cells per staged row
    const int tw = min(pitch, p.W - j_lo);
    RV_STAMP(3);

    float *row = out + (size_t)e * row_stride + col0;
    // surface height from four staged cells: the plane of the cell's triangle the ray falls in (cells split along the
    // (i, j) - (i+1, j+1) diagonal: what a ray-cast of the terrain's triangle mesh returns) or the bilinear patch.  For the
    // int16 tile the interpolation runs on the raw integers and is scaled once at the end: q_scale is a power of two, so
    // this is bit-identical to interpolating the scaled heights
    auto bilerp = [&](const cell_t *q, float fx, float fy) -> float {
        const float hh = patch_height<TRI>(q, pitch, fx, fy);
        return Q16 ? hh * p.q_scale : hh;
    };
    // one vertical ray: bilinear height of the staged tile at the yaw-rotated grid point.  FAST (whole-workgroup uniform,
    // decided once per env by scan_window()): every ray is inside the map and the window, so the bounds tests, clamps and
    // the window check of the general path are no-ops and are skipped.
    auto ray_obs = [&](int ray, auto fast_tag) -> float {
        constexpr bool FAST = decltype(fast_tag)::value;
        const int i = (int)(((float)ray + 0.5f) * inv_nx);   // ray / scan_nx, exact for ray < 2^20
        const int j = ray - (int)__umul24(i, c.scan_nx);
        const float oy = oy_tab[i];
        const float ox = ox_tab[j];
        const float x = px + (cy * ox - sy * oy);
        const float y = py + (sy * ox + cy * oy);
        float hgt;
        if (FAST) {
            const float u = (x - p.min_x) * inv_res;
            const float v = (y - p.min_y) * inv_res;
            const int j0 = (int)u, i0 = (int)v;
            const float fx = u - (float)j0, fy = v - (float)i0;
            hgt = bilerp(tile + (__umul24(i0 - i_lo, pitch) + (j0 - j_lo)), fx, fy);
        } else if (x < p.min_x || x > x_max || y < p.min_y || y > y_max) {
            hgt = INFINITY;  // ray leaves the terrain: ORBIT RayCaster reports +inf
        } else {
            float u = (x - p.min_x) * inv_res;
            float v = (y - p.min_y) * inv_res;
            u = clampf(u, 0.0f, (float)(p.W - 1));
            v = clampf(v, 0.0f, (float)(p.H - 1));
            int j0 = (int)u, i0 = (int)v;
            if (j0 > p.W - 2) j0 = p.W - 2;
            if (i0 > p.H - 2) i0 = p.H - 2;
            const float fx = u - (float)j0, fy = v - (float)i0;
            const int jl = j0 - j_lo, il = i0 - i_lo;
            // The staged window covers every in-map ray by construction (scan_window(): pattern extent + slack, tile sized
            // for the diagonal).  Indices are clamped so the LDS reads are always valid plain ds_read instructions (a
            // pointer that may be LDS or global would become a slow flat load); a ray outside the window -- a bug --
            // yields NaN, which the parity tests would catch.
            const bool in_tile = jl >= 0 && il >= 0 && jl + 1 < tw && il + 1 < th;
            const int jc = max(0, min(jl, tw - 2)), ic = max(0, min(il, th - 2));
            hgt = bilerp(tile + ic * pitch + jc, fx, fy);
            if (!in_tile) return __int_as_float(0x7fc00000);
        }
        return pz - hgt - c.scan_height_offset;  // observations.py:45
    };
    // two independent rays per thread first (their LDS reads overlap), then whatever is left for larger patterns
    auto all_rays = [&](auto fast_tag) {
        const int r0 = tid, r1 = tid + RV_K2_THREADS;
        const bool v0 = r0 < p.rays, v1 = r1 < p.rays;
        const float o0 = ray_obs(v0 ? r0 : 0, fast_tag);
        const float o1 = ray_obs(v1 ? r1 : 0, fast_tag);
        if (v0) row[r0] = o0;
        if (v1) row[r1] = o1;
        for (int ray = tid + 2 * RV_K2_THREADS; ray < p.rays; ray += RV_K2_THREADS) row[ray] = ray_obs(ray, fast_tag);
    };
    if (w.interior) all_rays(std::true_type{});
    else all_rays(std::false_type{});
    RV_STAMP(4);
    if (MODE == 1 && tid == 0) {
        const float cbx = state[(size_t)(ROVER_CMD_B + 0) * N + e];
        const float cby = state[(size_t)(ROVER_CMD_B + 1) * N + e];
        float *o = out + (size_t)e * row_stride;
        o[0] = state[(size_t)(ROVER_ACTION + 0) * N + e];
        o[1] = state[(size_t)(ROVER_ACTION + 1) * N + e];
        o[2] = sqrtf(cbx * cbx + cby * cby) * c.obs_scale_distance;
        o[3] = rv_atan2f(cby, cbx) * c.obs_scale_heading;
    }
    if (!more) break;
    if (!two_bufs) {
        __syncthreads();  // every ray of this env has read the single tile
        issue_tile(wn, tile_base);
    }
    w = wn;
    wn = wnn;
    e = e_next;
    }  // env loop
    if (MODE == 2 && blockIdx.x == 0 && log_out) {   // workgroup 0, after its last env: extras["log"] of this step (usually one counter read)
        __syncthreads();                  // the tiles are dead: the reduction reuses the LDS
        reduce_log_partials<RV_K2_THREADS>(p, lds, tid, log_partial, n_waves, log_out);
    }
}

// ------------------------------------------------------------------------------------------------ K2, step form
// The scan kernel of rover_step() when the host can promise 16-byte chunk staging (map width a multiple of the chunk,
// aligned base) and a pattern of at most 1024 rays -- every procedural or imported terrain of a power-of-two width.
// Same arithmetic as rover_scan_obs_kernel<2, ...>; what differs is what the measurements of round 2 asked for:
//   * a small SCALAR footprint.  8 waves per SIMD leave 80 SGPRs per wave (800 per SIMD, 16 of them the trap handler's); the
//     generic kernel keeps three decoded windows plus every uniform of its fallback paths live (68 spill slots, ~50
//     v_readlane / v_writelane per env).  Here the windows being cast live in VGPRs (8 v_mov per env; VGPRs are plentiful),
//     the windows staged next stay as raw 8-dword descriptors, and there are no spills;
//   * each thread's ray offset is computed once, not per env (no table reads, no index split), and a thread without a ray of
//     its own repeats ray 0, so the ray phase needs no execution masks;
//   * FEW, FAT workgroups and few synchronisation rounds: 1024 threads (one ray per thread of a 31 x 31 pattern), two per CU,
//     and EPI = 2 envs per round -- two tiles are staged together, one `s_waitcnt vmcnt(0)` + barrier releases both, every
//     thread casts its ray in both.  HIP-event time at 4096 / 16384 / 32768 envs: 256 threads 26.8 / 84 us; 512 threads
//     19.4 / 59 / 112 us; 1024 threads 18.6 / 55 / 104 us; 1024 threads x 2 envs 18.0 / 51 / 97 us (generic: 20.3 / 65 / 124);
//   * ONE tile per env (no double buffering): the other workgroup of the CU already overlaps copy and cast, and a second
//     buffer measured no faster at any of these sizes.
__device__ __forceinline__ float to_vgpr(float uniform) { float v; asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(uniform)); return v; }
__device__ __forceinline__ int to_vgpr(int uniform) { int v; asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(uniform)); return v; }

template <bool Q16, bool TRI, int THREADS, int EPI>   // THREADS x RPT = 1024 rays at most; EPI envs per iteration
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(8, 8), target("no-unaligned-access-mode"))) void rover_scan_step_kernel(
    RvParams p, float *__restrict__ out, int row_stride, int col0, const float *__restrict__ log_partial, int n_waves,
    float *__restrict__ log_out, const float *__restrict__ scan_desc)
{
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x;
    const rover_config &c = p.cfg;
    const int N = p.n;
    const int n_wg = (int)gridDim.x;
    using cell_t = typename std::conditional<Q16, int16_t, float>::type;
    constexpr int CC = Q16 ? 8 : 4;  // cells per 16-byte chunk
    typedef float v4f __attribute__((ext_vector_type(4)));
    float *inv_tab = lds + 128;  // same LDS carve as the generic kernel; the ray-offset tables at [0, 128) are not needed
    cell_t *tile_base = reinterpret_cast<cell_t *>(lds + 192);
    const int tile_cells = p.tile_dim * p.tile_pitch;
    const cell_t *hsrc = Q16 ? reinterpret_cast<const cell_t *>(p.height_q) : reinterpret_cast<const cell_t *>(p.height);
    // ORBIT grid_pattern: arange(-size/2, size/2 + 1e-9, res) evaluated in double, x fastest (App. C)
    auto pattern_x = [&](int j) { return (float)(-0.5 * (double)c.scan_size_x + (double)c.scan_resolution * (double)j); };
    auto pattern_y = [&](int i) { return (float)(-0.5 * (double)c.scan_size_y + (double)c.scan_resolution * (double)i); };
    if (tid >= 128 && tid < 192) inv_tab[tid - 128] = 1.0f / (float)max(tid - 128, 1);
    const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
    // this thread's RPT rays tid + m THREADS (the host sends patterns of more than 1024 rays to the generic kernel).  A thread
    // without an m-th ray of its own repeats its first ray (or ray 0): the duplicate stores write the same bits, and the ray
    // phase needs no execution masks
    constexpr int RPT = 1024 / THREADS;
    int ray[RPT];
    float ox[RPT], oy[RPT];
#pragma unroll
    for (int m = 0; m < RPT; ++m) {
        int r = tid + m * THREADS;
        ray[m] = r < p.rays ? r : (m == 0 ? 0 : ray[0]);
        ox[m] = pattern_x(ray[m] % c.scan_nx);
        oy[m] = pattern_y(ray[m] / c.scan_nx);
    }

    // envs of iteration `it`: group (blockIdx.x + it n_wg), envs group * EPI + j, j < EPI; indices past the end repeat env N - 1
    // (same bits again).  XCD-aware dealing (EPI = 2, p.xcd_rows): workgroup j runs on XCD j mod 8 (round-robin dispatch), and
    // inside every aligned block of 64 groups the group index is transposed, g = 8 y + x -> 8 x + y, so that the eight pairs
    // 8 b .. 8 b + 7 of a 16-row block all come from XCD b mod 8.  A bijection of the group indices: every env is still cast once.
    const int n_groups = (N + EPI - 1) / EPI;
    auto group_of = [&](int g) {
        if (EPI == 2 && p.xcd_rows && (g | 63) < n_groups) g = (g & ~63) | ((g & 7) << 3) | ((g >> 3) & 7);
        return g;
    };
    int gidx = blockIdx.x;
    if (gidx >= n_groups) return;
    int e0 = group_of(gidx) * EPI;
    __syncthreads();  // table

    // asynchronous dense copy of a th x tw4 chunk window into LDS (see rover_scan_obs_kernel)
    auto issue_tile = [&](const float4 &d1, cell_t *tile) {
        const int i_lo = __float_as_int(d1.y), j_lo = __float_as_int(d1.z), pk = __float_as_int(d1.w);
        const int th = pk & 0x7FFF, tw4 = pk >> 16;
        const v4f *src = reinterpret_cast<const v4f *>(hsrc + (size_t)i_lo * p.W + j_lo);
        v4f *dst = reinterpret_cast<v4f *>(tile);
        const int nchunk = th * tw4;
        const float inv_tw4 = inv_tab[tw4];
        for (int k0 = 0; k0 < nchunk; k0 += THREADS) {
            const int k = k0 + tid;
            if (k < nchunk) {
                const int r = (int)(((float)k + 0.5f) * inv_tw4);  // k / tw4, exact for k < 2^20
                const int cq = k - (int)__umul24(r, tw4);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(src + (size_t)(__umul24(r, p.wq) + cq)),
                    (__attribute__((address_space(3))) void *)(dst + (k0 + wave_base)), 16, 0, 0);
            }
        }
    };
    auto load_desc = [&](int env, float4 &d0, float4 &d1) {
        const float4 *d = reinterpret_cast<const float4 *>(scan_desc + (size_t)min(env, N - 1) * 8);
        d0 = d[0];
        d1 = d[1];
    };

    float4 a0[EPI], a1[EPI];  // descriptors of the envs whose tiles are staged next
    // the envs being cast: pose and window origin in VGPRs, the packed sizes as one scalar each
    float px[EPI], py[EPI], pz[EPI], cy[EPI], sy[EPI];
    int i_lo[EPI], j_lo[EPI], pk[EPI];
    auto to_cast = [&]() {
#pragma unroll
        for (int j = 0; j < EPI; ++j) {
            px[j] = to_vgpr(a0[j].x); py[j] = to_vgpr(a0[j].y); pz[j] = to_vgpr(a0[j].z); cy[j] = to_vgpr(a0[j].w);
            sy[j] = to_vgpr(a1[j].x);
            i_lo[j] = to_vgpr(__float_as_int(a1[j].y)); j_lo[j] = to_vgpr(__float_as_int(a1[j].z));
            pk[j] = __float_as_int(a1[j].w);
        }
    };
#pragma unroll
    for (int j = 0; j < EPI; ++j) load_desc(e0 + j, a0[j], a1[j]);
#pragma unroll
    for (int j = 0; j < EPI; ++j) issue_tile(a1[j], tile_base + j * tile_cells);
    to_cast();
    for (;;) {
        const int g_next = gidx + n_wg;
        const bool more = g_next < n_groups;
        const int e_next = more ? group_of(g_next) * EPI : N;
        if (more) {   // scalar loads: they arrive under the ray phase
#pragma unroll
            for (int j = 0; j < EPI; ++j) load_desc(e_next + j, a0[j], a1[j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPI; ++j) {
            const cell_t *tile = tile_base + j * tile_cells;
            const int th = pk[j] & 0x7FFF;
            const int pitch = (pk[j] >> 16) * CC;  // cells per staged row
            float *row = out + (size_t)min(e0 + j, N - 1) * row_stride + col0;
            auto ray_obs = [&](float rx, float ry, auto fast_tag) -> float {
                constexpr bool FAST = decltype(fast_tag)::value;
                const float x = px[j] + (cy[j] * rx - sy[j] * ry);
                const float y = py[j] + (sy[j] * rx + cy[j] * ry);
                float hgt;
                if (FAST) {
                    const float u = (x - p.min_x) * p.inv_res;
                    const float v = (y - p.min_y) * p.inv_res;
                    const int j0 = (int)u, i0 = (int)v;
                    const float fx = u - (float)j0, fy = v - (float)i0;
                    hgt = patch_height<TRI>(tile + (__umul24(i0 - i_lo[j], pitch) + (j0 - j_lo[j])), pitch, fx, fy);
                } else if (x < p.min_x || x > p.x_max || y < p.min_y || y > p.y_max) {
                    return pz[j] - INFINITY - c.scan_height_offset;  // ray leaves the terrain: ORBIT RayCaster reports +inf
                } else {
                    float u = (x - p.min_x) * p.inv_res;
                    float v = (y - p.min_y) * p.inv_res;
                    u = clampf(u, 0.0f, (float)(p.W - 1));
                    v = clampf(v, 0.0f, (float)(p.H - 1));
                    int j0 = (int)u, i0 = (int)v;
                    if (j0 > p.W - 2) j0 = p.W - 2;
                    if (i0 > p.H - 2) i0 = p.H - 2;
                    const float fx = u - (float)j0, fy = v - (float)i0;
                    const int jl = j0 - j_lo[j], il = i0 - i_lo[j];
                    const int tw = min(pitch, p.W - j_lo[j]);
                    const bool in_tile = jl >= 0 && il >= 0 && jl + 1 < tw && il + 1 < th;
                    const int jc = max(0, min(jl, tw - 2)), ic = max(0, min(il, th - 2));
                    hgt = patch_height<TRI>(tile + ic * pitch + jc, pitch, fx, fy);
                    if (!in_tile) return __int_as_float(0x7fc00000);  // a ray outside the staged window is a bug: NaN
                }
                if (Q16) hgt *= p.q_scale;
                return pz[j] - hgt - c.scan_height_offset;  // observations.py:45
            };
            auto all_rays = [&](auto fast_tag) {
                float o[RPT];
#pragma unroll
                for (int m = 0; m < RPT; ++m) o[m] = ray_obs(ox[m], oy[m], fast_tag);
#pragma unroll
                for (int m = 0; m < RPT; ++m) row[ray[m]] = o[m];
            };
            if ((pk[j] >> 15) & 1) all_rays(std::true_type{});
            else all_rays(std::false_type{});
        }
        if (!more) break;
        __syncthreads();  // every ray of this iteration has read its tile
#pragma unroll
        for (int j = 0; j < EPI; ++j) issue_tile(a1[j], tile_base + j * tile_cells);
        to_cast();
        e0 = e_next;
        gidx = g_next;
    }
    if (blockIdx.x == 0 && log_out) {   // workgroup 0, after its last pair of envs: extras["log"] of this step (usually one counter read)
        __syncthreads();     // the tiles are dead: the reduction reuses the LDS
        reduce_log_partials<THREADS>(p, lds, tid, log_partial, n_waves, log_out);
    }
}

// The same without copy waves (scan_single_tile_wave): 256-thread workgroups, two per CU.
template <bool TRI>
__global__ __launch_bounds__(RV_K1G_THREADS) RV_FUSED_ATTR void rover_step_scan1_kernel(
    RvParams p, float *__restrict__ state, const float *__restrict__ action, float *__restrict__ obs, float *__restrict__ reward,
    uint8_t *__restrict__ terminated, uint8_t *__restrict__ truncated, float *__restrict__ force, float *__restrict__ log_partial,
    const float2 *__restrict__ ray_xy)
{
    extern __shared__ __align__(16) float lds[];
    step_group_body<TRI ? 4 : 3>(p, state, action, obs, reward, terminated, truncated, force, log_partial, lds, ray_xy);
}
// extras["log"] behind the fused step kernel (the scan kernel's workgroup 0 does this on the two-launch path)
// (1024 threads: the summation order of the scan kernel's reduction, so that both paths produce the same bits)
__global__ __launch_bounds__(1024) void rover_log_kernel(RvParams p, const float *__restrict__ log_partial, int n_waves,
                                                         float *__restrict__ log_out)
{
    __shared__ float lds[1024 + 16];
    reduce_log_partials<1024>(p, lds, threadIdx.x, log_partial, n_waves, log_out);
}
template <bool TRI>
__global__ __launch_bounds__(RV_K1G_THREADS) RV_FUSED_ATTR void rover_scan_private_kernel(
    RvParams p, float *__restrict__ out, int row_stride, int col0, const float *__restrict__ log_partial, int n_waves,
    float *__restrict__ log_out, const float *__restrict__ scan_desc, const float2 *__restrict__ ray_xy)
{
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e_base = (blockIdx.x * (RV_K1G_THREADS / 64) + wv) * 4;
    const int tile_cells = p.tile_dim * p.tile_pitch;
    int16_t *tile0 = reinterpret_cast<int16_t *>(lds) + (size_t)(2 * wv) * tile_cells, *tile1 = tile0 + tile_cells;
    const int n_env = max(0, min(4, p.n - e_base));
    PrivateWindows w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 *d = reinterpret_cast<const float4 *>(scan_desc + (size_t)min(e_base + j, p.n - 1) * 8);
        const float4 d0 = d[0], d1 = d[1];
        w.px[j] = d0.x; w.py[j] = d0.y; w.pz[j] = d0.z; w.cy[j] = d0.w; w.sy[j] = d1.x;
        w.i_lo[j] = __float_as_int(d1.y); w.j_lo[j] = __float_as_int(d1.z); w.pk[j] = __float_as_int(d1.w);
    }
    scan_private_wave<TRI>(p, tile0, tile1, lane, n_env, e_base, w, out, row_stride, col0, ray_xy);
    if (blockIdx.x == 0 && log_out) {   // workgroup 0: extras["log"] of this step (usually one counter read)
        __syncthreads();     // the tiles are dead: the reduction reuses the LDS
        reduce_log_partials<RV_K1G_THREADS>(p, lds, tid, log_partial, n_waves, log_out);
    }
}

// ================================================================================================ unit kernels
__global__ void rover_ackermann_kernel(rover_config c, int n, const float *raw, float *processed, float *steer, float *wheel)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r[2] = {raw[2 * i], raw[2 * i + 1]}, pr[2], st[4], wh[6];
    ackermann_one(c, r, pr, st, wh);
    processed[2 * i] = pr[0]; processed[2 * i + 1] = pr[1];
    for (int k = 0; k < 4; ++k) steer[4 * i + k] = st[k];
    for (int k = 0; k < 6; ++k) wheel[6 * i + k] = wh[k];
}

// the observation-head / reward / termination term functions of the step kernel's tail, one row per thread
__global__ void rover_mdp_terms_kernel(rover_config c, int n, const float *cmd_b, const float *action, const float *prev_action,
                                       const int32_t *ep_len, const float *force, float *obs_distance, float *obs_angle,
                                       float *rew, uint8_t *term)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float F[ROVER_NUM_BODIES * 3];
    for (int k = 0; k < ROVER_NUM_BODIES * 3; ++k) F[k] = force[(size_t)i * (ROVER_NUM_BODIES * 3) + k];
    const float cb[3] = {cmd_b[3 * i], cmd_b[3 * i + 1], cmd_b[3 * i + 2]};
    const float a[2] = {action[2 * i], action[2 * i + 1]}, pa[2] = {prev_action[2 * i], prev_action[2 * i + 1]};
    float r[ROVER_NUM_REW];
    bool t[ROVER_NUM_TERM];
    mdp_terms_one(c, cb, a, pa, ep_len[i], F, r, t);
    for (int k = 0; k < ROVER_NUM_REW; ++k) rew[(size_t)i * ROVER_NUM_REW + k] = r[k];
    for (int k = 0; k < ROVER_NUM_TERM; ++k) term[(size_t)i * ROVER_NUM_TERM + k] = t[k] ? 1 : 0;
    // the two observation terms as write_obs_head evaluates them, before their scales (observations.py:15-32)
    obs_distance[i] = sqrtf(cb[0] * cb[0] + cb[1] * cb[1]);
    obs_angle[i] = rv_atan2f(cb[1], cb[0]);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void rover_physics_kernel(RvParams p, float *__restrict__ state, const float *steer_t,
                                                           const float *wheel_t, int substeps, float *force)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= p.n) return;
    const int N = p.n;
    float S[ROVER_STATE_WORDS];
#pragma unroll
    for (int i = 0; i < ROVER_STATE_WORDS; ++i) S[i] = state[(size_t)i * N + e];
    float st[4], wt[6], F[ROVER_NUM_BODIES * 3];
#pragma unroll
    for (int k = 0; k < 4; ++k) st[k] = steer_t[4 * e + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) wt[k] = wheel_t[6 * e + k];
#pragma unroll
    for (int i = 0; i < ROVER_NUM_BODIES * 3; ++i) F[i] = 0.0f;
    const StepConsts &K = p.K;
    for (int s = 0; s < substeps - 1; ++s) physics_substep<false>(p, K, S, st, wt, nullptr);
    if (substeps > 0) physics_substep<true>(p, K, S, st, wt, F);
#pragma unroll
    for (int i = 0; i < ROVER_STATE_WORDS; ++i) state[(size_t)i * N + e] = S[i];
    if (force) {
#pragma unroll
        for (int i = 0; i < ROVER_NUM_BODIES * 3; ++i) force[(size_t)i * N + e] = F[i];
    }
}

thread_local char g_err[512] = "";
int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}
#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t _e = (expr);                                                                                        \
        if (_e != hipSuccess) return fail(ROVER_ERR_HIP, #expr ": %s", hipGetErrorString(_e));                        \
    } while (0)

// Every entry point that launches work runs on the handle's device, whatever the calling thread's current device is.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

}  // namespace

int rover_internal_fail(int code, const char *fmt, const char *detail) { return fail(code, fmt, detail); }

// ==================================================================================================== C ABI
struct rover_sim {
    RvParams p;
    int device;
    bool have_terrain;
    float *state;
    float *log_partial;
    size_t ws_log_floats;
    size_t ws_bytes;
    int n_waves;       // log-partial rows written by the step kernel of the selected mapping
    int step_blocks;   // workgroups of the step kernel
    uint64_t counter;  // rover_reset / rover_step launches so far (keys the per-batch spawn permutation)
    bool group_mapping; // sixteen lanes per env
    size_t lds_bytes;
    int n_cu;          // compute units of the device
    int scan_wgs;      // persistent scan workgroups: what the device holds at once
    int scan_form;     // measurement hook: 1 = the generic scan kernel on the step path too
    bool markers;      // roctx ranges around the launches of rover_step (rover_set_markers)
    uint32_t log_serial; // tag of the log-partial rows of the launch under way
    size_t max_lds;      // LDS a workgroup may allocate on this device (hipDeviceAttributeMaxSharedMemoryPerBlock)
    int fused;           // one launch per step (rover_step_scan_kernel): -1 = decide (group mapping, int16 terrain copy, <= 1024 rays,
                         // the four waves' eight tiles fit the LDS, one workgroup per CU holds the batch), 0 = off, 1 = on where possible
    bool single_tile_ok; // the single-tile one-launch form may be chosen automatically beyond one round of workgroups (measured: see fused_form)
    bool log_deferred;   // rover_set_log_deferred: rover_step leaves `log` alone, rover_flush_log reduces it on demand
    bool phase_open;     // rover_step_begin has run, rover_step_finish has not
    int launch_error;    // set by launch_step_kernels when a launch could not be made (rover_step returns it)
    float2 *ray_xy;      // [1024] pattern offsets of ray i (rays past the pattern repeat ray 0): the wave-private scan's table (workspace)
};



static void configure_tile(rover_sim *sim, int chunk_cells);

// spawn_draw = 1: (a, b) of this launch's affine row bijection from (seed, call counter); then the counter advances.
// Same arithmetic as spawn_affine() of the oracle.
static uint32_t gcd_u32(uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; }
static void next_batch(rover_sim *sim)
{
    RvParams &p = sim->p;
    const uint32_t n = (uint32_t)p.n_spawns;
    uint32_t r[4];
    philox4x32((uint32_t)(sim->counter & 0xFFFFFFFFu), (uint32_t)(sim->counter >> 32), 0x5eedu, 2u, p.cfg.seed_lo, p.cfg.seed_hi, r);
    uint32_t a = n > 1 ? r[0] % n : 0u;
    while (n > 1 && gcd_u32(a, n) != 1u) a = (a + 1u) % n;
    p.spawn_a = a;
    p.spawn_b = n > 0 ? r[1] % n : 0u;
    p.step_tag = ++sim->log_serial;   // launches of THIS handle (not the restorable call counter: a resumed env must not
                                      // meet rows an earlier pass through the same counter values left behind); starts at 1
    ++sim->counter;
}

// Which scan kernel a launch of mode `mode` (0 unit scan, 1 reset, 2 step) uses: the spill-free step form (`simple`) with
// one or two envs per synchronisation round, or the generic kernel.  Shared by launch_scan and rover_kernel_names.
struct ScanForm {
    bool q16, tri, simple;
    int epi, grid;
    size_t step_lds;
};
static ScanForm scan_form_of(const rover_sim *sim, int mode)
{
    ScanForm f;
    f.grid = sim->p.n < sim->scan_wgs ? sim->p.n : sim->scan_wgs;
    f.q16 = sim->p.height_q != nullptr;
    f.tri = sim->p.cfg.scan_surface == 0;
    const int cc = f.q16 ? 8 : 4;
    const uintptr_t base = f.q16 ? reinterpret_cast<uintptr_t>(sim->p.height_q) : reinterpret_cast<uintptr_t>(sim->p.height);
    f.simple = mode == 2 && (sim->p.W & (cc - 1)) == 0 && (base & 15) == 0 && sim->p.rays <= 1024 && sim->scan_form != 1;
    const size_t tile_bytes = (size_t)sim->p.tile_dim * sim->p.tile_pitch * (f.q16 ? 2 : 4);
    // two envs per iteration when two workgroups with two tiles each fit the CU's LDS (measurement hook: form 2 = one env)
    f.epi = (sim->scan_form != 2 && 2 * (192 * sizeof(float) + 2 * tile_bytes) <= sim->max_lds) ? 2 : 1;
    f.step_lds = 192 * sizeof(float) + f.epi * tile_bytes < (1024 + 16) * sizeof(float) ? (1024 + 16) * sizeof(float)
                                                                                         : 192 * sizeof(float) + f.epi * tile_bytes;
    if (f.simple) {
        const int groups = (sim->p.n + f.epi - 1) / f.epi, wgs = 2 * sim->n_cu;   // two 1024-thread workgroups per CU (one per CU: 19.7 vs 17.8 us)
        f.grid = groups < wgs ? groups : wgs;
    }
    return f;
}

// Does rover_step run as ONE launch (rover_step_scan_kernel: the scan is the last phase of the step kernel's waves)?
static size_t single_tile_lds_bytes(const rover_sim *sim) { return (size_t)(RV_K1G_THREADS / 64) * (size_t)sim->p.tile_dim * sim->p.tile_pitch * 2; }
// eight tiles (two per step wave) + the windows' hand-over area (4 waves x 2 sets x 4 envs x 32 B) + the link points' (4 x 768 B)
static size_t fused_lds_bytes(const rover_sim *sim) { return (size_t)(RV_K1G_THREADS / 64) * 2 * (size_t)sim->p.tile_dim * sim->p.tile_pitch * 2 + 1024 + 4 * RV_HAND * sizeof(float); }
// 0 = two launches, 1 = one launch with copy waves (one workgroup per CU), 2 = one launch, one tile per wave (two workgroups per CU)
static int fused_form(const rover_sim *sim)
{
    if (sim->fused == 0 || !sim->group_mapping || sim->scan_form != 0) return 0;
    const ScanForm f = scan_form_of(sim, 2);
    if (!f.simple || !f.q16 || sim->p.rays > 1024) return 0;
    if (sim->p.tile_pitch / 8 > 64) return 0;   // private_issue stages whole rows per instruction: a row must fit a wave's 64 lanes
    // (the device's own limit: 160 KiB per workgroup on gfx950)
    const bool fits1 = fused_lds_bytes(sim) <= sim->max_lds, fits2 = 2 * single_tile_lds_bytes(sim) <= sim->max_lds;
    if (sim->fused == 1) return fits1 ? 1 : 0;        // measurement hooks: force a form wherever its tiles fit
    if (sim->fused == 2) return fits2 ? 2 : 0;
    // (without the on-demand log reduction rover_log_kernel follows the one launch: still shorter than step + scan kernel with
    // the reduction in its last workgroup -- 4096 envs: 33.9 + ~5 us against 45.3 us)
    // The copy-wave form holds ONE workgroup per CU: it is the one to use while one round of workgroups holds the batch -- and
    // not far below that either: the scan phase of a wave is four envs long whatever the batch, while the scan KERNEL shrinks
    // with it (N sweep: 1024 envs 36.7 vs 34.6 us, 4096 envs 42.0 vs 47.1 us per step; break-even near 2048 envs on 256 CUs).
    if (fits1 && sim->step_blocks <= sim->n_cu) return 2 * sim->step_blocks >= sim->n_cu ? 1 : 0;
    // More than one round of workgroups: the single-tile form keeps the two-launch path's two workgroups per CU (N sweep, us per
    // step, one launch / two: 8192 envs 62.7 / 70.4, 16384 envs 112.0 / 125.6, 32768 envs 197.6 / 228.6).
    return (sim->single_tile_ok && fits2) ? 2 : 0;
}
static bool fused_step(const rover_sim *sim) { return fused_form(sim) != 0; }
// The automatic mapping (cfg.step_mapping = 0) is sixteen lanes per env at every batch size: as one launch wherever the terrain /
// pattern allow it (fused_form), else as the two launches of the group mapping.  One env per lane (cfg.step_mapping = 1) is faster
// than those two launches from ~65536 envs on (tools/n_sweep.py: 164 against 131 M env-steps/s at 65536 envs; slower at 32768:
// 116 against 142 M) but its kernels carry 728 - 984 bytes of scratch per lane: an explicit choice, never the automatic one.
// Both mappings produce the same bits from the same state layout.
static void refresh_mapping(rover_sim *) {}
// the kernel launches of one env step (rover_step / rover_profile_step); ev: optional event recorded between the two launches
static void launch_step_kernels(rover_sim *sim, hipStream_t st, const float *action, float *obs, float *reward, uint8_t *terminated,
                                uint8_t *truncated, float *force, float *log, hipEvent_t mid);

template <int MODE>
static void launch_scan(rover_sim *sim, int grid, hipStream_t st, float *out, int row_stride, int col0, const float *log_partial,
                        int n_waves, float *log_out)
{
    const ScanForm f = scan_form_of(sim, MODE);
    grid = f.grid;
    const bool q16 = f.q16, tri = f.tri, simple = f.simple;
    const int epi = f.epi;
    const size_t step_lds = f.step_lds;
    if (MODE == 2 && sim->scan_form == 7 && simple && q16) {   // measurement hook: the wave-private form as a kernel of its own
        const size_t tile_bytes = (size_t)sim->p.tile_dim * sim->p.tile_pitch * 2;
        size_t lds = (RV_K1G_THREADS / 64) * 2 * tile_bytes;
        if (lds < (RV_K1G_THREADS + 16) * sizeof(float)) lds = (RV_K1G_THREADS + 16) * sizeof(float);
        if (lds <= sim->max_lds) {
            const int blocks = (sim->p.n + RV_K1G_ENVS - 1) / RV_K1G_ENVS;
            if (tri) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&rover_scan_private_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((rover_scan_private_kernel<true>), dim3(blocks), dim3(RV_K1G_THREADS), lds, st, sim->p, out, row_stride, col0,
                                   log_partial, n_waves, log_out, sim->p.scan_desc, sim->ray_xy);
            } else {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&rover_scan_private_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((rover_scan_private_kernel<false>), dim3(blocks), dim3(RV_K1G_THREADS), lds, st, sim->p, out, row_stride, col0,
                                   log_partial, n_waves, log_out, sim->p.scan_desc, sim->ray_xy);
            }
            return;
        }
    }
#define RV_LAUNCH_STEP(Q, T, E)                                                                                               \
    hipLaunchKernelGGL((rover_scan_step_kernel<Q, T, 1024, E>), dim3(grid), dim3(1024), step_lds, st, sim->p, out, row_stride, \
                       col0, log_partial, n_waves, log_out, sim->p.scan_desc)
#define RV_LAUNCH_SCAN_QT(Q, T)                                                                                               \
    do {                                                                                                                      \
        if (simple && epi == 2) RV_LAUNCH_STEP(Q, T, 2);                                                                      \
        else if (simple) RV_LAUNCH_STEP(Q, T, 1);                                                                             \
        else                                                                                                                  \
            hipLaunchKernelGGL((rover_scan_obs_kernel<MODE, Q, T>), dim3(grid), dim3(RV_K2_THREADS), sim->lds_bytes, st,       \
                               sim->p, sim->state, out, row_stride, col0, log_partial, n_waves, log_out, sim->p.scan_desc);   \
    } while (0)
    if (q16 && tri) RV_LAUNCH_SCAN_QT(true, true);
    else if (q16) RV_LAUNCH_SCAN_QT(true, false);
    else if (tri) RV_LAUNCH_SCAN_QT(false, true);
    else RV_LAUNCH_SCAN_QT(false, false);
#undef RV_LAUNCH_SCAN_QT
#undef RV_LAUNCH_STEP
}

// ---- roctx ranges around the two launches of a step (tracing): resolved lazily with dlopen so that the
// library carries no link-time dependency on a profiler; `rocprofv3 --marker-trace` then shows one "rover_step" range per
// env step with the "K1 ..." / "K2 ..." ranges inside.  Off unless rover_set_markers(sim, 1) was called.
typedef int (*roctx_push_fn)(const char *);
typedef int (*roctx_pop_fn)(void);
static roctx_push_fn g_roctx_push = nullptr;
static roctx_pop_fn g_roctx_pop = nullptr;
static int g_roctx_state = 0;   // 0 = not tried, 1 = resolved, -1 = unavailable
static bool roctx_resolve()
{
    if (g_roctx_state == 0) {
        g_roctx_state = -1;
        const char *names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
        for (const char *nm : names) {
            void *h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            g_roctx_push = reinterpret_cast<roctx_push_fn>(dlsym(h, "roctxRangePushA"));
            g_roctx_pop = reinterpret_cast<roctx_pop_fn>(dlsym(h, "roctxRangePop"));
            if (g_roctx_push && g_roctx_pop) { g_roctx_state = 1; break; }
        }
    }
    return g_roctx_state == 1;
}
struct MarkerRange {
    bool on;
    MarkerRange(const rover_sim *sim, const char *name) : on(sim->markers && roctx_resolve()) { if (on) g_roctx_push(name); }
    ~MarkerRange() { if (on) g_roctx_pop(); }
};

// The dynamic-LDS limit is a property of the kernel (per device), shared by every handle of the process: only ever RAISE it -- a
// handle with smaller tiles must not lower it under another handle's launches -- and not per launch (the call costs tens of
// microseconds of host time).  What has been granted is remembered per device under a mutex (handles may step from different host
// threads) and only when the runtime said yes; a failure is reported through rover_last_error() and retried by the next call.
#include <mutex>
static bool raise_dynamic_lds(rover_sim *sim, int form, size_t lds)
{
    static std::mutex mu;
    static std::vector<size_t> granted[3];   // [form][device]
    std::lock_guard<std::mutex> lock(mu);
    std::vector<size_t> &g = granted[form];
    if ((size_t)sim->device >= g.size()) g.resize((size_t)sim->device + 1, 0);
    if (lds <= g[sim->device]) return true;
    hipError_t e1, e2;
    if (form == 1) {
        e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&rover_step_scan_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&rover_step_scan_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    } else {
        e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(&rover_step_scan1_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&rover_step_scan1_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (e1 != hipSuccess || e2 != hipSuccess) {
        sim->launch_error = fail(ROVER_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
        return false;
    }
    g[sim->device] = lds;
    return true;
}
static void launch_step_kernels(rover_sim *sim, hipStream_t st, const float *action, float *obs, float *reward, uint8_t *terminated,
                                uint8_t *truncated, float *force, float *log, hipEvent_t mid)
{
    refresh_mapping(sim);
    const RvParams &p = sim->p;
    const int form = fused_form(sim);
    if (form == 2) {
        MarkerRange k1(sim, "rover_step_scan1_kernel");
        const size_t lds = single_tile_lds_bytes(sim);
        if (!raise_dynamic_lds(sim, 2, lds)) return;
        if (p.cfg.scan_surface == 0)
            hipLaunchKernelGGL((rover_step_scan1_kernel<true>), dim3(sim->step_blocks), dim3(RV_K1G_THREADS), lds, st, p, sim->state, action, obs,
                               reward, terminated, truncated, force, sim->log_partial, sim->ray_xy);
        else
            hipLaunchKernelGGL((rover_step_scan1_kernel<false>), dim3(sim->step_blocks), dim3(RV_K1G_THREADS), lds, st, p, sim->state, action, obs,
                               reward, terminated, truncated, force, sim->log_partial, sim->ray_xy);
        if (mid) (void)hipEventRecord(mid, st);
        if (!sim->log_deferred)
            hipLaunchKernelGGL(rover_log_kernel, dim3(1), dim3(1024), 0, st, p, sim->log_partial, sim->n_waves, log);
        return;
    }
    if (form == 1) {
        MarkerRange k1(sim, "rover_step_scan_kernel");
        const size_t lds = fused_lds_bytes(sim);
        if (!raise_dynamic_lds(sim, 1, lds)) return;
        if (p.cfg.scan_surface == 0) {
            hipLaunchKernelGGL((rover_step_scan_kernel<true>), dim3(sim->step_blocks), dim3(2 * RV_K1G_THREADS), lds, st, p, sim->state, action, obs,
                               reward, terminated, truncated, force, sim->log_partial, sim->ray_xy);
        } else {
            hipLaunchKernelGGL((rover_step_scan_kernel<false>), dim3(sim->step_blocks), dim3(2 * RV_K1G_THREADS), lds, st, p, sim->state, action, obs,
                               reward, terminated, truncated, force, sim->log_partial, sim->ray_xy);
        }
        if (mid) (void)hipEventRecord(mid, st);
        if (!sim->log_deferred)
            hipLaunchKernelGGL(rover_log_kernel, dim3(1), dim3(1024), 0, st, p, sim->log_partial, sim->n_waves, log);
        return;
    }
    {
        MarkerRange k1(sim, sim->group_mapping ? "K1 rover_step_kernel_group" : "K1 rover_step_kernel");
        if (sim->group_mapping)
            hipLaunchKernelGGL(rover_step_kernel_group, dim3(sim->step_blocks), dim3(RV_K1G_THREADS), 0, st, p, sim->state, action, obs, reward,
                               terminated, truncated, force, sim->log_partial);
        else
            hipLaunchKernelGGL(rover_step_kernel, dim3(sim->step_blocks), dim3(64), 0, st, p, sim->state, action, obs, reward,
                               terminated, truncated, force, sim->log_partial);
    }
    if (mid) (void)hipEventRecord(mid, st);
    {
        MarkerRange k2(sim, "K2 scan + observation rows");
        launch_scan<2>(sim, p.n + 1, st, obs, p.obs_w, 4, sim->log_partial, sim->n_waves, sim->log_deferred ? nullptr : log);
    }
}

extern "C" {

int rover_default_config(rover_config *c)
{
    if (!c) return fail(ROVER_ERR_INVALID, "cfg is NULL");
    memset(c, 0, sizeof(*c));
    c->scale_lin = 1.0f; c->scale_ang = 1.0f;              // actions_cfg.py:20
    c->offset_lin = -0.0135f; c->offset_ang = -0.0135f;    // aau_rover/env_cfg.py:30 (broadcast to both, B-4)
    c->wheel_radius = 0.1f; c->d_fr = 0.77f; c->d_mw = 0.894f; c->wheelbase = 0.849f;  // env_cfg.py:23-26
    c->sim_dt = 1.0f / 30.0f; c->decimation = 6;           // rover_env_cfg.py:269-270
    c->max_episode_length = 750; c->max_episode_length_s = 150.0f;  // :271
    c->success_threshold = 0.18f; c->far_threshold = 11.0f;          // :136,162,173,177
    c->target_distance = 9.0f;                             // terrain_importer.py:132
    c->heading_lo = -RV_PI_F; c->heading_hi = RV_PI_F; c->resample_time = 150.0f;  // rover_env_cfg.py:195-198
    const float w[ROVER_NUM_REW] = {5.0f, 5.0f, -0.1f, -1.5f, -0.5f, -2.0f, -2.0f};  // :126-163
    memcpy(c->rew_weight, w, sizeof(w));
    c->obs_scale_distance = 0.11f;                         // :104
    c->obs_scale_heading = (float)(1.0 / 3.141592653589793);  // :110
    c->scan_resolution = 0.1f; c->scan_size_x = 3.0f; c->scan_size_y = 3.0f;  // :82
    c->scan_nx = 31; c->scan_ny = 31;
    c->scan_height_offset = 0.26878f;                      // observations.py:45
    c->reset_z_offset = 0.5f;                              // randomizations.py:12
    c->reset_mode = 0;
    c->seed_lo = 0u; c->seed_hi = 0u;
    c->friction_mu = 0.75f;
    c->solver_iterations = 32;                             // aau_rover_simple.py:33 solver_position_iteration_count
    c->step_mapping = 0;
    c->max_target_tries = 32;
    c->scan_surface = 0;
    c->spawn_draw = 1;
    c->counter_lo = 0u; c->counter_hi = 0u;
    c->mass_model = 1;
    c->rew_success_threshold = 0.18f; c->rew_far_threshold = 11.0f;   // rover_env_cfg.py:136,162
    return ROVER_OK;
}

int rover_create(const rover_config *cfg, int32_t num_envs, int32_t env_id_offset, int32_t device, rover_sim **out)
{
    if (!cfg || !out) return fail(ROVER_ERR_INVALID, "cfg/out is NULL");
    if (num_envs <= 0 || env_id_offset < 0) return fail(ROVER_ERR_INVALID, "num_envs must be > 0 and env_id_offset >= 0");
    if ((uint64_t)ROVER_STATE_WORDS * (uint64_t)num_envs * 4u >= (1ull << 32))   // soa_word(): 32-bit byte offsets into the SoA arrays
        return fail(ROVER_ERR_UNSUPPORTED, "num_envs per handle must stay below 2^32 / (72 x 4) = 14.9 M (shard the batch over handles)");
    if (cfg->scan_nx > 64 || cfg->scan_ny > 64) return fail(ROVER_ERR_UNSUPPORTED, "scan grid larger than 64 x 64 rays");
    if (cfg->scan_nx <= 0 || cfg->scan_ny <= 0 || cfg->scan_resolution <= 0.0f || cfg->decimation < 0 ||
        cfg->solver_iterations < 0 || cfg->max_target_tries < 1 || cfg->sim_dt <= 0.0f || cfg->max_episode_length <= 0 ||
        cfg->scan_surface < 0 || cfg->scan_surface > 1 || cfg->spawn_draw < 0 || cfg->spawn_draw > 1 || cfg->mass_model < 0 ||
        cfg->mass_model > 1)
        return fail(ROVER_ERR_INVALID, "invalid rover_config");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(ROVER_ERR_INVALID, "device ordinal out of range");
    rover_sim *s = new (std::nothrow) rover_sim();
    if (!s) return fail(ROVER_ERR_INVALID, "out of host memory");
    memset(s, 0, sizeof(*s));
    s->p.cfg = *cfg;
    s->counter = ((uint64_t)cfg->counter_hi << 32) | cfg->counter_lo;
    make_step_consts(cfg->sim_dt, s->p.K);
    s->p.n = num_envs;
    s->p.env_id_offset = env_id_offset;
    s->p.rays = cfg->scan_nx * cfg->scan_ny;
    s->p.obs_w = 4 + s->p.rays;
    s->p.xcd_rows = 1;
    s->device = device;
    {
        int n_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n_cu <= 0) n_cu = 256;
        s->n_cu = n_cu;
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || lds <= 0) lds = 64 * 1024;
        s->max_lds = (size_t)lds;
    }
    if (cfg->step_mapping < 0 || cfg->step_mapping > 2) { delete s; return fail(ROVER_ERR_INVALID, "step_mapping must be 0, 1 or 2"); }
    // latency mapping (16 lanes per env) while one-env-per-lane would leave most SIMDs without a wave
    s->group_mapping = cfg->step_mapping != 1;   // one env per lane only on request (its kernels spill: see refresh_mapping)
    // log-partial rows = waves launched (the group mapping launches whole 256-thread workgroups)
    s->step_blocks = s->group_mapping ? (num_envs + RV_K1G_ENVS - 1) / RV_K1G_ENVS : (num_envs + 63) / 64;
    s->n_waves = s->group_mapping ? s->step_blocks * (RV_K1G_THREADS / 64) : s->step_blocks;
    // workspace: [log partials, padded to 128 B][scan descriptors: n x 32 B]
    s->ws_log_floats = (((size_t)(((num_envs + RV_K1G_ENVS - 1) / RV_K1G_ENVS) * (RV_K1G_THREADS / 64)) * ROVER_LOG_WORDS) + 31) & ~(size_t)31;
    // workspace: [log partials][scan descriptors n x 8 floats, padded to 128 B][the reset-wave counter, 128 B]
    // ... [ray pattern table of the wave-private scan, 1024 x 8 B] ...
    s->ws_bytes = (((s->ws_log_floats + (size_t)num_envs * 8) * sizeof(float) + 127) & ~(size_t)127) + 1024 * sizeof(float2) + 128;
    s->ray_xy = nullptr;
    s->fused = -1;
    s->single_tile_ok = true;
    s->log_deferred = false;
    *out = s;
    return ROVER_OK;
}

int rover_destroy(rover_sim *sim)
{
    delete sim;
    return ROVER_OK;
}

int rover_set_terrain(rover_sim *sim, const float *height, const float *obstacle, const uint8_t *safe_mask, int32_t H,
                      int32_t W, float resolution, float min_x, float min_y, const float *spawns, int32_t n_spawns)
{
    if (!sim) return fail(ROVER_ERR_INVALID, "sim is NULL");
    if (!height || !obstacle || !safe_mask || !spawns) return fail(ROVER_ERR_INVALID, "terrain pointer is NULL");
    if (H < 2 || W < 2 || resolution <= 0.0f || n_spawns < 1) return fail(ROVER_ERR_INVALID, "bad terrain shape");
    // spawn_draw = 1: row = (a * gid + b) mod n_spawns is a bijection of the global ids only while gid < n_spawns
    if (sim->p.cfg.spawn_draw == 1 && (int64_t)n_spawns < (int64_t)sim->p.env_id_offset + sim->p.n)
        return fail(ROVER_ERR_INVALID, "spawn_draw = 1 (distinct rows per reset batch) needs n_spawns >= env_id_offset + num_envs");
    RvParams &p = sim->p;
    p.height = height; p.lookup = height; p.obstacle = obstacle; p.safe_mask = safe_mask; p.spawns = spawns;
    p.H = H; p.W = W; p.n_spawns = n_spawns; p.res = resolution; p.min_x = min_x; p.min_y = min_y;
    // LDS tile: diagonal of the ray pattern in cells + slack
    const float diag = sqrtf(p.cfg.scan_size_x * p.cfg.scan_size_x + p.cfg.scan_size_y * p.cfg.scan_size_y);
    p.tile_dim = (int)ceilf(diag / resolution) + 6;
    p.height_q = nullptr;
    p.q_scale = 0.0f;
    configure_tile(sim, 4);
    if (sim->lds_bytes > 64 * 1024) return fail(ROVER_ERR_UNSUPPORTED, "ray pattern too large for the LDS tile (64 KiB)");
    sim->have_terrain = true;
    return ROVER_OK;
}

int rover_set_terrain_lookup(rover_sim *sim, const float *lookup_height)
{
    if (!sim) return fail(ROVER_ERR_INVALID, "sim is NULL");
    if (!sim->have_terrain) return fail(ROVER_ERR_STATE, "rover_set_terrain has not been called");
    sim->p.lookup = lookup_height ? lookup_height : sim->p.height;
    return ROVER_OK;
}

int rover_set_terrain_q16(rover_sim *sim, const int16_t *height_q, float q_scale)
{
    if (!sim) return fail(ROVER_ERR_INVALID, "sim is NULL");
    if (!sim->have_terrain) return fail(ROVER_ERR_STATE, "rover_set_terrain has not been called");
    if (!height_q) {  // back to the fp32 heightfield
        sim->p.height_q = nullptr;
        sim->p.q_scale = 0.0f;
        configure_tile(sim, 4);
        return ROVER_OK;
    }
    int q_exp = 0;
    if (!(q_scale > 0.0f) || frexpf(q_scale, &q_exp) != 0.5f)   // the kernel scales once, after interpolating the raw integers
        return fail(ROVER_ERR_INVALID, "q_scale must be a positive power of two");
    sim->p.height_q = height_q;
    sim->p.q_scale = q_scale;
    configure_tile(sim, 8);
    return ROVER_OK;
}

size_t rover_workspace_bytes(const rover_sim *sim) { return sim ? sim->ws_bytes : 0; }

int rover_bind(rover_sim *sim, float *state, void *workspace, size_t workspace_bytes)
{
    if (!sim || !state || !workspace) return fail(ROVER_ERR_INVALID, "NULL argument");
    if (workspace_bytes < sim->ws_bytes) return fail(ROVER_ERR_INVALID, "workspace too small");
    if (reinterpret_cast<uintptr_t>(workspace) & 127) return fail(ROVER_ERR_INVALID, "workspace must be 128-byte aligned");
    sim->state = state;
    sim->log_partial = static_cast<float *>(workspace);
    sim->p.scan_desc = sim->log_partial + sim->ws_log_floats;
    sim->p.log_counter = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + sim->ws_bytes - 128);
    sim->ray_xy = reinterpret_cast<float2 *>(static_cast<char *>(workspace) + sim->ws_bytes - 128 - 1024 * sizeof(float2));
    {
        DeviceGuard guard(sim->device);
        HIP_TRY(hipMemset(sim->p.log_counter, 0, 128));   // init-time, synchronous; the scan kernel returns it to zero after every reduction
        // the log-partial rows carry the tag of the launch that wrote them in word 15 and log_serial restarts at 1 with every
        // handle: a workspace that an earlier handle (or nobody) wrote must not hold rows that match this handle's tags
        HIP_TRY(hipMemset(sim->log_partial, 0, sim->ws_log_floats * sizeof(float)));
        // ORBIT grid_pattern: arange(-size/2, size/2 + 1e-9, res) evaluated in double, x fastest (App. C) -- the values the scan
        // kernels derive per thread
        const rover_config &c = sim->p.cfg;
        std::vector<float2> tab(1024);
        const int rays = c.scan_nx * c.scan_ny;
        for (int i = 0; i < 1024; ++i) {
            const int m = i >> 6, ln = i & 63, ri = ((m >> 2) << 8) + (ln << 2) + (m & 3);   // = private_ray_index(m, lane)
            const int r = ri < rays ? ri : 0;
            tab[i].x = (float)(-0.5 * (double)c.scan_size_x + (double)c.scan_resolution * (double)(r % c.scan_nx));
            tab[i].y = (float)(-0.5 * (double)c.scan_size_y + (double)c.scan_resolution * (double)(r / c.scan_nx));
        }
        HIP_TRY(hipMemcpy(sim->ray_xy, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    return ROVER_OK;
}

// LDS tile geometry for a heightfield array with `chunk_cells` cells per 16-byte chunk (4: fp32, 8: int16)
static void configure_tile(rover_sim *sim, int chunk_cells)
{
    RvParams &p = sim->p;
    p.chunk_cells = chunk_cells;
    // widest window: tile_dim cells + (chunk - 1) cells for the alignment of its left edge, rounded up to whole chunks
    p.tile_pitch = (p.tile_dim + 2 * (chunk_cells - 1)) & ~(chunk_cells - 1);
    const size_t cell_bytes = chunk_cells == 8 ? 2 : 4;
    const size_t tile_bytes = (size_t)p.tile_dim * p.tile_pitch * cell_bytes;
    // 8-wave workgroups: four per CU fill the 32 wave slots and may use 40 KiB of the 160 KiB LDS each
    p.tile_bufs = (192 * sizeof(float) + 2 * tile_bytes <= 40 * 1024) ? 2 : 1;
    sim->lds_bytes = 192 * sizeof(float) + p.tile_bufs * tile_bytes;
    const int by_lds = (int)(sim->max_lds / (sim->lds_bytes > 0 ? sim->lds_bytes : 1));   // the CU's LDS = what one workgroup may allocate
    const int by_waves = 32 / (RV_K2_THREADS / 64);
    const int per_cu = by_lds < by_waves ? (by_lds < 1 ? 1 : by_lds) : by_waves;
    sim->scan_wgs = sim->n_cu * per_cu;
    p.pq = p.tile_pitch / chunk_cells;
    p.wq = p.W / chunk_cells;
    p.cpr_log = 0;
    while ((1 << p.cpr_log) < p.pq) ++p.cpr_log;
    p.inv_res = 1.0f / p.res;
    p.inv_nx = 1.0f / (float)p.cfg.scan_nx;
    p.x_max = p.min_x + (float)(p.W - 1) * p.res;
    p.y_max = p.min_y + (float)(p.H - 1) * p.res;
    if (sim->lds_bytes < (RV_K2_THREADS + 16) * sizeof(float)) sim->lds_bytes = (RV_K2_THREADS + 16) * sizeof(float);
}

static int ready(rover_sim *sim)
{
    if (!sim) return fail(ROVER_ERR_INVALID, "sim is NULL");
    if (!sim->have_terrain) return fail(ROVER_ERR_STATE, "rover_set_terrain has not been called");
    if (!sim->state) return fail(ROVER_ERR_STATE, "rover_bind has not been called");
    return ROVER_OK;
}
// Between rover_step_begin and rover_step_finish the state holds a step whose reset / command update / observation rows are still
// owed: anything that advances the call counter, runs physics or resets envs there would reduce log rows under the wrong tag and
// step un-reset envs -- a caller bug that must not become silent state corruption.
static int closed(rover_sim *sim, const char *what)
{
    if (sim->phase_open) return fail(ROVER_ERR_STATE, "%s between rover_step_begin and rover_step_finish", what);
    return ROVER_OK;
}

int rover_reset(rover_sim *sim, float *obs, void *stream)
{
    if (int rc = ready(sim)) return rc;
    if (int rc = closed(sim, "rover_reset")) return rc;
    if (!obs) return fail(ROVER_ERR_INVALID, "obs is NULL");
    DeviceGuard guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    next_batch(sim);
    const RvParams &p = sim->p;
    hipLaunchKernelGGL(rover_reset_kernel, dim3((p.n + 63) / 64), dim3(64), 0, st, p, sim->state, nullptr, nullptr, nullptr,
                       nullptr, nullptr);
    launch_scan<1>(sim, p.n, st, obs, p.obs_w, 4, nullptr, 0, nullptr);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

int rover_reset_with_draws(rover_sim *sim, const uint8_t *mask, const int32_t *spawn_row, const float *yaw_u,
                           const float *theta_u, const float *heading_u, float *obs, void *stream)
{
    if (int rc = ready(sim)) return rc;
    if (int rc = closed(sim, "rover_reset_with_draws")) return rc;
    if (!spawn_row || !yaw_u || !theta_u || !heading_u || !obs) return fail(ROVER_ERR_INVALID, "NULL buffer");
    DeviceGuard guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const RvParams &p = sim->p;
    hipLaunchKernelGGL(rover_reset_kernel, dim3((p.n + 63) / 64), dim3(64), 0, st, p, sim->state, mask, spawn_row, yaw_u,
                       theta_u, heading_u);
    launch_scan<1>(sim, p.n, st, obs, p.obs_w, 4, nullptr, 0, nullptr);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

int rover_get_counter(const rover_sim *sim, uint64_t *counter)
{
    if (!sim || !counter) return fail(ROVER_ERR_INVALID, "NULL argument");
    *counter = sim->counter;
    return ROVER_OK;
}

int rover_set_counter(rover_sim *sim, uint64_t counter)
{
    if (!sim) return fail(ROVER_ERR_INVALID, "sim is NULL");
    if (int rc = closed(sim, "rover_set_counter")) return rc;
    sim->counter = counter;
    return ROVER_OK;
}

int rover_set_seed(rover_sim *sim, uint32_t seed_lo, uint32_t seed_hi)
{
    if (!sim) return fail(ROVER_ERR_INVALID, "sim is NULL");
    sim->p.cfg.seed_lo = seed_lo;
    sim->p.cfg.seed_hi = seed_hi;
    return ROVER_OK;
}

int rover_step(rover_sim *sim, const float *action, float *obs, float *reward, uint8_t *terminated, uint8_t *truncated,
               float *force, float *log, void *stream)
{
    if (int rc = ready(sim)) return rc;
    if (int rc = closed(sim, "rover_step")) return rc;
    if (!action || !obs || !reward || !terminated || !truncated || !log) return fail(ROVER_ERR_INVALID, "NULL buffer");
    DeviceGuard guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    next_batch(sim);
    MarkerRange whole(sim, "rover_step");
    sim->launch_error = ROVER_OK;
    launch_step_kernels(sim, st, action, obs, reward, terminated, truncated, force, log, nullptr);
    if (sim->launch_error != ROVER_OK) return sim->launch_error;
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

// ---- the step in two halves (slow path for user-written terms; see step_lane_body)
int rover_step_begin(rover_sim *sim, const float *action, float *reward, uint8_t *terminated, uint8_t *truncated, float *force,
                     void *stream)
{
    if (int rc = ready(sim)) return rc;
    if (int rc = closed(sim, "rover_step_begin")) return rc;
    if (!action || !reward || !terminated || !truncated || !force) return fail(ROVER_ERR_INVALID, "NULL buffer (the two-phase step needs the force rows)");
    DeviceGuard guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    next_batch(sim);
    sim->phase_open = true;
    MarkerRange whole(sim, "rover_step_begin");
    hipLaunchKernelGGL(rover_step_begin_kernel, dim3((sim->p.n + RV_K1G_ENVS - 1) / RV_K1G_ENVS), dim3(RV_K1G_THREADS), 0, st, sim->p, sim->state,
                       action, reward, terminated, truncated, force);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

int rover_step_finish(rover_sim *sim, const uint8_t *reset_mask, float *obs, float *force, float *log, void *stream)
{
    if (int rc = ready(sim)) return rc;
    if (!reset_mask || !obs || !force || !log) return fail(ROVER_ERR_INVALID, "NULL buffer");
    if (!sim->phase_open) return fail(ROVER_ERR_STATE, "rover_step_finish without rover_step_begin");
    DeviceGuard guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    sim->phase_open = false;
    MarkerRange whole(sim, "rover_step_finish");
    const int blocks = (sim->p.n + 63) / 64;
    hipLaunchKernelGGL(rover_step_finish_kernel, dim3(blocks), dim3(64), 0, st, sim->p, sim->state, obs, force, sim->log_partial, reset_mask);
    // the scan kernel of the two-launch path: observation rows from the descriptors the finish kernel wrote; its first workgroup
    // reduces extras["log"] (always eagerly here: this path is not the one-launch form, a deferred flush would find nothing)
    launch_scan<2>(sim, sim->p.n + 1, st, obs, sim->p.obs_w, 4, sim->log_partial, blocks, log);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

int rover_set_obs_streaming(rover_sim *sim, int32_t streaming)
{
    if (!sim) return fail(ROVER_ERR_INVALID, "sim is NULL");
    sim->p.nt_obs = streaming != 0;
    return ROVER_OK;
}

int rover_set_log_deferred(rover_sim *sim, int32_t deferred)
{
    if (!sim) return fail(ROVER_ERR_INVALID, "sim is NULL");
    sim->log_deferred = deferred != 0;
    return ROVER_OK;
}
int rover_flush_log(rover_sim *sim, float *log, void *stream)
{
    if (int rc = ready(sim)) return rc;
    if (!log) return fail(ROVER_ERR_INVALID, "log is NULL");
    DeviceGuard guard(sim->device);
    hipLaunchKernelGGL(rover_log_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), sim->p, sim->log_partial, sim->n_waves, log);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

int rover_set_markers(rover_sim *sim, int32_t enabled)
{
    if (!sim) return fail(ROVER_ERR_INVALID, "sim is NULL");
    if (enabled && !roctx_resolve()) return fail(ROVER_ERR_UNSUPPORTED, "no roctx library (librocprofiler-sdk-roctx / libroctx64) could be loaded");
    sim->markers = enabled != 0;
    return ROVER_OK;
}

int rover_kernel_names(const rover_sim *sim, char *step_kernel, char *scan_kernel, size_t cap)
{
    // The names rocprofv3's kernel trace prints for the two launches of rover_step with the CURRENT configuration and
    // terrain (without the "(anonymous namespace)::" qualifier and the parameter list): keys of bench.py's roofline block
    // and of profiles/hbm_traffic.json.
    if (!sim || !step_kernel || !scan_kernel || cap < 8) return fail(ROVER_ERR_INVALID, "bad argument");
    if (!sim->have_terrain) return fail(ROVER_ERR_STATE, "rover_set_terrain has not been called");
    refresh_mapping(const_cast<rover_sim *>(sim));
    if (fused_step(sim)) {   // one launch: the scan is the last phase of the step kernel; the second name is the log reduction's
        snprintf(step_kernel, cap, "%s<%s>", fused_form(sim) == 2 ? "rover_step_scan1_kernel" : "rover_step_scan_kernel",
                 sim->p.cfg.scan_surface == 0 ? "true" : "false");
        snprintf(scan_kernel, cap, "%s", sim->log_deferred ? "" : "rover_log_kernel");
        return ROVER_OK;
    }
    snprintf(step_kernel, cap, "%s", sim->group_mapping ? "rover_step_kernel_group" : "rover_step_kernel");
    const ScanForm f = scan_form_of(sim, 2);
    if (sim->scan_form == 7 && f.simple && f.q16)   // measurement hook: the wave-private scan as a kernel of its own
        snprintf(scan_kernel, cap, "rover_scan_private_kernel<%s>", f.tri ? "true" : "false");
    else if (f.simple)
        snprintf(scan_kernel, cap, "rover_scan_step_kernel<%s, %s, 1024, %d>", f.q16 ? "true" : "false", f.tri ? "true" : "false", f.epi);
    else
        snprintf(scan_kernel, cap, "rover_scan_obs_kernel<2, %s, %s>", f.q16 ? "true" : "false", f.tri ? "true" : "false");
    return ROVER_OK;
}

int rover_profile_step(rover_sim *sim, const float *action, float *obs, float *reward, uint8_t *terminated,
                       uint8_t *truncated, float *force, float *log, void *stream, float *ms_step_kernel,
                       float *ms_scan_kernel)
{
    // Same two launches as rover_step, bracketed by HIP events on `stream`; synchronises (profiling only).
    if (int rc = ready(sim)) return rc;
    if (int rc = closed(sim, "rover_profile_step")) return rc;
    if (!action || !obs || !reward || !terminated || !truncated || !log || !ms_step_kernel || !ms_scan_kernel)
        return fail(ROVER_ERR_INVALID, "NULL buffer");
    DeviceGuard guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    next_batch(sim);
    hipEvent_t ev[3];
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventCreate(&ev[i]));
    HIP_TRY(hipEventRecord(ev[0], st));
    sim->launch_error = ROVER_OK;
    launch_step_kernels(sim, st, action, obs, reward, terminated, truncated, force, log, ev[1]);
    if (sim->launch_error != ROVER_OK) {
        for (int i = 0; i < 3; ++i) (void)hipEventDestroy(ev[i]);
        return sim->launch_error;
    }
    HIP_TRY(hipEventRecord(ev[2], st));
    HIP_TRY(hipEventSynchronize(ev[2]));
    HIP_TRY(hipEventElapsedTime(ms_step_kernel, ev[0], ev[1]));
    HIP_TRY(hipEventElapsedTime(ms_scan_kernel, ev[1], ev[2]));
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventDestroy(ev[i]));
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

int rover_profile_event_overhead(rover_sim *sim, void *stream, int32_t reps, float *ms)
{
    // Elapsed time of an event pair with NOTHING between the two records, averaged over `reps`: the fixed cost every
    // interval of rover_profile_step carries.  Synchronises (measurement only).
    if (!sim || !ms || reps <= 0) return fail(ROVER_ERR_INVALID, "bad argument");
    DeviceGuard guard(sim->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t ev[3];
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventCreate(&ev[i]));
    double acc = 0.0;
    for (int r = 0; r < reps; ++r) {
        // three records like rover_profile_step; the middle interval pair is what a kernel-less step would report
        HIP_TRY(hipEventRecord(ev[0], st));
        HIP_TRY(hipEventRecord(ev[1], st));
        HIP_TRY(hipEventRecord(ev[2], st));
        HIP_TRY(hipEventSynchronize(ev[2]));
        float a = 0.0f, b = 0.0f;
        HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&b, ev[1], ev[2]));
        acc += 0.5 * ((double)a + (double)b);
    }
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventDestroy(ev[i]));
    *ms = (float)(acc / reps);
    return ROVER_OK;
}

int rover_mdp_terms(rover_sim *sim, int32_t n, const float *cmd_b, const float *action, const float *prev_action,
                    const int32_t *ep_len, const float *force, float *obs_distance, float *obs_angle, float *rew,
                    uint8_t *term, void *stream)
{
    if (!sim || n <= 0 || !cmd_b || !action || !prev_action || !ep_len || !force || !obs_distance || !obs_angle || !rew || !term)
        return fail(ROVER_ERR_INVALID, "bad argument");
    DeviceGuard guard(sim->device);
    hipLaunchKernelGGL(rover_mdp_terms_kernel, dim3((n + 127) / 128), dim3(128), 0, static_cast<hipStream_t>(stream), sim->p.cfg,
                       n, cmd_b, action, prev_action, ep_len, force, obs_distance, obs_angle, rew, term);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

int rover_ackermann(rover_sim *sim, int32_t n, const float *raw, float *processed, float *steer, float *wheel, void *stream)
{
    if (!sim || !raw || !processed || !steer || !wheel || n <= 0) return fail(ROVER_ERR_INVALID, "bad argument");
    DeviceGuard guard(sim->device);
    hipLaunchKernelGGL(rover_ackermann_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       sim->p.cfg, n, raw, processed, steer, wheel);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

int rover_height_scan(rover_sim *sim, float *scan, void *stream)
{
    if (int rc = ready(sim)) return rc;
    if (!scan) return fail(ROVER_ERR_INVALID, "scan is NULL");
    DeviceGuard guard(sim->device);
    const RvParams &p = sim->p;
    launch_scan<0>(sim, p.n, static_cast<hipStream_t>(stream), scan, p.rays, 0, nullptr, 0, nullptr);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

// measurement hook (tools/n_sweep.py): 0 = automatic choice, 1 = the generic scan kernel on the step path as well, 2 = the
// step form with one env per iteration, 7 = the wave-private scan (the scan phase of the one-launch kernels) as a kernel of its
// own behind the group-mapped step kernel.  (Forms 3 .. 6 -- 8 x 8 ray blocks per wave, XCD-aware pair dealing off / on -- were
// round-3 experiments; their outcome is in docs/history.md section 10, their code is gone.)
int rover_debug_set_scan_form(rover_sim *sim, int form)
{
    if (!sim || !(form == 0 || form == 1 || form == 2 || form == 7)) return ROVER_ERR_INVALID;
    sim->scan_form = form;
    return ROVER_OK;
}
// measurement hook: -1 = automatic, 0 = two launches per step, 1 / 2 = one launch (copy-wave form / single-tile form) wherever it can run
int rover_debug_set_fused(rover_sim *sim, int fused)
{
    if (!sim || fused < -1 || fused > 2) return ROVER_ERR_INVALID;   // 2 = the single-tile form (no copy waves)
    sim->fused = fused;
    return ROVER_OK;
}
#if defined(RV_K1_STAMP) || defined(RV_K1_LITE)
int rover_debug_set_k1_stamps(void *buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_k1_stamps), &buf, sizeof(buf)) == hipSuccess ? ROVER_OK : ROVER_ERR_HIP;
}
#endif

int rover_physics(rover_sim *sim, const float *steer_target, const float *wheel_target, int32_t substeps, float *force,
                  void *stream)
{
    if (int rc = ready(sim)) return rc;
    if (!steer_target || !wheel_target || substeps < 0) return fail(ROVER_ERR_INVALID, "bad argument");
    DeviceGuard guard(sim->device);
    if (sim->group_mapping)
        hipLaunchKernelGGL(rover_physics_kernel_group, dim3((sim->p.n + RV_K1G_ENVS - 1) / RV_K1G_ENVS), dim3(RV_K1G_THREADS), 0, static_cast<hipStream_t>(stream),
                           sim->p, sim->state, steer_target, wheel_target, substeps, force);
    else
        hipLaunchKernelGGL(rover_physics_kernel, dim3((sim->p.n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream),
                           sim->p, sim->state, steer_target, wheel_target, substeps, force);
    HIP_TRY(hipGetLastError());
    return ROVER_OK;
}

int rover_model_constants(float *out, int32_t cap)
{
    const float com[3] = RV_COM_B_INIT, inertia[3] = RV_INERTIA_B_INIT, wheel[6][3] = RV_WHEEL_B_INIT;
    const float pivot[3][3] = RV_BOGIE_PIVOT_INIT, axis[3][3] = RV_BOGIE_AXIS_INIT, binertia[3] = RV_BOGIE_INERTIA_INIT;
    float t[160];
    int n = 0;
    t[n++] = RV_M_TOTAL;
    for (int i = 0; i < 3; ++i) t[n++] = com[i];
    for (int i = 0; i < 3; ++i) t[n++] = inertia[i];
    for (int k = 0; k < 6; ++k) for (int i = 0; i < 3; ++i) t[n++] = wheel[k][i];
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) t[n++] = pivot[k][i];
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) t[n++] = axis[k][i];
    for (int k = 0; k < 3; ++k) t[n++] = binertia[k];
    t[n++] = RV_WHEEL_CONTACT_RADIUS;
    t[n++] = RV_STEER_INERTIA; t[n++] = RV_STEER_KP; t[n++] = RV_STEER_KD; t[n++] = RV_STEER_EFFORT; t[n++] = RV_STEER_VLIM;
    t[n++] = RV_WHEEL_INERTIA; t[n++] = RV_WHEEL_KP; t[n++] = RV_WHEEL_KD; t[n++] = RV_WHEEL_EFFORT; t[n++] = RV_WHEEL_VLIM;
    t[n++] = RV_BOGIE_QLIM; t[n++] = RV_BOGIE_DAMPING; t[n++] = RV_BAUMGARTE; t[n++] = RV_MAX_DEPENETRATION_VEL;
    t[n++] = RV_MAX_LINEAR_VEL; t[n++] = RV_GRAVITY; t[n++] = RV_OBSTACLE_EPS; t[n++] = RV_WARM_START; t[n++] = RV_STEER_QLIM;
    {
        const float lp[6][2][3] = RV_LINK_POINT_INIT;
        for (int sl = 0; sl < 6; ++sl) for (int r = 0; r < 2; ++r) for (int i = 0; i < 3; ++i) t[n++] = lp[sl][r][i];
        for (int sl = 0; sl < 6; ++sl) for (int r = 0; r < 2; ++r) for (int i = 0; i < 3; ++i)
            if (d_SLOT_host_lp(sl, r, i) != lp[sl][r][i]) return -1;      // the group mapping's table must be the same points
        t[n++] = RV_LINK_STIFFNESS;
    }
    {
        const float sm[3] = RV_SUBTREE_MASS_INIT, scom[3][3] = RV_SUBTREE_COM_INIT;
        for (int j = 0; j < 3; ++j) t[n++] = sm[j];
        for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) {
            if (d_SLOT_host_sc(j, i) != scom[j][i]) return -1;         // the group mapping's table must hold the same centres
            t[n++] = scom[j][i];
        }
    }
    if (out) for (int i = 0; i < n && i < cap; ++i) out[i] = t[i];
    return n;
}

int rover_state_words(void) { return ROVER_STATE_WORDS; }
size_t rover_config_bytes(void) { return sizeof(rover_config); }
const char *rover_last_error(void) { return g_err; }
const char *rover_version(void) { return "isaac_rover_orbit_amd 0.5.0 (gfx950)"; }

}  // extern "C"
