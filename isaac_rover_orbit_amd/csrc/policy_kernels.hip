// policy_kernels.hip -- fused policy / value network forward pass on the f32-input MFMA (gfx950 / CDNA4, wave64).
//
// Reference being replaced (inference only): rover_envs/envs/navigation/learning/skrl/models.py HeightmapEncoder :24-36,
// GaussianNeuralNetwork.compute :89-103, DeterministicNeuralNetwork.compute :151-163; architecture from
// rover_envs/learning/train/get_models.py:36-62.  See include/rover_policy.h for the contract.
//
// One 512-thread workgroup = 16 observation rows.  The rows (16 x 965 fp32 = 61.8 KB, one contiguous block) are copied to
// LDS once; every layer is a [16 x K] x [K x N] product on v_mfma_f32_16x16x4_f32 with the A fragments read from LDS and
// the B fragments streamed from a packed, fragment-ordered weight buffer (one coalesced 16-byte load per lane feeds
// four MFMAs); activations never leave LDS; the last layer writes the (n, N_last) result.  The f32 MFMA is bit-for-bit
// a k-ordered fmaf chain (MI355X guide, "FP32-input MFMA"), so the result is reproducible on a CPU:
//   out[n] = act(chain_k fmaf(in[k], W[n][k], acc) + bias[n]),   acc0 = 0, k ascending
// with the chain cut into 8 contiguous k ranges (one per wave) combined as ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7))
// for `split_k` layers (the 961-wide first layer: every wave streams an eighth of its 307 KB of weights; the last
// 128 -> 2 layer).  tanh / exp are explicit
// fp32 sequences (rv_tanhf / rv_expf), the same text as in oracle/policy_oracle.c.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../../include/rover_hip.h"
#include "../../include/rover_policy.h"
#include "rover_internal.hpp"

namespace {

constexpr int POL_THREADS = 512;  // 8 waves: two per SIMD, so that one wave's LDS / global latencies hide under the other's MFMAs
constexpr int POL_WAVES = POL_THREADS / 64;
constexpr int POL_ROWS = 16;      // observation rows per workgroup = M of the MFMA tile
constexpr int POL_MAXT = 6;       // accumulator tiles a wave carries at once in a split-K layer
constexpr int POL_PF = 3;         // k groups of B fragments in flight per wave in a split-K layer (x tiles x 16 B per lane)
// ... and in a full-K layer of a shape without a compile-time path (mfma_all below), by the number of column tiles carried
#ifndef POL_PF_FULL
#define POL_PF_FULL(NT) ((NT) >= 3 ? 3 : 6)
#endif

typedef float v4f __attribute__((ext_vector_type(4)));

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Cephes expf / tanhf as explicit fp32 sequences (no contraction): identical text in oracle/policy_oracle.c
__device__ __forceinline__ float rv_expf(float x)
{
    if (x > 88.0f) return INFINITY;
    if (x < -88.0f) return 0.0f;
    const float z = floorf(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    const float zz = x * x;
    float p = 1.9875691500e-4f;
    p = p * x + 1.3981999507e-3f;
    p = p * x + 8.3334519073e-3f;
    p = p * x + 4.1665795894e-2f;
    p = p * x + 1.6666665459e-1f;
    p = p * x + 5.0000001201e-1f;
    p = p * zz + x + 1.0f;
    return ldexpf(p, (int)z);
}
__device__ __forceinline__ float rv_tanhf(float x)
{
    const float z = fabsf(x);
    if (z > 44.0f) return x > 0.0f ? 1.0f : -1.0f;
    if (z >= 0.625f) {
        const float s = rv_expf(z + z);
        const float r = 1.0f - 2.0f / (s + 1.0f);
        return x < 0.0f ? -r : r;
    }
    if (x == 0.0f) return x;
    const float s = x * x;
    float p = -5.70498872745e-3f;
    p = p * s + 2.06390887954e-2f;
    p = p * s - 5.37397155531e-2f;
    p = p * s + 1.33314422036e-1f;
    p = p * s - 3.33332819422e-1f;
    return p * s * x + x;
}
__device__ __forceinline__ float activate(float v, int act, float slope)
{
    if (act == ROVER_ACT_LEAKY_RELU) return v > 0.0f ? v : v * slope;
    if (act == ROVER_ACT_TANH) return rv_tanhf(v);
    return v;
}

#ifdef POL_STAMP
__device__ unsigned long long *g_pol_stamps = nullptr;
#endif

struct PolLaunch {
    int n_copies;     // replicas of the packed buffer; workgroup b reads replica b % n_copies
    unsigned copy_floats;  // floats per replica
    int tile_floats;  // LDS floats of the observation tile (16 * obs_dim, padded to 4)
    int part_floats;  // LDS floats of the raw-accumulator buffer
    int act_pitch;    // floats per row of an activation buffer
};

// acc[i] += A[16 x k-range] x B[k-range x 16] for NT column tiles at once: k groups [g0, g1) of 16 inputs, B fragments of
// tile i at Wt + i * tile_stride (+ g * 64 per group, lane already added).  The B fragments of the next PF groups are
// kept in flight in a register queue (one wave per SIMD: nothing else hides the L2 latency); the loop body is
// branch-free (a group that holds k >= K, which can only be the last one, is peeled off) so that the compiler's
// s_waitcnt stays a counted vmcnt(N); the MFMAs of consecutive tiles are independent (the 16x16x4 f32 MFMA issues every
// 32 cycles but has 40 cycles of dependent latency).
template <int NT>
__device__ __forceinline__ void mfma_one_group(v4f (&acc)[NT], const float (&a)[4], const v4f (&b)[NT])
{
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) {
#ifdef POL_ABL_NOMFMA   // diagnostic build: weight streaming + LDS time only
            acc[i][j] += a[j] * b[i][j];
#else
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[i][j], acc[i], 0, 0, 0);
#endif
        }
}
template <int NT, int PF>
__device__ __forceinline__ void mfma_groups(v4f (&acc)[NT], const float *arow_ptr, int akq, int K, const v4f *Wt,
                                            size_t tile_stride, int g0, int g1, int G)
{
#ifdef POL_ABL_NOLOAD   // diagnostic build: every B fragment comes from group 0 (L1 hits): MFMA + LDS time only
#define POL_G(x) 0
#else
#define POL_G(x) (x)
#endif
    if (g0 >= g1) return;
    const bool ragged = (K & 15) != 0 && g1 == G;   // the last group of the layer reads past K: peeled off below
    const int g_main = ragged ? g1 - 1 : g1;
    const int n_full = ((g_main - g0) / PF) * PF, rem = (g_main - g0) - n_full;
    v4f bq[PF][NT];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int i = 0; i < NT; ++i) bq[u][i] = Wt[i * tile_stride + (size_t)POL_G(min(g0 + u, G - 1)) * 64];
    const float *ap = arow_ptr + 16 * g0 + akq;
    for (int gb = g0; gb < g0 + n_full; gb += PF) {   // PF groups per trip, slot u of the queue <-> group gb + u
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            float a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = ap[16 * u + 4 * j];
            mfma_one_group<NT>(acc, a, bq[u]);
#pragma unroll
            for (int i = 0; i < NT; ++i) bq[u][i] = Wt[i * tile_stride + (size_t)POL_G(min(gb + u + PF, G - 1)) * 64];
            __builtin_amdgcn_sched_barrier(0);   // keep the refill here: PF - 1 groups of MFMAs cover its latency
        }
        ap += 16 * PF;
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {   // fewer than PF groups left; their fragments are already in slots 0 .. rem - 1
        if (u < rem) {
            float a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = ap[16 * u + 4 * j];
            mfma_one_group<NT>(acc, a, bq[u]);
        } else if (u == rem && ragged) {
            float a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * g_main + 4 * j + akq;
                const float v = arow_ptr[min(k, K - 1)];
                a[j] = k < K ? v : 0.0f;
            }
            mfma_one_group<NT>(acc, a, bq[u]);
        }
    }
}

// Full-K layer with a COMPILE-TIME number of k groups (the reference architecture's 80 -> 60, 64 -> 256, 256 -> 160, 160 -> 128
// layers: 5 / 4 / 16 / 10 groups): every B fragment of the wave's tiles is requested up front -- straight-line code, so the
// waits before each group's MFMAs are counted vmcnt(N) and the first MFMA starts when the first fragment lands.  With the
// generic queue (3 - 6 groups in flight) a wave's weight stream was latency-bound: 6 KB per L2 round trip x 8 waves ~ 24 B/clk,
// below the CU's ~33 B/clk L2 port.  (Raising the queue depth of the generic loop instead made things WORSE: its ragged tail
// has run-time bounds, the compiler waits with vmcnt(0) there, and with everything in the tail nothing overlapped.)
template <int NT, int GC>
__device__ __forceinline__ void mfma_all(v4f (&acc)[NT], const float *arow_ptr, int akq, const v4f *Wt, size_t tile_stride)
{
    v4f b[GC][NT];
#pragma unroll
    for (int g = 0; g < GC; ++g)
#pragma unroll
        for (int i = 0; i < NT; ++i) b[g][i] = Wt[i * tile_stride + (size_t)g * 64];
    const float *ap = arow_ptr + akq;
#pragma unroll
    for (int g = 0; g < GC; ++g) {
        float a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = ap[16 * g + 4 * j];
        mfma_one_group<NT>(acc, a, b[g]);
    }
}

// A split-K layer for the 16 rows of the workgroup (see the header): the 8 waves each take a contiguous range of k groups for up
// to POL_MAXT column tiles at once, raw partial accumulators go through `part`, and every thread combines one (row, column) in
// the fixed order ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)), adds the bias and applies the activation.  Shared by the
// descriptor-driven kernel and the reference-architecture kernel (same numerics by construction).
struct SplitKArgs {
    const v4f *W4;
    const float *bias;
    int K, N, G, T, act, rows, dst_pitch;
    float slope;
    bool last;
    float *dst_out, *dst_act, *part;
    const float *arow_ptr;
};
// combine step of a split-K pass: thread -> (row, column) with the column fastest; 16 * nt columns starting at tile t0
__device__ __forceinline__ void split_k_combine(const SplitKArgs &A, int t0, int nt, int tid)
{
    constexpr int ppitch = 16 * POL_MAXT + 4;
    const int ncols = 16 * nt;
    const float inv = 1.0f / (float)ncols;
    __syncthreads();
    for (int e = tid; e < POL_ROWS * ncols; e += POL_THREADS) {
        const int r = (int)(((float)e + 0.5f) * inv), c = e - r * ncols, col = 16 * t0 + c;   // e / ncols, exact here
        float q[POL_WAVES];
#pragma unroll
        for (int w = 0; w < POL_WAVES; ++w) q[w] = A.part[(w * POL_ROWS + r) * ppitch + c];
        const float sum = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
        if (r < A.rows && col < A.N) {
            const float v = activate(sum + A.bias[col], A.act, A.slope);
            if (A.last) A.dst_out[r * A.dst_pitch + col] = v;
            else A.dst_act[r * A.dst_pitch + col] = v;
        }
    }
}
__device__ __forceinline__ void split_k_layer(const SplitKArgs &A, int tid, int lane, int wave, int arow, int akq)
{
    const v4f *W4 = A.W4;
    const float *bias = A.bias;
    const int K = A.K, N = A.N, G = A.G, T = A.T, rows = A.rows, dst_pitch = A.dst_pitch;
    const bool last = A.last;
    float *dst_out = A.dst_out, *dst_act = A.dst_act, *part = A.part;
    const float *arow_ptr = A.arow_ptr;
    struct { int act; } lay = {A.act};
    struct { float leaky_slope; } d = {A.slope};
    const int gw = ceil_div(G, POL_WAVES), g0 = min(wave * gw, G), g1 = min(g0 + gw, G);
    const int ppitch = 16 * POL_MAXT + 4;
    for (int t0 = 0; t0 < T; t0 += POL_MAXT) {
        const int nt = min(POL_MAXT, T - t0);
        const v4f *Wt = W4 + (size_t)t0 * G * 64 + lane;
        if (t0 > 0) __syncthreads();  // the previous pass's partials have been combined
        auto pass = [&](auto nt_tag) {
            constexpr int NT = decltype(nt_tag)::value;
            v4f acc[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            mfma_groups<NT, POL_PF>(acc, arow_ptr, akq, K, Wt, (size_t)G * 64, g0, g1, G);
            float *pw = part + (wave * POL_ROWS + 4 * akq) * ppitch + arow;
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[j * ppitch + 16 * i] = acc[i][j];
        };
        switch (nt) {
            case 1: pass(std::integral_constant<int, 1>{}); break;
            case 2: pass(std::integral_constant<int, 2>{}); break;
            case 3: pass(std::integral_constant<int, 3>{}); break;
            case 4: pass(std::integral_constant<int, 4>{}); break;
            case 5: pass(std::integral_constant<int, 5>{}); break;
            default: pass(std::integral_constant<int, 6>{}); break;
        }
        split_k_combine(A, t0, nt, tid);
    }
}

// Packed weights of one layer: for column tile t (16 outputs), k group g (16 inputs = 4 MFMA k-steps), lane l:
// a float4 {W[n][k0], W[n][k0 + 4], W[n][k0 + 8], W[n][k0 + 12]} with n = 16 t + (l & 15), k0 = 16 g + (l >> 4);
// zero outside (N, K).  Index ((t * G + g) * 64 + l) * 4.
__global__ __launch_bounds__(POL_THREADS) void rover_policy_kernel(rover_policy_desc d, PolLaunch L,
                                                                   const float *__restrict__ packed,
                                                                   const float *__restrict__ obs, int n,
                                                                   float *__restrict__ out)
{
    extern __shared__ __align__(16) float lds[];
    // every workgroup streams the same 640 KB of weights in lock step; replicas spread those reads over L2 channels
    packed += (size_t)(blockIdx.x % (unsigned)L.n_copies) * L.copy_floats;
#ifdef POL_STAMP   // diagnostic build: s_memtime per phase of workgroup 0..; `out` row space is not used for stamps, a global is
#define PSTAMP(k) do { if (threadIdx.x == 0 && g_pol_stamps) g_pol_stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif
    PSTAMP(0);
    float *tile = lds;
    float *part = tile + L.tile_floats;                       // raw accumulators: [4][16][16 * MAXT + 4] or [16][16 T + 4]
    float *buf0 = part + L.part_floats;                       // [16][act_pitch]
    float *buf1 = buf0 + POL_ROWS * L.act_pitch;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar loop bounds, counted waits
    const int row0 = blockIdx.x * POL_ROWS;
    const int rows = min(POL_ROWS, n - row0);
    const int obs_dim = d.obs_dim;

    // ---- observation rows -> LDS (one contiguous block; 16-byte loads when the block is aligned and full)
    {
        const float *src = obs + (size_t)row0 * obs_dim;
        const int total = rows * obs_dim, total_pad = POL_ROWS * obs_dim;
        if (rows == POL_ROWS && (total & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            const v4f *s4 = reinterpret_cast<const v4f *>(src);
            v4f *t4 = reinterpret_cast<v4f *>(tile);
            const int n4 = total / 4;
            for (int i0 = tid; i0 < n4; i0 += 8 * POL_THREADS) {   // eight 16-byte loads in flight per lane
                v4f r[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) r[u] = __builtin_nontemporal_load(s4 + min(i0 + u * POL_THREADS, n4 - 1));  // read once: keep the weights in L2
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + u * POL_THREADS < n4) t4[i0 + u * POL_THREADS] = r[u];
            }
        } else {
            for (int i = tid; i < total_pad; i += POL_THREADS) tile[i] = i < total ? src[i] : 0.0f;
        }
    }
    __syncthreads();
    PSTAMP(1);

    const int n_layers = d.n_enc + d.n_mlp;
    const float *in = tile + (d.n_enc > 0 ? d.enc_offset : 0);  // A operand of the current layer
    int in_pitch = obs_dim;
    float *cur = buf0, *nxt = buf1;
    const int arow = lane & 15, akq = lane >> 4;  // A fragment: row, k within a k-step; C fragment: col = arow, rows 4 akq + j

    for (int li = 0; li < n_layers; ++li) {
        const rover_policy_layer lay = d.layers[li];
        const int K = lay.K, N = lay.N, G = ceil_div(K, 16), T = ceil_div(N, 16);
        const v4f *W4 = reinterpret_cast<const v4f *>(packed + lay.w_off);
        const float *bias = packed + lay.b_off;
        const bool last = li == n_layers - 1;
        // where this layer's activations go: the last encoder layer writes behind the proprioceptive columns of the MLP input
        const int col0 = (d.n_enc > 0 && li == d.n_enc - 1) ? d.prop_dim : 0;
        // two typed destinations instead of one `last ? global : LDS` pointer: a pointer that may be either is a FLAT pointer,
        // and every epilogue store through it became a flat_store_dword (115 of them in the ISA)
        float *dst_out = out + (size_t)row0 * N;   // global: the last layer
        float *dst_act = cur + col0;               // LDS: every other layer
        const int dst_pitch = last ? N : L.act_pitch;
        const float *arow_ptr = in + arow * in_pitch;

        if (lay.split_k) {
            SplitKArgs A;
            A.W4 = W4; A.bias = bias; A.K = K; A.N = N; A.G = G; A.T = T; A.act = lay.act; A.rows = rows; A.dst_pitch = dst_pitch;
            A.slope = d.leaky_slope; A.last = last; A.dst_out = dst_out; A.dst_act = dst_act; A.part = part; A.arow_ptr = arow_ptr;
            split_k_layer(A, tid, lane, wave, arow, akq);
        } else {
            // column tiles wave, wave + 8, wave + 16, ... belong to this wave; up to four of them are carried at once
            for (int t0 = wave; t0 < T; t0 += 4 * POL_WAVES) {
                const int nt = min(4, ceil_div(T - t0, POL_WAVES));
                const v4f *Wt = W4 + (size_t)t0 * G * 64 + lane;
                auto pass = [&](auto nt_tag) {
                    constexpr int NT = decltype(nt_tag)::value;
                    v4f acc[NT];
                    float bv[NT];   // this lane's bias per tile, fetched before the k loop (its latency hides under it)
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        acc[i] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                        bv[i] = bias[min(16 * (t0 + POL_WAVES * i) + arow, N - 1)];
                    }
                    const size_t ts = (size_t)POL_WAVES * G * 64;
                    if constexpr (NT <= 2) {
                        if ((K & 15) == 0 && G == 4) mfma_all<NT, 4>(acc, arow_ptr, akq, Wt, ts);
                        else if ((K & 15) == 0 && G == 5) mfma_all<NT, 5>(acc, arow_ptr, akq, Wt, ts);
                        else if ((K & 15) == 0 && G == 10) mfma_all<NT, 10>(acc, arow_ptr, akq, Wt, ts);
                        else if ((K & 15) == 0 && G == 16) mfma_all<NT, 16>(acc, arow_ptr, akq, Wt, ts);
                        else mfma_groups<NT, POL_PF_FULL(NT)>(acc, arow_ptr, akq, K, Wt, ts, 0, G, G);
                    } else {
                        mfma_groups<NT, POL_PF_FULL(NT)>(acc, arow_ptr, akq, K, Wt, ts, 0, G, G);
                    }
                    const int pd_off = 4 * akq * dst_pitch + 16 * t0 + arow;
                    auto epi = [&](auto act_tag) {   // one specialised copy per activation: no per-element branching
                        constexpr int ACT = decltype(act_tag)::value;
                        auto stores = [&](float *pd) {
#pragma unroll
                            for (int i = 0; i < NT; ++i)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    if (4 * akq + j < rows && 16 * (t0 + POL_WAVES * i) + arow < N)
                                        pd[j * dst_pitch + 16 * POL_WAVES * i] = activate(acc[i][j] + bv[i], ACT, d.leaky_slope);
                        };
                        if (last) stores(dst_out + pd_off);
                        else stores(dst_act + pd_off);
                    };
                    if (lay.act == ROVER_ACT_LEAKY_RELU) epi(std::integral_constant<int, ROVER_ACT_LEAKY_RELU>{});
                    else if (lay.act == ROVER_ACT_TANH) epi(std::integral_constant<int, ROVER_ACT_TANH>{});
                    else epi(std::integral_constant<int, ROVER_ACT_NONE>{});
                };
                switch (nt) {
                    case 1: pass(std::integral_constant<int, 1>{}); break;
                    case 2: pass(std::integral_constant<int, 2>{}); break;
                    case 3: pass(std::integral_constant<int, 3>{}); break;
                    default: pass(std::integral_constant<int, 4>{}); break;
                }
            }
        }
        PSTAMP(2 + li);
        if (last) break;
        // proprioceptive columns in front of the encoder output (models.py:93-96: cat([states[:, :4], encoder_output]))
        if (d.n_enc > 0 && li == d.n_enc - 1)
            for (int e = tid; e < POL_ROWS * d.prop_dim; e += POL_THREADS) {
                const int r = e / d.prop_dim, c = e - r * d.prop_dim;
                cur[r * L.act_pitch + c] = tile[r * obs_dim + c];
            }
        __syncthreads();
        in = cur;
        in_pitch = L.act_pitch;
        float *t = cur; cur = nxt; nxt = t;
    }
}

// ---------------------------------------------------------------------------------------------------- reference architecture
// The network the reference builds (get_models.py:36-62): 965-wide rows, encoder 961 -> 80 -> 60 on obs[:, 3:-1], MLP
// (4 + 60) -> 256 -> 160 -> 128 -> out (<= 16), split-K on the first and the last layer.  Same tiles, same chains, same
// combine order as the descriptor-driven kernel above -- bit-identical results -- but every layer's shape is a compile-time
// constant, so the B fragments of layer i + 1 are requested while layer i computes and are CARRIED IN REGISTERS across the
// layer boundary: the generic loop starts every layer with an exposed L2 round trip (~4 k cycles of start-up per small layer,
// tools/policy_stamps.py).  Register blocks: L2 5, L3 8, L4 32 (waves 0, 1 hold two column tiles), L5 10, L6 1 float4 per
// lane; at most L4 + L5 = 42 float4 = 168 VGPRs are alive at once (two waves per SIMD: 256 available).
template <int NT, int GC, int STRIDE>   // fragments b[g * STRIDE + i], i < NT
__device__ __forceinline__ void ref_load(v4f (&b)[GC * STRIDE], const v4f *Wt, size_t tile_stride)
{
#pragma unroll
    for (int g = 0; g < GC; ++g)
#pragma unroll
        for (int i = 0; i < NT; ++i) b[g * STRIDE + i] = Wt[i * tile_stride + (size_t)g * 64];
}
template <int NT, int GC, int STRIDE>
__device__ __forceinline__ void ref_mfma(v4f (&acc)[NT], const float *arow_ptr, int akq, const v4f (&b)[GC * STRIDE])
{
    const float *ap = arow_ptr + akq;
#pragma unroll
    for (int g = 0; g < GC; ++g) {
        float a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = ap[16 * g + 4 * j];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[g * STRIDE + i][j], acc[i], 0, 0, 0);
    }
}
// epilogue of a full-K layer: tiles t0 + 8 i; LeakyReLU; into an LDS activation buffer
template <int NT>
__device__ __forceinline__ void ref_store(const v4f (&acc)[NT], const float (&bv)[NT], float *dst_act, int dst_pitch, int t0, int N,
                                          int rows, int arow, int akq, float slope)
{
    float *pd = dst_act + 4 * akq * dst_pitch + 16 * t0 + arow;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * akq + j < rows && 16 * (t0 + POL_WAVES * i) + arow < N)
                pd[j * dst_pitch + 16 * POL_WAVES * i] = activate(acc[i][j] + bv[i], ROVER_ACT_LEAKY_RELU, slope);
}

// STAGE_TILE: this network is the first of the launch and copies the observation rows into LDS; false: a second network on
// the tile the first one staged (rover_policy_ref_pair_kernel: actor + critic of one rollout step read the rows ONCE).
template <bool STAGE_TILE>
__device__ __forceinline__ void ref_network(const rover_policy_desc &d, const PolLaunch &L, const float *__restrict__ packed,
                                            const float *__restrict__ obs, int n, float *__restrict__ out, float *lds)
{
    packed += (size_t)(blockIdx.x % (unsigned)L.n_copies) * L.copy_floats;
    if (STAGE_TILE) PSTAMP(0);
    float *tile = lds;
    float *part = tile + L.tile_floats;
    float *buf0 = part + L.part_floats;
    float *buf1 = buf0 + POL_ROWS * L.act_pitch;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * POL_ROWS;
    const int rows = min(POL_ROWS, n - row0);
    constexpr int OBS = 965, PROP = 4, ENC_OFF = 3;
    constexpr int G2 = 5, G3 = 4, G4 = 16, G5 = 10, G6 = 8;      // k groups of layers 2 .. 6 (K = 80, 64, 256, 160, 128)
    const int pitch = L.act_pitch;
    const float slope = d.leaky_slope;
    const int arow = lane & 15, akq = lane >> 4;
    auto Wof = [&](int li) { return reinterpret_cast<const v4f *>(packed + d.layers[li].w_off) + lane; };
    auto Bof = [&](int li) { return packed + d.layers[li].b_off; };

    // ---- observation rows -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers), and, queued right behind the
    // copy, layer 1's weights: they do not depend on the observations.  Waves 0 .. 6 request their WHOLE share (8 k groups x 5
    // column tiles: 30 fragments in front of the barrier, the last 10 right behind it) so that the 307 KB stream through the L2
    // port while the rows arrive; the wait in front of the barrier is a counted vmcnt(30) -- the copy, not the weights -- and the MFMAs below run as straight-line
    // code behind counted waits.  Wave 7 holds the ragged end (k groups 56 .. 60, the last one a single input) and takes the
    // generic queue.  (Measured and dropped: no tile at all -- every wave fetching its own A fragments, 32 four-byte loads per
    // lane, straight from global memory: 51 k cycles instead of 42 k; sixteen 16-byte row segments per load instruction are too
    // many requests.)
    constexpr int G1 = 61, GW1 = 8, T1 = 5, GA1 = 6, GB1 = GW1 - GA1;   // 6 of the 8 k groups up front, 2 behind the barrier
    v4f f1a[GA1 * T1], f1b[GB1 * T1];
    const bool full1 = wave < 7;
    {
        const float *src = obs + (size_t)row0 * OBS;
        const int total = rows * OBS, total_pad = POL_ROWS * OBS;
        const bool dma = rows == POL_ROWS && (total & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
        if (!STAGE_TILE) {
            // the rows are already in LDS
        } else if (dma) {
            const v4f *s4 = reinterpret_cast<const v4f *>(src);
            v4f *t4 = reinterpret_cast<v4f *>(tile);
            constexpr int n4 = POL_ROWS * OBS / 4;
            const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
#pragma unroll
            for (int i0 = 0; i0 < n4; i0 += POL_THREADS)
                if (i0 + tid < n4)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s4 + i0 + tid),
                                                     (__attribute__((address_space(3))) void *)(t4 + i0 + wave_base), 16, 0, 0);
        } else {
            for (int i = tid; i < total_pad; i += POL_THREADS) tile[i] = i < total ? src[i] : 0.0f;
        }
        asm volatile("" ::: "memory");   // the weight loads below stay BEHIND the copy in issue order (the counted wait relies on it)
        if (full1) ref_load<T1, GA1, T1>(f1a, Wof(0) + (size_t)(wave * GW1) * 64, (size_t)G1 * 64);
        if (STAGE_TILE) {
            if (full1) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();   // first network: the tile is complete; second network: the first one's combine has read `part`, its buffers are dead
    if (STAGE_TILE) PSTAMP(1);

    v4f f2[G2];
    float bv2 = 0.0f;
    const bool has2 = wave < 4;
    // ---- layer 1: 961 -> 80, split-K, into buf0: accumulate (specialised for waves 0 .. 6), then the shared combine
    {
        SplitKArgs A;
        A.W4 = reinterpret_cast<const v4f *>(packed + d.layers[0].w_off); A.bias = Bof(0);
        A.K = d.layers[0].K; A.N = d.layers[0].N; A.G = G1; A.T = T1; A.act = d.layers[0].act;
        A.rows = rows; A.dst_pitch = pitch; A.slope = slope; A.last = false; A.dst_out = out; A.dst_act = buf0; A.part = part;
        A.arow_ptr = tile + ENC_OFF + arow * OBS;
        constexpr int ppitch = 16 * POL_MAXT + 4;
        v4f acc[T1];
#pragma unroll
        for (int i = 0; i < T1; ++i) acc[i] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        if (full1) {
            ref_load<T1, GB1, T1>(f1b, Wof(0) + (size_t)(wave * GW1 + GA1) * 64, (size_t)G1 * 64);
            ref_mfma<T1, GA1, T1>(acc, A.arow_ptr + 16 * (wave * GW1), akq, f1a);
            ref_mfma<T1, GB1, T1>(acc, A.arow_ptr + 16 * (wave * GW1 + GA1), akq, f1b);
        } else mfma_groups<T1, POL_PF>(acc, A.arow_ptr, akq, A.K, A.W4 + lane, (size_t)G1 * 64, 7 * GW1, G1, G1);
        float *pw = part + (wave * POL_ROWS + 4 * akq) * ppitch + arow;
#pragma unroll
        for (int i = 0; i < T1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) pw[j * ppitch + 16 * i] = acc[i][j];
        // layer 2's fragments (column tile `wave` of 4; waves 4 .. 7 have none) travel under layer 1's combine
        if (has2) {
            ref_load<1, G2, 1>(f2, Wof(1) + (size_t)wave * G2 * 64, 0);
            bv2 = Bof(1)[min(16 * wave + arow, d.layers[1].N - 1)];
        }
        split_k_combine(A, 0, T1, tid);
    }
    if (STAGE_TILE) PSTAMP(2);
    // layer 3's fragments (tiles wave, wave + 8 of 16) travel under layer 2
    v4f f3[G3 * 2];
    float bv3[2];
    ref_load<2, G3, 2>(f3, Wof(2) + (size_t)wave * G3 * 64, (size_t)POL_WAVES * G3 * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) bv3[i] = Bof(2)[16 * (wave + POL_WAVES * i) + arow];
    __syncthreads();   // buf0 = layer 1's activations

    // ---- layer 2: 80 -> 60 into buf1[:, 4 ..], proprioceptive columns in front (models.py:93-96)
    if (has2) {
        v4f acc[1] = {(v4f){0.0f, 0.0f, 0.0f, 0.0f}};
        ref_mfma<1, G2, 1>(acc, buf0 + arow * pitch, akq, f2);
        const float bv[1] = {bv2};
        ref_store<1>(acc, bv, buf1 + PROP, pitch, wave, d.layers[1].N, rows, arow, akq, slope);
    }
    for (int e = tid; e < POL_ROWS * PROP; e += POL_THREADS) {
        const int r = e / PROP, c = e - r * PROP;
        buf1[r * pitch + c] = tile[r * OBS + c];
    }
    if (STAGE_TILE) PSTAMP(3);
    // layer 4's fragments (tiles wave, wave + 8 of 10: two for waves 0 and 1, one otherwise) travel under layer 3
    v4f f4[G4 * 2];
    float bv4[2];
    const bool two4 = wave < 2;
    if (two4) ref_load<2, G4, 2>(f4, Wof(3) + (size_t)wave * G4 * 64, (size_t)POL_WAVES * G4 * 64);
    else ref_load<1, G4, 2>(f4, Wof(3) + (size_t)wave * G4 * 64, 0);
    bv4[0] = Bof(3)[16 * wave + arow];
    bv4[1] = Bof(3)[min(16 * (wave + POL_WAVES) + arow, d.layers[3].N - 1)];
    __syncthreads();   // buf1 = MLP input

    // ---- layer 3: 64 -> 256 into buf0
    {
        v4f acc[2] = {(v4f){0.0f, 0.0f, 0.0f, 0.0f}, (v4f){0.0f, 0.0f, 0.0f, 0.0f}};
        ref_mfma<2, G3, 2>(acc, buf1 + arow * pitch, akq, f3);
        ref_store<2>(acc, bv3, buf0, pitch, wave, d.layers[2].N, rows, arow, akq, slope);
    }
    if (STAGE_TILE) PSTAMP(4);
    // layer 5's fragments (tile `wave` of 8) travel under layer 4
    v4f f5[G5];
    ref_load<1, G5, 1>(f5, Wof(4) + (size_t)wave * G5 * 64, 0);
    const float bv5 = Bof(4)[16 * wave + arow];
    __syncthreads();   // buf0 = layer 3's activations

    // ---- layer 4: 256 -> 160 into buf1
    if (two4) {
        v4f acc[2] = {(v4f){0.0f, 0.0f, 0.0f, 0.0f}, (v4f){0.0f, 0.0f, 0.0f, 0.0f}};
        ref_mfma<2, G4, 2>(acc, buf0 + arow * pitch, akq, f4);
        ref_store<2>(acc, bv4, buf1, pitch, wave, d.layers[3].N, rows, arow, akq, slope);
    } else {
        v4f acc[1] = {(v4f){0.0f, 0.0f, 0.0f, 0.0f}};
        ref_mfma<1, G4, 2>(acc, buf0 + arow * pitch, akq, f4);
        const float bv[1] = {bv4[0]};
        ref_store<1>(acc, bv, buf1, pitch, wave, d.layers[3].N, rows, arow, akq, slope);
    }
    if (STAGE_TILE) PSTAMP(5);
    // layer 6's fragment (split-K: k group `wave` of 8, the one column tile) travels under layer 5
    const v4f f6 = Wof(5)[(size_t)wave * 64];
    __syncthreads();   // buf1 = layer 4's activations

    // ---- layer 5: 160 -> 128 into buf0
    {
        v4f acc[1] = {(v4f){0.0f, 0.0f, 0.0f, 0.0f}};
        ref_mfma<1, G5, 1>(acc, buf1 + arow * pitch, akq, f5);
        const float bv[1] = {bv5};
        ref_store<1>(acc, bv, buf0, pitch, wave, d.layers[4].N, rows, arow, akq, slope);
    }
    if (STAGE_TILE) PSTAMP(6);
    __syncthreads();   // buf0 = layer 5's activations

    // ---- layer 6: 128 -> out, split-K with one k group per wave: the generic arithmetic (mfma_groups over [wave, wave + 1),
    // partials through `part`, fixed combine order) with the fragment already in registers
    {
        const int N = d.layers[5].N, act = d.layers[5].act;
        constexpr int ppitch = 16 * POL_MAXT + 4;
        v4f acc = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        const float *ap = buf0 + arow * pitch + 16 * wave + akq;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * j], f6[j], acc, 0, 0, 0);
        float *pw = part + (wave * POL_ROWS + 4 * akq) * ppitch + arow;
#pragma unroll
        for (int j = 0; j < 4; ++j) pw[j * ppitch] = acc[j];
        __syncthreads();
        const float *bias = Bof(5);
        float *dst_out = out + (size_t)row0 * N;
        for (int e = tid; e < POL_ROWS * 16; e += POL_THREADS) {
            const int r = e >> 4, c = e & 15;
            float q[POL_WAVES];
#pragma unroll
            for (int w = 0; w < POL_WAVES; ++w) q[w] = part[(w * POL_ROWS + r) * ppitch + c];
            const float sum = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
            if (r < rows && c < N) dst_out[r * N + c] = activate(sum + bias[c], act, slope);
        }
    }
    if (STAGE_TILE) PSTAMP(7);
}

__global__ __launch_bounds__(POL_THREADS) void rover_policy_ref_kernel(rover_policy_desc d, PolLaunch L,
                                                                       const float *__restrict__ packed,
                                                                       const float *__restrict__ obs, int n,
                                                                       float *__restrict__ out)
{
    extern __shared__ __align__(16) float lds[];
    ref_network<true>(d, L, packed, obs, n, out, lds);
}
// ---------------------------------------------------------------------------------------------------- actor + critic pair
// Two networks of the reference architecture on the same observation rows in ONE launch (the policy mean and the value of a
// rollout step), layer by layer TOGETHER -- not one network after the other (round 3: 34.8 us, 1.9 x one network):
//   * layer 1 (961 -> 80, split-K): a wave carries the five column tiles of BOTH networks over its k range -- the A fragments
//     (observation rows from the LDS tile) are read once for ten accumulator tiles, and the 2 x 307 KB of weights stream through
//     a three-group register queue behind counted waits;
//   * layers 2 .. 5: the column tiles of both networks are dealt to the eight waves as ONE list (tile tt = wave, wave + 8, ...;
//     tt < T: actor, else critic), so a layer costs one barrier, one LDS round trip and one epilogue for the pair -- layer 2's
//     eight tiles fill the eight waves (four of them idled with one network), layer 4's twenty are 3 + 2 per SIMD;
//   * layer 6 (128 -> out, split-K over the waves): two MFMA chains per wave, the two combines side by side.
// LDS: the carve of one network (tile | part | two activation buffers) -- the critic's layer-1 partials and its two activation
// buffers live in the observation tile, which is dead once every wave has read its layer-1 A fragments (the four proprioceptive
// columns are saved in registers first).
// Per output element the arithmetic is the single network's: the same k-ordered chain per (network, tile), the same split-K
// ranges and combine order -- bit-identical to two rover_policy_ref_kernel launches and to oracle/policy_oracle.c.
// (Wt[i]: the WAVE-UNIFORM start of tile i's fragments -- the loads take it as a scalar base plus the lane's offset, one VGPR for
// all tiles, instead of a 64-bit address pair per tile)
template <int NT, int QD>
__device__ __forceinline__ void pq_preload(v4f (&q)[QD][NT], const v4f *const (&Wt)[NT], int lane)
{
#pragma unroll
    for (int u = 0; u < QD; ++u)
#pragma unroll
        for (int i = 0; i < NT; ++i) q[u][i] = Wt[i][u * 64 + lane];
}
// acc[i] += A_i[16 x 16 GC] x B_i for NT tiles, k groups 0 .. GC - 1 in order; the queue holds groups g .. g + QD - 1
template <int NT, int GC, int QD>
__device__ __forceinline__ void pq_run(v4f (&acc)[NT], const float *const (&ap)[NT], v4f (&q)[QD][NT], const v4f *const (&Wt)[NT], int lane)
{
    static_assert(QD <= GC, "queue deeper than the layer");
#pragma unroll
    for (int g = 0; g < GC; ++g) {
        const int slot = g % QD;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[i][16 * g + 4 * j], q[slot][i][j], acc[i], 0, 0, 0);
        if (g + QD < GC) {
#pragma unroll
            for (int i = 0; i < NT; ++i) q[slot][i] = Wt[i][(g + QD) * 64 + lane];
        }
    }
}
// epilogue of one column tile of a full-K layer: bias, LeakyReLU, into an LDS activation buffer
__device__ __forceinline__ void pq_store(const v4f &acc, float bv, float *dst, int pitch, int tile, int N, int rows, int arow, int akq,
                                         float slope)
{
    float *pd = dst + 4 * akq * pitch + 16 * tile + arow;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * akq + j < rows && 16 * tile + arow < N) pd[j * pitch] = activate(acc[j] + bv, ROVER_ACT_LEAKY_RELU, slope);
}
__device__ __forceinline__ void ref_pair_network(const rover_policy_desc &da, const rover_policy_desc &db, const PolLaunch &L,
                                                 unsigned copy_floats_b,
                                                 const float *__restrict__ packed_a, const float *__restrict__ packed_b,
                                                 const float *__restrict__ obs, int n, float *__restrict__ out_a,
                                                 float *__restrict__ out_b, float *lds)
{
    packed_a += (size_t)(blockIdx.x % (unsigned)L.n_copies) * L.copy_floats;
    packed_b += (size_t)(blockIdx.x % (unsigned)L.n_copies) * copy_floats_b;   // the critic's own replica stride: the padded bias of
                                                                                // the last layer (N rounded up to 4) may differ from the actor's
    PSTAMP(0);
    constexpr int OBS = 965, PROP = 4, ENC_OFF = 3;
    constexpr int G1 = 61, GW1 = 8, T1 = 5, G2 = 5, G3 = 4, G4 = 16, G5 = 10;
    constexpr int PP1 = 16 * T1 + 4;                         // row pitch of the layer-1 partials (five tiles)
    constexpr int PP6 = 20;                                  // ... of the layer-6 partials (one tile)
    const int pitch = L.act_pitch;
    float *tile = lds;
    float *partA = tile + L.tile_floats;
    float *bufA0 = partA + L.part_floats, *bufA1 = bufA0 + POL_ROWS * pitch;
    float *partB = tile;                                     // the critic's share of the tile region (see above)
    float *bufB0 = tile + POL_WAVES * POL_ROWS * PP1, *bufB1 = tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * POL_ROWS;
    const int rows = min(POL_ROWS, n - row0);
    const float slope = da.leaky_slope;
    const int arow = lane & 15, akq = lane >> 4;
    auto Wof = [&](int net, int li) { return reinterpret_cast<const v4f *>((net ? packed_b : packed_a) + (net ? db : da).layers[li].w_off); };   // wave-uniform
    auto Bof = [&](int net, int li) { return (net ? packed_b : packed_a) + (net ? db : da).layers[li].b_off; };

    // ---- observation rows -> LDS by LDS-DMA, layer 1's first k group of both networks queued right behind the copy.  A group is 40
    // MFMAs per wave (1.3 k cycles of the SIMD's MFMA unit, twice that with the second wave of the SIMD): one group of lead covers
    // the L2 round trip of the next.  Measured (tools/policy_stamps.py pair, cycles: tile / layer 1 / whole kernel) for queue depths
    // 1 / 2 / 3: 3.9 k / 27.9 k / 67.4 k, 5.4 k / 27.3 k / 68.5 k, 6.4 k / 26.3 k / 68.8 k -- layer 1 is bound by the MFMA unit (20.5 k
    // of its 27 k), and fragments queued behind the copy only slow the copy down.
#ifndef POL_QD1
#define POL_QD1 1
#endif
#ifndef POL_QD4
#define POL_QD4 4
#endif
#ifndef POL_QD5
#define POL_QD5 3
#endif
    constexpr int QD1 = POL_QD1;
    const bool full1 = wave < 7;
    v4f q1[QD1][2 * T1];
    const v4f *W1[2 * T1];
#pragma unroll
    for (int i = 0; i < 2 * T1; ++i) W1[i] = Wof(i >= T1, 0) + ((size_t)(i % T1) * G1 + (size_t)wave * GW1) * 64;
    {
        const float *src = obs + (size_t)row0 * OBS;
        const int total = rows * OBS, total_pad = POL_ROWS * OBS;
        const bool dma = rows == POL_ROWS && (total & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
        if (dma) {
            const v4f *s4 = reinterpret_cast<const v4f *>(src);
            v4f *t4 = reinterpret_cast<v4f *>(tile);
            constexpr int n4 = POL_ROWS * OBS / 4;
            const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
#pragma unroll
            for (int i0 = 0; i0 < n4; i0 += POL_THREADS)
                if (i0 + tid < n4)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s4 + i0 + tid),
                                                     (__attribute__((address_space(3))) void *)(t4 + i0 + wave_base), 16, 0, 0);
        } else {
            for (int i = tid; i < total_pad; i += POL_THREADS) tile[i] = i < total ? src[i] : 0.0f;
        }
        asm volatile("" ::: "memory");   // the weight loads below stay BEHIND the copy in issue order (the counted wait relies on it)
        if (full1) {
            pq_preload<2 * T1, QD1>(q1, W1, lane);
#ifndef POL_X_NOTILEWAIT   // TIMING EXPERIMENT ONLY (tools/build_diag.py P_NOTILEWAIT): layer 1 starts on whatever the tile holds -- wrong results
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QD1 * 2 * T1) : "memory");     // the copy, not the fragments queued behind it
#endif
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
#ifndef POL_X_NOTILEWAIT
    __syncthreads();   // the tile is complete
#endif
    PSTAMP(1);
    const float prop = tid < POL_ROWS * PROP ? tile[(tid >> 2) * OBS + (tid & 3)] : 0.0f;   // models.py:93-96, saved before the tile is reused

    // ---- layer 1 of both networks: 961 -> 80, split-K over the waves
    v4f acc1[2 * T1];
#pragma unroll
    for (int i = 0; i < 2 * T1; ++i) acc1[i] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
    {
        const float *a1 = tile + ENC_OFF + arow * OBS;
        if (full1) {
            const float *ap[2 * T1];
#pragma unroll
            for (int i = 0; i < 2 * T1; ++i) ap[i] = a1 + 16 * (wave * GW1) + akq;
            pq_run<2 * T1, GW1, QD1>(acc1, ap, q1, W1, lane);
        } else {   // the ragged end (k groups 56 .. 60, the last one a single input): the generic queue, one network after the other
            v4f accA[T1], accB[T1];
#pragma unroll
            for (int i = 0; i < T1; ++i) { accA[i] = (v4f){0.0f, 0.0f, 0.0f, 0.0f}; accB[i] = (v4f){0.0f, 0.0f, 0.0f, 0.0f}; }
            mfma_groups<T1, POL_PF>(accA, a1, akq, da.layers[0].K, Wof(0, 0) + lane, (size_t)G1 * 64, 7 * GW1, G1, G1);
            mfma_groups<T1, POL_PF>(accB, a1, akq, db.layers[0].K, Wof(1, 0) + lane, (size_t)G1 * 64, 7 * GW1, G1, G1);
#pragma unroll
            for (int i = 0; i < T1; ++i) { acc1[i] = accA[i]; acc1[T1 + i] = accB[i]; }
        }
    }
    // layer 2's fragments (eight column tiles for eight waves: waves 0 .. 3 the actor's, 4 .. 7 the critic's) travel under the combine
    const int net2 = wave >> 2, t2 = wave & 3;
    v4f q2[G2][1];
    const v4f *W2[1] = {Wof(net2, 1) + (size_t)t2 * G2 * 64};
    pq_preload<1, G2>(q2, W2, lane);
    const float bv2 = Bof(net2, 1)[min(16 * t2 + arow, da.layers[1].N - 1)];
    __syncthreads();   // every wave has read its layer-1 A fragments: the tile region is free
    {
        float *pa = partA + (wave * POL_ROWS + 4 * akq) * PP1 + arow, *pb = partB + (wave * POL_ROWS + 4 * akq) * PP1 + arow;
#pragma unroll
        for (int i = 0; i < T1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { pa[j * PP1 + 16 * i] = acc1[i][j]; pb[j * PP1 + 16 * i] = acc1[T1 + i][j]; }
    }
    __syncthreads();
    {   // combine: ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)), bias, LeakyReLU; 16 x 80 outputs per network
        const int N1 = da.layers[0].N;
        const float *b1a = Bof(0, 0), *b1b = Bof(1, 0);
        for (int e = tid; e < 2 * POL_ROWS * 16 * T1; e += POL_THREADS) {
            const int net = e >= POL_ROWS * 16 * T1, ee = e - net * POL_ROWS * 16 * T1;
            const int r = (int)(((float)ee + 0.5f) * (1.0f / (float)(16 * T1))), c = ee - r * 16 * T1;
            const float *pp = (net ? partB : partA) + r * PP1 + c;
            float qq[POL_WAVES];
#pragma unroll
            for (int w = 0; w < POL_WAVES; ++w) qq[w] = pp[w * POL_ROWS * PP1];
            const float sum = ((qq[0] + qq[1]) + (qq[2] + qq[3])) + ((qq[4] + qq[5]) + (qq[6] + qq[7]));
            if (r < rows && c < N1) (net ? bufB0 : bufA0)[r * pitch + c] = activate(sum + (net ? b1b : b1a)[c], ROVER_ACT_LEAKY_RELU, slope);
        }
    }
    PSTAMP(2);
    // layer 3's fragments: tiles tt = wave + 8 i of 32 (tt < 16: actor tile tt, else critic tile tt - 16)
    v4f q3[G3][4];
    const v4f *W3[4];
    float bv3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tt = wave + POL_WAVES * i, net = tt >> 4, t = tt & 15;
        W3[i] = Wof(net, 2) + (size_t)t * G3 * 64;
        bv3[i] = Bof(net, 2)[16 * t + arow];
    }
    pq_preload<4, G3>(q3, W3, lane);
    __syncthreads();   // bufA0 / bufB0 = layer 1's activations; partB is dead (bufB1 overlays it)

    // ---- layer 2: 80 -> 60 into buf?1[:, 4 ..], the proprioceptive columns in front
    {
        v4f acc[1] = {(v4f){0.0f, 0.0f, 0.0f, 0.0f}};
        const float *ap[1] = {(net2 ? bufB0 : bufA0) + arow * pitch + akq};
        pq_run<1, G2, G2>(acc, ap, q2, W2, lane);
        pq_store(acc[0], bv2, (net2 ? bufB1 : bufA1) + PROP, pitch, t2, da.layers[1].N, rows, arow, akq, slope);
    }
    if (tid < POL_ROWS * PROP) {
        bufA1[(tid >> 2) * pitch + (tid & 3)] = prop;
        bufB1[(tid >> 2) * pitch + (tid & 3)] = prop;
    }
    PSTAMP(3);
    // layer 4's fragments: twenty tiles, tt = wave + 8 i: three for waves 0 .. 3, two for waves 4 .. 7; the first six k groups
    constexpr int QD4 = POL_QD4;
    const bool three4 = wave < 4;
    v4f q4[QD4][3];
    const v4f *W4[3];
    float bv4[3];
    int net4[3], t4[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int tt = min(wave + POL_WAVES * i, 19);
        net4[i] = tt >= 10; t4[i] = tt - 10 * net4[i];
        W4[i] = Wof(net4[i], 3) + (size_t)t4[i] * G4 * 64;
        bv4[i] = Bof(net4[i], 3)[16 * t4[i] + arow];
    }
    if (three4) {
        pq_preload<3, QD4>(q4, W4, lane);
    } else {
        v4f q42[QD4][2];
        const v4f *W42[2] = {W4[0], W4[1]};
        pq_preload<2, QD4>(q42, W42, lane);
#pragma unroll
        for (int u = 0; u < QD4; ++u) { q4[u][0] = q42[u][0]; q4[u][1] = q42[u][1]; }
    }
    __syncthreads();   // buf?1 = the MLP inputs

    // ---- layer 3: 64 -> 256 into buf?0
    {
        v4f acc[4];
        const float *ap[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            ap[i] = (i >= 2 ? bufB1 : bufA1) + arow * pitch + akq;      // tt = wave + 8 i: i < 2 actor, else critic
        }
        pq_run<4, G3, G3>(acc, ap, q3, W3, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            pq_store(acc[i], bv3[i], i >= 2 ? bufB0 : bufA0, pitch, (wave + POL_WAVES * i) & 15, da.layers[2].N, rows, arow, akq, slope);
    }
    PSTAMP(4);
    // layer 5's fragments: actor tile `wave`, critic tile `wave`; the first five of ten k groups
    constexpr int QD5 = POL_QD5;
    v4f q5[QD5][2];
    const v4f *W5[2] = {Wof(0, 4) + (size_t)wave * G5 * 64, Wof(1, 4) + (size_t)wave * G5 * 64};
    const float bv5[2] = {Bof(0, 4)[16 * wave + arow], Bof(1, 4)[16 * wave + arow]};
    pq_preload<2, QD5>(q5, W5, lane);
    __syncthreads();   // buf?0 = layer 3's activations

    // ---- layer 4: 256 -> 160 into buf?1
    if (three4) {
        v4f acc[3];
        const float *ap[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            acc[i] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            ap[i] = (net4[i] ? bufB0 : bufA0) + arow * pitch + akq;
        }
        pq_run<3, G4, QD4>(acc, ap, q4, W4, lane);
#pragma unroll
        for (int i = 0; i < 3; ++i) pq_store(acc[i], bv4[i], net4[i] ? bufB1 : bufA1, pitch, t4[i], da.layers[3].N, rows, arow, akq, slope);
    } else {
        v4f acc[2];
        const float *ap[2];
        v4f q42[QD4][2];
        const v4f *W42[2] = {W4[0], W4[1]};
#pragma unroll
        for (int u = 0; u < QD4; ++u) { q42[u][0] = q4[u][0]; q42[u][1] = q4[u][1]; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc[i] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            ap[i] = (net4[i] ? bufB0 : bufA0) + arow * pitch + akq;
        }
        pq_run<2, G4, QD4>(acc, ap, q42, W42, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) pq_store(acc[i], bv4[i], net4[i] ? bufB1 : bufA1, pitch, t4[i], da.layers[3].N, rows, arow, akq, slope);
    }
    PSTAMP(5);
    // layer 6's fragments (split-K: k group `wave` of 8, the one column tile of each network) travel under layer 5
    const v4f f6a = Wof(0, 5)[wave * 64 + lane], f6b = Wof(1, 5)[wave * 64 + lane];
    __syncthreads();   // buf?1 = layer 4's activations

    // ---- layer 5: 160 -> 128 into buf?0
    {
        v4f acc[2] = {(v4f){0.0f, 0.0f, 0.0f, 0.0f}, (v4f){0.0f, 0.0f, 0.0f, 0.0f}};
        const float *ap[2] = {bufA1 + arow * pitch + akq, bufB1 + arow * pitch + akq};
        pq_run<2, G5, QD5>(acc, ap, q5, W5, lane);
        pq_store(acc[0], bv5[0], bufA0, pitch, wave, da.layers[4].N, rows, arow, akq, slope);
        pq_store(acc[1], bv5[1], bufB0, pitch, wave, db.layers[4].N, rows, arow, akq, slope);
    }
    PSTAMP(6);
    __syncthreads();   // buf?0 = layer 5's activations

    // ---- layer 6: 128 -> out, split-K with one k group per wave, both networks; partials through partA ([0, 2560) actor, then critic)
    {
        v4f acA = (v4f){0.0f, 0.0f, 0.0f, 0.0f}, acB = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        const float *apA = bufA0 + arow * pitch + 16 * wave + akq, *apB = bufB0 + arow * pitch + 16 * wave + akq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acA = __builtin_amdgcn_mfma_f32_16x16x4f32(apA[4 * j], f6a[j], acA, 0, 0, 0);
            acB = __builtin_amdgcn_mfma_f32_16x16x4f32(apB[4 * j], f6b[j], acB, 0, 0, 0);
        }
        float *pw = partA + (wave * POL_ROWS + 4 * akq) * PP6 + arow;
#pragma unroll
        for (int j = 0; j < 4; ++j) { pw[j * PP6] = acA[j]; pw[POL_WAVES * POL_ROWS * PP6 + j * PP6] = acB[j]; }
        __syncthreads();
        const int net = tid >> 8, e = tid & 255, r = e >> 4, c = e & 15;
        const rover_policy_desc &d = net ? db : da;
        const int N = d.layers[5].N, act = d.layers[5].act;
        const float *pp = partA + net * POL_WAVES * POL_ROWS * PP6 + r * PP6 + c;
        float qq[POL_WAVES];
#pragma unroll
        for (int w = 0; w < POL_WAVES; ++w) qq[w] = pp[w * POL_ROWS * PP6];
        const float sum = ((qq[0] + qq[1]) + (qq[2] + qq[3])) + ((qq[4] + qq[5]) + (qq[6] + qq[7]));
        float *dst = (net ? out_b : out_a) + (size_t)row0 * N;
        if (r < rows && c < N) dst[r * N + c] = activate(sum + Bof(net, 5)[c], act, d.leaky_slope);
    }
    PSTAMP(7);
}
__global__ __launch_bounds__(POL_THREADS) void rover_policy_ref_pair_kernel(rover_policy_desc da, rover_policy_desc db, PolLaunch La,
                                                                            PolLaunch Lb, const float *__restrict__ packed_a,
                                                                            const float *__restrict__ packed_b,
                                                                            const float *__restrict__ obs, int n,
                                                                            float *__restrict__ out_a, float *__restrict__ out_b)
{
    extern __shared__ __align__(16) float lds[];
#ifdef POL_PAIR_SEQUENTIAL   // round 3's form: one network after the other on the staged tile
    ref_network<true>(da, La, packed_a, obs, n, out_a, lds);
    ref_network<false>(db, Lb, packed_b, obs, n, out_b, lds);
#else
    // same LDS carve and the same hidden-layer slope for both networks (identical shapes up to the last layer's width; the host
    // entry refuses a pair with different slopes); the weight replicas of each network at its own stride
    ref_pair_network(da, db, La, Lb.copy_floats, packed_a, packed_b, obs, n, out_a, out_b, lds);
#endif
}

// the shapes rover_policy_ref_kernel is written for
bool is_reference_architecture(const rover_policy_desc *d)
{
    if (d->obs_dim != 965 || d->prop_dim != 4 || d->enc_offset != 3 || d->enc_dim != 961 || d->n_enc != 2 || d->n_mlp != 4) return false;
    const int K[6] = {961, 80, 64, 256, 160, 128}, N[5] = {80, 60, 256, 160, 128};
    for (int i = 0; i < 6; ++i) {
        if (d->layers[i].K != K[i]) return false;
        if (i < 5 && (d->layers[i].N != N[i] || d->layers[i].act != ROVER_ACT_LEAKY_RELU)) return false;
        if ((d->layers[i].split_k != 0) != (i == 0 || i == 5)) return false;
    }
    return d->layers[5].N >= 1 && d->layers[5].N <= 16;
}

int check_desc(const rover_policy_desc *d)
{
    if (!d) return rover_internal_fail(ROVER_ERR_INVALID, "desc is NULL");
    const int nl = d->n_enc + d->n_mlp;
    if (d->n_enc < 0 || d->n_mlp < 1 || nl > ROVER_POLICY_MAX_LAYERS) return rover_internal_fail(ROVER_ERR_INVALID, "bad layer counts");
    if (d->obs_dim < 1 || d->prop_dim < 0 || d->prop_dim > d->obs_dim) return rover_internal_fail(ROVER_ERR_INVALID, "bad obs_dim / prop_dim");
    int width = d->prop_dim;
    if (d->n_enc > 0) {
        if (d->enc_dim < 1 || d->enc_offset < 0 || d->enc_offset + d->enc_dim > d->obs_dim)
            return rover_internal_fail(ROVER_ERR_INVALID, "encoder slice outside the observation row");
        int k = d->enc_dim;
        for (int i = 0; i < d->n_enc; ++i) {
            if (d->layers[i].K != k || d->layers[i].N < 1) return rover_internal_fail(ROVER_ERR_INVALID, "encoder layer shapes do not chain");
            k = d->layers[i].N;
        }
        width += k;
    }
    for (int i = d->n_enc; i < nl; ++i) {
        if (d->layers[i].K != width || d->layers[i].N < 1) return rover_internal_fail(ROVER_ERR_INVALID, "MLP layer shapes do not chain");
        width = d->layers[i].N;
    }
    for (int i = 0; i < nl; ++i)
        if (d->layers[i].act < 0 || d->layers[i].act > 2) return rover_internal_fail(ROVER_ERR_INVALID, "unknown activation");
    return ROVER_OK;
}

size_t layer_weight_floats(const rover_policy_layer &l) { return (size_t)ceil_div(l.N, 16) * ceil_div(l.K, 16) * 64 * 4; }
size_t layer_bias_floats(const rover_policy_layer &l) { return ((size_t)l.N + 3) & ~(size_t)3; }

}  // namespace

extern "C" {

int rover_policy_default_desc(rover_policy_desc *d, int32_t out_dim, int32_t final_tanh)
{
    if (!d || out_dim < 1) return rover_internal_fail(ROVER_ERR_INVALID, "bad argument");
    memset(d, 0, sizeof(*d));
    d->obs_dim = 965; d->prop_dim = 4; d->enc_offset = 3; d->enc_dim = 961;   // models.py:94-95
    d->n_enc = 2; d->n_mlp = 4; d->leaky_slope = 0.01f;
    const int K[6] = {961, 80, 64, 256, 160, 128}, N[6] = {80, 60, 256, 160, 128, out_dim};   // get_models.py:43-49
    for (int i = 0; i < 6; ++i) {
        d->layers[i].K = K[i]; d->layers[i].N = N[i];
        d->layers[i].act = ROVER_ACT_LEAKY_RELU;
        d->layers[i].split_k = 0;
    }
    d->layers[0].split_k = 1;   // 307 KB of weights: a quarter per wave
    d->layers[5].split_k = 1;   // one column tile only
    d->layers[5].act = final_tanh ? ROVER_ACT_TANH : ROVER_ACT_NONE;
    return ROVER_OK;
}

size_t rover_policy_packed_floats(const rover_policy_desc *d)
{
    if (!d || d->n_enc + d->n_mlp > ROVER_POLICY_MAX_LAYERS || d->n_enc < 0 || d->n_mlp < 0) return 0;
    size_t n = 0;
    for (int i = 0; i < d->n_enc + d->n_mlp; ++i) n += layer_weight_floats(d->layers[i]) + layer_bias_floats(d->layers[i]);
    return n;
}

int rover_policy_pack(rover_policy_desc *d, const float *const *weights, const float *const *biases, float *packed)
{
    if (int rc = check_desc(d)) return rc;
    if (!weights || !biases || !packed) return rover_internal_fail(ROVER_ERR_INVALID, "NULL argument");
    size_t off = 0;
    for (int li = 0; li < d->n_enc + d->n_mlp; ++li) {
        rover_policy_layer &l = d->layers[li];
        if (!weights[li] || !biases[li]) return rover_internal_fail(ROVER_ERR_INVALID, "layer weight / bias is NULL");
        const int G = ceil_div(l.K, 16), T = ceil_div(l.N, 16);
        l.w_off = (uint32_t)off;
        float *w = packed + off;
        for (int t = 0; t < T; ++t)
            for (int g = 0; g < G; ++g)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int nn = 16 * t + (lane & 15), k = 16 * g + 4 * j + (lane >> 4);
                        w[(((size_t)t * G + g) * 64 + lane) * 4 + j] = (nn < l.N && k < l.K) ? weights[li][(size_t)nn * l.K + k] : 0.0f;
                    }
        off += layer_weight_floats(l);
        l.b_off = (uint32_t)off;
        for (size_t i = 0; i < layer_bias_floats(l); ++i) packed[off + i] = i < (size_t)l.N ? biases[li][i] : 0.0f;
        off += layer_bias_floats(l);
    }
    return ROVER_OK;
}

int rover_policy_forward(const rover_policy_desc *d, const float *packed, int32_t n_copies, const float *obs, int32_t n,
                         float *out, void *stream)
{
    if (int rc = check_desc(d)) return rc;
    if (!packed || !obs || !out || n < 1 || n_copies < 1) return rover_internal_fail(ROVER_ERR_INVALID, "bad argument");
    if (reinterpret_cast<uintptr_t>(packed) & 15) return rover_internal_fail(ROVER_ERR_INVALID, "packed weights must be 16-byte aligned");
    PolLaunch L;
    L.n_copies = n_copies;
    L.copy_floats = (unsigned)rover_policy_packed_floats(d);
    L.tile_floats = (POL_ROWS * d->obs_dim + 3) & ~3;
    int width = d->prop_dim, part_floats = 0;
    const int nl = d->n_enc + d->n_mlp;
    for (int i = 0; i < nl; ++i) {
        const int T = ceil_div(d->layers[i].N, 16);
        const int pf = d->layers[i].split_k ? POL_WAVES * POL_ROWS * (16 * POL_MAXT + 4) : 0;
        part_floats = part_floats > pf ? part_floats : pf;
        const int w = 16 * T + ((d->n_enc > 0 && i == d->n_enc - 1) ? d->prop_dim : 0);
        width = width > w ? width : w;
    }
    L.part_floats = (part_floats + 3) & ~3;
    L.act_pitch = ((width + 3) & ~3) + 4;
    size_t lds = sizeof(float) * ((size_t)L.tile_floats + L.part_floats + 2 * POL_ROWS * L.act_pitch);
    if (lds > 160 * 1024) return rover_internal_fail(ROVER_ERR_UNSUPPORTED, "network too wide for the 160 KiB LDS");
    // the reference's architecture has its own kernel (layer shapes at compile time, fragments carried across layers); every
    // other descriptor runs the generic one.  ROVER_POLICY_GENERIC=1 forces the generic kernel (A/B measurements, tests).
    static const bool force_generic = getenv("ROVER_POLICY_GENERIC") != nullptr && getenv("ROVER_POLICY_GENERIC")[0] == '1';
    const bool ref = !force_generic && is_reference_architecture(d);
    const void *kfn = ref ? reinterpret_cast<const void *>(rover_policy_ref_kernel) : reinterpret_cast<const void *>(rover_policy_kernel);
    hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return rover_internal_fail(ROVER_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    if (ref)
        hipLaunchKernelGGL(rover_policy_ref_kernel, dim3(ceil_div(n, POL_ROWS)), dim3(POL_THREADS), lds, static_cast<hipStream_t>(stream),
                           *d, L, packed, obs, n, out);
    else
        hipLaunchKernelGGL(rover_policy_kernel, dim3(ceil_div(n, POL_ROWS)), dim3(POL_THREADS), lds, static_cast<hipStream_t>(stream),
                           *d, L, packed, obs, n, out);
    e = hipGetLastError();
    if (e != hipSuccess) return rover_internal_fail(ROVER_ERR_HIP, "rover_policy_kernel launch: %s", hipGetErrorString(e));
    return ROVER_OK;
}

int rover_policy_forward_pair(const rover_policy_desc *da, const float *packed_a, const rover_policy_desc *db, const float *packed_b,
                              int32_t n_copies, const float *obs, int32_t n, float *out_a, float *out_b, void *stream)
{
    if (int rc = check_desc(da)) return rc;
    if (int rc = check_desc(db)) return rc;
    if (!packed_a || !packed_b || !obs || !out_a || !out_b || n < 1 || n_copies < 1) return rover_internal_fail(ROVER_ERR_INVALID, "bad argument");
    if ((reinterpret_cast<uintptr_t>(packed_a) | reinterpret_cast<uintptr_t>(packed_b)) & 15)
        return rover_internal_fail(ROVER_ERR_INVALID, "packed weights must be 16-byte aligned");
    if (!is_reference_architecture(da) || !is_reference_architecture(db))
        return rover_internal_fail(ROVER_ERR_UNSUPPORTED, "rover_policy_forward_pair: both networks must have the reference architecture "
                                                          "(call rover_policy_forward twice otherwise)");
    if (da->leaky_slope != db->leaky_slope)   // the pair kernel activates the hidden layers of both networks with one slope
        return rover_internal_fail(ROVER_ERR_UNSUPPORTED, "rover_policy_forward_pair: the two networks use different leaky-ReLU slopes "
                                                          "(call rover_policy_forward twice)");
    PolLaunch L[2];
    size_t lds = 0;
    const rover_policy_desc *dd[2] = {da, db};
    for (int k = 0; k < 2; ++k) {   // same LDS carve for both (identical shapes up to the last layer's width <= 16)
        const rover_policy_desc *d = dd[k];
        L[k].n_copies = n_copies;
        L[k].copy_floats = (unsigned)rover_policy_packed_floats(d);
        L[k].tile_floats = (POL_ROWS * d->obs_dim + 3) & ~3;
        int width = d->prop_dim, part_floats = 0;
        const int nl = d->n_enc + d->n_mlp;
        for (int i = 0; i < nl; ++i) {
            const int T = ceil_div(d->layers[i].N, 16);
            const int pf = d->layers[i].split_k ? POL_WAVES * POL_ROWS * (16 * POL_MAXT + 4) : 0;
            part_floats = part_floats > pf ? part_floats : pf;
            const int w = 16 * T + ((d->n_enc > 0 && i == d->n_enc - 1) ? d->prop_dim : 0);
            width = width > w ? width : w;
        }
        L[k].part_floats = (part_floats + 3) & ~3;
        L[k].act_pitch = ((width + 3) & ~3) + 4;
        const size_t need = sizeof(float) * ((size_t)L[k].tile_floats + L[k].part_floats + 2 * POL_ROWS * L[k].act_pitch);
        lds = lds > need ? lds : need;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rover_policy_ref_pair_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return rover_internal_fail(ROVER_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(rover_policy_ref_pair_kernel, dim3(ceil_div(n, POL_ROWS)), dim3(POL_THREADS), lds, static_cast<hipStream_t>(stream),
                       *da, *db, L[0], L[1], packed_a, packed_b, obs, n, out_a, out_b);
    e = hipGetLastError();
    if (e != hipSuccess) return rover_internal_fail(ROVER_ERR_HIP, "rover_policy_ref_pair_kernel launch: %s", hipGetErrorString(e));
    return ROVER_OK;
}

#ifdef POL_STAMP
int rover_debug_set_policy_stamps(void *buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pol_stamps), &buf, sizeof(buf)) == hipSuccess ? ROVER_OK : ROVER_ERR_HIP;
}
#endif

}  // extern "C"
