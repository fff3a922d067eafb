// Dev tool (GPU box): issue / dependent-latency cost of the instruction kinds K1's solver loop is made of, for ONE wave
// per SIMD (K1's regime at N = 4096) and for two waves per SIMD.   hipcc --offload-arch=gfx950 -O2 -o build/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int T>
__global__ void k(unsigned long long *out, float seed, int iters)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, c};
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
        if (T == 0) {  // dependent v_fma_f32 chain, 64 per block
            asm volatile(REP16(REP4("v_fma_f32 %0, %0, %1, %2\n\t")) : "+v"(a0) : "v"(b), "v"(c));
        } else if (T == 1) {  // 8 independent v_fma_f32 chains, 64 per block
            asm volatile(REP4(REP4("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"))
                         REP4(REP4("v_fma_f32 %4, %4, %8, %9\n\t")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if (T == 2) {  // dependent v_pk_fma_f32 chain
            asm volatile(REP16(REP4("v_pk_fma_f32 %0, %0, %1, %1\n\t")) : "+v"(p0) : "v"(pb));
        } else if (T == 3) {  // 4 independent v_pk_fma_f32 chains
            asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4\n\t")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        } else if (T == 4) {  // dependent v_add_f32_dpp chain with the 2 wait states the hazard needs (s_nop 1): 32 x (nop + add)
            asm volatile(REP16("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t") : "+v"(a0));
        } else if (T == 5) {  // 7 independent dpp adds per level (K1's shape): 1 nop + 7 adds, 8 levels = 64 instrs
            asm volatile(REP4("s_nop 1\n\t"
                              "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "s_nop 1\n\t"
                              "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                              "v_add_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6));
        } else if (T == 6) {  // dependent cmp + cndmask pairs (the clamp idiom): 32 pairs
            asm volatile(REP16("v_cmp_gt_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cmp_lt_f32_e32 vcc, %0, %2\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\t")
                         : "+v"(a0) : "v"(b), "v"(c) : "vcc");
        } else if (T == 7) {  // dependent v_med3_f32 chain
            asm volatile(REP16(REP4("v_med3_f32 %0, %0, %1, %2\n\t")) : "+v"(a0) : "v"(b), "v"(c));
        } else if (T == 8) {  // dependent alternation fma -> pk_fma (pair register written by a scalar op, read by a packed op)
            asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n\tv_pk_fma_f32 %3, %3, %4, %4\n\tv_add_f32 %0, %0, %1\n\tv_pk_mul_f32 %3, %3, %4\n\t")
                         : "+v"(a0) : "v"(b), "v"(c), "v"(p0), "v"(pb));
        } else if (T == 9) {  // dependent chain fma -> add -> fma -> max (scalar ops only, K1's impulse update idiom)
            asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n\tv_sub_f32 %0, %0, %1\n\tv_max_f32 %0, %0, %2\n\tv_mul_f32 %0, %0, %1\n\t")
                         : "+v"(a0) : "v"(b), "v"(c));
        } else if (T == 10) {  // s_nop 0 x 64
            asm volatile(REP16(REP4("s_nop 0\n\t")));
        } else if (T == 11) {  // scalar alu x 64 (dependent)
            int s = i;
            asm volatile(REP16(REP4("s_add_u32 %0, %0, 1\n\t")) : "+s"(s));
            a0 += (float)s;
        } else if (T == 12) {  // dependent pk_mul -> pk_fma -> add of the two halves -> fmac (row dot idiom), compiler-scheduled
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                p0 = p0 * pb;
                p0 = __builtin_elementwise_fma(p0, pb, pb);
                a0 = p0.x + p0.y;
                a0 = __builtin_fmaf(a0, b, a0);
                p0.x = a0;
            }
        } else if (T == 13) {  // v_mov_b32_dpp + add (what the compiler emits around pk adds)
            asm volatile(REP16("s_nop 1\n\tv_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32 %0, %0, %1\n\t") : "+v"(a0), "+v"(a1));
        } else if (T == 14) {  // dependent dpp add chain WITHOUT nops (is the hardware interlocked? result is discarded)
            asm volatile(REP16(REP4("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")) : "+v"(a0));
        } else if (T == 15) {  // v_rcp_f32 dependent (transcendental)
            asm volatile(REP16(REP4("v_rcp_f32 %0, %0\n\t")) : "+v"(a0));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.x + p3.x + pb.x;
    if (sink == 12345.678f) out[4096] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int T>
static double run(int blocks, int threads, int iters, unsigned long long *d, std::vector<unsigned long long> &h)
{
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f, 8);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, d, 1.0f, iters);
    hipDeviceSynchronize();
    int nw = blocks * threads / 64;
    hipMemcpy(h.data(), d, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<unsigned long long> v(h.begin(), h.begin() + nw);
    std::sort(v.begin(), v.end());
    return (double)v[nw / 2] / iters;
}

int main()
{
    unsigned long long *d;
    hipMalloc(&d, 8192 * sizeof(unsigned long long));
    std::vector<unsigned long long> h(8192);
    const int iters = 2000;
    const char *names[] = {"dep v_fma_f32 x64", "8-indep v_fma_f32 x64", "dep v_pk_fma_f32 x64", "4-indep v_pk_fma_f32 x64",
                           "dep (s_nop1 + v_add_dpp) x32", "K1-shape dpp: 8 x (s_nop1 + 7 adds)", "dep (cmp+cndmask) x32 pairs",
                           "dep v_med3_f32 x64", "dep fma/pk_fma/add/pk_mul x16", "dep fma/sub/max/mul x16", "s_nop 0 x64",
                           "dep s_add_u32 x64", "dep pk_mul/pk_fma/add/fmac x16", "(s_nop1 + mov_dpp + add) x16",
                           "dep v_add_dpp no nop x64", "dep v_rcp_f32 x64"};
    struct Cfg { int blocks, threads; const char *what; } cfgs[] = {
        {256, 64, "1 wave per CU"}, {256, 256, "1 wave per SIMD (4 per CU)"}, {256, 512, "2 waves per SIMD"}, {256, 1024, "4 waves per SIMD"}};
    for (auto &c : cfgs) {
        printf("== %s (%d blocks x %d threads): median cycles per block of instructions\n", c.what, c.blocks, c.threads);
        double r[16];
        r[0] = run<0>(c.blocks, c.threads, iters, d, h); r[1] = run<1>(c.blocks, c.threads, iters, d, h);
        r[2] = run<2>(c.blocks, c.threads, iters, d, h); r[3] = run<3>(c.blocks, c.threads, iters, d, h);
        r[4] = run<4>(c.blocks, c.threads, iters, d, h); r[5] = run<5>(c.blocks, c.threads, iters, d, h);
        r[6] = run<6>(c.blocks, c.threads, iters, d, h); r[7] = run<7>(c.blocks, c.threads, iters, d, h);
        r[8] = run<8>(c.blocks, c.threads, iters, d, h); r[9] = run<9>(c.blocks, c.threads, iters, d, h);
        r[10] = run<10>(c.blocks, c.threads, iters, d, h); r[11] = run<11>(c.blocks, c.threads, iters, d, h);
        r[12] = run<12>(c.blocks, c.threads, iters, d, h); r[13] = run<13>(c.blocks, c.threads, iters, d, h);
        r[14] = run<14>(c.blocks, c.threads, iters, d, h); r[15] = run<15>(c.blocks, c.threads, iters, d, h);
        for (int i = 0; i < 16; ++i) printf("  %-40s %8.1f cycles  (%.2f per instruction of 64)\n", names[i], r[i], r[i] / 64.0);
    }
    return 0;
}
