// Dev tool (GPU box): does a wave64 VALU instruction get cheaper when only part of the wave is active?  One wave per SIMD, a chain
// of independent v_fma_f32 / v_pk_fma_f32 under EXEC = all 64 lanes, the low 32, the low 16, and alternating quads.  s_memtime ticks
// per instruction (100 MHz-class constant clock; the ratio between the rows is what is wanted).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111"
template <int PK>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, int iters, unsigned long long mask)
{
    unsigned long long t0, t1, saved;
    asm volatile("v_mov_b32 v100, 1.0\n\tv_mov_b32 v101, 0.5\n\tv_mov_b32 v102, 0.25\n\tv_mov_b32 v103, 2.0\n\tv_mov_b32 v104, 1.0\n\tv_mov_b32 v105, 0.5\n\t"
                 "v_mov_b32 v106, 1.0\n\tv_mov_b32 v107, 1.0\n\tv_mov_b32 v108, 1.0\n\tv_mov_b32 v109, 0.5\n\tv_mov_b32 v110, 0.25\n\tv_mov_b32 v111, 2.0\n\t" ::: CLOB);
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=&s"(saved) : "s"(mask));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
        if (PK == 1) asm volatile(REP16(REP4("v_pk_fma_f32 v[100:101], v[102:103], v[104:105], v[106:107]\n\t")) ::: CLOB);
        else if (PK == 2) asm volatile(REP16(REP4("v_add_f32 v100, v101, v102\n\t")) ::: CLOB);
        else if (PK == 3) asm volatile(REP16(REP4("v_cvt_f32_i32 v100, v101\n\t")) ::: CLOB);
        else if (PK == 4) asm volatile(REP16(REP4("v_mad_u32_u24 v100, v101, v102, v103\n\t")) ::: CLOB);
        else if (PK == 5) asm volatile(REP16(REP4("v_add_f32_dpp v100, v101, v102 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")) ::: CLOB);
        else if (PK == 6) asm volatile(REP16(REP4("v_cndmask_b32 v100, v101, v102, vcc\n\t")) ::: CLOB);
        else if (PK == 7) asm volatile(REP16(REP4("v_fract_f32 v100, v101\n\t")) ::: CLOB);
        else if (PK == 8) asm volatile(REP16(REP4("v_cndmask_b32_e64 v100, v101, v102, s[20:21]\n\t")) ::: CLOB, "s20", "s21");
        else if (PK == 9) asm volatile(REP16("v_cndmask_b32 v100, v101, v102, vcc\n\tv_add_f32 v103, v104, v105\n\tv_add_f32 v106, v107, v108\n\tv_add_f32 v109, v110, v111\n\t") ::: CLOB);
        else if (PK == 10) asm volatile(REP16("v_cmp_lt_f32 vcc, v101, v102\n\tv_cndmask_b32 v100, v101, v102, vcc\n\tv_cmp_lt_f32 vcc, v104, v105\n\tv_cndmask_b32 v103, v104, v105, vcc\n\t") ::: CLOB, "vcc");
        else if (PK == 11) asm volatile(REP16(REP4("v_max_f32 v100, v101, v102\n\t")) ::: CLOB);
        else if (PK == 12) asm volatile(REP16(REP4("v_med3_f32 v100, v101, v102, v103\n\t")) ::: CLOB);
        else if (PK == 13) asm volatile(REP16(REP4("v_bfi_b32 v100, v101, v102, v103\n\t")) ::: CLOB);
        else if (PK == 14) asm volatile(REP16(REP4("v_mov_b32 v100, v101\n\t")) ::: CLOB);
        else if (PK == 15) asm volatile(REP16(REP4("v_mov_b32_dpp v100, v101 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")) ::: CLOB);
        else if (PK == 16) asm volatile(REP16(REP4("v_readfirstlane_b32 s20, v101\n\t")) ::: CLOB, "s20");
        else if (PK == 17) asm volatile(REP16(REP4("v_lshl_add_u32 v100, v101, 1, v102\n\t")) ::: CLOB);
        else asm volatile(REP16(REP4("v_fma_f32 v100, v101, v102, v103\n\t")) ::: CLOB);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_mov_b64 exec, %0" ::"s"(saved));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int PK> static double run(unsigned long long *d, std::vector<unsigned long long> &h, unsigned long long mask, int waves_per_simd)
{
    const int iters = 1000, blocks = 256, threads = 256 * waves_per_simd;
    hipLaunchKernelGGL(k<PK>, dim3(blocks), dim3(threads), 0, 0, d, 8, mask); hipDeviceSynchronize();
    hipLaunchKernelGGL(k<PK>, dim3(blocks), dim3(threads), 0, 0, d, iters, mask); hipDeviceSynchronize();
    int nw = blocks * threads / 64;
    hipMemcpy(h.data(), d, nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + nw);
    return (double)h[nw / 2] / iters / 64.0;
}
int main()
{
    unsigned long long *d; hipMalloc(&d, 65536 * 8); std::vector<unsigned long long> h(65536);
    for (int w = 1; w <= 4; w *= 2) {
        printf("%d wave(s) per SIMD, ticks per instruction and wave: v_fma %.2f  v_pk_fma %.2f  v_add %.2f  v_cvt_f32_i32 %.2f  v_mad_u32_u24 %.2f  v_add_dpp %.2f  cndmask(vcc) %.2f  v_fract %.2f\n",
               w, run<0>(d, h, ~0ull, w), run<1>(d, h, ~0ull, w), run<2>(d, h, ~0ull, w), run<3>(d, h, ~0ull, w), run<4>(d, h, ~0ull, w), run<5>(d, h, ~0ull, w), run<6>(d, h, ~0ull, w), run<7>(d, h, ~0ull, w));
        printf("    cndmask_e64(sgpr pair) %.2f  [cndmask + 3 v_add]/4 %.2f  [cmp, cndmask]x2 /4 %.2f  v_max %.2f  v_med3 %.2f  v_bfi %.2f  v_mov %.2f  v_mov_dpp %.2f  v_readfirstlane %.2f  v_lshl_add %.2f\n",
               run<8>(d, h, ~0ull, w), run<9>(d, h, ~0ull, w), run<10>(d, h, ~0ull, w), run<11>(d, h, ~0ull, w), run<12>(d, h, ~0ull, w), run<13>(d, h, ~0ull, w), run<14>(d, h, ~0ull, w), run<15>(d, h, ~0ull, w),
               run<16>(d, h, ~0ull, w), run<17>(d, h, ~0ull, w));
    }
    return 0;
}
