// Dev tool (GPU box): VGPR bank-conflict and operand-form costs of the instruction kinds in K1's solver loop, one wave per
// SIMD, explicit physical registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define CLOB "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131"

template <int T>
__global__ void k(unsigned long long *out, int iters)
{
    unsigned long long t0, t1;
    asm volatile(REP16("v_mov_b32 v100, 1.0\n\t") ::: CLOB);
    asm volatile("v_mov_b32 v100, 1.0\n\tv_mov_b32 v101, 0.5\n\tv_mov_b32 v102, 0.25\n\tv_mov_b32 v103, 2.0\n\tv_mov_b32 v104, 1.0\n\tv_mov_b32 v105, 0.5\n\tv_mov_b32 v106, 1.0\n\tv_mov_b32 v107, 1.0\n\t"
                 "v_mov_b32 v108, 1.0\n\tv_mov_b32 v109, 0.5\n\tv_mov_b32 v110, 0.25\n\tv_mov_b32 v111, 2.0\n\tv_mov_b32 v112, 1.0\n\tv_mov_b32 v113, 0.5\n\tv_mov_b32 v114, 1.0\n\tv_mov_b32 v115, 1.0\n\t"
                 "v_mov_b32 v116, 1.0\n\tv_mov_b32 v117, 0.5\n\tv_mov_b32 v118, 0.25\n\tv_mov_b32 v119, 2.0\n\tv_mov_b32 v120, 1.0\n\tv_mov_b32 v121, 0.5\n\tv_mov_b32 v122, 1.0\n\tv_mov_b32 v123, 1.0\n\t"
                 "v_mov_b32 v124, 1.0\n\tv_mov_b32 v125, 0.5\n\tv_mov_b32 v126, 0.25\n\tv_mov_b32 v127, 2.0\n\tv_mov_b32 v128, 1.0\n\tv_mov_b32 v129, 0.5\n\tv_mov_b32 v130, 1.0\n\tv_mov_b32 v131, 1.0\n\t" ::: CLOB);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < iters; ++i) {
        if (T == 0) asm volatile(REP16(REP4("v_fma_f32 v100, v101, v102, v103\n\t")) ::: CLOB);             // banks 1,2,3 distinct
        if (T == 1) asm volatile(REP16(REP4("v_fma_f32 v100, v104, v108, v112\n\t")) ::: CLOB);             // all bank 0
        if (T == 2) asm volatile(REP16(REP4("v_fma_f32 v100, v104, v108, v101\n\t")) ::: CLOB);             // two in bank 0
        if (T == 3) asm volatile(REP16(REP4("v_pk_fma_f32 v[100:101], v[102:103], v[104:105], v[106:107]\n\t")) ::: CLOB);  // pairs at banks (2,3),(0,1),(2,3)
        if (T == 4) asm volatile(REP16(REP4("v_pk_fma_f32 v[100:101], v[104:105], v[108:109], v[112:113]\n\t")) ::: CLOB);  // all (0,1)
        if (T == 5) asm volatile(REP16(REP4("v_pk_fma_f32 v[100:101], v[104:105], v[106:107], v[100:101]\n\t")) ::: CLOB);  // (0,1),(2,3),(0,1) dependent acc
        if (T == 6) asm volatile(REP16(REP4("v_pk_mul_f32 v[100:101], v[104:105], v[106:107]\n\t")) ::: CLOB);   // (0,1),(2,3)
        if (T == 7) asm volatile(REP16(REP4("v_pk_mul_f32 v[100:101], v[104:105], v[108:109]\n\t")) ::: CLOB);   // (0,1),(0,1)
        if (T == 8) asm volatile(REP16(REP4("v_add_f32 v100, v101, v102\n\t")) ::: CLOB);
        if (T == 9) asm volatile(REP16(REP4("v_add_f32 v100, v104, v108\n\t")) ::: CLOB);                        // same bank
        if (T == 10) asm volatile(REP16(REP4("v_add_f32_dpp v100, v101, v102 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")) ::: CLOB);  // indep dpp
        if (T == 11) asm volatile(REP16(REP4("v_pk_fma_f32 v[100:101], v[102:103], v[104:105], v[106:107] op_sel_hi:[1,0,1]\n\t")) ::: CLOB);
        if (T == 12) asm volatile(REP16("v_pk_mul_f32 v[100:101], v[104:105], v[106:107]\n\tv_pk_mul_f32 v[102:103], v[108:109], v[110:111]\n\tv_pk_mul_f32 v[112:113], v[116:117], v[118:119]\n\tv_pk_mul_f32 v[114:115], v[120:121], v[122:123]\n\t") ::: CLOB);  // 4 independent
        if (T == 13) asm volatile(REP16("v_pk_mul_f32 v[100:101], v[104:105], v[106:107]\n\tv_add_f32 v102, v100, v101\n\tv_pk_mul_f32 v[112:113], v[116:117], v[118:119]\n\tv_add_f32 v103, v112, v113\n\t") ::: CLOB);  // pk then dependent add
        if (T == 14) asm volatile(REP16("v_pk_mul_f32 v[100:101], v[104:105], v[106:107]\n\tv_pk_mul_f32 v[112:113], v[116:117], v[118:119]\n\tv_add_f32 v102, v100, v101\n\tv_add_f32 v103, v112, v113\n\t") ::: CLOB);  // same, interleaved
        if (T == 15) asm volatile(REP16(REP4("v_med3_f32 v100, v101, -v102, v102\n\t")) ::: CLOB);
        if (T == 16) asm volatile(REP16(REP4("v_fmac_f32 v100, v101, v102\n\t")) ::: CLOB);
        if (T == 17) asm volatile(REP16(REP4("v_mov_b32 v100, v101\n\t")) ::: CLOB);
        if (T == 18) asm volatile(REP16("s_nop 1\n\tv_add_f32 v100, v101, v102\n\tv_add_f32 v103, v105, v106\n\tv_add_f32 v104, v109, v110\n\t") ::: CLOB);  // cost of s_nop 1 among plain valu
        if (T == 19) asm volatile(REP16("v_add_f32 v100, v101, v102\n\tv_add_f32_dpp v100, v100, v100 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_add_f32 v103, v105, v106\n\tv_add_f32_dpp v103, v103, v103 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t") ::: CLOB);  // valu -> dpp back to back (hazard: result may be wrong; timing only)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int T> static double run(unsigned long long *d, std::vector<unsigned long long> &h)
{
    const int iters = 1000, blocks = 256, threads = 256;
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, d, 8); hipDeviceSynchronize();
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, d, iters); hipDeviceSynchronize();
    int nw = blocks * threads / 64;
    hipMemcpy(h.data(), d, nw * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.begin() + nw);
    return (double)h[nw / 2] / iters / 64.0;
}
int main()
{
    unsigned long long *d; hipMalloc(&d, 8192 * 8); std::vector<unsigned long long> h(8192);
    const char *nm[] = {"v_fma distinct banks", "v_fma 3 srcs bank 0", "v_fma 2 srcs bank 0", "pk_fma pairs (2,3)(0,1)(2,3)", "pk_fma all pairs (0,1)",
                        "pk_fma dep acc", "pk_mul (0,1)(2,3)", "pk_mul (0,1)(0,1)", "v_add distinct", "v_add same bank", "v_add_dpp indep", "pk_fma op_sel_hi",
                        "4 indep pk_mul", "pk_mul -> dep add (x2)", "pk_mul x2 -> adds interleaved", "v_med3 neg", "v_fmac", "v_mov", "s_nop1 + 3 add",
                        "add -> dpp back to back"};
    double r[20];
    r[0]=run<0>(d,h); r[1]=run<1>(d,h); r[2]=run<2>(d,h); r[3]=run<3>(d,h); r[4]=run<4>(d,h); r[5]=run<5>(d,h); r[6]=run<6>(d,h); r[7]=run<7>(d,h);
    r[8]=run<8>(d,h); r[9]=run<9>(d,h); r[10]=run<10>(d,h); r[11]=run<11>(d,h); r[12]=run<12>(d,h); r[13]=run<13>(d,h); r[14]=run<14>(d,h);
    r[15]=run<15>(d,h); r[16]=run<16>(d,h); r[17]=run<17>(d,h); r[18]=run<18>(d,h); r[19]=run<19>(d,h);
    for (int i = 0; i < 20; ++i) printf("%-36s %.2f cycles / instruction\n", nm[i], r[i]);
    return 0;
}
