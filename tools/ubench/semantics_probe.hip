// Dev tool (GPU box): pins the instruction semantics the K1 solver relies on -- sign of zero of v_med3_f32 / v_max_f32 /
// v_min_f32, DPP row_ror:8, row_half_mirror, bank_mask -- so that oracle/rover_oracle.c can mirror them exactly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>

__global__ void probe_zero(const float *in, float *out, int n)
{
    // in: triples (x, lo, hi)
    int i = threadIdx.x;
    if (i >= n) return;
    float x = in[3 * i], lo = in[3 * i + 1], hi = in[3 * i + 2];
    float m3, mx, mn;
    asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(x), "v"(lo), "v"(hi));
    asm volatile("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(x), "v"(lo));
    asm volatile("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(x), "v"(hi));
    out[4 * i] = m3; out[4 * i + 1] = mx; out[4 * i + 2] = mn;
    out[4 * i + 3] = __builtin_amdgcn_fmed3f(x, lo, hi);
}

__global__ void probe_dpp(float *out)
{
    const int l = threadIdx.x;
    float x = (float)l, a, b, c, d;
    a = x; b = x; c = x; d = 1000.0f + x;
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(b));
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0x3 bound_ctrl:1" : "+v"(c));
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(d) : "v"(x));
    out[l] = a; out[64 + l] = b; out[128 + l] = c; out[192 + l] = d;
}

int main()
{
    const float Z = 0.0f, NZ = -0.0f;
    float h[] = {Z, NZ, Z,   NZ, NZ, Z,   Z, Z, Z,   NZ, Z, Z,   NZ, NZ, NZ,   Z, NZ, NZ,  1.0f, NZ, Z,  -1.0f, NZ, Z,
                 0.5f, -1.0f, 1.0f,  2.0f, -1.0f, 1.0f, -2.0f, -1.0f, 1.0f,  Z, -1.0f, 1.0f, NZ, -1.0f, 1.0f};
    const int n = sizeof(h) / sizeof(float) / 3;
    float *din, *dout;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, 4096);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_zero, dim3(1), dim3(64), 0, 0, din, dout, n);
    float o[256];
    hipMemcpy(o, dout, n * 16, hipMemcpyDeviceToHost);
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    printf("x lo hi -> v_med3(x,lo,hi) v_max(x,lo) v_min(x,hi) builtin_fmed3 (bit patterns)\n");
    for (int i = 0; i < n; ++i)
        printf("%08x %08x %08x -> %08x %08x %08x %08x\n", bits(h[3 * i]), bits(h[3 * i + 1]), bits(h[3 * i + 2]), bits(o[4 * i]),
               bits(o[4 * i + 1]), bits(o[4 * i + 2]), bits(o[4 * i + 3]));
    hipLaunchKernelGGL(probe_dpp, dim3(1), dim3(64), 0, 0, dout);
    hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
    const char *nm[] = {"x + row_ror:8(x)", "x + row_half_mirror(x)", "x + quad_perm xor2 (x), bank_mask 0x3", "mov row_mirror(x)"};
    for (int k = 0; k < 4; ++k) {
        printf("%s:\n ", nm[k]);
        for (int l = 0; l < 32; ++l) printf(" %g", o[64 * k + l]);
        printf("\n");
    }
    return 0;
}
