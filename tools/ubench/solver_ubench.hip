// Dev tool (GPU box): cycles per projected-Jacobi iteration of K1's group mapping (solver_iterations_group of
// rover_kernels.hip), one wave per SIMD, 256 workgroups x 256 threads like the real launch at N = 4096.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DVARIANT=n] -o build/solver_ubench tools/ubench/solver_ubench.hip
#include "../../isaac_rover_orbit_amd/csrc/rover_kernels.hip"
#include <vector>
#include <algorithm>

namespace {
__global__ __launch_bounds__(256) void solver_bench(const float *in, float *out, unsigned long long *cyc, int iters)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const float *q = in + (size_t)(tid & 1023) * 64;
    StepConsts K;
    make_step_consts(1.0f / 30.0f, K);
    Contact ct;
    for (int i = 0; i < 3; ++i) { ct.n[i] = q[i]; ct.t[i] = q[3 + i]; ct.s[i] = q[6 + i]; ct.jn_a[i] = q[9 + i]; ct.jt_a[i] = q[12 + i]; ct.js_a[i] = q[15 + i]; }
    ct.jn_b = q[18]; ct.jt_b = q[19]; ct.js_b = q[20];
    ct.mn = q[21]; ct.mt = q[22]; ct.ms = q[23]; ct.a_nt = q[24]; ct.a_ns = q[25]; ct.a_ts = q[26]; ct.bias = q[27];
    ct.ln = q[28]; ct.lt = 0.0f; ct.ls = 0.0f; ct.obst = 0.0f;
    const bool rb = (threadIdx.x & 8) != 0;
    RoleRows rr;
    const f2 minv0 = {K.inv_m, rb ? K.inv_I[2] : K.inv_I[0]};
    const f2 minv1 = {rb ? K.b_winv[0] : K.inv_m, rb ? 0.0f : K.inv_I[1]};
    make_role(ct, rb, minv0, minv1, rr);
    f2 V[2] = {{q[30], q[31]}, {q[32], rb ? 0.0f : q[33]}};
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    solver_iterations_group(K, ct, rr, V, q[34], 0.75f, iters);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[tid] = V[0].x + V[0].y + V[1].x + V[1].y + ct.ln + ct.lt + ct.ls;
    if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}
}  // namespace

int main()
{
    const int blocks = 256, threads = 256, iters = 4096;
    std::vector<float> h(1024 * 64);
    unsigned s = 12345u;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xFFFF) / 65536.0f * 0.2f - 0.1f; }
    for (int r = 0; r < 1024; ++r) { h[r * 64 + 21] = 2.0f; h[r * 64 + 22] = 2.0f; h[r * 64 + 23] = 2.0f; }
    float *din, *dout; unsigned long long *dc;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, blocks * threads * 4); hipMalloc(&dc, 8192 * 8);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(solver_bench, dim3(blocks), dim3(threads), 0, 0, din, dout, dc, rep == 0 ? 16 : iters);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> c(blocks * threads / 64);
    hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    printf("solver iteration: median %.1f cycles (min %.1f, max %.1f) per iteration over %d iterations\n",
           (double)c[c.size() / 2] / iters, (double)c[0] / iters, (double)c.back() / iters, iters);
    return 0;
}
