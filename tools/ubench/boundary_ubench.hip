// boundary_ubench.hip -- what do the first microseconds of a kernel that FOLLOWS a dependent kernel consist of?
// The scan kernel's grid (512 x 1024 threads, 38 KB of LDS) returning at once costs 1.85 us on an idle GPU and 4.65 us behind K1.
// Producers of different kinds run in front of the same empty follower; every follower has its own template tag so that
// `rocprofv3 --kernel-trace --stats` lists the cases separately.
//   hipcc --offload-arch=gfx950 -O3 -o boundary_ubench boundary_ubench.hip && rocprofv3 --kernel-trace --stats -- ./boundary_ubench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int TAG>
__global__ void follower(float *out, int n)
{
    extern __shared__ float lds[];
    if (n < 0) { lds[threadIdx.x] = 1.0f; out[blockIdx.x] = lds[0]; }
}

// K1's launch shape: 256 workgroups x 256 threads, one wave per SIMD.  Spins `spin` shader clocks, then every thread stores
// `words` floats (plain or non-temporal) at a stride that makes each wave write whole 256-byte segments.
template <bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void producer(float *buf, int words, long long spin)
{
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(2);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, total = gridDim.x * blockDim.x;
    for (int w = 0; w < words; ++w) {
        float v = float(tid + w);
        if (NT) __builtin_nontemporal_store(v, buf + size_t(w) * total + tid);
        else buf[size_t(w) * total + tid] = v;
    }
}

// a long straight-line instruction stream (evicts the instruction cache the way K1's ~40 KB of code does): 4096 dependent FMAs
// unrolled 2 x, each a distinct 8-byte encoding.
#define F8(a) a = __builtin_fmaf(a, 1.0000001f, 0.5f); a = __builtin_fmaf(a, 0.9999999f, 0.25f); a = __builtin_fmaf(a, 1.0000002f, 0.125f); \
              a = __builtin_fmaf(a, 0.9999998f, 0.75f); a = __builtin_fmaf(a, 1.0000003f, 0.375f); a = __builtin_fmaf(a, 0.9999997f, 0.625f); \
              a = __builtin_fmaf(a, 1.0000004f, 0.875f); a = __builtin_fmaf(a, 0.9999996f, 0.0625f);
#define F64(a) F8(a) F8(a) F8(a) F8(a) F8(a) F8(a) F8(a) F8(a)
#define F512(a) F64(a) F64(a) F64(a) F64(a) F64(a) F64(a) F64(a) F64(a)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void big_code(float *buf, int n)
{
    float a = float(threadIdx.x);
    F512(a) F512(a) F512(a) F512(a) F512(a) F512(a) F512(a) F512(a)
    if (n < 0) buf[threadIdx.x] = a;
}

#define CASE(TAG, LAUNCH_PRODUCER, WHAT)                                                                    \
    for (int r = 0; r < 200; ++r) {                                                                         \
        LAUNCH_PRODUCER;                                                                                    \
        hipLaunchKernelGGL(follower<TAG>, dim3(512), dim3(1024), 38 * 1024, 0, out, 1);                     \
    }                                                                                                       \
    (void)hipDeviceSynchronize();                                                                             \
    printf("follower<%d>: behind %s\n", TAG, WHAT);

int main()
{
    float *out, *buf;
    (void)hipMalloc(&out, 1 << 20);
    (void)hipMalloc(&buf, 64 << 20);
    const int T = 256 * 256;
    const long long spin = 60000;      // ~25 us of shader clocks (s_memtime runs at the shader clock on gfx950)
    CASE(0, hipLaunchKernelGGL(follower<99>, dim3(1), dim3(64), 0, 0, out, 1), "a one-wave empty kernel")
    CASE(1, hipLaunchKernelGGL(producer<false>, dim3(256), dim3(256), 0, 0, buf, 0, spin), "25 us spin, no stores")
    CASE(2, hipLaunchKernelGGL(producer<false>, dim3(256), dim3(256), 0, 0, buf, 2400000 / 4 / T, spin), "25 us spin + 2.4 MB plain stores")
    CASE(3, hipLaunchKernelGGL(producer<true>, dim3(256), dim3(256), 0, 0, buf, 2400000 / 4 / T, spin), "25 us spin + 2.4 MB non-temporal stores")
    CASE(4, hipLaunchKernelGGL(producer<false>, dim3(256), dim3(256), 0, 0, buf, 16000000 / 4 / T, spin), "25 us spin + 16 MB plain stores")
    CASE(5, hipLaunchKernelGGL(producer<true>, dim3(256), dim3(256), 0, 0, buf, 16000000 / 4 / T, spin), "25 us spin + 16 MB non-temporal stores")
    CASE(6, hipLaunchKernelGGL(producer<false>, dim3(256), dim3(256), 0, 0, buf, 2400000 / 4 / T, 0LL), "2.4 MB plain stores, no spin")
    CASE(7, hipLaunchKernelGGL(big_code, dim3(256), dim3(256), 0, 0, buf, 1), "32 KB of straight-line code, no stores")
    CASE(8, hipLaunchKernelGGL(producer<false>, dim3(4096), dim3(256), 0, 0, buf, 0, 0LL), "an empty 4096 x 256 grid")
    return 0;
}
