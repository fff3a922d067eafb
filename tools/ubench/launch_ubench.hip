// launch_ubench.hip -- what does it cost to dispatch and retire an (almost) empty grid of a given shape on gfx950?
// Each variant is its own kernel name so that `rocprofv3 --kernel-trace --stats` lists them separately.
//   hipcc --offload-arch=gfx950 -O3 -o launch_ubench launch_ubench.hip && rocprofv3 --kernel-trace --stats -- ./launch_ubench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int TAG>
__global__ void empty_kernel(float *out, int n)
{
    extern __shared__ float lds[];
    if (n < 0) { lds[threadIdx.x] = 1.0f; out[blockIdx.x] = lds[0]; }   // never taken: keeps the LDS allocation and the argument alive
}
// the same with the register footprint of the scan kernel's waves_per_eu(8, 8) / of K1's one wave per SIMD
template <int TAG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void empty_fat_kernel(float *out, int n)
{
    if (n < 0) out[blockIdx.x] = 1.0f;
}

#define RUN(TAG, GRID, BLOCK, LDS)                                                                  \
    for (int r = 0; r < 200; ++r) {                                                                 \
        hipLaunchKernelGGL(empty_kernel<TAG>, dim3(GRID), dim3(BLOCK), LDS, 0, out, 1);             \
        hipLaunchKernelGGL(empty_kernel<999>, dim3(1), dim3(64), 0, 0, out, 1);                     \
    }                                                                                               \
    hipDeviceSynchronize();                                                                         \
    printf("tag %d: grid %d x block %d, lds %d B\n", TAG, GRID, BLOCK, LDS);

int main()
{
    float *out;
    hipMalloc(&out, 1 << 20);
    RUN(1, 512, 1024, 38 * 1024)     // the scan kernel's grid
    RUN(2, 512, 1024, 0)
    RUN(3, 256, 1024, 38 * 1024)
    RUN(4, 1024, 512, 19 * 1024)
    RUN(5, 512, 512, 38 * 1024)
    RUN(6, 256, 256, 0)              // K1's grid (group mapping, N = 4096)
    RUN(7, 2048, 256, 0)
    RUN(8, 256, 64, 0)               // the lift kernel's grid
    for (int r = 0; r < 200; ++r) {
        hipLaunchKernelGGL(empty_fat_kernel<1>, dim3(256), dim3(256), 0, 0, out, 1);
        hipLaunchKernelGGL(empty_kernel<999>, dim3(1), dim3(64), 0, 0, out, 1);
    }
    hipDeviceSynchronize();
    return 0;
}
