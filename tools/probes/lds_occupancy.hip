// How many 256-thread workgroups with a given dynamic LDS size co-reside on one gfx950 CU (160 KB LDS)?  Measured, not the occupancy API:
// every workgroup spins until `expect` workgroups have arrived on its CU... simpler: time a kernel that only sleeps, with 2 x CUs workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256, 2) void sleeper(int iters, float *out)
{
    extern __shared__ float smem[];
    smem[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
    if (threadIdx.x == 0 && smem[5] < 0.f) out[0] = smem[7];
}
int main()
{
    float *out; hipMalloc(&out, 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d, sharedMemPerMultiprocessor %zu, sharedMemPerBlock %zu, optin %zu\n", p.multiProcessorCount, p.sharedMemPerMultiprocessor, p.sharedMemPerBlock, (size_t)p.sharedMemPerBlockOptin);
    const int sizes[] = {32768, 65536, 73728, 77824, 79872, 80896, 81920, 83968};
    for (int s : sizes) {
        hipFuncSetAttribute((const void *)sleeper, hipFuncAttributeMaxDynamicSharedMemorySize, s);
        int nb = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sleeper, 256, s);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float t1, t2;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a); hipLaunchKernelGGL(sleeper, dim3(p.multiProcessorCount), dim3(256), s, 0, 2000, out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&t1, a, b);
            hipEventRecord(a); hipLaunchKernelGGL(sleeper, dim3(2 * p.multiProcessorCount), dim3(256), s, 0, 2000, out); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&t2, a, b);
        }
        printf("lds %6d B: occupancy API %d blocks/CU; 1 WG/CU %.3f ms, 2 WG/CU %.3f ms (ratio %.2f: ~1 = co-resident, ~2 = serialised)\n", s, nb, t1, t2, t2 / t1);
    }
    return 0;
}
