// Does anything running on another stream write into the LDS of a resident workgroup?  Every workgroup fills `bytes` of dynamic LDS with a pattern, waits
// `spin_us` microseconds (wall clock), re-reads it and reports the number of words that changed.  Built as a shared object for tools/probes/lds_sentinel.py:
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/lds_sentinel.hip -o tools/probes/lds_sentinel.so
#include <hip/hip_runtime.h>
__global__ void lds_sentinel_kernel(int words, long spin_ticks, unsigned *bad, unsigned *first_word, unsigned *first_val)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 20);
    __syncthreads();
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(32);
    __syncthreads();
    unsigned cnt = 0;
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
        const unsigned v = lds[i];
        if (v != (0xC0DE0000u ^ (unsigned)i ^ (blockIdx.x << 20))) {
            if (atomicAdd(&bad[blockIdx.x], 1u) == 0) { first_word[blockIdx.x] = i; first_val[blockIdx.x] = v; }
            ++cnt;
        }
    }
}
extern "C" int lds_sentinel_launch(int blocks, int bytes, long spin_us, unsigned *bad, unsigned *first_word, unsigned *first_val, void *stream)
{
    hipFuncSetAttribute((const void *)lds_sentinel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(lds_sentinel_kernel, dim3(blocks), dim3(256), bytes, (hipStream_t)stream, bytes / 4, spin_us * 100, bad, first_word, first_val);   // wall clock: 100 MHz
    return (int)hipGetLastError();
}
