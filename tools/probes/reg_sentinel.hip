// Register-file sentinels: every lane parks NV known values in VGPRs (and every wave NS values in SGPRs), sleeps spin_us, and counts the registers that changed.
// With a co-runner on another stream this answers: does anything write into the registers of a resident wave?  (tools/probes/reg_sentinel.py)
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/reg_sentinel.hip -o tools/probes/reg_sentinel.so
#include <hip/hip_runtime.h>
constexpr int NV = 96, NS = 40;
__global__ __launch_bounds__(256) void reg_sentinel_kernel(long spin_ticks, unsigned *bad_v, unsigned *bad_s, unsigned *example)
{
    unsigned v[NV];
    unsigned s[NS];
    const unsigned seed = blockIdx.x * 4099u + threadIdx.x * 17u;
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = seed * 2654435761u + i * 40503u; asm volatile("" : "+v"(v[i])); }
#pragma unroll
    for (int i = 0; i < NS; ++i) { s[i] = __builtin_amdgcn_readfirstlane(blockIdx.x * 977u + (threadIdx.x >> 6) * 131u + i * 7919u); asm volatile("" : "+s"(s[i])); }
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(16);
    unsigned cv = 0, cs = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) { asm volatile("" : "+v"(v[i])); if (v[i] != seed * 2654435761u + i * 40503u) { ++cv; example[0] = i; example[1] = v[i]; } }
#pragma unroll
    for (int i = 0; i < NS; ++i) { asm volatile("" : "+s"(s[i])); if (s[i] != blockIdx.x * 977u + (threadIdx.x >> 6) * 131u + i * 7919u) { ++cs; example[2] = i; example[3] = s[i]; } }
    if (cv) atomicAdd(bad_v, cv);
    if (cs && (threadIdx.x & 63) == 0) atomicAdd(bad_s, cs);
}
extern "C" int reg_sentinel_launch(int blocks, long spin_us, unsigned *bad_v, unsigned *bad_s, unsigned *example, void *stream)
{
    hipLaunchKernelGGL(reg_sentinel_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, spin_us * 100, bad_v, bad_s, example);
    return (int)hipGetLastError();
}
