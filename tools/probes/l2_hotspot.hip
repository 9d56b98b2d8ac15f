// Probe: do workgroups that stream the SAME small L2-resident window in lock-step (the chain kernels' weight stream: every workgroup reads the
// same packed weights in the same order at the same time) get less L2 -> CU bandwidth than workgroups whose positions in the window are spread?
//   hipcc --offload-arch=gfx950 -O3 -o l2_hotspot.bin l2_hotspot.hip && ./l2_hotspot.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// every wave: `iters` 1 KB chunk loads (16 B per lane), 8 in flight; chunk sequence = (start + it) % nchunks of its 64-column group stream
template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const float *__restrict__ w, int nchunks, int iters, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *base = w + (size_t)wave * nchunks * 256;          // four column-group streams, like the shared-tile chain kernels
    int start = 0;
    if (MODE == 1) start = (int)((blockIdx.x * 37u) % (unsigned)nchunks);               // spread over the window
    if (MODE == 2) start = (int)(((blockIdx.x >> 3) & 3u) * (unsigned)(nchunks / 4));    // four phases per XCD
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    u32x4 v[8];
    int c = start;
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, c * 1024, 0); c = c + 1 == nchunks ? 0 : c + 1; }
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc += v[u];
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, c * 1024, 0);
            c = c + 1 == nchunks ? 0 : c + 1;
            __builtin_amdgcn_s_sleep(1);
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
    if (acc.x == 0x12345678u) out[threadIdx.x] = 1.f;
}

template <int MODE>
float run(const float *w, int nchunks, int iters, int grid, float *out)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(stream_kernel<MODE>, dim3(grid), dim3(256), 0, 0, w, nchunks, iters, out);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(stream_kernel<MODE>, dim3(grid), dim3(256), 0, 0, w, nchunks, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main()
{
    float *w, *out;
    const int nchunks = 64;                       // 64 KB per column-group stream, 256 KB window (one 256 x 256 fp32 layer)
    hipMalloc(&w, (size_t)4 * 1024 * 1024); hipMemset(w, 0, 4 * 1024 * 1024);
    hipMalloc(&out, 4096);
    for (int grid : {256, 512, 1024, 2048}) {
        for (int iters : {64, 512}) {
            const double bytes = (double)grid * 4 * iters * 1024;
            const float t0 = run<0>(w, nchunks, iters, grid, out), t1 = run<1>(w, nchunks, iters, grid, out), t2 = run<2>(w, nchunks, iters, grid, out);
            printf("grid %4d iters %3d: lock-step %.1f us (%.2f TB/s)  spread %.1f us (%.2f TB/s)  4-phase %.1f us (%.2f TB/s)\n", grid, iters,
                   t0 * 1e3, bytes / t0 / 1e9, t1 * 1e3, bytes / t1 / 1e9, t2 * 1e3, bytes / t2 / 1e9);
        }
    }
    return 0;
}
