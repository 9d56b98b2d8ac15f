// What the MI355X memory system delivers to plain streaming kernels at the grouping kernel's traffic mix (K5, SURVEY 8d micro-benchmark: 1.11 GB read,
// 2.68 GB written per launch): read-only, write-only, copy, and one 16-byte read per 2.4 16-byte non-temporal writes, all with 16-byte accesses,
// grid-stride, 4 GiB of buffers.  hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_mix.hip -o tools/probes/hbm_mix.bin && tools/probes/hbm_mix.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k_read(const v4f *__restrict__ a, size_t n, float *sink)
{
    v4f s = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s.x + s.y + s.z + s.w == 1.2345f) *sink = s.x;
}
__global__ void k_write(v4f *__restrict__ o, size_t n)
{
    const v4f v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, o + i);
}
__global__ void k_copy(const v4f *__restrict__ a, v4f *__restrict__ o, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(a[i], o + i);
}
// 5 reads : 12 writes (1 : 2.4): every thread reads five vectors and writes twelve
__global__ void k_mix(const v4f *__restrict__ a, v4f *__restrict__ o, size_t groups)
{
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
        v4f r[5];
        const size_t stride = groups;
#pragma unroll
        for (int j = 0; j < 5; ++j) r[j] = a[g + j * stride];
#pragma unroll
        for (int j = 0; j < 12; ++j) __builtin_nontemporal_store(r[j % 5] + (float)j, o + g + j * stride);
    }
}
int main()
{
    const size_t NB = (size_t)3 << 30;                  // 3 GiB per buffer
    v4f *a, *o; float *sink;
    hipMalloc(&a, NB); hipMalloc(&o, NB); hipMalloc(&sink, 4);
    hipMemset(a, 0, NB); hipMemset(o, 0, NB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t n = NB / 16;
    auto time = [&](const char *name, double bytes, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-34s %8.3f ms  %7.1f GB/s  (%.2f of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
    };
    for (int blocks : {2048, 8192, 32768}) {
        printf("grid %d x 256\n", blocks);
        time("read only, 3 GiB", (double)NB, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, sink); });
        time("write only (non-temporal), 3 GiB", (double)NB, [&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, o, n); });
        time("copy, 3 + 3 GiB", 2.0 * NB, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, o, n); });
        const size_t groups = n / 12;
        time("5 reads : 12 writes (grouping's mix)", 17.0 * 16 * groups, [&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, a, o, groups); });
    }
    return 0;
}
