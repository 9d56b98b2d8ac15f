// Wall-clock fp32 MFMA ceiling of the whole chip: every SIMD of every CU issues independent v_mfma_f32_16x16x4_f32 from registers
// (WPS waves per SIMD), timed with HIP events.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void peak(int iters, float *out)
{
    floatx4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (floatx4){0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}
int main()
{
    float *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 2; ++wps)
        for (int iters : {2000, 20000, 100000}) {
            const int blocks = 256 * wps;
            peak<<<blocks, 256>>>(100, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            peak<<<blocks, 256>>>(iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * iters * 64 * 2048.0;
            printf("waves/SIMD %d  iters %6d  %.3f ms  %.1f TFLOP/s fp32 MFMA  (%.1f %% of 157.3)\n", wps, iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
        }
    return 0;
}
