// Micro-probe: cycles per v_mfma_f32_16x16x4_f32 in the chain kernel's loop shape (RT x NC accumulators), with operands from
// (0) registers only, (1) + A from LDS, (2) + B from global/L2 (dword loads, 64-B segments), (3) B from LDS instead.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE, int RT, int NC>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ w, int n, int ksteps, float *out, long long *cyc)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *act = lds + wave * (RT * 16 * 262);
    for (int i = lane; i < RT * 16 * 262; i += 64) act[i] = (float)(i % 7) * 0.01f;
    float *wl = lds + 4 * (RT * 16 * 262);     // MODE 3: 8 k-rows x 256 + pad weights in LDS
    if (MODE == 3) for (int i = threadIdx.x; i < 8 * 272; i += 256) wl[i] = 0.001f * i;
    __syncthreads();
    floatx4 acc[RT][NC];
    for (int rt = 0; rt < RT; ++rt) for (int ct = 0; ct < NC; ++ct) acc[rt][ct] = (floatx4){0, 0, 0, 0};
    const float *ap = act + (lane & 15) * 262 + (lane >> 4);
    const float *wp = w + (size_t)(lane >> 4) * n + (lane & 15);
    float a[RT], b[NC];
    for (int rt = 0; rt < RT; ++rt) a[rt] = 1.0f + rt;
    for (int ct = 0; ct < NC; ++ct) b[ct] = 0.5f + ct;
    const long long t0 = __builtin_readcyclecounter();
    for (int ks = 0; ks < ksteps; ++ks) {
        if (MODE >= 1) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a[rt] = ap[rt * 16 * 262 + (ks & 63) * 4];
        }
        if (MODE == 2) {
            const float *wn = wp + (size_t)(ks & 63) * 4 * n;
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) b[ct] = wn[ct * 16];
        }
        if (MODE == 3) {
            const float *wn = wl + ((ks & 1) * 4 + (lane >> 4)) * 272 + (lane & 15);
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) b[ct] = wn[ct * 16];
        }
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt], b[ct], acc[rt][ct], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int rt = 0; rt < RT; ++rt) for (int ct = 0; ct < NC; ++ct) s += acc[rt][ct][0] + acc[rt][ct][1] + acc[rt][ct][2] + acc[rt][ct][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE, int RT, int NC>
void run(const char *name, int blocks, int waves)
{
    const int n = 256, ksteps = 256;
    float *w, *out; long long *cyc;
    hipMalloc(&w, 256 * n * 4); hipMemset(w, 0, 256 * n * 4);
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
    size_t ldsb = (size_t)(4 * RT * 16 * 262 + 8 * 272) * 4;
    hipFuncSetAttribute((const void *)probe<MODE, RT, NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<MODE, RT, NC>), dim3(blocks), dim3(64 * waves), ldsb, 0, w, n, ksteps, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double mf = (double)ksteps * RT * NC;
    printf("%-34s blocks %4d waves/blk %d : %.1f cyc/MFMA (wave0)  %.3f ms  %.1f TFLOP/s\n", name, blocks, waves, h[0] / mf, ms,
           blocks * waves * mf * 2048.0 / (ms * 1e-3) / 1e12);
    hipFree(w); hipFree(out); hipFree(cyc);
}

int main()
{
    run<0, 2, 16>("regs only RT2 NC16", 1024, 4);
    run<1, 2, 16>("A from LDS RT2 NC16", 1024, 4);
    run<2, 2, 16>("A LDS + B global RT2 NC16", 1024, 4);
    run<3, 2, 16>("A LDS + B LDS RT2 NC16", 1024, 4);
    run<2, 8, 4>("A LDS + B global RT8 NC4", 1024, 4);
    run<0, 2, 16>("regs only, 1 block/1 wave", 1, 1);
    run<2, 2, 16>("A LDS + B global, 1 block 1 wave", 1, 1);
    run<2, 2, 16>("A LDS + B global, 256 blocks", 256, 4);
    return 0;
}
