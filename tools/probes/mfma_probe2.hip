// Probe 2: the production k-loop shape (two register sets refilled two k-steps ahead, sched_group_barrier interleave) with
// (A) 16 dword weight loads per k-step (K-major weights) vs (B) 4 dwordx4 loads per k-step (fragment-major packed weights).
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifndef PWAVES
#define PWAVES 4
#endif
typedef float floatx4 __attribute__((ext_vector_type(4)));
#ifndef PRT
#define PRT 2
#endif
constexpr int RT = PRT, NC = 16, STR = 262;

template <int PACKED, int SCHED>
__global__ __launch_bounds__(64 * PWAVES) void probe(const float *__restrict__ w, int n, int ksteps, float *out, long long *cyc)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *act = lds + wave * (RT * 16 * STR);
    for (int i = lane; i < RT * 16 * STR; i += 64) act[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    floatx4 acc[RT][NC];
    for (int rt = 0; rt < RT; ++rt) for (int ct = 0; ct < NC; ++ct) acc[rt][ct] = (floatx4){0, 0, 0, 0};
    const float *ap = act + (lane & 15) * STR + (lane >> 4);
    const int last = ksteps - 1;
    float b0[NC], b1[NC], a0[RT], a1[RT];
    auto ldb = [&](float (&b)[NC], int ks) {
        if (PACKED) {
            const float4 *p = reinterpret_cast<const float4 *>(w + ((size_t)(ks & 63) * 64 + lane) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = p[q]; b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w; }
        } else {
            const float *p = w + (size_t)((ks & 63) * 4 + (lane >> 4)) * n + (lane & 15);
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) b[ct] = p[ct * 16];
        }
    };
    ldb(b0, 0); ldb(b1, 1);
    for (int rt = 0; rt < RT; ++rt) { a0[rt] = ap[rt * 16 * STR]; a1[rt] = ap[rt * 16 * STR + 4]; }
    const long long t0 = __builtin_readcyclecounter();
#define BLOCK(A, B, NX)                                                                                              \
    {                                                                                                                \
        const int nx = min((NX), last);                                                                              \
        float an[RT];                                                                                                \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) an[rt] = ap[rt * 16 * STR + (nx & 63) * 4];                \
        if (PACKED) {                                                                                                \
            const float4 *p = reinterpret_cast<const float4 *>(w + ((size_t)(nx & 63) * 64 + lane) * 16);            \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                          \
                _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                        \
                    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                \
                        acc[rt][4 * q + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rt], B[4 * q + c], acc[rt][4 * q + c], 0, 0, 0); \
                const float4 v = p[q];                                                                               \
                B[4 * q] = v.x; B[4 * q + 1] = v.y; B[4 * q + 2] = v.z; B[4 * q + 3] = v.w;                          \
                if (SCHED) { __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); } \
            }                                                                                                        \
        } else {                                                                                                     \
            const float *p = w + (size_t)((nx & 63) * 4 + (lane >> 4)) * n + (lane & 15);                            \
            _Pragma("unroll") for (int ct = 0; ct < NC; ++ct) {                                                      \
                _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                    \
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rt], B[ct], acc[rt][ct], 0, 0, 0);          \
                B[ct] = p[ct * 16];                                                                                  \
                if (SCHED) { __builtin_amdgcn_sched_group_barrier(0x008, RT, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); } \
            }                                                                                                        \
        }                                                                                                            \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) A[rt] = an[rt];                                            \
    }
    for (int ks = 0; ks + 2 <= ksteps; ks += 2) {
        BLOCK(a0, b0, ks + 2)
        BLOCK(a1, b1, ks + 3)
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int rt = 0; rt < RT; ++rt) for (int ct = 0; ct < NC; ++ct) s += acc[rt][ct][0] + acc[rt][ct][1] + acc[rt][ct][2] + acc[rt][ct][3];
    out[blockIdx.x * 64 * PWAVES + threadIdx.x] = s;
    if (lane == 0 && wave < 4) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int PACKED, int SCHED>
void run(const char *name, int blocks, int waves)
{
    const int n = 256, ksteps = 256;
    float *w, *out; long long *cyc;
    (void)hipMalloc(&w, 256 * n * 4); (void)hipMemset(w, 0, 256 * n * 4);
    (void)hipMalloc(&out, blocks * 64 * PWAVES * 4); (void)hipMalloc(&cyc, blocks * 4 * 8);
    size_t ldsb = (size_t)(PWAVES * RT * 16 * STR) * 4;
    (void)hipFuncSetAttribute((const void *)probe<PACKED, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<PACKED, SCHED>), dim3(blocks), dim3(64 * waves), ldsb, 0, w, n, ksteps, out, cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double mf = (double)ksteps * RT * NC;
    printf("%-44s blocks %4d waves %d : %.1f cyc/MFMA  %.3f ms  %.1f TFLOP/s\n", name, blocks, waves, h[0] / mf, ms,
           blocks * waves * mf * 2048.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    run<1, 1>("packed dwordx4, sched", 1024, PWAVES);
    run<1, 0>("packed dwordx4, compiler order", 1024, PWAVES);
    run<0, 1>("dword, sched", 1024, PWAVES);
    return 0;
}
