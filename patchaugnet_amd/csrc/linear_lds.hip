// One dense layer on point-major rows with a 128-column half of the weight matrix RESIDENT IN LDS:  out = [residual +] act(x Wt + bias).
//
// The layers this serves are the pre-multiplies of the feature-propagation levels (engine.py fp_premul: the coarse level's features times the
// first layer's interpolated-part weights, utils/model_util/pt_util.py:16-41 folded through pointops.interpolation) and the 1x1 convolutions
// around PPT-Net's attention (pptnet.py:261-282): a few thousand to a few ten thousand rows, 256 -> 256 / 512.  The chain kernel (mlp_chain.hip)
// runs them with wave-private 16-row tiles, every tile streaming the whole 256 KB weight matrix out of L2 -- 512 MB per launch at 32 768 rows,
// which is what bounds it (11 TB/s of L2 -> L1 reads, 54 % MFMA-busy).  Here a workgroup is pinned to one 128-column half, loads that half ONCE
// (128 KB, MFMA fragment order, 16-byte reads), and its eight wavefronts walk the rows: the activation tile of a wave lives in registers (the
// MFMA operand layout is read straight from global memory, the next tile's 64 registers are in flight under the current tile's MFMAs), so LDS
// holds nothing but weights and the L2 sees each weight once per workgroup.  Same arithmetic as the chain kernel (exact fp32 MFMA, k ascending,
// then bias, activation, residual): bit-identical results (tests/test_gpu_chain.py).
#include <stdlib.h>

#include "pa_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int LL_COLS = 128, LL_CT = LL_COLS / 16;

// 4 x 4 transpose across the lanes {l, l + 16, l + 32, l + 48} and the four registers of v (gfx950 lane-swap instructions, no LDS): afterwards
// component r of lane (l % 16, q = l / 16) holds what component q of lane (l % 16, r) held.
__device__ __forceinline__ void transpose4_rows(float4 &v)
{
    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
    uint2v t;
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.x), __float_as_uint(v.z), false, false);      // bit 1 of (row, register)
    v.x = __uint_as_float(t[0]); v.z = __uint_as_float(t[1]);
    t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.y), __float_as_uint(v.w), false, false);
    v.y = __uint_as_float(t[0]); v.w = __uint_as_float(t[1]);
    t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v.x), __float_as_uint(v.y), false, false);      // bit 0
    v.x = __uint_as_float(t[0]); v.y = __uint_as_float(t[1]);
    t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v.z), __float_as_uint(v.w), false, false);
    v.z = __uint_as_float(t[0]); v.w = __uint_as_float(t[1]);
}

// KS = k / 4 k-steps (compile time: the activation tile is KS registers per lane)
template <int KS, int LL_WAVES>
__global__ __launch_bounds__(LL_WAVES * 64) void linear_lds_kernel(long rows, int n, const float *__restrict__ x, int ldx, const float *__restrict__ wt,
                                                                     const float *__restrict__ bias, int relu, const float *__restrict__ residual, int ldr,
                                                                     float *__restrict__ out, int ldo, long tiles_per_group)
{
    extern __shared__ __attribute__((aligned(16))) float wf[];      // [KS][2][64 lanes][4]: lane's fragments of column tiles 4 c4 .. 4 c4 + 3 of k-step s
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    const int nbase = blockIdx.y * LL_COLS;
    const long ntiles = (rows + 15) / 16;
    const long t_begin = blockIdx.x * tiles_per_group, t_end = min(t_begin + tiles_per_group, ntiles);
    long tile = t_begin + wave;
    // Activations: the MFMA B operand of k-step s wants channel 4 s + l / 16 of row 16 tile + l % 16 in lane l.  Read that way (4 bytes a lane, 16
    // cache lines an instruction, 64 instructions a tile) the eight waves' lines do not survive in L1 from one k-step to the next and the kernel runs
    // at the L2's bandwidth (measured: 13 TB/s of L2 -> L1 for 0.25 TB/s of useful reads).  So a lane reads 16 bytes -- channels 16 j + 4 (l / 16) .. + 3,
    // 16 instructions a tile, issued in pairs that finish a 128-byte line -- and the 4 x 4 block is transposed across the four lane rows in registers.
    // Rows past the end read the last row (never stored).
    float4 xa[KS / 4], xn[KS / 4];
    {
        const float *p = x + min(tile * 16 + li, rows - 1) * ldx + 4 * lq;     // the wave's first tile: in flight under the weight load
#pragma unroll
        for (int j = 0; j < KS / 4; ++j) xa[j] = *reinterpret_cast<const float4 *>(p + 16 * j);
    }
    // ---- this half of the weights, once: wf[((s * 2 + c4) * 64 + l) * 4 + j] = Wt[4 s + l / 16][nbase + 64 c4 + 16 j + l % 16].  A thread gathers the four
    // values of one LDS float4 (4-byte reads, 16 lanes = 64 contiguous bytes; one conflict-free 16-byte LDS write), all KS / 4 float4s of a thread in
    // flight together: one L2 round trip, not KS / 4 of them
    {
        constexpr int NQ = KS * 2 * 64, NT = LL_WAVES * 64, NIT = (NQ + NT - 1) / NT;
        float4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int q = min(it * NT + tid, NQ - 1), l = q & 63, c4 = (q >> 6) & 1, s = q >> 7;
            const float *src = wt + (size_t)(4 * s + (l >> 4)) * n + nbase + 64 * c4 + (l & 15);
            v[it] = make_float4(src[0], src[16], src[32], src[48]);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (NQ % NT == 0 || it * NT + tid < NQ) reinterpret_cast<float4 *>(wf)[it * NT + tid] = v[it];
    }
    __syncthreads();
    if (tile >= t_end) return;                        // wave-uniform; no workgroup barrier below
    const float floor_v = relu ? 0.f : -INFINITY;
    const float4 *wl = reinterpret_cast<const float4 *>(wf) + lane;
    for (; tile < t_end; tile += LL_WAVES) {
        const float *pn = x + min((tile + LL_WAVES) * 16 + li, rows - 1) * ldx + 4 * lq;      // the wave's next tile (clamped: read, never used, past the end)
#pragma unroll
        for (int j = 0; j < KS / 4; ++j) transpose4_rows(xa[j]);
        floatx4 acc[LL_CT];
#pragma unroll
        for (int ct = 0; ct < LL_CT; ++ct) acc[ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
        // per k-step: the NEXT k-step's two weight reads (and, every eighth k-step, two loads of the next tile's activations) are issued in front of
        // this k-step's eight MFMAs, and the scheduler is pinned to that order (left alone it hoists all 128 LDS reads and spills)
        float4 w0 = wl[0], w1 = wl[64];
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float4 n0 = w0, n1 = w1;
            if (s + 1 < KS) {
                n0 = wl[((s + 1) * 2 + 0) * 64];
                n1 = wl[((s + 1) * 2 + 1) * 64];
            }
            if (s % 8 == 0) {
                xn[s / 4] = *reinterpret_cast<const float4 *>(pn + 4 * s);
                xn[s / 4 + 1] = *reinterpret_cast<const float4 *>(pn + 4 * s + 16);
            }
            const float xs = s % 4 == 0 ? xa[s / 4].x : s % 4 == 1 ? xa[s / 4].y : s % 4 == 2 ? xa[s / 4].z : xa[s / 4].w;
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, xs, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, xs, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, xs, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, xs, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, xs, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, xs, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, xs, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, xs, acc[7], 0, 0, 0);
            if (s + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            if (s % 8 == 0) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            w0 = n0;
            w1 = n1;
        }
        // the weight fragment was the MFMA's A operand: a lane holds channels nbase + 16 ct + 4 (l / 16) + r of row 16 tile + l % 16 -> 16-byte stores
        const long row = tile * 16 + li;
        if (row < rows) {
#pragma unroll
            for (int ct = 0; ct < LL_CT; ++ct) {
                const int col = nbase + 16 * ct + 4 * lq;
                const float4 bs = *reinterpret_cast<const float4 *>(bias + col);
                float4 v = make_float4(fmaxf(acc[ct][0] + bs.x, floor_v), fmaxf(acc[ct][1] + bs.y, floor_v), fmaxf(acc[ct][2] + bs.z, floor_v),
                                       fmaxf(acc[ct][3] + bs.w, floor_v));
                if (residual) {
                    const float4 rr = *reinterpret_cast<const float4 *>(residual + row * ldr + col);
                    v.x = rr.x + v.x; v.y = rr.y + v.y; v.z = rr.z + v.z; v.w = rr.w + v.w;
                }
                *reinterpret_cast<float4 *>(out + row * ldo + col) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < KS / 4; ++j) xa[j] = xn[j];
    }
}

}  // namespace

static int g_linear_lds = -1;
// test / A/B switch: 1 = wherever the shape applies, 0 = never, -1 = the default rule (PA_LINEAR_NO_LDS=1 turns it off)
PA_API void pa_linear_lds_enable(int on) { g_linear_lds = on; }

// 1 when the LDS-resident kernel took the call (pa_linear, mlp_chain.hip, asks first), 0 when the shape is not one it is built for.
int pa_linear_lds_try(long rows, int k, int n, const float *x, int ldx, const float *wt, const float *bias, int relu, const float *residual, int ldr,
                      float *out, int ldo, hipStream_t st)
{
    static const bool off = getenv("PA_LINEAR_NO_LDS") != nullptr;
    static const long min_rows = getenv("PA_LINEAR_LDS_MIN_ROWS") ? atol(getenv("PA_LINEAR_LDS_MIN_ROWS")) : 8192;     // below: the chain kernel's column-sliced tiling is as fast (4096 rows: 14.5 vs 13.7 us)
    if (g_linear_lds == 0 || (g_linear_lds < 0 && (off || rows < min_rows))) return 0;
    if (k != 256 || ldx % 4 || ((uintptr_t)x & 15) || n % LL_COLS != 0 || n <= 0 || rows <= 0 || !x || !wt || !bias || !out) return 0;
    if (ldo % 4 || ((uintptr_t)out & 15) || ((uintptr_t)bias & 15) || (residual && (ldr % 4 || ((uintptr_t)residual & 15)))) return 0;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    static const int waves = getenv("PA_LINEAR_LDS_WAVES") ? atoi(getenv("PA_LINEAR_LDS_WAVES")) : 8;       // 8 = 2 wavefronts per SIMD; 12 = 3 (168 VGPRs) measured no faster
    const int halves = n / LL_COLS;
    const long ntiles = (rows + 15) / 16;
    const long cap = cus / halves > 0 ? cus / halves : 1;             // one workgroup per CU; few rows: spread over the CUs (a 16-row tile is
    long groups = ntiles < cap ? ntiles : cap;                         // 7 us of MFMA on its SIMD, the 128 KB weight load 2-3 us) rather than fill a workgroup's waves
    const long tpg = (ntiles + groups - 1) / groups;
    groups = (ntiles + tpg - 1) / tpg;
    constexpr int KS = 64;
    const size_t lds = (size_t)KS * 2 * 64 * 4 * sizeof(float);        // 128 KB
    auto launch = [&](auto kern, int nw) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)groups, halves), dim3(nw * 64), lds, st, rows, n, x, ldx, wt, bias, relu, residual, ldr, out, ldo, tpg);
    };
    if (waves == 8) launch(&linear_lds_kernel<KS, 8>, 8);
    else launch(&linear_lds_kernel<KS, 12>, 12);
    return 1;
}
