// Finest feature-propagation level (patch_aug_net.py:350-362 / pptnet.py FP level 0 after pa_fp_chain_premul's fold: two remaining
// 256 -> 256 layers) with fp32 operands SPLIT into fp16 pairs -- OPT-IN (model.mlp_dtype = "f32x3"), never the default.
//
// Every product a * w of the two dense layers is evaluated as   hi(a) hi(w) + lo(a) hi(w) + hi(a) lo(w),   hi(v) = fp16(v), lo(v) = fp16(v - hi(v)),
// on v_mfma_f32_16x16x32_f16 with fp32 accumulation.  fp16 x fp16 products are exact in fp32, hi + lo carries 22 of v's 24 mantissa bits, the
// dropped lo lo term is 2^-22 relative: a product is accurate to ~2^-21, against 2^-24 of the fp32 MFMA and 2^-11 of a TF32 product (what the
// reference's cuDNN 1x1 convolutions use by default on the hardware it was published on).  The weights are scaled by a power of two per layer
// (exact; undone in the epilogue) so that lo(w) stays out of fp16's subnormal range.  Three fp16 MFMAs cost 3/16 of one fp32 MFMA per product.
//
// Structure = fpx_f16.hip (weights global -> LDS once per workgroup in double-buffered slabs, activations in registers, r-ordered accumulators =
// next layer's operands), with 16-row wave tiles (the (hi, lo) operand pairs double the operand registers), one k-step x (hi, lo) x 16 column
// tiles = 32 KB per slab, and the fp32 pre-multiplied table.
#include <stdlib.h>

#include "pa_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

struct Fpx3Args {
    long rows;
    const float *g;       // (b * m_known, 256) fp32: features already multiplied by the first layer's interpolated-part weights
    const int *idx3;
    const float *w3;
    const float *skip;
    const float *wskip;   // (c1, 256) K-major
    const float *bias0;
    const half8 *wq[2];   // per layer: pa_pack_weights_f16(256, 256) of hi(W 2^s), then of lo(W 2^s) (2 x 65536 halfs)
    float inv_scale[2];   // 2^-s
    const float *b[2];
    float *out;
    int ldo, n_unknown, m_known, c1, xcd_remap;
    long long *dbg;       // profiling only (pa_chain_debug_buffer): cycle stamps of the first 512 wave tiles
    const float *x;       // PREMUL form: out[r][:] = x[r][:256] . W (one layer, no bias / ReLU), rows = rows of x
    int ldx;
};

__device__ __forceinline__ int r_ofs3(int ct, int g) { return 32 * (ct >> 1) + 8 * g + 4 * (ct & 1); }

template <int WAVES, bool PREMUL, bool DBG = false>
__global__ __launch_bounds__(WAVES * 64, 2) void fpx3_kernel(Fpx3Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fpx3_lds[];
    constexpr int SLAB = 32 * 1024;         // one k-step: (hi, lo) x 16 column tiles x 64 lanes x 16 B
    constexpr int PER = 32 / WAVES;         // 1 KB pieces per wave and slab
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mq = lane & 15, g = lane >> 4;
    float *cst = reinterpret_cast<float *>(fpx3_lds + 2 * SLAB);   // wskip[4][256], bias0[256], b2[256], b3[256]
    const long nblk = gridDim.x;
    const long blk = (a.xcd_remap && (nblk & 7) == 0) ? (long)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    unsigned *stamps = reinterpret_cast<unsigned *>(fpx3_lds + 2 * SLAB + 7 * 1024);   // profiling build only: branch-free LDS stamps
#define FPX3_STAMP(i) do { if (DBG) stamps[wave * 8 + (i)] = (unsigned)__builtin_readcyclecounter(); } while (0)
    FPX3_STAMP(0);

    // slab s = (layer s / 8, k-step s % 8) into buffer s & 1; piece p = (part p / 16: hi, lo; column tile p % 16), source lanes permuted so that
    // the fragment's columns come out in r-order (fpx_f16.hip)
    const unsigned lane_src = ((mq >> 3) * 512 + g * 16 + ((mq >> 2) & 1) * 8 + (mq & 3)) * 16u;
    auto fetch = [&](int s) {
        const char *src = reinterpret_cast<const char *>(a.wq[s >> 3]);
        const int ks = s & 7;
        unsigned char *dst = fpx3_lds + (s & 1) * SLAB;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int p = u * WAVES + wave, part = p >> 4, ct = p & 15;
            const char *piece = src + (size_t)part * 131072 + (size_t)((((ct >> 1) * 16 + ks) * 64 + 4 * (ct & 1)) * 16);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(piece + lane_src),
                                             (void __attribute__((address_space(3))) *)(dst + p * 1024), 16, 0, 0);
        }
    };

    const long row = (blk * WAVES + wave) * 16 + mq, rowc = row < a.rows ? row : a.rows - 1;
    half8 hhi[8], hlo[8];
    auto split = [&](int p, const float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const _Float16 hi = (_Float16)v[e];
            hhi[p][e] = hi;
            hlo[p][e] = (_Float16)(v[e] - (float)hi);
        }
    };
    if constexpr (PREMUL) {
        // ---- operand = the rows themselves: lane (mq, g) holds x[row][32 ks + 8 g .. + 7] ---------------------------------------------------
        const float4 *xp = reinterpret_cast<const float4 *>(a.x + (size_t)rowc * a.ldx + 8 * g);
        float4 f[8][2];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { f[ks][0] = xp[8 * ks]; f[ks][1] = xp[8 * ks + 1]; }
        __builtin_amdgcn_sched_barrier(0);
        fetch(0);
        fetch(1);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const float v[8] = {f[ks][0].x, f[ks][0].y, f[ks][0].z, f[ks][0].w, f[ks][1].x, f[ks][1].y, f[ks][1].z, f[ks][1].w};
            split(ks, v);
        }
    } else {
    // ---- h1 (hi, lo) in registers: interpolation + skip term + bias, ReLU ------------------------------------------------------------------
    const long cloud = rowc / a.n_unknown;
    int nb[3];
    float wj[3], sv[4];
#pragma unroll
    for (int t = 0; t < 3; ++t) { nb[t] = a.idx3[rowc * 3 + t]; wj[t] = a.w3[rowc * 3 + t]; }
#pragma unroll
    for (int t = 0; t < 4; ++t) sv[t] = t < a.c1 ? a.skip[rowc * a.c1 + t] : 0.f;
    constexpr int CPT = (7 * 256 + WAVES * 64 - 1) / (WAVES * 64);
    float cv[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int t = tid + u * WAVES * 64;
        const float *src = t < 1024 ? a.wskip + ((t >> 8) < a.c1 ? t : 0) : t < 1280 ? a.bias0 + (t - 1024) : t < 1536 ? a.b[0] + (t - 1280) : a.b[1] + (t < 1792 ? t - 1536 : 0);
        cv[u] = *src;
        if (t < 1024 && (t >> 8) >= a.c1) cv[u] = 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    fetch(0);
    fetch(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
        const int t = tid + u * WAVES * 64;
        if (t < 7 * 256) cst[t] = cv[u];
    }
    const float *gp[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) gp[t] = a.g + (size_t)(cloud * a.m_known + nb[t]) * 256 + 8 * g;
    __syncthreads();   // cst visible
    FPX3_STAMP(1);
#pragma unroll
    for (int q = 0; q < 8; q += 4) {   // 24 16-byte gathers in flight
        __builtin_amdgcn_sched_barrier(0);
        float4 f4[4][3][2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                f4[u][t][0] = *reinterpret_cast<const float4 *>(gp[t] + 32 * (q + u));
                f4[u][t][1] = *reinterpret_cast<const float4 *>(gp[t] + 32 * (q + u) + 4);
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = 32 * (q + u) + 8 * g;
            float v[8];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float4 bz = *reinterpret_cast<const float4 *>(cst + 1024 + c + 4 * e);
                v[4 * e] = bz.x; v[4 * e + 1] = bz.y; v[4 * e + 2] = bz.z; v[4 * e + 3] = bz.w;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 wv = *reinterpret_cast<const float4 *>(cst + t * 256 + c + 4 * e);
                    v[4 * e] = fmaf(sv[t], wv.x, v[4 * e]); v[4 * e + 1] = fmaf(sv[t], wv.y, v[4 * e + 1]);
                    v[4 * e + 2] = fmaf(sv[t], wv.z, v[4 * e + 2]); v[4 * e + 3] = fmaf(sv[t], wv.w, v[4 * e + 3]);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float4 f = f4[u][t][e];
                    v[4 * e] = fmaf(wj[t], f.x, v[4 * e]); v[4 * e + 1] = fmaf(wj[t], f.y, v[4 * e + 1]);
                    v[4 * e + 2] = fmaf(wj[t], f.z, v[4 * e + 2]); v[4 * e + 3] = fmaf(wj[t], f.w, v[4 * e + 3]);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            split(q + u, v);
        }
    }

    }

    FPX3_STAMP(2);
    // ---- the two layers: one barrier per k-step slab ---------------------------------------------------------------------------------------
    floatx4 acc[16];
    unsigned off0 = lane * 16u, off1 = SLAB + lane * 16u;
    asm volatile("" : "+v"(off0), "+v"(off1));
    const half8 *buf0 = reinterpret_cast<const half8 *>(fpx3_lds + off0), *buf1 = reinterpret_cast<const half8 *>(fpx3_lds + off1);
    constexpr int NSLAB = PREMUL ? 8 : 16;
#pragma unroll
    for (int s = 0; s < NSLAB; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s == 0) FPX3_STAMP(3);
        if (s >= 1 && s + 1 < NSLAB) fetch(s + 1);
        if ((s & 7) == 0) {
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) acc[ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
        }
        const half8 *buf = (s & 1) ? buf1 : buf0;
        const int ks = s & 7;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            const half8 whi = buf[ct * 64], wlo = buf[(16 + ct) * 64];
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, hhi[ks], acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, hlo[ks], acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, hhi[ks], acc[ct], 0, 0, 0);
        }
        // fragment pairs run four ahead of the MFMAs that consume them
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        if (s == 7) FPX3_STAMP(4);
        if (!PREMUL && s == 7) {   // h2 = relu(acc 2^-s + b2), split again, straight from the accumulators
            const float is = a.inv_scale[0];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float4 b0 = *reinterpret_cast<const float4 *>(cst + 1280 + r_ofs3(2 * p, g)), b1 = *reinterpret_cast<const float4 *>(cst + 1280 + r_ofs3(2 * p + 1, g));
                const float v[8] = {fmaxf(fmaf(acc[2 * p][0], is, b0.x), 0.f), fmaxf(fmaf(acc[2 * p][1], is, b0.y), 0.f), fmaxf(fmaf(acc[2 * p][2], is, b0.z), 0.f),
                                    fmaxf(fmaf(acc[2 * p][3], is, b0.w), 0.f), fmaxf(fmaf(acc[2 * p + 1][0], is, b1.x), 0.f), fmaxf(fmaf(acc[2 * p + 1][1], is, b1.y), 0.f),
                                    fmaxf(fmaf(acc[2 * p + 1][2], is, b1.z), 0.f), fmaxf(fmaf(acc[2 * p + 1][3], is, b1.w), 0.f)};
                split(p, v);
            }
        }
    }

    FPX3_STAMP(5);
    // ---- out = relu(acc 2^-s + b3)  (PREMUL: out = acc 2^-s) ------------------------------------------------------------------------------
    if (row < a.rows) {
        const float is = a.inv_scale[PREMUL ? 0 : 1];
        float *o = a.out + (size_t)row * a.ldo;
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            const int c = r_ofs3(ct, g);
            if (PREMUL) {
                *reinterpret_cast<float4 *>(o + c) = make_float4(acc[ct][0] * is, acc[ct][1] * is, acc[ct][2] * is, acc[ct][3] * is);
            } else {
                const float4 bz = *reinterpret_cast<const float4 *>(cst + 1536 + c);
                *reinterpret_cast<float4 *>(o + c) = make_float4(fmaxf(fmaf(acc[ct][0], is, bz.x), 0.f), fmaxf(fmaf(acc[ct][1], is, bz.y), 0.f),
                                                                 fmaxf(fmaf(acc[ct][2], is, bz.z), 0.f), fmaxf(fmaf(acc[ct][3], is, bz.w), 0.f));
            }
        }
    }
    if (DBG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FPX3_STAMP(6);
        if (blk * WAVES + wave < 512 && lane < 7) a.dbg[(blk * WAVES + wave) * 8 + lane] = (long long)stamps[wave * 8 + lane];
    }
#undef FPX3_STAMP
}

// Measured limit of this form (tools/fpx3_time.py, phase stamps): of 61.8 k cycles per 64-row workgroup 35.7 k are the layers, and those are bound by
// the LDS fragment reads -- a 16-row wave tile reads 32 KB of (hi, lo) fragments per k-step for 48 MFMAs; with two co-resident workgroups that is
// 256 KB per k-step and CU = 2 k cycles at the LDS's 128 B/clk.  Twice the rows per fragment read needs the (hi, lo) operands of 32 rows + 128
// accumulators = 256 registers before anything else, i.e. one wave per SIMD and the memory phases hidden in software; a persistent build of that
// shape (next tile's gathers consumed into a second operand set under the current tile's layers) was built three ways and dropped: both operand sets
// in registers spills 144; sixteen unrolled slab steps are 200 KB of code (instruction-cache bound: 428 us); half of the second set in a wave-private
// LDS stage with the layer loop rolled still spills 73 and runs in 214 us (all three bit-correct).  Open: 32x32x16 MFMAs do not change the count
// (the tile's operands and accumulators are the same registers); the operand set of row tile 1 read from LDS per k-step instead of held.

}  // namespace

long long *pa_chain_dbg_ptr();   // mlp_chain.hip

// pa_fp_chain_premul for the finest level's shape (c2 = 256, 1 <= c1 <= 4, two remaining 256 -> 256 layers) with every dense-layer product
// evaluated from (hi, lo) fp16 operand pairs (three fp16 MFMAs, ~2^-21 relative).  wp16x3[l]: 131072 halfs = pa_pack_weights_f16(256, 256) of
// hi(W_l 2^s_l), then of lo(W_l 2^s_l); inv_scale[l] = 2^-s_l.
PA_API int pa_fp_chain_premul_x3(int nlayers, const void *const *wp16x3, const float *inv_scale, const float *const *bias, long rows, const float *g,
                                 const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1, const float *wskip,
                                 const float *bias0, float *out, int ldo, pa_stream_t stream)
{
    PA_REQUIRE(wp16x3 && inv_scale && bias && rows > 0 && g && idx3 && w3 && skip && wskip && bias0 && out, "pa_fp_chain_premul_x3: null argument");
    if (nlayers != 2 || c2 != 256 || c1 < 1 || c1 > 4 || !wp16x3[0] || !wp16x3[1] || ldo % 4 != 0 || ((uintptr_t)out & 15) != 0 || ((uintptr_t)g & 15) != 0 ||
        n_unknown <= 0 || rows % n_unknown != 0) {
        pa_set_error("pa_fp_chain_premul_x3: only c2 = 256, 1 <= c1 <= 4 and two 256 -> 256 layers (got nlayers=%d c2=%d c1=%d)", nlayers, c2, c1);
        return PA_EUNSUPPORTED;
    }
    Fpx3Args a = {};
    a.rows = rows; a.g = g; a.idx3 = idx3; a.w3 = w3; a.skip = skip; a.wskip = wskip; a.bias0 = bias0;
    for (int l = 0; l < 2; ++l) { a.wq[l] = reinterpret_cast<const half8 *>(wp16x3[l]); a.inv_scale[l] = inv_scale[l]; a.b[l] = bias[l]; }
    a.out = out; a.ldo = ldo; a.n_unknown = n_unknown; a.m_known = m_known; a.c1 = c1;
    static const bool no_xcd = getenv("PA_CHAIN_NO_XCD_REMAP") != nullptr;
    a.xcd_remap = no_xcd ? 0 : 1;
    a.dbg = pa_chain_dbg_ptr();
    const size_t lds = 2 * 32 * 1024 + 7 * 1024 + (a.dbg ? 256 : 0);
    auto kern = a.dbg ? fpx3_kernel<4, false, true> : fpx3_kernel<4, false, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(pa_div_up(rows, 64)), dim3(256), lds, (hipStream_t)stream, a);
    PA_CHECK_LAUNCH("pa_fp_chain_premul_x3");
    return PA_OK;
}

// The pre-multiply of that level in the same arithmetic: out[r][:] = x[r][:256] . W for the (256 x 256) interpolated-part slice W of the first
// layer; wp16x3 / inv_scale as above (one layer).  fp32 output (the table pa_fp_chain_premul_x3 gathers from).
PA_API int pa_linear_x3(long rows, const float *x, int ldx, const void *wp16x3, float inv_scale, float *out, int ldo, pa_stream_t stream)
{
    PA_REQUIRE(rows > 0 && x && wp16x3 && out && ldx >= 256 && ldx % 4 == 0 && ldo >= 256 && ldo % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0,
               "pa_linear_x3: needs 256-wide 16-byte aligned rows");
    Fpx3Args a = {};
    a.rows = rows; a.x = x; a.ldx = ldx; a.wq[0] = reinterpret_cast<const half8 *>(wp16x3); a.inv_scale[0] = inv_scale; a.out = out; a.ldo = ldo;
    a.xcd_remap = 0;
    const size_t lds = 2 * 32 * 1024 + 7 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fpx3_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((fpx3_kernel<4, true>), dim3(pa_div_up(rows, 64)), dim3(256), lds, (hipStream_t)stream, a);
    PA_CHECK_LAUNCH("pa_linear_x3");
    return PA_OK;
}
