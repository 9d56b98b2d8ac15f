// EXPERIMENTAL (opt-in: PA_ENGINE_FPX_REG=1; the shipped path is the LDS-tiled chain kernel of mlp_chain.hip).
// Finest feature-propagation level with the activations held in REGISTERS between the layers (gfx950).
//
// Same computation as pa_fp_chain_premul with c1 <= 4 and two remaining 256 -> 256 layers (patch_aug_net.py:350-362 at fp0):
//     h1 = relu(interp(g) + skip . Wskip + b0);  h2 = relu(h1 W2 + b2);  out = relu(h2 W3 + b3)
// organised around the operand-swapped MFMA:  D[m = out channel][n = point] += A[m][k] * B[k][n]  with A = W^T and B = h^T.
// In that form a lane's accumulator registers acc[ct][r] hold channel 16 ct + 4 (l/16) + r of point l%16 -- which is exactly a B
// operand (k slot l/16, column l%16) of the NEXT layer for the k-step that contracts channels {16 ct + 4 g + r : g = 0..3}.  So the
// output of one layer feeds the next one straight from the accumulators: no activation tile in LDS, no transposition, and the first
// layer's output is produced in the same layout by the interpolation prologue.  The weights are packed so that the four k-steps
// r = 0..3 of one (input-channel tile q, output tile ot) pair are ONE 16-byte fragment:
//     wp[((q * 16 + ot) * 64 + l) * 4 + s] = Wt[16 q + 4 (l/16) + s][16 ot + l%16]          (pa_fpx256 in the header)
// Weight pipeline: chunk = the 16 output tiles of 16 input channels (16 KB).  Chunk C + 2 goes global -> LDS without registers
// (global_load_lds_dwordx4) into a three-deep stage shared by the workgroup's four 16-point waves; a wave reads the eight fragments of
// the next half-chunk from the stage while the 32 MFMAs of the current half-chunk issue; s_waitcnt vmcnt(0) + one barrier per chunk.
// ~200 live registers: two waves per SIMD, so one wave's gather prologue / store epilogue runs under another's MFMAs.
//
// Status (round 1, MI355X, 131 072 rows): 0.355 ms = the shipped kernel's time (0.35-0.36 ms), deterministic, within the 1e-4
// descriptor tolerance (it contracts the channels in a different order, so not the same bits).  History: each wave streaming its own
// weights from L2 0.44-0.54 ms (half of the CU's L1 bandwidth on weights); LDS stage filled through registers 0.41-0.43 ms (hipcc sank
// the fetch to the end of the chunk and, with a deeper register look-ahead, spilled it to scratch behind s_waitcnt vmcnt(0));
// sched_group_barrier patterns made it worse (the scheduler pairs every DS read with the MFMAs that consume it).  Left to do before it
// can replace the shipped kernel: the layer pipeline alone (no gathers, no stores) takes 0.32 ms = 69 % of the MFMA rate, gathers and
// stores add 0.02 ms each; 32 KB chunks with one barrier per 128 MFMAs were no faster (0.368 ms).  Open: why two waves per SIMD fed from
// LDS a half-chunk ahead stop at 69 %, the 22 spilled registers of the two-waves build, an LDS-staged epilogue.
#include "pa_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

struct FpxArgs {
    long rows;
    const float *g;       // (b * m_known, 256) features already multiplied by the first layer's interpolated-part weights
    const int *idx3;      // (rows, 3) neighbour indices inside the cloud
    const float *w3;      // (rows, 3)
    const float *skip;    // (rows, c1)
    const float *wskip;   // (c1, 256) K-major
    const float *bias0;   // (256)
    const float *wp2, *b2, *wp3, *b3;
    float *out;
    int ldo, n_unknown, m_known, c1, xcd_remap;
};

// 64 MFMAs of weight chunk c in two halves of eight output tiles.  Half A runs from wa[8] (read during the previous chunk) while the
// eight fragments of half B are read from this chunk's stage into wb[8]; half B runs while half A of chunk c + 1 is read into wa[8].
// Eight independent accumulators per half: no MFMA waits on its predecessor; a fragment has >= 32 MFMAs to arrive.
__device__ __forceinline__ void fpx_chunk(const float4 *__restrict__ cur_stage, const float4 *__restrict__ next_stage, bool have_next,
                                          const floatx4 &hq, float4 (&wa)[8], float4 (&wb)[8], floatx4 (&acc)[16], int lane)
{
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
        wb[ot] = cur_stage[(8 + ot) * 64 + lane];
        acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ot].x, hq[0], acc[ot], 0, 0, 0);
    }
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ot].y, hq[1], acc[ot], 0, 0, 0);
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ot].z, hq[2], acc[ot], 0, 0, 0);
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ot].w, hq[3], acc[ot], 0, 0, 0);
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
        if (have_next) wa[ot] = next_stage[ot * 64 + lane];
        acc[8 + ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[ot].x, hq[0], acc[8 + ot], 0, 0, 0);
    }
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) acc[8 + ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[ot].y, hq[1], acc[8 + ot], 0, 0, 0);
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) acc[8 + ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[ot].z, hq[2], acc[8 + ot], 0, 0, 0);
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) acc[8 + ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[ot].w, hq[3], acc[8 + ot], 0, 0, 0);
}

// global -> LDS without registers (global_load_lds_dwordx4: lane l of the wave writes 16 bytes at M0 + 16 l)
__device__ __forceinline__ void fpx_fetch_chunk(const float *wp, int chunk, float4 *stage_buf, int tid, int lane, int wave)
{
    const float4 *src = reinterpret_cast<const float4 *>(wp) + (size_t)chunk * 1024 + tid;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + u * 256),
                                         (void __attribute__((address_space(3))) *)(stage_buf + u * 256 + wave * 64), 16, 0, 0);
}

// iteration C of the 32-chunk pipeline as a compile-time recursion (every register array index must be a constant).  Chunk C + 2 is
// fetched straight into stage[(C + 2) % 3] while chunk C (second half) and chunk C + 1 (first half) are read; vmcnt(0) + barrier at the end.
template <int C>
__device__ __forceinline__ void fpx_pipeline(const FpxArgs &a, float4 *stage, floatx4 (&h)[16], float4 (&wa)[8], float4 (&wb)[8], floatx4 (&acc)[16],
                                             int tid, int lane, int wave, int gq)
{
    if constexpr (C < 32) {
        if constexpr (C + 2 < 32) fpx_fetch_chunk(C + 2 < 16 ? a.wp2 : a.wp3, (C + 2) & 15, stage + ((C + 2) % 3) * 1024, tid, lane, wave);
        fpx_chunk(stage + (C % 3) * 1024, stage + ((C + 1) % 3) * 1024, C + 1 < 32, h[C & 15], wa, wb, acc, lane);
        if constexpr (C == 15) {
#pragma unroll
            for (int ot = 0; ot < 16; ++ot) {
                const float4 bz = *reinterpret_cast<const float4 *>(a.b2 + ot * 16 + gq * 4);
                h[ot] = (floatx4){fmaxf(acc[ot][0] + bz.x, 0.f), fmaxf(acc[ot][1] + bz.y, 0.f), fmaxf(acc[ot][2] + bz.z, 0.f), fmaxf(acc[ot][3] + bz.w, 0.f)};
                acc[ot] = (floatx4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if constexpr (C + 2 < 32) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        fpx_pipeline<C + 1>(a, stage, h, wa, wb, acc, tid, lane, wave, gq);
    }
}

__global__ __launch_bounds__(256, 2) void fpx_reg_kernel(FpxArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float fpx_lds[];
    float4 *stage = reinterpret_cast<float4 *>(fpx_lds);                 // [3][1024]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long nblk = gridDim.x;
    const long blk = (a.xcd_remap && (nblk & 7) == 0) ? (long)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    const long tile = blk * 4 + wave;
    const long row = tile * 16 + (lane & 15);
    const long rowc = row < a.rows ? row : a.rows - 1;
    const int gq = lane >> 4;
    fpx_fetch_chunk(a.wp2, 0, stage, tid, lane, wave);
    fpx_fetch_chunk(a.wp2, 1, stage + 1024, tid, lane, wave);
    const long cloud = rowc / a.n_unknown;
    const float4 *g4[3];
    float wj[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        g4[t] = reinterpret_cast<const float4 *>(a.g + (size_t)(cloud * a.m_known + a.idx3[rowc * 3 + t]) * 256) + gq;
        wj[t] = a.w3[rowc * 3 + t];
    }
    float sv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) sv[t] = t < a.c1 ? a.skip[rowc * a.c1 + t] : 0.f;
    floatx4 h[16], acc[16];
#pragma unroll
    for (int q4 = 0; q4 < 16; q4 += 4) {                                 // twelve gathers issued, then consumed
        float4 f[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 3; ++t) f[u][t] = g4[t][(q4 + u) * 4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = (q4 + u) * 16 + gq * 4;
            const float4 bz = *reinterpret_cast<const float4 *>(a.bias0 + c);
            float v[4] = {bz.x, bz.y, bz.z, bz.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < a.c1) {
                    const float4 wv = *reinterpret_cast<const float4 *>(a.wskip + (size_t)t * 256 + c);
                    v[0] = fmaf(sv[t], wv.x, v[0]); v[1] = fmaf(sv[t], wv.y, v[1]); v[2] = fmaf(sv[t], wv.z, v[2]); v[3] = fmaf(sv[t], wv.w, v[3]);
                }
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                v[0] = fmaf(wj[t], f[u][t].x, v[0]); v[1] = fmaf(wj[t], f[u][t].y, v[1]); v[2] = fmaf(wj[t], f[u][t].z, v[2]); v[3] = fmaf(wj[t], f[u][t].w, v[3]);
            }
            h[q4 + u] = (floatx4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float4 wa[8], wb[8];
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) wa[ot] = stage[ot * 64 + lane];
#pragma unroll
    for (int ot = 0; ot < 16; ++ot) acc[ot] = (floatx4){0.f, 0.f, 0.f, 0.f};
    fpx_pipeline<0>(a, stage, h, wa, wb, acc, tid, lane, wave, gq);
    if (row < a.rows) {
        float *o = a.out + (size_t)row * a.ldo + gq * 4;
#pragma unroll
        for (int ot = 0; ot < 16; ++ot) {
            const float4 bz = *reinterpret_cast<const float4 *>(a.b3 + ot * 16 + gq * 4);
            *reinterpret_cast<float4 *>(o + ot * 16) =
                make_float4(fmaxf(acc[ot][0] + bz.x, 0.f), fmaxf(acc[ot][1] + bz.y, 0.f), fmaxf(acc[ot][2] + bz.z, 0.f), fmaxf(acc[ot][3] + bz.w, 0.f));
        }
    }
}

}  // namespace

PA_API int pa_fpx256(long rows, const float *g, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c1,
                     const float *wskip, const float *bias0, const float *wp2, const float *b2, const float *wp3, const float *b3,
                     float *out, int ldo, pa_stream_t stream)
{
    PA_REQUIRE(rows > 0 && g && idx3 && w3 && skip && wskip && bias0 && wp2 && b2 && wp3 && b3 && out, "pa_fpx256: null argument");
    PA_REQUIRE(c1 >= 1 && c1 <= 4 && n_unknown > 0 && m_known > 0 && rows % n_unknown == 0, "pa_fpx256: needs 1 <= c1 <= 4 and rows = b * n_unknown");
    PA_REQUIRE(ldo % 4 == 0 && ((uintptr_t)out & 15) == 0, "pa_fpx256: out rows must be 16-byte aligned");
    FpxArgs a;
    a.rows = rows; a.g = g; a.idx3 = idx3; a.w3 = w3; a.skip = skip; a.wskip = wskip; a.bias0 = bias0;
    a.wp2 = wp2; a.b2 = b2; a.wp3 = wp3; a.b3 = b3; a.out = out; a.ldo = ldo; a.n_unknown = n_unknown; a.m_known = m_known; a.c1 = c1;
    static const bool no_xcd = getenv("PA_CHAIN_NO_XCD_REMAP") != nullptr;
    a.xcd_remap = no_xcd ? 0 : 1;
    static const int pad_kb = getenv("PA_FPX_LDS_PAD") ? atoi(getenv("PA_FPX_LDS_PAD")) : 0;   // tuning knob: extra LDS per workgroup steers co-residency
    const long nblk = (rows + 63) / 64;
    const size_t lds = (size_t)(48 + pad_kb) * 1024;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fpx_reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fpx_reg_kernel, dim3(nblk), dim3(256), lds, (hipStream_t)stream, a);
    PA_CHECK_LAUNCH("pa_fpx256");
    return PA_OK;
}
