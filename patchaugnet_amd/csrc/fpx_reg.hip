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
// and travel global -> registers -> a double-buffered 2 x 16 KB LDS stage shared by the workgroup's four 16-point waves (each wave
// streaming its own copy from L2 put half of the CU's L1 bandwidth on weights: 0.44-0.54 ms).  ~190 live registers: two waves per SIMD.
//
// Status (round 1, MI355X, 131 072 rows): bit-for-bit deterministic and within the 1e-4 descriptor tolerance, but 0.41 ms against
// 0.35 ms for the LDS-tiled kernel.  Even without gathers and stores the layer pipeline runs at 55 % of the MFMA rate: hipcc sinks the
// next chunk's global fetch to the end of the chunk (latency exposed before every barrier) and issues the LDS fragment reads only four
// MFMAs ahead.  One workgroup per CU (LDS-padded launch) takes 0.57 ms, two take 0.44 ms: a lone wave reaches 39 % of the MFMA rate
// and the second wave per SIMD hides only part of its stalls.  Needs explicit software pipelining (sched_group_barrier or inline asm)
// before it can replace the shipped kernel.  Tried in source and undone by the machine scheduler: a three-deep stage with the fragments of
// chunk c + 1 read into a second register set during chunk c (one wave per SIMD, 16 x float4 ahead) -- the emitted code reads every
// fragment immediately before the four MFMAs that consume it, with s_waitcnt lgkmcnt(0) in between (0.73 ms).
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

// 64 MFMAs of one weight chunk q (16 output tiles x 4 k-steps) from the LDS stage: acc[ot] += W^T fragment * h_q.s
__device__ __forceinline__ void fpx_chunk(const float4 *__restrict__ stage, const floatx4 &hq, floatx4 (&acc)[16], int lane)
{
#pragma unroll
    for (int o4 = 0; o4 < 16; o4 += 4) {                                  // four output tiles in flight: no MFMA waits on the one before it
        float4 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = stage[(o4 + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[o4 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].x, hq[0], acc[o4 + u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[o4 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].y, hq[1], acc[o4 + u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[o4 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].z, hq[2], acc[o4 + u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[o4 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u].w, hq[3], acc[o4 + u], 0, 0, 0);
    }
}

__global__ __launch_bounds__(256, 2) void fpx_reg_kernel(FpxArgs a)
{
    // weight stage: 2 x 16 KB (one chunk = the 16 output tiles of 16 input channels), shared by the four waves of the workgroup:
    // every wave streaming its own copy of the 256 KB matrix would put half of the CU's L1 bandwidth on weight traffic
    extern __shared__ __attribute__((aligned(16))) float fpx_lds[];
    float4 *stage = reinterpret_cast<float4 *>(fpx_lds);                 // [2][1024]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long nblk = gridDim.x;
    const long blk = (a.xcd_remap && (nblk & 7) == 0) ? (long)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    const long tile = blk * 4 + wave;
    const long row = tile * 16 + (lane & 15);
    const long rowc = row < a.rows ? row : a.rows - 1;                    // lanes past the end compute on a clamped row and store nothing
    const int gq = lane >> 4;

    // chunk 0 of the first layer goes to the stage while the prologue gathers
    float4 pre[4];
    {
        const float4 *src = reinterpret_cast<const float4 *>(a.wp2);
#pragma unroll
        for (int u = 0; u < 4; ++u) pre[u] = src[tid + u * 256];
    }

    // ---- first layer: bias + skip . Wskip + interpolation of the pre-multiplied coarse features, ReLU (same fmaf chain as pa_chain.h)
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
    for (int q = 0; q < 16; ++q) {
        const int c = q * 16 + gq * 4;
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
            const float4 f = g4[t][q * 4];
            v[0] = fmaf(wj[t], f.x, v[0]); v[1] = fmaf(wj[t], f.y, v[1]); v[2] = fmaf(wj[t], f.z, v[2]); v[3] = fmaf(wj[t], f.w, v[3]);
        }
        h[q] = (floatx4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);           // at most twelve 16-byte gathers in flight
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) stage[tid + u * 256] = pre[u];
    __syncthreads();

    // ---- two 256 -> 256 layers as one pipeline of 32 weight chunks: chunk c is read from stage[c & 1] while chunk c + 1 travels
    // global -> registers -> stage[(c + 1) & 1]; one barrier per chunk.  The accumulators of layer 1 (bias + ReLU) ARE layer 2's B operands.
#pragma unroll
    for (int ot = 0; ot < 16; ++ot) acc[ot] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        if (c + 1 < 32) {
            const float4 *src = reinterpret_cast<const float4 *>(c + 1 < 16 ? a.wp2 : a.wp3) + (size_t)((c + 1) & 15) * 1024;
#pragma unroll
            for (int u = 0; u < 4; ++u) pre[u] = src[tid + u * 256];
        }
        fpx_chunk(stage + (c & 1) * 1024, h[c & 15], acc, lane);
        if (c == 15) {                                                   // layer boundary: h <- relu(acc + b2), in place of the consumed inputs
#pragma unroll
            for (int ot = 0; ot < 16; ++ot) {
                const float4 bz = *reinterpret_cast<const float4 *>(a.b2 + ot * 16 + gq * 4);
                h[ot] = (floatx4){fmaxf(acc[ot][0] + bz.x, 0.f), fmaxf(acc[ot][1] + bz.y, 0.f), fmaxf(acc[ot][2] + bz.z, 0.f), fmaxf(acc[ot][3] + bz.w, 0.f)};
                acc[ot] = (floatx4){0.f, 0.f, 0.f, 0.f};
            }
        }
        if (c + 1 < 32) {
#pragma unroll
            for (int u = 0; u < 4; ++u) stage[((c + 1) & 1) * 1024 + tid + u * 256] = pre[u];
            __syncthreads();
        }
    }
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
    const size_t lds = (size_t)(32 + pad_kb) * 1024;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fpx_reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fpx_reg_kernel, dim3(nblk), dim3(256), lds, (hipStream_t)stream, a);
    PA_CHECK_LAUNCH("pa_fpx256");
    return PA_OK;
}
