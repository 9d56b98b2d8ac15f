// NetVLAD pyramid + adaptive feature aggregator (APFA) on the MFMA pipe (gfx950, exact fp32 MFMA).
//
// Reference semantics (eval mode):
//   NetVLADBase.forward      place_recognition/patch_aug_net/models/loupe.py:191-222
//        A = softmax_K( BN1d_K( X Wc ) );  a_sum = sum_n A;  V = (A^T X)^T - a_sum * W2;  V /= max(||V||_2 over C, 1e-12)
//   MLPAttentionLayer        loupe.py:24-41   w = softmax_k( max_o ( Watt x )[o,k] );  x <- relu(x + x*w)
//   AdaptiveFeatureAggregator loupe.py:57-66  fc(21504 -> 256) + BN1d + L2 normalise
//
// MI355X design.  The reference runs ~14 torch kernels per scale and reads the (B, N, 256) feature map twice (once
// transposed).  Here X is already point-major (the MLP chain writes it that way); one kernel per scale stages 64-row
// tiles of X in LDS ONCE and uses them for both contractions: the assignment GEMM (A operand) and the aggregation
// GEMM A^T X (B operand), with the soft-max done in registers on the MFMA C/D layout in between.  A^T X is
// accumulated in registers across a workgroup's row tiles; per-workgroup partials are reduced by a tiny finalize
// kernel that also subtracts a_sum*W2, intra-normalises and writes the concatenated (B, 256, sum K) layout.
// The aggregator's 22 MB FC weight is streamed exactly once by a split-K MFMA kernel over ~170 workgroups.
#include <string.h>

#include "pa_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

// all-reduce over the 16 lanes of a DPP row with row rotations (one VALU op per step; __shfl_xor would be an LDS round trip each)
#define PA_DPP_ROW_ROR(n) (0x120 + (n))
template <int N>
__device__ __forceinline__ float row_ror(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), PA_DPP_ROW_ROR(N), 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v)
{
    v = fmaxf(v, row_ror<8>(v)); v = fmaxf(v, row_ror<4>(v)); v = fmaxf(v, row_ror<2>(v)); v = fmaxf(v, row_ror<1>(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v)
{
    v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
    return v;
}

constexpr int VC = 256;         // feature channels (every shipped config; checked on the host)
constexpr int VROWS = 64;       // rows of X per LDS tile

// LDS layout of the NetVLAD kernel.  The X tile (64 rows x 256 channels) is stored UNPADDED with whole float4 groups XOR-swizzled per row,
//     X[row][c] at Xs[row * 256 + (c ^ 4 (row % 16))],
// so that the tile (64 KB) plus the assignment tile (16 KB at K = 64) is exactly half of the CU's 160 KB: two workgroups co-reside and one's
// soft-max / barriers / refill run under the other's MFMAs.  Every MFMA fragment is fetched with 16-byte LDS reads (ds_read_b32 reaches a
// fifth of its rate from one or two waves per SIMD, MI355X_MICROARCH.md section LDS); the contraction / tile indices are permuted to fit:
//   assignment GEMM  (A operand = X rows): lane (i, q) reads X[16 w + i][16 j + 4 q .. + 3]; its four values feed k-steps 4 j .. 4 j + 3, so
//       k-step s = 4 j + e contracts channels {16 j + 4 q + e}: the packed weights are permuted accordingly (pa_netvlad_pack_weights);
//   aggregation GEMM (A operand = X columns, B operand = assignments): lane (i, q) reads X[4 ks + q][64 w + 4 i .. + 3] and, at K = 64,
//       A[4 ks + q][4 i .. + 3]: MFMA tile e has row i <-> channel 64 w + 4 i + e and column i <-> cluster 4 i + e.
// A ds_read_b128 is served in groups of 16 lanes, {i 0-3, 12-15 of q} + {i 4-11 of q + 1}; (4 j + q) ^ i (resp. i ^ (4 ks + q) % 16) takes 16
// distinct values over such a group: conflict-free.  The assignment tile uses the same swizzle, A[row][k] at As[row * 64 + (k ^ 4 (row % 16))];
// its 4-byte writes (rows 16 w + 4 q + r, clusters 16 ct + i) land on (16 ct + i) ^ (16 q + 4 r): 32 distinct banks per 32 lanes.
__device__ __forceinline__ int vlad_swz(int row) { return (row & 15) << 2; }

// ------------------------------------------------------------------------------------------------ NetVLAD accumulate
// grid (chunks, B); each workgroup walks rows [chunk*rows_per_wg, +rows_per_wg) of cloud b in 64-row tiles.
// part[b][chunk][kp][VC] = sum over its rows of A[row][k] * X[row][c];   asum_part[b][chunk][kp] = sum of A[row][k]
#ifdef PA_VLAD_DEBUG
__device__ long long vlad_stamps[16];
#define VSTAMP(i) do { if (KT == 4 && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && r0 == row_begin + VROWS) vlad_stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define VSTAMPK(i) do { if (KT == 4 && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) vlad_stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define VSTAMP(i) do { } while (0)
#define VSTAMPK(i) do { } while (0)
#endif

#ifndef PA_VLAD_WGS
#define PA_VLAD_WGS 2           // workgroups per CU at the finest scale
#endif

// channel contracted by k-step s (0..63) in k-group q (0..3) of the assignment GEMM
__host__ __device__ __forceinline__ int vlad_chan(int s, int q) { return 16 * (s >> 2) + 4 * q + (s & 3); }

template <int KT>
__global__ __launch_bounds__(256, KT == 4 ? PA_VLAD_WGS : 1) void vlad_accum_kernel(int n, int k_true, int rows_per_wg, const float *__restrict__ x_all,
                                                           const float *__restrict__ wc_t,   // [VC][16*KT] K-major, BN folded
                                                           const float *__restrict__ wc_p,   // optional pa_netvlad_pack_weights(wc_t) (KT == 4), or null
                                                           const float *__restrict__ bias,   // [16*KT]
                                                           float *__restrict__ part, float *__restrict__ asum_part)
{
    constexpr int KP = 16 * KT;
    constexpr bool SWZ_A = KT == 4;        // unpadded, swizzled assignment tile read 16 bytes at a time (needs all 64 columns)
    constexpr int AS = SWZ_A ? KP : KP + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Xs = smem;                      // [VROWS][VC], float4 groups swizzled (vlad_swz)
    float *As = smem + VROWS * VC;         // [VROWS][AS]
    float *red = As;                       // [4][KP] cross-wave a_sum, after the last tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int row_begin = chunk * rows_per_wg;
    const int row_end = min(row_begin + rows_per_wg, n);
    const float *x = x_all + (size_t)b * n * VC;

    floatx4 acc2[KT][4];                   // (A^T X)^T: [cluster tile rt][channel tile e]; rows of an MFMA tile = channels, columns = clusters
#pragma unroll
    for (int rt = 0; rt < KT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc2[rt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float asum[KT];
#pragma unroll
    for (int ct = 0; ct < KT; ++ct) asum[ct] = 0.f;
    float bia[KT];
#pragma unroll
    for (int ct = 0; ct < KT; ++ct) bia[ct] = bias[ct * 16 + li];

    // X tiles are prefetched one tile ahead into registers (16 float4 per thread): the global loads of tile t+1 are in flight
    // during the two GEMM phases of tile t, and LDS is refilled right after the barrier that ends phase 4.
    constexpr int PF = VROWS * (VC / 4) / 256;
    float4 pre[PF];
    // Through a buffer descriptor over this workgroup's rows: one loop-invariant 32-bit lane offset and no 64-bit address math (flat
    // addressing kept 16 address pairs alive across the tile, which spilled at two waves per SIMD); rows past row_end read as zero.
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (size_t)row_begin * VC), 0,
                                                                         (row_end - row_begin) * VC * 4, 0x00020000);
    auto fetch_part = [&](int r0, int u0, int u1) {
        const unsigned base = (unsigned)(r0 - row_begin) * (VC * 4) + (unsigned)tid * 16u;       // the range check covers voffset + imm
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (u < u0 || u >= u1) continue;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrs, base + (unsigned)u * 4096u, 0, 0);
            pre[u] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
    };
    auto fetch = [&](int r0) { fetch_part(r0, 0, PF); };
    VSTAMPK(9);
    fetch(row_begin);
    // A fragments of the assignment GEMM: byte address of X[16 w + i][16 j + 4 q] = arow * 1024 + ((64 j + 16 q) ^ 16 i) = abase ^ (64 j)
    const char *xs_bytes = reinterpret_cast<const char *>(Xs);
    const unsigned abase = (unsigned)(wave * 16 + li) * (VC * 4) + (unsigned)((16 * lq) ^ (16 * li));
    auto lda = [&](int j) { return *reinterpret_cast<const float4 *>(xs_bytes + (abase ^ ((unsigned)j << 6))); };
    for (int r0 = row_begin; r0 < row_end; r0 += VROWS) {
        const int cnt = min(VROWS, row_end - r0);
        if (r0 == row_begin) VSTAMPK(10);
        VSTAMP(0);
        // 1. registers -> LDS (the swizzle moves whole float4 groups: 16-byte stores), then start fetching the next tile
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int q = tid + u * 256, r = q / (VC / 4), part4 = q - r * (VC / 4);
            *reinterpret_cast<float4 *>(Xs + r * VC + ((part4 * 4) ^ vlad_swz(r))) = pre[u];
        }
        VSTAMP(1);
        __syncthreads();
        VSTAMP(2);
#ifndef PA_VLAD_FETCH_INTERLEAVE
#define PA_VLAD_FETCH_INTERLEAVE 1
#endif
        // the next tile's 16 loads per thread in TWO halves: back to back they took 3.3 k cycles to issue (phase stamps: every wave of the CU queues on
        // the one address path) with the matrix pipe idle; the second half goes out at the head of the soft-max, a VALU-only phase
        fetch_part(r0 + VROWS, 0, PA_VLAD_FETCH_INTERLEAVE ? PF / 2 : PF);
        VSTAMP(3);
        // 2. assignment logits for this wave's 16 rows: X[16 x 256] * Wc[256 x KP]
        floatx4 acc[KT];
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) acc[ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
        if (KT == 4 && wc_p != nullptr) {
            // packed weights: one 16-byte load per k-step brings the lane's four cluster-tile fragments, refilled four k-steps (16 MFMAs)
            // ahead through a buffer descriptor with the k-step as a scalar offset; one 16-byte LDS read per four k-steps for X.
            const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wc_p), 0, 0x7fffffff, 0x00020000);
            const unsigned wvo = (unsigned)lane * 16u;
            auto ldw = [&](int s) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, s * 1024, 0);
                return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            };
            float4 bw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bw[e] = ldw(e);
            float4 a4 = lda(0);
#pragma unroll 4
            for (int j = 0; j < VC / 16; ++j) {
                const int jn = min(j + 1, VC / 16 - 1);
                const float4 an = lda(jn);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].x, acc[0], 0, 0, 0);
                    acc[1 % KT] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].y, acc[1 % KT], 0, 0, 0);
                    acc[2 % KT] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].z, acc[2 % KT], 0, 0, 0);
                    acc[3 % KT] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bw[e].w, acc[3 % KT], 0, 0, 0);
                    bw[e] = ldw(jn * 4 + e);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    if (e == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                a4 = an;
            }
        } else {
            const float *wp = wc_t + li;
            for (int j = 0; j < VC / 16; ++j) {
                const float4 a4 = lda(j);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float bc[KT];
#pragma unroll
                    for (int ct = 0; ct < KT; ++ct) bc[ct] = wp[(size_t)vlad_chan(4 * j + e, lq) * KP + ct * 16];
#pragma unroll
                    for (int ct = 0; ct < KT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bc[ct], acc[ct], 0, 0, 0);
                }
            }
        }
        VSTAMP(4);
#ifndef PA_VLAD_SOFTMAX_PRIO
#define PA_VLAD_SOFTMAX_PRIO 2
#endif
        // The soft-max is a few hundred VALU / DPP / exp instructions per wave; the co-resident workgroup's waves are in their GEMM phases meanwhile
        // and, at equal priority, the older wave's MFMA stream wins the issue slots: the phase stamps put this block at 8.2 k cycles.  Raised priority
        // for its duration lets it through (the partner loses a few hundred cycles of issue, not eight thousand).
        if (PA_VLAD_FETCH_INTERLEAVE) fetch_part(r0 + VROWS, PF / 2, PF);
        if (PA_VLAD_SOFTMAX_PRIO) __builtin_amdgcn_s_setprio(PA_VLAD_SOFTMAX_PRIO);
        // 3. soft-max over the k_true clusters of each row.  C/D layout: column (cluster) = 16*ct + lane%16,
        //    row = 4*(lane/16) + r: a row's clusters live in the 16 lanes of a DPP row and the KT tiles.
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v[KT], mx = -3.0e38f;
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) {
                const bool live = ct * 16 + li < k_true;
                v[ct] = live ? acc[ct][r] + bia[ct] : -3.0e38f;
                mx = fmaxf(mx, v[ct]);
            }
            mx = row16_max(mx);
            float s = 0.f;
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) {
                v[ct] = (ct * 16 + li < k_true) ? __expf(v[ct] - mx) : 0.f;
                s += v[ct];
            }
            s = row16_sum(s);
            const float inv = 1.0f / s;
            const int row = wave * 16 + lq * 4 + r;
            const bool row_live = row < cnt;
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) {
                const float a = row_live ? v[ct] * inv : 0.f;
                As[row * AS + (SWZ_A ? ((ct * 16 + li) ^ vlad_swz(row)) : ct * 16 + li)] = a;
                asum[ct] += a;
            }
        }
        if (PA_VLAD_SOFTMAX_PRIO) __builtin_amdgcn_s_setprio(0);
        VSTAMP(5);
        __syncthreads();
        VSTAMP(6);
        // 4. aggregation: acc2[rt][e] += X^T[this wave's channels 64 w + 4 i + e][64 rows] * A[64 rows][clusters of tile rt].
        //    Fragments of k-step ks + 1 are read before the 16 MFMAs of k-step ks (register double buffer).
        {
            auto ldx = [&](int ks) {                                          // contraction index of k-group lq in k-step ks: row 4 ks + lq
                const int row = ks * 4 + lq, sw = ((ks & 3) << 4) | (lq << 2);   // sw = vlad_swz(row)
                return *reinterpret_cast<const float4 *>(Xs + row * VC + wave * 64 + ((4 * li) ^ sw));
            };
            auto ldas = [&](int ks) {
                const int row = ks * 4 + lq, sw = ((ks & 3) << 4) | (lq << 2);
                return *reinterpret_cast<const float4 *>(As + row * AS + ((4 * li) ^ sw));
            };
            float4 x4 = ldx(0), a4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (SWZ_A) a4 = ldas(0);
#pragma unroll
            for (int ks = 0; ks < VROWS / 4; ++ks) {
#ifdef PA_VLAD_NO_PIPE
                x4 = ldx(ks);
                if (SWZ_A) a4 = ldas(ks);
                const float4 xn = x4, an = a4;
#else
                const int kn = ks + 1 < VROWS / 4 ? ks + 1 : ks;
                const float4 xn = ldx(kn);
                float4 an = a4;
                if (SWZ_A) an = ldas(kn);
#endif
                const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
                float af[KT];
                if (SWZ_A) {
                    af[0] = a4.x; af[1 % KT] = a4.y; af[2 % KT] = a4.z; af[3 % KT] = a4.w;
                } else {
#pragma unroll
                    for (int rt = 0; rt < KT; ++rt) af[rt] = As[(ks * 4 + lq) * AS + rt * 16 + li];
                }
#pragma unroll
                for (int rt = 0; rt < KT; ++rt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[rt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[e], af[rt], acc2[rt][e], 0, 0, 0);
#ifndef PA_VLAD_NO_PIPE
                if (SWZ_A) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                }
#endif
                x4 = xn;
                a4 = an;
            }
        }
        VSTAMP(7);
        __syncthreads();
        VSTAMP(8);
    }
    VSTAMPK(11);
    // partial A^T X.  MFMA tile (rt, e): row m = 4 q + r <-> channel 64 w + 4 m + e, column i <-> cluster (K = 64: 4 i + rt, else 16 rt + i),
    // so (acc2[rt][0..3][r]) are four CONSECUTIVE CHANNELS of one cluster: 16-byte stores (the A^T X orientation with cluster-major tiles
    // needed 4 x as many 4-byte stores, ~16 k cycles per workgroup).
    float *po = part + ((size_t)b * nchunks + chunk) * KP * VC;
#pragma unroll
    for (int rt = 0; rt < KT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cl = SWZ_A ? 4 * li + rt : rt * 16 + li;
            *reinterpret_cast<float4 *>(po + (size_t)cl * VC + wave * 64 + 16 * lq + 4 * r) =
                make_float4(acc2[rt][0][r], acc2[rt][1][r], acc2[rt][2][r], acc2[rt][3][r]);
        }
    // partial a_sum: lanes sharing lane%16 hold different rows of the same cluster
#pragma unroll
    for (int ct = 0; ct < KT; ++ct) {
        float s = asum[ct];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lane < 16) red[wave * KP + ct * 16 + lane] = s;
    }
    __syncthreads();
    if (tid < KP) asum_part[((size_t)b * nchunks + chunk) * KP + tid] = (red[tid] + red[KP + tid]) + (red[2 * KP + tid] + red[3 * KP + tid]);
    VSTAMPK(12);
}

// ------------------------------------------------------------------------------------------------ NetVLAD accumulate, fp16 operands
// The K = 64 scale of the fp16 path (model.mlp_dtype = "f16", BASELINE.json configs[4]): same partials, same finalize, both contractions on
// the fp16 MFMA with fp32 accumulation; logits bias, soft-max, the a_sum partials and everything after the partials stay fp32.  The fp32 kernel
// above is bound by its MFMAs (8.6 GFLOP at 60 % of the fp32 matrix rate); here they are a few percent of the launch and it becomes a stream over X.
//   * the LOGITS keep fp32 accuracy: x = hi + lo and W = Whi + Wlo in fp16 pairs, logits = hi Whi + lo Whi + hi Wlo (three MFMAs, the dropped
//     term is 2^-22 relative).  A plain fp16 rounding of x is an ABSOLUTE error |x| |W| 2^-11 sqrt(256) in the exponent of the soft-max: with
//     features of magnitude 50-100 (PPT-Net with seeded random weights: logits of standard deviation 17, nearly one-hot assignments) it flips
//     assignments -- descriptor cosine 0.52-0.98 in the first build (tools/probes/vlad16_diag.py);
//   * a lane loads 8 consecutive channels of ITS point per 32-channel k-step (two 16-byte global loads): after the conversion these registers
//     ARE the A operands (M = point) of the assignment GEMM v_mfma_f32_16x16x32_f16, and one 16-byte LDS store puts the hi part into the
//     row-major fp16 tile Xh[128][256 + 8];
//   * the assignment weights (pa_pack_weights_f16(256, 64) of W and of W - fp16(W): lane (k % 16, g) holds W[32 ks + 8 g + e][16 kt + k % 16])
//     are copied into LDS once per workgroup (global_load_lds) and read lane-linearly;
//   * the logits come out as D[point 4 g + i][cluster 16 kt + l % 16]: a row's clusters are the 16 lanes of a DPP row x 4 tiles (the soft-max of
//     the fp32 kernel), and a lane's four values ARE four consecutive points of one cluster = one A operand (M = cluster, K = 16 points) of the
//     aggregation GEMM v_mfma_f32_16x16x16_f16: they pass through LDS as one 8-byte word per (point quad, cluster) only because the aggregation
//     splits the CHANNELS over the waves (every wave contracts all 128 points of the tile for its 32 channels);
//   * the aggregation's B operand (N = channel, K = 16 points) is four 2-byte reads down a column of Xh; the 528-byte row pitch puts the four
//     k-groups' rows 16 banks apart;
//   * the AGGREGATION runs in bf16 (v_mfma_f32_16x16x16_bf16), not fp16: a cluster nobody is assigned to still gets a row of the descriptor --
//     intra-normalisation scales its 1e-8-and-below masses to unit norm -- and those masses are under fp16's range (first build: such rows came
//     out with cosine ~0 against the fp32 kernel).  bf16 has fp32's exponent; its 8-bit mantissa costs 2^-9 relative per term of sums over
//     hundreds to thousands of points (measured: every row within cosine 0.999997 of the fp32 kernel, tests/test_gpu_f16.py).
// One eight-wave workgroup per CU (152 KB of LDS: the tile, both weight halves, the assignment words).
typedef _Float16 vhalf8 __attribute__((ext_vector_type(8)));
typedef _Float16 vhalf4 __attribute__((ext_vector_type(4)));
typedef __bf16 vbf8 __attribute__((ext_vector_type(8)));
typedef __bf16 vbf4 __attribute__((ext_vector_type(4)));
typedef short vshort4 __attribute__((ext_vector_type(4)));
constexpr int V16_ROWS = 128;                                                // rows of X per tile (16 per wave)
constexpr int V16_PITCH = VC + 8;                                            // halfs per row of Xh
constexpr int V16_AQ = 66;                                                   // (point quad) pitch of the assignment tile, in clusters
constexpr size_t V16_LDS = (size_t)V16_ROWS * V16_PITCH * 2 + 2 * 32768 + (V16_ROWS / 4) * V16_AQ * 8 + 8 * 64 * 4;

// X16: the feature map arrives as fp16 rows (pa_fp_chain_premul_g16h wrote it for this kernel alone): half the read, no (hi, lo) split of x
// (the map IS its hi part; its own rounding, 2^-11 relative per feature, is that of every other operand of the fp16 path)
template <bool X16>
__global__ __launch_bounds__(512, 1) void vlad_accum16_kernel(int n, int k_true, int rows_per_wg, const float *__restrict__ x_all,
                                                              const _Float16 *__restrict__ wc16,   // pa_pack_weights_f16(256, 64) of W, then of W - fp16(W)
                                                              const float *__restrict__ bias, float *__restrict__ part, float *__restrict__ asum_part)
{
    constexpr int KT = 4, KP = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    __bf16 *Xh = reinterpret_cast<__bf16 *>(smem16);                          // [128][V16_PITCH] bf16
    _Float16 *Wh = reinterpret_cast<_Float16 *>(Xh + V16_ROWS * V16_PITCH);   // [hi, lo][4 kt][8 ks][64 lanes][8] fp16
    __bf16 *Aq = reinterpret_cast<__bf16 *>(Wh + 2 * 16384);                  // [32 point quads][V16_AQ clusters][4 points] bf16
    float *red = reinterpret_cast<float *>(Aq + (V16_ROWS / 4) * V16_AQ * 4); // [8][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int row_begin = chunk * rows_per_wg;
    const int row_end = min(row_begin + rows_per_wg, n);
    const float *x = x_all + (size_t)b * n * VC;
    const _Float16 *xh16 = reinterpret_cast<const _Float16 *>(x_all) + (size_t)b * n * VC;      // X16

    // assignment weights -> LDS (64 pieces of 1 KB, eight per wave)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int p = u * 8 + wave;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(wc16 + (size_t)p * 512 + lane * 8),
                                         (void __attribute__((address_space(3))) *)(reinterpret_cast<unsigned char *>(Wh) + p * 1024), 16, 0, 0);
    }
    floatx4 acc2[KT][2];                   // V: [cluster tile kt][channel tile ct]: row m = 4 g + i <-> cluster 16 kt + m, column <-> channel 32 w + 16 ct + li
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc2[kt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float asum[KT], bia[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { asum[kt] = 0.f; bia[kt] = bias[kt * 16 + li]; }

    // a lane's point of the tile at r0: row r0 + 16 w + li (clamped: rows past the end get a = 0 below), channels 32 ks + 8 lq .. + 7
    float4 pre[X16 ? 8 : 16];
    auto fetch = [&](int r0) {
        const int row = min(r0 + wave * 16 + li, row_end - 1);
        if constexpr (X16) {
            const float4 *src = reinterpret_cast<const float4 *>(xh16 + (size_t)row * VC + 8 * lq);      // eight halfs = 16 bytes per k-step
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) pre[ks] = src[4 * ks];
        } else {
            const float4 *src = reinterpret_cast<const float4 *>(x + (size_t)row * VC + 8 * lq);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) { pre[2 * ks] = src[8 * ks]; pre[2 * ks + 1] = src[8 * ks + 1]; }
        }
    };
    fetch(row_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the weight copies (and the first tile) have landed ...
    __syncthreads();                                            // ... everybody's
    const vhalf8 *wf = reinterpret_cast<const vhalf8 *>(Wh) + lane;
    for (int r0 = row_begin; r0 < row_end; r0 += V16_ROWS) {
        const int cnt = min(V16_ROWS, row_end - r0);
        // 1 + 2. convert (hi, lo = x - hi); logits of this wave's 16 points D[point 4 g + i][cluster 16 kt + li] = hi Whi + lo Whi + hi Wlo; the hi
        //        part into the row-major tile
        floatx4 acc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc[kt] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            vhalf8 hi, lo;
            vbf8 xb;
            if constexpr (X16) {
                hi = __builtin_bit_cast(vhalf8, pre[ks]);
#pragma unroll
                for (int e = 0; e < 8; ++e) xb[e] = (__bf16)(float)hi[e];
            } else {
                const float4 a0 = pre[2 * ks], a1 = pre[2 * ks + 1];
                const float xv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    hi[e] = (_Float16)xv[e];
                    lo[e] = (_Float16)(xv[e] - (float)hi[e]);
                    xb[e] = (__bf16)xv[e];
                }
            }
            *reinterpret_cast<vbf8 *>(Xh + (wave * 16 + li) * V16_PITCH + 32 * ks + 8 * lq) = xb;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const vhalf8 wh = wf[(kt * 8 + ks) * 64], wl = wf[(32 + kt * 8 + ks) * 64];
                acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, wh, acc[kt], 0, 0, 0);
                if constexpr (!X16) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, wh, acc[kt], 0, 0, 0);
                acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, wl, acc[kt], 0, 0, 0);
            }
        }
        if (r0 + V16_ROWS < row_end) fetch(r0 + V16_ROWS);     // in flight under the soft-max and the aggregation
        // 3. soft-max over the clusters of each point (fp32), the assignments as (point quad, cluster) words
        float av[KT][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v[KT], mx = -3.0e38f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                v[kt] = (kt * 16 + li < k_true) ? acc[kt][r] + bia[kt] : -3.0e38f;
                mx = fmaxf(mx, v[kt]);
            }
            mx = row16_max(mx);
            float sm = 0.f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                v[kt] = (kt * 16 + li < k_true) ? __expf(v[kt] - mx) : 0.f;
                sm += v[kt];
            }
            sm = row16_sum(sm);
            const float inv = 1.0f / sm;
            const bool row_live = wave * 16 + lq * 4 + r < cnt;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                av[kt][r] = row_live ? v[kt] * inv : 0.f;
                asum[kt] += av[kt][r];
            }
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
            *reinterpret_cast<vbf4 *>(Aq + ((wave * 4 + lq) * V16_AQ + kt * 16 + li) * 4) =
                (vbf4){(__bf16)av[kt][0], (__bf16)av[kt][1], (__bf16)av[kt][2], (__bf16)av[kt][3]};
        __syncthreads();
        // 4. aggregation over the tile's 128 points for this wave's 32 channels: K = 16 points per MFMA
#pragma unroll
        for (int kk = 0; kk < V16_ROWS / 16; ++kk) {
            vbf4 xf[2], af[KT];
            const __bf16 *xc = Xh + (16 * kk + 4 * lq) * V16_PITCH + wave * 32 + li;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                xf[ct] = (vbf4){xc[16 * ct], xc[16 * ct + V16_PITCH], xc[16 * ct + 2 * V16_PITCH], xc[16 * ct + 3 * V16_PITCH]};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) af[kt] = *reinterpret_cast<const vbf4 *>(Aq + ((4 * kk + lq) * V16_AQ + kt * 16 + li) * 4);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc2[kt][ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(vshort4, af[kt]), __builtin_bit_cast(vshort4, xf[ct]), acc2[kt][ct], 0, 0, 0);
        }
        __syncthreads();
    }
    // partials in the fp32 kernel's layout: part[b][chunk][cluster][channel]
    float *po = part + ((size_t)b * nchunks + chunk) * KP * VC;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) po[(size_t)(kt * 16 + 4 * lq + r) * VC + wave * 32 + ct * 16 + li] = acc2[kt][ct][r];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        float sv = asum[kt];
        sv += __shfl_xor(sv, 16);
        sv += __shfl_xor(sv, 32);
        if (lane < 16) red[wave * KP + kt * 16 + lane] = sv;
    }
    __syncthreads();
    if (tid < KP) {
        float sv = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sv += red[w * KP + tid];
        asum_part[((size_t)b * nchunks + chunk) * KP + tid] = sv;
    }
}

constexpr int VLAD_MAX_SCALES = 4;

// wc_t [256][64] K-major -> the fragment order of the assignment GEMM above: wc_p[(s * 64 + lane) * 4 + ct] = wc_t[vlad_chan(s, lane / 16)][16 ct + lane % 16]
__global__ __launch_bounds__(256) void vlad_pack_kernel(const float *__restrict__ wc_t, float *__restrict__ wc_p)
{
    const int t = blockIdx.x * 256 + threadIdx.x;                  // (s, lane)
    const int s = t >> 6, lane = t & 63;
    const float *src = wc_t + (size_t)vlad_chan(s, lane >> 4) * 64 + (lane & 15);
    reinterpret_cast<float4 *>(wc_p)[t] = make_float4(src[0], src[16], src[32], src[48]);
}

// grid (k_true, B), 256 threads = channels: reduce partials, subtract a_sum*W2, intra-normalise, write (B, VC, ldo) at column koff+k
__global__ __launch_bounds__(256) void vlad_finalize_kernel(int nchunks, int kp, const float *__restrict__ part, const float *__restrict__ asum_part,
                                                              const float *__restrict__ w2,  // [VC][k_true]
                                                              int k_true, float *__restrict__ out, int ldo, int koff, int rows_layout)
{
    __shared__ float red[4];
    const int k = blockIdx.x, b = blockIdx.y, c = threadIdx.x;
    float v = 0.f, a = 0.f;
    for (int ch = 0; ch < nchunks; ++ch) {
        v += part[(((size_t)b * nchunks + ch) * kp + k) * VC + c];
        a += asum_part[((size_t)b * nchunks + ch) * kp + k];
    }
    v = v - a * w2[(size_t)c * k_true + k];          // loupe.py:213-219
    float ss = v * v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) ss += __shfl_xor(ss, o);
    if ((c & 63) == 0) red[c >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    const float r = v / fmaxf(nrm, 1e-12f);                                // F.normalize(dim=1), loupe.py:221
    if (rows_layout) out[((size_t)b * ldo + koff + k) * VC + c] = r;        // (B, sum K, C): one contiguous 1 KB row per cluster
    else out[((size_t)b * VC + c) * ldo + koff + k] = r;                    // (B, C, sum K): the reference's layout
}

// ------------------------------------------------------------------------------------------------ APFA attention
// grid (8, B): workgroup p computes r[o][k] = sum_c Watt[o][c] * x[c][k] for its 32 output channels o and all ktot
// columns, then the column-wise max over those o.  pmax[b][p][k].
__global__ __launch_bounds__(256) void afa_colmax_kernel(int ktot, const float *__restrict__ v_all,   // (B, VC, ktot)
                                                           const float *__restrict__ watt,             // [VC][VC] row-major (o, c)
                                                           float *__restrict__ pmax)                   // (B, 8, ktot)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];   // x tile [VC][ktp + 2] zero-padded columns
    const int ktp = (ktot + 15) & ~15;
    const int xs = ktp + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, p = blockIdx.x;
    const float *x = v_all + (size_t)b * VC * ktot;
    for (int q = tid; q < VC * ktp; q += 256) {
        const int c = q / ktp, k = q - c * ktp;
        smem[c * xs + k] = k < ktot ? x[(size_t)c * ktot + k] : 0.f;
    }
    __syncthreads();
    // wave w: output rows o = p*32 + (w&1)*16 + lane%16, column tiles (w>>1), (w>>1)+2, ... of the ktp/16 tiles
    const int o0 = p * 32 + (wave & 1) * 16;
    const int nct = ktp >> 4;
    const float *ap = watt + (size_t)(o0 + (lane & 15)) * VC + (lane >> 4);     // A[i = o][k = c]
    float af[VC / 4];                                                             // the wave's whole A panel: reused by every column tile
#pragma unroll
    for (int ks = 0; ks < VC / 4; ++ks) af[ks] = ap[ks * 4];
    for (int ct = wave >> 1; ct < nct; ct += 2) {
        floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
        const float *bp = smem + (lane >> 4) * xs + ct * 16 + (lane & 15);      // B[k = c][j = column]
#pragma unroll
        for (int ks = 0; ks < VC / 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], bp[ks * 4 * xs], acc, 0, 0, 0);
        float m = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));            // rows 4*(lane/16) + r
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        // two waves (w&1 = 0/1) cover the 32 rows of this workgroup for the same columns: combine through LDS below
        if (lane < 16) smem[VC * xs + (wave & 1) * ktp + ct * 16 + lane] = m;
    }
    __syncthreads();
    for (int k = tid; k < ktot; k += 256) pmax[((size_t)b * 8 + p) * ktot + k] = fmaxf(smem[VC * xs + k], smem[VC * xs + ktp + k]);
}

// grid B: w = softmax_k(max_p pmax);  y[c][k] = relu(x + x*w)    (flat (B, VC*ktot), the FC input)
__global__ __launch_bounds__(256) void afa_reweight_kernel(int ktot, const float *__restrict__ v_all, const float *__restrict__ pmax,
                                                             float *__restrict__ y_all)
{
    __shared__ float w[256];
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    float m = -3.0e38f;
    if (tid < ktot)
        for (int p = 0; p < 8; ++p) m = fmaxf(m, pmax[((size_t)b * 8 + p) * ktot + tid]);
    float mx = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float e = tid < ktot ? __expf(m - mx) : 0.f;
    float s = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = s;
    __syncthreads();
    s = (red[4] + red[5]) + (red[6] + red[7]);
    w[tid] = e / s;
    __syncthreads();
    const float *x = v_all + (size_t)b * VC * ktot;
    float *y = y_all + (size_t)b * VC * ktot;
    for (int q = tid; q < VC * ktot; q += 256) {
        const float xv = x[q];
        y[q] = fmaxf(xv + xv * w[q % ktot], 0.f);          // loupe.py:33-38
    }
}

// ------------------------------------------------------------------------------------------------ split-K FC
// out_part[s][b][n] = sum_{k in slice s} y[b][k] * Wt[k][n];  Wt K-major (kdim x nout), slices of KS rows, grid = nslices.
// The 22 MB weight stream comes from HBM once: every operand is fetched 16 bytes per lane.  A lane's float4 of Wt is four CONSECUTIVE
// COLUMNS of one k row (MFMA tile j of the wave's 64-column group holds columns 4 i + j in its column i), its float4 of y four consecutive
// k of one batch row (k-step (u, e) contracts k0 + 16 u + 4 q + e in k-group q) -- 256-byte segments instead of 64-byte ones, a quarter of
// the load instructions (17 -> ~10 us at b = 32, 21 504 x 256); needs nout % 64 == 0, 16-byte aligned rows, else the 4-byte path.
template <int RT>
__global__ __launch_bounds__(256) void fc_splitk_kernel(int bsz, int kdim, int nout, int ks_rows, const float *__restrict__ y,
                                                          const float *__restrict__ wt, float *__restrict__ out_part, int vec)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int s = blockIdx.x;
    const int k_begin = s * ks_rows, k_end = min(k_begin + ks_rows, kdim);
    const int nct = nout >> 4;
    for (int c0 = wave * 4; c0 < nct; c0 += 16) {   // this wave: column tiles c0..c0+3
        floatx4 acc[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
        if (vec) {
            for (int k0 = k_begin; k0 < k_end; k0 += 32) {   // 8 k-steps of operands in flight
                float4 a4[2][RT], b4[8];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kk = k0 + 16 * u + 4 * lq;     // k_end - k_begin and kdim are multiples of 4 here: a float4 is all in or all out
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const int row = rt * 16 + li;
                        a4[u][rt] = (row < bsz && kk < k_end) ? *reinterpret_cast<const float4 *>(y + (size_t)row * kdim + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kr = kk + e;
                        b4[u * 4 + e] = kr < k_end ? *reinterpret_cast<const float4 *>(wt + (size_t)kr * nout + c0 * 16 + 4 * li) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float4 bw = b4[u * 4 + e];
                        const float bv[4] = {bw.x, bw.y, bw.z, bw.w};
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            const float av = e == 0 ? a4[u][rt].x : e == 1 ? a4[u][rt].y : e == 2 ? a4[u][rt].z : a4[u][rt].w;
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[ct], acc[rt][ct], 0, 0, 0);
                        }
                    }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rt * 16 + lq * 4 + r;
                    if (row < bsz)
                        *reinterpret_cast<float4 *>(out_part + ((size_t)s * bsz + row) * nout + c0 * 16 + 4 * li) =
                            make_float4(acc[rt][0][r], acc[rt][1][r], acc[rt][2][r], acc[rt][3][r]);
                }
            continue;
        }
        for (int k0 = k_begin; k0 < k_end; k0 += 32) {   // 8 k-steps of operands in flight: the weight stream comes from HBM
            float af[8][RT], bf[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k0 + u * 4 + (lane >> 4);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const int row = rt * 16 + (lane & 15);
                    af[u][rt] = (row < bsz && kk < k_end) ? y[(size_t)row * kdim + kk] : 0.f;
                }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    bf[u][ct] = (kk < k_end && c0 + ct < nct) ? wt[(size_t)kk * nout + (c0 + ct) * 16 + (lane & 15)] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[u][rt], bf[u][ct], acc[rt][ct], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rt * 16 + (lane >> 4) * 4 + r;
                    if (row < bsz && c0 + ct < nct) out_part[((size_t)s * bsz + row) * nout + (c0 + ct) * 16 + (lane & 15)] = acc[rt][ct][r];
                }
    }
}

// grid B, 4 x nout threads (nout <= 256) or nout threads: sum the slices, + bias, BatchNorm (eval) as scale/shift, optional gating,
// optional L2 normalise.  The slice sum is pure load latency, so four thread groups take every fourth slice with eight
// independent accumulators each and meet in LDS.
__global__ void fc_finalize_kernel(int bsz, int nout, int nslices, const float *__restrict__ out_part, const float *__restrict__ fc_bias,
                                   const float *__restrict__ scale, const float *__restrict__ shift, int l2norm, const float *__restrict__ gate_x,
                                   float *__restrict__ desc)
{
    __shared__ float red[16];
    __shared__ float grp[3 * 256];
    const int b = blockIdx.x, n = threadIdx.x % nout, g = threadIdx.x / nout, ng = blockDim.x / nout;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = g;
    for (; s + 7 * ng < nslices; s += 8 * ng)
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += out_part[((size_t)(s + u * ng) * bsz + b) * nout + n];
    for (; s < nslices; s += ng) a[0] += out_part[((size_t)s * bsz + b) * nout + n];
    float v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    if (ng > 1) {
        if (g > 0) grp[(g - 1) * nout + n] = v;
        __syncthreads();
        if (g > 0) return;
        for (int o = 0; o < ng - 1; ++o) v += grp[o * nout + n];
    }
    v = (v + (fc_bias ? fc_bias[n] : 0.f)) * scale[n] + shift[n];
    if (gate_x) v = gate_x[(size_t)b * nout + n] * (1.0f / (1.0f + expf(-v)));   // GatingContext, loupe.py:332-361: x * sigmoid(BN(x W))
    if (l2norm) {
        float ss = v * v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) ss += __shfl_xor(ss, o);
        if ((n & 63) == 0) red[n >> 6] = ss;
        __syncthreads();                                    // only group 0 (nout threads, whole waves) is still here
        float t = 0.f;
        for (int w = 0; w < (nout + 63) / 64; ++w) t += red[w];
        v = v / fmaxf(sqrtf(t), 1e-12f);                   // F.normalize, loupe.py:63-64
    }
    desc[(size_t)b * nout + n] = v;
}

// cluster-major head, grid B x 256 threads: w = softmax_k(max_o logits[b][k][o]) (loupe.py:33-36), y[b][k][:] = relu(x + x*w[k])
__global__ __launch_bounds__(256) void afa_rows_reweight_kernel(int ktot, const float *__restrict__ vt_all, const float *__restrict__ logits_all,
                                                                  float *__restrict__ y_all)
{
    __shared__ float w[256];
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 *lg = reinterpret_cast<const float4 *>(logits_all + (size_t)b * ktot * VC);
    for (int k = wave; k < ktot; k += 4) {                      // row max: one wavefront per 1 KB row
        const float4 v = lg[(size_t)k * 64 + lane];
        float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) w[k] = m;
    }
    __syncthreads();
    const float m = tid < ktot ? w[tid] : -3.0e38f;
    float mx = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float e = tid < ktot ? __expf(m - mx) : 0.f;
    float s = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[4 + wave] = s;
    __syncthreads();
    s = (red[4] + red[5]) + (red[6] + red[7]);
    w[tid] = e / s;
    __syncthreads();
    const float4 *x4 = reinterpret_cast<const float4 *>(vt_all + (size_t)b * ktot * VC);
    float4 *y4 = reinterpret_cast<float4 *>(y_all + (size_t)b * ktot * VC);
    for (int q = tid; q < ktot * 64; q += 256) {
        const float wk = w[q >> 6];
        const float4 xv = x4[q];
        y4[q] = make_float4(fmaxf(xv.x + xv.x * wk, 0.f), fmaxf(xv.y + xv.y * wk, 0.f), fmaxf(xv.z + xv.z * wk, 0.f), fmaxf(xv.w + xv.w * wk, 0.f));
    }
}

// two workgroups per CU: 4096 points x 32 clouds = 512 workgroups of four 64-row tiles
int vlad_rows_per_wg(int n) { return n >= 2048 ? 512 / PA_VLAD_WGS : 64; }
int vlad_chunks(int n) { const int rows = vlad_rows_per_wg(n); return (n + rows - 1) / rows; }

// split-K FC + finalize for any batch size: rows are processed 64 at a time (the split-K kernel holds <= 4 row tiles)
int fc_launch(int b, int kdim, int nout, const float *y, const float *fc_wt, const float *fc_bias, const float *scale, const float *shift, int l2norm,
              const float *gate_x, float *opart, float *out, hipStream_t st)
{
    const int ks_rows = 128;
    const int nslices = (kdim + ks_rows - 1) / ks_rows;
    // 16-byte operand path: whole 64-column groups, k in whole float4s, aligned rows
    const int vec = (nout % 64 == 0 && kdim % 4 == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)fc_wt & 15) == 0 && ((uintptr_t)opart & 15) == 0) ? 1 : 0;
    for (int b0 = 0; b0 < b; b0 += 64) {
        const int bc = b - b0 < 64 ? b - b0 : 64;
        const float *yc = y + (size_t)b0 * kdim;
        switch ((bc + 15) / 16) {
            case 1: hipLaunchKernelGGL(fc_splitk_kernel<1>, dim3(nslices), dim3(256), 0, st, bc, kdim, nout, ks_rows, yc, fc_wt, opart, vec); break;
            case 2: hipLaunchKernelGGL(fc_splitk_kernel<2>, dim3(nslices), dim3(256), 0, st, bc, kdim, nout, ks_rows, yc, fc_wt, opart, vec); break;
            case 3: hipLaunchKernelGGL(fc_splitk_kernel<3>, dim3(nslices), dim3(256), 0, st, bc, kdim, nout, ks_rows, yc, fc_wt, opart, vec); break;
            default: hipLaunchKernelGGL(fc_splitk_kernel<4>, dim3(nslices), dim3(256), 0, st, bc, kdim, nout, ks_rows, yc, fc_wt, opart, vec); break;
        }
        hipLaunchKernelGGL(fc_finalize_kernel, dim3(bc), dim3(nout <= 256 ? 4 * nout : nout), 0, st, bc, nout, nslices, opart, fc_bias, scale, shift, l2norm,
                           gate_x ? gate_x + (size_t)b0 * nout : nullptr, out + (size_t)b0 * nout);
    }
    return PA_OK;
}

}  // namespace

// Fragment-ordered copy of the K = 64 assignment weights for pa_netvlad_rows: wc_t (256, 64) K-major (BatchNorm folded) -> wc_p (256 * 64 floats).
PA_API int pa_netvlad_pack_weights(int c, int kp, const float *wc_t, float *wc_p, pa_stream_t stream)
{
    PA_REQUIRE(wc_t && wc_p, "pa_netvlad_pack_weights: null pointer");
    if (c != VC || kp != 64) { pa_set_error("pa_netvlad_pack_weights: built for 256 channels x 64 padded clusters (got c=%d kp=%d)", c, kp); return PA_EUNSUPPORTED; }
    hipLaunchKernelGGL(vlad_pack_kernel, dim3(VC / 4 * 64 / 256), dim3(256), 0, (hipStream_t)stream, wc_t, wc_p);
    PA_CHECK_LAUNCH("pa_netvlad_pack_weights");
    return PA_OK;
}

PA_API long pa_netvlad_scratch_floats(int b, int n, int k)
{
    const int kp = (k + 15) & ~15;
    return (long)b * vlad_chunks(n) * ((long)kp * VC + kp);
}

// X (b, n, 256) point-major -> out[b][c][koff + k], k < k_true, ldo floats per (b, c) row.
static int netvlad_impl(int b, int n, int c, int k, const float *x, const float *wc_t, const float *wc_p, const float *bias, const float *w2,
                        float *scratch, float *out, int ldo, int koff, int rows_layout, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && k > 0 && x && wc_t && bias && w2 && scratch && out, "pa_netvlad: bad arguments");
    PA_REQUIRE(b <= 65535, "pa_netvlad: b=%d exceeds the grid limit", b);
    if (c != VC || k > 64) { pa_set_error("pa_netvlad: built for 256 channels and <= 64 clusters (got c=%d k=%d)", c, k); return PA_EUNSUPPORTED; }
    hipStream_t st = (hipStream_t)stream;
    const int kp = (k + 15) & ~15, kt = kp / 16;
    const int chunks = vlad_chunks(n), rows = vlad_rows_per_wg(n);
    float *part = scratch;
    float *asum = scratch + (size_t)b * chunks * kp * VC;
    const size_t lds = (size_t)(VROWS * VC + VROWS * (kt == 4 ? kp : kp + 2)) * 4;      // K = 64: 80 KB, half a CU
#define PA_VLAD_LAUNCH(KT)                                                                                                              \
    do {                                                                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vlad_accum_kernel<KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(vlad_accum_kernel<KT>, dim3(chunks, b), dim3(256), lds, st, n, k, rows, x, wc_t, wc_p, bias, part, asum);         \
    } while (0)
    switch (kt) {
        case 1: PA_VLAD_LAUNCH(1); break;
        case 2: PA_VLAD_LAUNCH(2); break;
        case 3: PA_VLAD_LAUNCH(3); break;
        default: PA_VLAD_LAUNCH(4); break;
    }
#undef PA_VLAD_LAUNCH
    hipLaunchKernelGGL(vlad_finalize_kernel, dim3(k, b), dim3(256), 0, st, chunks, kp, part, asum, w2, k, out, ldo, koff, rows_layout);
    PA_CHECK_LAUNCH("pa_netvlad");
    return PA_OK;
}

PA_API int pa_netvlad(int b, int n, int c, int k, const float *x, const float *wc_t, const float *bias, const float *w2, float *scratch,
                      float *out, int ldo, int koff, pa_stream_t stream)
{
    return netvlad_impl(b, n, c, k, x, wc_t, nullptr, bias, w2, scratch, out, ldo, koff, 0, stream);
}

// Same, writing cluster-major rows: out[b][koff + j][c] with ktot = ldo rows of 256 floats per batch element -- the layout the
// row-oriented head (pa_afa_rows) consumes with whole-row loads.
PA_API int pa_netvlad_rows(int b, int n, int c, int k, const float *x, const float *wc_t, const float *wc_p, const float *bias, const float *w2,
                           float *scratch, float *out, int ktot, int koff, pa_stream_t stream)
{
    return netvlad_impl(b, n, c, k, x, wc_t, (k > 48 ? wc_p : nullptr), bias, w2, scratch, out, ktot, koff, 1, stream);
}

PA_API long pa_afa_scratch_floats(int b, int c, int ktot, int nout)
{
    const long kdim = (long)c * ktot;
    const long nslices = (kdim + 127) / 128;
    return (long)b * 8 * ktot + (long)b * kdim + nslices * (b < 64 ? b : 64) * nout;
}

// v (b, 256, ktot) -> desc (b, nout).  watt: (256, 256) row-major (o, c);  fc_wt: K-major (256*ktot, nout);
// scale/shift: BatchNorm1d (eval) folded to y*scale + shift;  l2norm != 0: F.normalize.
PA_API int pa_afa(int b, int c, int ktot, int nout, const float *v, const float *watt, const float *fc_wt, const float *fc_bias,
                  const float *scale, const float *shift, int l2norm, float *scratch, float *desc, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && ktot > 0 && nout > 0 && v && watt && fc_wt && fc_bias && scale && shift && scratch && desc, "pa_afa: bad arguments");
    if (c != VC || ktot > 256 || nout % 16 || nout > 1024 || b > 65535) {
        pa_set_error("pa_afa: built for 256 channels, <= 256 columns, nout %% 16 == 0, nout <= 1024 (got c=%d ktot=%d nout=%d b=%d)", c, ktot, nout, b);
        return PA_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const int kdim = c * ktot;
    float *pmax = scratch;
    float *y = pmax + (size_t)b * 8 * ktot;
    float *opart = y + (size_t)b * kdim;
    const int ktp = (ktot + 15) & ~15;
    const size_t lds = (size_t)(VC * (ktp + 2) + 2 * ktp) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&afa_colmax_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(afa_colmax_kernel, dim3(8, b), dim3(256), lds, st, ktot, v, watt, pmax);
    hipLaunchKernelGGL(afa_reweight_kernel, dim3(b), dim3(256), 0, st, ktot, v, pmax, y);
    const int rc = fc_launch(b, kdim, nout, y, fc_wt, fc_bias, scale, shift, l2norm, nullptr, opart, desc, st);
    if (rc != PA_OK) return rc;
    PA_CHECK_LAUNCH("pa_afa");
    return PA_OK;
}

PA_API long pa_fc_scratch_floats(int b, int kdim, int nout)
{
    const long nslices = (kdim + 127) / 128;
    return nslices * (b < 64 ? b : 64) * nout;
}

// out (b, nout) = [gate_x *] f(BN(y (b, kdim) . fc_wt (kdim x nout, K-major) + fc_bias)), f = sigmoid when gate_x is given
// (GatingContext, loupe.py:332-361) else identity; then optional F.normalize.  fc_bias / gate_x may be NULL.
PA_API int pa_fc(int b, int kdim, int nout, const float *y, const float *fc_wt, const float *fc_bias, const float *scale, const float *shift,
                 int l2norm, const float *gate_x, float *scratch, float *out, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && kdim > 0 && nout > 0 && y && fc_wt && scale && shift && scratch && out, "pa_fc: bad arguments");
    if (nout % 16 || nout > 1024) { pa_set_error("pa_fc: nout=%d must be a multiple of 16 and <= 1024", nout); return PA_EUNSUPPORTED; }
    const int rc = fc_launch(b, kdim, nout, y, fc_wt, fc_bias, scale, shift, l2norm, gate_x, scratch, out, (hipStream_t)stream);
    if (rc != PA_OK) return rc;
    PA_CHECK_LAUNCH("pa_fc");
    return PA_OK;
}

PA_API long pa_afa_rows_scratch_floats(int b, int c, int ktot, int nout)
{
    const long kdim = (long)c * ktot;
    const long nslices = (kdim + 127) / 128;
    return 2 * (long)b * kdim + nslices * (b < 64 ? b : 64) * nout;
}

// Cluster-major APFA head.  vt (b, ktot, 256) from pa_netvlad_rows; watt_t: the attention conv as a K-major (in, out) matrix (+ its
// packed copy or NULL); zero_bias: 256 zeros; fc_wt: K-major (ktot*256, nout) with ROWS ORDERED k*256 + c (the reference's FC weight
// has rows c*ktot + k; the caller permutes once).  Launches: logits = vt . watt_t (MFMA chain kernel), row max + soft-max +
// re-weighting, split-K FC, finalize.
PA_API int pa_afa_rows(int b, int c, int ktot, int nout, const float *vt, const float *watt_t, const float *watt_p, const float *zero_bias,
                       const float *fc_wt, const float *fc_bias, const float *scale, const float *shift, int l2norm, float *scratch,
                       float *desc, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && ktot > 0 && nout > 0 && vt && watt_t && zero_bias && fc_wt && fc_bias && scale && shift && scratch && desc,
               "pa_afa_rows: bad arguments");
    if (c != VC || ktot > 256 || nout % 16 || nout > 1024) {
        pa_set_error("pa_afa_rows: built for 256 channels, <= 256 clusters, nout %% 16 == 0, nout <= 1024 (got c=%d ktot=%d nout=%d)", c, ktot, nout);
        return PA_EUNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const int kdim = c * ktot;
    float *logits = scratch;
    float *y = logits + (size_t)b * kdim;
    float *opart = y + (size_t)b * kdim;
    int rc = pa_linear((long)b * ktot, VC, VC, vt, VC, watt_t, watt_p, zero_bias, 0, nullptr, 0, logits, VC, stream);
    if (rc != PA_OK) return rc;
    hipLaunchKernelGGL(afa_rows_reweight_kernel, dim3(b), dim3(256), 0, st, ktot, vt, logits, y);
    rc = fc_launch(b, kdim, nout, y, fc_wt, fc_bias, scale, shift, l2norm, nullptr, opart, desc, st);
    if (rc != PA_OK) return rc;
    PA_CHECK_LAUNCH("pa_afa_rows");
    return PA_OK;
}

namespace {
// grid B x 256 threads (= channels): max over the cluster rows, optional L2 normalisation over the 256 channels
__global__ __launch_bounds__(256) void vlad_maxpool_kernel(int ktot, const float *__restrict__ vt_all, int l2norm, float *__restrict__ out)
{
    __shared__ float red[4];
    const int b = blockIdx.x, c = threadIdx.x;
    const float *vt = vt_all + (size_t)b * ktot * VC;
    float m = vt[c];
    for (int k = 1; k < ktot; ++k) m = fmaxf(m, vt[(size_t)k * VC + c]);
    if (l2norm) {
        float ss = m * m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) ss += __shfl_xor(ss, o);
        if ((c & 63) == 0) red[c >> 6] = ss;
        __syncthreads();
        m = m / fmaxf(sqrtf((red[0] + red[1]) + (red[2] + red[3])), 1e-12f);
    }
    out[(size_t)b * VC + c] = m;
}
}  // namespace

PA_API int pa_vlad_maxpool(int b, int ktot, int c, const float *vt, int l2norm, float *out, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && ktot > 0 && vt && out, "pa_vlad_maxpool: bad arguments");
    if (c != VC) { pa_set_error("pa_vlad_maxpool: built for 256 channels (got c=%d)", c); return PA_EUNSUPPORTED; }
    hipLaunchKernelGGL(vlad_maxpool_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, ktot, vt, l2norm, out);
    PA_CHECK_LAUNCH("pa_vlad_maxpool");
    return PA_OK;
}

// ------------------------------------------------------------------------------------------------ whole pyramid + fused APFA head
namespace {

// finalize of ALL scales in one launch: grid (ktot, B); cluster row kk of the concatenation belongs to scale s with koff[s] <= kk < koff[s+1]
struct VladFinalizeMulti {
    int nscales;
    int koff[VLAD_MAX_SCALES + 1], k[VLAD_MAX_SCALES], kp[VLAD_MAX_SCALES], nchunks[VLAD_MAX_SCALES];
    const float *part[VLAD_MAX_SCALES], *asum[VLAD_MAX_SCALES], *w2[VLAD_MAX_SCALES];
};

__global__ __launch_bounds__(256) void vlad_finalize_multi_kernel(VladFinalizeMulti m, float *__restrict__ out, int ktot)
{
    __shared__ float red[4];
    const int kk = blockIdx.x, b = blockIdx.y, c = threadIdx.x;
    int s = 0;
#pragma unroll
    for (int t = 1; t < VLAD_MAX_SCALES; ++t)
        if (t < m.nscales && kk >= m.koff[t]) s = t;
    int koff = m.koff[0], k_true = m.k[0], kp = m.kp[0], nchunks = m.nchunks[0];
    const float *part = m.part[0], *asum_part = m.asum[0], *w2 = m.w2[0];
#pragma unroll
    for (int t = 1; t < VLAD_MAX_SCALES; ++t)
        if (s == t) { koff = m.koff[t]; k_true = m.k[t]; kp = m.kp[t]; nchunks = m.nchunks[t]; part = m.part[t]; asum_part = m.asum[t]; w2 = m.w2[t]; }
    const int k = kk - koff;
    float v = 0.f, a = 0.f;
    // same chunk order as vlad_finalize_kernel (bit-identical rows), but sixteen partial rows are IN FLIGHT before the first add: the pass
    // is 33 MB of reads at B = 32 and a dependent load -> add chain per chunk ran at a third of the cache bandwidth
    for (int c0 = 0; c0 < nchunks; c0 += 16) {
        float pv[16], pa[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const bool live = c0 + u < nchunks;
            const size_t row = ((size_t)b * nchunks + (live ? c0 + u : c0)) * kp + k;
            pv[u] = live ? part[row * VC + c] : 0.f;
            pa[u] = live ? asum_part[row] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (c0 + u < nchunks) { v += pv[u]; a += pa[u]; }
    }
    v = v - a * w2[(size_t)c * k_true + k];          // loupe.py:213-219
    float ss = v * v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) ss += __shfl_xor(ss, o);
    if ((c & 63) == 0) red[c >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    out[((size_t)b * ktot + kk) * VC + c] = v / fmaxf(nrm, 1e-12f);        // F.normalize(dim=1), loupe.py:221
}

// APFA head, stage 1.  grid (ktot, NS column slices of 64, batch tiles of 32): for cluster row k and a tile of <= 32 clouds
//   logits[b][o]  = sum_c v[b][k][c] * Watt_t[c][o]                    o in the slice  -> mpart[b][k][slice] = max_o      (loupe.py:33-34)
//   P[b][k][n]    = sum_c relu(v[b][k][c]) * Wfc[k*256 + c][n]         n in the slice
// The reference computes y = relu(x + x*w[k]) and desc = y_flat . Wfc (loupe.py:36-38, :60-62).  1 + w[k] > 0 (w is a soft-max output), so
// y[k][c] = (1 + w[k]) * relu(v[k][c]) and desc = sum_k (1 + w[k]) * P[k]: P does not depend on the attention weights, so the 22 MB FC
// weight is streamed ONCE, here, by the same workgroups that compute the attention logits, and the soft-max over clusters only scales
// 84 rows of 256 afterwards (stage 2).  Five launches (logits, re-weight, split-K FC, finalize: 53 us at B = 32) become two.
// The contraction (256 channels) is split over the four waves (64 channels each) and reduced through LDS; operands are fetched 16 bytes
// per lane: a lane's float4 of v feeds four k-steps (k-step (j, e) of k-group q contracts channel 16 j + 4 q + e), its float4 of W is four
// consecutive columns of one k row (MFMA tile t holds column 4 i + t in its column i).
constexpr int HB = 32;                 // clouds per tile
constexpr int HAS = VC + 4;            // LDS row stride of the v tile (16-byte aligned rows)
// grid (ktot, 4 attention slices + nout / 64 FC slices, batch tiles): ONE (cluster, 64-column slice, pass) unit per workgroup -- 672 equal
// units of 128 MFMAs per wave at B = 32, 33 KB of LDS each, all resident at once (a workgroup doing both passes of a slice was 336 units of
// twice the work: the CUs that drew two of them set the kernel's time).
__global__ __launch_bounds__(256, 2) void afa_cluster_kernel(int bsz, int ktot, int nout, const float *__restrict__ vt,   // (B, ktot, VC)
                                                               const float *__restrict__ watt_t,                            // (VC, VC) K-major (c, o)
                                                               const float *__restrict__ fc_wt,                             // (ktot*VC, nout) rows k*VC + c
                                                               float *__restrict__ mpart,                                   // (B, ktot, 4)
                                                               float *__restrict__ P)                                       // (B, ktot, nout)
{
    __shared__ __attribute__((aligned(16))) float smem[HB * HAS];      // the v tile; after the MFMAs the four waves' partial tiles (4 x 32 x 64 floats)
    float *As = smem, *red = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    const int k = blockIdx.x, b0 = blockIdx.z * HB;
    const bool fc = blockIdx.y >= VC / 64;                             // block-uniform: attention logits (raw v) or FC partial (relu(v))
    const int y = fc ? blockIdx.y - VC / 64 : blockIdx.y;
    const int cnt = min(HB, bsz - b0);
    // A workgroup is one dependent chain (operands -> MFMAs -> reduce -> store) and the FC weights come from HBM: every global load of the
    // chain -- the v tile and the weight fragments (16 x 16 bytes per lane) -- is issued before the first use: one memory latency, not two.
    float4 vreg[HB * (VC / 4) / 256];
#pragma unroll
    for (int u = 0; u < HB * (VC / 4) / 256; ++u) {
        const int q = tid + u * 256, r = q >> 6, p4 = q & 63;
        vreg[u] = r < cnt ? *reinterpret_cast<const float4 *>(vt + ((size_t)(b0 + r) * ktot + k) * VC + p4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float *w = fc ? fc_wt + (size_t)k * VC * nout + y * 64 : watt_t + y * 64;
    const int ldw = fc ? nout : VC;
    float4 bw[16];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            bw[jj * 4 + e] = *reinterpret_cast<const float4 *>(w + (size_t)(16 * (wave * 4 + jj) + 4 * lq + e) * ldw + 4 * li);
#pragma unroll
    for (int u = 0; u < HB * (VC / 4) / 256; ++u) {
        const int q = tid + u * 256, r = q >> 6, p4 = q & 63;
        float4 v = vreg[u];
        if (fc) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        *reinterpret_cast<float4 *>(As + r * HAS + p4 * 4) = v;
    }
    __syncthreads();
    floatx4 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[rt][t] = (floatx4){0.f, 0.f, 0.f, 0.f};
    // the contraction (256 channels) is split over the four waves, 64 channels each
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        float4 a4[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) a4[rt] = *reinterpret_cast<const float4 *>(As + (rt * 16 + li) * HAS + 16 * (wave * 4 + jj) + 4 * lq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float4 b4 = bw[jj * 4 + e];
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float av = e == 0 ? a4[rt].x : e == 1 ? a4[rt].y : e == 2 ? a4[rt].z : a4[rt].w;
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[rt][t], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                     // every wave has read its A fragments: the tile becomes the reduction buffer
    // this wave's partial: row rt*16 + 4 lq + r, columns 4 li .. 4 li + 3 (tile t = column 4 li + t)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4 *>(red + (size_t)wave * HB * 64 + (rt * 16 + 4 * lq + r) * 64 + 4 * li) =
                make_float4(acc[rt][0][r], acc[rt][1][r], acc[rt][2][r], acc[rt][3][r]);
    __syncthreads();
    // thread (row = tid / 8, column group = tid % 8): 8 columns, summed over the four waves in wave order
    const int row = tid >> 3, cg = tid & 7;
    float o[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 sum = *reinterpret_cast<const float4 *>(red + row * 64 + cg * 8 + h * 4);
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) {
            const float4 t4 = *reinterpret_cast<const float4 *>(red + (size_t)wv * HB * 64 + row * 64 + cg * 8 + h * 4);
            sum.x += t4.x; sum.y += t4.y; sum.z += t4.z; sum.w += t4.w;
        }
        o[h * 4] = sum.x; o[h * 4 + 1] = sum.y; o[h * 4 + 2] = sum.z; o[h * 4 + 3] = sum.w;
    }
    if (!fc) {
        float mx = fmaxf(fmaxf(fmaxf(o[0], o[1]), fmaxf(o[2], o[3])), fmaxf(fmaxf(o[4], o[5]), fmaxf(o[6], o[7])));
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2)); mx = fmaxf(mx, __shfl_xor(mx, 4));
        if (cg == 0 && row < cnt) mpart[((size_t)(b0 + row) * ktot + k) * 4 + y] = mx;
    } else if (row < cnt) {
        float *dst = P + ((size_t)(b0 + row) * ktot + k) * nout + y * 64 + cg * 8;
        *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4 *>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
}

// APFA head, stage 2.  grid B x 1024 threads: w = softmax_k(max over the 4 slices of mpart) (loupe.py:34-36); desc = BN(sum_k (1 + w[k]) P[k] +
// bias) (loupe.py:60-62), optional F.normalize (:63-64).  nout <= 1024.  The sum over the ktot rows is pure load latency: thread group
// g = tid / 256 takes the rows k = g mod 4 with independent accumulators and the groups meet in LDS (fixed order: deterministic).
__global__ __launch_bounds__(1024) void afa_combine_kernel(int ktot, int nout, const float *__restrict__ mpart, const float *__restrict__ P,
                                                             const float *__restrict__ fc_bias, const float *__restrict__ scale, const float *__restrict__ shift,
                                                             int l2norm, float *__restrict__ desc)
{
    __shared__ float wk[256];
    __shared__ float red[32];
    __shared__ float grp[3 * 1024];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = tid >> 8, t = tid & 255;
    const float *Pb = P + (size_t)b * ktot * nout;
    // the P rows do not depend on the soft-max: start the first loads of every thread before the reductions
    float m = -3.0e38f;
    if (tid < ktot) {
        const float4 mp = *reinterpret_cast<const float4 *>(mpart + ((size_t)b * ktot + tid) * 4);
        m = fmaxf(fmaxf(mp.x, mp.y), fmaxf(mp.z, mp.w));
    }
    float mx = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));       // ktot <= 256: waves 0..3 hold every live row
    const float e = tid < ktot ? __expf(m - mx) : 0.f;
    float s = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[16 + wave] = s;
    __syncthreads();
    s = (red[16] + red[17]) + (red[18] + red[19]);
    if (tid < 256) wk[tid] = 1.0f + e / s;
    __syncthreads();
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int n = t + u * 256;
        if (n < nout) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int k = g;
            for (; k + 12 < ktot; k += 16) {
                a0 = fmaf(wk[k], Pb[(size_t)k * nout + n], a0);
                a1 = fmaf(wk[k + 4], Pb[(size_t)(k + 4) * nout + n], a1);
                a2 = fmaf(wk[k + 8], Pb[(size_t)(k + 8) * nout + n], a2);
                a3 = fmaf(wk[k + 12], Pb[(size_t)(k + 12) * nout + n], a3);
            }
            for (; k < ktot; k += 4) a0 = fmaf(wk[k], Pb[(size_t)k * nout + n], a0);
            v[u] = (a0 + a1) + (a2 + a3);
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (u * 256 >= nout) break;                                   // block-uniform
        if (g > 0) grp[(g - 1) * 1024 + u * 256 + t] = v[u];
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = t + u * 256;
            if (n < nout) {
                const float acc = ((v[u] + grp[u * 256 + t]) + grp[1024 + u * 256 + t]) + grp[2048 + u * 256 + t];
                v[u] = (acc + fc_bias[n]) * scale[n] + shift[n];
                ss += v[u] * v[u];
            }
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) red[wave] = ss;
    }
    __syncthreads();
    if (g == 0) {
        const float inv = l2norm ? 1.0f / fmaxf(sqrtf((red[0] + red[1]) + (red[2] + red[3])), 1e-12f) : 1.0f;     // F.normalize
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t + u * 256 < nout) desc[(size_t)b * nout + t + u * 256] = v[u] * inv;
    }
}

}  // namespace

// The whole NetVLAD pyramid (loupe.py:191-222 per scale, :301-303 concatenation) -> out (b, ktot, 256) cluster-major rows, ktot = sum k[s].
// Arrays of nscales (<= 4) entries, coarse to fine; x[s] (b, n[s], 256) point-major; wc_t / wc_p / bias / w2 as for pa_netvlad_rows (wc_p[s]
// NULL except for k[s] > 48); scratch[s]: pa_netvlad_scratch_floats(b, n[s], k[s]) floats.  Launches: one accumulate kernel per scale, then ONE
// finalize launch for every scale (pa_netvlad_rows per scale: 2 launches each; 21.7 -> 18.6 us for the three finalize passes at B = 32).
// Measured and dropped: the <= 16-cluster scales sharing one accumulate launch -- 576 workgroups of 70 KB of LDS on the chip's 512 two-per-CU
// slots run as two rounds (28.4 us against 13.7 + 7.7 us for the two launches).  Results are bit-identical to pa_netvlad_rows.
// phases: bit 0 = the accumulate launches of the scales with <= 16 clusters, bit 1 = the accumulate launches of the other scales,
// bit 2 = the finalize launch (7 = everything).  A caller whose coarse feature maps are ready early (the decoder writes them first) issues
// phase 1 right behind their producer, while they are still cache-resident, and phases 2 | 4 after the finest level; x[s] of a scale that
// the requested phases do not read may be NULL.
static int netvlad_pyramid(int b, int nscales, const int *n, const int *k, const float *const *x, const float *const *wc_t, const float *const *wc_p,
                           const void *const *wc16, const float *const *bias, const float *const *w2, float *const *scratch, float *out, int phases,
                           pa_stream_t stream, int x16_mask = 0)
{
    PA_REQUIRE(b > 0 && b <= 65535 && nscales > 0 && nscales <= VLAD_MAX_SCALES && n && k && x && wc_t && bias && w2 && scratch && (out || !(phases & 4)) && (phases & 7),
               "pa_netvlad_pyramid: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    VladFinalizeMulti fm;
    memset(&fm, 0, sizeof(fm));
    fm.nscales = nscales;
    int ktot = 0;
    for (int s = 0; s < nscales; ++s) {
        const int kp = (k[s] + 15) & ~15, kt = kp / 16;
        int chunks = vlad_chunks(n[s]), rows = vlad_rows_per_wg(n[s]);
        // the 16-bit kernel is one eight-wave workgroup per CU: 512 rows each = one round of 256 workgroups at B = 32 and half the partial volume
        // (16.7 instead of 33 MB; the scratch is sized for the smaller chunks).  PA_VLAD16_ROWS = A/B knob
        static const int rows16 = getenv("PA_VLAD16_ROWS") ? atoi(getenv("PA_VLAD16_ROWS")) : 512;
        if (kt == 4 && wc16 && wc16[s] && n[s] >= 2048 && rows16 >= rows && rows16 % V16_ROWS == 0) { rows = rows16; chunks = (n[s] + rows - 1) / rows; }
        const bool run = kp == 16 ? (phases & 1) : (phases & 2);
        PA_REQUIRE(n[s] > 0 && k[s] > 0 && k[s] <= 64 && (x[s] || !run) && wc_t[s] && bias[s] && w2[s] && scratch[s],
                   "pa_netvlad_pyramid: scale %d: bad arguments (<= 64 clusters)", s);
        float *part = scratch[s], *asum = scratch[s] + (size_t)b * chunks * kp * VC;
        fm.koff[s] = ktot; fm.k[s] = k[s]; fm.kp[s] = kp; fm.nchunks[s] = chunks; fm.part[s] = part; fm.asum[s] = asum; fm.w2[s] = w2[s];
        ktot += k[s];
        if (!run) continue;
        const size_t lds = (size_t)(VROWS * VC + VROWS * (kt == 4 ? kp : kp + 2)) * 4;
        const float *wp = (wc_p && k[s] > 48) ? wc_p[s] : nullptr;
#define PA_VLAD_LAUNCH(KT)                                                                                                              \
    do {                                                                                                                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vlad_accum_kernel<KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(vlad_accum_kernel<KT>, dim3(chunks, b), dim3(256), lds, st, n[s], k[s], rows, x[s], wc_t[s], wp, bias[s], part, asum);  \
    } while (0)
        if (kt == 4 && wc16 && wc16[s] && ((x16_mask >> s) & 1)) {   // fp16 path, fp16 feature map
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vlad_accum16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)V16_LDS);
            hipLaunchKernelGGL(vlad_accum16_kernel<true>, dim3(chunks, b), dim3(512), V16_LDS, st, n[s], k[s], rows, x[s], reinterpret_cast<const _Float16 *>(wc16[s]),
                               bias[s], part, asum);
        } else if (kt == 4 && wc16 && wc16[s]) {   // fp16 path: both contractions on the fp16 MFMA, same partials
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&vlad_accum16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)V16_LDS);
            hipLaunchKernelGGL(vlad_accum16_kernel<false>, dim3(chunks, b), dim3(512), V16_LDS, st, n[s], k[s], rows, x[s], reinterpret_cast<const _Float16 *>(wc16[s]),
                               bias[s], part, asum);
        } else if (kt == 1) PA_VLAD_LAUNCH(1);
        else if (kt == 2) PA_VLAD_LAUNCH(2);
        else if (kt == 3) PA_VLAD_LAUNCH(3);
        else PA_VLAD_LAUNCH(4);
#undef PA_VLAD_LAUNCH
    }
    fm.koff[nscales] = ktot;
    for (int s = nscales; s < VLAD_MAX_SCALES; ++s) fm.koff[s + 1] = ktot;
    if (phases & 4) hipLaunchKernelGGL(vlad_finalize_multi_kernel, dim3(ktot, b), dim3(256), 0, st, fm, out, ktot);
    PA_CHECK_LAUNCH("pa_netvlad_pyramid");
    return PA_OK;
}

PA_API int pa_netvlad_pyramid(int b, int nscales, const int *n, const int *k, const float *const *x, const float *const *wc_t, const float *const *wc_p,
                              const float *const *bias, const float *const *w2, float *const *scratch, float *out, int phases, pa_stream_t stream)
{
    return netvlad_pyramid(b, nscales, n, k, x, wc_t, wc_p, nullptr, bias, w2, scratch, out, phases, stream);
}

// The same with fp16 operands at the scales that have 49..64 clusters: wc16[s] = pa_pack_weights_f16(256, 64, wc_t[s]) there (NULL elsewhere: those
// scales run the fp32 kernel; 32 768 halfs: the packing of W, then of W - fp16(W)).  The logits keep fp32 accuracy ((hi, lo) operand pairs), the
// aggregation takes features and assignments rounded to fp16, accumulation and everything else fp32:
// part of the model's fp16 path (descriptors within cosine 0.999 of the fp32 path, not 1e-4).
PA_API int pa_netvlad_pyramid_f16(int b, int nscales, const int *n, const int *k, const float *const *x, const float *const *wc_t, const float *const *wc_p,
                                  const void *const *wc16, const float *const *bias, const float *const *w2, float *const *scratch, float *out, int phases,
                                  pa_stream_t stream)
{
    PA_REQUIRE(wc16, "pa_netvlad_pyramid_f16: null wc16");
    return netvlad_pyramid(b, nscales, n, k, x, wc_t, wc_p, wc16, bias, w2, scratch, out, phases, stream);
}

// pa_netvlad_pyramid_f16 where the feature maps of the scales in x16_mask (bit s) are fp16 rows of 256 halfs (x[s] then points at halfs): only
// scales that run the fp16 kernel (wc16[s] != NULL, 49..64 clusters, >= 2048 points) may be flagged.  The producer is pa_fp_chain_premul_g16h.
PA_API int pa_netvlad_pyramid_f16h(int b, int nscales, const int *n, const int *k, const float *const *x, const float *const *wc_t, const float *const *wc_p,
                                   const void *const *wc16, const float *const *bias, const float *const *w2, float *const *scratch, float *out, int phases,
                                   int x16_mask, pa_stream_t stream)
{
    PA_REQUIRE(wc16, "pa_netvlad_pyramid_f16h: null wc16");
    for (int s = 0; s < nscales && s < VLAD_MAX_SCALES; ++s)
        PA_REQUIRE(!((x16_mask >> s) & 1) || (wc16[s] && ((k[s] + 15) & ~15) == 64), "pa_netvlad_pyramid_f16h: scale %d is not an fp16-kernel scale", s);
    return netvlad_pyramid(b, nscales, n, k, x, wc_t, wc_p, wc16, bias, w2, scratch, out, phases, stream, x16_mask);
}

PA_API long pa_afa_fused_scratch_floats(int b, int ktot, int nout) { return (long)b * ktot * 4 + (long)b * ktot * nout; }

// Cluster-major APFA head in two launches (afa_cluster_kernel / afa_combine_kernel above).  vt (b, ktot, 256) from pa_netvlad_pyramid /
// pa_netvlad_rows; watt_t: the attention conv as a K-major (in, out) matrix; fc_wt: K-major (ktot*256, nout) with rows ordered k*256 + c;
// scratch: pa_afa_fused_scratch_floats(b, ktot, nout) floats.  Same function as pa_afa_rows; the FC sum is re-associated (per-cluster
// partial sums scaled by 1 + w[k]), so descriptors agree to fp32 rounding, not bit for bit.
PA_API int pa_afa_fused(int b, int c, int ktot, int nout, const float *vt, const float *watt_t, const float *fc_wt, const float *fc_bias,
                        const float *scale, const float *shift, int l2norm, float *scratch, float *desc, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && ktot > 0 && nout > 0 && vt && watt_t && fc_wt && fc_bias && scale && shift && scratch && desc, "pa_afa_fused: bad arguments");
    if (c != VC || ktot > 256 || nout % 64 || nout > 1024 || ktot > 65535) {
        pa_set_error("pa_afa_fused: built for 256 channels, <= 256 clusters, nout %% 64 == 0, nout <= 1024 (got c=%d ktot=%d nout=%d)", c, ktot, nout);
        return PA_EUNSUPPORTED;
    }
    PA_REQUIRE(((uintptr_t)vt & 15) == 0 && ((uintptr_t)watt_t & 15) == 0 && ((uintptr_t)fc_wt & 15) == 0 && ((uintptr_t)scratch & 15) == 0,
               "pa_afa_fused: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    float *mpart = scratch, *P = scratch + (size_t)b * ktot * 4;
    hipLaunchKernelGGL(afa_cluster_kernel, dim3(ktot, VC / 64 + nout / 64, (b + HB - 1) / HB), dim3(256), 0, st, b, ktot, nout, vt, watt_t, fc_wt, mpart, P);
    hipLaunchKernelGGL(afa_combine_kernel, dim3(b), dim3(1024), 0, st, ktot, nout, mpart, P, fc_bias, scale, shift, l2norm, desc);
    PA_CHECK_LAUNCH("pa_afa_fused");
    return PA_OK;
}

#ifdef PA_VLAD_DEBUG
PA_API int pa_vlad_debug_read(long long *host16)
{
    return hipMemcpyFromSymbol(host16, HIP_SYMBOL(vlad_stamps), sizeof(long long) * 16) == hipSuccess ? PA_OK : PA_EINVAL;
}
#endif
