// Training-mode dense path on the MFMA pipe (gfx950, exact fp32 MFMA): the building blocks of the hand-written forward AND backward of
//   pt_util.SharedMLP in train() mode   utils/model_util/pt_util.py:16-41, :98-152   (1x1 conv without bias -> BatchNorm(batch statistics) -> ReLU)
//   PointNetDecoder                     place_recognition/patch_aug_net/models/pointnet_autoencoder.py:85-111 (Linear + BatchNorm1d + ReLU, x2; Linear + tanh)
// as driven by the training step (place_recognition/train_place_recognition.py:142-169, :386-392).  The reference runs every layer as
// cuDNN/cuBLAS conv + a BatchNorm kernel + a ReLU kernel (and their three backward kernels), each a full pass over the activation.
//
// MI355X plan.  Activations stay in the reference's CHANNEL-MAJOR layout (B, C, P) (P = points, or m*k grouped points), so a 1x1 conv is,
// per cloud, Y (O x P) = W (O x C) . X (C x P): the big operand is read in whole contiguous rows.  BatchNorm in train mode needs global
// statistics between layers, so a layer is one launch, but everything elementwise rides inside the GEMMs:
//   * forward  (tgemm_nn): the B-operand loader applies the PREVIOUS layer's BatchNorm + ReLU on the fly (per-channel scale/shift), the
//     epilogue accumulates this layer's per-channel sum / sum of squares (fp64 atomics) -- the normalised activation is never written;
//   * backward dX (tgemm_nn): the B-operand loader turns (dZ, raw Y) into the BatchNorm/ReLU input gradient on the fly
//         dY = (mask(dZ) - mean(mask dZ) - xhat * mean(mask dZ * xhat)) * gamma * rstd ,
//   * backward dW (tgemm_kk): both operands are built on the fly (dY as above, the layer input as BN+ReLU of the previous raw output) and
//     contracted over all points of all clouds with split-K partial tiles combined by fp32 atomics;
//   * two thin HBM-bound passes remain per layer: the statistics finalize (a few hundred channels) and the dZ reduction that BatchNorm's
//     backward needs before any dY can be formed.
// Kernels are LDS-tiled (BM x 128 x 16, register-staged double buffer, one barrier per k-tile), v_mfma_f32_16x16x4_f32.
#include <stdlib.h>
#include <string.h>

#include "pa_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------------------------------------ operand transforms
enum { TF_NONE = 0, TF_AFFINE_RELU = 1, TF_BN_BWD_RELU = 2, TF_BN_BWD = 3 };

struct TOp {
    int mode;
    const float *aux;   // modes 2/3: the layer's raw (pre-BatchNorm) output, same layout as the operand
    const float *p;     // per-channel parameters, SoA p[j * nch + ch]: mode 1: j = 0 scale, 1 shift;
                        // modes 2/3: 0 scale (gamma*rstd), 1 shift, 2 mean, 3 rstd, 4 mean(mask g), 5 mean(mask g * xhat), 6 gamma*rstd
    int nch;
};

struct ChanP { float v[7]; };

template <int MODE>
__device__ __forceinline__ ChanP load_chan(const TOp &t, int ch, bool live)
{
    ChanP c;
    constexpr int NP = MODE == TF_NONE ? 0 : (MODE == TF_AFFINE_RELU ? 2 : 7);
#pragma unroll
    for (int j = 0; j < 7; ++j) c.v[j] = (j < NP && live) ? t.p[(size_t)j * t.nch + ch] : 0.f;
    return c;
}

// the same parameters fetched UNCONDITIONALLY from a clamped channel (no branch around the loads; the caller masks the result)
template <int MODE>
__device__ __forceinline__ ChanP load_chan_clamped(const TOp &t, int ch)
{
    ChanP c;
    constexpr int NP = MODE == TF_NONE ? 0 : (MODE == TF_AFFINE_RELU ? 2 : 7);
    const int cc = min(ch, t.nch - 1);
#pragma unroll
    for (int j = 0; j < 7; ++j) c.v[j] = j < NP ? t.p[(size_t)j * t.nch + cc] : 0.f;
    return c;
}

template <int MODE>
__device__ __forceinline__ float tf_apply(float g, float y, const ChanP &c)
{
    if (MODE == TF_NONE) return g;
    if (MODE == TF_AFFINE_RELU) return fmaxf(fmaf(g, c.v[0], c.v[1]), 0.f);
    const float z = fmaf(y, c.v[0], c.v[1]);
    const float gm = (MODE == TF_BN_BWD || z > 0.f) ? g : 0.f;
    const float xhat = (y - c.v[2]) * c.v[3];
    return (gm - c.v[4] - xhat * c.v[5]) * c.v[6];
}

// ------------------------------------------------------------------------------------------------ tgemm_nn
// C_b (M x N) = [beta * C_b +] act( A_b (M x K) . f(B_b) (K x N) + bias[m] ),   B and C n-contiguous, f = per-k (channel) transform.
struct NNArgs {
    int M, N, K;
    const float *A; long sAb; int lda;   // A_KCONTIG: A(m,k) = A[m*lda + k]; else A(m,k) = A[k*lda + m].  sAb = 0: shared by the batch
    const float *B; long sBb; int ldb;
    TOp tb;
    float *C; long sCb; int ldc;
    int beta;
    const float *bias;                   // per m, or null
    const float *colv;                   // act 2: per n
    int act;                             // 0 none, 1 tanh, 2 squared distance: max(bias[m] + colv[n] - 2 acc, 0)
    double *stats;                       // [PA_BN_STAT_SLOTS][2*M]: sum and sum of squares of the stored values per row m over (batch, n); or null
    long sPb, sStatb;                    // per-batch strides of tb.p (floats) and stats (doubles); 0 = one block shared by the batch
    int vecA, vecB;                      // 16-byte loads allowed (alignment / divisibility checked on the host)
    int xcd_order;                       // XCD-contiguous tile order (xcd_contiguous_id)
    int vecC;                            // 16-byte stores of C allowed (alignment, N % 4 == 0)
};

#ifndef PA_TGEMM_VEC_STORE
#define PA_TGEMM_VEC_STORE 1
#endif
#ifndef PA_TGEMM_PREFETCH2
#define PA_TGEMM_PREFETCH2 0
#endif
#ifndef PA_TGEMM_LDS_AHEAD
#define PA_TGEMM_LDS_AHEAD 1
#endif

constexpr int NN_BN = 128;

// Workgroups are dealt to the eight XCDs round-robin in launch order (id % 8), and every XCD has its own L2.  This maps the launch-order id to a
// tile id such that an XCD owns a CONTIGUOUS range of tile ids: tiles that share an operand (the M tiles of one column block; the tiles of one
// split of a dW GEMM) are then neighbours in time on one XCD and the shared operand is fetched into that L2 once instead of once per XCD.
// (The tail of a launch whose size is not a multiple of eight keeps its order.)  PA_TGEMM_XCD_ORDER = bit 0: tgemm_nn, bit 1: tgemm_kk (A/B knob).
__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned id, unsigned total)
{
    const unsigned t8 = total & ~7u;
    return id < t8 ? (id & 7u) * (t8 >> 3) + (id >> 3) : id;
}

// NN_BK = 32 halves the number of (barrier, fetch) rounds of a long contraction: with few workgroups per CU a round is bound by the
// latency of its global loads, not by its MFMAs; 16 stays for the short contractions (K <= 16: first layers, cluster counts).
// VA / VB: 16-byte global loads of the A / B operand (host-checked alignment and divisibility) -- compile-time, so that the k-loop has no
// branch between its loads (a run-time flag made the compiler wait for all outstanding loads at every block boundary).
template <int BM, int NN_BK, bool A_KCONTIG, int BMODE, bool VA, bool VB>
__global__ __launch_bounds__(256) void tgemm_nn_kernel(NNArgs a)
{
    constexpr int SA = BM + 16, SB = NN_BN + 16;          // row strides = 16 mod 32 floats: conflict-free fragment reads (rows k, k+1 per 32-lane half)
    constexpr int WM = BM == 128 ? 2 : 1, WN = 4 / WM;     // wave grid
    constexpr int MT = BM / WM / 16, NT = NN_BN / WN / 16; // 16x16 tiles per wave
    constexpr int AE = BM * NN_BK / 256;                   // A elements staged per thread (4 or 8)
    __shared__ __attribute__((aligned(16))) float As[2][NN_BK * SA];
    __shared__ __attribute__((aligned(16))) float Bs[2][NN_BK * SB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // tile order: M tiles fastest (they share the column block of B, the big operand), then column blocks, then clouds
    unsigned tix = blockIdx.x, tiy = blockIdx.y, tiz = blockIdx.z;
    if (a.xcd_order) {
        const unsigned t = xcd_contiguous_id(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
        tiy = t % gridDim.y;
        tix = (t / gridDim.y) % gridDim.x;
        tiz = t / (gridDim.y * gridDim.x);
    }
    const int n0 = tix * NN_BN, m0 = tiy * BM, b = tiz;
    const float *A = a.A + (size_t)b * a.sAb;
    const float *B = a.B + (size_t)b * a.sBb;
    const float *B2 = (BMODE >= TF_BN_BWD_RELU) ? a.tb.aux + (size_t)b * a.sBb : nullptr;
    float *C = a.C + (size_t)b * a.sCb;
    a.tb.p += (size_t)b * a.sPb;

    floatx4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};

    // Operand staging, global -> registers -> (transform) -> LDS.  fetch() only ISSUES loads: every load is unconditional from a clamped
    // address (out-of-range elements are zeroed by a select in stash()), and the BatchNorm / ReLU transform of the B operand is applied in
    // stash(), after the MFMAs of the current tile -- so the loads of tile kt + 1 are in flight during the whole MFMA block of tile kt.
    // (Before: predicated loads compiled to a branch per load and the transform right behind them put s_waitcnt vmcnt(0) BEFORE the
    // MFMAs: the global latency of every k-tile was exposed in every wave.)
    // Register stage(s) of the operand loads.  PA_TGEMM_PREFETCH2=1 (compile time; measured and NOT the default) requests tile kt + 2 while tile
    // kt is multiplied and tile kt + 1 is stashed: hipcc 7.2 then drains the load queue (s_waitcnt vmcnt(0)) at the top of the second half of
    // the unrolled loop and waits for the older stage in front of the MFMAs, and the second stage's registers cost a workgroup per CU: 122 ->
    // 159 us at the fp0 shape (tools/probes/tgemm_scale.py).
    constexpr int BP = NN_BK / 8;                           // B staging passes of 8 rows
    struct Stage {
        float ra[AE];
        unsigned fa;                                       // validity bits of ra
        float rb[BP * 4], ry[BP * 4];
        ChanP rcp[BP];
        unsigned fb;                                       // validity bits of rb
    };
    Stage S0;
#if PA_TGEMM_PREFETCH2
    Stage S1;
#endif
    // B staging: thread -> rows kb, kb + 8; 4 consecutive columns nb4
    const int kb = tid >> 5, nb4 = (tid & 31) * 4;
    const int Km1 = a.K - 1, Mm1 = a.M - 1, Nm1 = a.N - 1;
    auto fetch = [&](int k0, Stage &S) {
        float (&ra)[AE] = S.ra, (&rb)[BP * 4] = S.rb, (&ry)[BP * 4] = S.ry;
        ChanP (&rcp)[BP] = S.rcp;
        unsigned &fa = S.fa, &fb = S.fb;
        fa = 0; fb = 0;
        // ---- A tile
        if (A_KCONTIG) {                                   // 4 consecutive k per load: m = q / (BK/4), k4 = (q % (BK/4)) * 4
#pragma unroll
            for (int u = 0; u < AE / 4; ++u) {
                const int q = tid + u * 256, m = q / (NN_BK / 4), k4 = (q % (NN_BK / 4)) * 4;
                const int gm = m0 + m, gk = k0 + k4;
                const float *row = A + (size_t)min(gm, Mm1) * a.lda;
                if (VA) {                                  // K % 4 == 0 here (host): a group is all in or all out
                    const float4 v = *reinterpret_cast<const float4 *>(row + min(gk, a.K - 4));
                    ra[u * 4] = v.x; ra[u * 4 + 1] = v.y; ra[u * 4 + 2] = v.z; ra[u * 4 + 3] = v.w;
                    if (gm < a.M && gk < a.K) fa |= 0xfu << (u * 4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ra[u * 4 + e] = row[min(gk + e, Km1)];
                        if (gm < a.M && gk + e < a.K) fa |= 1u << (u * 4 + e);
                    }
                }
            }
        } else {                                           // 4 consecutive m per load: k = q / (BM/4), m4 = (q % (BM/4)) * 4
#pragma unroll
            for (int u = 0; u < AE / 4; ++u) {
                const int q = tid + u * 256, k = q / (BM / 4), m4 = (q % (BM / 4)) * 4;
                const int gm = m0 + m4, gk = k0 + k;
                const float *row = A + (size_t)min(gk, Km1) * a.lda;
                if (VA) {                                  // M % 4 == 0 here (host)
                    const float4 v = *reinterpret_cast<const float4 *>(row + min(gm, a.M - 4));
                    ra[u * 4] = v.x; ra[u * 4 + 1] = v.y; ra[u * 4 + 2] = v.z; ra[u * 4 + 3] = v.w;
                    if (gk < a.K && gm < a.M) fa |= 0xfu << (u * 4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ra[u * 4 + e] = row[min(gm + e, Mm1)];
                        if (gk < a.K && gm + e < a.M) fa |= 1u << (u * 4 + e);
                    }
                }
            }
        }
        // ---- B tile (raw; transformed in stash)
#pragma unroll
        for (int h = 0; h < BP; ++h) {
            const int gk = k0 + kb + h * 8, gn = n0 + nb4;
            const bool klive = gk < a.K;
            rcp[h] = load_chan_clamped<BMODE>(a.tb, gk);
            const size_t roff = (size_t)min(gk, Km1) * a.ldb;
            if (VB) {                                      // N % 4 == 0 here (host)
                const int cn = min(gn, a.N - 4);
                const float4 v = *reinterpret_cast<const float4 *>(B + roff + cn);
                rb[h * 4] = v.x; rb[h * 4 + 1] = v.y; rb[h * 4 + 2] = v.z; rb[h * 4 + 3] = v.w;
                if (BMODE >= TF_BN_BWD_RELU) {
                    const float4 w = *reinterpret_cast<const float4 *>(B2 + roff + cn);
                    ry[h * 4] = w.x; ry[h * 4 + 1] = w.y; ry[h * 4 + 2] = w.z; ry[h * 4 + 3] = w.w;
                }
                if (klive && gn < a.N) fb |= 0xfu << (h * 4);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cn = min(gn + e, Nm1);
                    rb[h * 4 + e] = B[roff + cn];
                    if (BMODE >= TF_BN_BWD_RELU) ry[h * 4 + e] = B2[roff + cn];
                    if (klive && gn + e < a.N) fb |= 1u << (h * 4 + e);
                }
            }
        }
    };
    auto stash = [&](int buf, Stage &S) {
        float (&ra)[AE] = S.ra, (&rb)[BP * 4] = S.rb, (&ry)[BP * 4] = S.ry;
        ChanP (&rcp)[BP] = S.rcp;
        const unsigned fa = S.fa, fb = S.fb;
        float *as = As[buf], *bs = Bs[buf];
#pragma unroll
        for (int e = 0; e < AE; ++e) ra[e] = (fa >> e) & 1u ? ra[e] : 0.f;
        if (A_KCONTIG) {
#pragma unroll
            for (int u = 0; u < AE / 4; ++u) {
                const int q = tid + u * 256, m = q / (NN_BK / 4), k4 = (q % (NN_BK / 4)) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) as[(k4 + e) * SA + m] = ra[u * 4 + e];
            }
        } else {
#pragma unroll
            for (int u = 0; u < AE / 4; ++u) {
                const int q = tid + u * 256, k = q / (BM / 4), m4 = (q % (BM / 4)) * 4;
                *reinterpret_cast<float4 *>(as + k * SA + m4) = make_float4(ra[u * 4], ra[u * 4 + 1], ra[u * 4 + 2], ra[u * 4 + 3]);
            }
        }
#pragma unroll
        for (int h = 0; h < BP; ++h) {
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = tf_apply<BMODE>(rb[h * 4 + e], BMODE >= TF_BN_BWD_RELU ? ry[h * 4 + e] : 0.f, rcp[h]);
                t[e] = (fb >> (h * 4 + e)) & 1u ? v : 0.f;
            }
            *reinterpret_cast<float4 *>(bs + (kb + h * 8) * SB + nb4) = make_float4(t[0], t[1], t[2], t[3]);
        }
    };

    const int nk = (a.K + NN_BK - 1) / NN_BK;
    // The MFMA block of one k-tile.  The operand fragments of k-step ks + 1 are read from LDS BEFORE the MFMAs of k-step ks are issued and the
    // scheduler is pinned to that order: left alone hipcc emits read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs, eight times per tile, i.e. every wave
    // pays the LDS round trip per 128 cycles of MFMA work (PA_TGEMM_LDS_AHEAD=0 keeps that form for A/B).
    auto multiply = [&](int cur) {
        const float *as = As[cur] + (lane >> 4) * SA + wm * (BM / WM) + (lane & 15);
        const float *bs = Bs[cur] + (lane >> 4) * SB + wn * (NN_BN / WN) + (lane & 15);
#if PA_TGEMM_LDS_AHEAD
        float af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = as[i * 16];
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[j] = bs[j * 16];
#pragma unroll
        for (int ks = 0; ks < NN_BK / 4; ++ks) {
            float an[MT], bn[NT];
            if (ks + 1 < NN_BK / 4) {
#pragma unroll
                for (int i = 0; i < MT; ++i) an[i] = as[(ks + 1) * 4 * SA + i * 16];
#pragma unroll
                for (int j = 0; j < NT; ++j) bn[j] = bs[(ks + 1) * 4 * SB + j * 16];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            if (ks + 1 < NN_BK / 4) {
                __builtin_amdgcn_sched_group_barrier(0x100, (MT + NT + 1) / 2, 0);      // the next k-step's reads (pairs merge into ds_read2_b32) ...
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);                // ... then this k-step's MFMAs
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = an[i];
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[j] = bn[j];
            }
        }
#else
#pragma unroll
        for (int ks = 0; ks < NN_BK / 4; ++ks) {
            float af[MT], bf[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = as[ks * 4 * SA + i * 16];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = bs[ks * 4 * SB + j * 16];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
#endif
    };
#if PA_TGEMM_PREFETCH2
    fetch(0, S0);
    if (nk > 1) fetch(NN_BK, S1);
    stash(0, S0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        if (kt + 2 < nk) fetch((kt + 2) * NN_BK, S0);       // S0 is free: tile kt was stashed from it
        multiply(0);
        if (kt + 1 < nk) stash(1, S1);
        __syncthreads();
        if (kt + 1 >= nk) break;
        if (kt + 3 < nk) fetch((kt + 3) * NN_BK, S1);
        multiply(1);
        if (kt + 2 < nk) stash(0, S0);
        __syncthreads();
    }
#else
    fetch(0, S0);
    stash(0, S0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
#ifndef PA_TGEMM_DBG_NO_LOADS                              // decomposition probes (never built into the library): the k-loop without its global
        if (kt + 1 < nk) fetch((kt + 1) * NN_BK, S0);      // loads / without its MFMAs -- tools/probes/tgemm_scale.py on a variant build
#endif
#ifndef PA_TGEMM_DBG_NO_MFMA
        multiply(cur);
#endif
#ifndef PA_TGEMM_DBG_NO_STASH
        if (kt + 1 < nk) stash(cur ^ 1, S0);
        __syncthreads();
#endif
    }
#endif

    // ---- epilogue: C/D layout: column n = l % 16, row m = 4 * (l / 16) + r
    // (vec_store: 64-row tiles with the plain epilogue and 16-byte-aligned output rows -- workgroup-uniform)
    const bool vec_store = PA_TGEMM_VEC_STORE && BM == 64 && NN_BK * (NN_BN + 16) * 2 >= 4 * 32 * 36 && a.act == 0 && !a.beta && a.vecC;
    float s1[MT][4], s2[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = m0 + wm * (BM / WM) + i * 16 + (lane >> 4) * 4 + r;
            const float bias = (a.bias && gm < a.M) ? a.bias[gm] : 0.f;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int gn = n0 + wn * (NN_BN / WN) + j * 16 + (lane & 15);
                if (gm < a.M && gn < a.N) {
                    float v = acc[i][j][r] + bias;
                    if (a.act == 1) v = tanhf(v);
                    else if (a.act == 2) { v = (bias + a.colv[gn]) - 2.f * acc[i][j][r]; v = v > 0.f ? v : 0.f; }
                    float *dst = C + (size_t)gm * a.ldc + gn;
                    if (a.beta) v += *dst;
#ifdef PA_TGEMM_DBG_NO_STORE
                    if (v == 1.2345e33f)                    // decomposition probe: the epilogue without its stores (the value still has to exist)
#endif
                    if (!vec_store) *dst = v;
                    if (vec_store) acc[i][j][r] = v;       // the stored value, for the transposed 16-byte store below
                    t1 += v;
                    t2 += v * v;
                }
            }
            s1[i][r] = t1;
            s2[i][r] = t2;
        }
    if (vec_store) {
        // 16-byte stores: the accumulator layout gives a lane ONE column of four rows (4-byte stores, 64 contiguous bytes per row and instruction:
        // ~12 us of the 124 us launch at the fp0 shape).  Each wave passes its 64 x 32 tile through a private 32 x 36-float slab of the B stage
        // (free after the loop's last barrier; a wave's LDS operations execute in order, no barrier needed) in two halves and writes rows of
        // 128 contiguous bytes, 8 lanes per row.
        float *slab = &Bs[0][0] + wave * (32 * 36);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < NT; ++j) slab[(ii * 16 + (lane >> 4) * 4 + r) * 36 + j * 16 + (lane & 15)] = acc[half * 2 + ii][j][r];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = it * 64 + lane, rl = idx >> 3, c4 = (idx & 7) * 4;
                const int gm = m0 + half * 32 + rl, gn = n0 + wn * 32 + c4;
                const float4 v = *reinterpret_cast<const float4 *>(slab + rl * 36 + c4);
                if (gm < a.M && gn < a.N) *reinterpret_cast<float4 *>(C + (size_t)gm * a.ldc + gn) = v;      // N % 4 == 0: a group of four is in or out
            }
        }
    }
    if (a.stats) {
        // per-row sums over this workgroup's columns: 16 lanes of a DPP row share a row m, the WN waves of a wave row meet in LDS (the
        // operand stage is free after the loop's last barrier), then ONE fp64 atomic pair per row and workgroup -- into one of
        // PA_BN_STAT_SLOTS replicas of the statistics block, so a few thousand workgroups do not queue up on 2*M addresses.
        float *red = &As[0][0];                    // [WN][BM][2]
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t1 = s1[i][r], t2 = s2[i][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
                const int lm = wm * (BM / WM) + i * 16 + (lane >> 4) * 4 + r;
                if ((lane & 15) == 0) { red[(wn * BM + lm) * 2] = t1; red[(wn * BM + lm) * 2 + 1] = t2; }
            }
        __syncthreads();
        if (tid < BM && m0 + tid < a.M) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) { t1 += red[(w * BM + tid) * 2]; t2 += red[(w * BM + tid) * 2 + 1]; }
            const unsigned slot = (tix + tiz * gridDim.x) % PA_BN_STAT_SLOTS;
            double *st = a.stats + (size_t)b * a.sStatb + (size_t)slot * 2 * a.M;
            atomicAdd(st + m0 + tid, (double)t1);
            atomicAdd(st + a.M + m0 + tid, (double)t2);
        }
    }
}


// ------------------------------------------------------------------------------------------------ tgemm_kk
// C (M x N) += sum over (batch, k) of fA(A_b)(m,k) * fB(B_b)(n,k); both operands k-contiguous: A_b(m,k) = A[b*sAb + m*lda + k].
// fA: per-m channel transform (none / bn-bwd, two tensors), fB: per-n channel transform (none / affine+relu).
// grid (N tiles, M tiles, batch * ksplits); partial tiles are combined with fp32 atomics (C zero-filled by the caller), or stored
// accumulated per batch when `per_batch` is set (C_b = C + b*sCb).
struct KKArgs {
    int M, N, K;
    const float *A; long sAb; int lda; TOp ta;
    const float *B; long sBb; int ldb; TOp tb;
    float *C; long sCb; int ldc;
    int ksplits, kchunk;
    int per_batch;
    int vecA, vecB;
    long sAPb, sBPb;                     // per-batch strides of ta.p / tb.p (floats); 0 = shared
    int xcd_order;
    int reps; long srep;                 // > 0: C is `reps` zero-filled replicas srep floats apart; workgroup z adds into replica z % reps (pa_tgemm_kk_rep)
};

constexpr int KK_BM = 64, KK_BN = 64, KK_BK = 32, KK_S = KK_BK + 2;   // row stride 34: (2m + k) mod 32 distinct for m < 16, k < 2

// VEC: 16-byte global loads of both operands (host: alignment, K and the split size multiples of 4); compile-time for a branch-free k-loop
template <int AMODE, int BMODE, bool VEC>
__global__ __launch_bounds__(256) void tgemm_kk_kernel(KKArgs a)
{
    __shared__ __attribute__((aligned(16))) float As[2][KK_BM * KK_S];
    __shared__ __attribute__((aligned(16))) float Bs[2][KK_BN * KK_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                 // 2 x 2 waves, 32 x 32 each
    unsigned tix = blockIdx.x, tiy = blockIdx.y, tiz = blockIdx.z;
    if (a.xcd_order) {                  // the tiles of one (cloud, k-split) share its two operand chunks: neighbours on one XCD
        const unsigned t = xcd_contiguous_id(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
        tix = t % gridDim.x;
        tiy = (t / gridDim.x) % gridDim.y;
        tiz = t / (gridDim.x * gridDim.y);
    }
    const int n0 = tix * KK_BN, m0 = tiy * KK_BM;
    const int b = tiz / a.ksplits, split = tiz % a.ksplits;
    const int kbeg = split * a.kchunk, kend = min(kbeg + a.kchunk, a.K);
    const float *A = a.A + (size_t)b * a.sAb, *B = a.B + (size_t)b * a.sBb;
    const float *A2 = AMODE >= TF_BN_BWD_RELU ? a.ta.aux + (size_t)b * a.sAb : nullptr;
    a.ta.p += (size_t)b * a.sAPb;
    a.tb.p += (size_t)b * a.sBPb;
    floatx4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
    // staging: 64 rows x 32 k = 512 float4 per operand: thread -> row r = q / 8, k4 = (q % 8) * 4, q = tid, tid + 256
    // fetch() only issues loads (unconditional, clamped addresses); masks and the BatchNorm transforms are applied in stash(), after the
    // MFMAs of the current tile (see tgemm_nn_kernel)
    float ra[8], ry[8], rb[8];
    unsigned fa = 0, fb = 0;
    ChanP ca[2], cb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = (tid + u * 256) >> 3;
        ca[u] = load_chan<AMODE>(a.ta, m0 + r, m0 + r < a.M);
        cb[u] = load_chan<BMODE>(a.tb, n0 + r, n0 + r < a.N);
    }
    const int Mm1 = a.M - 1, Nm1 = a.N - 1, kl1 = kend - 1;
    auto fetch = [&](int k0) {
        fa = 0; fb = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = tid + u * 256, r = q >> 3, k4 = (q & 7) * 4, gk = k0 + k4;
            const int gm = m0 + r, gn = n0 + r;
            const size_t aoff = (size_t)min(gm, Mm1) * a.lda, boff = (size_t)min(gn, Nm1) * a.ldb;
            if (VEC) {                                     // kend % 4 == 0: a group of four k is all in or all out
                const int ck = min(gk, kend - 4);
                const float4 v = *reinterpret_cast<const float4 *>(A + aoff + ck);
                ra[u * 4] = v.x; ra[u * 4 + 1] = v.y; ra[u * 4 + 2] = v.z; ra[u * 4 + 3] = v.w;
                if (AMODE >= TF_BN_BWD_RELU) {
                    const float4 w = *reinterpret_cast<const float4 *>(A2 + aoff + ck);
                    ry[u * 4] = w.x; ry[u * 4 + 1] = w.y; ry[u * 4 + 2] = w.z; ry[u * 4 + 3] = w.w;
                }
                const float4 x = *reinterpret_cast<const float4 *>(B + boff + ck);
                rb[u * 4] = x.x; rb[u * 4 + 1] = x.y; rb[u * 4 + 2] = x.z; rb[u * 4 + 3] = x.w;
                if (gm < a.M && gk < kend) fa |= 0xfu << (u * 4);
                if (gn < a.N && gk < kend) fb |= 0xfu << (u * 4);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ck = min(gk + e, kl1);
                    ra[u * 4 + e] = A[aoff + ck];
                    if (AMODE >= TF_BN_BWD_RELU) ry[u * 4 + e] = A2[aoff + ck];
                    rb[u * 4 + e] = B[boff + ck];
                    if (gm < a.M && gk + e < kend) fa |= 1u << (u * 4 + e);
                    if (gn < a.N && gk + e < kend) fb |= 1u << (u * 4 + e);
                }
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = tid + u * 256, r = q >> 3, k4 = (q & 7) * 4;
            float ta[4], tb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float va = tf_apply<AMODE>(ra[u * 4 + e], AMODE >= TF_BN_BWD_RELU ? ry[u * 4 + e] : 0.f, ca[u]);
                const float vb = tf_apply<BMODE>(rb[u * 4 + e], 0.f, cb[u]);
                ta[e] = (fa >> (u * 4 + e)) & 1u ? va : 0.f;
                tb[e] = (fb >> (u * 4 + e)) & 1u ? vb : 0.f;
            }
            float2 *da = reinterpret_cast<float2 *>(As[buf] + r * KK_S + k4);
            da[0] = make_float2(ta[0], ta[1]);
            da[1] = make_float2(ta[2], ta[3]);
            float2 *db = reinterpret_cast<float2 *>(Bs[buf] + r * KK_S + k4);
            db[0] = make_float2(tb[0], tb[1]);
            db[1] = make_float2(tb[2], tb[3]);
        }
    };
    const int nk = (kend - kbeg + KK_BK - 1) / KK_BK;
    if (nk <= 0) return;
    fetch(kbeg);
    stash(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) fetch(kbeg + (kt + 1) * KK_BK);
        const float *as = As[cur] + (wm * 32 + (lane & 15)) * KK_S + (lane >> 4);
        const float *bs = Bs[cur] + (wn * 32 + (lane & 15)) * KK_S + (lane >> 4);
#if PA_TGEMM_LDS_AHEAD
        // the fragments of k-step ks + 1 are read before the MFMAs of k-step ks are issued (see tgemm_nn_kernel's multiply)
        float a0 = as[0], a1 = as[16 * KK_S], b0 = bs[0], b1 = bs[16 * KK_S];
#pragma unroll
        for (int ks = 0; ks < KK_BK / 4; ++ks) {
            float na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
            if (ks + 1 < KK_BK / 4) {
                na0 = as[(ks + 1) * 4]; na1 = as[16 * KK_S + (ks + 1) * 4];
                nb0 = bs[(ks + 1) * 4]; nb1 = bs[16 * KK_S + (ks + 1) * 4];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
            if (ks + 1 < KK_BK / 4) {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
#else
#pragma unroll
        for (int ks = 0; ks < KK_BK / 4; ++ks) {
            const float a0 = as[ks * 4], a1 = as[16 * KK_S + ks * 4];
            const float b0 = bs[ks * 4], b1 = bs[16 * KK_S + ks * 4];
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
#endif
        if (kt + 1 < nk) stash(cur ^ 1);
        __syncthreads();
    }
    float *C = a.C + (a.per_batch ? (size_t)b * a.sCb : (size_t)0) + (a.reps > 0 ? (size_t)(tiz % (unsigned)a.reps) * a.srep : (size_t)0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r, gn = n0 + wn * 32 + j * 16 + (lane & 15);
                if (gm < a.M && gn < a.N) atomicAdd(C + (size_t)gm * a.ldc + gn, acc[i][j][r]);
            }
}

// ------------------------------------------------------------------------------------------------ BatchNorm helpers
// finalize: stats (fp64 sum, sum of squares per channel over `count` values) -> p[0] scale = gamma*rstd, p[1] shift = beta - mean*scale,
// p[2] mean, p[3] rstd (biased variance, as torch's training forward); running statistics updated in place with momentum and the
// unbiased variance (torch.nn.BatchNorm semantics).
__global__ void bn_finalize_kernel(int nch, int batch, double count, const double *__restrict__ stats, const float *__restrict__ gamma, const float *__restrict__ beta,
                                   float eps, float momentum, float *__restrict__ running_mean, float *__restrict__ running_var, float *__restrict__ p,
                                   long long *__restrict__ counter)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && counter) *counter += batch;      // num_batches_tracked: one forward pass per statistics group
    if (c >= nch) return;
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_mean ? running_var[c] : 0.f;
    for (int b = 0; b < batch; ++b) {            // batch > 1: one statistics group per batch entry, running statistics updated in order
        const double *st = stats + (size_t)b * PA_BN_STAT_SLOTS * 2 * nch;
        float *pb = p + (size_t)b * 7 * nch;
        double su = 0.0, sq = 0.0;
        for (int sl = 0; sl < PA_BN_STAT_SLOTS; ++sl) { su += st[(size_t)sl * 2 * nch + c]; sq += st[(size_t)sl * 2 * nch + nch + c]; }
        const double mean = su / count;
        double var = sq / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scale = g * rstd;
        pb[c] = scale;
        pb[nch + c] = bt - (float)mean * scale;
        pb[2 * nch + c] = (float)mean;
        pb[3 * nch + c] = rstd;
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rm = (1.f - momentum) * rm + momentum * (float)mean;
        rv = (1.f - momentum) * rv + momentum * (float)unbiased;
    }
    if (running_mean) { running_mean[c] = rm; running_var[c] = rv; }
}

// Evaluation-mode BatchNorm as the same parameter block: rows 0..3 from the RUNNING statistics (scale = gamma / sqrt(running_var + eps), shift = beta -
// running_mean * scale, mean, rstd), rows 4, 5 (the batch-statistics terms of the input gradient) zero, row 6 = scale; `groups` identical copies.
__global__ void bn_eval_params_kernel(int nch, int groups, const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ running_mean,
                                      const float *__restrict__ running_var, float eps, float *__restrict__ p)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch) return;
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float mean = running_mean[c];
    const float rstd = (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
    const float scale = g * rstd;
    for (int b = 0; b < groups; ++b) {
        float *pb = p + (size_t)b * 7 * nch;
        pb[c] = scale;
        pb[nch + c] = bt - mean * scale;
        pb[2 * nch + c] = mean;
        pb[3 * nch + c] = rstd;
        pb[4 * nch + c] = 0.f;
        pb[5 * nch + c] = 0.f;
        pb[6 * nch + c] = scale;
    }
}

// backward reduction over a channel-major (B, C, P) pair (g = gradient w.r.t. the post-activation, y = raw pre-BatchNorm output):
// sums[c] += sum mask(g), sums[C + c] += sum mask(g) * xhat   (fp64 atomics); grid (chunks of P, C, B)
template <bool RELU>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(int C, long P, const float *__restrict__ g, const float *__restrict__ y, const float *__restrict__ p,
                                                              double *__restrict__ sums, int vec, long sPb, long sSumb)
{
    __shared__ float red[8];
    const int c = blockIdx.y, b = blockIdx.z;
    p += (size_t)b * sPb;
    sums += (size_t)b * sSumb;
    const float scale = p[c], shift = p[C + c], mean = p[2 * C + c], rstd = p[3 * C + c];
    const size_t base = ((size_t)b * C + c) * P;
    const long chunk = (P + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * chunk, hi = min(lo + chunk, P);
    float s1 = 0.f, s2 = 0.f;
    if (vec && (lo & 3) == 0) {
        long i = lo + (long)threadIdx.x * 4;
        for (; i + 3 < hi; i += 1024) {
            const float4 gv = *reinterpret_cast<const float4 *>(g + base + i), yv = *reinterpret_cast<const float4 *>(y + base + i);
            const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, ya[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gm = (!RELU || fmaf(ya[e], scale, shift) > 0.f) ? ga[e] : 0.f;
                s1 += gm;
                s2 += gm * ((ya[e] - mean) * rstd);
            }
        }
        for (; i < hi; ++i) {   // the (at most 3) leftover elements of the thread that reaches the tail
            if (i >= lo + ((hi - lo) & ~3L)) {
                const float gm = (!RELU || fmaf(y[base + i], scale, shift) > 0.f) ? g[base + i] : 0.f;
                s1 += gm;
                s2 += gm * ((y[base + i] - mean) * rstd);
            } else break;
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += 256) {
            const float gm = (!RELU || fmaf(y[base + i], scale, shift) > 0.f) ? g[base + i] : 0.f;
            s1 += gm;
            s2 += gm * ((y[base + i] - mean) * rstd);
        }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s1; red[4 + (threadIdx.x >> 6)] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sums + c, (double)((red[0] + red[1]) + (red[2] + red[3])));
        atomicAdd(sums + C + c, (double)((red[4] + red[5]) + (red[6] + red[7])));
    }
}

// p[4] = sum mask(g) / count, p[5] = sum(mask(g) * xhat) / count, p[6] = gamma * rstd (= p[0]); dgamma = sum(mask g * xhat), dbeta = sum(mask g)
__global__ void bn_bwd_finalize_kernel(int nch, int batch, double count, const double *__restrict__ sums, float *__restrict__ p, float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch) return;
    double dg = 0.0, db = 0.0;
    for (int b = 0; b < batch; ++b) {            // batch > 1: one statistics group per batch entry; the parameter gradients add up
        const double *sm = sums + (size_t)b * 2 * nch;
        float *pb = p + (size_t)b * 7 * nch;
        pb[4 * nch + c] = (float)(sm[c] / count);
        pb[5 * nch + c] = (float)(sm[nch + c] / count);
        pb[6 * nch + c] = pb[c];
        dg += sm[nch + c];
        db += sm[c];
    }
    if (dgamma) dgamma[c] = (float)dg;
    if (dbeta) dbeta[c] = (float)db;
}

// out = relu(y * scale + shift) on (B, C, P); with pool > 0: max over groups of `pool` consecutive points -> out (B, C, P / pool), arg (int8 slot)
__global__ __launch_bounds__(256) void bn_apply_kernel(int C, long P, int pool, int relu, const float *__restrict__ y, const float *__restrict__ p, float *__restrict__ out,
                                                        signed char *__restrict__ arg, long sPb, long rows)
{
    const long Pout = pool > 0 ? P / pool : P;
    for (long row = blockIdx.y; row < rows; row += gridDim.y) {     // B * C rows, grid-stride over y (gridDim.y <= 65535)
    const int c = (int)(row % C);
    const float *pr = p + (size_t)(row / C) * sPb;
    const float scale = pr[c], shift = pr[C + c];
    const float *src = y + (size_t)row * P;
    float *dst = out + (size_t)row * Pout;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < Pout; j += (long)gridDim.x * 256) {
        if (pool <= 0) {
            const float v = fmaf(src[j], scale, shift);
            dst[j] = relu ? fmaxf(v, 0.f) : v;
        } else {
            float best = -INFINITY;
            int bi = 0;
            for (int s = 0; s < pool; ++s) {
                float v = fmaf(src[j * pool + s], scale, shift);
                if (relu) v = fmaxf(v, 0.f);
                if (v > best) { best = v; bi = s; }     // first maximum, like torch.max
            }
            dst[j] = best;
            arg[(size_t)row * Pout + j] = (signed char)bi;
        }
    }
    }
}

// The un-pooled form with 16-byte accesses (P % 4 == 0, aligned rows): P4 = P / 4
__global__ __launch_bounds__(256) void bn_apply_vec_kernel(int C, long P4, int relu, const float *__restrict__ y, const float *__restrict__ p, float *__restrict__ out, long sPb,
                                                            long rows)
{
    for (long row = blockIdx.y; row < rows; row += gridDim.y) {
        const int c = (int)(row % C);
        const float *pr = p + (size_t)(row / C) * sPb;
        const float scale = pr[c], shift = pr[C + c];
        const float4 *src = reinterpret_cast<const float4 *>(y) + (size_t)row * P4;
        float4 *dst = reinterpret_cast<float4 *>(out) + (size_t)row * P4;
        for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < P4; j += (long)gridDim.x * 256) {
            float4 v = src[j];
            v.x = fmaf(v.x, scale, shift); v.y = fmaf(v.y, scale, shift); v.z = fmaf(v.z, scale, shift); v.w = fmaf(v.w, scale, shift);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            dst[j] = v;
        }
    }
}

// The pooled form through LDS: a workgroup reads the 256 * pool consecutive inputs of 256 outputs with contiguous (16-byte when VEC) loads into rows of
// pool + 1 floats, then thread j walks its own row.  (The direct form above reads `pool` floats per thread at a stride of `pool` floats: 1.1 TB/s on
// the first set-abstraction level's 94 MB.)  Dynamic LDS: 256 * (pool + 1) floats.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_pool_lds_kernel(int C, long P, int pool, int relu, const float *__restrict__ y, const float *__restrict__ p,
                                                                 float *__restrict__ out, signed char *__restrict__ arg, long sPb, long rows)
{
    extern __shared__ float tile[];
    const long Pout = P / pool;
    const int st = pool + 1;
    for (long row = blockIdx.y; row < rows; row += gridDim.y) {
        const int c = (int)(row % C);
        const float *pr = p + (size_t)(row / C) * sPb;
        const float scale = pr[c], shift = pr[C + c];
        for (long j0 = (long)blockIdx.x * 256; j0 < Pout; j0 += (long)gridDim.x * 256) {
            const int cnt = (int)(Pout - j0 < 256 ? Pout - j0 : 256), nel = cnt * pool;
            const float *src = y + (size_t)row * P + j0 * pool;
            if (VEC) {
                for (int i = threadIdx.x; i < nel / 4; i += 256) {
                    float4 v = reinterpret_cast<const float4 *>(src)[i];
                    v.x = fmaf(v.x, scale, shift); v.y = fmaf(v.y, scale, shift); v.z = fmaf(v.z, scale, shift); v.w = fmaf(v.w, scale, shift);
                    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    const int e = 4 * i, r = e / pool;
                    float *d = tile + r * st + (e - r * pool);
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                }
            } else {
                for (int e = threadIdx.x; e < nel; e += 256) {
                    float v = fmaf(src[e], scale, shift);
                    if (relu) v = fmaxf(v, 0.f);
                    const int r = e / pool;
                    tile[r * st + (e - r * pool)] = v;
                }
            }
            __syncthreads();
            if ((int)threadIdx.x < cnt) {
                const float *t = tile + threadIdx.x * st;
                float best = -INFINITY;
                int bi = 0;
                for (int q = 0; q < pool; ++q)
                    if (t[q] > best) { best = t[q]; bi = q; }     // first maximum, like torch.max
                out[(size_t)row * Pout + j0 + threadIdx.x] = best;
                arg[(size_t)row * Pout + j0 + threadIdx.x] = (signed char)bi;
            }
            __syncthreads();
        }
    }
}

// gradient of the pooled output scattered back to the full (B, C, P) grid: g[j*pool + s] = (s == arg[j]) ? gp[j] : 0
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(long Pout, int pool, const float *__restrict__ gp, const signed char *__restrict__ arg, float *__restrict__ g, long rows)
{
    for (long row = blockIdx.y; row < rows; row += gridDim.y)
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < Pout; j += (long)gridDim.x * 256) {
        const float v = gp[(size_t)row * Pout + j];
        const int a = arg[(size_t)row * Pout + j];
        float *dst = g + ((size_t)row * Pout + j) * pool;
        for (int s = 0; s < pool; ++s) dst[s] = s == a ? v : 0.f;
    }
}

// pool % 4 == 0 and aligned rows: 16-byte stores
__global__ __launch_bounds__(256) void maxpool_bwd_vec_kernel(long Pout, int pool, const float *__restrict__ gp, const signed char *__restrict__ arg, float *__restrict__ g, long rows)
{
    for (long row = blockIdx.y; row < rows; row += gridDim.y)
        for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < Pout; j += (long)gridDim.x * 256) {
            const float v = gp[(size_t)row * Pout + j];
            const int a = arg[(size_t)row * Pout + j];
            float4 *dst = reinterpret_cast<float4 *>(g + ((size_t)row * Pout + j) * pool);
            for (int q = 0; q < pool / 4; ++q)
                dst[q] = make_float4(4 * q == a ? v : 0.f, 4 * q + 1 == a ? v : 0.f, 4 * q + 2 == a ? v : 0.f, 4 * q + 3 == a ? v : 0.f);
        }
}

// maxpool_bwd with the pooled layer's BatchNorm-backward sums on the way: the scattered gradient is zero except at the arg-max position of every
// pooled point, so   sum_P mask(g) = sum_j mask_j gp[j],   sum_P mask(g) xhat = sum_j mask_j gp[j] xhat(y[j pool + arg[j]])
// -- Pout gathered values of y per row instead of bn_bwd_reduce_kernel's pass over the dense (g, y) pair (pool = 20: a twentieth of 2 x 94 MB at
// the first set-abstraction level).  grid (chunks of Pout, C, B) like the dense reduce; one fp64 atomic pair per workgroup.
template <bool RELU, bool VEC>
__global__ __launch_bounds__(256) void maxpool_bwd_bnred_kernel(int C, long Pout, int pool, const float *__restrict__ gp, const signed char *__restrict__ arg,
                                                                  float *__restrict__ g, const float *__restrict__ y, const float *__restrict__ p,
                                                                  double *__restrict__ sums)
{
    __shared__ float red[8];
    const int c = blockIdx.y, b = blockIdx.z;
    const float scale = p[c], shift = p[C + c], mean = p[2 * C + c], rstd = p[3 * C + c];
    const size_t row = (size_t)b * C + c;
    float s1 = 0.f, s2 = 0.f;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < Pout; j += (long)gridDim.x * 256) {
        const float v = gp[row * Pout + j];
        const int a = arg[row * Pout + j];
        const size_t o = (row * Pout + j) * pool;
        if (VEC) {
            float4 *dst = reinterpret_cast<float4 *>(g + o);
            for (int q = 0; q < pool / 4; ++q)
                dst[q] = make_float4(4 * q == a ? v : 0.f, 4 * q + 1 == a ? v : 0.f, 4 * q + 2 == a ? v : 0.f, 4 * q + 3 == a ? v : 0.f);
        } else {
            for (int q = 0; q < pool; ++q) g[o + q] = q == a ? v : 0.f;
        }
        const float yy = y[o + a];
        const float gm = (!RELU || fmaf(yy, scale, shift) > 0.f) ? v : 0.f;
        s1 += gm;
        s2 += gm * ((yy - mean) * rstd);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s1; red[4 + (threadIdx.x >> 6)] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sums + c, (double)((red[0] + red[1]) + (red[2] + red[3])));
        atomicAdd(sums + C + c, (double)((red[4] + red[5]) + (red[6] + red[7])));
    }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

TOp make_top(int mode, const float *aux, const float *p, int nch)
{
    TOp t;
    t.mode = mode; t.aux = aux; t.p = p; t.nch = nch;
    return t;
}

}  // namespace

int pa_tgemm_cm_try(int batch, int M, int N, int K, const float *A, long sAb, int lda, int a_kcontig, const float *B, long sBb, int ldb, int bmode,
                    const float *baux, const float *bp, float *C, long sCb, int ldc, int beta, const float *bias, const float *colv, int act,
                    double *stats, int per_batch_stats, hipStream_t st, const float *ynext = nullptr, const float *pnext = nullptr, int relu_next = 0,
                    double *sums_next = nullptr);      // train_gemm_cm.hip


// C_b (M x N) = [beta C_b +] act(A_b . f(B_b) + bias): see tgemm_nn_kernel.  a_kcontig: A(m,k) = A[m*lda + k] (else A[k*lda + m]);
// sAb = 0 shares A over the batch.  bmode 0 none / 1 affine+relu (bp: 2*K floats) / 2 bn-bwd with ReLU mask / 3 bn-bwd (baux = raw output,
// bp: 7*K floats).  stats (PA_BN_STAT_SLOTS x 2*M doubles, accumulated; pa_bn_finalize adds the replicas up) or NULL.
PA_API int pa_tgemm_nn(int batch, int M, int N, int K, const float *A, long sAb, int lda, int a_kcontig,
                       const float *B, long sBb, int ldb, int bmode, const float *baux, const float *bp,
                       float *C, long sCb, int ldc, int beta, const float *bias, const float *colv, int act, double *stats, int per_batch_stats, pa_stream_t stream)
{
    PA_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && A && B && C, "pa_tgemm_nn: bad arguments");
    PA_REQUIRE(act != 2 || (bias && colv), "pa_tgemm_nn: the squared-distance epilogue needs the row and column norms");
    PA_REQUIRE(bmode >= 0 && bmode <= 3 && (bmode == 0 || bp) && (bmode < 2 || baux), "pa_tgemm_nn: transform %d needs its parameter / auxiliary tensors", bmode);
    PA_REQUIRE(batch <= 65535 && (M + 63) / 64 <= 65535, "pa_tgemm_nn: grid limits");
    NNArgs a;
    memset(&a, 0, sizeof(a));
    a.M = M; a.N = N; a.K = K;
    a.A = A; a.sAb = sAb; a.lda = lda;
    a.B = B; a.sBb = sBb; a.ldb = ldb;
    a.tb = make_top(bmode, baux, bp, K);
    a.C = C; a.sCb = sCb; a.ldc = ldc; a.beta = beta; a.bias = bias; a.colv = colv; a.act = act; a.stats = stats;
    a.sPb = per_batch_stats ? 7L * K : 0;                       // the operand's parameter block belongs to ITS layer: K channels
    a.sStatb = per_batch_stats ? (long)PA_BN_STAT_SLOTS * 2 * M : 0;
    a.vecA = aligned16(A) && lda % 4 == 0 && sAb % 4 == 0 && (a_kcontig ? (K % 4 == 0 && K >= 4) : (M % 4 == 0 && M >= 4));
    a.vecB = aligned16(B) && ldb % 4 == 0 && sBb % 4 == 0 && (bmode < 2 || aligned16(baux)) && N % 4 == 0 && N >= 4;
    static const int xcd_mode = getenv("PA_TGEMM_XCD_ORDER") ? atoi(getenv("PA_TGEMM_XCD_ORDER")) : 2;      // bit 0: tgemm_nn, bit 1: tgemm_kk
    a.xcd_order = xcd_mode & 1;
    static const bool no_vec_store = getenv("PA_TGEMM_NO_VEC_STORE") != nullptr;                               // A/B knob
    a.vecC = !no_vec_store && aligned16(C) && ldc % 4 == 0 && sCb % 4 == 0 && N % 4 == 0;
    // 64-row tiles (half the accumulators: 4-5 workgroups per CU instead of 3) for every shape.  128-row tiles remain behind
    // PA_TGEMM_BIG_MIN (minimum number of 128 x 128 tiles to use them): measured on MI355X they lose even where they fill the chip -- the
    // finest level's 256 x 4096 x 256 per cloud x 18 is 1152 such tiles on 768 resident slots = two rounds (153 us; 64-row tiles 145 us), the
    // whole step's GEMMs 5.35 -> 5.18 ms, the 20 k x 20 k retrieval distance GEMM + select 7.64 -> 7.50 ms.
    const long t128 = (long)((M + 127) / 128) * ((N + NN_BN - 1) / NN_BN) * batch;
    static const long big_min = getenv("PA_TGEMM_BIG_MIN") ? atol(getenv("PA_TGEMM_BIG_MIN")) : (1L << 40);   // tuning knob
    const bool big = M > 64 && t128 >= big_min;
    const long t64 = (long)((M + 63) / 64) * ((N + NN_BN - 1) / NN_BN) * batch;
    const bool deep = K > 16 && !big && t64 < 1024;     // few workgroups: latency-bound rounds, halve their number (measured: hurts the chip-filling launches)
    hipStream_t st = (hipStream_t)stream;
    // the chip-filling aligned shapes (K = 64 / 128 / 256, M and N multiples of 64, A shared by the batch) run on LDS-resident weights with the B
    // operand straight from 16-byte global loads (csrc/train_gemm_cm.hip); everything else stays here
    {
        const int took = pa_tgemm_cm_try(batch, M, N, K, A, sAb, lda, a_kcontig, B, sBb, ldb, bmode, baux, bp, C, sCb, ldc, beta, bias, colv, act, stats,
                                         per_batch_stats, st);
        if (took < 0) { pa_set_error("pa_tgemm_nn (LDS-resident weights): launch failed"); return PA_EINVAL; }
        if (took) return PA_OK;
    }
    dim3 grid((N + NN_BN - 1) / NN_BN, (M + (big ? 127 : 63)) / (big ? 128 : 64), batch);
#define PA_NN(BMv, BKv, KC, MODE, VAv, VBv) hipLaunchKernelGGL((tgemm_nn_kernel<BMv, BKv, KC, MODE, VAv, VBv>), grid, dim3(256), 0, st, a)
#define PA_NN_VEC(BMv, BKv, KC, MODE)                                                                       \
    if (a.vecA) { if (a.vecB) PA_NN(BMv, BKv, KC, MODE, true, true); else PA_NN(BMv, BKv, KC, MODE, true, false); } \
    else { if (a.vecB) PA_NN(BMv, BKv, KC, MODE, false, true); else PA_NN(BMv, BKv, KC, MODE, false, false); }
#define PA_NN_MODE(BMv, BKv, KC)                                                                            \
    switch (bmode) { case 0: PA_NN_VEC(BMv, BKv, KC, 0) break; case 1: PA_NN_VEC(BMv, BKv, KC, 1) break;    \
                     case 2: PA_NN_VEC(BMv, BKv, KC, 2) break; default: PA_NN_VEC(BMv, BKv, KC, 3) break; }
#define PA_NN_KC(BMv, BKv) if (a_kcontig) { PA_NN_MODE(BMv, BKv, true) } else { PA_NN_MODE(BMv, BKv, false) }
    if (big) { if (deep) { PA_NN_KC(128, 32) } else { PA_NN_KC(128, 16) } }
    else { if (deep) { PA_NN_KC(64, 32) } else { PA_NN_KC(64, 16) } }
#undef PA_NN_KC
#undef PA_NN_MODE
#undef PA_NN_VEC
#undef PA_NN
    PA_CHECK_LAUNCH("pa_tgemm_nn");
    return PA_OK;
}

// C (M x N) += sum_b sum_k fA(A_b)(m,k) fB(B_b)(n,k)  (per_batch = 0: C must be zero-filled or hold the value to add to), or
// C_b += ... per batch (per_batch = 1).  amode 0 / 2 / 3 (aaux, ap: 7*M floats), bmode 0 / 1 (bp: 2*N floats).
static int tgemm_kk_impl(int batch, int M, int N, long K, const float *A, long sAb, int lda, int amode, const float *aaux, const float *ap,
                         const float *B, long sBb, int ldb, int bmode, const float *bp,
                         float *C, long sCb, int ldc, int per_batch, int per_batch_stats, pa_stream_t stream, int reps, long srep)
{
    PA_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && K < 2147483647L && A && B && C, "pa_tgemm_kk: bad arguments");
    PA_REQUIRE((amode == 0 || ((amode == 2 || amode == 3) && aaux && ap)) && (bmode == 0 || (bmode == 1 && bp)), "pa_tgemm_kk: transform arguments");
    KKArgs a;
    memset(&a, 0, sizeof(a));
    a.M = M; a.N = N; a.K = (int)K;
    a.A = A; a.sAb = sAb; a.lda = lda; a.ta = make_top(amode, aaux, ap, M);
    a.B = B; a.sBb = sBb; a.ldb = ldb; a.tb = make_top(bmode, nullptr, bp, N);
    a.C = C; a.sCb = sCb; a.ldc = ldc; a.per_batch = per_batch; a.reps = reps; a.srep = srep;
    a.sAPb = per_batch_stats ? 7L * M : 0;
    a.sBPb = per_batch_stats ? 7L * N : 0;
    a.vecA = aligned16(A) && lda % 4 == 0 && sAb % 4 == 0 && (amode == 0 || aligned16(aaux)) && K % 4 == 0 && K >= 4;
    a.vecB = aligned16(B) && ldb % 4 == 0 && sBb % 4 == 0;
    static const int xcd_mode = getenv("PA_TGEMM_XCD_ORDER") ? atoi(getenv("PA_TGEMM_XCD_ORDER")) : 2;
    a.xcd_order = (xcd_mode >> 1) & 1;
    const long tiles = (long)((M + KK_BM - 1) / KK_BM) * ((N + KK_BN - 1) / KK_BN) * batch;
    long splits = (2048 + tiles - 1) / tiles;                  // aim at ~2048 workgroups
    const long maxsplits = (K + 4 * KK_BK - 1) / (4 * KK_BK);  // at least 4 k-tiles per split
    if (splits > maxsplits) splits = maxsplits;
    // every split adds its partial tile with fp32 atomics: (batch x splits) workgroups per output address.  With few output tiles (the
    // 32 x 6 weight gradient of the first set-abstraction layer over 18 x 20 480 points) 2000 workgroups queue up on 192 addresses (186 us
    // for 56 MB of operands); the sum of the k-loop (~ K / 32 / splits tile rounds) and of the queue (~ contributors x one atomic) is smallest
    // at splits = sqrt(c (K / 32) / contributors per split); c = 30 from a sweep of the three first-level shapes (tools/kk_splits.py:
    // 32 splits: 186 / 85 / 92 us -> 100 / 69 / 72 us)
    {
        // (replicated output, pa_tgemm_kk_rep: the contributors of an address are divided by the number of replicas)
        const double per_split = per_batch ? 1.0 : (reps > 0 ? (double)batch / reps : (double)batch);
        const long cap = (long)(sqrt(30.0 * ((double)K / KK_BK) / (per_split < 1.0 / 64 ? 1.0 / 64 : per_split)) + 0.999);
        if (splits > cap) splits = cap < 1 ? 1 : cap;
    }
    {
        static const long force = getenv("PA_KK_SPLITS") ? atol(getenv("PA_KK_SPLITS")) : 0;   // tuning knob
        if (force > 0) splits = force > maxsplits ? maxsplits : force;
    }
    if (splits < 1) splits = 1;
    long chunk = ((K + splits - 1) / splits + KK_BK - 1) / KK_BK * KK_BK;
    splits = (K + chunk - 1) / chunk;
    PA_REQUIRE((long)batch * splits <= 65535, "pa_tgemm_kk: grid limit (batch %d x %ld splits)", batch, splits);
    a.ksplits = (int)splits; a.kchunk = (int)chunk;
    dim3 grid((N + KK_BN - 1) / KK_BN, (M + KK_BM - 1) / KK_BM, (unsigned)(batch * splits));
    hipStream_t st = (hipStream_t)stream;
    const bool vec = a.vecA && a.vecB;       // the split size is a multiple of 32, K of 4: every split ends on a multiple of 4
#define PA_KK(AM, BMo) do { if (vec) hipLaunchKernelGGL((tgemm_kk_kernel<AM, BMo, true>), grid, dim3(256), 0, st, a); \
                            else hipLaunchKernelGGL((tgemm_kk_kernel<AM, BMo, false>), grid, dim3(256), 0, st, a); } while (0)
    if (amode == 0) { if (bmode == 0) PA_KK(0, 0); else PA_KK(0, 1); }
    else if (amode == 2) { if (bmode == 0) PA_KK(2, 0); else PA_KK(2, 1); }
    else { if (bmode == 0) PA_KK(3, 0); else PA_KK(3, 1); }
#undef PA_KK
    PA_CHECK_LAUNCH("pa_tgemm_kk");
    return PA_OK;
}

PA_API int pa_tgemm_kk(int batch, int M, int N, long K, const float *A, long sAb, int lda, int amode, const float *aaux, const float *ap,
                       const float *B, long sBb, int ldb, int bmode, const float *bp,
                       float *C, long sCb, int ldc, int per_batch, int per_batch_stats, pa_stream_t stream)
{
    return tgemm_kk_impl(batch, M, N, K, A, sAb, lda, amode, aaux, ap, B, sBb, ldb, bmode, bp, C, sCb, ldc, per_batch, per_batch_stats, stream, 0, 0);
}

namespace {
// C[m * ldc + n] += sum_r scratch[r][m * N + n]: the replicas of pa_tgemm_kk_rep added up in replica order (deterministic given the replicas)
__global__ __launch_bounds__(256) void kk_reduce_kernel(int M, int N, int reps, const float *__restrict__ scratch, float *__restrict__ C, int ldc)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * N) return;
    float s = 0.f;
    for (int r = 0; r < reps; ++r) s += scratch[(size_t)r * M * N + i];
    C[(size_t)(i / N) * ldc + i % N] += s;
}
}  // namespace

// pa_tgemm_kk (shared C, per_batch = 0) for SMALL outputs contracted over very long k: the (32 x 6) ... (64 x 64) weight gradients of the first
// set-abstraction level over 18 x 20 480 grouped points.  There a few hundred workgroups queue their fp32 atomics on a few hundred addresses
// (the splits were capped for it: 41-85 us for 100-140 MB of operands).  Here workgroup z adds its partial tile into replica z % reps of a
// zero-filled scratch (reps x M x N floats, caller-provided), and a second small launch adds the replicas to C: more splits, a thirty-second of
// the contention.
PA_API int pa_tgemm_kk_rep(int batch, int M, int N, long K, const float *A, long sAb, int lda, int amode, const float *aaux, const float *ap,
                           const float *B, long sBb, int ldb, int bmode, const float *bp, float *C, int ldc, float *scratch, int reps, pa_stream_t stream)
{
    PA_REQUIRE(scratch && reps > 0 && reps <= 256, "pa_tgemm_kk_rep: scratch of 1..256 replicas");
    const int rc = tgemm_kk_impl(batch, M, N, K, A, sAb, lda, amode, aaux, ap, B, sBb, ldb, bmode, bp, scratch, 0, N, 0, 0, stream, reps, (long)M * N);
    if (rc != PA_OK) return rc;
    hipLaunchKernelGGL(kk_reduce_kernel, dim3(pa_div_up((long)M * N, 256)), dim3(256), 0, (hipStream_t)stream, M, N, reps, scratch, C, ldc);
    PA_CHECK_LAUNCH("pa_tgemm_kk_rep");
    return PA_OK;
}


// BatchNorm (training) statistics -> per-channel parameter block p (7 * nch floats, rows 0..3 written here, 4..6 by pa_bn_bwd_finalize).
PA_API int pa_bn_finalize(int nch, int groups, double count, const double *stats, const float *gamma, const float *beta, float eps, float momentum,
                          float *running_mean, float *running_var, float *p, long long *num_batches_tracked, pa_stream_t stream)
{
    PA_REQUIRE(nch > 0 && groups > 0 && count > 0 && stats && p, "pa_bn_finalize: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(pa_div_up(nch, 256)), dim3(256), 0, (hipStream_t)stream, nch, groups, count, stats, gamma, beta, eps, momentum,
                       running_mean, running_var, p, num_batches_tracked);
    PA_CHECK_LAUNCH("pa_bn_finalize");
    return PA_OK;
}

PA_API int pa_bn_eval_params(int nch, int groups, const float *gamma, const float *beta, const float *running_mean, const float *running_var, float eps,
                             float *p, pa_stream_t stream)
{
    PA_REQUIRE(nch > 0 && groups > 0 && running_mean && running_var && p, "pa_bn_eval_params: bad arguments");
    hipLaunchKernelGGL(bn_eval_params_kernel, dim3(pa_div_up(nch, 256)), dim3(256), 0, (hipStream_t)stream, nch, groups, gamma, beta, running_mean, running_var, eps, p);
    PA_CHECK_LAUNCH("pa_bn_eval_params");
    return PA_OK;
}

// sums (2*C doubles, zero-filled by the caller) over g, y (B, C, P) channel-major; relu != 0 masks g where BN(y) <= 0.
PA_API int pa_bn_bwd_reduce(int B, int C, long P, const float *g, const float *y, const float *p, int relu, double *sums, int per_batch_stats, pa_stream_t stream)
{
    const long sPb = per_batch_stats ? 7L * C : 0, sSumb = per_batch_stats ? 2L * C : 0;
    PA_REQUIRE(B > 0 && C > 0 && P > 0 && g && y && p && sums && B <= 65535 && C <= 65535, "pa_bn_bwd_reduce: bad arguments");
    long chunks = (2048 + (long)B * C - 1) / ((long)B * C);
    const long maxc = (P + 4095) / 4096;
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
    const int vec = aligned16(g) && aligned16(y) && P % 4 == 0 && ((P + chunks - 1) / chunks) % 4 == 0;
    dim3 grid((unsigned)chunks, C, B);
    if (relu) hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, C, P, g, y, p, sums, vec, sPb, sSumb);
    else hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, C, P, g, y, p, sums, vec, sPb, sSumb);
    PA_CHECK_LAUNCH("pa_bn_bwd_reduce");
    return PA_OK;
}

// pa_maxpool_bwd(B * C, Pout, pool, gp, arg, g) together with pa_bn_bwd_reduce(B, C, Pout * pool, g, y, p, relu, sums) of the pooled layer: the sums
// come from the Pout pooled gradients and the raw outputs at their arg-max positions (the dense pair is not read).  y (B, C, Pout * pool) raw layer
// output, p its 7 x C parameter block, sums 2 x C doubles (accumulated: zero-filled by the caller).
PA_API int pa_maxpool_bwd_bnred(int B, int C, long Pout, int pool, const float *gp, const signed char *arg, float *g, const float *y, const float *p, int relu,
                                double *sums, pa_stream_t stream)
{
    PA_REQUIRE(B > 0 && C > 0 && Pout > 0 && pool > 0 && pool <= 127 && gp && arg && g && y && p && sums && B <= 65535 && C <= 65535, "pa_maxpool_bwd_bnred: bad arguments");
    long chunks = (Pout + 1023) / 1024;                          // four pooled points per thread
    if (chunks > 64) chunks = 64;
    const dim3 grid((unsigned)chunks, (unsigned)C, (unsigned)B);
    const bool vec = pool % 4 == 0 && aligned16(g);
    hipStream_t st = (hipStream_t)stream;
#define PA_MPB(R, V) hipLaunchKernelGGL((maxpool_bwd_bnred_kernel<R, V>), grid, dim3(256), 0, st, C, Pout, pool, gp, arg, g, y, p, sums)
    if (relu) { if (vec) PA_MPB(true, true); else PA_MPB(true, false); }
    else { if (vec) PA_MPB(false, true); else PA_MPB(false, false); }
#undef PA_MPB
    PA_CHECK_LAUNCH("pa_maxpool_bwd_bnred");
    return PA_OK;
}

// pa_tgemm_nn for an input-gradient contraction (bmode 2 / 3) whose result C (batch, M, N) is the gradient of the NEXT (earlier) layer's activation,
// together with that layer's pa_bn_bwd_reduce(batch, M, N, C, ynext, pnext, relu_next, sums_next): on the LDS-resident-weights kernel the sums ride on
// the contraction's epilogue (C is not read back); shapes that kernel does not take run the two launches.  A shared by the batch (sAb = 0).
PA_API int pa_tgemm_nn_bnred(int batch, int M, int N, int K, const float *A, int lda, int a_kcontig, const float *B, long sBb, int ldb, int bmode,
                             const float *baux, const float *bp, float *C, long sCb, int ldc, const float *ynext, const float *pnext, int relu_next,
                             double *sums_next, pa_stream_t stream)
{
    PA_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && A && B && C && ynext && pnext && sums_next, "pa_tgemm_nn_bnred: bad arguments");
    PA_REQUIRE(bmode >= 2 && bmode <= 3 && baux && bp, "pa_tgemm_nn_bnred: an input-gradient contraction (bmode 2 / 3)");
    PA_REQUIRE(sCb == (long)M * N && ldc == N, "pa_tgemm_nn_bnred: C contiguous (batch, M, N) like ynext");
    const int took = pa_tgemm_cm_try(batch, M, N, K, A, 0, lda, a_kcontig, B, sBb, ldb, bmode, baux, bp, C, sCb, ldc, 0, nullptr, nullptr, 0, nullptr, 0,
                                     (hipStream_t)stream, ynext, pnext, relu_next, sums_next);
    if (took < 0) { pa_set_error("pa_tgemm_nn_bnred: launch failed"); return PA_EINVAL; }
    if (took) return PA_OK;
    const int rc = pa_tgemm_nn(batch, M, N, K, A, 0, lda, a_kcontig, B, sBb, ldb, bmode, baux, bp, C, sCb, ldc, 0, nullptr, nullptr, 0, nullptr, 0, stream);
    if (rc != PA_OK) return rc;
    return pa_bn_bwd_reduce(batch, M, N, C, ynext, pnext, relu_next, sums_next, 0, stream);
}

PA_API int pa_bn_bwd_finalize(int nch, int groups, double count, const double *sums, float *p, float *dgamma, float *dbeta, pa_stream_t stream)
{
    PA_REQUIRE(nch > 0 && groups > 0 && count > 0 && sums && p, "pa_bn_bwd_finalize: bad arguments");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(pa_div_up(nch, 256)), dim3(256), 0, (hipStream_t)stream, nch, groups, count, sums, p, dgamma, dbeta);
    PA_CHECK_LAUNCH("pa_bn_bwd_finalize");
    return PA_OK;
}

// out = [relu](y*scale + shift) over (B, C, P); pool > 0 additionally takes the max over groups of `pool` consecutive points
// (out (B, C, P/pool), arg (B, C, P/pool) int8 = winning slot; patch_aug_net.py:236).
PA_API int pa_bn_apply(int B, int C, long P, int pool, int relu, const float *y, const float *p, float *out, signed char *arg, int per_batch_stats, pa_stream_t stream)
{
    PA_REQUIRE(B > 0 && C > 0 && P > 0 && y && p && out && (pool <= 0 || (arg && P % pool == 0 && pool < 128)), "pa_bn_apply: bad arguments");
    const long Pout = pool > 0 ? P / pool : P;
    long gx = (Pout + 255) / 256;
    if (gx > 64) gx = 64;
    const long rows = (long)B * C;
    const dim3 grid((unsigned)gx, (unsigned)(rows < 65535 ? rows : 65535));
    const long sPb = per_batch_stats ? 7L * C : 0L;
    hipStream_t st = (hipStream_t)stream;
    static const bool plain = getenv("PA_BN_APPLY_PLAIN") != nullptr;          // A/B switch: the direct kernels only
    if (!plain && pool <= 0 && P % 4 == 0 && aligned16(y) && aligned16(out)) {
        long g4 = (P / 4 + 255) / 256;
        hipLaunchKernelGGL(bn_apply_vec_kernel, dim3((unsigned)(g4 > 64 ? 64 : g4), grid.y), dim3(256), 0, st, C, P / 4, relu, y, p, out, sPb, rows);
    } else if (!plain && pool > 0 && pool <= 63) {
        const size_t lds = 256 * (size_t)(pool + 1) * sizeof(float);
        if (pool % 4 == 0 && aligned16(y)) hipLaunchKernelGGL(bn_apply_pool_lds_kernel<true>, grid, dim3(256), lds, st, C, P, pool, relu, y, p, out, arg, sPb, rows);
        else hipLaunchKernelGGL(bn_apply_pool_lds_kernel<false>, grid, dim3(256), lds, st, C, P, pool, relu, y, p, out, arg, sPb, rows);
    } else {
        hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(256), 0, st, C, P, pool, relu, y, p, out, arg, sPb, rows);
    }
    PA_CHECK_LAUNCH("pa_bn_apply");
    return PA_OK;
}

PA_API int pa_maxpool_bwd(int rows, long Pout, int pool, const float *gp, const signed char *arg, float *g, pa_stream_t stream)
{
    PA_REQUIRE(rows > 0 && Pout > 0 && pool > 0 && gp && arg && g, "pa_maxpool_bwd: bad arguments");
    long gx = (Pout + 255) / 256;
    if (gx > 64) gx = 64;
    const dim3 grid((unsigned)gx, (unsigned)(rows < 65535 ? rows : 65535));
    if (pool % 4 == 0 && aligned16(g)) hipLaunchKernelGGL(maxpool_bwd_vec_kernel, grid, dim3(256), 0, (hipStream_t)stream, Pout, pool, gp, arg, g, (long)rows);
    else hipLaunchKernelGGL(maxpool_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, Pout, pool, gp, arg, g, (long)rows);
    PA_CHECK_LAUNCH("pa_maxpool_bwd");
    return PA_OK;
}
