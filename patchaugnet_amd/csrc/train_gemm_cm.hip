// Training-mode 1x1 convolution on channel-major activations with the WEIGHTS RESIDENT IN LDS (gfx950, exact fp32 MFMA):
//     C_b (M x N) = A (M x K) . f(B_b) (K x N) [+ bias[m]],   b = 0 .. batch - 1,   B_b and C_b n-contiguous (points),
// the chip-filling shapes of pa_tgemm_nn (csrc/train_gemm.hip): forward and input-gradient contractions of the 64..256-wide layers of
// pt_util.SharedMLP in train() mode (utils/model_util/pt_util.py:16-41, :98-152) and the NetVLAD assignment (loupe.py:196-204) as the training
// step drives them (train_place_recognition.py:142-169).  f = the operand transform of the LDS-tiled kernel (none / BatchNorm+ReLU of the
// previous layer / BatchNorm-backward of (dZ, raw Y)); the epilogue accumulates the per-row sum and sum of squares BatchNorm needs.
//
// Why a second kernel.  The LDS-tiled kernel stages BOTH operands through LDS in 64 x 128 x 16 tiles, one barrier per k-tile: at 18 x (256 x 4096 x
// 256) it is 51 % MFMA-busy, its waves wait 57 % of their cycles (profiles/r03, tools/pmc_tgemm.sh) -- 0.47-0.49 of the fp32 MFMA peak.  Here
//   * a workgroup is pinned to a 128-row block of A (both layouts of the weight matrix are packed into MFMA fragment order on the way in:
//     128 x 256 floats = 128 KB, read once per workgroup) and its eight wavefronts walk (cloud, 64-column block) tiles of that row block;
//   * the B operand never touches LDS: with the point axis contiguous, a lane's 16-byte load of B[4 s + l / 16][n0 + 4 (l % 16) .. + 3] IS the B
//     fragment of k-step s for FOUR column tiles (column tile t = columns n0 + 4 j + t, j = 0 .. 15: the MFMA does not care which sixteen columns a
//     tile holds), and a lane's accumulators of those four tiles are four CONSECUTIVE columns of one output row: loads and stores are 16 bytes per
//     lane with no transposition anywhere; the transform f runs on the loaded registers;
//   * no workgroup barrier after the weights have landed: a wave keeps two register groups of eight k-steps (four in the backward modes, whose
//     transform carries a second operand), the loads of group g + 1 (and of the next tile's first group) in flight under the MFMAs of group g;
//   * statistics: per tile a 16-lane DPP row sum, accumulated in double precision in a wave-private LDS block (a wave's LDS operations execute
//     in order: no atomics, fixed order), one fp64 atomic pair per row and WORKGROUP at the end (the LDS-tiled kernel: per row and 64 x 128 tile).
// Same arithmetic contract as the LDS-tiled kernel (exact fp32 MFMA, k ascending within a row); the two are interchangeable and tested against
// each other and against float64 (tests/test_gpu_train_ops.py).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "pa_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

struct CMArgs {
    int M, N, K, batch;
    const float *A; int lda; int a_kcontig;            // A(m,k) = A[m*lda + k] (a_kcontig) or A[k*lda + m]
    const float *B; long sBb; int ldb;
    const float *aux;                                  // modes 2/3: raw layer output, layout of B
    const float *p;                                    // per-channel parameters, SoA p[j*K + ch] (mode 1: j < 2; modes 2/3: j < 7)
    float *C; long sCb; int ldc;
    const float *bias;                                 // per m or null
    int beta;                                          // 1: C += (the result is added to what C holds: a second gradient contribution)
    double *stats;                                     // STATS 1: [PA_BN_STAT_SLOTS][2*M] sum / sum of squares of C; STATS 2: [2*M] BatchNorm-backward sums; or null
    const float *ynext, *pnext; int relu_next;         // STATS 2: raw output (layout of C) and 7*M parameter block of the layer whose activation gradient C is
    int coltiles_per_cloud;                            // N / (16 CT)
    long coltiles;                                     // batch * N / (16 CT)
    long tiles_per_group;                              // wave tiles (column tile x 64-row half) per workgroup
    int dbg;                                           // PA_TGEMM_CM_DBG (measurement only): 1 = no tiles (the fixed cost of a launch), 2 = tiles without the weight copy
};



__device__ __forceinline__ float dpp_row_sum16(float v)
{
    // sum over the 16 lanes of a DPP row; the total is valid in the row's last lane (l % 16 == 15)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), PA_DPP_ROW_SHR(1), 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), PA_DPP_ROW_SHR(2), 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), PA_DPP_ROW_SHR(4), 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), PA_DPP_ROW_SHR(8), 0xf, 0xf, true));
    return v;
}

template <int MODE>
__device__ __forceinline__ float cm_tf(float g, float y, const float4 &c0, const float4 &c1)
{
    if (MODE == 0) return g;
    if (MODE == 1) return fmaxf(fmaf(g, c0.x, c0.y), 0.f);
    const float z = fmaf(y, c0.x, c0.y);
    const float gm = (MODE == 3 || z > 0.f) ? g : 0.f;
    const float xhat = (y - c0.z) * c0.w;
    return (gm - c1.x - xhat * c1.y) * c1.z;
}

// NG = K / 32 (1, 2, 4 or 8: K = 32, 64, 128, 256); MH = 64-row halves of the workgroup's row block (1 or 2);
// CT = columns per lane = column tiles per wave tile (4: 64-column tiles, 16-byte accesses; 2: 32-column tiles, 8-byte accesses -- twice the tiles
// for launches whose 64-column tile count leaves SIMDs idle in the last round, and half the accumulators: twelve wavefronts per workgroup)
template <int NG, int MODE, int MH, int STATS, int CT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void tgemm_cm_kernel(CMArgs a)
{
    constexpr int KS = NG * 8;                          // k-steps of four channels
    constexpr int GS = (MODE >= 2 || NG == 1) ? 4 : 8;  // k-steps per register group (the backward transforms carry a second operand: half the group; K = 32: two groups of four)
    constexpr int NGR = KS / GS;                        // register groups per tile (even)
    constexpr int TW = 16 * CT;                         // columns of a wave tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *wf = reinterpret_cast<float4 *>(smem);                               // [KS][MH][64 lanes] float4: lane's A fragments of m-tiles 0..3 of half h, k-step s
    float4 *ptab = wf + KS * MH * 64;                                           // [K][2] float4 (modes >= 1)
    double *sst = reinterpret_cast<double *>(ptab + (MODE ? 2 * KS * 4 : 0));   // [WAVES][64][2] (STATS)
    float4 *qtab = reinterpret_cast<float4 *>(sst + (STATS ? WAVES * 128 : 0));  // [64 MH] (scale, shift, mean, rstd) of the NEXT layer's rows (STATS 2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, slot = lane & 15, kq = lane >> 4;
    const int mbase = blockIdx.y * 64 * MH;

    const long t_begin = (long)blockIdx.x * a.tiles_per_group, t_end = a.dbg == 1 ? t_begin : min(t_begin + a.tiles_per_group, a.coltiles * MH);
    long tile = t_begin + wave;
    const int h = MH == 2 ? (int)(tile & 1) : 0;                                // fixed per wave: tiles advance by WAVES (even), t_begin is even

    // Addressing through buffer descriptors: the per-lane part of an address (cloud, column block, lane) is ONE 32-bit VGPR offset per tile, the
    // k-step / output row part a SCALAR offset -- a load or store is one VMEM instruction with no vector address arithmetic (flat 64-bit
    // addressing cost a v_lshl_add_u64 per access and two VGPRs per live pointer, which pushed the forward kernel to 252 VGPRs and made the
    // allocator rotate the accumulators through registers that still had LDS reads in flight: a wait in front of every k-step's MFMAs).
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.B), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(MODE >= 2 ? a.aux : a.B), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(STATS == 2 ? a.ynext : a.B), 0, 0x7fffffff, 0x00020000);
    const unsigned krow = 16u * (unsigned)a.ldb;        // bytes from one k-step's rows to the next (four channels)
    auto tile_offs = [&](long t, unsigned &bo, unsigned &co) {
        const long ct = MH == 2 ? (t >> 1) : t;
        const long b = ct / a.coltiles_per_cloud;
        const int n0 = (int)(ct - b * a.coltiles_per_cloud) * TW;
        bo = (unsigned)(((size_t)b * a.sBb + (size_t)kq * a.ldb + n0 + CT * slot) * 4);
        co = (unsigned)(((size_t)b * a.sCb + (size_t)(mbase + 64 * h + 4 * kq) * a.ldc + n0 + CT * slot) * 4);
    };
    struct Vec { float v[CT]; };
    struct Group { Vec x[GS]; Vec y[MODE >= 2 ? GS : 1]; };
    auto ldv = [&](const __amdgpu_buffer_rsrc_t &r, unsigned vo, unsigned so) {
        Vec o;
        if constexpr (CT == 4) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
            o.v[0] = __uint_as_float(v.x); o.v[1] = __uint_as_float(v.y); o.v[2] = __uint_as_float(v.z); o.v[3] = __uint_as_float(v.w);
        } else {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0);
            o.v[0] = __uint_as_float(v.x); o.v[1] = __uint_as_float(v.y);
        }
        return o;
    };
    auto load = [&](Group &G, unsigned bo, int g) {
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            G.x[j] = ldv(brs, bo, (unsigned)(g * GS + j) * krow);
            if (MODE >= 2) G.y[j] = ldv(yrs, bo, (unsigned)(g * GS + j) * krow);
        }
    };

    Group G0, G1;
    unsigned bo = 0, co = 0;
    const long tlast = t_end - 1;
    tile_offs(min(tile, tlast < 0 ? 0 : tlast), bo, co);
    if (t_begin < t_end) load(G0, bo, 0);               // the wave's first group: in flight under the weight copy

    // ---- the row block of A, once, into fragment order: float index ((s*MH + hh)*64 + kq*16 + m%16)*4 + j  <-  A(mbase + 64 hh + 16 j + m%16, 4 s + kq)
    {
        constexpr int NT = WAVES * 64;
        const int rows = 64 * MH, K = KS * 4;
        // (consecutive lanes take consecutive ROWS: the fragment order puts rows m, m + 1 four floats apart and m, m + 16 one float apart, so a
        // wave's 4-byte LDS writes land in 64 different banks; with consecutive lanes along k every lane of a write hit the same bank: 104 vs 113 us)
        // (the loads of a batch of PU iterations are issued before their first LDS write: one L2 round trip per batch, not per iteration)
#ifndef CM_PU
#define CM_PU 6
#endif
        constexpr int PU = CM_PU;
        if (a.dbg == 2) {                               // measurement only: tiles without the weight copy
        } else if (a.a_kcontig) {                              // 16-byte reads along k: element e of the read is k = 4 s + e, i.e. k-step s, lane row kq = e
            for (int q0 = tid; q0 < rows * KS; q0 += NT * PU) {
                float4 v[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int q = q0 + u * NT, m = q % rows, s = q / rows;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);          // rows past M (M not a multiple of 64): zero fragments, never stored
                    if (q < rows * KS && mbase + m < a.M) v[u] = *reinterpret_cast<const float4 *>(a.A + (size_t)(mbase + m) * a.lda + 4 * s);
                }
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int q = q0 + u * NT, m = q % rows, s = q / rows;
                    if (q >= rows * KS) break;
                    float *dst = smem + ((size_t)(s * MH + (m >> 6)) * 64 + (m & 15)) * 4 + ((m & 63) >> 4);
                    dst[0] = v[u].x; dst[64] = v[u].y; dst[128] = v[u].z; dst[192] = v[u].w;       // kq = 0..3: lane + 16 -> + 64 floats
                }
            }
        } else {                                        // 16-byte reads along m: four consecutive rows of one k
            for (int q0 = tid; q0 < (rows / 4) * K; q0 += NT * PU) {
                float4 v[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int q = q0 + u * NT, k = q / (rows / 4), m = (q - k * (rows / 4)) * 4;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (q < (rows / 4) * K && mbase + m < a.M) v[u] = *reinterpret_cast<const float4 *>(a.A + (size_t)k * a.lda + mbase + m);      // M % 4 == 0 (host)
                }
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int q = q0 + u * NT, k = q / (rows / 4), m = (q - k * (rows / 4)) * 4;
                    if (q >= (rows / 4) * K) break;
                    float *dst = smem + ((size_t)((k >> 2) * MH + (m >> 6)) * 64 + (k & 3) * 16 + (m & 15)) * 4 + ((m & 63) >> 4);
                    dst[0] = v[u].x; dst[4] = v[u].y; dst[8] = v[u].z; dst[12] = v[u].w;           // m % 16 + 1 -> next lane -> + 4 floats
                }
            }
        }
        if (MODE) {
            constexpr int NP = MODE == 1 ? 2 : 7;
            float *pt = reinterpret_cast<float *>(ptab);
            for (int q = tid; q < K * 8; q += NT) {
                const int ch = q >> 3, j = q & 7;
                pt[q] = j < NP ? a.p[(size_t)j * K + ch] : 0.f;
            }
        }
        if (STATS)
            for (int q = tid; q < WAVES * 64 * 2; q += NT) sst[q] = 0.0;
        if (STATS == 2)
            for (int q = tid; q < rows; q += NT) {
                const int m = min(mbase + q, a.M - 1);
                qtab[q] = make_float4(a.pnext[m], a.pnext[(size_t)a.M + m], a.pnext[(size_t)2 * a.M + m], a.pnext[(size_t)3 * a.M + m]);
            }
    }
    __syncthreads();

#ifdef CM_STAGGER                                        // experiment: the SIMD's second / third wave start a third / two thirds of a tile late
    for (int d = 0; d < (wave >> 2); ++d) __builtin_amdgcn_s_sleep(CM_STAGGER);
#endif
    const float4 *wl = wf + h * 64 + lane;              // + s * MH * 64 per k-step
    const float4 *pl = ptab + 2 * kq;                   // + 8 s per k-step (channel 4 s + kq)
    double *myst = sst + (size_t)wave * 128;
    const bool plain = mbase + 64 * MH <= a.M && !a.bias && !a.beta;            // uniform: the epilogue without guards, bias or read-back

    // the LDS pipeline's state entering k-step 0: fragments of steps 0 and 1, parameters of step 1, the transformed B fragments of step 0
    float4 w0 = wl[0], w1 = wl[(size_t)MH * 64];
    float4 c0n = make_float4(0.f, 0.f, 0.f, 0.f), c1n = c0n;
    float bf[CT];
    {
        float4 c00 = c0n, c10 = c0n;
        if (MODE) { c00 = pl[0]; c0n = pl[8]; }
        if (MODE >= 2) { c10 = pl[1]; c1n = pl[9]; }
#pragma unroll
        for (int t = 0; t < CT; ++t) bf[t] = cm_tf<MODE>(G0.x[0].v[t], MODE >= 2 ? G0.y[0].v[t] : 0.f, c00, c10);
    }

    for (; tile < t_end; tile += WAVES) {
        unsigned nbo, nco;
        tile_offs(min(tile + WAVES, tlast), nbo, nco);                           // the wave's next tile (clamped: loaded, never used, past the end)
        floatx4 acc[4][CT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < CT; ++t) acc[i][t] = (floatx4){0.f, 0.f, 0.f, 0.f};

        // One register group = GS k-steps.  The LDS side is ONE software pipeline that runs through the groups and tiles of the wave (the k-steps of
        // every tile read the same fragment / parameter addresses): in k-step s the reads of step s + 2 are issued first (a scheduling fence
        // keeps them in front of the MFMAs: left alone hipcc sinks them to their first use, or hoists a group's reads and spills), then half of
        // the step's 4 CT MFMAs, the operand transform of step s + 1 (parameters read one step earlier: sixteen MFMAs of cover for an LDS round
        // trip; with one step of cover a wave sat in s_waitcnt lgkmcnt for a fifth of its cycles, SQ_WAIT_ANY in profiles/r05_tgemm_cm_pmc.txt),
        // the other half.  Gn = the register group that follows G (the next tile's first group after the tile's last).
        auto compute = [&](Group &G, Group &Gn, int g) {
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                const int s2 = (g * GS + j + 2) % KS;
                const float4 w2 = wl[(size_t)s2 * MH * 64];
                float4 c02 = c0n, c12 = c1n;
                if (MODE) c02 = pl[s2 * 8];
                if (MODE >= 2) c12 = pl[s2 * 8 + 1];
                __builtin_amdgcn_sched_barrier(0);
#define CM_ROW(i, wi)                                                                                                   \
                _Pragma("unroll") for (int t = 0; t < CT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wi, bf[t], acc[i][t], 0, 0, 0);
                CM_ROW(0, w0.x) CM_ROW(1, w0.y)
                __builtin_amdgcn_sched_barrier(0);
                float nf[CT];
                const Vec &xn = j + 1 < GS ? G.x[j + 1 < GS ? j + 1 : 0] : Gn.x[0];
                const Vec &yn = j + 1 < GS ? G.y[MODE >= 2 ? (j + 1 < GS ? j + 1 : 0) : 0] : Gn.y[0];
#pragma unroll
                for (int t = 0; t < CT; ++t) nf[t] = cm_tf<MODE>(xn.v[t], MODE >= 2 ? yn.v[t] : 0.f, c0n, c1n);
                CM_ROW(2, w0.z) CM_ROW(3, w0.w)
#undef CM_ROW
                __builtin_amdgcn_sched_barrier(0);
                w0 = w1; w1 = w2; c0n = c02; c1n = c12;
#pragma unroll
                for (int t = 0; t < CT; ++t) bf[t] = nf[t];
            }
        };

#pragma unroll
        for (int g = 0; g < NGR; g += 2) {
#ifndef CM_DBG_NOLOAD                                    // decomposition probe (never built into the library): the k-loop without its global loads
            load(G1, bo, g + 1);
#endif
            compute(G0, G1, g);
#ifndef CM_DBG_NOLOAD
            if (g + 2 < NGR) load(G0, bo, g + 2);
            else load(G0, nbo, 0);                      // next tile's first group under this tile's last MFMAs and the epilogue
#endif
            compute(G1, G0, g + 1);
        }

        // ---- epilogue: lane (slot, kq) holds C[m = 16 i + 4 kq + r][n0 + CT slot + t] in acc[i][t][r]: one 16- (8-) byte store per (i, r).
        // Nothing in the common form waits: stores are fire-and-forget, the statistics go to the wave's LDS block as fp64 LDS adds without
        // return (a wave's LDS operations execute in order: fixed summation order).  The first build read-modified-wrote that block (a round
        // trip per row) behind a per-row `bias ? load : 0` whose s_waitcnt vmcnt(0) also waited out the previous row's STORE: 16 write
        // acknowledgements per tile in series, SQ_WAIT_ANY = 22 % of a wave's cycles (profiles/r05_tgemm_cm_pmc.txt).
        const int mrow0 = mbase + 64 * h + 4 * kq;
        auto add_stat = [&](int rowslot, float t1, float t2) {
            if (slot == 15) {
                double *d = myst + rowslot * 2;
                (void)__hip_atomic_fetch_add(d, (double)t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_add(d + 1, (double)t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        auto epilogue = [&](auto plain_tag) {
            constexpr bool PLAIN = decltype(plain_tag)::value;          // whole row block inside M, no bias, beta = 0
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Vec yv[4], oldv[4];
                float bsv[4] = {0.f, 0.f, 0.f, 0.f};
                if (!PLAIN) {                           // bias and (beta) the values already in C for the four rows first: one wait for the four
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = mrow0 + 16 * i + r;
                        oldv[r] = Vec{};
                        if (m >= a.M) continue;
                        if (a.bias) bsv[r] = a.bias[m];
                        if (a.beta) oldv[r] = ldv(crs, co, (unsigned)(16 * i + r) * 4u * (unsigned)a.ldc);
                    }
                }
                if (STATS == 2) {                       // the four rows' raw outputs of the next layer first: one wait for the four
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (!PLAIN && mrow0 + 16 * i + r >= a.M) { yv[r] = Vec{}; continue; }         // padding rows: nothing behind them to read
                        yv[r] = ldv(nrs, co, (unsigned)(16 * i + r) * 4u * (unsigned)a.ldc);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mrow0 + 16 * i + r;
                    if (!PLAIN && m >= a.M) continue;   // padding rows of the last row block
                    float v[CT];
#pragma unroll
                    for (int t = 0; t < CT; ++t) v[t] = acc[i][t][r];
                    const unsigned so = (unsigned)(16 * i + r) * 4u * (unsigned)a.ldc;
                    if (!PLAIN) {
#pragma unroll
                        for (int t = 0; t < CT; ++t) v[t] = (v[t] + bsv[r]) + oldv[r].v[t];
                    }
                    if constexpr (CT == 4) {
                        const u32x4 uv = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                        __builtin_amdgcn_raw_buffer_store_b128(uv, crs, co, so, 0);
                        // gfx950 reads the data registers of a store wider than 64 bits for two more cycles after issue; hipcc covers that for
                        // the forms it knows (immediate offset) and exempts the SGPR-offset form, whose extra issue cycle hides ONE of the two:
                        // the next row's gather (v_mov into the same four registers) then overwrote lanes 12..15 of each row of the first dword
                        // about once in two thousand stores.  Found by the engine-vs-module test at 8192 points; tests/test_gpu_train_ops.py
                        // now runs the K = 32 / 64 shapes many times over.
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_nop 2" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        const u32x2 uv = {__float_as_uint(v[0]), __float_as_uint(v[1])};
                        __builtin_amdgcn_raw_buffer_store_b64(uv, crs, co, so, 0);
                    }
                    if (STATS == 1) {
                        float p1, p2;
                        if constexpr (CT == 4) { p1 = (v[0] + v[1]) + (v[2] + v[3]); p2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]); }
                        else { p1 = v[0] + v[1]; p2 = v[0] * v[0] + v[1] * v[1]; }
                        add_stat(16 * i + 4 * kq + r, dpp_row_sum16(p1), dpp_row_sum16(p2));
                    }
                    if (STATS == 2) {
                        // C is the gradient of the NEXT (earlier) layer's activation: that layer's BatchNorm-backward sums ride on this epilogue
                        // (sum mask(g), sum mask(g) xhat per row; train_gemm.hip: bn_bwd_reduce_kernel) instead of a pass of their own over (g, y)
                        const float4 q = qtab[64 * h + 16 * i + 4 * kq + r];
                        float p1 = 0.f, p2 = 0.f;
#pragma unroll
                        for (int t = 0; t < CT; ++t) {
                            const float gm = (!a.relu_next || fmaf(yv[r].v[t], q.x, q.y) > 0.f) ? v[t] : 0.f;
                            p1 += gm;
                            p2 += gm * ((yv[r].v[t] - q.z) * q.w);
                        }
                        add_stat(16 * i + 4 * kq + r, dpp_row_sum16(p1), dpp_row_sum16(p2));
                    }
                }
            }
        };
        if (plain) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        bo = nbo; co = nco;
    }
    if (STATS) {
        __syncthreads();
        // rows of half hh were accumulated by the waves with h == hh (fixed per wave), summed here in wave order; one atomic pair per row
        if (tid < 64 * MH) {
            const int hh = tid >> 6, row = tid & 63;
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const long wt0 = t_begin + w;
                const int wh = MH == 2 ? (int)(wt0 & 1) : 0;
                if (wh == hh && wt0 < t_end) { s1 += sst[(size_t)w * 128 + row * 2]; s2 += sst[(size_t)w * 128 + row * 2 + 1]; }
            }
            const unsigned sl = STATS == 1 ? (blockIdx.x + blockIdx.y * gridDim.x) % PA_BN_STAT_SLOTS : 0u;
            double *st = a.stats + (size_t)sl * 2 * a.M;
            if (mbase + 64 * hh + row < a.M) {
                atomicAdd(st + mbase + 64 * hh + row, s1);
                atomicAdd(st + a.M + mbase + 64 * hh + row, s2);
            }
        }
    }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int NG, int MODE, int MH, int CT>
void launch_cm(const CMArgs &a, dim3 grid, int stats, hipStream_t st)
{
    constexpr int WAVES = CT == 4 ? 8 : 12;
    const size_t lds = (size_t)NG * 8 * MH * 64 * 16 + (MODE ? (size_t)NG * 32 * 32 : 0) + (stats ? (size_t)WAVES * 128 * 8 : 0) + (stats == 2 ? (size_t)64 * MH * 16 : 0);
    if (stats == 1) {
        if constexpr (MODE < 2) {                       // output statistics ride on forward contractions only
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tgemm_cm_kernel<NG, MODE, MH, 1, CT, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((tgemm_cm_kernel<NG, MODE, MH, 1, CT, WAVES>), grid, dim3(WAVES * 64), lds, st, a);
        }
    } else if (stats == 2) {
        if constexpr (MODE >= 2) {                      // the next layer's BatchNorm-backward sums ride on input-gradient contractions only
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tgemm_cm_kernel<NG, MODE, MH, 2, CT, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((tgemm_cm_kernel<NG, MODE, MH, 2, CT, WAVES>), grid, dim3(WAVES * 64), lds, st, a);
        }
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&tgemm_cm_kernel<NG, MODE, MH, 0, CT, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((tgemm_cm_kernel<NG, MODE, MH, 0, CT, WAVES>), grid, dim3(WAVES * 64), lds, st, a);
    }
}

}  // namespace

static int g_tgemm_cm = -1;
// test / A/B switch: 1 = wherever the shape rules hold, 0 = never, -1 = the default rule (PA_TGEMM_NO_CM=1 turns it off)
PA_API void pa_tgemm_cm_enable(int on) { g_tgemm_cm = on; }

// 1 when the LDS-resident-weights kernel took the call (pa_tgemm_nn asks first), 0 when the shape is not one it is built for.
int pa_tgemm_cm_try(int batch, int M, int N, int K, const float *A, long sAb, int lda, int a_kcontig, const float *B, long sBb, int ldb, int bmode,
                    const float *baux, const float *bp, float *C, long sCb, int ldc, int beta, const float *bias, const float *colv, int act,
                    double *stats, int per_batch_stats, hipStream_t st, const float *ynext, const float *pnext, int relu_next, double *sums_next)
{
    static const bool off = getenv("PA_TGEMM_NO_CM") != nullptr;
    // (default from a sweep of the whole training step after the round-5 epilogue work, 1024 / 512 / 256 / 128 / 32: 4.778 / 4.765 / 4.743 / 4.722 /
    // 4.730 ms -- the kernel's launch cost fell, so it now also wins on the N = 128 .. 320 levels it used to leave to the LDS-tiled kernel)
    static const long min_tiles = getenv("PA_TGEMM_CM_MIN_TILES") ? atol(getenv("PA_TGEMM_CM_MIN_TILES")) : 128;
    if (g_tgemm_cm == 0 || (g_tgemm_cm < 0 && off)) return 0;
    if (act != 0 || (beta && (stats || sums_next)) || colv || sAb != 0 || (per_batch_stats && (bmode || stats)) || (stats && bmode >= 2)) return 0;
    if (sums_next && (bmode < 2 || stats || per_batch_stats || !ynext || !pnext || !aligned16(ynext))) return 0;
    // rows: whole 64-row blocks, or a last block at least half full (its padding rows are zero fragments: wasted MFMAs, which the narrow layers
    // that need this -- 32 output channels over 20 480 grouped points -- do not miss: they are bound by their HBM traffic)
    if (M % 4 || M < 32 || (M % 64 != 0 && M % 64 < 32) || N % 32 || (K != 32 && K != 64 && K != 128 && K != 256)) return 0;
    if (!aligned16(A) || lda % 4 || !aligned16(B) || ldb % 4 || sBb % 4 || !aligned16(C) || ldc % 4 || sCb % 4 || (bmode >= 2 && !aligned16(baux))) return 0;
    // 32-bit buffer offsets: the largest byte offset any lane forms -- the last cloud's base plus the operand's whole extent inside it (B: K rows of
    // ldb, also when sBb == 0 or ldb * K > sBb; C: M rows of ldc) -- must stay below 2^31, else the LDS-tiled kernel (64-bit addresses) takes the call
    if (((double)(batch - 1) * (double)sBb + (double)K * (double)ldb + (double)N) * 4.0 >= 2147483647.0 ||
        ((double)(batch - 1) * (double)sCb + (double)M * (double)ldc + (double)N) * 4.0 >= 2147483647.0 ||
        (bmode >= 2 && ((double)(batch - 1) * (double)sBb + (double)K * (double)ldb + (double)N) * 4.0 >= 2147483647.0)) return 0;
    const int MH = (M + 63) / 64 % 2 == 0 ? 2 : 1;
    const int chunks = (M + 64 * MH - 1) / (64 * MH);
    if (g_tgemm_cm < 0 && (long)batch * (N / 32) * MH * chunks < 2 * min_tiles) return 0;      // few tiles: the LDS-tiled kernel's finer tiles fill the chip better
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // Tiling: one workgroup per CU, its wave tiles dealt round-robin to the wavefronts, i.e. to the four SIMDs.  The launch lasts as long as its
    // busiest SIMD: ceil(tiles per workgroup / 4) tile times.  64-column tiles (16-byte accesses, 16 MFMAs per fragment read) unless the
    // 32-column form's finer quantum shortens that by more than its ~4 % of extra per-tile overhead (18 clouds x 4096 points x 256 rows: 4.5 tiles
    // per SIMD -> five rounds of 64-column tiles or nine of 32-column ones = 4.5).  PA_TGEMM_CM_CT forces a form (A/B knob).
    auto plan = [&](int ct, long &groups, long &tpg) {
        const long wtiles = (long)batch * (N / (16 * ct)) * MH;
        groups = cus / chunks > 0 ? cus / chunks : 1;
        tpg = (wtiles + groups - 1) / groups;
        tpg = (tpg + 3) / 4 * 4;                                         // a multiple of four: every SIMD of the workgroup the same count, and even
        groups = (wtiles + tpg - 1) / tpg;                               // (a column tile's two halves stay in one workgroup: adjacent waves share the B lines)
        return (double)(tpg / 4) * (ct == 4 ? 1.0 : 0.52);
    };
    static const int ct_env = getenv("PA_TGEMM_CM_CT") ? atoi(getenv("PA_TGEMM_CM_CT")) : 0;
    long g4 = 0, t4 = 0, g2 = 0, t2 = 0;
    const double c4 = N % 64 == 0 ? plan(4, g4, t4) : 1e30, c2 = plan(2, g2, t2);
    const int CT = (ct_env == 4 && N % 64 == 0) ? 4 : (ct_env == 2 ? 2 : (c4 <= c2 ? 4 : 2));
    const long groups = CT == 4 ? g4 : g2, tpg = CT == 4 ? t4 : t2;
    const long coltiles = (long)batch * (N / (16 * CT));
    CMArgs a;
    memset(&a, 0, sizeof(a));
    a.M = M; a.N = N; a.K = K; a.batch = batch;
    a.A = A; a.lda = lda; a.a_kcontig = a_kcontig;
    a.B = B; a.sBb = sBb; a.ldb = ldb; a.aux = baux; a.p = bp;
    a.C = C; a.sCb = sCb; a.ldc = ldc; a.bias = bias; a.beta = beta ? 1 : 0; a.stats = sums_next ? sums_next : stats;
    a.ynext = ynext; a.pnext = pnext; a.relu_next = relu_next;
    a.coltiles_per_cloud = N / (16 * CT); a.coltiles = coltiles; a.tiles_per_group = tpg;
#ifdef PA_TGEMM_CM_PROBE      // probe builds only (tools/build_variant.sh cmprobe "-DPA_TGEMM_CM_PROBE" train_gemm_cm.hip): the shipped library never skips tiles or the weight copy
    static const int dbg = getenv("PA_TGEMM_CM_DBG") ? atoi(getenv("PA_TGEMM_CM_DBG")) : 0;
    a.dbg = dbg;
#endif
    const dim3 grid((unsigned)groups, (unsigned)chunks);
    const int s = sums_next ? 2 : (stats != nullptr ? 1 : 0);
#define CM_CT(NGv, MODEv, MHv) { if (CT == 4) launch_cm<NGv, MODEv, MHv, 4>(a, grid, s, st); else launch_cm<NGv, MODEv, MHv, 2>(a, grid, s, st); }
#define CM_MH(NGv, MODEv) { if (MH == 2) CM_CT(NGv, MODEv, 2) else CM_CT(NGv, MODEv, 1) }
#define CM_MODE(NGv) switch (bmode) { case 0: CM_MH(NGv, 0) break; case 1: CM_MH(NGv, 1) break; case 2: CM_MH(NGv, 2) break; default: CM_MH(NGv, 3) break; }
    if (K == 256) { CM_MODE(8) } else if (K == 128) { CM_MODE(4) } else if (K == 64) { CM_MODE(2) } else { CM_MODE(1) }
#undef CM_MODE
#undef CM_MH
#undef CM_CT
    return hipGetLastError() == hipSuccess ? 1 : -1;
}
