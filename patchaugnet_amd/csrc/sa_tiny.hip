// First set-abstraction level as a PERSISTENT kernel: the level's weights in registers, every layer fed from the previous layer's accumulators.
//
// Same function as chain_kernel<5, 4, MODE_SA, true, 1> (mlp_chain.hip): gather the k neighbours of every centre, subtract the centre (coordinates
// AND features: libs/pointops/functions/pointops.py:559-570), three 1x1 conv + folded BatchNorm + ReLU layers (utils/model_util/pt_util.py:16-41) and
// the max over the neighbourhood (place_recognition/patch_aug_net/models/patch_aug_net.py:236) -- for the ONE shape both models use at the finest
// level: 3 + c_feat <= 8 input channels -> 32 -> 32 -> 64, nsample in 13..20.
//
// Why a kernel of its own: at this level the matrix work of an 80-row tile is 260 MFMAs (8 320 cycles) but the generic chain
// kernel spends 46 000 cycles on it -- every layer starts with an L2 round trip for its first weight fragments and ends with one for its bias, and
// the neighbour gather in front is two dependent round trips with nothing to hide under.  Round 2 kept the 13 KB of weights and the activation
// tiles in LDS (56 us at B = 32, 100 KB of LDS per workgroup); round 6 (below) keeps both in REGISTERS: 51 us, 26 KB of LDS.  Eight wavefronts per
// workgroup (two per SIMD, one workgroup per CU) each own a wave-private 80-row tile and LOOP over tiles; the gather is software-pipelined
// across the loop: while tile t is multiplied, the coordinates of tile t + 1 and the indices of tile t + 2 are in flight.
#include <stdlib.h>

#include "pa_common.h"

#include "pa_chain.h"

namespace {

constexpr int ST_N0 = 32, ST_N1 = 32, ST_N2 = 64, ST_K0 = 8;
constexpr int ST_BTOT = ST_N0 + ST_N1 + ST_N2;

struct RowsRaw {           // the loaded indices as they came (see load_rows); base = first row of the cloud, -1 = padding row
    int nbr[2], cen[2], base[2];
};
struct Pts {
    float p[2][6], c[2][6];
};

// The activations stay in REGISTERS between the layers and the weights in registers for the whole launch.
//
// In the operand-swapped MFMA (A = W^T fragment, B = activation fragment) the accumulator of lane (i = l % 16, q = l / 16) for column tile ct holds
// channels 16 ct + 4 q + r (r = 0..3) of point i -- which IS the B fragment (k slot q, column i) of the NEXT layer for a k-step that contracts the
// channels {16 ct + 4 q' + r : q' = 0..3}.  So layer l + 1 runs straight from layer l's accumulators (bias folded into the accumulator's initial
// value, ReLU in place): no activation tile in LDS, no fence, no transposition; only the contraction ORDER differs from the k-ascending kernels
// (k-step (ct, r) instead of 4 ks + q), i.e. the fp32 additions of a dot product are taken in another order -- equal to the generic kernel up to
// that, not bit for bit (tests/test_gpu_chain.py holds both to float64), and the kernel is chosen by the layer shapes only, never by the batch.
// The level's 3 328 weights are 52 registers per lane in that fragment order (13 KB spread over the wave): loaded once, no LDS reads in the loop.
// What stays in LDS: the biases (512 B) and the K = 8 input tile of the first layer (80 rows x 10 floats per wave, written by the pipelined gather,
// read back as one 8-byte word per row tile: the row is stored as (c0, c4, c1, c5, c2, c6, c3, c7) so that a lane's two k-steps are adjacent).
// 26 KB of LDS per workgroup instead of 100 KB: the other streams' workgroups (neighbour search, sampling, the coarser chains) co-reside with it.
constexpr int S2_STRIDE = 10;

template <int RT, int ST_WAVES, bool WIN = false>
__global__ __launch_bounds__(ST_WAVES * 64) void sa_tiny_reg_kernel(PaChain a, long ntiles)
{
    constexpr int R = RT * 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *bs = smem;                                                   // [128] biases of the three layers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    float *act = smem + ST_BTOT + wave * (R * S2_STRIDE);
    // profiling only (tools/probes/sa_tiny_phases.py): per-WAVE stamps -- 0 entry, 1 weights + biases in place, 2 first tile gathered, 3 + t end of the wave's tile t
    const long wid = (long)blockIdx.x * ST_WAVES + wave;
#define ST_WSTAMP(i) do { if (a.dbg && wid < 512 && lane == 0) a.dbg[wid * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
    ST_WSTAMP(0);
    for (int q = tid; q < ST_BTOT; q += ST_WAVES * 64)
        bs[q] = q < ST_N0 ? a.L[0].bias[q] : q < ST_N0 + ST_N1 ? a.L[1].bias[q - ST_N0] : a.L[2].bias[q - ST_N0 - ST_N1];
    // weight fragments, once per lane: w0[ks][ct'] = Wt0[4 ks + q][16 ct' + i];  w1[ct][ct'][r] = Wt1[16 ct + 4 q + r][16 ct' + i];  w2 likewise
    float w0[2][2];
    floatx4 w1[2][2], w2[2][4];
    {
        const float *t0 = a.L[0].wt, *t1 = a.L[1].wt, *t2 = a.L[2].wt;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int co = 0; co < 2; ++co) w0[ks][co] = t0[(size_t)(4 * ks + lq) * ST_N0 + 16 * co + li];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int co = 0; co < 2; ++co)
#pragma unroll
                for (int r = 0; r < 4; ++r) w1[ct][co][r] = t1[(size_t)(16 * ct + 4 * lq + r) * ST_N1 + 16 * co + li];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int co = 0; co < 4; ++co)
#pragma unroll
                for (int r = 0; r < 4; ++r) w2[ct][co][r] = t2[(size_t)(16 * ct + 4 * lq + r) * ST_N2 + 16 * co + li];
    }
    __syncthreads();
    ST_WSTAMP(1);
    const float *b0 = bs, *b1 = bs + ST_N0, *b2 = bs + ST_N0 + ST_N1;

    const long nw = (long)gridDim.x * ST_WAVES;
    long tile = (long)blockIdx.x * ST_WAVES + wave;
    if (tile >= ntiles) return;                          // wave-uniform; no workgroup barrier below
    const int C = a.c_feat;

    // Index loads of a tile are ISSUED two tiles ahead and only CONSUMED (turned into row addresses) one tile later, by load_pts: the raw values stay
    // in registers meanwhile.  (Computing `cloud base + index` right behind the load -- the round-2 form -- made every tile wait for an L2 round trip
    // in front of its MFMAs: s_waitcnt vmcnt(0) after each index load in the ISA.)  Rows past the end read entry 0 and are masked when used: no
    // divergent branch around the loads either.
    auto load_rows = [&](long t, RowsRaw &rw) {          // neighbour-major rows: row = slot * 4 + group (pa_chain.h chain_prologue, POOLED)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = h * 64 + lane;
            const long gid = t * 4 + (r & 3);
            int sl = r >> 2;
            if (sl >= a.ns) sl = 0;
            const bool ok = r < R && gid < a.rows;
            unsigned g32 = ok ? (unsigned)gid : 0u;                                   // groups < 2^31 (host-checked): 32-bit division, not the 64-bit loop
            if (WIN) { const unsigned wb = g32 / (unsigned)a.win_len; g32 = wb * (unsigned)a.m_ctr + (unsigned)a.win_off + (g32 - wb * (unsigned)a.win_len); }
            rw.base[h] = ok ? (int)((g32 / (unsigned)a.m_ctr) * (unsigned)a.n_src) : -1;
            rw.nbr[h] = a.nbr_idx[(size_t)g32 * a.ns + (ok ? sl : 0)];
            rw.cen[h] = a.center_idx[g32];
        }
    };
    const bool same = a.feat == a.xyz && C == 3;             // workgroup-uniform
    typedef float f3 __attribute__((ext_vector_type(3)));
    auto load_pts = [&](const RowsRaw &rw, Pts &pt) {     // raw coordinates / features of the neighbour and of its centre (padding rows: point 0, masked in write_tile)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int bs_ = max(rw.base[h], 0);
            const int sp = bs_ + rw.nbr[h], c = bs_ + rw.cen[h];
            const f3 ps = *reinterpret_cast<const f3 *>(a.xyz + (size_t)sp * 3), pc = *reinterpret_cast<const f3 *>(a.xyz + (size_t)c * 3);
            pt.p[h][0] = ps.x; pt.p[h][1] = ps.y; pt.p[h][2] = ps.z;
            pt.c[h][0] = pc.x; pt.c[h][1] = pc.y; pt.c[h][2] = pc.z;
            if (!same) {         // (same: the features ARE the coordinates; write_tile duplicates them -- touching the loaded registers here would wait for the loads)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    pt.p[h][3 + t] = t < C ? a.feat[(size_t)sp * C + t] : 0.f;
                    pt.c[h][3 + t] = t < C ? a.feat[(size_t)c * C + t] : 0.f;
                }
            }
        }
    };
    auto write_tile = [&](const RowsRaw &rw, const Pts &pt) {   // centred coordinates / features, zero padding to 8; stored as (c0, c4, c1, c5, c2, c6, c3, c7)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = h * 64 + lane;
            if (r >= R) continue;
            const bool live = rw.base[h] >= 0;
            float v[6];
#pragma unroll
            for (int t = 0; t < 3; ++t) v[t] = live ? pt.p[h][t] - pt.c[h][t] : 0.f;
#pragma unroll
            for (int t = 3; t < 6; ++t) v[t] = same ? v[t - 3] : (live ? pt.p[h][t] - pt.c[h][t] : 0.f);
            float2 *d = reinterpret_cast<float2 *>(act + r * S2_STRIDE);
            d[0] = make_float2(v[0], v[4]);
            d[1] = make_float2(v[1], v[5]);
            d[2] = make_float2(v[2], 0.f);
            d[3] = make_float2(v[3], 0.f);
        }
    };

    RowsRaw rw_cur, rw_nxt;
    Pts pt_cur, pt_nxt;
    load_rows(tile, rw_cur);
    load_pts(rw_cur, pt_cur);
    const bool has1 = tile + nw < ntiles;
    if (has1) load_rows(tile + nw, rw_nxt);
    if ((wave >> 2) & 1)
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);

#define ST_STAMP(i) do { } while (0)
    ST_WSTAMP(2);
    int it = 0;
    for (; tile < ntiles; tile += nw, ++it) {
        ST_STAMP(0);
        write_tile(rw_cur, pt_cur);
        const bool more = tile + nw < ntiles;
        RowsRaw rw_n2;
        if (more) {
            load_pts(rw_nxt, pt_nxt);
            load_rows(min(tile + 2 * nw, ntiles - 1), rw_n2);      // (clamped: the last tiles' look-ahead re-reads a valid tile and is never used)
        }
        lds_fence();
        ST_STAMP(1);
        // ---- layer 0: 8 -> 32 from the LDS tile (both k-steps of a row tile in one 8-byte read); the accumulators start at the bias
        floatx4 h0[RT][2];
        {
            float2 x[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) x[rt] = *reinterpret_cast<const float2 *>(act + (rt * 16 + li) * S2_STRIDE + 2 * lq);
#pragma unroll
            for (int co = 0; co < 2; ++co) {
                const float4 bz = *reinterpret_cast<const float4 *>(b0 + 16 * co + 4 * lq);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) h0[rt][co] = (floatx4){bz.x, bz.y, bz.z, bz.w};
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int co = 0; co < 2; ++co) h0[rt][co] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[0][co], x[rt].x, h0[rt][co], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int co = 0; co < 2; ++co) h0[rt][co] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[1][co], x[rt].y, h0[rt][co], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int co = 0; co < 2; ++co)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h0[rt][co][r] = fmaxf(h0[rt][co][r], 0.f);
        }
        ST_STAMP(2);
        // ---- layer 1: 32 -> 32, B fragments = layer 0's accumulators
        floatx4 h1[RT][2];
        {
#pragma unroll
            for (int co = 0; co < 2; ++co) {
                const float4 bz = *reinterpret_cast<const float4 *>(b1 + 16 * co + 4 * lq);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) h1[rt][co] = (floatx4){bz.x, bz.y, bz.z, bz.w};
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int co = 0; co < 2; ++co) h1[rt][co] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[ct][co][r], h0[rt][ct][r], h1[rt][co], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int co = 0; co < 2; ++co)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h1[rt][co][r] = fmaxf(h1[rt][co][r], 0.f);
        }
        ST_STAMP(3);
        // ---- layer 2: 32 -> 64 + max over the neighbourhood (row tiles in registers, lanes l % 16 = g, g + 4, g + 8, g + 12 by two DPP row rotations),
        // then bias + ReLU (both monotone: exact) and one 16-byte store per group and channel quad
        {
            floatx4 acc[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int co = 0; co < 4; ++co) acc[rt][co] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int co = 0; co < 4; ++co) acc[rt][co] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[ct][co][r], h1[rt][ct][r], acc[rt][co], 0, 0, 0);
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                floatx4 m = acc[0][co];
#pragma unroll
                for (int rt = 1; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], acc[rt][co][r]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m[r] = fmaxf(m[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m[r]), 0x120 + 4, 0xf, 0xf, true)));   // row_ror:4
                    m[r] = fmaxf(m[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m[r]), 0x120 + 8, 0xf, 0xf, true)));   // row_ror:8
                }
                long grp = tile * 4 + li;
                const bool live = li < 4 && grp < a.rows;
                if (WIN) { const unsigned wb = (unsigned)grp / (unsigned)a.win_len; grp = (long)wb * a.m_ctr + a.win_off + ((unsigned)grp - wb * (unsigned)a.win_len); }
                if (live) {
                    const int col = co * 16 + lq * 4;
                    const float4 bias = *reinterpret_cast<const float4 *>(b2 + col);
                    const float4 v = make_float4(fmaxf(m[0] + bias.x, 0.f), fmaxf(m[1] + bias.y, 0.f), fmaxf(m[2] + bias.z, 0.f), fmaxf(m[3] + bias.w, 0.f));
                    float *o = a.out + grp * a.ldo + col;
                    if (a.vec_out) *reinterpret_cast<float4 *>(o) = v;
                    else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
                }
            }
        }
        ST_STAMP(4);
        if (it < 5) ST_WSTAMP(3 + it);
        rw_cur = rw_nxt;
        pt_cur = pt_nxt;
        rw_nxt = rw_n2;
    }
#undef ST_STAMP
#undef ST_WSTAMP
}

}  // namespace

// Does the persistent tiny-chain kernel take this launch?  (mlp_chain.hip chain_dispatch asks for pooled set-abstraction chains.)
bool pa_sa_tiny_applies(const PaChain &a, int rt)
{
    return a.nlayers == 3 && (rt == 4 || rt == 5) && a.L[0].kpad == ST_K0 && a.L[0].n == ST_N0 && a.L[1].kpad == ST_N0 && a.L[1].n == ST_N1 &&
           a.L[2].kpad == ST_N1 && a.L[2].n == ST_N2 && a.c_feat >= 1 && a.c_feat <= 3 && a.L[0].wt && a.L[1].wt && a.L[2].wt;
}

template <int RT>
static void sa_tiny_launch_t(const PaChain &a_in, long ntiles, int cus, hipStream_t st)
{
    constexpr int W = 8;      // 52 weight + up to 120 activation registers per lane: two waves per SIMD (eight per workgroup), never three
    static const int stagger = getenv("PA_SA_TINY_STAGGER") ? atoi(getenv("PA_SA_TINY_STAGGER")) : 1;      // units of 8 128 cycles
    PaChain a = a_in;
    a.stagger = (ntiles + (long)W * cus - 1) / ((long)W * cus) >= 2 ? stagger : 0;                        // only when the waves loop
    long grid = (ntiles + W - 1) / W;
    if (grid > cus) grid = cus;                              // one persistent workgroup per CU; waves loop over the tiles
    const size_t lds = (size_t)(ST_BTOT + W * RT * 16 * S2_STRIDE) * 4;
    if (a.win_len > 0) hipLaunchKernelGGL((sa_tiny_reg_kernel<RT, W, true>), dim3((unsigned)grid), dim3(W * 64), lds, st, a, ntiles);
    else hipLaunchKernelGGL((sa_tiny_reg_kernel<RT, W>), dim3((unsigned)grid), dim3(W * 64), lds, st, a, ntiles);
}

int pa_sa_tiny_launch(const PaChain &a, int rt, long ntiles, hipStream_t st)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (rt == 5) sa_tiny_launch_t<5>(a, ntiles, cus, st);
    else sa_tiny_launch_t<4>(a, ntiles, cus, st);
    return 0;
}
