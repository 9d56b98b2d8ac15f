// First set-abstraction level as a PERSISTENT kernel with the level's weights resident in LDS.
//
// Same function as chain_kernel<5, 4, MODE_SA, true, 1> (mlp_chain.hip): gather the k neighbours of every centre, subtract the centre (coordinates
// AND features: libs/pointops/functions/pointops.py:559-570), three 1x1 conv + folded BatchNorm + ReLU layers (utils/model_util/pt_util.py:16-41) and
// the max over the neighbourhood (place_recognition/patch_aug_net/models/patch_aug_net.py:236) -- for the ONE shape both models use at the finest
// level: 3 + c_feat <= 8 input channels -> 32 -> 32 -> 64, nsample in 13..20.
//
// Why a kernel of its own: at this level the matrix work of a tile is 260 MFMAs (8 320 cycles) but the generic chain kernel
// spends 46 000 cycles on it -- every layer starts with an L2 round trip for its first weight fragments and ends with one for its bias, and the
// neighbour gather in front is two dependent round trips with nothing to hide under (40 % MFMA-busy, 63 us at B = 32).  The level's weights are
// 13 KB.  Here a workgroup loads them ONCE into LDS in MFMA fragment order, its twelve wavefronts (three per SIMD, one workgroup per CU) each own a
// wave-private 80-row tile and LOOP over tiles, and the gather is software-pipelined across the loop: while tile t is multiplied, the coordinates
// of tile t + 1 and the indices of tile t + 2 are in flight.  The arithmetic (exact fp32 MFMA, k ascending, bias after the max, ReLU) is the
// generic kernel's, so the results are bit-identical (tests/test_gpu_chain.py).
#include <stdlib.h>

#include "pa_common.h"

#include "pa_chain.h"

namespace {

constexpr int ST_N0 = 32, ST_N1 = 32, ST_N2 = 64, ST_K0 = 8;
constexpr int ST_STRIDE = 34;                                   // floats per LDS row: 32 hidden columns + 2 (conflict-free fragment reads)
constexpr int ST_W0 = (ST_K0 / 4) * (ST_N0 / 16) * 64;          // fragment-major weights: [k-step][column tile][lane]
constexpr int ST_W1 = (ST_N0 / 4) * (ST_N1 / 16) * 64;
constexpr int ST_W2 = (ST_N1 / 4) * (ST_N2 / 16) * 64;
constexpr int ST_WTOT = ST_W0 + ST_W1 + ST_W2;
constexpr int ST_BTOT = ST_N0 + ST_N1 + ST_N2;

struct Rows {              // what a lane fetches for a tile: rows `lane` and `64 + lane` (the second only below the tile's row count)
    int src[2], ctr[2];
};
struct Pts {
    float p[2][6], c[2][6];
};

// One layer's MFMAs for a wave's RT row tiles x NC column tiles: acc[rt][ct] += W^T fragment (ks, ct) x activation fragment (rt, ks), k-steps
// ascending.  The operands of k-step ks + 1 (NC weight + RT activation fragments, all from LDS) are requested BEFORE the RT * NC MFMAs of k-step
// ks and the scheduler is pinned to that order (sched_group_barrier): left alone, hipcc placed each k-step's reads right in front of their first
// use and every k-step paid the LDS round trip (measured: 45 instead of 32 cycles per MFMA with the SIMD to itself).
template <int RT, int NC, int KS>
__device__ __forceinline__ void layer_mfma(const float *__restrict__ wfl, const float *__restrict__ ap, int lane, floatx4 (&acc)[RT][NC])
{
    float w[NC], x[RT];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) w[ct] = wfl[ct * 64 + lane];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) x[rt] = ap[rt * 16 * ST_STRIDE];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float wn[NC], xn[RT];
        if (ks + 1 < KS) {
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) wn[ct] = wfl[((ks + 1) * NC + ct) * 64 + lane];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) xn[rt] = ap[rt * 16 * ST_STRIDE + (ks + 1) * 4];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct], x[rt], acc[rt][ct], 0, 0, 0);
        if (ks + 1 < KS) {
            __builtin_amdgcn_sched_group_barrier(0x100, NC + RT, 0);      // the next k-step's LDS reads ...
            __builtin_amdgcn_sched_group_barrier(0x008, RT * NC, 0);      // ... then this k-step's MFMAs
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) w[ct] = wn[ct];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) x[rt] = xn[rt];
        }
    }
}

// ST_WAVES wavefronts per workgroup (one workgroup per CU): 12 = three per SIMD, 8 = two per SIMD.  The layers run MFMA-bound with the waves
// of a SIMD in lock-step (measured: layer 2 takes ST_WAVES / 4 x its 5 120 MFMA cycles), so a launch costs about rounds x ST_WAVES / 4 tile
// times with rounds = ceil(tiles / (CUs x ST_WAVES)): the launcher picks the count that minimises it (B = 32: 8 192 tiles = exactly 4 rounds
// of 2 048 waves, against 3 rounds of 3 072 of which the last is two-thirds empty).
// WIN: the launch computes a window of every cloud's centres (PaChain::win_len / win_off, pa_sa_group_window) -- an instantiation of its own, so the
// index arithmetic of the window costs the ordinary launch nothing (compiled into the one kernel it took 14 registers and 6 us of 51)
template <int RT, int ST_WAVES, bool WIN = false>
__global__ __launch_bounds__(ST_WAVES * 64) void sa_tiny_kernel(PaChain a, long ntiles)
{
    constexpr int R = RT * 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wf = smem, *bs = smem + ST_WTOT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
    float *act = smem + ST_WTOT + ST_BTOT + wave * (R * ST_STRIDE);

    // ---- the level's weights and biases, once per workgroup: wf_l[(ks * nct + ct) * 64 + l] = Wt_l[4 ks + l / 16][16 ct + l % 16]
    {
        const int off[3] = {0, ST_W0, ST_W0 + ST_W1}, nct[3] = {ST_N0 / 16, ST_N1 / 16, ST_N2 / 16}, cnt[3] = {ST_W0, ST_W1, ST_W2};
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const float *wt = a.L[l].wt;
            const int n = a.L[l].n;
            for (int q = tid; q < cnt[l]; q += ST_WAVES * 64) {
                const int ln = q & 63, f = q >> 6, ct = f % nct[l], ks = f / nct[l];
                wf[off[l] + q] = wt[(size_t)(4 * ks + (ln >> 4)) * n + 16 * ct + (ln & 15)];
            }
        }
        for (int q = tid; q < ST_BTOT; q += ST_WAVES * 64)
            bs[q] = q < ST_N0 ? a.L[0].bias[q] : q < ST_N0 + ST_N1 ? a.L[1].bias[q - ST_N0] : a.L[2].bias[q - ST_N0 - ST_N1];
    }
    __syncthreads();
    const float *wf0 = wf, *wf1 = wf + ST_W0, *wf2 = wf + ST_W0 + ST_W1;
    const float *b0 = bs, *b1 = bs + ST_N0, *b2 = bs + ST_N0 + ST_N1;

    const long nw = (long)gridDim.x * ST_WAVES;
    long tile = (long)blockIdx.x * ST_WAVES + wave;
    if (tile >= ntiles) return;                          // wave-uniform; no workgroup barrier below
    const int C = a.c_feat;

    auto load_rows = [&](long t, Rows &rw) {             // neighbour-major rows: row = slot * 4 + group (pa_chain.h chain_prologue, POOLED)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = h * 64 + lane;
            const long gid = t * 4 + (r & 3);
            int s = r >> 2;
            if (s >= a.ns) s = 0;
            rw.src[h] = -1;
            rw.ctr[h] = 0;
            if (r < R && gid < a.rows) {
                unsigned g32 = (unsigned)gid;                                       // groups < 2^31 (host-checked): 32-bit division, not the 64-bit loop
                if (WIN) { const unsigned wb = g32 / (unsigned)a.win_len; g32 = wb * (unsigned)a.m_ctr + (unsigned)a.win_off + (g32 - wb * (unsigned)a.win_len); }
                const unsigned b = g32 / (unsigned)a.m_ctr;
                rw.src[h] = (int)(b * (unsigned)a.n_src + (unsigned)a.nbr_idx[(size_t)g32 * a.ns + s]);
                rw.ctr[h] = (int)(b * (unsigned)a.n_src + (unsigned)a.center_idx[g32]);
            }
        }
    };
    // At the first level the features ARE the coordinates (both models feed xyz as the level-0 feature map): the neighbour's 12 bytes are then
    // fetched once, as one 3-dword load.  A scattered wave-load costs the texture path one cache line per lane whatever its width, and with twelve
    // waves per CU the six dword loads per row were as much L1 time as the tile's MFMAs.
    const bool same = a.feat == a.xyz && C == 3;             // workgroup-uniform
    typedef float f3 __attribute__((ext_vector_type(3)));
    auto load_pts = [&](const Rows &rw, Pts &pt) {       // raw coordinates / features of the neighbour and of its centre
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s = max(rw.src[h], 0), c = rw.ctr[h];
            const f3 ps = *reinterpret_cast<const f3 *>(a.xyz + (size_t)s * 3), pc = *reinterpret_cast<const f3 *>(a.xyz + (size_t)c * 3);
            pt.p[h][0] = ps.x; pt.p[h][1] = ps.y; pt.p[h][2] = ps.z;
            pt.c[h][0] = pc.x; pt.c[h][1] = pc.y; pt.c[h][2] = pc.z;
            if (same) {
#pragma unroll
                for (int t = 0; t < 3; ++t) { pt.p[h][3 + t] = pt.p[h][t]; pt.c[h][3 + t] = pt.c[h][t]; }
            } else {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    pt.p[h][3 + t] = t < C ? a.feat[(size_t)s * C + t] : 0.f;
                    pt.c[h][3 + t] = t < C ? a.feat[(size_t)c * C + t] : 0.f;
                }
            }
        }
    };
    auto write_tile = [&](const Rows &rw, const Pts &pt) {   // centred coordinates -> channels 0..2, centred features -> 3..3+C, zero padding to 8
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = h * 64 + lane;
            if (r >= R) continue;
            const bool live = rw.src[h] >= 0;
            float v[8];
#pragma unroll
            for (int t = 0; t < 6; ++t) v[t] = live ? pt.p[h][t] - pt.c[h][t] : 0.f;
            v[6] = 0.f;
            v[7] = 0.f;
            float2 *d = reinterpret_cast<float2 *>(act + r * ST_STRIDE);
            d[0] = make_float2(v[0], v[1]);
            d[1] = make_float2(v[2], v[3]);
            d[2] = make_float2(v[4], v[5]);
            d[3] = make_float2(v[6], v[7]);
        }
    };

    // ---- software pipeline over this wave's tiles: indices two tiles ahead, coordinates one tile ahead
    Rows rw_cur, rw_nxt;
    Pts pt_cur, pt_nxt;
    load_rows(tile, rw_cur);
    load_pts(rw_cur, pt_cur);
    const bool has1 = tile + nw < ntiles;
    if (has1) load_rows(tile + nw, rw_nxt);
    // The waves of a SIMD start together and do identical work, so left alone they sit in their gather / write-back phases at the same moments
    // and fight for the matrix pipe in the same moments.  Delaying every second wave of a SIMD by part of a tile ONCE (its first loads are
    // already in flight) puts them in anti-phase for the rest of their tile loops.  Pure scheduling.
    if ((wave >> 2) & 1)
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);

#define ST_STAMP(i) do { if (a.dbg && tile < 512 && lane == 0) a.dbg[tile * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
    for (; tile < ntiles; tile += nw) {
        ST_STAMP(0);
        write_tile(rw_cur, pt_cur);
        const bool more = tile + nw < ntiles;
        Rows rw_n2;
        if (more) {
            load_pts(rw_nxt, pt_nxt);                        // tile t + 1: coordinates (its indices arrived during tile t - 1)
            if (tile + 2 * nw < ntiles) load_rows(tile + 2 * nw, rw_n2);
        }
        lds_fence();
        ST_STAMP(1);

        const float *ap = act + li * ST_STRIDE + lq;          // activation fragment (MFMA B operand): point 16 rt + l % 16, channel 4 ks + l / 16
        // ---- layer 0: 8 -> 32
        {
            floatx4 acc[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
            layer_mfma<RT, 2, ST_K0 / 4>(wf0, ap, lane, acc);
            lds_fence();                                      // every fragment read of this layer has landed before its rows are overwritten
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = ct * 16 + lq * 4;
                const float4 bias = *reinterpret_cast<const float4 *>(b0 + col);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    float2 *d = reinterpret_cast<float2 *>(act + (rt * 16 + li) * ST_STRIDE + col);
                    d[0] = make_float2(fmaxf(acc[rt][ct][0] + bias.x, 0.f), fmaxf(acc[rt][ct][1] + bias.y, 0.f));
                    d[1] = make_float2(fmaxf(acc[rt][ct][2] + bias.z, 0.f), fmaxf(acc[rt][ct][3] + bias.w, 0.f));
                }
            }
            lds_fence();
        }
        ST_STAMP(2);
        // ---- layer 1: 32 -> 32
        {
            floatx4 acc[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
            layer_mfma<RT, 2, ST_N0 / 4>(wf1, ap, lane, acc);
            lds_fence();
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = ct * 16 + lq * 4;
                const float4 bias = *reinterpret_cast<const float4 *>(b1 + col);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    float2 *d = reinterpret_cast<float2 *>(act + (rt * 16 + li) * ST_STRIDE + col);
                    d[0] = make_float2(fmaxf(acc[rt][ct][0] + bias.x, 0.f), fmaxf(acc[rt][ct][1] + bias.y, 0.f));
                    d[1] = make_float2(fmaxf(acc[rt][ct][2] + bias.z, 0.f), fmaxf(acc[rt][ct][3] + bias.w, 0.f));
                }
            }
            lds_fence();
        }
        ST_STAMP(3);
        // ---- layer 2: 32 -> 64, max over the neighbourhood: a lane's point 16 rt + l % 16 belongs to group l % 4, so the max over a group is a max
        // across the row tiles (registers) and across the lanes l % 16 = g, g + 4, g + 8, g + 12 (two DPP row rotations); then bias + ReLU (both
        // monotone: exact) and one 16-byte store per group and channel quad
        {
            floatx4 acc[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
            layer_mfma<RT, 4, ST_N1 / 4>(wf2, ap, lane, acc);
            lds_fence();                                      // the tile is dead: the next iteration overwrites it
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                floatx4 m = acc[0][ct];
#pragma unroll
                for (int rt = 1; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], acc[rt][ct][r]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m[r] = fmaxf(m[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m[r]), 0x120 + 4, 0xf, 0xf, true)));   // row_ror:4
                    m[r] = fmaxf(m[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m[r]), 0x120 + 8, 0xf, 0xf, true)));   // row_ror:8
                }
                long grp = tile * 4 + li;
                const bool live = li < 4 && grp < a.rows;
                if (WIN) { const unsigned wb = (unsigned)grp / (unsigned)a.win_len; grp = (long)wb * a.m_ctr + a.win_off + ((unsigned)grp - wb * (unsigned)a.win_len); }
                if (live) {
                    const int col = ct * 16 + lq * 4;
                    const float4 bias = *reinterpret_cast<const float4 *>(b2 + col);
                    const float4 v = make_float4(fmaxf(m[0] + bias.x, 0.f), fmaxf(m[1] + bias.y, 0.f), fmaxf(m[2] + bias.z, 0.f), fmaxf(m[3] + bias.w, 0.f));
                    float *o = a.out + grp * a.ldo + col;
                    if (a.vec_out) *reinterpret_cast<float4 *>(o) = v;
                    else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
                }
            }
        }
        ST_STAMP(4);
        rw_cur = rw_nxt;
        pt_cur = pt_nxt;
        rw_nxt = rw_n2;
    }
#undef ST_STAMP
}

}  // namespace

// Does the persistent tiny-chain kernel take this launch?  (mlp_chain.hip chain_dispatch asks for pooled set-abstraction chains.)
bool pa_sa_tiny_applies(const PaChain &a, int rt)
{
    return a.nlayers == 3 && (rt == 4 || rt == 5) && a.L[0].kpad == ST_K0 && a.L[0].n == ST_N0 && a.L[1].kpad == ST_N0 && a.L[1].n == ST_N1 &&
           a.L[2].kpad == ST_N1 && a.L[2].n == ST_N2 && a.c_feat >= 1 && a.c_feat <= 3 && a.L[0].wt && a.L[1].wt && a.L[2].wt;
}

template <int RT, int W>
static void sa_tiny_launch_t(const PaChain &a_in, long ntiles, int cus, hipStream_t st)
{
    static const int stagger = getenv("PA_SA_TINY_STAGGER") ? atoi(getenv("PA_SA_TINY_STAGGER")) : 1;      // units of 8 128 cycles
    PaChain a = a_in;
    a.stagger = (ntiles + (long)W * cus - 1) / ((long)W * cus) >= 2 ? stagger : 0;                        // only when the waves loop
    const size_t lds = (size_t)(ST_WTOT + ST_BTOT + W * RT * 16 * ST_STRIDE) * 4;
    long grid = (ntiles + W - 1) / W;
    if (grid > cus) grid = cus;                              // one persistent workgroup per CU; waves loop over the tiles
    if (a.win_len > 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_tiny_kernel<RT, W, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((sa_tiny_kernel<RT, W, true>), dim3((unsigned)grid), dim3(W * 64), lds, st, a, ntiles);
        return;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_tiny_kernel<RT, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((sa_tiny_kernel<RT, W>), dim3((unsigned)grid), dim3(W * 64), lds, st, a, ntiles);
}

int pa_sa_tiny_launch(const PaChain &a, int rt, long ntiles, hipStream_t st)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    static const int forced = getenv("PA_SA_TINY_WAVES") ? atoi(getenv("PA_SA_TINY_WAVES")) : 0;     // A/B knob: 8 or 12
    const long r8 = (ntiles + 8L * cus - 1) / (8L * cus), r12 = (ntiles + 12L * cus - 1) / (12L * cus);
    const bool w8 = forced ? forced == 8 : r8 * 2 <= r12 * 3;      // cost ~ rounds x waves per SIMD
    if (rt == 5) { if (w8) sa_tiny_launch_t<5, 8>(a, ntiles, cus, st); else sa_tiny_launch_t<5, 12>(a, ntiles, cus, st); }
    else { if (w8) sa_tiny_launch_t<4, 8>(a, ntiles, cus, st); else sa_tiny_launch_t<4, 12>(a, ntiles, cus, st); }
    return 0;
}
