// The second set-abstraction level of both models (grouping x 2 + subtract + cat + SharedMLP 67 -> 64 -> 64 -> N2 + max over the 20
// neighbours; pointops.py:559-570, pt_util.py:16-41, patch_aug_net.py:236 / pptnet.py SA level 1) as a persistent kernel with the level's
// weights RESIDENT IN LDS and the activations in REGISTERS.
//
// Why: the shared-tile pooled kernel (chain_pooled.hip, four waves split the columns of one 80-row tile) streams the 99 KB of weights from
// L2 per tile and passes the hidden activations through LDS between barriers: 49 us at B = 32 where the MFMAs alone take 26 us.  Here one
// workgroup per CU packs the weights into LDS once; a WAVE owns an 80-row tile (4 groups x 20 neighbours = five 16-row MFMA tiles) from the
// gather to the pooled store, with no barrier after the weights have landed:
//   * v_mfma_f32_16x16x4_f32 with the weights as the A operand leaves a lane with acc[ct][i] = output channel 16 ct + 4 (l/16) + i of point
//     l%16 -- which IS the B operand (k slot l/16) of the next layer's k-step that contracts channels {16 ct + 4 g + i : g = 0..3}.  The hidden
//     layers never leave the registers; the weight fragments are packed for that contraction order (four k-steps i = 0..3 = one 16-byte LDS
//     read feeding 4 x RT MFMAs);
//   * the gather builds the first operand in the same order: a lane reads 16 bytes of its neighbour's (and centre's) feature row per 16
//     channels; the three centred coordinates are one k-step of their own (k slots 0..2 = dx, dy, dz);
//   * row 16 rt + l%16 of a tile is neighbour 4 rt + (l%16)/4 of group l%4 (as in sa_tiny.hip), so the max over a neighbourhood is a max over
//     the RT accumulators and two DPP row rotations; bias + ReLU after the max (both monotone: exact).
// Same values as the generic chain kernels up to the order of the fp32 additions inside a dot product (tests/test_gpu_chain.py: <= 2e-5
// of the tensor's scale against float64, like every other chain variant).
#include <stdlib.h>

#include "pa_chain.h"

namespace {

constexpr int SM_C = 64, SM_N0 = 64, SM_N1 = 64;

// Accumulator slot <-> channel.  The hidden layers use rho(ct, g, i) = 16 ct + 4 i + g: k-step ks = 4 ct + i of the next layer then contracts
// the CONSECUTIVE channels 4 ks + g, the order the standard fragment packing (pa_pack_weights) is in -- so layers 2 and 3 go global -> LDS
// verbatim (global_load_lds, no registers); a layer PRODUCES rho-order when lane m of its A fragment holds column 16 ct + 4 (m % 4) + m / 4,
// which for layer 2 is a permuted lane address of the LDS read (all 64 lanes still read distinct 16-byte words of one 1 KB fragment).  The
// last layer reads its fragments lane-linearly (slot = channel 16 ct + 4 g + i: 16-byte bias loads and stores).  Only the first layer (its K
// order is the gather's: 16 bytes of a feature row per lane) is packed by the kernel itself.
// LDS image (floats): w1f [4 ct][4 q][64 lanes][4]  W1[3 + 16 q + 4 (l/16) + i][16 ct + rho-lane(l%16)]
//                     w1x [4 ct][64 lanes]          W1[l/16][16 ct + rho-lane(l%16)], 0 for l/16 == 3
//                     w2  [16 ks][64][4]            pa_pack_weights(64, 64):  W2[4 ks + l/16][16 j + l%16]
//                     w3  [n2/64 cg][16 ks][64][4]  pa_pack_weights(64, n2):  W3[4 ks + l/16][64 cg + 16 j + l%16]
//                     bias [64 + 64 + n2]
__host__ __device__ constexpr int sm_lds_floats(int n2) { return 4096 + 256 + 4096 + n2 * 64 + 128 + n2; }

__device__ __forceinline__ int sm_rho_lane(int m) { return 4 * (m & 3) + (m >> 2); }

__device__ __forceinline__ void sm_pack(float *dst, const float *__restrict__ wt, int ldw, int kofs, int nct, int tid)
{
    // dst[((ct * 4 + q) * 64 + l) * 4 + i] = wt[(kofs + 16 q + 4 (l / 16) + i) * ldw + 16 ct + rho-lane(l % 16)]
    for (int e = tid; e < nct * 256; e += 256) {
        const int l = e & 63, q = (e >> 6) & 3, ct = e >> 8;
        const float *src = wt + (size_t)(kofs + 16 * q + 4 * (l >> 4)) * ldw + 16 * ct + sm_rho_lane(l & 15);
        *reinterpret_cast<float4 *>(dst + (size_t)e * 4) = make_float4(src[0], src[ldw], src[2 * ldw], src[3 * ldw]);
    }
}

// PREFETCH: the next tile's gather is issued after layer 1 (needs ~350 registers: one wave per SIMD owns the register file); without it the
// kernel fits 256 registers, so waves of OTHER streams' kernels can share the SIMDs -- the engine's launches at B = 32 have one tile per wave.
template <int RT, bool PREFETCH>
__global__ __launch_bounds__(256, PREFETCH ? 1 : 2) void sa_mid_kernel(PaChain a, long ntiles)
{
    extern __shared__ __attribute__((aligned(16))) float sm_lds[];
    const int n2 = a.L[2].n;
    float *w1f = sm_lds, *w1x = w1f + 4096, *w2 = w1x + 256, *w3 = w2 + 4096, *bs = w3 + n2 * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kg = lane >> 4;
    const long nblk = gridDim.x;
    const long blk = (a.xcd_remap && (nblk & 7) == 0) ? (long)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    const float4 *f4 = reinterpret_cast<const float4 *>(a.feat);

    // gather of one tile: raw neighbour / centre rows into registers (16 bytes per 16 channels and lane; coordinate kg), consumed at the top of
    // the tile's iteration -- the loads of tile t + 1 are issued after layer 1 of tile t and fly under its layers 2 and 3
    float4 pf[RT][4], cf[4];
    float px[RT], cx;
    auto gather = [&](long t) {
        long gid = t * 4 + (li & 3);
        if (gid >= a.rows) gid = a.rows - 1;   // ragged last tile: recomputed, never stored
        const long b = gid / a.m_ctr;
        const size_t ctr = (size_t)(b * a.n_src + a.center_idx[gid]);
        size_t src[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            int s = rt * 4 + (li >> 2);
            if (s >= a.ns) s = 0;              // padding rows repeat neighbour 0: the max is unchanged
            src[rt] = (size_t)(b * a.n_src + a.nbr_idx[gid * a.ns + s]);
        }
        const int kc = kg < 3 ? kg : 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) cf[q] = f4[ctr * 16 + 4 * q + kg];
        cx = a.xyz[ctr * 3 + kc];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pf[rt][q] = f4[src[rt] * 16 + 4 * q + kg];
            px[rt] = a.xyz[src[rt] * 3 + kc];
        }
    };

    long tile = blk * 4 + wave;
    if (tile < ntiles) gather(tile);   // in flight while the weights are packed

    // ---- weights -> LDS, once per workgroup: layers 2 and 3 verbatim (1 KB per wave instruction), layer 1 packed here -----------------------
    {
        const char *s2 = reinterpret_cast<const char *>(a.L[1].wp), *s3 = reinterpret_cast<const char *>(a.L[2].wp);
        const int n3 = n2 >> 2;   // KB of the last layer's packing (64 x n2 floats)
        for (int p = wave; p < 16 + n3; p += 4) {
            const char *src = p < 16 ? s2 + (size_t)p * 1024 : s3 + (size_t)(p - 16) * 1024;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + lane * 16),
                                             (void __attribute__((address_space(3))) *)(reinterpret_cast<char *>(w2) + (size_t)p * 1024), 16, 0, 0);
        }
    }
    sm_pack(w1f, a.L[0].wt, a.L[0].ldw, 3, 4, tid);
    {
        const int l = tid & 63, ct = tid >> 6;   // 256 threads = 4 ct x 64 lanes
        w1x[tid] = (l >> 4) < 3 ? a.L[0].wt[(size_t)(l >> 4) * a.L[0].ldw + 16 * ct + sm_rho_lane(l & 15)] : 0.f;
        for (int e = tid; e < 128 + n2; e += 256) bs[e] = e < 64 ? a.L[0].bias[e] : e < 128 ? a.L[1].bias[e - 64] : a.L[2].bias[e - 128];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const floatx4 *w1f4 = reinterpret_cast<const floatx4 *>(w1f) + lane;
    const floatx4 *w24 = reinterpret_cast<const floatx4 *>(w2) + kg * 16 + sm_rho_lane(li);   // layer 2 produces rho-order: permuted lane
    const floatx4 *w34 = reinterpret_cast<const floatx4 *>(w3) + lane;
    for (; tile < ntiles; tile += nblk * 4) {
        asm volatile("" ::: "memory");   // keeps the (loop-invariant) weight fragments out of registers across tiles: 128 of them would spill
        // ---- first operands: h0[rt][q][i] = centred feature 16 q + 4 kg + i, hx[rt] = centred coordinate kg (k slot 3 = 0) -------------------------
        floatx4 h0[RT][4];
        float hx[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                h0[rt][q] = (floatx4){pf[rt][q].x - cf[q].x, pf[rt][q].y - cf[q].y, pf[rt][q].z - cf[q].z, pf[rt][q].w - cf[q].w};
            hx[rt] = kg < 3 ? px[rt] - cx : 0.f;
        }
        // ---- layer 1: 3 + 64 -> 64 ---------------------------------------------------------------------------------------------------------
        floatx4 h1[RT][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            floatx4 acc[RT];
            const float wx = w1x[ct * 64 + lane];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wx, hx[rt], (floatx4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const floatx4 w = w1f4[(ct * 4 + q) * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i], h0[rt][q][i], acc[rt], 0, 0, 0);
            }
            const float *bp = bs + 16 * ct + kg;   // slot (ct, kg, i) = channel 16 ct + 4 i + kg
            const float bz[4] = {bp[0], bp[4], bp[8], bp[12]};
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                h1[rt][ct] = (floatx4){fmaxf(acc[rt][0] + bz[0], 0.f), fmaxf(acc[rt][1] + bz[1], 0.f), fmaxf(acc[rt][2] + bz[2], 0.f), fmaxf(acc[rt][3] + bz[3], 0.f)};
        }
        // the next tile's gather flies under layers 2 and 3
        const long next = tile + nblk * 4;
        if (PREFETCH && next < ntiles) gather(next);
        // ---- layer 2: 64 -> 64: k-step ks = 4 ct1 + i contracts channels 4 ks + g = h1[.][ct1][i]; one fragment = the four column tiles ----------
        floatx4 h2[RT][4];
        {
            floatx4 acc[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[rt][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const floatx4 w = w24[ks * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], h1[rt][ks >> 2][ks & 3], acc[rt][j], 0, 0, 0);
            }
            // fragment reads run two ahead of the 4 RT MFMAs they feed (one wave per SIMD: nobody else hides the LDS latency)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int ks = 0; ks < 14; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8 * RT, 0);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const float *bp = bs + 64 + 16 * ct + kg;
                const float bz[4] = {bp[0], bp[4], bp[8], bp[12]};
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    h2[rt][ct] = (floatx4){fmaxf(acc[rt][ct][0] + bz[0], 0.f), fmaxf(acc[rt][ct][1] + bz[1], 0.f), fmaxf(acc[rt][ct][2] + bz[2], 0.f),
                                           fmaxf(acc[rt][ct][3] + bz[3], 0.f)};
            }
        }
        // ---- layer 3: 64 -> n2, max over the neighbourhood, bias + ReLU, one 16-byte store per group and channel quad -------------------------
        const long grp = tile * 4 + li;
        const bool live = li < 4 && grp < a.rows;
        for (int cg = 0; cg < (n2 >> 6); ++cg) {
            floatx4 acc[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[rt][j] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                const floatx4 w = w34[(cg * 16 + ks) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], h2[rt][ks >> 2][ks & 3], acc[rt][j], 0, 0, 0);
            }
            // fragment reads run two ahead of the 4 RT MFMAs they feed (one wave per SIMD: nobody else hides the LDS latency)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int ks = 0; ks < 14; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8 * RT, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                floatx4 m = acc[0][j];
#pragma unroll
                for (int rt = 1; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], acc[rt][j][r]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m[r] = fmaxf(m[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m[r]), 0x120 + 4, 0xf, 0xf, true)));   // row_ror:4
                    m[r] = fmaxf(m[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m[r]), 0x120 + 8, 0xf, 0xf, true)));   // row_ror:8
                }
                if (live) {
                    const int col = 64 * cg + 16 * j + 4 * kg;
                    const float4 bz = *reinterpret_cast<const float4 *>(bs + 128 + col);
                    const float4 v = make_float4(fmaxf(m[0] + bz.x, 0.f), fmaxf(m[1] + bz.y, 0.f), fmaxf(m[2] + bz.z, 0.f), fmaxf(m[3] + bz.w, 0.f));
                    float *o = a.out + grp * a.ldo + col;
                    if (a.vec_out) *reinterpret_cast<float4 *>(o) = v;
                    else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
                }
            }
        }
        if (!PREFETCH && next < ntiles) gather(next);   // register-lean build: the next tile's rows are requested when this tile is done
    }
}

}  // namespace

// Does the LDS-resident second-level kernel take this launch?  (mlp_chain.hip chain_dispatch asks for pooled fp32 set-abstraction chains.)
bool pa_sa_mid_applies(const PaChain &a, int rt)
{
    return a.nlayers == 3 && (rt == 4 || rt == 5) && a.c_feat == SM_C && a.L[0].kpad == 68 && a.L[0].n == SM_N0 && a.L[1].kpad == SM_N0 && a.L[1].n == SM_N1 &&
           a.L[2].kpad == SM_N1 && a.L[2].n % 64 == 0 && a.L[2].n >= 64 && a.L[2].n <= 256 && a.L[0].wt && a.L[1].wp && a.L[2].wp &&
           (reinterpret_cast<uintptr_t>(a.feat) & 15) == 0 && a.win_len == 0;
}

int pa_sa_mid_launch(const PaChain &a, int rt, long ntiles, hipStream_t st)
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { pa_set_error("pa_mlp_chain(sa_mid): hipGetDeviceProperties failed"); return PA_EINVAL; }
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const size_t lds = (size_t)sm_lds_floats(a.L[2].n) * 4;
    long wgs = (ntiles + 3) / 4;
    if (wgs > cus) wgs = cus;
    static const int pf_env = getenv("PA_SA_MID_PREFETCH") ? atoi(getenv("PA_SA_MID_PREFETCH")) : -1;   // A/B knob
    const bool prefetch = pf_env >= 0 ? pf_env != 0 : ntiles > wgs * 4;
#define SM_LAUNCH(RTV, PF)                                                                                                                      \
    do {                                                                                                                                        \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_mid_kernel<RTV, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((sa_mid_kernel<RTV, PF>), dim3(wgs), dim3(256), lds, st, a, ntiles);                                                 \
    } while (0)
    if (rt == 5) { if (prefetch) SM_LAUNCH(5, true); else SM_LAUNCH(5, false); }
    else { if (prefetch) SM_LAUNCH(4, true); else SM_LAUNCH(4, false); }
#undef SM_LAUNCH
    return PA_OK;
}
