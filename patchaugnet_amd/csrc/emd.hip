// Earth mover's distance by the auction algorithm -- forward (assignment + squared distances) and backward.
//
// Replaces libs/emd_module/emd_cuda.cu:228-282 (emd_cuda_forward: `iters` rounds of the seven kernels clear /
// calc_unass_cnt / calc_unass_cnt_sum / calc_unass_idx / Bid / GetMax / Assign, then CalcDist) and :284-317
// (NmDistanceGradKernel).  The reference pays 7 launches per round (7 168 launches at iters = 1024) and re-reads xyz2
// and the prices from global memory in every Bid block.
//
// MI355X plan: ONE persistent workgroup (16 wavefronts) per cloud pair runs every round on chip.
//   * xyz2 (SoA) and the prices live in LDS for the whole launch (16 n bytes, n <= 8192), the list of unassigned
//     points is rebuilt each round with ballot/popcount prefix sums into LDS (ascending order);
//   * Bid: one wavefront per unassigned point, lanes stride the objects (conflict-free LDS reads), the
//     (best, best index, second best) triple is merged across the wavefront with xor-shuffles;
//   * GetMax / Assign are per-point passes separated by workgroup barriers; nothing leaves the CU between rounds,
//     and a cloud whose points are all assigned stops early (later rounds are no-ops in the reference too).
//
// Determinism.  The reference's result depends on thread timing (GetMax/Assign "last writer wins", :181-215) and on a
// data-dependent split of the object range in Bid (:108-118).  This kernel resolves those free choices exactly like
// the CPU oracle (oracle/pointops_oracle.c, oracle_emd_forward): lowest object index among equal values, highest
// bidder index among increments that match the maximum within 1e-6, evictions seen only by the next round, and on
// the last round bidders are applied in ascending order.  Arithmetic follows the source: value =
// (float)(3.0 - (double)sqrtf(d2) - (double)price) (the literal 3.0 is a double, :146), increment =
// (best - better) + eps in fp32.
#include <stdlib.h>

#include "pa_common.h"

namespace {

constexpr int EMD_THREADS = 1024;
constexpr int EMD_WAVES = EMD_THREADS / 64;
#ifndef EMD_DIRECT_MAX
#define EMD_DIRECT_MAX 8     // bidders of a workgroup up to which its scan reads global memory instead of staging the objects in LDS (0 = never)
#endif

struct Bid3 {
    float best, better;
    int best_i;
};

__device__ __forceinline__ Bid3 bid_merge(Bid3 a, float ob, float obt, int oi)
{
    // order statistics of the union: best = max (lowest index among equals), better = second largest of the multiset
    const bool other_wins = (ob > a.best) || (ob == a.best && oi < a.best_i);
    const float lo = other_wins ? a.best : ob;
    Bid3 r;
    r.best = other_wins ? ob : a.best;
    r.best_i = other_wins ? oi : a.best_i;
    r.better = fmaxf(fmaxf(a.better, obt), lo);
    return r;
}

// One lane's share of a bidder's scan over the objects k0 + lane, k0 + lane + 64, ... < k1 (Bid, emd_cuda.cu:95-179): value = (float)(3.0 - (double)
// sqrtf(d2) - (double)price) exactly as the source writes it (:146: the literal is a double), best / second best with the lowest index among equal
// values.  Four objects per trip: the square roots and the double-precision subtractions of independent objects overlap each other and the LDS
// reads (one object per trip ran at ~570 cycles per trip with one wavefront per SIMD); the running (best, second) update stays in ascending k.
__device__ __forceinline__ Bid3 bid_scan(const float *x2, const float *y2, const float *z2, const float *pr, float x1, float y1, float z1, int k0, int k1, int lane)
{
    Bid3 b = {-1e9f, -1e9f, -1};
    auto value = [&](int k) {
        const float dx = x2[k] - x1, dy = y2[k] - y1, dz = z2[k] - z1;
        return (float)(3.0 - (double)sqrtf(dx * dx + dy * dy + dz * dz) - (double)pr[k]);
    };
    auto take = [&](float d, int k) {
        if (d > b.best) {
            b.better = b.best;
            b.best = d;
            b.best_i = k;
        } else if (d > b.better) {
            b.better = d;
        }
    };
    int k = k0 + lane;
    for (; k + 192 < k1; k += 256) {
        const float d0 = value(k), d1 = value(k + 64), d2 = value(k + 128), d3 = value(k + 192);
        take(d0, k); take(d1, k + 64); take(d2, k + 128); take(d3, k + 192);
    }
    for (; k < k1; k += 64) take(value(k), k);
    return b;
}

__device__ __forceinline__ float ld_agent_f(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent_i(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent_f(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent_i(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// bid_scan straight from global memory (objects (n, 3) as the caller passes them, this round's prices with agent-scope loads): for the workgroups of
// the auction's long tail, which have one or two bidders -- staging all n objects and prices in LDS first (64 KB, sixteen loads a thread and a barrier)
// costs more than the few objects a lane then looks at.  Same operations in the same order on the same values: bit-identical bids.
__device__ __forceinline__ Bid3 bid_scan_direct(const float *xyz2, const float *price, float x1, float y1, float z1, int k0, int k1, int lane)
{
    Bid3 b = {-1e9f, -1e9f, -1};
    auto take = [&](float d, int k) {
        if (d > b.best) {
            b.better = b.best;
            b.best = d;
            b.best_i = k;
        } else if (d > b.better) {
            b.better = d;
        }
    };
    int k = k0 + lane;
    for (; k + 192 < k1; k += 256) {
        float ox[4], oy[4], oz[4], op[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kk = k + 64 * t;
            ox[t] = xyz2[kk * 3 + 0]; oy[t] = xyz2[kk * 3 + 1]; oz[t] = xyz2[kk * 3 + 2];
            op[t] = ld_agent_f(price + kk);
        }
        float d[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float dx = ox[t] - x1, dy = oy[t] - y1, dz = oz[t] - z1;
            d[t] = (float)(3.0 - (double)sqrtf(dx * dx + dy * dy + dz * dz) - (double)op[t]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) take(d[t], k + 64 * t);
    }
    for (; k < k1; k += 64) {
        const float dx = xyz2[k * 3 + 0] - x1, dy = xyz2[k * 3 + 1] - y1, dz = xyz2[k * 3 + 2] - z1;
        take((float)(3.0 - (double)sqrtf(dx * dx + dy * dy + dz * dz) - (double)ld_agent_f(price + k)), k);
    }
    return b;
}

// emd_cuda.cu:10-20 -- float atomicMax by compare-and-swap (replace only when val > stored)
__device__ __forceinline__ void atomic_max_f(float *addr, float val)
{
    int old = __float_as_int(ld_agent_f(addr));
    while (val > __int_as_float(old)) {
        const int seen = atomicCAS((int *)addr, old, __float_as_int(val));
        if (seen == old) break;
        old = seen;
    }
}

__global__ __launch_bounds__(EMD_THREADS) void emd_auction_kernel(int n, const float *__restrict__ xyz1_all, const float *__restrict__ xyz2_all,
                                                                  float *dist_all, int *assignment_all, float *price_all, int *assignment_inv_all,
                                                                  int *bid_all, float *bid_inc_all, float *max_inc_all, int *max_idx_all, float eps,
                                                                  int iters)
{
    extern __shared__ float lds[];
    float *x2 = lds, *y2 = lds + n, *z2 = lds + 2 * n, *pr = lds + 3 * n;
    unsigned short *unass = (unsigned short *)(lds + 4 * n);
    __shared__ int wcnt[EMD_WAVES];

    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t off = (size_t)i * n;
    const float *xyz1 = xyz1_all + off * 3, *xyz2 = xyz2_all + off * 3;
    int *ass = assignment_all + off, *ass_inv = assignment_inv_all + off, *bid = bid_all + off, *max_idx = max_idx_all + off;
    float *price = price_all + off, *binc = bid_inc_all + off, *minc = max_inc_all + off, *dist = dist_all + off;

    for (int k = tid; k < n; k += EMD_THREADS) {
        x2[k] = xyz2[k * 3 + 0];
        y2[k] = xyz2[k * 3 + 1];
        z2[k] = xyz2[k * 3 + 2];
        pr[k] = price[k];
    }
    __syncthreads();

    for (int it = 0; it < iters; ++it) {
        const bool last = (it == iters - 1);
        // ---- list the unassigned points in ascending order (calc_unass_cnt / _sum / _idx, :30-93) -------------
        int U = 0;
        for (int c = 0; c < n; c += EMD_THREADS) {
            const int j = c + tid;
            const bool un = ld_agent_i(ass + j) == -1;
            const u64 mask = __ballot(un);
            if (lane == 0) wcnt[wave] = __popcll(mask);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < EMD_WAVES; ++w) {
                const int cw = wcnt[w];
                before += (w < wave) ? cw : 0;
                total += cw;
            }
            if (un) unass[U + before + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)j;
            U += total;
            __syncthreads();
        }
        if (U == 0) break;   // every later round would find nothing to do

        // ---- Bid (:95-179): one wavefront per unassigned point ----------------------------------------------
        for (int u = wave; u < U; u += EMD_WAVES) {
            const int j = unass[u];
            const float x1 = xyz1[j * 3 + 0], y1 = xyz1[j * 3 + 1], z1 = xyz1[j * 3 + 2];
            Bid3 b = bid_scan(x2, y2, z2, pr, x1, y1, z1, 0, n, lane);
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const float ob = __shfl_xor(b.best, s), obt = __shfl_xor(b.better, s);
                const int oi = __shfl_xor(b.best_i, s);
                b = bid_merge(b, ob, obt, oi);
            }
            if (lane == 0) {
                const float inc = b.best - b.better + eps;
                bid[j] = b.best_i;
                binc[j] = inc;
                if (b.best_i >= 0) atomic_max_f(minc + b.best_i, inc);
            }
        }
        __syncthreads();

        // ---- GetMax (:181-194): the highest-indexed bidder whose increment matches the maximum within 1e-6 ---
        for (int pass = 0; pass < 2; ++pass) {
            for (int u = tid; u < U; u += EMD_THREADS) {
                const int j = unass[u];
                const int bid_id = bid[j];
                if (bid_id < 0) continue;
                const double bi = (double)binc[j], mx = (double)ld_agent_f(minc + bid_id);
                if (bi - 1e-6 <= mx && mx <= bi + 1e-6) {
                    if (pass == 0) st_agent_i(max_idx + bid_id, -1);
                    else atomicMax(max_idx + bid_id, j);
                }
            }
            __syncthreads();
        }

        // ---- Assign (:196-215) --------------------------------------------------------------------------------
        if (!last) {
            for (int u = tid; u < U; u += EMD_THREADS) {
                const int j = unass[u];
                const int bid_id = bid[j];
                if (bid_id < 0 || ld_agent_i(max_idx + bid_id) != j) continue;
                const int prev = ld_agent_i(ass_inv + bid_id);
                if (prev != -1) st_agent_i(ass + prev, -1);
                st_agent_i(ass_inv + bid_id, j);
                st_agent_i(ass + j, bid_id);
                pr[bid_id] += binc[j];
                st_agent_f(minc + bid_id, -1e9f);
            }
        } else if (tid == 0) {
            // last round: every unassigned point takes the object it bid for, applied in ascending point order
            for (int u = 0; u < U; ++u) {
                const int j = unass[u];
                const int bid_id = bid[j];
                st_agent_i(ass + j, bid_id);
                if (bid_id < 0) continue;
                st_agent_i(ass_inv + bid_id, j);
                pr[bid_id] += binc[j];
                st_agent_f(minc + bid_id, -1e9f);
            }
        }
        __syncthreads();
    }

    // ---- CalcDist (:217-226) + the prices back to the caller's state tensor -------------------------------------
    for (int j = tid; j < n; j += EMD_THREADS) {
        price[j] = pr[j];
        const int k = ld_agent_i(ass + j);
        float d = 0.f;
        if (k >= 0) {
            const float dx = xyz1[j * 3 + 0] - x2[k], dy = xyz1[j * 3 + 1] - y2[k], dz = xyz1[j * 3 + 2] - z2[k];
            d = dx * dx + dy * dy + dz * dz;
        }
        dist[j] = d;
    }
}

// ---- chip-wide form: ONE LAUNCH PER ROUND, G workgroups per cloud pair --------------------------------------------------------------------
// The persistent kernel above keeps a cloud pair on ONE CU: at the reference's call shape (16, 4096, 3) that is 16 of 256 CUs, and a round costs
// ~1 us per bidder (a wavefront scans all n objects) however few bidders are left -- 19.5 ms for 64 rounds, 53 ms for 1024.  Here the bidders of
// a round are dealt to G workgroups per cloud (slices of the point range; Bid is independent per bidder, its only shared write is the atomic
// maximum), and the per-object GetMax / Assign pass, which needs every bid of the cloud, is done by whichever of the G workgroups finishes LAST
// (an arrival counter per cloud; producers release with __threadfence, the last arriver acquires).  The kernel boundary between rounds is the
// only grid-wide synchronisation, so nothing can dead-lock against other streams' work.  Prices live in global memory between rounds (16 KB per
// cloud from L2 per workgroup and round).  Same deterministic choices, same arithmetic, same state as the persistent kernel and the oracle.
__global__ __launch_bounds__(EMD_THREADS) void emd_round_kernel(int n, const float *__restrict__ xyz1_all, const float *__restrict__ xyz2_all,
                                                                 float *dist_all, int *assignment_all, float *price_all, int *assignment_inv_all,
                                                                 int *bid_all, float *bid_inc_all, float *max_inc_all, int *max_idx_all, float eps,
                                                                 int last)
{
    extern __shared__ float lds[];
    float *x2 = lds, *y2 = lds + n, *z2 = lds + 2 * n, *pr = lds + 3 * n;
    unsigned short *unass = (unsigned short *)(lds + 4 * n);
    __shared__ int wcnt[EMD_WAVES];
    __shared__ int is_last, list_base;
    __shared__ Bid3 part_bid[EMD_WAVES];

    const int i = blockIdx.y, g = blockIdx.x, G = gridDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t off = (size_t)i * n;
    const float *xyz1 = xyz1_all + off * 3, *xyz2 = xyz2_all + off * 3;
    int *ass = assignment_all + off, *ass_inv = assignment_inv_all + off, *bid = bid_all + off, *max_idx = max_idx_all + off;
    float *price = price_all + off, *binc = bid_inc_all + off, *minc = max_inc_all + off, *dist = dist_all + off;
    // Per-call round state in the cloud's `dist` row, which holds nothing until the squared distances are written after the last round (two calls in
    // flight never share it): words 0 .. n/2 - 1 = this round's bidders (16-bit point indices, any order), word n/2 = arrival counter,
    // word n/2 + 1 = number of bidders.  Both counters are zeroed by the launcher and left at zero by every round's last arriver.
    unsigned short *glist = reinterpret_cast<unsigned short *>(dist);
    int *arrive = reinterpret_cast<int *>(dist) + n / 2, *gcount = arrive + 1;

    // ---- this workgroup's bidders: the unassigned points of its slice of the point range ---------------------------------------------------
    const int per = (n + G - 1) / G, s0 = g * per, s1 = min(s0 + per, n);
    int U = 0;
    for (int c = s0; c < s1; c += EMD_THREADS) {
        const int j = c + tid;
        const bool un = j < s1 && ld_agent_i(ass + j) == -1;
        const u64 mask = __ballot(un);
        if (lane == 0) wcnt[wave] = __popcll(mask);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < EMD_WAVES; ++w) {
            const int cw = wcnt[w];
            before += (w < wave) ? cw : 0;
            total += cw;
        }
        if (un) unass[U + before + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)j;
        U += total;
        __syncthreads();
    }
    if (U > 0) {
        // the objects and this round's prices: only workgroups that have a bidder pay for the 16 n bytes -- and with one or two bidders (EMD_DIRECT_MAX)
        // not even those: the scan reads its few objects per lane straight from global memory (bid_scan_direct)
        const bool direct = U <= EMD_DIRECT_MAX;                  // workgroup-uniform
        if (!direct)
            for (int k = tid; k < n; k += EMD_THREADS) {
                x2[k] = xyz2[k * 3 + 0];
                y2[k] = xyz2[k * 3 + 1];
                z2[k] = xyz2[k * 3 + 2];
                pr[k] = ld_agent_f(price + k);
            }
        if (tid == 0) list_base = atomicAdd(gcount, U);          // this workgroup's segment of the cloud's bidder list
        __syncthreads();
        for (int u = tid; u < U; u += EMD_THREADS) __hip_atomic_store(glist + list_base + u, unass[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- Bid (:95-179).  A wavefront per bidder; with fewer bidders than wavefronts (the long tail of the auction: a handful per round) W
        // wavefronts share a bidder, each scanning 1/W of the objects, and the partial (best, second, index) triples are merged in object order.
        int W = 1;
        while (W < EMD_WAVES && U * W * 2 <= EMD_WAVES) W *= 2;
        const int slice = n / W;                                  // n is a multiple of 1024
        for (int u0 = 0; u0 < U; u0 += EMD_WAVES / W) {
            const int u = u0 + wave / W, part = wave % W;
            Bid3 b = {-1e9f, -1e9f, -1};
            int j = -1;
            if (u < U) {
                j = unass[u];
                const float x1 = xyz1[j * 3 + 0], y1 = xyz1[j * 3 + 1], z1 = xyz1[j * 3 + 2];
                b = direct ? bid_scan_direct(xyz2, price, x1, y1, z1, part * slice, (part + 1) * slice, lane)
                           : bid_scan(x2, y2, z2, pr, x1, y1, z1, part * slice, (part + 1) * slice, lane);
#pragma unroll
                for (int s = 1; s < 64; s <<= 1) {
                    const float ob = __shfl_xor(b.best, s), obt = __shfl_xor(b.better, s);
                    const int oi = __shfl_xor(b.best_i, s);
                    b = bid_merge(b, ob, obt, oi);
                }
            }
            if (W > 1) {                                          // W is workgroup-uniform
                if (lane == 0) part_bid[wave] = b;
                __syncthreads();
                if (part == 0 && u < U)
                    for (int q = 1; q < W; ++q) { const Bid3 o = part_bid[wave + q]; b = bid_merge(b, o.best, o.better, o.best_i); }
                __syncthreads();
            }
            if (lane == 0 && part == 0 && u < U) {
                const float inc = b.best - b.better + eps;
                st_agent_i(bid + j, b.best_i);
                st_agent_f(binc + j, inc);
                if (b.best_i >= 0) atomic_max_f(minc + b.best_i, inc);
            }
        }
    }
    // ---- arrival: the last of the cloud's G workgroups resolves the round --------------------------------------------------------------------
    // Everything another workgroup reads from this one (bidder list, bid, increment, the atomic maximum) was written with agent-scope write-through
    // stores / atomics and is read back with agent-scope loads that bypass the L1: publishing needs every wave's stores RETIRED (vmcnt 0), the
    // workgroup barrier, and ONE lane's release in front of the arrival -- not 16 wavefronts x 512 workgroups of L2 write-back fences (that form
    // measured 176 us per round).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (hipcc 7.2 may drop the wait behind the write-back when it thinks nothing is outstanding)
        const int last_one = (atomicAdd(arrive, 1) == G - 1) ? 1 : 0;
        if (last_one) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            st_agent_i(arrive, 0);                          // ready for the next round's launch
        }
        is_last = last_one;
    }
    __syncthreads();
    if (!is_last) return;
    U = ld_agent_i(gcount);                                  // the cloud's bidders of this round
    __syncthreads();
    if (tid == 0) st_agent_i(gcount, 0);
    for (int u = tid; u < U; u += EMD_THREADS) unass[u] = __hip_atomic_load(glist + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // GetMax (:181-194): the highest-indexed bidder whose increment matches the maximum within 1e-6.  max_idx[k] is left at -1 by the winner
    // that consumes it (Assign below) and starts at the caller's 0, which a real bidder index can only raise: no separate reset pass.
    for (int u = tid; u < U; u += EMD_THREADS) {
        const int j = unass[u];
        const int bid_id = ld_agent_i(bid + j);
        if (bid_id < 0) continue;
        const double bi = (double)ld_agent_f(binc + j), mx = (double)ld_agent_f(minc + bid_id);
        if (bi - 1e-6 <= mx && mx <= bi + 1e-6) atomicMax(max_idx + bid_id, j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the atomics are at the L2 before anyone reads them
    __syncthreads();
    // Assign (:196-215): prices are updated in GLOBAL memory (the next round's workgroups load them)
    if (!last) {
        for (int u = tid; u < U; u += EMD_THREADS) {
            const int j = unass[u];
            const int bid_id = ld_agent_i(bid + j);
            if (bid_id < 0 || ld_agent_i(max_idx + bid_id) != j) continue;
            const int prev = ld_agent_i(ass_inv + bid_id);
            if (prev != -1) st_agent_i(ass + prev, -1);
            st_agent_i(ass_inv + bid_id, j);
            st_agent_i(ass + j, bid_id);
            st_agent_f(price + bid_id, ld_agent_f(price + bid_id) + ld_agent_f(binc + j));     // one winner per object: nobody else touches this price
            st_agent_f(minc + bid_id, -1e9f);
            st_agent_i(max_idx + bid_id, -1);
        }
    } else {
        // last round: every bidder takes the object it bid for, applied in ASCENDING point order (several may take the same object): sort the list
        // (any order until now) by counting ranks -- U is small by then, and n^2 / 1024 comparisons per thread even when it is not
        for (int u = tid; u < U; u += EMD_THREADS) {
            const unsigned short mine = unass[u];
            int rank = 0;
            for (int t = 0; t < U; ++t) rank += unass[t] < mine ? 1 : 0;
            reinterpret_cast<unsigned short *>(pr)[rank] = mine;     // pr[] (LDS) is free: this workgroup may not even have loaded it
        }
        __syncthreads();
        if (tid == 0) {
            const unsigned short *srt = reinterpret_cast<const unsigned short *>(pr);
            for (int u = 0; u < U; ++u) {
                const int j = srt[u];
                const int bid_id = ld_agent_i(bid + j);
                st_agent_i(ass + j, bid_id);
                if (bid_id < 0) continue;
                st_agent_i(ass_inv + bid_id, j);
                st_agent_f(price + bid_id, ld_agent_f(price + bid_id) + ld_agent_f(binc + j));
                st_agent_f(minc + bid_id, -1e9f);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // CalcDist (:217-226), straight from global memory (this workgroup may have had no bidder and no LDS copy of the objects)
        for (int j = tid; j < n; j += EMD_THREADS) {
            const int k = ld_agent_i(ass + j);
            float d = 0.f;
            if (k >= 0) {
                const float dx = xyz1[j * 3 + 0] - xyz2[k * 3 + 0], dy = xyz1[j * 3 + 1] - xyz2[k * 3 + 1], dz = xyz1[j * 3 + 2] - xyz2[k * 3 + 2];
                d = dx * dx + dy * dy + dz * dz;
            }
            dist[j] = d;
        }
    }
}


// NmDistanceGradKernel (:284-300): grad_xyz[i,j] += 2 g (xyz1[i,j] - xyz2[i,idx[i,j]]); one thread owns one point.
__global__ void emd_backward_kernel(long total, int n, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                    const float *__restrict__ grad_dist, const int *__restrict__ idx, float *grad_xyz)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const long i = t / n;
    const int j2 = idx[t];
    const float g = grad_dist[t] * 2;
    const float *p = xyz1 + t * 3, *q = xyz2 + (i * n + j2) * 3;
    float *o = grad_xyz + t * 3;
    o[0] += g * (p[0] - q[0]);
    o[1] += g * (p[1] - q[1]);
    o[2] += g * (p[2] - q[2]);
}

}  // namespace

static int g_emd_persistent = -1;
// test / A/B switch: 1 = the persistent one-workgroup-per-cloud kernel, 0 = the chip-wide one-launch-per-round form (default; PA_EMD_PERSISTENT=1 flips it)
// 1 = the one-workgroup persistent kernel, 0 = chip-wide round launches only, -1 = the default rule (round launches); 2 (test-only library) = the
// resident multi-round form
PA_API void pa_emd_persistent_enable(int on) { g_emd_persistent = on < 0 ? -1 : (on > 2 ? 2 : on); }

PA_API int pa_emd_forward(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *assignment, float *price,
                          int *assignment_inv, int *bid, float *bid_increments, float *max_increments, int *max_idx, float eps, int iters,
                          pa_stream_t stream)
{
    // emd_cuda.cu:236-249 -- the reference's shape rules (it returns -1 for these)
    if (n != m) { pa_set_error("pa_emd_forward: the two point clouds must have the same size (n=%d, m=%d)", n, m); return PA_EUNSUPPORTED; }
    if (b > 512) { pa_set_error("pa_emd_forward: batch size %d > 512", b); return PA_EUNSUPPORTED; }
    if (n % 1024 != 0 || n <= 0) { pa_set_error("pa_emd_forward: n=%d is not a positive multiple of 1024", n); return PA_EUNSUPPORTED; }
    if (n > 8192) { pa_set_error("pa_emd_forward: n=%d > 8192 does not fit the on-chip auction state", n); return PA_EUNSUPPORTED; }
    PA_REQUIRE(b > 0 && iters >= 1, "pa_emd_forward: need b >= 1 and iters >= 1 (b=%d, iters=%d)", b, iters);
    PA_REQUIRE(xyz1 && xyz2 && dist && assignment && price && assignment_inv && bid && bid_increments && max_increments && max_idx,
               "pa_emd_forward: null pointer");
    const size_t lds_bytes = (size_t)n * 16 + (size_t)n * 2;
    static const bool persistent = getenv("PA_EMD_PERSISTENT") != nullptr;      // A/B and test knob: one workgroup per cloud, every round on chip
    if (g_emd_persistent == 1 || (g_emd_persistent < 0 && persistent)) {
        // the opt-in is a per-DEVICE attribute of the function: set it on every call (a process may drive several GPUs)
        hipError_t e = hipFuncSetAttribute((const void *)emd_auction_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e != hipSuccess) { pa_set_error("pa_emd_forward: cannot raise the LDS limit: %s", hipGetErrorString(e)); return (int)e; }
        hipLaunchKernelGGL(emd_auction_kernel, dim3(b), dim3(EMD_THREADS), lds_bytes, (hipStream_t)stream, n, xyz1, xyz2, dist, assignment, price,
                           assignment_inv, bid, bid_increments, max_increments, max_idx, eps, iters);
        PA_CHECK_LAUNCH("pa_emd_forward");
        return PA_OK;
    }
    // chip-wide: G workgroups per cloud, ONE workgroup per CU in total.  (Two per CU -- 512 workgroups of 1024 threads and 73 KB of LDS at the reference's
    // call shape -- was the first tuning; the long tail of the auction has a handful of bidders per round and a round then costs what the launch costs:
    // 21 us with 512 workgroups, 15.8 us with 256, and the early rounds do not lose either: (16, 4096, 3) 64 / 1024 rounds 3.11 / 23.3 -> 2.49 / 17.6 ms.)
    int cus_emd = 256;
    {
        int dev_emd = 0;
        if (hipGetDevice(&dev_emd) == hipSuccess) (void)hipDeviceGetAttribute(&cus_emd, hipDeviceAttributeMultiprocessorCount, dev_emd);
    }
    int G = cus_emd / b;
    if (G > 32) G = 32;
    if (G < 1) G = 1;
    while (G > 1 && n / G < 64) G >>= 1;
    // round state in every cloud's dist row (see the kernels): five words from int index n/2 -- zero them, stream-ordered
    if (pa_fill32_2d(dist + n / 2, (size_t)n, 0u, 5, (size_t)b, (hipStream_t)stream) != PA_OK) {      // a kernel, not a memset node (pa_common.h)
        pa_set_error("pa_emd_forward: zero fill failed");
        return PA_EINVAL;
    }
    hipError_t e = hipFuncSetAttribute((const void *)emd_round_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    if (e != hipSuccess) { pa_set_error("pa_emd_forward: cannot raise the LDS limit: %s", hipGetErrorString(e)); return (int)e; }
    int r1 = iters;
    // The workgroups per cloud may follow a schedule over the round index (A/B knob PA_EMD_SCHED="from:G,from:G,...", default none: measured, one
    // workgroup per CU from the first round on is the best of them, profiles/r04_ab_log.txt) -- a function of the round only, so results cannot depend on
    // it: the slices are recomputed from the launch's own grid; the bidder list, the counters and the arrival count are per cloud and per launch.
    struct Step { int from, g; };
    static Step sched[8];
    static int nsched = -1;
    if (nsched < 0) {
        const char *e = getenv("PA_EMD_SCHED");
        const char *txt = e ? e : "";
        int k = 0;
        while (*txt && k < 8) {
            char *end = nullptr;
            const long f = strtol(txt, &end, 10);
            if (end == txt || *end != ':') break;
            txt = end + 1;
            const long gg = strtol(txt, &end, 10);
            if (end == txt) break;
            sched[k].from = (int)f; sched[k].g = (int)gg; ++k;
            txt = (*end == ',') ? end + 1 : end;
        }
        nsched = k;
    }
    for (int it = 0; it < r1; ++it) {
        int Gi = G;
        for (int k = 0; k < nsched; ++k)
            if (it >= sched[k].from && sched[k].g > 0 && sched[k].g < G) Gi = sched[k].g;
        hipLaunchKernelGGL(emd_round_kernel, dim3(Gi, b), dim3(EMD_THREADS), lds_bytes, (hipStream_t)stream, n, xyz1, xyz2, dist, assignment, price,
                           assignment_inv, bid, bid_increments, max_increments, max_idx, eps, it == iters - 1 ? 1 : 0);
    }
    PA_CHECK_LAUNCH("pa_emd_forward");
    return PA_OK;
}

PA_API int pa_emd_backward(int b, int n, const float *xyz1, const float *xyz2, float *grad_xyz, const float *grad_dist, const int *idx,
                           pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && xyz1 && xyz2 && grad_xyz && grad_dist && idx, "pa_emd_backward: bad arguments");
    const long total = (long)b * n;
    hipLaunchKernelGGL(emd_backward_kernel, dim3(pa_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, total, n, xyz1, xyz2, grad_dist, idx,
                       grad_xyz);
    PA_CHECK_LAUNCH("pa_emd_backward");
    return PA_OK;
}
