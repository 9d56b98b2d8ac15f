// K4, fourth form: FOUR lanes per query over an 8 x 8 x 8 cell grid (the set-abstraction kNN at 2048-4096 source points).
//
// Reference semantics (knnquery_cuda_kernel.cu:6-50): per query the first nsample entries of the stable ascending
// sort of (d2, index), d2 = (qx-x)*(qx-x) + (qy-y)*(qy-y) + (qz-z)*(qz-z) in fp32 without contraction; unfilled slots (0, +inf).
//
// What round 2 measured: the wave-per-query grid kernel (knn.hip) costs ~2000 wave instructions per query (0.149 ms per batch); one LANE per
// query (knn_lane.hip) costs ~200 but leaves 512 waves with 600 k-cycle chains for 1024 SIMDs (0.287 ms); the same lane-per-query walk for
// the 3-NN (three_nn_grid.hip: 2048 waves, short chains) took 0.030 ms instead of 0.085.  This kernel gives the kNN that shape: a QUAD of
// adjacent lanes shares a query, lane j takes every fourth candidate of every cell row, so a batch is 2048 waves again and a lane's chain
// is a quarter as long.
//   pass 1  each lane keeps the K smallest TAGGED distances of its quarter in a sorted register list: the distance's bit pattern with the low
//           12 mantissa bits replaced by the candidate's position, two integer min / max per slot and candidate; the quad's K-th smallest
//           comes from min / max merges over quad shuffles; shells are added until the top of the K-th's truncation bucket is provably
//           smaller than anything unvisited (stop rule of three_nn_grid.hip);
//   pass 2  (none: the survivors' positions are in the lists) every list entry whose truncated distance is <= the K-th's is a candidate --
//           the exact K best are among them;
//   pass 3  the quad's candidates become exact 64-bit (d2 bits, index) keys in LDS; a candidate's output slot is its rank among them
//           (exact (d2, index) order whatever the ties), slots beyond the candidates are (0, +inf).
// A lane whose whole list qualifies (it may have dropped candidates) or more than KQ_TCAP candidates in a quad (a dozen ties inside one
// 2^-12 bucket: lattice clouds) send the query down a slow exact path: lane 0 of the quad inserts every candidate of the visited cells into
// a sorted key list.
//
// Measured (MI355X, b = 32, n = 4096, m = 1024, k = 20; tools/knn_time.py), uniform / plane-like clouds, wave-per-query kernel 148 / 150 us:
//   256 threads, 128 queries per workgroup (one wave per SIMD, every LDS access at full latency)   158 / 219 us
//   512 threads, 128 queries per workgroup (two waves per SIMD)                                      97 / 172 us
//   + two candidates per trip, four keys per rank trip                                              88 / 151 us
//   + tagged distances instead of a second walk (shipped)                                           see DESIGN.md
//   1024 threads (64-register budget: spills)                                                      242 / 482 us
// i.e. -35 % on uniform clouds (the benchmark distribution), +15 % on plane-like ones (dense cells: more candidates per neighbourhood and
// more imbalance inside a wave); the cell-grid 3-NN gains on both (85 -> 30 / 45 us).
#include <stdlib.h>

#include "pa_common.h"
#include "pa_cellsort.h"

namespace {

constexpr u64 KQ_INF0 = ((u64)0x7F800000u) << 32;
constexpr u32 KQ_NONE = 0x7F800000u;     // list entry "nothing": the bit pattern of +inf (every admissible tagged distance is below it)
constexpr int KQ_TAG = 12;               // low mantissa bits of a list entry that hold the candidate's position in the sorted cloud (n <= 4096)
#ifndef KQ_TCAP_V
#define KQ_TCAP_V 32
#endif
#ifndef KQ_NT
#define KQ_NT 512
#endif
constexpr int KQ_TCAP = KQ_TCAP_V;      // keys per query
constexpr int KQ_AUX_FLOATS = KG_AUX_FLOATS + 8;

__device__ __forceinline__ u32 med3_u32(u32 a, u32 b, u32 c)
{
    u32 r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// K smallest (ascending) of two ascending K-lists held by this lane (a) and the lane `mask` away (quad shuffle): bitonic merge on 32 slots
template <int K>
__device__ __forceinline__ void merge_pair(u32 (&a)[K], int mask)
{
    static_assert(K <= 32, "padded to 32 slots");
    u32 c[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int jb = 31 - j;                                   // partner element paired with slot j
        u32 pb = KQ_NONE;
        if (jb < K) pb = (u32)__shfl_xor((int)a[jb], mask);
        c[j] = j < K ? (jb < K ? min(a[j], pb) : a[j]) : pb;      // min(A[j] or none, B[31 - j] or none)
    }
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1)
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if ((i & s) == 0) {
                const u32 lo = min(c[i], c[i + s]), hi = max(c[i], c[i + s]);
                c[i] = lo;
                c[i + s] = hi;
            }
#pragma unroll
    for (int j = 0; j < K; ++j) a[j] = c[j];
}

// KL: entries of a LANE's list.  The K nearest of a query are spread over the quad's four strided candidate streams about evenly (5 +- 2 of them per
// lane at K = 20), so a lane keeps only its KL = 0.7 K smallest: every candidate costs KL instead of K v_med3 steps.  Exactness is unchanged: the
// quad's K-th comes from the union of the four lists (>= the true K-th, so the candidate band and the stop rule only get more conservative), and a
// lane whose WHOLE list falls inside the band may have dropped a candidate -> that query takes the exact slow path (4 sigma: a few queries per
// batch).
// Per-cloud record of pa_cloud_cellsort (floats): [0, 4n) the sorted cloud as (x, y, z, index bits); then KG_CELLS + 1 cell END offsets (ints);
// then lo.xyz, scale.xyz of the cell function.  All workgroups of a cloud used to repeat this counting sort (a quarter of the kernel's time at
// 128 queries per workgroup); with PRESORT they copy the record -- 16-byte coalesced loads -- instead.
constexpr int KQ_CNT_PAD = (KG_CELLS + 1 + 3) & ~3;     // cell offsets padded so that every cloud's record stays 16-byte aligned
__host__ __device__ constexpr long kq_cells_floats(int n) { return 4L * n + KQ_CNT_PAD + 8; }

template <int PTS, int NT>
__global__ __launch_bounds__(NT) void cellsort_kernel(int n, const float *__restrict__ xyz_all, float *__restrict__ cells_all)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *sorted = reinterpret_cast<float4 *>(smem);
    float *box = smem + 4 * (size_t)n;
    int *cnt = reinterpret_cast<int *>(box + 64 * 8);
    float *red = reinterpret_cast<float *>(cnt + KG_CELLS + 1);
    float *grid = red + 16 * 6;
    const int b = blockIdx.x, tid = threadIdx.x;
    int nchunks;
    cell_sort_cloud<PTS, NT, true>(n, xyz_all + (size_t)b * n * 3, sorted, box, cnt, red, &nchunks, grid);
    float *rec = cells_all + (size_t)b * kq_cells_floats(n);
    for (int i = tid; i < n; i += NT) reinterpret_cast<float4 *>(rec)[i] = sorted[i];
    for (int c = tid; c <= KG_CELLS; c += NT) reinterpret_cast<int *>(rec + 4 * (size_t)n)[c] = cnt[c];
    if (tid < 6) rec[4 * (size_t)n + KQ_CNT_PAD + tid] = grid[tid];
}

template <int K, int KL, int PTS, int NT, bool PRESORT = false>
__global__ __launch_bounds__(NT) void knn_quad_kernel(int n, int m, int q_per_block, const float *__restrict__ xyz_all, const float *__restrict__ new_xyz_all,
                                                       int *__restrict__ idx_all, float *__restrict__ dist2_all, long long *dbg, int mq, const float *__restrict__ cells_all = nullptr)
{
    // m: queries per cloud in the buffers (the stride); mq <= m: queries this launch answers per cloud (a window: the base pointers are offset by
    // the caller -- pa_knnquery_window, the chunked first level of the engine's latency mode)
#define KQ_STAMP(i) do { if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
    constexpr int NQ = NT / 4;        // queries per pass
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *sorted = reinterpret_cast<float4 *>(smem);
    float *box = smem + 4 * (size_t)n;
    int *cnt = reinterpret_cast<int *>(box + 64 * 8);
    float *red = reinterpret_cast<float *>(cnt + KG_CELLS + 1);
    float *grid = red + 16 * 6;
    int *qcnt = reinterpret_cast<int *>(grid + 8);
    u64 *keys = reinterpret_cast<u64 *>((reinterpret_cast<uintptr_t>(qcnt + KG_CELLS + 1) + 7) & ~(uintptr_t)7);   // [NQ][KQ_TCAP + 4], 8-byte aligned
    unsigned short *qorder = reinterpret_cast<unsigned short *>(keys + NQ * (KQ_TCAP + 4));   // [q_per_block]
    const int b = blockIdx.y, tid = threadIdx.x, part = tid & 3, ql = tid >> 2;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    int nchunks;
    KQ_STAMP(0);
    if (PRESORT) {
        const float *rec = cells_all + (size_t)b * kq_cells_floats(n);
        for (int i = tid; i < n; i += NT) sorted[i] = reinterpret_cast<const float4 *>(rec)[i];
        for (int c = tid; c <= KG_CELLS; c += NT) cnt[c] = reinterpret_cast<const int *>(rec + 4 * (size_t)n)[c];
        if (tid < 6) grid[tid] = rec[4 * (size_t)n + KQ_CNT_PAD + tid];
        __syncthreads();
        (void)nchunks; (void)xyz;
    } else {
        cell_sort_cloud<PTS, NT, true>(n, xyz, sorted, box, cnt, red, &nchunks, grid);
    }
    KQ_STAMP(1);
    const float lo0 = grid[0], lo1 = grid[1], lo2 = grid[2], sc0 = grid[3], sc1 = grid[4], sc2 = grid[5];

    // this workgroup's queries in cell order
    const int q_begin = blockIdx.x * q_per_block, q_end = min(q_begin + q_per_block, mq);
    {
        for (int c = tid; c <= KG_CELLS; c += NT) qcnt[c] = 0;
        __syncthreads();
        auto qcell = [&](int qi) {
            const float *qp = new_xyz_all + ((size_t)b * m + qi) * 3;
            const int cx = min(max((int)((qp[0] - lo0) * sc0), 0), 7), cy = min(max((int)((qp[1] - lo1) * sc1), 0), 7), cz = min(max((int)((qp[2] - lo2) * sc2), 0), 7);
            return (cz * 8 + cy) * 8 + cx;
        };
        for (int qi = q_begin + tid; qi < q_end; qi += NT) atomicAdd(&qcnt[qcell(qi)], 1);
        __syncthreads();
        if (tid < 64) {
            int v[9], sum = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) { const int c = tid * 9 + t; v[t] = c <= KG_CELLS ? qcnt[c] : 0; sum += v[t]; }
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (tid >= o) incl += u; }
            int run = incl - sum;
#pragma unroll
            for (int t = 0; t < 9; ++t) { const int c = tid * 9 + t; if (c <= KG_CELLS) qcnt[c] = run; run += v[t]; }
        }
        __syncthreads();
        for (int qi = q_begin + tid; qi < q_end; qi += NT) qorder[atomicAdd(&qcnt[qcell(qi)], 1)] = (unsigned short)(qi - q_begin);
        __syncthreads();
    }

    KQ_STAMP(2);
    for (int qs = ql; qs < q_end - q_begin; qs += NQ) {          // the four lanes of a quad run this loop together
        const int q = q_begin + qorder[qs];
        const float *qp = new_xyz_all + ((size_t)b * m + q) * 3;
        const float qx = qp[0], qy = qp[1], qz = qp[2];
        const float f0 = (qx - lo0) * sc0, f1 = (qy - lo1) * sc1, f2 = (qz - lo2) * sc2;
        const bool tame = fabsf(f0) < 1000.f && fabsf(f1) < 1000.f && fabsf(f2) < 1000.f;   // false for NaN / inf
        const int cx = min(max((int)f0, 0), 7), cy = min(max((int)f1, 0), 7), cz = min(max((int)f2, 0), 7);

        // every `step`-th point (from `first`) of the cells at Chebyshev distance r_from .. r_to from the query's cell
        auto scan = [&](int r_from, int r_to, int first, int step, auto &&fn) {
            // one contiguous range of the sorted cloud (the cells c0 .. c1 of a row), two candidates per trip with the NEXT trip's two LDS reads issued
            // before this trip's candidates are processed (a trip is ~55 VALU instructions behind a ~130-cycle read; with two waves per SIMD the
            // read of a trip was exposed about half of the time).  The look-ahead reads up to 3 steps + 1 past `end`: other cells' points or the
            // first bytes behind the array (the box / counter area of the same LDS allocation) -- never used.
            auto seg = [&](int c0, int c1) {
                const int beg = c0 ? cnt[c0 - 1] : 0, end = cnt[c1];
                int pos = beg + first;
                if (pos >= end) return;
                float4 pa = sorted[pos], pb = sorted[pos + step];
                while (pos + step < end) {
                    const float4 na = sorted[pos + 2 * step], nb = sorted[pos + 3 * step];
                    fn(pos, pa);
                    fn(pos + step, pb);
                    pa = na; pb = nb;
                    pos += 2 * step;
                }
                if (pos < end) fn(pos, pa);
            };
            for (int dz = -r_to; dz <= r_to; ++dz) {
                const int z = cz + dz;
                if (z < 0 || z > 7) continue;
                for (int dy = -r_to; dy <= r_to; ++dy) {
                    const int y = cy + dy;
                    if (y < 0 || y > 7) continue;
                    const int rowbase = (z * 8 + y) * 8;
                    const int xl = max(cx - r_to, 0), xh = min(cx + r_to, 7);
                    const bool interior = r_from > 0 && abs(dz) < r_from && abs(dy) < r_from;   // the middle of the row was scanned before
                    {
                        const int c0 = rowbase + xl, c1 = rowbase + (interior ? min(cx - r_from, xh) : xh);
                        if (c1 >= c0) seg(c0, c1);
                    }
                    if (interior) {
                        const int c0 = rowbase + max(cx + r_from, xl), c1 = rowbase + xh;
                        if (c1 >= c0) seg(c0, c1);
                    }
                }
            }
        };
        // (x, y) as one packed subtract / multiply on the register pair the LDS read delivers, z scalar: the same IEEE operations in the same order
        // (dx*dx + dy*dy) + dz*dz; left to itself the compiler pairs components of TWO candidates and spends nine v_mov per trip assembling them
        typedef float kq_f2 __attribute__((ext_vector_type(2)));
        // the subtraction as a plain packed add of the negated query, p + (-q) = -(q - p) exactly and the square drops the sign: no operand modifiers on
        // packed fp32 (pa_common.h, pa_pk_plain)
        const kq_f2 nqxy = pa_pk_plain((kq_f2){-qx, -qy});
        auto dist = [&](const float4 &p) {                                                                                       // :31
            kq_f2 d = (kq_f2){p.x, p.y} + nqxy;
            d = d * d;
            const float dz = qz - p.z;
            return (d.x + d.y) + dz * dz;
        };
        auto outside_bound = [&](int R) {
            float best = INFINITY;
            auto face = [&](float f, int c, float sc) {
                if (!(sc > 0.f)) return;
                const float inv = 0.999f / sc;
                if (c + R + 1 <= 7) { const float g = fmaxf((float)(c + R + 1) - f - 1e-3f, 0.f) * inv; best = fminf(best, g * g); }
                if (c - R - 1 >= 0) { const float g = fmaxf(f - (float)(c - R) - 1e-3f, 0.f) * inv; best = fminf(best, g * g); }
            };
            face(f0, cx, sc0); face(f1, cy, sc1); face(f2, cz, sc2);
            return best;
        };

        // ---- pass 1: this lane's K smallest TAGGED distances.  A list entry is the distance's bit pattern with its low KQ_TAG bits replaced
        // by the candidate's position in the sorted cloud: d >= 0, so unsigned order on the patterns is the order of the truncated distances
        // (ties broken by position), the lists stay plain 32-bit integers (v_med3_u32 / v_min_u32 / v_max_u32: no float modes involved), and the
        // positions of the survivors come back out of the list -- no second walk over the cells.
        u32 L[KL];
#pragma unroll
        for (int j = 0; j < KL; ++j) L[j] = KQ_NONE;
        auto pass1 = [&](int pos, const float4 &p) {
            const u32 bits = __float_as_uint(dist(p));
            const u32 d = bits < KQ_NONE ? ((bits & ~((1u << KQ_TAG) - 1u)) | (u32)pos) : KQ_NONE;     // +inf / NaN (either sign): never admitted
#pragma unroll
            for (int j = KL - 1; j >= 1; --j) L[j] = med3_u32(L[j - 1], d, L[j]);
            L[0] = min(L[0], d);
        };
        auto quad_kth = [&]() {                        // K-th smallest entry of the four lanes' lists (same value in all four lanes)
            u32 M[K];
#pragma unroll
            for (int j = 0; j < K; ++j) M[j] = j < KL ? L[j < KL ? j : 0] : KQ_NONE;
            merge_pair<K>(M, 1);                        // lanes {0,1} and {2,3}: K smallest of each pair, ascending
            u32 kth = 0u;
#pragma unroll
            for (int j = 0; j < K; ++j) kth = max(kth, min(M[j], (u32)__shfl_xor((int)M[K - 1 - j], 2)));   // the K smallest of the union are min(M[j], P[K-1-j])
            return kth;
        };
        int R = 1;
        scan(0, 1, part, 4, pass1);
        u32 kth;
        while (true) {
            kth = quad_kth();                          // < KQ_NONE <=> the quad has seen K admissible candidates
            // every distance whose truncation equals the K-th's is still a candidate: compare the stop bound with the top of that bucket
            const float kth_hi = __uint_as_float(kth | ((1u << KQ_TAG) - 1u));
            if (R >= 7 || (tame && kth < KQ_NONE && kth_hi < outside_bound(R))) break;
            ++R;
            scan(R, R, part, 4, pass1);
        }
        KQ_STAMP(3);
        if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) dbg[8] = R;
        // ---- candidates: the exact K best all have a truncated distance <= the K-th's (at least K entries are <= the K-th entry, and anything
        // with a larger truncated distance is farther than all of them).  A lane's candidates are a prefix of its sorted list; a lane whose
        // whole list qualifies may have dropped some -> slow path.
        const u32 tmax = kth >> KQ_TAG;
        int qn = 0;
#pragma unroll
        for (int j = 0; j < KL; ++j) qn += (L[j] < KQ_NONE && (L[j] >> KQ_TAG) <= tmax) ? 1 : 0;
        KQ_STAMP(4);
        const int base = (tid & 63) & ~3;
        const int c0 = __shfl(qn, base), c1 = __shfl(qn, base + 1), c2 = __shfl(qn, base + 2), c3 = __shfl(qn, base + 3);
        const int total = c0 + c1 + c2 + c3;
        const bool overflow = c0 >= KL || c1 >= KL || c2 >= KL || c3 >= KL || total > KQ_TCAP;
        const size_t o = ((size_t)b * m + q) * K;
        if (!overflow) {
            // ---- pass 3: exact keys to LDS, output slot = rank among the quad's keys
            const int off = (part > 0 ? c0 : 0) + (part > 1 ? c1 : 0) + (part > 2 ? c2 : 0);
            u64 *kq = keys + ql * (KQ_TCAP + 4);
            // Every list entry's point is fetched in ONE batch of LDS reads (an entry that is no candidate reads position 0 or its own point: harmless)
            // and its exact key built; the stores are predicated.  (One entry at a time behind a wave vote, each LDS read waited for, this block and
            // the rank loop below were 18.5 k of the workgroup's 90 k cycles.)
            u64 mine[KL];
            {
                float4 pj[KL];
#pragma unroll
                for (int j = 0; j < KL; ++j) pj[j] = sorted[L[j] & ((1u << KQ_TAG) - 1u)];
#pragma unroll
                for (int j = 0; j < KL; ++j) {
                    mine[j] = pa_make_key(dist(pj[j]), (u32)__float_as_int(pj[j].w));
                    if (j < qn) kq[off + j] = mine[j];
                }
            }
            if (part == 0) { kq[total] = ~0ull; kq[total + 1] = ~0ull; kq[total + 2] = ~0ull; }      // the rank loop reads four keys per trip
            // the four lanes of a quad read each other's keys: same wavefront, LDS operations retire in order; the fence keeps the compiler
            // from moving the reads above the writes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ranks: every block of four keys is read once (the next block is in flight meanwhile) and compared with ALL of this lane's candidates --
            // the loop over a lane's candidates with an LDS round trip per block inside it paid that latency qn times over
            int rank[KL];
#pragma unroll
            for (int j = 0; j < KL; ++j) rank[j] = 0;
            const int qmax = max(max(c0, c1), max(c2, c3));
            u64 a0 = kq[0], a1 = kq[1], a2 = kq[2], a3 = kq[3];
            for (int t = 0; t < total; t += 4) {
                const u64 n0 = kq[t + 4], n1 = kq[t + 5], n2 = kq[t + 6], n3 = kq[t + 7];       // past the quad's keys: the next quad's (or the query order behind the last): never compared
#pragma unroll
                for (int j = 0; j < KL; ++j) {
                    if (!__any(j < qmax)) break;
                    rank[j] += (a0 < mine[j] ? 1 : 0) + (a1 < mine[j] ? 1 : 0) + (a2 < mine[j] ? 1 : 0) + (a3 < mine[j] ? 1 : 0);
                }
                a0 = n0; a1 = n1; a2 = n2; a3 = n3;
            }
#pragma unroll
            for (int j = 0; j < KL; ++j) {
                if (j < qn && rank[j] < K) {
                    idx_all[o + rank[j]] = (int)(u32)mine[j];
                    dist2_all[o + rank[j]] = __uint_as_float((u32)(mine[j] >> 32));
                }
            }
            for (int j = total + part; j < K; j += 4) {          // fewer than K admissible points: (0, +inf) like the reference
                idx_all[o + j] = 0;
                dist2_all[o + j] = __uint_as_float(0x7F800000u);
            }
        } else if (part == 0) {
            // ---- slow exact path: every candidate of the visited cells into a sorted key list
            u64 Lk[K];
#pragma unroll
            for (int j = 0; j < K; ++j) Lk[j] = KQ_INF0;
            auto slow = [&](int, const float4 &p) {
                u64 key = pa_make_key(dist(p), (u32)__float_as_int(p.w));
                if (key < Lk[K - 1]) {                 // key >= KQ_INF0 (inf / NaN distance) never passes
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const bool lt = key < Lk[j];
                        const u64 lo = lt ? key : Lk[j], hi = lt ? Lk[j] : key;
                        Lk[j] = lo;
                        key = hi;
                    }
                }
            };
            scan(0, R, 0, 1, slow);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                idx_all[o + j] = (int)(u32)Lk[j];
                dist2_all[o + j] = __uint_as_float((u32)(Lk[j] >> 32));
            }
        }
        KQ_STAMP(5);
        if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { dbg[9] = total; dbg[10] = overflow; }
    }
#undef KQ_STAMP
}

#ifndef KQ_QPB
#define KQ_QPB 128
#endif

template <int K>
int launch_quad(int b, int n, int m, const float *xyz, const float *new_xyz, int *idx, float *dist2, hipStream_t st, long long *dbg, int mq, const float *cells)
{
    constexpr int NT = KQ_NT, PTS = 4096 / KQ_NT;
    const int qpb = KQ_QPB;
    const size_t lds = (size_t)n * 16 + (size_t)KQ_AUX_FLOATS * 4 + (size_t)(KG_CELLS + 3) * 4 + (size_t)(NT / 4) * (KQ_TCAP + 4) * 8 + (size_t)qpb * 2;
#ifndef KQ_KL_NUM
#define KQ_KL_NUM 7      // lane list length = ceil(K * KQ_KL_NUM / 10); 10 = the full K
#endif
    constexpr int KL = (K * KQ_KL_NUM + 9) / 10;
    if (cells) {
        auto kern = knn_quad_kernel<K, KL, PTS, NT, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(pa_div_up(mq, qpb), b), dim3(NT), lds, st, n, m, qpb, xyz, new_xyz, idx, dist2, dbg, mq, cells);
        return 0;
    }
    auto kern = knn_quad_kernel<K, KL, PTS, NT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(pa_div_up(mq, qpb), b), dim3(NT), lds, st, n, m, qpb, xyz, new_xyz, idx, dist2, dbg, mq, (const float *)nullptr);
    return 0;
}

}  // namespace

static int g_quad_on = -1;
// A/B and test switch: 1 = the quad kernel where it applies (default; PA_KNN_NO_QUAD=1 turns it off), 0 = the wave-per-query kernels
PA_API void pa_knn_quad_enable(int on) { g_quad_on = on ? 1 : 0; }

// 1 when the quad kernel took the call, 0 when it is off or the shape is not one it is built for.
int pa_knn_quad_try(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx, float *dist2, hipStream_t st, long long *dbg, int mq,
                    const float *cells)
{
    if (mq <= 0) mq = m;
    if (g_quad_on < 0) g_quad_on = getenv("PA_KNN_NO_QUAD") != nullptr ? 0 : 1;
    // From 1024 source points and 128 queries (round 4; 2048 / 256 before): at the second level of the 4096-point models (1024 -> 128 centres) the
    // wave-per-query kernel fills the chip with 1024 workgroups for 26 us, this one runs ONE workgroup per cloud for 47 us -- longer on the step's own
    // stream, but a batch then holds 32 CUs instead of all of them while the other streams' dense kernels run: 37.2 -> 37.5 k submaps/s.
    static const int nmin = getenv("PA_KNN_QUAD_NMIN") ? atoi(getenv("PA_KNN_QUAD_NMIN")) : 1024, mmin = getenv("PA_KNN_QUAD_MMIN") ? atoi(getenv("PA_KNN_QUAD_MMIN")) : 128;   // tuning knobs
    if (!g_quad_on || n < nmin || n > 4096 || m < mmin) return 0;      // a function of the level's shape (n, m), never of the window
    switch (nsample) {
        case 16: launch_quad<16>(b, n, m, xyz, new_xyz, idx, dist2, st, dbg, mq, cells); return 1;
        case 20: launch_quad<20>(b, n, m, xyz, new_xyz, idx, dist2, st, dbg, mq, cells); return 1;
        case 32: launch_quad<32>(b, n, m, xyz, new_xyz, idx, dist2, st, dbg, mq, cells); return 1;
        default: return 0;
    }
}

#ifdef KG_STAMPS
PA_API int pa_kg_stamps_read(long long *host16) { return hipMemcpyFromSymbol(host16, HIP_SYMBOL(kg_stamps), 16 * sizeof(long long)) == hipSuccess ? 0 : -1; }
#endif

// ---- the cloud's counting sort as a launch of its own (one workgroup per cloud), for callers that know the cloud before they know the queries:
// the engine sorts the input cloud while the first level's sampling chain has not even started; the first level's neighbour search then copies
// the record instead of sorting (pa_knnquery_presorted).  cells: pa_cloud_cellsort_floats(b, n) floats.
PA_API long pa_cloud_cellsort_floats(int b, int n) { return (long)b * kq_cells_floats(n); }

PA_API int pa_cloud_cellsort(int b, int n, const float *xyz, float *cells, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && n <= 4096 && xyz && cells && ((uintptr_t)cells & 15) == 0, "pa_cloud_cellsort: needs n <= 4096 and a 16-byte aligned record buffer (n=%d)", n);
    constexpr int NT = KQ_NT, PTS = 4096 / KQ_NT;
    const size_t lds = (size_t)n * 16 + (size_t)KQ_AUX_FLOATS * 4;
    auto kern = cellsort_kernel<PTS, NT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(b), dim3(NT), lds, (hipStream_t)stream, n, xyz, cells);
    PA_CHECK_LAUNCH("pa_cloud_cellsort");
    return PA_OK;
}

// pa_knnquery with the source cloud's record from pa_cloud_cellsort (same results bit for bit); PA_EUNSUPPORTED when the level's shape does not run
// the cell-grid kernel (2048..4096 source points, >= 256 queries, nsample 16 / 20 / 32): use pa_knnquery then.
PA_API int pa_knnquery_presorted(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, const float *cells, int *idx, float *dist2, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && m > 0 && nsample > 0 && xyz && new_xyz && cells && idx && dist2 && b <= 65535, "pa_knnquery_presorted: bad arguments");
    PA_REQUIRE(((uintptr_t)cells & 15) == 0, "pa_knnquery_presorted: the record buffer must be 16-byte aligned");
    if (!pa_knn_quad_try(b, n, m, nsample, xyz, new_xyz, idx, dist2, (hipStream_t)stream, nullptr, 0, cells)) {
        pa_set_error("pa_knnquery_presorted: the level (n=%d, m=%d, nsample=%d) does not run the cell-grid kernel", n, m, nsample);
        return PA_EUNSUPPORTED;
    }
    PA_CHECK_LAUNCH("pa_knnquery_presorted");
    return PA_OK;
}
