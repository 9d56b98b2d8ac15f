// K4, third form: one LANE per query (the set-abstraction kNN at 2048-4096 source points, 20 neighbours).
//
// Reference semantics (knnquery_cuda_kernel.cu:6-50): per query the first nsample entries of the stable ascending
// sort of (d2, index), d2 = (qx-x)*(qx-x) + (qy-y)*(qy-y) + (qz-z)*(qz-z) in fp32 without contraction; unfilled slots (0, +inf).
//
// Why a third kernel.  knn_grid_kernel gives a query a whole wavefront: 64 candidates are evaluated per step, but choosing the next
// chunk, the ballot and every list insertion are wave-wide serial steps -- about 2000 instructions PER QUERY, 0.148 ms for the 32 768
// queries of a batch (the largest non-MFMA item of a step after FPS).  Here the cloud is sorted into an 8 x 8 x 8 grid in LDS (row-major cell
// order: the three cells cx-1..cx+1 of a row are one contiguous range) and every lane walks the 3 x 3 x 3 neighbourhood of ITS query's cell:
//   pass 1  keeps the K smallest DISTANCES (no indices) in a sorted register array: inserting is one v_med3_f32 per slot, unconditional,
//           so there is no divergent branch; a candidate not farther than the CURRENT K-th distance has its position queued in LDS;
//           shells are added until the K-th distance is provably smaller than anything outside;
//   pass 2  only if the queue overflowed: walks the same cells again and queues the candidates with d <= the FINAL K-th distance;
//   pass 3  inserts the queued candidates as 64-bit (d, index) keys into a sorted register list (exact tie order) and writes it out.
// About 150 instructions per query.  A lane whose queue overflows even in pass 2 (more than QCAP candidates tie with the K-th distance:
// lattice clouds, duplicates) redoes its scan with direct key insertion -- slow and exact.
//
// Exactness of the stop rule.  Cells come from c_a(p) = clamp((int)f_a(p), 0, 7), f_a(p) = fl(fl(p_a - lo_a) * scale_a), monotone in p_a.
// After all cells within Chebyshev distance R of the query's cell are scanned, an unscanned point p differs by >= R + 1 cells on some
// axis a with scale_a > 0, hence |f_a(p) - f_a(q)| >= R when f_a(q) lies inside [0, 8] (cell index <= f < cell index + 1 up to the clamp
// at the upper face, which only makes the inequality non-strict).  f carries a relative rounding error of 2 ulp on values <= 8, so in real
// numbers |p_a - q_a| >= (R - 4e-6) / scale_a, and the fp32 distance of p is at least that squared times (1 - 6 ulp).  The kernel stops
// when kth < ((R - 1e-3) * 0.999 * min_a 1/scale_a)^2 -- far inside the bound -- with strict <, so equal-distance points are never cut
// off.  A query outside the cloud's bounding box (possible through the operator API, never in the model) simply never stops early.
#include <stdlib.h>

#include "pa_common.h"
#include "pa_cellsort.h"

namespace {

constexpr u64 KL_INF0 = ((u64)0x7F800000u) << 32;
constexpr int KL_QCAP = 96;

// LDS floats behind the sorted points: the cell-sort aux block + grid (8) ; the queue (u16 [QCAP][NT]) follows
constexpr int KL_AUX_FLOATS = KG_AUX_FLOATS + 8;

template <int K, int PTS, int NT>
__global__ __launch_bounds__(NT) void knn_lane_kernel(int n, int m, int q_per_block, const float *__restrict__ xyz_all, const float *__restrict__ new_xyz_all,
                                                       int *__restrict__ idx_all, float *__restrict__ dist2_all, long long *dbg)
{
#define KL_STAMP(i) do { if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *sorted = reinterpret_cast<float4 *>(smem);
    float *box = smem + 4 * (size_t)n;
    int *cnt = reinterpret_cast<int *>(box + 64 * 8);
    float *red = reinterpret_cast<float *>(cnt + KG_CELLS + 1);
    float *grid = red + 16 * 6;
    int *qcnt = reinterpret_cast<int *>(grid + 8);                                    // [KG_CELLS + 1] query histogram -> offsets
    unsigned short *qorder = reinterpret_cast<unsigned short *>(qcnt + KG_CELLS + 1);  // [q_per_block] this workgroup's queries in cell order
    unsigned short *queue = qorder + q_per_block;
#ifdef KL_SOA
    float *soa = reinterpret_cast<float *>(queue + KL_QCAP * NT);      // [4][n]: x, y, z, index bits
#endif
    const int b = blockIdx.y, tid = threadIdx.x;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    int nchunks;
    KL_STAMP(0);
    cell_sort_cloud<PTS, NT, true>(n, xyz, sorted, box, cnt, red, &nchunks, grid);
#ifdef KL_SOA
    for (int i = tid; i < n; i += NT) { const float4 p = sorted[i]; soa[i] = p.x; soa[n + i] = p.y; soa[2 * n + i] = p.z; soa[3 * n + i] = p.w; }
    __syncthreads();
#define KL_LOAD(pos) make_float4(soa[pos], soa[n + (pos)], soa[2 * n + (pos)], soa[3 * n + (pos)])
#else
#define KL_LOAD(pos) sorted[pos]
#endif
    KL_STAMP(1);
    const float lo0 = grid[0], lo1 = grid[1], lo2 = grid[2], sc0 = grid[3], sc1 = grid[4], sc2 = grid[5];

    // ---- this workgroup's queries in cell order: the 64 lanes of a wave then walk neighbouring rows -- similar trip counts, LDS reads that
    // hit nearby addresses.  A counting sort over the same 512 cells; the order inside a cell is whatever the atomics give (every query of
    // the slice is processed exactly once either way).
    const int q_begin = blockIdx.x * q_per_block, q_end = min(q_begin + q_per_block, m);
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

    for (int qs = tid; qs < q_end - q_begin; qs += NT) {
        const int q = q_begin + qorder[qs];
        const float *qp = new_xyz_all + ((size_t)b * m + q) * 3;
        const float qx = qp[0], qy = qp[1], qz = qp[2];
        const float f0 = (qx - lo0) * sc0, f1 = (qy - lo1) * sc1, f2 = (qz - lo2) * sc2;
        const bool inside = f0 >= 0.f && f0 <= 8.f && f1 >= 0.f && f1 <= 8.f && f2 >= 0.f && f2 <= 8.f;   // false for NaN
        const int cx = min(max((int)f0, 0), 7), cy = min(max((int)f1, 0), 7), cz = min(max((int)f2, 0), 7);

        // visit every point of the cells at Chebyshev distance r_from .. r_to from (cx, cy, cz): fn(position in `sorted`, point)
        auto scan = [&](int r_from, int r_to, auto &&fn) {
            for (int dz = -r_to; dz <= r_to; ++dz) {
                const int z = cz + dz;
                if (z < 0 || z > 7) continue;
                for (int dy = -r_to; dy <= r_to; ++dy) {
                    const int y = cy + dy;
                    if (y < 0 || y > 7) continue;
                    const int rowbase = (z * 8 + y) * 8;
                    const int xl = max(cx - r_to, 0), xh = min(cx + r_to, 7);
                    const bool interior = r_from > 0 && abs(dz) < r_from && abs(dy) < r_from;   // the middle of the row was scanned before
                    // ranges: [xl, xa) and (xb, xh] when interior, else [xl, xh]
                    const int xa = interior ? cx - r_from + 1 : xh + 1;      // first range covers cells xl .. xa - 1
                    const int xb = interior ? cx + r_from - 1 : xh;          // second range covers cells xb + 1 .. xh
                    {
                        const int c0 = rowbase + xl, c1 = rowbase + min(xa - 1, xh);
                        if (c1 >= c0) {
                            const int beg = c0 ? cnt[c0 - 1] : 0, end = cnt[c1];
#ifndef KL_NO_UNROLL2
                            int pos = beg;
                            for (; pos + 1 < end; pos += 2) { const float4 pa = KL_LOAD(pos), pb = KL_LOAD(pos + 1); fn(pos, pa); fn(pos + 1, pb); }
                            if (pos < end) fn(pos, KL_LOAD(pos));
#else
                            for (int pos = beg; pos < end; ++pos) fn(pos, KL_LOAD(pos));
#endif
                        }
                    }
                    if (interior) {
                        const int c0 = rowbase + max(xb + 1, xl), c1 = rowbase + xh;
                        if (c1 >= c0) {
                            const int beg = c0 ? cnt[c0 - 1] : 0, end = cnt[c1];
#ifndef KL_NO_UNROLL2
                            int pos = beg;
                            for (; pos + 1 < end; pos += 2) { const float4 pa = KL_LOAD(pos), pb = KL_LOAD(pos + 1); fn(pos, pa); fn(pos + 1, pb); }
                            if (pos < end) fn(pos, KL_LOAD(pos));
#else
                            for (int pos = beg; pos < end; ++pos) fn(pos, KL_LOAD(pos));
#endif
                        }
                    }
                }
            }
        };
        auto dist = [&](const float4 &p) { return (qx - p.x) * (qx - p.x) + (qy - p.y) * (qy - p.y) + (qz - p.z) * (qz - p.z); };   // :31

        // ---- pass 1: the K smallest distances
        float L[K];
#pragma unroll
        for (int j = 0; j < K; ++j) L[j] = INFINITY;
        int nfin = 0, iters = 0, qn = 0;
        auto pass1 = [&](int pos, const float4 &p) {
            ++iters;
            float d = dist(p);
            d = d < INFINITY ? d : INFINITY;          // NaN / inf: never admitted
            nfin += d < INFINITY ? 1 : 0;
            // a candidate farther than the CURRENT K-th distance can never be among the final K (the K-th only shrinks): everything else
            // is remembered by position (about K (1 + ln(T / K)) of T candidates)
            if (d <= L[K - 1] && d < INFINITY) {
                if (qn < KL_QCAP) queue[qn * NT + tid] = (unsigned short)pos;
                ++qn;
            }
#pragma unroll
            for (int j = K - 1; j >= 1; --j) L[j] = __builtin_amdgcn_fmed3f(L[j - 1], d, L[j]);
            L[0] = fminf(L[0], d);
        };
        // smallest distance an unscanned point can have after the cells within Chebyshev distance R are done: it lies beyond one of the six
        // faces of that block; a face outside the grid has nothing behind it.  In cell units the gap to the +a face is (c_a + R + 1) - f_a,
        // to the -a face f_a - (c_a - R) (see the header for why 1e-3 / 0.999 are far inside the rounding slack).
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
        int R = 1;
        scan(0, 1, pass1);
        while (R < 7) {
#ifdef KL_OLD_BOUND
            { float hm = INFINITY; if (sc0 > 0.f) hm = fminf(hm, 1.f / sc0); if (sc1 > 0.f) hm = fminf(hm, 1.f / sc1); if (sc2 > 0.f) hm = fminf(hm, 1.f / sc2);
              const float rb = ((float)R - 1e-3f) * 0.999f * hm; if (inside && nfin >= K && L[K - 1] < rb * rb) break; }
#else
            if (inside && nfin >= K && L[K - 1] < outside_bound(R)) break;
#endif
            ++R;
            scan(R, R, pass1);
        }
        const float kth = L[K - 1];
        KL_STAMP(2);
        if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid < 64) {
            int mx = iters, sm = iters;
            for (int o = 1; o < 64; o <<= 1) { mx = max(mx, __shfl_xor(mx, o)); sm += __shfl_xor(sm, o); }
            if (tid == 0) { dbg[6] = R; dbg[8] = mx; dbg[9] = sm; }
        }

        // ---- pass 2 (only when the running queue overflowed): positions of the candidates with d <= the final K-th distance
        if (qn > KL_QCAP) {
            qn = 0;
            auto pass2 = [&](int pos, const float4 &p) {
                const float d = dist(p);
                if (d <= kth && d < INFINITY) {
                    if (qn < KL_QCAP) queue[qn * NT + tid] = (unsigned short)pos;
                    ++qn;
                }
            };
            scan(0, R, pass2);
        }
        KL_STAMP(3);
        if (dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) dbg[7] = qn;

        // ---- pass 3: exact (d, index) order
        u64 Lk[K];
#pragma unroll
        for (int j = 0; j < K; ++j) Lk[j] = KL_INF0;
        auto insert = [&](u64 key) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const bool lt = key < Lk[j];
                const u64 lo = lt ? key : Lk[j], hi = lt ? Lk[j] : key;
                Lk[j] = lo;
                key = hi;
            }
        };
        if (qn <= KL_QCAP) {
            for (int e = 0; e < qn; ++e) {
                const float4 p = KL_LOAD(queue[e * NT + tid]);
                const u64 key = pa_make_key(dist(p), (u32)__float_as_int(p.w));
                if (key < Lk[K - 1]) insert(key);
            }
        } else {          // more ties with the K-th distance than the queue holds: direct insertion over the scanned cells
            auto slow = [&](int, const float4 &p) {
                const u64 key = pa_make_key(dist(p), (u32)__float_as_int(p.w));
                if (key < Lk[K - 1]) insert(key);     // key >= KL_INF0 (inf / NaN distance) never passes
            };
            scan(0, R, slow);
        }
        KL_STAMP(4);
        const size_t o = ((size_t)b * m + q) * K;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            idx_all[o + j] = (int)(u32)Lk[j];
            dist2_all[o + j] = __uint_as_float((u32)(Lk[j] >> 32));
        }
        KL_STAMP(5);
    }
#undef KL_STAMP
}

template <int K>
int launch_lane(int b, int n, int m, const float *xyz, const float *new_xyz, int *idx, float *dist2, hipStream_t st, long long *dbg)
{
    constexpr int NT = 256, PTS = 16;
    const int qpb = 256;
    size_t lds = (size_t)n * 16 + (size_t)KL_AUX_FLOATS * 4 + (size_t)(KG_CELLS + 1) * 4 + (size_t)qpb * 2 + (size_t)KL_QCAP * NT * 2;
#ifdef KL_SOA
    lds += (size_t)n * 16;
#endif
    auto kern = knn_lane_kernel<K, PTS, NT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(pa_div_up(m, qpb), b), dim3(NT), lds, st, n, m, qpb, xyz, new_xyz, idx, dist2, dbg);
    return 0;
}

}  // namespace

// MEASURED AND NOT SHIPPED AS THE DEFAULT (round 2, MI355X, b = 32, n = 4096, m = 1024, k = 20): 287 us against 149 us for the wave-per-query
// grid kernel (467 us vs 151 us on plane-like clouds).  The instruction count per query did drop ~8x, but 32 768 queries are only 512
// wavefronts for 1024 SIMDs: every wave runs a ~600 k-cycle dependent chain (LDS read -> distance -> med3 chain) alone on its SIMD, with
// nothing to hide the LDS latency behind, and FPS centres near the cloud boundary need a second shell that the whole wave waits for.
// Sorting the workgroup's queries by cell (-20 %), the face-wise stop rule (-25 % candidates), 2x unrolling (-12 %), an SoA point layout
// (no change) and a single pass with a running-threshold queue (no change) were tried.  It stays as an exact, tested alternative behind
// pa_knn_lane_enable(1) / PA_KNN_LANE=1: with >= 4x more queries per launch (8 waves per SIMD) it is the cheaper formulation.
static int g_lane_on = -1;
PA_API void pa_knn_lane_enable(int on) { g_lane_on = on ? 1 : 0; }

// Returns 1 when the per-lane kernel took the call, 0 when it is off or the shape is not one it is built for (the caller falls through).
int pa_knn_lane_try(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx, float *dist2, hipStream_t st, long long *dbg)
{
    if (g_lane_on < 0) g_lane_on = getenv("PA_KNN_LANE") != nullptr ? 1 : 0;
    if (!g_lane_on || n < 2048 || n > 4096 || m < 256) return 0;
    switch (nsample) {
        case 16: launch_lane<16>(b, n, m, xyz, new_xyz, idx, dist2, st, dbg); return 1;
        case 20: launch_lane<20>(b, n, m, xyz, new_xyz, idx, dist2, st, dbg); return 1;
        case 32: launch_lane<32>(b, n, m, xyz, new_xyz, idx, dist2, st, dbg); return 1;
        default: return 0;
    }
}
