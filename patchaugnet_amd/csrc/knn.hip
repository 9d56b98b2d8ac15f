// K4 -- k-nearest-neighbour query for gfx950.
//
// Reference semantics: libs/pointops/src/knnquery/knnquery_cuda_kernel.cu:6-50: for every
// query the first nsample entries of the stable ascending sort of (d2, index), d2 in fp32 as
// (qx-x)*(qx-x) + (qy-y)*(qy-y) + (qz-z)*(qz-z) without FMA contraction; unfilled slots hold (index 0, +inf).
//
// MI355X design (not the reference's one-thread-per-query local-memory insertion sort): one WAVEFRONT owns one
// query.  The cloud is staged once per workgroup into LDS as SoA; each lane evaluates one source point per step;
// the running top-k lives in the wave's registers as ONE 64-bit key per lane, sorted across lanes
//      key = float_bits(d2) << 32 | index        (d2 >= 0, so unsigned order == (d2, index) order)
// A ballot finds the lanes whose candidate beats the current k-th key (kept in SGPRs); each such candidate is
// inserted with a single wave_shr:1 DPP shift + two compares.  No local memory, no shared-memory sort.
#include <stdlib.h>

#include "pa_common.h"
#include "pa_cellsort.h"

namespace {

constexpr u64 KNN_INF0 = ((u64)0x7F800000u) << 32;  // (+inf, index 0): the reference's empty slot (:23-26)

// insert candidate ck into the lane-distributed ascending list (lane l holds entry l)
__device__ __forceinline__ u64 knn_insert(u64 list, u64 ck)
{
    const u64 up = pa_dpp_u64<PA_DPP_WAVE_SHR1, 0xf>(list);  // lane l <- entry l-1, lane 0 <- 0
    const u64 ins = up > ck ? up : ck;
    return list > ck ? ins : list;
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void knn_wave_kernel(int n, int m, int k, int q_per_block, const float *__restrict__ xyz_all,
                                                         const float *__restrict__ new_xyz_all, int *__restrict__ idx_all,
                                                         float *__restrict__ dist2_all)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    const float *sx, *sy, *sz;
    if (USE_LDS) {
        float *wx = smem, *wy = smem + n, *wz = smem + 2 * n;
        for (int i = tid; i < n; i += 256) {
            wx[i] = xyz[i * 3 + 0];
            wy[i] = xyz[i * 3 + 1];
            wz[i] = xyz[i * 3 + 2];
        }
        __syncthreads();
        sx = wx; sy = wy; sz = wz;
    }
    const int q_begin = blockIdx.x * q_per_block;
    const int q_end = min(q_begin + q_per_block, m);
    for (int q = q_begin + wave; q < q_end; q += 4) {
        const float *qp = new_xyz_all + ((size_t)b * m + q) * 3;
        const float qx = qp[0], qy = qp[1], qz = qp[2];
        u64 list = KNN_INF0;
        u64 thresh = KNN_INF0;  // wave-uniform: current k-th best
        for (int c = 0; c < n; c += 64) {
            const int i = c + lane;
            float x = 0.f, y = 0.f, z = 0.f;
            if (i < n) {
                if (USE_LDS) { x = sx[i]; y = sy[i]; z = sz[i]; }
                else { x = xyz[i * 3 + 0]; y = xyz[i * 3 + 1]; z = xyz[i * 3 + 2]; }
            }
            const float d2 = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);  // :31
            const u64 key = pa_make_key(d2, (u32)i);
            // key < thresh also rejects +inf and NaN distances (their bit patterns are >= +inf's): the reference's
            // strict "d2 < best[j]" against 1e40 never admits them either
            u64 mask = __ballot(i < n && key < thresh);
            while (mask) {
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const u64 ck = pa_readlane_u64(key, src);
                if (ck < thresh) {
                    list = knn_insert(list, ck);
                    thresh = pa_readlane_u64(list, k - 1);
                }
            }
        }
        if (lane < k) {
            const size_t o = ((size_t)b * m + q) * k + lane;
            idx_all[o] = (int)(u32)list;
            dist2_all[o] = __uint_as_float((u32)(list >> 32));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Spatially pruned variant (same results, bit for bit).  The brute-force kernel above evaluates all n sources for every
// query and spends most of its instructions inserting early, far-away candidates into the running list.  Here every
// workgroup first sorts its cloud into 512 Morton-ordered cells (counting sort in LDS, a few microseconds), so that each
// run of 64 consecutive sorted points -- a "chunk" -- is spatially compact, and records each chunk's bounding box.  A query
// (still one wavefront) then
//   1. computes in one wave-wide step the squared distance to all <= 64 chunk boxes (one lane per chunk);
//   2. visits chunks in ascending box distance (a wave-min per visit) and stops at the first box farther than the current
//      k-th neighbour -- typically ~10 of 64 chunks at n = 4096, k = 20;
//   3. seeds the list from the nearest chunk with one bitonic sort instead of ~45 serial insertions.
// Exactness: a box distance is computed with the same fp32 operation order as a point distance on per-axis differences that
// are, by monotonicity of IEEE rounding, no larger in magnitude than those of any point inside the box, so it never exceeds
// the fp32 distance of a member point; a chunk is skipped only when its box distance is STRICTLY greater than the k-th
// distance, so equal-distance candidates with a lower index are still seen.  The order inside a cell depends on LDS atomics
// and varies between runs; the selected (distance, index) keys do not.  Points with non-finite coordinates can never be
// selected (their distance is never < a finite or infinite best) and are left out of the chunks.
// ascending bitonic sort of one u64 key per lane across the wavefront
__device__ __forceinline__ u64 kg_sort64(u64 key, int lane)
{
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const u64 other = kg_shfl_xor_u64(key, j);
            const bool up = (lane & k) == 0, low = (lane & j) == 0;
            const u64 mn = key < other ? key : other, mx = key < other ? other : key;
            key = (low == up) ? mn : mx;
        }
    return key;
}

template <int PTS, int NT, bool DBG>   // NT threads per workgroup, PTS points per thread: n <= NT * PTS; DBG: cycle stamps for tools/knn_phases.py
__global__ __launch_bounds__(NT, NT / 128) void knn_grid_kernel(int n, int m, int k, int q_per_block, const float *__restrict__ xyz_all,
                                                         const float *__restrict__ new_xyz_all, int *__restrict__ idx_all,
                                                         float *__restrict__ dist2_all, long long *dbg)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const long long t_start = DBG ? (long long)__builtin_readcyclecounter() : 0;
    long long n_visit = 0, n_ins = 0, t_sort = 0;
    float4 *sorted = reinterpret_cast<float4 *>(smem);                        // [n] x, y, z, original index (bits)
    float *box = smem + 4 * (size_t)n;                                        // [64][8] lo.xyz, hi.xyz
    int *cnt = reinterpret_cast<int *>(box + 64 * 8);                         // [KG_CELLS + 1] histogram -> running offsets
    float *red = reinterpret_cast<float *>(cnt + KG_CELLS + 1);               // [NW][6] cross-wave bounding box
    constexpr int NW = NT / 64;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    int nchunks;
    const int n_valid = cell_sort_cloud<PTS, NT>(n, xyz, sorted, box, cnt, red, &nchunks);
    const long long t_pro = DBG ? (long long)__builtin_readcyclecounter() : 0;
    // ---- 5. queries
    const int q_begin = blockIdx.x * q_per_block;
    const int q_end = min(q_begin + q_per_block, m);
    for (int q = q_begin + wave; q < q_end; q += NW) {
        const float *qp = new_xyz_all + ((size_t)b * m + q) * 3;
        const float qx = qp[0], qy = qp[1], qz = qp[2];
        // box distance of chunk `lane`: per axis the signed difference to the nearest face (0 inside), same order of operations as a point
        u64 ckey = ~0ull;
        if (lane < nchunks) {
            const float *bx = box + lane * 8;
            const float ax = qx < bx[0] ? qx - bx[0] : (qx > bx[3] ? qx - bx[3] : 0.f);
            const float ay = qy < bx[1] ? qy - bx[1] : (qy > bx[4] ? qy - bx[4] : 0.f);
            const float az = qz < bx[2] ? qz - bx[2] : (qz > bx[5] ? qz - bx[5] : 0.f);
            const float lb = ax * ax + ay * ay + az * az;
            ckey = pa_make_key(lb, (u32)lane);
        }
        u64 list = KNN_INF0, thresh = KNN_INF0;
        // seed: the chunk with the nearest box, sorted in one go
        int c = -1;
        {
            const u64 best = kg_wave_min_u64(ckey);
            if (best != ~0ull && (u32)(best >> 32) <= 0x7F800000u) c = (int)(u32)best;   // NaN box distance (NaN query): nothing to do
        }
        bool seed = true;
        while (c >= 0) {
            if (DBG) ++n_visit;
            const int i = c * 64 + lane;
            u64 key = ~0ull;
            if (i < n_valid) {
                const float4 p = sorted[i];
                const float d2 = (qx - p.x) * (qx - p.x) + (qy - p.y) * (qy - p.y) + (qz - p.z) * (qz - p.z);  // knnquery_cuda_kernel.cu:31
                key = pa_make_key(d2, (u32)__float_as_int(p.w));
                if (key >= KNN_INF0) key = ~0ull;                                 // inf / NaN distance: never admitted (:32 strict <)
            }
            if (seed) {
                seed = false;
                const long long ts = DBG ? (long long)__builtin_readcyclecounter() : 0;
                const u64 s = kg_sort64(key, lane);
                if (DBG) t_sort += (long long)__builtin_readcyclecounter() - ts;
                list = s == ~0ull ? KNN_INF0 : s;
            } else {
                // every candidate below the k-th key AS OF THE START of this chunk is inserted; one that turns out not to be among the
                // k best just lands behind position k-1, so the per-candidate re-check (two readlanes on the critical path) is dropped
                u64 mask = __ballot(key < thresh);
                while (mask) {
                    const int src = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    list = knn_insert(list, pa_readlane_u64(key, src));
                    if (DBG) ++n_ins;
                }
            }
            thresh = pa_readlane_u64(list, k - 1);
            // next: the nearest unvisited box, if it is not farther than the k-th neighbour (ascending box distance keeps both the
            // number of chunks visited and the number of insertions about 2x lower than index order)
            if (lane == c) ckey = ~0ull;
            const u64 best = kg_wave_min_u64(ckey);
            c = (best != ~0ull && (u32)(best >> 32) <= (u32)(thresh >> 32)) ? (int)(u32)best : -1;
        }
        if (lane < k) {
            const size_t o = ((size_t)b * m + q) * k + lane;
            idx_all[o] = (int)(u32)list;
            dist2_all[o] = __uint_as_float((u32)(list >> 32));
        }
    }
    if (DBG && dbg && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        dbg[0] = t_pro - t_start; dbg[1] = (long long)__builtin_readcyclecounter() - t_pro; dbg[2] = n_visit; dbg[3] = n_ins; dbg[4] = t_sort;
        dbg[5] = (q_end - q_begin + NW - 1) / NW;
    }
}

// nsample > 64: successive-minimum selection, one wave per query, O(k * n / 64) per query.  Rarely used
// (every shipped config has nsample <= 40); kept so that any nsample the reference accepts works.
__global__ __launch_bounds__(256) void knn_select_kernel(int n, int m, int k, const float *__restrict__ xyz_all,
                                                           const float *__restrict__ new_xyz_all, int *__restrict__ idx_all,
                                                           float *__restrict__ dist2_all)
{
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= m) return;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    const float *qp = new_xyz_all + ((size_t)b * m + q) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const size_t o = ((size_t)b * m + q) * k;
    bool have_prev = false;
    u64 prev = 0;
    for (int s = 0; s < k; ++s) {
        u64 best = ~0ull;  // min over keys > prev
        for (int i = lane; i < n; i += 64) {
            const float x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
            const float d2 = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);
            const u64 key = pa_make_key(d2, (u32)i);
            if (key < KNN_INF0 && (!have_prev || key > prev) && key < best) best = key;
        }
        const u64 g = ~pa_wave_max_u64(~best);  // wave min
        if (g == ~0ull) {  // no more admissible points: remaining slots stay (0, +inf)
            for (int r = s + lane; r < k; r += 64) { idx_all[o + r] = 0; dist2_all[o + r] = __uint_as_float(0x7F800000u); }
            break;
        }
        if (lane == 0) { idx_all[o + s] = (int)(u32)g; dist2_all[o + s] = __uint_as_float((u32)(g >> 32)); }
        prev = g;
        have_prev = true;
    }
}

}  // namespace


int pa_knn_quad_try(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx, float *dist2, hipStream_t st, long long *dbg, int mq = 0,
                    const float *cells = nullptr);   // knn_quad.hip

static long long *g_knn_dbg = nullptr;
PA_API void pa_knn_debug_buffer(long long *buf) { g_knn_dbg = buf; }   // profiling hook (6 int64), NULL = off

PA_API int pa_knnquery(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx, float *dist2, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && m > 0 && nsample > 0, "pa_knnquery: b=%d n=%d m=%d nsample=%d must be positive", b, n, m, nsample);
    PA_REQUIRE(xyz && new_xyz && idx && dist2, "pa_knnquery: null pointer");
    PA_REQUIRE(b <= 65535, "pa_knnquery: b=%d exceeds the grid limit 65535", b);
    hipStream_t st = (hipStream_t)stream;
    if (nsample > 64) {
        hipLaunchKernelGGL(knn_select_kernel, dim3(pa_div_up(m, 4), b), dim3(256), 0, st, n, m, nsample, xyz, new_xyz, idx, dist2);
        PA_CHECK_LAUNCH("pa_knnquery(select)");
        return PA_OK;
    }
    // 2048-4096 source points, 16 / 20 / 32 neighbours: four lanes per query over an 8 x 8 x 8 grid (knn_quad.hip)
    if (pa_knn_quad_try(b, n, m, nsample, xyz, new_xyz, idx, dist2, st, g_knn_dbg)) {
        PA_CHECK_LAUNCH("pa_knnquery(quad)");
        return PA_OK;
    }
    // queries per workgroup: enough workgroups to fill 256 CUs several times over, but amortise the LDS staging
    int qpb = 4;
    while (qpb < 64 && (long)b * pa_div_up(m, qpb * 2) >= 1024) qpb *= 2;
    static const bool no_grid = getenv("PA_KNN_NO_GRID") != nullptr;   // A/B knob
    // measured: pays from 2048 source points, and from 1024 when there are at least 256 queries to amortise the per-workgroup sort
    // (PPT-Net's second level 0.050 -> 0.039 ms; with 128 queries 0.029 -> 0.032 ms).  A function of (n, m) only: never of the batch size.
    if (!no_grid && n <= 4096 && (n >= 2048 || (n >= 1024 && m >= 256))) {
        // spatially pruned kernel: <= 64 chunks of 64 points, sorted cloud (16 n bytes) in LDS, 8 waves per workgroup so that the two
        // resident workgroups of a CU put four waves on every SIMD (the per-query work is a chain of dependent cross-lane steps)
        qpb = 16;
        while (qpb < 128 && (long)b * pa_div_up(m, qpb * 2) >= 512) qpb *= 2;
        const size_t lds = (size_t)n * 16 + 64 * 8 * 4 + (KG_CELLS + 1) * 4 + 16 * 6 * 4;
        auto kern = g_knn_dbg ? knn_grid_kernel<8, 512, true> : knn_grid_kernel<8, 512, false>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(pa_div_up(m, qpb), b), dim3(512), lds, st, n, m, nsample, qpb, xyz, new_xyz, idx, dist2, g_knn_dbg);
        PA_CHECK_LAUNCH("pa_knnquery(grid)");
        return PA_OK;
    }
    const size_t lds = (size_t)n * 12;
    if (lds <= 96 * 1024) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&knn_wave_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(knn_wave_kernel<true>, dim3(pa_div_up(m, qpb), b), dim3(256), lds, st, n, m, nsample, qpb, xyz, new_xyz, idx, dist2);
    } else {
        hipLaunchKernelGGL(knn_wave_kernel<false>, dim3(pa_div_up(m, qpb), b), dim3(256), 0, st, n, m, nsample, qpb, xyz, new_xyz, idx, dist2);
    }
    PA_CHECK_LAUNCH("pa_knnquery");
    return PA_OK;
}

// pa_knnquery for a WINDOW of every cloud's queries: queries q0 .. q0 + mq - 1 of the m per cloud (rows of new_xyz / idx / dist2 keep their m-query
// stride).  Same kernels, same results for those rows; returns PA_EUNSUPPORTED when the level's shape does not run the cell-grid kernel that
// takes windows (the caller then answers the whole level with pa_knnquery).
PA_API int pa_knnquery_window(int b, int n, int m, int nsample, int q0, int mq, const float *xyz, const float *new_xyz, int *idx, float *dist2, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && m > 0 && nsample > 0 && q0 >= 0 && mq > 0 && q0 + mq <= m, "pa_knnquery_window: bad arguments (m=%d q0=%d mq=%d)", m, q0, mq);
    PA_REQUIRE(xyz && new_xyz && idx && dist2 && b <= 65535, "pa_knnquery_window: null pointer or b > 65535");
    if (!pa_knn_quad_try(b, n, m, nsample, xyz, new_xyz + (size_t)q0 * 3, idx + (size_t)q0 * nsample, dist2 + (size_t)q0 * nsample, (hipStream_t)stream, g_knn_dbg, mq)) {
        pa_set_error("pa_knnquery_window: the level (n=%d, m=%d, nsample=%d) does not run the windowed cell-grid kernel", n, m, nsample);
        return PA_EUNSUPPORTED;
    }
    PA_CHECK_LAUNCH("pa_knnquery_window");
    return PA_OK;
}
