// N1-N3 -- generic-dimension brute-force kNN with the KNN_CUDA operator contract (gfx950).
//
// Reference semantics: libs/KNN_CUDA/knn_cuda/csrc/cuda/knn.cu:29-93 (full nr x nq squared-distance matrix,
// ssd accumulated in dimension order as ssd += tmp*tmp), :105-167 (per-column insertion, order (dist asc, row asc)),
// :178-183 (sqrt), indices 1-based int64 (knn.cpp:23-56 returns them; the Python wrapper subtracts 1).
//
// MI355X design: no O(nr*nq) scratch matrix and no serial per-column insertion sort.  One wavefront owns one query:
// its column is held in LDS, each lane accumulates the distance to one reference row per step (reference rows are
// contiguous along nr, so the loads are coalesced), and the running top-k is the same lane-distributed sorted
// 64-bit-key list as in knn.hip.  k <= 64 uses that list; larger k falls back to successive-minimum selection.
#include "pa_common.h"

namespace {

constexpr u64 INF0 = ((u64)0x7F800000u) << 32;

__device__ __forceinline__ u64 list_insert(u64 list, u64 ck)
{
    const u64 up = pa_dpp_u64<PA_DPP_WAVE_SHR1, 0xf>(list);
    const u64 ins = up > ck ? up : ck;
    return list > ck ? ins : list;
}

__device__ __forceinline__ float row_dist(const float *__restrict__ ref, int nr, int r, const float *qcol, int dim)
{
    float ssd = 0.f;
    for (int d = 0; d < dim; ++d) {
        const float tmp = ref[(size_t)d * nr + r] - qcol[d];  // knn.cu:80-83
        ssd += tmp * tmp;
    }
    return ssd;
}

__global__ __launch_bounds__(256) void knn_generic_kernel(const float *__restrict__ ref, int nr, const float *__restrict__ query, int nq, int dim,
                                                            int k, float *__restrict__ dist_out, long long *__restrict__ ind_out)
{
    extern __shared__ float qcols[];  // [4][dim]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    float *qcol = qcols + wave * dim;
    if (q < nq)
        for (int d = lane; d < dim; d += 64) qcol[d] = query[(size_t)d * nq + q];
    __syncthreads();
    if (q >= nq) return;
    if (k <= 64) {
        u64 list = INF0, thresh = INF0;
        for (int c = 0; c < nr; c += 64) {
            const int r = c + lane;
            const float ssd = r < nr ? row_dist(ref, nr, r, qcol, dim) : 0.f;
            const u64 key = pa_make_key(ssd, (u32)r);
            u64 mask = __ballot(r < nr && key < thresh);
            while (mask) {
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const u64 ck = pa_readlane_u64(key, src);
                if (ck < thresh) {
                    list = list_insert(list, ck);
                    thresh = pa_readlane_u64(list, k - 1);
                }
            }
        }
        if (lane < k) {
            dist_out[(size_t)lane * nq + q] = sqrtf(__uint_as_float((u32)(list >> 32)));
            ind_out[(size_t)lane * nq + q] = (long long)(u32)list + 1;
        }
    } else {
        bool have_prev = false;
        u64 prev = 0;
        for (int s = 0; s < k; ++s) {
            u64 best = ~0ull;
            for (int r = lane; r < nr; r += 64) {
                const u64 key = pa_make_key(row_dist(ref, nr, r, qcol, dim), (u32)r);
                if (key < INF0 && (!have_prev || key > prev) && key < best) best = key;
            }
            const u64 g = ~pa_wave_max_u64(~best);
            if (g == ~0ull) break;
            if (lane == 0) {
                dist_out[(size_t)s * nq + q] = sqrtf(__uint_as_float((u32)(g >> 32)));
                ind_out[(size_t)s * nq + q] = (long long)(u32)g + 1;
            }
            prev = g;
            have_prev = true;
        }
    }
}

// ---- kNN over a per-query CANDIDATE LIST (the hard-negative refresh of training: datasets/scene_dataset.py:1101-1113 builds a KD-tree over
// every query's own <= 3000 sampled negatives).  One wavefront per query: the query row sits in LDS, lane l takes candidate positions l, l + 64, ..
// of the query's list, accumulates the squared distance in dimension order (ssd += tmp * tmp: the arithmetic of row_dist / knn.cu:80-83, so the
// result equals pa_knn_generic on the gathered rows bit for bit) and the running top-k is the lane-distributed sorted key list with the LIST
// POSITION as the low word: ties resolve to the earlier position, as a kNN over the gathered rows would.  Rows are row-major (nref, dim):
// a lane streams its own candidate's row with 16-byte loads.
__global__ __launch_bounds__(256) void knn_cand_kernel(const float *__restrict__ ref_rows, int dim, const float *__restrict__ q_rows, int nq,
                                                         const long long *__restrict__ cand, int L, int k, long long *__restrict__ out)
{
    extern __shared__ float qcols[];  // [4][dim]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    float *qcol = qcols + wave * dim;
    if (q < nq)
        for (int d = lane; d < dim; d += 64) qcol[d] = q_rows[(size_t)q * dim + d];
    __syncthreads();
    if (q >= nq) return;
    const long long *cl = cand + (size_t)q * L;
    u64 list = INF0, thresh = INF0;
    const bool vec = (dim & 3) == 0 && (reinterpret_cast<uintptr_t>(ref_rows) & 15) == 0;
    for (int c = 0; c < L; c += 64) {
        const int pos = c + lane;
        const long long row = pos < L ? cl[pos] : -1;
        float ssd = 0.f;
        if (row >= 0) {
            const float *r = ref_rows + (size_t)row * dim;
            if (vec) {
                for (int d = 0; d < dim; d += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(r + d);
                    float t;
                    t = v.x - qcol[d]; ssd += t * t;
                    t = v.y - qcol[d + 1]; ssd += t * t;
                    t = v.z - qcol[d + 2]; ssd += t * t;
                    t = v.w - qcol[d + 3]; ssd += t * t;
                }
            } else {
                for (int d = 0; d < dim; ++d) { const float t = r[d] - qcol[d]; ssd += t * t; }
            }
        }
        const u64 key = pa_make_key(ssd, (u32)pos);
        u64 mask = __ballot(row >= 0 && key < thresh);
        while (mask) {
            const int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const u64 ck = pa_readlane_u64(key, src);
            if (ck < thresh) {
                list = list_insert(list, ck);
                thresh = pa_readlane_u64(list, k - 1);
            }
        }
    }
    if (lane < k) out[(size_t)q * k + lane] = list < INF0 ? cl[(u32)list] : -1;      // fewer than k live candidates: -1
}

}  // namespace

// out (nq, k) int64: for every query the k candidates of ITS list (cand (nq, L) int64 row indices into ref_rows, -1 = padding) nearest to it, nearest
// first, ties to the earlier list position; -1 where the list has fewer than k live entries.  ref_rows (nref, dim), q_rows (nq, dim) row-major; k <= 64.
PA_API int pa_knn_candidates(const float *ref_rows, int dim, const float *q_rows, int nq, const int64_t *cand, int L, int k, int64_t *out, pa_stream_t stream)
{
    PA_REQUIRE(ref_rows && q_rows && cand && out && dim > 0 && nq > 0 && L > 0, "pa_knn_candidates: bad arguments");
    PA_REQUIRE(k > 0 && k <= 64, "pa_knn_candidates: k=%d must be 1..64", k);
    PA_REQUIRE((size_t)dim * 16 <= 64 * 1024, "pa_knn_candidates: dim=%d too large for the LDS query tile", dim);
    hipLaunchKernelGGL(knn_cand_kernel, dim3(pa_div_up(nq, 4)), dim3(256), (size_t)dim * 16, (hipStream_t)stream, ref_rows, dim, q_rows, nq,
                       reinterpret_cast<const long long *>(cand), L, k, reinterpret_cast<long long *>(out));
    PA_CHECK_LAUNCH("pa_knn_candidates");
    return PA_OK;
}

PA_API int pa_knn_generic(const float *ref, int nr, const float *query, int nq, int dim, int k, float *dist, int64_t *ind, pa_stream_t stream)
{
    PA_REQUIRE(nr > 0 && nq > 0 && dim > 0 && k > 0, "pa_knn_generic: nr=%d nq=%d dim=%d k=%d must be positive", nr, nq, dim, k);
    PA_REQUIRE(k <= nr, "pa_knn_generic: k=%d exceeds the number of reference points %d", k, nr);
    PA_REQUIRE(ref && query && dist && ind, "pa_knn_generic: null pointer");
    PA_REQUIRE((size_t)dim * 16 <= 64 * 1024, "pa_knn_generic: dim=%d too large for the LDS query tile", dim);
    hipLaunchKernelGGL(knn_generic_kernel, dim3(pa_div_up(nq, 4)), dim3(256), (size_t)dim * 16, (hipStream_t)stream, ref, nr, query, nq, dim, k,
                       dist, reinterpret_cast<long long *>(ind));
    PA_CHECK_LAUNCH("pa_knn_generic");
    return PA_OK;
}
