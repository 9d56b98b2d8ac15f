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

}  // namespace

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
