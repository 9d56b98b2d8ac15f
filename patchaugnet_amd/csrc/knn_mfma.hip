// Retrieval kNN at database scale: brute-force k nearest descriptors with the distance matrix on the MFMA pipe, results IDENTICAL to
// the exact direct-sum kernel (knn_generic.hip, i.e. KNN_CUDA knn.cu:29-183 semantics: ssd accumulated in dimension order, ties by row).
//
// Reference call site: the recall harness builds a KDTree per reference trip and queries k = max(26, |db|/100 + 1) neighbours for every
// submap of every other trip (datasets/scene_dataset.py:1016-1099, place_recognition_dataset.py:52-70); KNN_CUDA's own formulation is the
// full nr x nq distance matrix followed by a per-column insertion sort (knn.cu:232-269).  knn_generic_kernel (wave per query, VALU) does
// 3 k x 3 k x 256 in 1 ms; a 20 k x 20 k database would take ~50 ms there.
//
// Three steps, the first on pa_tgemm_nn (train_gemm.hip, act = 2 epilogue), the other two here:
//   1. approximate squared distances  a(q, r) = max(|q|^2 + |r|^2 - 2 q.r, 0)  for a block of queries: fp32 MFMA GEMM, norms in the epilogue;
//   2. per query (one wavefront): the k-th smallest a by a three-pass radix select over the row's float bits (non-negative floats order
//      like unsigned integers), then every row with a <= a_k + 2 eps goes to a candidate list.  eps bounds |a - e| where e is the exact
//      direct sum: eps = 2 (dim + 8) 2^-24 (|q| + max|r|)^2.  Claim: the exact top-k is inside the candidates -- any row of the exact
//      top-k S has a <= e + eps <= max_S e + eps, and a_k >= max_S e - eps (otherwise k rows would have e < max_S e);
//   3. exact re-rank: each candidate's e is recomputed with the SAME arithmetic as knn_generic_kernel (sequential fp32 sum over the
//      dimensions, no contraction), the 64-bit keys (e bits, row) are sorted in LDS and the first k leave as (sqrt(e), row + 1).
// A query whose candidate list would overflow (more than KM_CAP rows within 2 eps of the k-th: near-duplicate-heavy databases) is
// flagged instead; the caller reruns flagged queries through the exact kernel.  So every returned column equals pa_knn_generic's, bit
// for bit (tests/test_gpu_retrieval_mfma.py).
#include "pa_common.h"

namespace {

constexpr int KM_CAP = 1024;      // candidate slots per query
constexpr int KM_BINS = 2048;

__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// k-th smallest (1-based rank `rank`) among the row's keys that match `prefix` on the bits above `shift + bits`; returns the bin and
// updates rank to the rank inside that bin.
__device__ __forceinline__ u32 radix_pass(const float *__restrict__ row, int nd, u32 *hist, int lane, u32 prefix, int hi_shift, int shift, int bits, int &rank)
{
    const int nbins = 1 << bits;
    for (int b = lane; b < nbins; b += 64) hist[b] = 0u;
    __syncthreads();
    for (int i = lane; i < nd; i += 64) {
        const u32 key = __float_as_uint(row[i]);
        if (hi_shift >= 32 || (key >> hi_shift) == prefix) atomicAdd(&hist[(key >> shift) & (nbins - 1)], 1u);
    }
    __syncthreads();
    // lane l owns bins [l * per, (l + 1) * per)
    const int per = nbins / 64;
    int mine = 0;
    for (int j = 0; j < per; ++j) mine += (int)hist[lane * per + j];
    const int incl = wave_incl_scan(mine, lane);
    const int excl = incl - mine;
    const u64 hit = __ballot(incl >= rank);
    const int owner = __ffsll((long long)hit) - 1;
    int bin = 0, newrank = 0;
    if (lane == owner) {
        int acc = excl;
        for (int j = 0; j < per; ++j) {
            const int c = (int)hist[lane * per + j];
            if (acc + c >= rank) { bin = lane * per + j; newrank = rank - acc; break; }
            acc += c;
        }
    }
    bin = __shfl(bin, owner);
    rank = __shfl(newrank, owner);
    __syncthreads();
    return (u32)bin;
}

// One workgroup (= one wavefront) per query.  a: (nq_blk, lda) approximate distances of this block of queries; ref_rows (nr, dim) and
// query_rows (nq_blk, dim) row-major; dist_out / ind_out: (k, nq_total) column q0 + q; flags (nq_total).
__global__ __launch_bounds__(64) void knn_select_kernel(const float *__restrict__ a, long lda, int nr, int dim, int k, const float *__restrict__ ref_rows,
                                                         const float *__restrict__ query_rows, const float *__restrict__ qnorm, const float *__restrict__ rnorm_max,
                                                         int q0, int nq_total, float *__restrict__ dist_out, long long *__restrict__ ind_out, int *__restrict__ flags)
{
    __shared__ u32 hist[KM_BINS];
    __shared__ u64 keys[KM_CAP];
    __shared__ int ncand;
    extern __shared__ float qrow[];   // [dim]
    const int q = blockIdx.x, lane = threadIdx.x;
    const float *row = a + (size_t)q * lda;
    for (int d = lane; d < dim; d += 64) qrow[d] = query_rows[(size_t)q * dim + d];
    if (lane == 0) ncand = 0;
    // ---- k-th smallest approximate distance: 11 + 11 + 10 bits
    int rank = k;
    const u32 b0 = radix_pass(row, nr, hist, lane, 0u, 32, 21, 11, rank);
    const u32 b1 = radix_pass(row, nr, hist, lane, b0, 21, 10, 11, rank);
    const u32 p1 = (b0 << 11) | b1;
    const u32 b2 = radix_pass(row, nr, hist, lane, p1, 10, 0, 10, rank);
    const float ak = __uint_as_float((p1 << 10) | b2);
    const float s = sqrtf(qnorm[q]) + sqrtf(rnorm_max[0]);
    const float eps = 2.f * (float)(dim + 8) * 5.9604645e-8f * s * s;
    const float thr = ak + 2.f * eps;
    // ---- candidates: every row within 2 eps of the k-th
    for (int c = 0; c < nr; c += 64) {
        const int i = c + lane;
        const bool take = i < nr && row[i] <= thr;
        const u64 m = __ballot(take);
        if (m) {
            const int base = ncand;                          // wave-uniform read (single wave per workgroup; LDS ops are in order)
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (take && pos < KM_CAP) keys[pos] = (u64)(u32)i;
            __syncthreads();
            if (lane == 0) ncand = base + __popcll(m);
            __syncthreads();
        }
    }
    const int nc = ncand;
    if (nc > KM_CAP) {                                       // too many rows tie with the k-th within the error bound: exact kernel for this query
        if (lane == 0) flags[q0 + q] = 1;
        return;
    }
    if (lane == 0) flags[q0 + q] = 0;
    // ---- exact distances of the candidates, knn_generic_kernel's arithmetic: ssd += (ref - q)^2 in dimension order
    int n2 = 64;
    while (n2 < nc) n2 <<= 1;
    for (int j = lane; j < n2; j += 64) {
        u64 key = ~0ull;
        if (j < nc) {
            const u32 r = (u32)keys[j];
            const float *rp = ref_rows + (size_t)r * dim;
            float ssd = 0.f;
            if ((dim & 3) == 0) {
                for (int d = 0; d < dim; d += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(rp + d);
                    float t = v.x - qrow[d];     ssd += t * t;
                    t = v.y - qrow[d + 1];       ssd += t * t;
                    t = v.z - qrow[d + 2];       ssd += t * t;
                    t = v.w - qrow[d + 3];       ssd += t * t;
                }
            } else {
                for (int d = 0; d < dim; ++d) { const float t = rp[d] - qrow[d]; ssd += t * t; }
            }
            key = pa_make_key(ssd, r);
        }
        keys[j] = key;
    }
    __syncthreads();
    // ---- bitonic sort of n2 keys (ascending (distance, row))
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (n2 >> 1); t += 64) {
                const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const u64 x = keys[lo], y = keys[hi];
                if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
            }
            __syncthreads();
        }
    for (int j = lane; j < k; j += 64) {
        const u64 key = keys[j];
        dist_out[(size_t)j * nq_total + q0 + q] = sqrtf(__uint_as_float((u32)(key >> 32)));
        ind_out[(size_t)j * nq_total + q0 + q] = (long long)(u32)key + 1;
    }
}

}  // namespace

// Steps 2 + 3 for a block of nq_blk queries whose approximate distances a (nq_blk x lda, row per query) pa_tgemm_nn has produced.
// ref_rows (nr, dim), query_rows (nq_blk, dim): row-major copies; qnorm (nq_blk): |q|^2; rnorm_max (1): max |r|^2 (device scalar).
// Writes columns q0 .. q0 + nq_blk - 1 of dist / ind ((k, nq_total), KNN_CUDA layout, 1-based int64 indices) and flags[q0 ..]:
// 1 = candidate overflow, column not written (rerun that query through pa_knn_generic).
PA_API int pa_knn_mfma_select(const float *a, long lda, int nq_blk, int nr, int dim, int k, const float *ref_rows, const float *query_rows,
                              const float *qnorm, const float *rnorm_max, int q0, int nq_total, float *dist, int64_t *ind, int *flags, pa_stream_t stream)
{
    PA_REQUIRE(a && ref_rows && query_rows && qnorm && rnorm_max && dist && ind && flags, "pa_knn_mfma_select: null pointer");
    PA_REQUIRE(nq_blk > 0 && nr > 0 && dim > 0 && k > 0 && k <= nr && k <= KM_CAP / 2, "pa_knn_mfma_select: nq=%d nr=%d dim=%d k=%d (k <= %d)", nq_blk, nr, dim, k, KM_CAP / 2);
    PA_REQUIRE((size_t)dim * 4 <= 32 * 1024, "pa_knn_mfma_select: dim=%d too large for the LDS query row", dim);
    hipLaunchKernelGGL(knn_select_kernel, dim3(nq_blk), dim3(64), (size_t)dim * 4, (hipStream_t)stream, a, lda, nr, dim, k, ref_rows, query_rows, qnorm, rnorm_max,
                       q0, nq_total, dist, reinterpret_cast<long long *>(ind), flags);
    PA_CHECK_LAUNCH("pa_knn_mfma_select");
    return PA_OK;
}
