// Patch overlap-pair selection for the contrastive patch-feature term of the training step, on the device.
//
// Reference: the Python loops of train_one_epoch, place_recognition/train_place_recognition.py:308-372.  For a (query cloud m, positive
// cloud n) pair the precomputed overlap table holds up to thousands of records (idx1; near_indices2; far candidates), all in ORIGINAL
// point indices.  Per record the reference looks idx1 up among m's 1024 FPS centres (np.where ... [0]: first match), intersects the near
// list with n's centres (np.where(np.isin(n_centres, near))[0]: positions in n's centre list, ascending, unique), does the same for the far
// candidates, drops the record when any of the three is empty, and emits len(positives) triplets (query position, positive position,
// negative position drawn uniformly with replacement from the far positions).  It is O(records x 1024) numpy work plus three tiny H2D
// copies per triplet; the reference caps it at 500 records per pair because of that.
//
// Here: the two centre lists become inverse maps (original index -> first position), one workgroup (one wave) per record marks the near /
// far positions in two LDS bitmaps (bit p = position p of n's centre list: enumerating set bits IS the ascending unique order of np.where),
// pass 1 writes the triplet count per record, the host-side wrapper turns counts into offsets (a device cumsum), pass 2 writes the
// triplets.  The negative draw uses a counter-based hash of (seed, record, slot) instead of numpy's global generator: same distribution
// (uniform with replacement over the far positions), not the same stream -- stated in DESIGN.md and checked as such by the tests.
#include "pa_common.h"

namespace {

constexpr int PP_MAXW = 128;         // bitmap words: centre lists of up to 4096 positions
constexpr int PP_ABSENT = 0x7f7f7f7f;

__global__ void pp_invmap_kernel(int npoints, int m0, const int *__restrict__ center_m, const int *__restrict__ center_n, int *__restrict__ inv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m0) return;
    const int a = center_m[i], b = center_n[i];
    if (a >= 0 && a < npoints) atomicMin(inv + a, i);                 // first position of a value, like np.where(...)[0][0]
    if (b >= 0 && b < npoints) atomicMin(inv + npoints + b, i);
}

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ u64 mix64(u64 z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <bool FILL>
__global__ __launch_bounds__(64) void pp_record_kernel(int nrec, const int *__restrict__ idx1, const int *__restrict__ near_off, const int *__restrict__ near_v,
                                                        const int *__restrict__ far_off, const int *__restrict__ far_v, int npoints, int m0,
                                                        const int *__restrict__ inv, u64 seed, int *__restrict__ counts, const int *__restrict__ offsets,
                                                        int *__restrict__ out1, int *__restrict__ out2, int *__restrict__ out3)
{
    __shared__ u32 bm[2][PP_MAXW];
    __shared__ int pre[2][64];
    const int k = blockIdx.x, lane = threadIdx.x;
    bm[0][lane] = bm[0][lane + 64] = bm[1][lane] = bm[1][lane + 64] = 0u;
    __syncthreads();
    const int *inv_m = inv, *inv_n = inv + npoints;
    for (int i = near_off[k] + lane; i < near_off[k + 1]; i += 64) {
        const int v = near_v[i];
        if (v >= 0 && v < npoints) { const int p = inv_n[v]; if (p < m0) atomicOr(&bm[0][p >> 5], 1u << (p & 31)); }
    }
    for (int i = far_off[k] + lane; i < far_off[k + 1]; i += 64) {
        const int v = far_v[i];
        if (v >= 0 && v < npoints) { const int p = inv_n[v]; if (p < m0) atomicOr(&bm[1][p >> 5], 1u << (p & 31)); }
    }
    __syncthreads();
    // lane l owns words 2l, 2l+1 of each bitmap (contiguous, so lane order = position order)
    const u32 p0 = bm[0][2 * lane], p1 = bm[0][2 * lane + 1], n0 = bm[1][2 * lane], n1 = bm[1][2 * lane + 1];
    const int cp = __popc(p0) + __popc(p1), cn = __popc(n0) + __popc(n1);
    const int npos = wave_sum(cp), nneg = wave_sum(cn);
    const int q1 = idx1[k];
    const int q = (q1 >= 0 && q1 < npoints) ? inv_m[q1] : PP_ABSENT;
    const bool valid = q < m0 && npos > 0 && nneg > 0;
    if (!FILL) {
        if (lane == 0) counts[k] = valid ? npos : 0;
        return;
    }
    if (!valid) return;
    // exclusive prefix of the per-lane counts
    int ip = cp, in = cn;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int tp = __shfl_up(ip, o), tn = __shfl_up(in, o);
        if (lane >= o) { ip += tp; in += tn; }
    }
    pre[0][lane] = ip - cp;
    pre[1][lane] = in - cn;
    __syncthreads();
    const int base = offsets[k];
    // positives: this lane's set bits, ascending
    int w = pre[0][lane];
    for (u32 b = p0; b; b &= b - 1) { out2[base + w] = 64 * lane + __ffs(b) - 1; out1[base + w] = q; ++w; }
    for (u32 b = p1; b; b &= b - 1) { out2[base + w] = 64 * lane + 32 + __ffs(b) - 1; out1[base + w] = q; ++w; }
    // negatives: slot j draws the r-th far position, r uniform in [0, nneg)
    for (int j = lane; j < npos; j += 64) {
        const u64 h = mix64(seed ^ mix64(((u64)(u32)k << 32) | (u32)j));
        int r = (int)(((h >> 32) * (u64)nneg) >> 32);
        // owner lane: the last one whose exclusive prefix is <= r
        int lo = 0, hi = 63;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (pre[1][mid] <= r) lo = mid; else hi = mid - 1; }
        r -= pre[1][lo];
        u32 b = bm[1][2 * lo];
        int basebit = 64 * lo;
        if (r >= __popc(b)) { r -= __popc(b); b = bm[1][2 * lo + 1]; basebit += 32; }
        for (; r > 0; --r) b &= b - 1;
        out3[base + j] = basebit + __ffs(b) - 1;
    }
}

}  // namespace

// Pass 1.  idx1 (nrec), near_off / far_off (nrec + 1, CSR), near / far: original point indices; center_m / center_n: the m0 FPS centre indices
// of the two clouds (original indices, < npoints); scratch_inv: 2 * npoints ints (filled here, read again by pass 2);
// counts (nrec): triplets the record contributes (0 = dropped like the reference's `continue`s).
PA_API int pa_patch_pairs_count(int nrec, const int *idx1, const int *near_off, const int *near_v, const int *far_off, const int *far_v, int npoints, int m0,
                                const int *center_m, const int *center_n, int *scratch_inv, int *counts, pa_stream_t stream)
{
    PA_REQUIRE(nrec > 0 && idx1 && near_off && far_off && center_m && center_n && scratch_inv && counts, "pa_patch_pairs_count: bad arguments");
    PA_REQUIRE(m0 > 0 && m0 <= 32 * PP_MAXW && npoints > 0, "pa_patch_pairs_count: centre lists of up to %d positions (got %d)", 32 * PP_MAXW, m0);
    hipStream_t st = (hipStream_t)stream;
    if (pa_fill32(scratch_inv, 0x7f7f7f7fu, 2 * (size_t)npoints, st) != PA_OK) { pa_set_error("pa_patch_pairs_count: fill failed"); return PA_EINVAL; }   // a kernel, not a memset node
    hipLaunchKernelGGL(pp_invmap_kernel, dim3(pa_div_up(m0, 256)), dim3(256), 0, st, npoints, m0, center_m, center_n, scratch_inv);
    hipLaunchKernelGGL(pp_record_kernel<false>, dim3(nrec), dim3(64), 0, st, nrec, idx1, near_off, near_v, far_off, far_v, npoints, m0, scratch_inv, 0ull, counts,
                       (const int *)nullptr, (int *)nullptr, (int *)nullptr, (int *)nullptr);
    PA_CHECK_LAUNCH("pa_patch_pairs_count");
    return PA_OK;
}

// Pass 2.  offsets (nrec): exclusive prefix sum of pass 1's counts; out_*: (sum of counts) ints each: position of the query patch in m's
// centre list, of the positive and of the negative patch in n's (train_place_recognition.py:366-371: indices1, pos_indices2, neg_indices2).
PA_API int pa_patch_pairs_fill(int nrec, const int *idx1, const int *near_off, const int *near_v, const int *far_off, const int *far_v, int npoints, int m0,
                               const int *scratch_inv, unsigned long long seed, const int *offsets, int *out_idx1, int *out_pos2, int *out_neg2, pa_stream_t stream)
{
    PA_REQUIRE(nrec > 0 && idx1 && near_off && far_off && scratch_inv && offsets && out_idx1 && out_pos2 && out_neg2, "pa_patch_pairs_fill: bad arguments");
    PA_REQUIRE(m0 > 0 && m0 <= 32 * PP_MAXW && npoints > 0, "pa_patch_pairs_fill: centre lists of up to %d positions (got %d)", 32 * PP_MAXW, m0);
    hipLaunchKernelGGL(pp_record_kernel<true>, dim3(nrec), dim3(64), 0, (hipStream_t)stream, nrec, idx1, near_off, near_v, far_off, far_v, npoints, m0, scratch_inv,
                       (u64)seed, (int *)nullptr, offsets, out_idx1, out_pos2, out_neg2);
    PA_CHECK_LAUNCH("pa_patch_pairs_fill");
    return PA_OK;
}
