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
#include "pa_common.h"

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

PA_API int pa_knnquery(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx, float *dist2,
                       pa_stream_t stream)
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
    // queries per workgroup: enough workgroups to fill 256 CUs several times over, but amortise the LDS staging
    int qpb = 4;
    while (qpb < 64 && (long)b * pa_div_up(m, qpb * 2) >= 1024) qpb *= 2;
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
