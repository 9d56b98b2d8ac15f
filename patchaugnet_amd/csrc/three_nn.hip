// K9 -- three nearest neighbours, K13 -- ball query, K14 -- featuredistribute, K16 -- label statistics (gfx950).
//
// Reference semantics:
//   K9  libs/pointops/src/interpolation/interpolation_cuda_kernel.cu:134-176 : sorted 3 best by strict '<' in scan order
//       => (d2 asc, index asc); d2 = (ux-x)*(ux-x) + (uy-y)*(uy-y) + (uz-z)*(uz-z), fp32, no FMA; empty slot = (0, +inf).
//   K13 libs/pointops/src/ballquery/ballquery_cuda_kernel.cu:47-80 : first nsample indices in scan order with
//       d2 < radius*radius; remaining slots repeat the first hit; no hit => slots untouched.
//   K14 libs/pointops/src/featuredistribute/featuredistribute_cuda_kernel.cu:4-30
//   K16 libs/pointops/src/labelstat/labelstat_cuda_kernel.cu:6-49, :74-105, :131-151
//
// These are scans of ONE small cloud by MANY queries: the cloud is staged through LDS in float4-padded tiles and
// every lane reads the same element per step (an LDS broadcast, conflict-free), instead of each thread streaming
// the cloud from global memory as the reference does.
#include "pa_common.h"

namespace {

constexpr int TILE = 2048;  // points per LDS tile (32 KiB as float4)

__device__ __forceinline__ void stage_tile(float4 *s, const float *__restrict__ src, int count, int tid, int nt)
{
    for (int i = tid; i < count; i += nt) s[i] = make_float4(src[i * 3 + 0], src[i * 3 + 1], src[i * 3 + 2], 0.f);
}

// one candidate against the running (d2 asc, index asc) top three: the strict compares of the reference
// (interpolation_cuda_kernel.cu:156-170), a tie keeps the earlier index.
__device__ __forceinline__ void top3_insert(float d, int gi, float &b1, float &b2, float &b3, int &i1, int &i2, int &i3)
{
    const bool c1 = d < b1, c2 = d < b2, c3 = d < b3;
    b3 = c2 ? b2 : (c3 ? d : b3);
    i3 = c2 ? i2 : (c3 ? gi : i3);
    b2 = c1 ? b1 : (c2 ? d : b2);
    i2 = c1 ? i1 : (c2 ? gi : i2);
    b1 = c1 ? d : b1;
    i1 = c1 ? gi : i1;
}

// K9.  The brute-force scan is VALU-issue bound (4096 x 1024 pairs per cloud at the finest level, ~21 instructions per pair).
// Candidates are staged in PAIRS ({x0,x1,y0,y1 | z0,z1,-,-}: 32-byte LDS records, every lane reads the same record: broadcast) and the
// loop is unrolled over 4 records: 89.8 -> 80.2 us at (32, 4096, 1024).  Measured and rejected on gfx950 (tools/probes/tnn_time.py):
// v_pk_add/mul_f32 for the two distances (no gain: the packed fp32 ops issue at half rate) and v_med3_f32 for the value updates
// (101 us: slower than the compare/select chain it replaces).
template <bool WEIGHTS>
__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m, const float *__restrict__ unknown_all,
                                                         const float *__restrict__ known_all, float *__restrict__ dist2_all,
                                                         int *__restrict__ idx_all)
{
    __shared__ float4 s[TILE];                       // TILE / 2 pair records of two float4
    const int b = blockIdx.y, tid = threadIdx.x;
    const int pt = blockIdx.x * 256 + tid;
    const float *known = known_all + (size_t)b * m * 3;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (pt < n) {
        const float *u = unknown_all + ((size_t)b * n + pt) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    const float inf = __uint_as_float(0x7F800000u);
    float b1 = inf, b2 = inf, b3 = inf;  // (double)1e40 in the reference (:149): same admits, prints as +inf
    int i1 = 0, i2 = 0, i3 = 0;
    for (int base = 0; base < m; base += TILE) {
        const int cnt = min(TILE, m - base);
        const int pairs = (cnt + 1) >> 1;
        __syncthreads();
        for (int p = tid; p < pairs; p += 256) {     // a missing second candidate sits at +inf: its distance is +inf and never admitted
            const float *a = known + (size_t)(base + 2 * p) * 3;
            const bool two = 2 * p + 1 < cnt;
            s[2 * p] = make_float4(a[0], two ? a[3] : inf, a[1], two ? a[4] : inf);
            s[2 * p + 1] = make_float4(a[2], two ? a[5] : inf, 0.f, 0.f);
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < pairs; ++p) {
            const float4 q0 = s[2 * p], q1 = s[2 * p + 1];
            const float d0 = (ux - q0.x) * (ux - q0.x) + (uy - q0.z) * (uy - q0.z) + (uz - q1.x) * (uz - q1.x);  // :155
            const float d1 = (ux - q0.y) * (ux - q0.y) + (uy - q0.w) * (uy - q0.w) + (uz - q1.y) * (uz - q1.y);
            const int gi = base + 2 * p;
            top3_insert(d0, gi, b1, b2, b3, i1, i2, i3);
            top3_insert(d1, gi + 1, b1, b2, b3, i1, i2, i3);
        }
    }
    if (pt < n) {
        float *od = dist2_all + ((size_t)b * n + pt) * 3;
        int *oi = idx_all + ((size_t)b * n + pt) * 3;
        if (WEIGHTS) {  // patch_aug_net.py:350-353: d = sqrt(d2); r = 1/(d + 1e-8); w = r / ((r0 + r1) + r2)
            const float r1 = 1.0f / (sqrtf(b1) + 1e-8f), r2 = 1.0f / (sqrtf(b2) + 1e-8f), r3 = 1.0f / (sqrtf(b3) + 1e-8f);
            const float norm = (r1 + r2) + r3;
            od[0] = r1 / norm; od[1] = r2 / norm; od[2] = r3 / norm;
        } else {
            od[0] = b1; od[1] = b2; od[2] = b3;
        }
        oi[0] = i1; oi[1] = i2; oi[2] = i3;
    }
}

// WITH_STAT: also accumulate label_stat rows of the hits (labelstat_and_ballquery); WITH_IDX: write the ball indices.
template <bool WITH_IDX, bool WITH_STAT>
__global__ __launch_bounds__(256) void ball_kernel(int n, int m, float radius, int nsample, int nclass,
                                                     const float *__restrict__ new_xyz_all, const float *__restrict__ xyz_all,
                                                     const int *__restrict__ label_stat_all, int *__restrict__ idx_all,
                                                     int *__restrict__ new_label_stat_all)
{
    __shared__ float4 s[TILE];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int pt = blockIdx.x * 256 + tid;
    const bool live = pt < m;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    const float radius2 = radius * radius;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) {
        const float *q = new_xyz_all + ((size_t)b * m + pt) * 3;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
    int *oi = WITH_IDX && live ? idx_all + ((size_t)b * m + pt) * nsample : nullptr;
    int *ls = WITH_STAT && live ? new_label_stat_all + ((size_t)b * m + pt) * nclass : nullptr;
    const int *lab = WITH_STAT ? label_stat_all + (size_t)b * n * nclass : nullptr;
    if (WITH_STAT && live)
        for (int i = 0; i < nclass; ++i) ls[i] = 0;
    int cnt = 0, first = 0;
    bool done = !live;
    for (int base = 0; base < n; base += TILE) {
        const int c = min(TILE, n - base);
        __syncthreads();
        stage_tile(s, xyz + (size_t)base * 3, c, tid, 256);
        __syncthreads();
        if (done) continue;
        for (int k = 0; k < c; ++k) {
            const float4 p = s[k];
            const float d2 = (qx - p.x) * (qx - p.x) + (qy - p.y) * (qy - p.y) + (qz - p.z) * (qz - p.z);
            if (d2 < radius2) {
                const int gi = base + k;
                if (WITH_STAT)
                    for (int i = 0; i < nclass; ++i) ls[i] += lab[(size_t)gi * nclass + i];
                if (WITH_IDX) {
                    if (cnt == 0) first = gi;
                    oi[cnt] = gi;
                    ++cnt;
                    if (cnt >= nsample) { done = true; break; }
                }
            }
        }
    }
    if (WITH_IDX && live && cnt > 0)
        for (int l = cnt; l < nsample; ++l) oi[l] = first;  // ballquery_cuda_kernel.cu:68-72
}

__global__ __launch_bounds__(256) void featuredistribute_kernel(int n, int m, const float *__restrict__ max_xyz_all,
                                                                  const float *__restrict__ xyz_all, int *__restrict__ out_all)
{
    __shared__ float4 s[TILE];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int pt = blockIdx.x * 256 + tid;
    float x = 0.f, y = 0.f, z = 0.f;
    if (pt < m) {
        const float *p = xyz_all + ((size_t)b * m + pt) * 3;
        x = p[0]; y = p[1]; z = p[2];
    }
    float min_dist2 = 100000.f;  // featuredistribute_cuda_kernel.cu:17-18
    int min_idx = -1;
    for (int base = 0; base < n; base += TILE) {
        const int c = min(TILE, n - base);
        __syncthreads();
        stage_tile(s, max_xyz_all + ((size_t)b * n + base) * 3, c, tid, 256);
        __syncthreads();
        for (int k = 0; k < c; ++k) {
            const float4 p = s[k];
            const float d2 = (p.x - x) * (p.x - x) + (p.y - y) * (p.y - y) + (p.z - z) * (p.z - z);
            if (d2 < min_dist2) { min_idx = base + k; min_dist2 = d2; }
        }
    }
    if (pt < m) out_all[(size_t)b * m + pt] = min_idx;
}

__global__ __launch_bounds__(256) void labelstat_idx_kernel(int n, int m, int nsample, int nclass, const int *__restrict__ label_stat,
                                                              const int *__restrict__ idx, int *__restrict__ out)
{
    const int b = blockIdx.y;
    const int pt = blockIdx.x * 256 + threadIdx.x;
    if (pt >= m) return;
    const int *id = idx + ((size_t)b * m + pt) * nsample;
    const int *lab = label_stat + (size_t)b * n * nclass;
    int *o = out + ((size_t)b * m + pt) * nclass;
    for (int i = 0; i < nclass; ++i) o[i] = 0;
    for (int k = 0; k < nsample; ++k) {
        const int *row = lab + (size_t)id[k] * nclass;
        for (int i = 0; i < nclass; ++i) o[i] += row[i];
    }
}

}  // namespace

#define PA_GRID_B(b, name) PA_REQUIRE((b) <= 65535, name ": b=%d exceeds the grid limit 65535", (b))

int pa_three_nn_grid_try(int b, int n, int m, const float *unknown, const float *known, float *out, int *idx, int weights, hipStream_t st);   // three_nn_grid.hip

PA_API int pa_nearestneighbor(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && m > 0, "pa_nearestneighbor: b=%d n=%d m=%d must be positive", b, n, m);
    PA_REQUIRE(unknown && known && dist2 && idx, "pa_nearestneighbor: null pointer");
    PA_GRID_B(b, "pa_nearestneighbor");
    if (pa_three_nn_grid_try(b, n, m, unknown, known, dist2, idx, 0, (hipStream_t)stream)) { PA_CHECK_LAUNCH("pa_nearestneighbor(grid)"); return PA_OK; }
    hipLaunchKernelGGL(three_nn_kernel<false>, dim3(pa_div_up(n, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, unknown, known, dist2, idx);
    PA_CHECK_LAUNCH("pa_nearestneighbor");
    return PA_OK;
}

// 3-NN with the inverse-distance interpolation weights of the FP module fused in (patch_aug_net.py:350-353):
// weight (b, n, 3) instead of dist2.  Same neighbours, same order.
PA_API int pa_three_nn_weights(int b, int n, int m, const float *unknown, const float *known, float *weight, int *idx, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && m > 0, "pa_three_nn_weights: b=%d n=%d m=%d must be positive", b, n, m);
    PA_REQUIRE(unknown && known && weight && idx, "pa_three_nn_weights: null pointer");
    PA_GRID_B(b, "pa_three_nn_weights");
    if (pa_three_nn_grid_try(b, n, m, unknown, known, weight, idx, 1, (hipStream_t)stream)) { PA_CHECK_LAUNCH("pa_three_nn_weights(grid)"); return PA_OK; }
    hipLaunchKernelGGL(three_nn_kernel<true>, dim3(pa_div_up(n, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, unknown, known, weight, idx);
    PA_CHECK_LAUNCH("pa_three_nn_weights");
    return PA_OK;
}

PA_API int pa_ballquery(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && m > 0 && nsample > 0, "pa_ballquery: b=%d n=%d m=%d nsample=%d must be positive", b, n, m, nsample);
    PA_REQUIRE(new_xyz && xyz && idx, "pa_ballquery: null pointer");
    PA_GRID_B(b, "pa_ballquery");
    hipLaunchKernelGGL((ball_kernel<true, false>), dim3(pa_div_up(m, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, radius, nsample, 0,
                       new_xyz, xyz, (const int *)nullptr, idx, (int *)nullptr);
    PA_CHECK_LAUNCH("pa_ballquery");
    return PA_OK;
}

PA_API int pa_labelstat_and_ballquery(int b, int n, int m, float radius, int nsample, int nclass, const float *new_xyz, const float *xyz,
                                      const int *label_stat, int *idx, int *new_label_stat, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && m > 0 && nsample > 0 && nclass > 0, "pa_labelstat_and_ballquery: sizes must be positive");
    PA_REQUIRE(new_xyz && xyz && label_stat && idx && new_label_stat, "pa_labelstat_and_ballquery: null pointer");
    PA_GRID_B(b, "pa_labelstat_and_ballquery");
    hipLaunchKernelGGL((ball_kernel<true, true>), dim3(pa_div_up(m, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, radius, nsample, nclass,
                       new_xyz, xyz, label_stat, idx, new_label_stat);
    PA_CHECK_LAUNCH("pa_labelstat_and_ballquery");
    return PA_OK;
}

PA_API int pa_labelstat_ballrange(int b, int n, int m, float radius, int nclass, const float *new_xyz, const float *xyz,
                                  const int *label_stat, int *new_label_stat, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && m > 0 && nclass > 0, "pa_labelstat_ballrange: sizes must be positive");
    PA_REQUIRE(new_xyz && xyz && label_stat && new_label_stat, "pa_labelstat_ballrange: null pointer");
    PA_GRID_B(b, "pa_labelstat_ballrange");
    hipLaunchKernelGGL((ball_kernel<false, true>), dim3(pa_div_up(m, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, radius, 0, nclass,
                       new_xyz, xyz, label_stat, (int *)nullptr, new_label_stat);
    PA_CHECK_LAUNCH("pa_labelstat_ballrange");
    return PA_OK;
}

PA_API int pa_labelstat_idx(int b, int n, int m, int nsample, int nclass, const int *label_stat, const int *idx, int *new_label_stat, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && m > 0 && nsample > 0 && nclass > 0, "pa_labelstat_idx: sizes must be positive");
    PA_REQUIRE(label_stat && idx && new_label_stat, "pa_labelstat_idx: null pointer");
    PA_GRID_B(b, "pa_labelstat_idx");
    hipLaunchKernelGGL(labelstat_idx_kernel, dim3(pa_div_up(m, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, nsample, nclass, label_stat, idx, new_label_stat);
    PA_CHECK_LAUNCH("pa_labelstat_idx");
    return PA_OK;
}

PA_API int pa_featuredistribute(int b, int n, int m, const float *max_xyz, const float *xyz, int *distribute_idx, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && m > 0, "pa_featuredistribute: sizes must be positive");
    PA_REQUIRE(max_xyz && xyz && distribute_idx, "pa_featuredistribute: null pointer");
    PA_GRID_B(b, "pa_featuredistribute");
    hipLaunchKernelGGL(featuredistribute_kernel, dim3(pa_div_up(m, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, max_xyz, xyz, distribute_idx);
    PA_CHECK_LAUNCH("pa_featuredistribute");
    return PA_OK;
}
