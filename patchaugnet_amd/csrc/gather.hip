// K2 gathering, K5 grouping (the roofline-graded neighbourhood gather), K8 grouping_int, K10 interpolation,
// and their scatter-add backward passes K3 / K6 / K11 / K15 (gfx950).
//
// Reference semantics (all pure data movement except K10's fixed-order 3-term sum):
//   K2  libs/pointops/src/sampling/sampling_cuda_kernel.cu:6-19    out[b,c,j]   = points[b,c,idx[b,j]]
//   K5  libs/pointops/src/grouping/grouping_cuda_kernel.cu:60-74   out[b,c,j,s] = points[b,c,idx[b,j,s]]
//   K10 libs/pointops/src/interpolation/interpolation_cuda_kernel.cu:181-195
//                                   out[b,c,j] = (w0*p[i0] + w1*p[i1]) + w2*p[i2]
//   K3/K6/K11/K15: atomicAdd scatters (:23-36, grouping :28-46, interpolation :90-114, featuredistribute :89-101)
//
// MI355X design.  The reference launches one thread per OUTPUT element: every 4-byte read is an uncoalesced HBM/L2
// access and idx is re-read once per channel.  Here a workgroup owns (batch b, a tile of CT channel rows, a slice
// of the output columns): the CT rows are staged into LDS with coalesced 16-byte loads, every lane then loads FOUR
// consecutive indices once (int4), reuses them for all CT rows, gathers from LDS and writes one coalesced 16-byte
// store per row.  HBM traffic is therefore the algorithmic minimum 4*(c*n + m*k + c*m*k) bytes per batch element
// plus an idx re-read per channel tile (c/CT times, L2-resident).  K2 is K5 with nsample = 1.
#include <stdlib.h>

#include "pa_common.h"

namespace {

constexpr int GT = 256;  // threads per workgroup
typedef float v4f __attribute__((ext_vector_type(4)));

// rows: CT channel rows of length n staged in LDS; cols: flattened (j,s) output columns [col0, col1)
template <int CT>
__global__ __launch_bounds__(GT) void group_lds_kernel(int c, int n, int mk, int cols_per_block, const float *__restrict__ points,
                                                         const int *__restrict__ idx, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];  // [CT][n]
    const int b = blockIdx.z, c0 = blockIdx.y * CT, tid = threadIdx.x;
    const int ct = min(CT, c - c0);
    const float *src = points + ((size_t)b * c + c0) * n;
    const int total = ct * n;
    // stage: rows are contiguous in memory ((b,c,n) layout), so this is one linear coalesced copy
    if ((((uintptr_t)src) & 15) == 0) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(rows);
        for (int i = tid; i < total / 4; i += GT) d4[i] = s4[i];
        for (int i = (total & ~3) + tid; i < total; i += GT) rows[i] = src[i];
    } else {
        for (int i = tid; i < total; i += GT) rows[i] = src[i];
    }
    __syncthreads();
    const int col0 = blockIdx.x * cols_per_block;
    const int col1 = min(col0 + cols_per_block, mk);
    const int *id = idx + (size_t)b * mk;
    float *o = out + ((size_t)b * c + c0) * mk;
    const bool vec = ((mk & 3) == 0) && ((((uintptr_t)id) & 15) == 0) && ((((uintptr_t)o) & 15) == 0);
    if (vec) {  // col0 is a multiple of 4 by construction
        for (int t = col0 + tid * 4; t < col1; t += GT * 4) {
            const int4 i4 = *reinterpret_cast<const int4 *>(id + t);
#pragma unroll
            for (int r = 0; r < CT; ++r) {
                if (r < ct) {
                    const float *row = rows + r * n;
                    const v4f v = {row[i4.x], row[i4.y], row[i4.z], row[i4.w]};
                    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(o + (size_t)r * mk + t));   // write-once stream
                }
            }
        }
    } else {
        for (int t = col0 + tid; t < col1; t += GT) {
            const int i = id[t];
            for (int r = 0; r < ct; ++r) o[(size_t)r * mk + t] = rows[r * n + i];
        }
    }
}

// fallback when a single row does not fit in LDS: straight from global memory, indices still loaded once per lane
template <typename T>
__global__ __launch_bounds__(GT) void group_direct_kernel(int c, int n, int mk, const T *__restrict__ points, const int *__restrict__ idx,
                                                            T *__restrict__ out)
{
    const int b = blockIdx.z, ch = blockIdx.y;
    const int t = blockIdx.x * GT + threadIdx.x;
    if (t >= mk) return;
    out[((size_t)b * c + ch) * mk + t] = points[((size_t)b * c + ch) * n + idx[(size_t)b * mk + t]];
}

template <int CT>
__global__ __launch_bounds__(GT) void interp_lds_kernel(int c, int m, int n, int cols_per_block, const float *__restrict__ points,
                                                          const int *__restrict__ idx, const float *__restrict__ weight,
                                                          float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];  // [CT][m]
    const int b = blockIdx.z, c0 = blockIdx.y * CT, tid = threadIdx.x;
    const int ct = min(CT, c - c0);
    const float *src = points + ((size_t)b * c + c0) * m;
    const int total = ct * m;
    for (int i = tid; i < total; i += GT) rows[i] = src[i];
    __syncthreads();
    const int col0 = blockIdx.x * cols_per_block;
    const int col1 = min(col0 + cols_per_block, n);
    const int *id = idx + (size_t)b * n * 3;
    const float *w = weight + (size_t)b * n * 3;
    float *o = out + ((size_t)b * c + c0) * n;
    for (int j = col0 + tid; j < col1; j += GT) {
        const int i0 = id[j * 3 + 0], i1 = id[j * 3 + 1], i2 = id[j * 3 + 2];
        const float w0 = w[j * 3 + 0], w1 = w[j * 3 + 1], w2 = w[j * 3 + 2];
#pragma unroll
        for (int r = 0; r < CT; ++r) {
            if (r < ct) {
                const float *row = rows + r * m;
                o[(size_t)r * n + j] = w0 * row[i0] + w1 * row[i1] + w2 * row[i2];  // interpolation_cuda_kernel.cu:194
            }
        }
    }
}

__global__ __launch_bounds__(GT) void interp_direct_kernel(int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
                                                             const float *__restrict__ weight, float *__restrict__ out)
{
    const int b = blockIdx.z, ch = blockIdx.y;
    const int j = blockIdx.x * GT + threadIdx.x;
    if (j >= n) return;
    const int *id = idx + ((size_t)b * n + j) * 3;
    const float *w = weight + ((size_t)b * n + j) * 3;
    const float *row = points + ((size_t)b * c + ch) * m;
    out[((size_t)b * c + ch) * n + j] = w[0] * row[id[0]] + w[1] * row[id[1]] + w[2] * row[id[2]];
}

// scatter-add: grad_points[b,ch,idx[b,t]] += grad_out[b,ch,t]   (K3 with mk = m, K6 with mk = m*nsample, K15)
__global__ __launch_bounds__(GT) void scatter_add_kernel(int c, int n, int mk, const float *__restrict__ grad_out, const int *__restrict__ idx,
                                                           float *__restrict__ grad_points)
{
    const int b = blockIdx.z, ch = blockIdx.y;
    const int t = blockIdx.x * GT + threadIdx.x;
    if (t >= mk) return;
    atomicAdd(grad_points + ((size_t)b * c + ch) * n + idx[(size_t)b * mk + t], grad_out[((size_t)b * c + ch) * mk + t]);
}

__global__ __launch_bounds__(GT) void interp_backward_kernel(int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
                                                               const float *__restrict__ weight, float *__restrict__ grad_points)
{
    const int b = blockIdx.z, ch = blockIdx.y;
    const int j = blockIdx.x * GT + threadIdx.x;
    if (j >= n) return;
    const int *id = idx + ((size_t)b * n + j) * 3;
    const float *w = weight + ((size_t)b * n + j) * 3;
    const float g = grad_out[((size_t)b * c + ch) * n + j];
    float *gp = grad_points + ((size_t)b * c + ch) * m;
    atomicAdd(gp + id[0], g * w[0]);
    atomicAdd(gp + id[1], g * w[1]);
    atomicAdd(gp + id[2], g * w[2]);
}

// Backward scatters through LDS (K3 / K6 / K11 / K15): a workgroup owns CT channel rows of one cloud's grad_points (CT * dst_len
// floats <= 64 KiB), accumulates every source element into them with LDS atomics (ds_add_f32) and adds the finished rows to
// global memory once.  The reference issues one global atomicAdd per (channel, source element)
// (interpolation_cuda_kernel.cu:198-230, grouping_cuda_kernel.cu:33-52, sampling_cuda_kernel.cu:21-36); here global traffic is
// grad_out read once + grad_points read-modify-written once.  W3 = the three weighted neighbours of the interpolation backward.
// When b * c / CT alone cannot fill the chip the source range is split over blockIdx.x and the partial rows are added atomically.
template <int CT, bool W3>
__global__ __launch_bounds__(256) void lds_scatter_kernel(int c, int dst_len, int src_len, int per_split, const float *__restrict__ grad_out,
                                                            const int *__restrict__ idx, const float *__restrict__ weight,
                                                            float *__restrict__ grad_points)
{
    extern __shared__ float acc[];
    const int b = blockIdx.z, c0 = blockIdx.y * CT, tid = threadIdx.x;
    const int ct = min(CT, c - c0);
    for (int i = tid; i < CT * dst_len; i += 256) acc[i] = 0.f;
    __syncthreads();
    const int t0 = blockIdx.x * per_split, t1 = min(src_len, t0 + per_split);
    const float *g = grad_out + ((size_t)b * c + c0) * src_len;
    for (int t = t0 + tid; t < t1; t += 256) {
        if (W3) {
            const int *id = idx + ((size_t)b * src_len + t) * 3;
            const float *w = weight + ((size_t)b * src_len + t) * 3;
            const int i0 = id[0], i1 = id[1], i2 = id[2];
            const float w0 = w[0], w1 = w[1], w2 = w[2];
#pragma unroll
            for (int r = 0; r < CT; ++r) {
                if (r < ct) {
                    const float v = g[(size_t)r * src_len + t];
                    atomicAdd(acc + r * dst_len + i0, v * w0);
                    atomicAdd(acc + r * dst_len + i1, v * w1);
                    atomicAdd(acc + r * dst_len + i2, v * w2);
                }
            }
        } else {
            const int i0 = idx[(size_t)b * src_len + t];
#pragma unroll
            for (int r = 0; r < CT; ++r)
                if (r < ct) atomicAdd(acc + r * dst_len + i0, g[(size_t)r * src_len + t]);
        }
    }
    __syncthreads();
    float *gp = grad_points + ((size_t)b * c + c0) * dst_len;
    if (gridDim.x == 1) {
        for (int i = tid; i < ct * dst_len; i += 256) gp[i] += acc[i];
    } else {
        for (int i = tid; i < ct * dst_len; i += 256)
            if (acc[i] != 0.f) atomicAdd(gp + i, acc[i]);
    }
}

template <bool W3>
bool launch_lds_scatter(int b, int c, int dst_len, int src_len, const float *grad_out, const int *idx, const float *weight, float *grad_points,
                        hipStream_t st)
{
    if ((size_t)dst_len * 4 > 64 * 1024 || getenv("PA_SCATTER_GLOBAL_ATOMICS")) return false;   // rows do not fit: caller uses the global-atomic kernel
    int ct = 16;
    while (ct > 1 && ((size_t)ct * dst_len * 4 > 64 * 1024 || ct > c)) ct >>= 1;
    const int row_blocks = pa_div_up(c, ct) * b;
    int splits = 1;
    if (row_blocks < 512) splits = max(1, min(pa_div_up(512, row_blocks), src_len / 1024));
    const int per_split = pa_div_up(pa_div_up(src_len, splits), 256) * 256;
    splits = pa_div_up(src_len, per_split);
    const dim3 grid(splits, pa_div_up(c, ct), b);
    const size_t lds = (size_t)ct * dst_len * 4;
    switch (ct) {
        case 16: hipLaunchKernelGGL((lds_scatter_kernel<16, W3>), grid, dim3(256), lds, st, c, dst_len, src_len, per_split, grad_out, idx, weight, grad_points); break;
        case 8: hipLaunchKernelGGL((lds_scatter_kernel<8, W3>), grid, dim3(256), lds, st, c, dst_len, src_len, per_split, grad_out, idx, weight, grad_points); break;
        case 4: hipLaunchKernelGGL((lds_scatter_kernel<4, W3>), grid, dim3(256), lds, st, c, dst_len, src_len, per_split, grad_out, idx, weight, grad_points); break;
        case 2: hipLaunchKernelGGL((lds_scatter_kernel<2, W3>), grid, dim3(256), lds, st, c, dst_len, src_len, per_split, grad_out, idx, weight, grad_points); break;
        default: hipLaunchKernelGGL((lds_scatter_kernel<1, W3>), grid, dim3(256), lds, st, c, dst_len, src_len, per_split, grad_out, idx, weight, grad_points); break;
    }
    return true;
}

// pick the channel tile so CT rows fit in 32 KiB of LDS (>= 4 workgroups per CU: one stages while others gather and
// stream out) and the column split so that the launch has >= ~1024 workgroups whenever the problem is big enough
struct GatherPlan { int ct; int cols_per_block; int col_blocks; size_t lds; };

GatherPlan plan(int b, int c, int row_len, int cols)
{
    GatherPlan p;
    const size_t row_bytes = (size_t)row_len * 4;
    static const int ct_max = getenv("PA_GROUP_CT") ? atoi(getenv("PA_GROUP_CT")) : 4;   // tuning knobs; defaults from the MI355X sweep in profiles/r01_grouping_sweep.txt
    static const size_t lds_cap = getenv("PA_GROUP_LDS_KB") ? (size_t)atoi(getenv("PA_GROUP_LDS_KB")) * 1024 : 32 * 1024;
    p.ct = ct_max;
    while (p.ct > 1 && (p.ct * row_bytes > lds_cap || p.ct / 2 >= c)) p.ct /= 2;
    p.lds = (size_t)p.ct * row_bytes;
    const long wg = (long)b * pa_div_up(c, p.ct);
    int split = 1;
    while (wg * split < 1024 && pa_div_up(cols, split * 2) >= 1024) split *= 2;
    p.cols_per_block = ((pa_div_up(cols, split) + 3) / 4) * 4;
    p.col_blocks = pa_div_up(cols, p.cols_per_block);
    return p;
}

template <int CT>
void launch_group(const GatherPlan &p, int b, int c, int n, int mk, const float *points, const int *idx, float *out, hipStream_t st)
{
    if (p.lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&group_lds_kernel<CT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    hipLaunchKernelGGL(group_lds_kernel<CT>, dim3(p.col_blocks, pa_div_up(c, CT), b), dim3(GT), p.lds, st, c, n, mk, p.cols_per_block, points, idx, out);
}

template <int CT>
void launch_interp(const GatherPlan &p, int b, int c, int m, int n, const float *points, const int *idx, const float *w, float *out, hipStream_t st)
{
    if (p.lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&interp_lds_kernel<CT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    hipLaunchKernelGGL(interp_lds_kernel<CT>, dim3(p.col_blocks, pa_div_up(c, CT), b), dim3(GT), p.lds, st, c, m, n, p.cols_per_block, points, idx, w, out);
}

int group_forward(const char *name, int b, int c, int n, int mk, const float *points, const int *idx, float *out, hipStream_t st)
{
    PA_REQUIRE(b > 0 && c > 0 && n > 0 && mk > 0, "%s: sizes must be positive (b=%d c=%d n=%d cols=%d)", name, b, c, n, mk);
    PA_REQUIRE(points && idx && out, "%s: null pointer", name);
    PA_REQUIRE(b <= 65535 && c <= 65535 * 16, "%s: b=%d / c=%d exceed the grid limits", name, b, c);
    if ((size_t)n * 4 > 128 * 1024) {
        PA_REQUIRE(c <= 65535, "%s: c=%d exceeds the grid limit", name, c);
        hipLaunchKernelGGL(group_direct_kernel<float>, dim3(pa_div_up(mk, GT), c, b), dim3(GT), 0, st, c, n, mk, points, idx, out);
    } else {
        GatherPlan p = plan(b, c, n, mk);
        if ((size_t)n * 4 > 64 * 1024) { p.ct = 1; p.lds = (size_t)n * 4; }
        switch (p.ct) {
            case 16: launch_group<16>(p, b, c, n, mk, points, idx, out, st); break;
            case 8: launch_group<8>(p, b, c, n, mk, points, idx, out, st); break;
            case 4: launch_group<4>(p, b, c, n, mk, points, idx, out, st); break;
            case 2: launch_group<2>(p, b, c, n, mk, points, idx, out, st); break;
            default: launch_group<1>(p, b, c, n, mk, points, idx, out, st); break;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pa_set_error("%s: launch failed: %s", name, hipGetErrorString(e)); return (int)e; }
    return PA_OK;
}

int scatter_add(const char *name, int b, int c, int n, int mk, const float *grad_out, const int *idx, float *grad_points, hipStream_t st)
{
    PA_REQUIRE(b > 0 && c > 0 && n > 0 && mk > 0, "%s: sizes must be positive", name);
    PA_REQUIRE(grad_out && idx && grad_points, "%s: null pointer", name);
    PA_REQUIRE(b <= 65535 && c <= 65535, "%s: b=%d / c=%d exceed the grid limits", name, b, c);
    if (!launch_lds_scatter<false>(b, c, n, mk, grad_out, idx, nullptr, grad_points, st))
        hipLaunchKernelGGL(scatter_add_kernel, dim3(pa_div_up(mk, GT), c, b), dim3(GT), 0, st, c, n, mk, grad_out, idx, grad_points);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { pa_set_error("%s: launch failed: %s", name, hipGetErrorString(e)); return (int)e; }
    return PA_OK;
}

}  // namespace

PA_API int pa_grouping_forward(int b, int c, int n, int m, int nsample, const float *points, const int *idx, float *out, pa_stream_t stream)
{
    PA_REQUIRE(m > 0 && nsample > 0, "pa_grouping_forward: m=%d nsample=%d must be positive", m, nsample);
    return group_forward("pa_grouping_forward", b, c, n, m * nsample, points, idx, out, (hipStream_t)stream);
}

PA_API int pa_gathering_forward(int b, int c, int n, int m, const float *points, const int *idx, float *out, pa_stream_t stream)
{
    return group_forward("pa_gathering_forward", b, c, n, m, points, idx, out, (hipStream_t)stream);
}

PA_API int pa_featuregather_forward(int b, int n, int m, int c, const float *max_feature, const int *distribute_idx, float *distribute_feature, pa_stream_t stream)
{
    return group_forward("pa_featuregather_forward", b, c, n, m, max_feature, distribute_idx, distribute_feature, (hipStream_t)stream);
}

PA_API int pa_grouping_int_forward(int b, int c, int n, int m, int nsample, const int64_t *points, const int *idx, int64_t *out, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && c > 0 && n > 0 && m > 0 && nsample > 0, "pa_grouping_int_forward: sizes must be positive");
    PA_REQUIRE(points && idx && out, "pa_grouping_int_forward: null pointer");
    PA_REQUIRE(b <= 65535 && c <= 65535, "pa_grouping_int_forward: b=%d / c=%d exceed the grid limits", b, c);
    hipLaunchKernelGGL(group_direct_kernel<int64_t>, dim3(pa_div_up(m * nsample, GT), c, b), dim3(GT), 0, (hipStream_t)stream, c, n, m * nsample, points, idx, out);
    PA_CHECK_LAUNCH("pa_grouping_int_forward");
    return PA_OK;
}

PA_API int pa_grouping_backward(int b, int c, int n, int m, int nsample, const float *grad_out, const int *idx, float *grad_points, pa_stream_t stream)
{
    PA_REQUIRE(m > 0 && nsample > 0, "pa_grouping_backward: m=%d nsample=%d must be positive", m, nsample);
    return scatter_add("pa_grouping_backward", b, c, n, m * nsample, grad_out, idx, grad_points, (hipStream_t)stream);
}

PA_API int pa_gathering_backward(int b, int c, int n, int m, const float *grad_out, const int *idx, float *grad_points, pa_stream_t stream)
{
    return scatter_add("pa_gathering_backward", b, c, n, m, grad_out, idx, grad_points, (hipStream_t)stream);
}

PA_API int pa_featuregather_backward(int b, int n, int m, int c, const float *grad_distribute_feature, const int *distribute_idx, float *grad_max_feature, pa_stream_t stream)
{
    return scatter_add("pa_featuregather_backward", b, c, n, m, grad_distribute_feature, distribute_idx, grad_max_feature, (hipStream_t)stream);
}

PA_API int pa_interpolation_forward(int b, int c, int m, int n, const float *points, const int *idx, const float *weight, float *out, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && c > 0 && m > 0 && n > 0, "pa_interpolation_forward: sizes must be positive");
    PA_REQUIRE(points && idx && weight && out, "pa_interpolation_forward: null pointer");
    PA_REQUIRE(b <= 65535 && c <= 65535, "pa_interpolation_forward: b=%d / c=%d exceed the grid limits", b, c);
    hipStream_t st = (hipStream_t)stream;
    if ((size_t)m * 4 > 64 * 1024) {
        hipLaunchKernelGGL(interp_direct_kernel, dim3(pa_div_up(n, GT), c, b), dim3(GT), 0, st, c, m, n, points, idx, weight, out);
    } else {
        GatherPlan p = plan(b, c, m, n);
        switch (p.ct) {
            case 16: launch_interp<16>(p, b, c, m, n, points, idx, weight, out, st); break;
            case 8: launch_interp<8>(p, b, c, m, n, points, idx, weight, out, st); break;
            case 4: launch_interp<4>(p, b, c, m, n, points, idx, weight, out, st); break;
            case 2: launch_interp<2>(p, b, c, m, n, points, idx, weight, out, st); break;
            default: launch_interp<1>(p, b, c, m, n, points, idx, weight, out, st); break;
        }
    }
    PA_CHECK_LAUNCH("pa_interpolation_forward");
    return PA_OK;
}

// ------------------------------------------------------------------------------------------------ interpolation backward without atomics
// grad_points[b][c][j] += sum over the (i, t) with idx[b][i][t] == j of weight[b][i][t] * grad_out[b][c][i]   (interpolation_cuda_kernel.cu:198-230).
// The scatter form pays one LDS float atomic per (channel, point, neighbour) -- 57 M of them at the finest level of the training step, and LDS
// float atomics retire about one lane per cycle (0.18 ms per launch).  The gather form inverts the index list ONCE per launch (it is the same
// for every channel): a counting sort of the 3 n references by target point gives each target its list of (source point, weight); a workgroup
// then stages four channel rows of grad_out in LDS and every thread sums its targets' lists with plain LDS reads and owns its outputs.
namespace {

// grid (b), 256 threads; LDS (m + 1) ints.  off: (b, m + 1) list starts; ent: (b, 3 n) int2 = (source point, weight bits)
__global__ __launch_bounds__(256) void interp_csr_build_kernel(int n, int m, const int *__restrict__ idx_all, const float *__restrict__ w_all,
                                                                 int *__restrict__ off_all, int2 *__restrict__ ent_all)
{
    extern __shared__ int cnt[];                 // [m + 1]
    __shared__ int part[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int *idx = idx_all + (size_t)b * n * 3;
    const float *w = w_all + (size_t)b * n * 3;
    int *off = off_all + (size_t)b * (m + 1);
    int2 *ent = ent_all + (size_t)b * n * 3;
    for (int j = tid; j <= m; j += 256) cnt[j] = 0;
    __syncthreads();
    for (int i = tid; i < 3 * n; i += 256) atomicAdd(&cnt[min(max(idx[i], 0), m - 1)], 1);
    __syncthreads();
    // exclusive scan of cnt[0 .. m]: every thread owns a run of consecutive bins
    const int per = (m + 1 + 255) / 256, lo = tid * per, hi = min(lo + per, m + 1);
    int sum = 0;
    for (int j = lo; j < hi; ++j) sum += cnt[j];
    part[tid] = sum;
    __syncthreads();
    if (tid < 64) {                              // 256 partial sums: 4 per lane of one wavefront
        int v[4], s4 = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) { v[t] = part[tid * 4 + t]; s4 += v[t]; }
        int incl = s4;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (tid >= o) incl += u; }
        int run = incl - s4;
#pragma unroll
        for (int t = 0; t < 4; ++t) { part[tid * 4 + t] = run; run += v[t]; }
    }
    __syncthreads();
    int run = part[tid];
    for (int j = lo; j < hi; ++j) { const int c = cnt[j]; cnt[j] = run; off[j] = run; run += c; }
    __syncthreads();
    for (int i = tid; i < 3 * n; i += 256) {
        const int pos = atomicAdd(&cnt[min(max(idx[i], 0), m - 1)], 1);
        ent[pos] = make_int2(i / 3, __float_as_int(w[i]));
    }
}

// grid (ceil(c / 4), b), NT threads (1024 when there are that many known points: a thread walks its point's list with dependent global reads, so
// the launch lives on the number of wavefronts in flight); LDS 4 n floats.  A list is read four entries at a time (clamped addresses, the tail
// predicated off): one round trip per four neighbours instead of four.
template <int NT>
__global__ __launch_bounds__(NT) void interp_bwd_gather_kernel(int c, int n, int m, const float *__restrict__ grad_out, long gstride, const int *__restrict__ off_all,
                                                               const int2 *__restrict__ ent_all, float *__restrict__ grad_points)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];      // [4][n]
    const int b = blockIdx.y, c0 = blockIdx.x * 4, tid = threadIdx.x;
    const int nc = min(4, c - c0);
    const float *src = grad_out + (size_t)b * gstride + (size_t)c0 * n;      // gstride: floats between clouds (c n, or more for a channel slice)
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(rows);
        for (int i = tid; i < nc * (n >> 2); i += NT) d4[i] = s4[i];
    } else {
        for (int i = tid; i < nc * n; i += NT) rows[i] = src[i];
    }
    for (int i = nc * n + tid; i < 4 * n; i += NT) rows[i] = 0.f;
    __syncthreads();
    const int *off = off_all + (size_t)b * (m + 1);
    const int2 *ent = ent_all + (size_t)b * n * 3;
    float *dst = grad_points + ((size_t)b * c + c0) * m;
    for (int j = tid; j < m; j += NT) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const int beg = off[j], end = off[j + 1];
        for (int e = beg; e < end; e += 4) {
            int2 en[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) en[q] = ent[min(e + q, end - 1)];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (e + q < end) {
                    const float w = __int_as_float(en[q].y);
                    a0 += w * rows[en[q].x];
                    a1 += w * rows[n + en[q].x];
                    a2 += w * rows[2 * n + en[q].x];
                    a3 += w * rows[3 * n + en[q].x];
                }
        }
        dst[j] += a0;                                                  // the reference accumulates into the caller's buffer
        if (nc > 1) dst[(size_t)m + j] += a1;
        if (nc > 2) dst[2 * (size_t)m + j] += a2;
        if (nc > 3) dst[3 * (size_t)m + j] += a3;
    }
}

}  // namespace

// ints of scratch pa_interpolation_backward_gather needs (list starts + (source, weight) pairs)
PA_API long pa_interpolation_backward_scratch_ints(int b, int n, int m) { return (long)b * ((long)m + 1 + 6L * n) + 4; }

// Same result as pa_interpolation_backward up to the order of the float sums (neither is ordered like the reference's atomics); needs
// n <= 4096 (four channel rows in LDS), m <= 8192; scratch: pa_interpolation_backward_scratch_ints ints, 8-byte aligned.  grad_out_batch_stride:
// floats between consecutive clouds of grad_out (0 = c n): the gradient of a channel SLICE of a wider tensor (autograd's backward of the
// torch.cat that appends the skip features) is read in place instead of through a contiguous copy (75 MB at the finest level).
PA_API int pa_interpolation_backward_gather(int b, int c, int n, int m, const float *grad_out, long grad_out_batch_stride, const int *idx, const float *weight,
                                            float *grad_points, int *scratch, pa_stream_t stream)
{
    const long gstride = grad_out_batch_stride > 0 ? grad_out_batch_stride : (long)c * n;
    PA_REQUIRE(gstride >= (long)c * n, "pa_interpolation_backward_gather: batch stride %ld < c n", gstride);
    PA_REQUIRE(b > 0 && c > 0 && m > 0 && n > 0, "pa_interpolation_backward_gather: sizes must be positive");
    PA_REQUIRE(grad_out && grad_points && scratch && ((idx == nullptr) == (weight == nullptr)), "pa_interpolation_backward_gather: null pointer");
    PA_REQUIRE(b <= 65535 && (c + 3) / 4 <= 2147483647, "pa_interpolation_backward_gather: b=%d exceeds the grid limit", b);
    if (n > 4096 || m > 8192) { pa_set_error("pa_interpolation_backward_gather: built for n <= 4096, m <= 8192 (got n=%d m=%d)", n, m); return PA_EUNSUPPORTED; }
    PA_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7) == 0, "pa_interpolation_backward_gather: scratch must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int *off = scratch;
    int2 *ent = reinterpret_cast<int2 *>(scratch + (((size_t)b * (m + 1) + 1) & ~(size_t)1));
    if (idx) hipLaunchKernelGGL(interp_csr_build_kernel, dim3(b), dim3(256), (size_t)(m + 1) * 4, st, n, m, idx, weight, off, ent);      // else: pa_interpolation_backward_lists ran
    const size_t lds = (size_t)4 * n * 4;
    if (m >= 1024) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&interp_bwd_gather_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(interp_bwd_gather_kernel<1024>, dim3((c + 3) / 4, b), dim3(1024), lds, st, c, n, m, grad_out, gstride, off, ent, grad_points);
    } else {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&interp_bwd_gather_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(interp_bwd_gather_kernel<256>, dim3((c + 3) / 4, b), dim3(256), lds, st, c, n, m, grad_out, gstride, off, ent, grad_points);
    }
    PA_CHECK_LAUNCH("pa_interpolation_backward_gather");
    return PA_OK;
}

// The inversion alone: scratch then serves any number of pa_interpolation_backward_gather(idx = weight = NULL) calls of the same (b, n, m).  It depends
// on the neighbour lists only, i.e. on coordinates: a training loop runs it with the next batch's sampling / neighbour searches, off the step's path.
PA_API int pa_interpolation_backward_lists(int b, int n, int m, const int *idx, const float *weight, int *scratch, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && m > 0 && n > 0 && idx && weight && scratch && b <= 65535, "pa_interpolation_backward_lists: bad arguments");
    if (n > 4096 || m > 8192) { pa_set_error("pa_interpolation_backward_lists: built for n <= 4096, m <= 8192 (got n=%d m=%d)", n, m); return PA_EUNSUPPORTED; }
    PA_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7) == 0, "pa_interpolation_backward_lists: scratch must be 8-byte aligned");
    int *off = scratch;
    int2 *ent = reinterpret_cast<int2 *>(scratch + (((size_t)b * (m + 1) + 1) & ~(size_t)1));
    hipLaunchKernelGGL(interp_csr_build_kernel, dim3(b), dim3(256), (size_t)(m + 1) * 4, (hipStream_t)stream, n, m, idx, weight, off, ent);
    PA_CHECK_LAUNCH("pa_interpolation_backward_lists");
    return PA_OK;
}

PA_API int pa_interpolation_backward(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight, float *grad_points, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && c > 0 && m > 0 && n > 0, "pa_interpolation_backward: sizes must be positive");
    PA_REQUIRE(grad_out && idx && weight && grad_points, "pa_interpolation_backward: null pointer");
    PA_REQUIRE(b <= 65535 && c <= 65535, "pa_interpolation_backward: b=%d / c=%d exceed the grid limits", b, c);
    if (!launch_lds_scatter<true>(b, c, m, n, grad_out, idx, weight, grad_points, (hipStream_t)stream))
        hipLaunchKernelGGL(interp_backward_kernel, dim3(pa_div_up(n, GT), c, b), dim3(GT), 0, (hipStream_t)stream, c, n, m, grad_out, idx, weight, grad_points);
    PA_CHECK_LAUNCH("pa_interpolation_backward");
    return PA_OK;
}
