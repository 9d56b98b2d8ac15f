// EdgeConv-style grouping of a set-abstraction level as ONE op each way (gfx950), for the autograd (training) path:
//     out[b, 0:3, j, s]   = grouped_xyz[b, :, j, s]                                              (neighbour coordinates minus the centre: coordinate-only, given)
//     out[b, 3 + c, j, s] = features[b, c, idx[b, j, s]] - features[b, c, center_idx[b, j]]      (pointops.py:559-570: gathering, grouping, subtract, cat)
// and its backward: d features[b, c, i] = sum over (j, s) with idx = i of g[b, 3 + c, j, s]  -  sum over j with center_idx = i of sum_s g[b, 3 + c, j, s].
// The module path spelled this as pa_gathering + pa_grouping + a torch subtraction + torch.cat forward and, backward, the cat's slices, a negation, a
// sum over s, a contiguous copy of the sliced gradient, two scatter kernels and autograd's add of the two feature gradients: ~14 launches per level
// and step (profiles/r05_train_step_per_replay.csv: the at::native sub / cat / neg / sum / add / copy rows).  The tensors are small (18 x 67 x 2560 and
// 18 x 259 x 320 elements at the two levels that have features): plain gathers from the L2-resident feature rows, one (cloud, channel) row of the
// gradient accumulated in LDS per workgroup.
#include "pa_common.h"

namespace {

// grid (ceil(m k / 256), 3 + c, b)
__global__ __launch_bounds__(256) void group_edge_fwd_kernel(int c, int n, int m, int k, const float *__restrict__ feat, const int *__restrict__ cidx,
                                                             const int *__restrict__ idx, const float *__restrict__ gxyz, float *__restrict__ out)
{
    const int b = blockIdx.z, ch = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x, mk = m * k;
    if (t >= mk) return;
    float v;
    if (ch < 3) {
        v = gxyz[((size_t)b * 3 + ch) * mk + t];
    } else {
        const float *row = feat + ((size_t)b * c + (ch - 3)) * n;
        v = row[idx[(size_t)b * mk + t]] - row[cidx[(size_t)b * m + t / k]];
    }
    out[((size_t)b * (3 + c) + ch) * mk + t] = v;
}

// grid (c, b), 256 threads; LDS n floats: the (cloud, channel) row of d features
__global__ __launch_bounds__(256) void group_edge_bwd_kernel(int c, int n, int m, int k, const float *__restrict__ g, const int *__restrict__ cidx,
                                                             const int *__restrict__ idx, float *__restrict__ dfeat)
{
    extern __shared__ float row[];
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, mk = m * k;
    for (int i = tid; i < n; i += 256) row[i] = 0.f;
    __syncthreads();
    const float *gr = g + ((size_t)b * (3 + c) + 3 + ch) * mk;
    const int *id = idx + (size_t)b * mk;
    for (int j = tid; j < m; j += 256) {
        float s = 0.f;
        for (int q = 0; q < k; ++q) {
            const float v = gr[j * k + q];
            atomicAdd(&row[id[j * k + q]], v);
            s += v;
        }
        atomicAdd(&row[cidx[(size_t)b * m + j]], -s);
    }
    __syncthreads();
    float *dst = dfeat + ((size_t)b * c + ch) * n;
    for (int i = tid; i < n; i += 256) dst[i] = row[i];
}

// grid (ceil(m k / 256), b): the coordinate-only part of the grouping -- neighbour coordinates and neighbour minus centre, channel-major
__global__ __launch_bounds__(256) void group_xyz_kernel(int n, int m, int k, int reps, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                        const int *__restrict__ idx, float *__restrict__ o_grouped, float *__restrict__ centred)
{
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x, mk = m * k;
    if (t >= mk) return;
    const float *p = xyz + ((size_t)b * n + idx[(size_t)b * mk + t]) * 3;
    const float *q = new_xyz + ((size_t)b * m + t / k) * 3;
    for (int d = 0; d < 3; ++d) {
        const float v = p[d], w = v - q[d];
        if (o_grouped) o_grouped[((size_t)b * 3 + d) * mk + t] = v;
        for (int r = 0; r < reps; ++r) centred[((size_t)b * 3 * reps + r * 3 + d) * mk + t] = w;
    }
}

// out[b, t] = table[b, idx[b, t]] (int32): level-local indices composed into indices of the input cloud (patch_aug_net.py:169-177)
__global__ __launch_bounds__(256) void compose_indices_kernel(int n, long count, const int *__restrict__ table, const int *__restrict__ idx, int *__restrict__ out)
{
    const int b = blockIdx.y;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t < count) out[(size_t)b * count + t] = table[(size_t)b * n + idx[(size_t)b * count + t]];
}

}  // namespace

// Coordinate-only grouping (pointops.py:559-562): o_grouped (b, 3, m, k) = xyz[b, idx[b, j, s], :] (may be NULL) and centred (b, 3 * reps, m, k) =
// that minus new_xyz[b, j, :], written reps times along the channel axis (reps = 2: the first level, whose features ARE the coordinates, so the
// module's cat([grouped_xyz, grouped_features - centre_features]) is the same three channels twice).  xyz (b, n, 3), new_xyz (b, m, 3), idx (b, m, k).
PA_API int pa_group_xyz(int b, int n, int m, int k, int reps, const float *xyz, const float *new_xyz, const int *idx, float *o_grouped, float *centred,
                        pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && b <= 65535 && n > 0 && m > 0 && k > 0 && (reps == 1 || reps == 2) && xyz && new_xyz && idx && centred, "pa_group_xyz: bad arguments");
    hipLaunchKernelGGL(group_xyz_kernel, dim3(pa_div_up((long)m * k, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, k, reps, xyz, new_xyz, idx, o_grouped, centred);
    PA_CHECK_LAUNCH("pa_group_xyz");
    return PA_OK;
}

// out (b, count) = table (b, n) indexed by idx (b, count), all int32 (torch.gather on the level-local index lists, patch_aug_net.py:169-177)
PA_API int pa_compose_indices(int b, int n, long count, const int *table, const int *idx, int *out, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && b <= 65535 && n > 0 && count > 0 && table && idx && out, "pa_compose_indices: bad arguments");
    hipLaunchKernelGGL(compose_indices_kernel, dim3(pa_div_up(count, 256), b), dim3(256), 0, (hipStream_t)stream, n, count, table, idx, out);
    PA_CHECK_LAUNCH("pa_compose_indices");
    return PA_OK;
}

// out (b, 3 + c, m, k); features (b, c, n), center_idx (b, m), idx (b, m, k), grouped_xyz (b, 3, m, k): see the header comment
PA_API int pa_group_edge_forward(int b, int c, int n, int m, int k, const float *features, const int *center_idx, const int *idx, const float *grouped_xyz,
                                 float *out, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && c > 0 && n > 0 && m > 0 && k > 0 && features && center_idx && idx && grouped_xyz && out, "pa_group_edge_forward: bad arguments");
    PA_REQUIRE(b <= 65535 && c + 3 <= 65535, "pa_group_edge_forward: grid limits");
    hipLaunchKernelGGL(group_edge_fwd_kernel, dim3(pa_div_up((long)m * k, 256), c + 3, b), dim3(256), 0, (hipStream_t)stream, c, n, m, k, features, center_idx, idx,
                       grouped_xyz, out);
    PA_CHECK_LAUNCH("pa_group_edge_forward");
    return PA_OK;
}

// dfeatures (b, c, n), WRITTEN (not accumulated); grad_out (b, 3 + c, m, k) contiguous; n <= 16384
PA_API int pa_group_edge_backward(int b, int c, int n, int m, int k, const float *grad_out, const int *center_idx, const int *idx, float *dfeatures,
                                  pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && c > 0 && n > 0 && m > 0 && k > 0 && grad_out && center_idx && idx && dfeatures, "pa_group_edge_backward: bad arguments");
    PA_REQUIRE(b <= 65535 && n <= 16384, "pa_group_edge_backward: b <= 65535, n <= 16384 (one gradient row in LDS)");
    const size_t lds = (size_t)n * sizeof(float);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&group_edge_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(group_edge_bwd_kernel, dim3(c, b), dim3(256), lds, (hipStream_t)stream, c, n, m, k, grad_out, center_idx, idx, dfeatures);
    PA_CHECK_LAUNCH("pa_group_edge_backward");
    return PA_OK;
}
