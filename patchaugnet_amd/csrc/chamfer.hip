// C1/C2 -- Chamfer distance forward / backward (gfx950).
//
// Reference semantics: libs/chamfer_dist/chamfer.cu:15-145 (forward): for each p in xyz1 the
// nearest q in xyz2 with d = (qx-px)^2 + (qy-py)^2 + (qz-pz)^2 ("buf - x1" form, :42-45, fp32, no FMA); inside a
// 512-point tile the first element is taken unconditionally and later ones on strict '<'; across tiles the stored
// result is replaced on strict '>' (:137) => (d asc, index asc).  Backward: chamfer.cu:173-229.
//
// MI355X design.  The reference launches a fixed (32,16) x 512 grid; at the training shape (B = R*1024 patches of
// n = m = 20 points) 20 of every 8192 threads work.  Here work items are (batch, point) pairs packed densely
// into wavefronts: for small clouds (m <= 512, one tile) a lane owns one pair and reads the <= 6 KiB partner cloud
// straight from L1/L2; larger clouds go through 512-point LDS tiles (same tile size as the reference so the tile
// boundary rule is literally the same).
#include "pa_common.h"

namespace {

constexpr int CT_TILE = 512;

__device__ __forceinline__ void chamfer_scan(const float *__restrict__ q, int cnt, int base, float x1, float y1, float z1,
                                              float &best_dist, int &best_idx)
{
    best_dist = 0.f;
    best_idx = 0;
    for (int k = 0; k < cnt; ++k) {
        const float x2 = q[k * 3 + 0] - x1, y2 = q[k * 3 + 1] - y1, z2 = q[k * 3 + 2] - z1;
        const float d = x2 * x2 + y2 * y2 + z2 * z2;
        if (k == 0 || d < best_dist) { best_dist = d; best_idx = k + base; }
    }
}

// L1 loss: the workgroup's sum of sqrt(dist) as one double (uniform call: every thread of the 256 arrives)
__device__ __forceinline__ void sqrt_sum_to_partial(float d, double *__restrict__ out)
{
    __shared__ double wsum[4];
    double a = (double)sqrtf(d);
    for (int o = 32; o; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// one lane per (batch, point of xyz1); xyz2[batch] (<= 512 points) read from global/L2
__global__ __launch_bounds__(256) void chamfer_small_kernel(long total, int n, int m, const float *__restrict__ xyz1,
                                                              const float *__restrict__ xyz2, float *__restrict__ dist,
                                                              int *__restrict__ indexes, double *__restrict__ partial)
{
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    float bd = 0.f;
    if (g < total) {
        const long i = g / n;
        const float *p = xyz1 + g * 3;
        int bi;
        chamfer_scan(xyz2 + i * m * 3, m, 0, p[0], p[1], p[2], bd, bi);
        dist[g] = bd;
        indexes[g] = bi;
    }
    if (partial) sqrt_sum_to_partial(g < total ? bd : 0.f, partial + blockIdx.x);
}

__global__ __launch_bounds__(256) void chamfer_tiled_kernel(int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                              float *__restrict__ dist, int *__restrict__ indexes, double *__restrict__ partial)
{
    __shared__ float buf[CT_TILE * 3];
    const int i = blockIdx.y, tid = threadIdx.x;
    const int j = blockIdx.x * 256 + tid;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (j < n) {
        const float *p = xyz1 + ((size_t)i * n + j) * 3;
        x1 = p[0]; y1 = p[1]; z1 = p[2];
    }
    float res_d = 0.f;
    int res_i = 0;
    for (int k2 = 0; k2 < m; k2 += CT_TILE) {
        const int end_k = min(m, k2 + CT_TILE) - k2;
        __syncthreads();
        for (int t = tid; t < end_k * 3; t += 256) buf[t] = xyz2[((size_t)i * m + k2) * 3 + t];
        __syncthreads();
        float bd;
        int bi;
        chamfer_scan(buf, end_k, k2, x1, y1, z1, bd, bi);
        if (k2 == 0 || res_d > bd) { res_d = bd; res_i = bi; }  // chamfer.cu:137
    }
    if (j < n) {
        dist[(size_t)i * n + j] = res_d;
        indexes[(size_t)i * n + j] = res_i;
    }
    if (partial) sqrt_sum_to_partial(j < n ? res_d : 0.f, partial + (size_t)blockIdx.y * gridDim.x + blockIdx.x);
}

// one lane per (batch, point of xyz1): grad_xyz1[b,j] += g*(p-q);  grad_xyz2[b,idx] -= g*(p-q)   (chamfer.cu:173-201).
// L1: grad_dist1 is the distance itself and the incoming gradient is the L1 loss's, one scalar: d loss / d dist = gout * coef / sqrt(dist)
// with coef = 1 / (4 * count) -- the sqrt, the mean and the halving of ChamferDistanceL1 (__init__.py:79-84) differentiated in place.
template <bool L1>
__global__ __launch_bounds__(256) void chamfer_grad_kernel(long total, int n, int m, const float *__restrict__ xyz1,
                                                             const float *__restrict__ xyz2, const float *__restrict__ grad_dist1,
                                                             const int *__restrict__ idx1, float *__restrict__ grad_xyz1,
                                                             float *__restrict__ grad_xyz2, const float *__restrict__ gout, float coef)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long i = t / n;
    const float x1 = xyz1[t * 3 + 0], y1 = xyz1[t * 3 + 1], z1 = xyz1[t * 3 + 2];
    const int j2 = idx1[t];
    const float *q = xyz2 + (i * m + j2) * 3;
    const float g = (L1 ? gout[0] * coef / sqrtf(grad_dist1[t]) : grad_dist1[t]) * 2;
    atomicAdd(grad_xyz1 + t * 3 + 0, g * (x1 - q[0]));
    atomicAdd(grad_xyz1 + t * 3 + 1, g * (y1 - q[1]));
    atomicAdd(grad_xyz1 + t * 3 + 2, g * (z1 - q[2]));
    float *o = grad_xyz2 + (i * m + j2) * 3;
    atomicAdd(o + 0, -(g * (x1 - q[0])));
    atomicAdd(o + 1, -(g * (y1 - q[1])));
    atomicAdd(o + 2, -(g * (z1 - q[2])));
}

// (mean(sqrt(dist1)) + mean(sqrt(dist2))) / 2 from the search kernels' per-workgroup sums (fp64, fixed order: the value does not depend on timing)
__global__ __launch_bounds__(256) void chamfer_l1_value_kernel(int np1, int np2, double t1, double t2, const double *__restrict__ partial, float *__restrict__ loss)
{
    __shared__ double part[2][4];
    const int tid = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int t = tid; t < np1; t += 256) a += partial[t];
    for (int t = tid; t < np2; t += 256) b += partial[np1 + t];
    for (int o = 32; o; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    if ((tid & 63) == 0) {
        part[0][tid >> 6] = a;
        part[1][tid >> 6] = b;
    }
    __syncthreads();
    if (tid == 0) {
        const double sa = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]), sb = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
        loss[0] = (float)((sa / t1 + sb / t2) * 0.5);
    }
}

// number of per-workgroup partial sums one direction writes
int partial_count(int B, int n, int m) { return m <= CT_TILE ? (int)pa_div_up((long)B * n, 256) : B * (int)pa_div_up(n, 256); }

// grad_xyz1 (n1 floats) and grad_xyz2 (n2 floats) := 0 in ONE launch.  Not hipMemsetAsync: inside a captured hipGraph (train.GraphedTrainer) the two
// memset nodes were not reliably ordered in front of the scatter kernels that accumulate into these buffers -- replays then added this step's
// gradient to stale memory (|gradient| of 1e20 .. inf in the whole reconstruction branch, more often when an eager launch ran between replays;
// tools/probes/dbg_perturb2.py).  A kernel node is.
__global__ __launch_bounds__(256) void chamfer_zero2_kernel(float *__restrict__ a, long n1, float *__restrict__ b, long n2)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n1 + n2; i += (long)gridDim.x * 256) {
        if (i < n1) a[i] = 0.f;
        else b[i - n1] = 0.f;
    }
}

void zero_two(float *a, long n1, float *b, long n2, hipStream_t st)
{
    const long blocks = pa_div_up(n1 + n2, 256);
    hipLaunchKernelGGL(chamfer_zero2_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, a, n1, b, n2);
}

int one_direction(int B, int n, int m, const float *a, const float *bq, float *dist, int *idx, double *partial, hipStream_t st)
{
    if (m <= CT_TILE) {
        const long total = (long)B * n;
        hipLaunchKernelGGL(chamfer_small_kernel, dim3(pa_div_up(total, 256)), dim3(256), 0, st, total, n, m, a, bq, dist, idx, partial);
    } else {
        PA_REQUIRE(B <= 65535, "pa_chamfer_forward: B=%d exceeds the grid limit for clouds above 512 points", B);
        hipLaunchKernelGGL(chamfer_tiled_kernel, dim3(pa_div_up(n, 256), B), dim3(256), 0, st, n, m, a, bq, dist, idx, partial);
    }
    return PA_OK;
}

}  // namespace

PA_API int pa_chamfer_forward(int B, int n, int m, const float *xyz1, const float *xyz2, float *dist1, float *dist2, int *idx1, int *idx2,
                              pa_stream_t stream)
{
    PA_REQUIRE(B > 0 && n > 0 && m > 0, "pa_chamfer_forward: B=%d n=%d m=%d must be positive", B, n, m);
    PA_REQUIRE(xyz1 && xyz2 && dist1 && dist2 && idx1 && idx2, "pa_chamfer_forward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int r = one_direction(B, n, m, xyz1, xyz2, dist1, idx1, nullptr, st);
    if (r) return r;
    r = one_direction(B, m, n, xyz2, xyz1, dist2, idx2, nullptr, st);
    if (r) return r;
    PA_CHECK_LAUNCH("pa_chamfer_forward");
    return PA_OK;
}

PA_API int pa_chamfer_backward(int B, int n, int m, const float *xyz1, const float *xyz2, const int *idx1, const int *idx2,
                               const float *grad_dist1, const float *grad_dist2, float *grad_xyz1, float *grad_xyz2, pa_stream_t stream)
{
    PA_REQUIRE(B > 0 && n > 0 && m > 0, "pa_chamfer_backward: B=%d n=%d m=%d must be positive", B, n, m);
    PA_REQUIRE(xyz1 && xyz2 && idx1 && idx2 && grad_dist1 && grad_dist2 && grad_xyz1 && grad_xyz2, "pa_chamfer_backward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    zero_two(grad_xyz1, (long)B * n * 3, grad_xyz2, (long)B * m * 3, st);      // chamfer.cu:212-213 (zeros_like)
    const long t1 = (long)B * n, t2 = (long)B * m;
    hipLaunchKernelGGL(chamfer_grad_kernel<false>, dim3(pa_div_up(t1, 256)), dim3(256), 0, st, t1, n, m, xyz1, xyz2, grad_dist1, idx1, grad_xyz1, grad_xyz2,
                       (const float *)nullptr, 0.f);
    hipLaunchKernelGGL(chamfer_grad_kernel<false>, dim3(pa_div_up(t2, 256)), dim3(256), 0, st, t2, m, n, xyz2, xyz1, grad_dist2, idx2, grad_xyz2, grad_xyz1,
                       (const float *)nullptr, 0.f);
    PA_CHECK_LAUNCH("pa_chamfer_backward");
    return PA_OK;
}

// ChamferDistanceL1 (libs/chamfer_dist/__init__.py:79-84) as one call each way: pa_chamfer_forward plus the value
// (mean(sqrt(dist1)) + mean(sqrt(dist2))) / 2 into loss[0]; partial = scratch of B * (ceil(n / 256) + ceil(m / 256)) doubles ...
PA_API int pa_chamfer_l1_forward(int B, int n, int m, const float *xyz1, const float *xyz2, float *dist1, float *dist2, int *idx1, int *idx2,
                                 float *loss, double *partial, pa_stream_t stream)
{
    PA_REQUIRE(B > 0 && n > 0 && m > 0, "pa_chamfer_l1_forward: B=%d n=%d m=%d must be positive", B, n, m);
    PA_REQUIRE(xyz1 && xyz2 && dist1 && dist2 && idx1 && idx2 && loss && partial, "pa_chamfer_l1_forward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int np1 = partial_count(B, n, m), np2 = partial_count(B, m, n);
    int r = one_direction(B, n, m, xyz1, xyz2, dist1, idx1, partial, st);
    if (r) return r;
    r = one_direction(B, m, n, xyz2, xyz1, dist2, idx2, partial + np1, st);
    if (r) return r;
    hipLaunchKernelGGL(chamfer_l1_value_kernel, dim3(1), dim3(256), 0, st, np1, np2, (double)B * n, (double)B * m, partial, loss);
    PA_CHECK_LAUNCH("pa_chamfer_l1_forward");
    return PA_OK;
}

// ... and its gradient from the scalar gout[0] (device memory): the chain through the halving, the means and the square roots
// is applied per point inside the scatter kernels (a zero distance gives the same inf * 0 the unfused chain gives).
PA_API int pa_chamfer_l1_backward(int B, int n, int m, const float *xyz1, const float *xyz2, const int *idx1, const int *idx2, const float *dist1,
                                  const float *dist2, const float *gout, float *grad_xyz1, float *grad_xyz2, pa_stream_t stream)
{
    PA_REQUIRE(B > 0 && n > 0 && m > 0, "pa_chamfer_l1_backward: B=%d n=%d m=%d must be positive", B, n, m);
    PA_REQUIRE(xyz1 && xyz2 && idx1 && idx2 && dist1 && dist2 && gout && grad_xyz1 && grad_xyz2, "pa_chamfer_l1_backward: null pointer");
    hipStream_t st = (hipStream_t)stream;
    zero_two(grad_xyz1, (long)B * n * 3, grad_xyz2, (long)B * m * 3, st);
    const long t1 = (long)B * n, t2 = (long)B * m;
    hipLaunchKernelGGL(chamfer_grad_kernel<true>, dim3(pa_div_up(t1, 256)), dim3(256), 0, st, t1, n, m, xyz1, xyz2, dist1, idx1, grad_xyz1, grad_xyz2, gout,
                       0.25f / (float)t1);
    hipLaunchKernelGGL(chamfer_grad_kernel<true>, dim3(pa_div_up(t2, 256)), dim3(256), 0, st, t2, m, n, xyz2, xyz1, dist2, idx2, grad_xyz2, grad_xyz1, gout,
                       0.25f / (float)t2);
    PA_CHECK_LAUNCH("pa_chamfer_l1_backward");
    return PA_OK;
}
