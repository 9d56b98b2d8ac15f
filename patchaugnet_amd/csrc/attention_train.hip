// Grouped self-attention of PPT-Net in TRAINING / autograd mode: the soft-max + column re-normalisation between the two batched GEMMs, forward and
// backward (place_recognition/pptnet_origin/models/pptnet.py:261-282; twin: patch_aug_net/models/loupe.py:69-114):
//
//     energy = Y^T Y                                  (N x N per cloud; the sum over the gp groups of per-group Grams IS the full Gram, :273-275)
//     P      = softmax(energy, dim = -1)              (row soft-max, :276)
//     A      = P / (1e-9 + P.sum(dim = 1))            (every COLUMN divided by its sum over the rows, :277)
//     x_r    = x_v @ A                                (:278)
//
// The evaluation engine never builds the N x N matrices (csrc/attention.hip recomputes Y^T Y tiles in two MFMA passes).  Autograd needs A for
// the backward GEMMs (dx_v = dx_r A^T, dA = x_v^T dx_r), so the training path keeps ONE N x N matrix per cloud (the reference keeps the
// (B, gp, N, N) energy, the summed energy, P and A: 4 + gp of them) and these kernels turn the energy into A in place and dA into dEnergy in
// place; the GEMMs around them are the MFMA kernels of train_gemm.hip (patchaugnet_amd/train_ops.py: sa_attention_train).
//
//   backward:  c_j = 1e-9 + sum_i P_ij,  A_ij = P_ij / c_j
//              dP_ij = (dA_ij - s_j) / c_j,            s_j = sum_i dA_ij A_ij            (quotient rule down every column)
//              dE_ij = P_ij (dP_ij - r_i),             r_i = sum_j dP_ij P_ij            (soft-max along every row),   P_ij = A_ij c_j
//
// Column reductions are two-step and deterministic: per row-chunk partial sums, then a fixed-order sum (no float atomics).
#include "pa_common.h"

namespace {

constexpr int AT_CHUNK = 64;       // rows per partial column sum

__device__ __forceinline__ float block_max(float v, float *red)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_sum(float v, float *red)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// grid (N, B) x 256: row soft-max in place
__global__ __launch_bounds__(256) void at_row_softmax_kernel(int n, float *__restrict__ e)
{
    __shared__ float red[4];
    float *row = e + ((size_t)blockIdx.y * n + blockIdx.x) * n;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < n; j += 256) m = fmaxf(m, row[j]);
    m = block_max(m, red);
    float s = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) { const float v = expf(row[j] - m); row[j] = v; s += v; }
    s = block_sum(s, red);
    const float inv = 1.0f / s;
    for (int j = threadIdx.x; j < n; j += 256) row[j] *= inv;
}

// grid (ceil(N / 256), chunks, B) x 256: part[b][chunk][j] = sum over the chunk's rows of a[i][j] (* w[i][j] when w != null)
__global__ __launch_bounds__(256) void at_col_partial_kernel(int n, const float *__restrict__ a, const float *__restrict__ w, float *__restrict__ part)
{
    const int j = blockIdx.x * 256 + threadIdx.x, chunk = blockIdx.y, b = blockIdx.z, chunks = gridDim.y;
    if (j >= n) return;
    const int i0 = chunk * AT_CHUNK, i1 = min(i0 + AT_CHUNK, n);
    const float *pa = a + ((size_t)b * n + i0) * n + j;
    const float *pw = w ? w + ((size_t)b * n + i0) * n + j : nullptr;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = i0;
    for (; i + 3 < i1; i += 4) {
        const size_t o = (size_t)(i - i0) * n;
        s0 += pa[o] * (pw ? pw[o] : 1.f);
        s1 += pa[o + n] * (pw ? pw[o + n] : 1.f);
        s2 += pa[o + 2 * (size_t)n] * (pw ? pw[o + 2 * (size_t)n] : 1.f);
        s3 += pa[o + 3 * (size_t)n] * (pw ? pw[o + 3 * (size_t)n] : 1.f);
    }
    for (; i < i1; ++i) s0 += pa[(size_t)(i - i0) * n] * (pw ? pw[(size_t)(i - i0) * n] : 1.f);
    part[((size_t)b * chunks + chunk) * n + j] = (s0 + s1) + (s2 + s3);
}

// grid (ceil(N / 256), B): out[b][j] = add + sum over chunks (in order) of part[b][chunk][j]
__global__ __launch_bounds__(256) void at_col_final_kernel(int n, int chunks, const float *__restrict__ part, float add, float *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (j >= n) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += part[((size_t)b * chunks + c) * n + j];
    out[(size_t)b * n + j] = add + s;
}

// grid (N, B) x 256: row[j] /= c[j]
__global__ __launch_bounds__(256) void at_col_scale_kernel(int n, float *__restrict__ p, const float *__restrict__ c)
{
    float *row = p + ((size_t)blockIdx.y * n + blockIdx.x) * n;
    const float *cb = c + (size_t)blockIdx.y * n;
    for (int j = threadIdx.x; j < n; j += 256) row[j] = row[j] / cb[j];
}

// grid (N, B) x 256: dA row -> dEnergy row in place (see the header)
__global__ __launch_bounds__(256) void at_bwd_row_kernel(int n, const float *__restrict__ a, const float *__restrict__ c, const float *__restrict__ s, float *__restrict__ g)
{
    __shared__ float red[4];
    const size_t ro = ((size_t)blockIdx.y * n + blockIdx.x) * n;
    const float *ar = a + ro, *cb = c + (size_t)blockIdx.y * n, *sb = s + (size_t)blockIdx.y * n;
    float *gr = g + ro;
    float r = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) {
        const float cj = cb[j], dp = (gr[j] - sb[j]) / cj, p = ar[j] * cj;
        gr[j] = dp;                     // dP, finished below
        r += dp * p;
    }
    r = block_sum(r, red);
    for (int j = threadIdx.x; j < n; j += 256) gr[j] = (ar[j] * cb[j]) * (gr[j] - r);
}

}  // namespace

PA_API long pa_attn_train_scratch_floats(int b, int n) { return (long)b * ((n + AT_CHUNK - 1) / AT_CHUNK) * n; }

// energy (b, n, n) -> A in place; colsum (b, n) = 1e-9 + column sums of the row soft-max; scratch: pa_attn_train_scratch_floats(b, n) floats
PA_API int pa_attn_softmax_renorm(int b, int n, float *energy, float *colsum, float *scratch, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && b <= 65535 && n <= 65535 && energy && colsum && scratch, "pa_attn_softmax_renorm: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = (n + AT_CHUNK - 1) / AT_CHUNK;
    hipLaunchKernelGGL(at_row_softmax_kernel, dim3(n, b), dim3(256), 0, st, n, energy);
    hipLaunchKernelGGL(at_col_partial_kernel, dim3(pa_div_up(n, 256), chunks, b), dim3(256), 0, st, n, energy, nullptr, scratch);
    hipLaunchKernelGGL(at_col_final_kernel, dim3(pa_div_up(n, 256), b), dim3(256), 0, st, n, chunks, scratch, 1e-9f, colsum);
    hipLaunchKernelGGL(at_col_scale_kernel, dim3(n, b), dim3(256), 0, st, n, energy, colsum);
    PA_CHECK_LAUNCH("pa_attn_softmax_renorm");
    return PA_OK;
}

// grad (b, n, n): dL/dA on entry, dL/dEnergy on return; attn = A and colsum from the forward call; scratch as above + b*n floats (s_j)
PA_API int pa_attn_softmax_renorm_backward(int b, int n, const float *attn, const float *colsum, float *grad, float *scratch, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && b <= 65535 && n <= 65535 && attn && colsum && grad && scratch, "pa_attn_softmax_renorm_backward: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int chunks = (n + AT_CHUNK - 1) / AT_CHUNK;
    float *s = scratch + (size_t)b * chunks * n;
    hipLaunchKernelGGL(at_col_partial_kernel, dim3(pa_div_up(n, 256), chunks, b), dim3(256), 0, st, n, grad, attn, scratch);
    hipLaunchKernelGGL(at_col_final_kernel, dim3(pa_div_up(n, 256), b), dim3(256), 0, st, n, chunks, scratch, 0.f, s);
    hipLaunchKernelGGL(at_bwd_row_kernel, dim3(n, b), dim3(256), 0, st, n, attn, colsum, s, grad);
    PA_CHECK_LAUNCH("pa_attn_softmax_renorm_backward");
    return PA_OK;
}
