// The small non-GEMM steps of the aggregation heads in TRAINING / autograd mode, forward and backward, one kernel each -- what the module path
// otherwise spells as strings of library elementwise / reduction launches (a step of the reference's training loop,
// train_place_recognition.py:255-392, issued ~400 of them, 2.2 ms of a 7.3 ms step on MI355X):
//
//   NetVLAD (place_recognition/patch_aug_net/models/loupe.py:196-222; pptnet_origin/models/loupe.py:52-71), after the assignment GEMM + BatchNorm:
//       act  = softmax over the K clusters of every point                      pa_softmax_cols        (+ per-block partial sums over the points)
//       vlad = X . act^T  (MFMA GEMM, train_gemm.hip)
//       out  = L2-normalise over the channels of (vlad - a_sum * cluster_weights2), a_sum = sum over the points of act     pa_vlad_residual_normalize
//   adaptive pyramid feature aggregator (patch_aug_net/models/loupe.py:8-66):
//       w = softmax over the columns of (max over the channels of conv(x)); out = relu(x + x * w)                        pa_afa_attention
//       BatchNorm1d over the rows of the (clouds x 256) FC output                                                        pa_bn_rows_train
//       F.normalize over each row / over dim 1                                                                           pa_l2_normalize
//
// The tensors are a few KB to a few MB: these kernels are launch-latency work, written for few launches and deterministic sums (no float atomics:
// column sums go through per-block partials added in a fixed order).
#include "pa_common.h"

namespace {

constexpr float L2_EPS = 1e-12f;          // torch.nn.functional.normalize's default eps

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// sum over the 256 threads of a workgroup, the same value returned to every thread (fixed order)
__device__ __forceinline__ float block256_sum(float v, float *red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- soft-max over the K rows of every column of a (B, K, N) tensor ------------------------------------------------------------------------------
// grid (ceil(N / 256), B) x 256, dynamic LDS 4 * K floats; part (B, nblk, K): this block's sum over its columns of act[b][k][.]
__global__ __launch_bounds__(256) void softmax_cols_kernel(int k, int n, const float *__restrict__ in, float *__restrict__ act, float *__restrict__ part)
{
    extern __shared__ float red[];                       // [4 waves][k]
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x, wave = threadIdx.x >> 6;
    const bool on = j < n;
    const float *src = in + (size_t)b * k * n + (on ? j : 0);
    float *dst = act + (size_t)b * k * n + (on ? j : 0);
    float m = -INFINITY;
    for (int c = 0; c < k; ++c) m = fmaxf(m, src[(size_t)c * n]);
    float s = 0.f;
    for (int c = 0; c < k; ++c) s += expf(src[(size_t)c * n] - m);
    for (int c = 0; c < k; ++c) {
        const float a = on ? expf(src[(size_t)c * n] - m) / s : 0.f;
        if (on) dst[(size_t)c * n] = a;
        const float w = wave_sum(a);
        if ((threadIdx.x & 63) == 0) red[wave * k + c] = w;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < k; c += 256) part[((size_t)b * gridDim.x + blockIdx.x) * k + c] = (red[c] + red[k + c]) + (red[2 * k + c] + red[3 * k + c]);
}

// The same for K = KT known at compile time (NetVLAD's 1 / 4 / 16 / 64 clusters): a thread keeps its column in registers -- one pass of loads, all in
// flight together, instead of three dependent walks.
template <int KT>
__global__ __launch_bounds__(256) void softmax_cols_reg_kernel(int n, const float *__restrict__ in, float *__restrict__ act, float *__restrict__ part)
{
    __shared__ float red[4 * KT];
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x, wave = threadIdx.x >> 6;
    const bool on = j < n;
    const size_t base = (size_t)b * KT * n + (on ? j : 0);
    float v[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) v[c] = in[base + (size_t)c * n];
    float m = v[0];
#pragma unroll
    for (int c = 1; c < KT; ++c) m = fmaxf(m, v[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < KT; ++c) { v[c] = expf(v[c] - m); s += v[c]; }
#pragma unroll
    for (int c = 0; c < KT; ++c) {
        const float a = on ? v[c] / s : 0.f;
        if (on) act[base + (size_t)c * n] = a;
        const float w = wave_sum(a);
        if ((threadIdx.x & 63) == 0) red[wave * KT + c] = w;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < KT; c += 256) part[((size_t)b * gridDim.x + blockIdx.x) * KT + c] = (red[c] + red[KT + c]) + (red[2 * KT + c] + red[3 * KT + c]);
}

// dpre = act * (g - sum_k g act), g = dact + dasum[b][k] (the gradient that reached a_sum spreads over every point of the cluster's row); dpre may
// be dact itself (a thread reads its whole column before it writes it)
__global__ __launch_bounds__(256) void softmax_cols_bwd_kernel(int k, int n, const float *__restrict__ act, const float *dact, const float *__restrict__ dasum,
                                                               float *dpre)
{
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const size_t base = (size_t)b * k * n + j;
    float dot = 0.f;
    for (int c = 0; c < k; ++c) {
        const float g = dact[base + (size_t)c * n] + (dasum ? dasum[b * k + c] : 0.f);
        dot += g * act[base + (size_t)c * n];
    }
    for (int c = 0; c < k; ++c) {
        const float g = dact[base + (size_t)c * n] + (dasum ? dasum[b * k + c] : 0.f);
        dpre[base + (size_t)c * n] = act[base + (size_t)c * n] * (g - dot);
    }
}

template <int KT>
__global__ __launch_bounds__(256) void softmax_cols_bwd_reg_kernel(int n, const float *__restrict__ act, const float *dact, const float *__restrict__ dasum, float *dpre)
{
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const size_t base = (size_t)b * KT * n + j;
    float g[KT], a[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c) {
        g[c] = dact[base + (size_t)c * n];
        a[c] = act[base + (size_t)c * n];
    }
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < KT; ++c) {
        g[c] += dasum ? dasum[b * KT + c] : 0.f;
        dot += g[c] * a[c];
    }
#pragma unroll
    for (int c = 0; c < KT; ++c) dpre[base + (size_t)c * n] = a[c] * (g[c] - dot);
}

// ---- out = normalise over the channels of (raw - a_sum * cw2) -------------------------------------------------------------------------------------
// grid (K, B) x 256: one (cloud, cluster) column of C channels per workgroup.  a_sum[b][k] = sum over the nblk partials (fixed order); nrm = the
// column's L2 norm BEFORE the eps clamp.
__global__ __launch_bounds__(256) void vlad_resnorm_kernel(int c, int k, int nblk, const float *__restrict__ raw, const float *__restrict__ part, const float *__restrict__ cw2,
                                                           float *__restrict__ out, float *__restrict__ asum, float *__restrict__ nrm)
{
    __shared__ float red[4];
    const int kk = blockIdx.x, b = blockIdx.y;
    float a = 0.f;
    for (int q = 0; q < nblk; ++q) a += part[((size_t)b * nblk + q) * k + kk];
    float ss = 0.f;
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        const float v = raw[((size_t)b * c + ch) * k + kk] - a * cw2[ch * k + kk];
        ss += v * v;
    }
    const float norm = sqrtf(block256_sum(ss, red));
    const float d = fmaxf(norm, L2_EPS);
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        const float v = raw[((size_t)b * c + ch) * k + kk] - a * cw2[ch * k + kk];
        out[((size_t)b * c + ch) * k + kk] = v / d;
    }
    if (threadIdx.x == 0) {
        asum[b * k + kk] = a;
        nrm[b * k + kk] = norm;
    }
}

// dv = (dout - out * sum_c(dout out)) / max(nrm, eps)   (nrm < eps: the clamp passes no gradient, dv = dout / eps);  draw = dv,
// dasum[b][k] = - sum_c dv cw2[c][k]
__global__ __launch_bounds__(256) void vlad_resnorm_bwd_kernel(int c, int k, const float *__restrict__ dout, const float *__restrict__ out, const float *__restrict__ nrm,
                                                               const float *__restrict__ cw2, float *__restrict__ dv, float *__restrict__ dasum)
{
    __shared__ float red[4];
    const int kk = blockIdx.x, b = blockIdx.y;
    const float norm = nrm[b * k + kk], d = fmaxf(norm, L2_EPS);
    float dot = 0.f;
    if (norm >= L2_EPS) {
        for (int ch = threadIdx.x; ch < c; ch += 256) dot += dout[((size_t)b * c + ch) * k + kk] * out[((size_t)b * c + ch) * k + kk];
        dot = block256_sum(dot, red);
    }
    float da = 0.f;
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        const size_t o = ((size_t)b * c + ch) * k + kk;
        const float g = (dout[o] - out[o] * dot) / d;
        dv[o] = g;
        da += g * cw2[ch * k + kk];
    }
    da = block256_sum(da, red);
    if (threadIdx.x == 0) dasum[b * k + kk] = -da;
}

// The same two for k <= 1024, one 1024-thread workgroup per cloud: thread t = (channel lane t / k, cluster t % k) walks the (C, K) slab with contiguous
// reads (the (K, B) grid above strides by k floats); the 1024 / k lanes of a cluster are added up in lane order through LDS.
__global__ __launch_bounds__(1024) void vlad_resnorm_slab_kernel(int c, int k, int nblk, const float *__restrict__ raw, const float *__restrict__ part,
                                                                const float *__restrict__ cw2, float *__restrict__ out, float *__restrict__ asum, float *__restrict__ nrm)
{
    __shared__ float red[1024], as[1024], ns[1024];
    const int b = blockIdx.x, lanes = 1024 / k, cl = threadIdx.x / k, kk = threadIdx.x % k;
    const bool on = cl < lanes;
    if (threadIdx.x < k) {
        float a = 0.f;
        for (int q = 0; q < nblk; ++q) a += part[((size_t)b * nblk + q) * k + threadIdx.x];
        as[threadIdx.x] = a;
    }
    __syncthreads();
    const float a = as[kk];
    float ss = 0.f;
    if (on)
        for (int ch = cl; ch < c; ch += lanes) {
            const float v = raw[((size_t)b * c + ch) * k + kk] - a * cw2[ch * k + kk];
            ss += v * v;
        }
    red[threadIdx.x] = ss;
    __syncthreads();
    if (threadIdx.x < k) {
        float t = 0.f;
        for (int l = 0; l < lanes; ++l) t += red[l * k + threadIdx.x];
        const float norm = sqrtf(t);
        ns[threadIdx.x] = norm;
        asum[b * k + threadIdx.x] = as[threadIdx.x];
        nrm[b * k + threadIdx.x] = norm;
    }
    __syncthreads();
    const float d = fmaxf(ns[kk], L2_EPS);
    if (on)
        for (int ch = cl; ch < c; ch += lanes) {
            const float v = raw[((size_t)b * c + ch) * k + kk] - a * cw2[ch * k + kk];
            out[((size_t)b * c + ch) * k + kk] = v / d;
        }
}
__global__ __launch_bounds__(1024) void vlad_resnorm_slab_bwd_kernel(int c, int k, const float *__restrict__ dout, const float *__restrict__ out, const float *__restrict__ nrm,
                                                                    const float *__restrict__ cw2, float *__restrict__ dv, float *__restrict__ dasum)
{
    __shared__ float red[1024], ds[1024];
    const int b = blockIdx.x, lanes = 1024 / k, cl = threadIdx.x / k, kk = threadIdx.x % k;
    const bool on = cl < lanes;
    const float norm = nrm[b * k + kk], d = fmaxf(norm, L2_EPS);
    float dot = 0.f;
    if (on)
        for (int ch = cl; ch < c; ch += lanes) dot += dout[((size_t)b * c + ch) * k + kk] * out[((size_t)b * c + ch) * k + kk];
    red[threadIdx.x] = dot;
    __syncthreads();
    if (threadIdx.x < k) {
        float t = 0.f;
        for (int l = 0; l < lanes; ++l) t += red[l * k + threadIdx.x];
        ds[threadIdx.x] = norm >= L2_EPS ? t : 0.f;          // threadIdx.x < k: kk == threadIdx.x, `norm` is this cluster's
    }
    __syncthreads();
    dot = ds[kk];
    float da = 0.f;
    if (on)
        for (int ch = cl; ch < c; ch += lanes) {
            const size_t o = ((size_t)b * c + ch) * k + kk;
            const float g = (dout[o] - out[o] * dot) / d;
            dv[o] = g;
            da += g * cw2[ch * k + kk];
        }
    red[threadIdx.x] = da;
    __syncthreads();
    if (threadIdx.x < k) {
        float t = 0.f;
        for (int l = 0; l < lanes; ++l) t += red[l * k + threadIdx.x];
        dasum[b * k + threadIdx.x] = -t;
    }
}

// dcw2[c][k] = - sum_b dv[b][c][k] a_sum[b][k]: one thread per (c, k), clouds in order
__global__ __launch_bounds__(256) void vlad_resnorm_dcw2_kernel(int b, int ck, int k, const float *__restrict__ dv, const float *__restrict__ asum, float *__restrict__ dcw2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ck) return;
    float s = 0.f;
    for (int q = 0; q < b; ++q) s += dv[(size_t)q * ck + i] * asum[q * k + i % k];
    dcw2[i] = -s;
}

// ---- F.normalize over the rows of an (R, F) matrix: one wavefront per row ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void l2_rows_kernel(int f, const float *__restrict__ x, float *__restrict__ out, float *__restrict__ nrm)
{
    const float *row = x + (size_t)blockIdx.x * f;
    float ss = 0.f;
    for (int i = threadIdx.x; i < f; i += 64) ss += row[i] * row[i];
    const float norm = sqrtf(wave_sum(ss)), d = fmaxf(norm, L2_EPS);
    for (int i = threadIdx.x; i < f; i += 64) out[(size_t)blockIdx.x * f + i] = row[i] / d;
    if (threadIdx.x == 0) nrm[blockIdx.x] = norm;
}
__global__ __launch_bounds__(64) void l2_rows_bwd_kernel(int f, const float *__restrict__ dout, const float *__restrict__ out, const float *__restrict__ nrm, float *__restrict__ dx)
{
    const size_t base = (size_t)blockIdx.x * f;
    const float norm = nrm[blockIdx.x], d = fmaxf(norm, L2_EPS);
    float dot = 0.f;
    if (norm >= L2_EPS) {
        for (int i = threadIdx.x; i < f; i += 64) dot += dout[base + i] * out[base + i];
        dot = wave_sum(dot);
    }
    for (int i = threadIdx.x; i < f; i += 64) dx[base + i] = (dout[base + i] - out[base + i] * dot) / d;
}

// the same over dim 1 of a (B, C, M) tensor, M > 1: grid (ceil(M / 32), B) x 256 = 32 columns x 8 channel lanes (a lane walks every 8th channel, 128
// contiguous bytes per channel row), the lanes of a column added up in lane order through LDS
__global__ __launch_bounds__(256) void l2_dim1_kernel(int c, int m, const float *__restrict__ x, float *__restrict__ out, float *__restrict__ nrm)
{
    __shared__ float red[256];
    const int col = threadIdx.x & 31, cl = threadIdx.x >> 5, j = blockIdx.x * 32 + col, b = blockIdx.y;
    const bool on = j < m;
    const size_t base = (size_t)b * c * m + (on ? j : 0);
    float ss = 0.f;
    if (on)
        for (int ch = cl; ch < c; ch += 8) ss += x[base + (size_t)ch * m] * x[base + (size_t)ch * m];
    red[threadIdx.x] = ss;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) t += red[l * 32 + col];
    const float norm = sqrtf(t), d = fmaxf(norm, L2_EPS);
    if (!on) return;
    for (int ch = cl; ch < c; ch += 8) out[base + (size_t)ch * m] = x[base + (size_t)ch * m] / d;
    if (cl == 0) nrm[(size_t)b * m + j] = norm;
}
__global__ __launch_bounds__(256) void l2_dim1_bwd_kernel(int c, int m, const float *__restrict__ dout, const float *__restrict__ out, const float *__restrict__ nrm,
                                                          float *__restrict__ dx)
{
    __shared__ float red[256];
    const int col = threadIdx.x & 31, cl = threadIdx.x >> 5, j = blockIdx.x * 32 + col, b = blockIdx.y;
    const bool on = j < m;
    const size_t base = (size_t)b * c * m + (on ? j : 0);
    const float norm = on ? nrm[(size_t)b * m + j] : 1.f, d = fmaxf(norm, L2_EPS);
    float dot = 0.f;
    if (on && norm >= L2_EPS)
        for (int ch = cl; ch < c; ch += 8) dot += dout[base + (size_t)ch * m] * out[base + (size_t)ch * m];
    red[threadIdx.x] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) dot += red[l * 32 + col];
    if (!on) return;
    for (int ch = cl; ch < c; ch += 8) dx[base + (size_t)ch * m] = (dout[base + (size_t)ch * m] - out[base + (size_t)ch * m] * dot) / d;
}

// ---- BatchNorm1d (train mode) over the rows of an (R, F) matrix: one thread per feature, rows in order ------------------------------------------------
__global__ __launch_bounds__(256) void bn_rows_kernel(int r, int f, const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                      float momentum, float *__restrict__ running_mean, float *__restrict__ running_var, long long *__restrict__ counter,
                                                      float *__restrict__ out, float *__restrict__ mean_out, float *__restrict__ rstd_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && counter) *counter += 1;
    if (i >= f) return;
    float s = 0.f;
    for (int q = 0; q < r; ++q) s += x[(size_t)q * f + i];
    const float mean = s / (float)r;
    float v = 0.f;
    for (int q = 0; q < r; ++q) {
        const float d = x[(size_t)q * f + i] - mean;
        v += d * d;
    }
    const float var = v / (float)r, rstd = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[i] : 1.f, bt = beta ? beta[i] : 0.f;
    for (int q = 0; q < r; ++q) out[(size_t)q * f + i] = (x[(size_t)q * f + i] - mean) * rstd * g + bt;
    mean_out[i] = mean;
    rstd_out[i] = rstd;
    if (running_mean) running_mean[i] = running_mean[i] * (1.f - momentum) + momentum * mean;
    if (running_var) running_var[i] = running_var[i] * (1.f - momentum) + momentum * (v / (float)(r > 1 ? r - 1 : 1));
}
// dx = gamma rstd (dy - mean_r(dy) - xhat mean_r(dy xhat)),  dgamma = sum_r dy xhat,  dbeta = sum_r dy
__global__ __launch_bounds__(256) void bn_rows_bwd_kernel(int r, int f, const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ mean,
                                                          const float *__restrict__ rstd, const float *__restrict__ gamma, float *__restrict__ dx, float *__restrict__ dgamma,
                                                          float *__restrict__ dbeta)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= f) return;
    const float m = mean[i], rs = rstd[i], g = gamma ? gamma[i] : 1.f;
    float s1 = 0.f, s2 = 0.f;
    for (int q = 0; q < r; ++q) {
        const float d = dy[(size_t)q * f + i];
        s1 += d;
        s2 += d * ((x[(size_t)q * f + i] - m) * rs);
    }
    const float a = s1 / (float)r, bq = s2 / (float)r;
    for (int q = 0; q < r; ++q) {
        const float xh = (x[(size_t)q * f + i] - m) * rs;
        dx[(size_t)q * f + i] = g * rs * (dy[(size_t)q * f + i] - a - xh * bq);
    }
    if (dgamma) dgamma[i] = s2;
    if (dbeta) dbeta[i] = s1;
}

// ---- APFA attention: w = softmax_k(max_c r[c][k]); out = relu(x + x w) ----------------------------------------------------------------------------
// grid (B) x 256, dynamic LDS k floats.  arg[b][k] = the (first) channel holding the column's maximum: where its gradient goes.
__global__ __launch_bounds__(256) void afa_attn_kernel(int c, int k, const float *__restrict__ x, const float *__restrict__ r, float *__restrict__ out, float *__restrict__ w,
                                                       int *__restrict__ arg)
{
    extern __shared__ float ws[];                        // [k] column maxima, then the soft-max weights
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float *rb = r + (size_t)b * c * k, *xb = x + (size_t)b * c * k;
    for (int kk = threadIdx.x; kk < k; kk += 256) {
        float m = -INFINITY;
        int am = 0;
        for (int ch = 0; ch < c; ++ch) {
            const float v = rb[ch * k + kk];
            if (v > m) { m = v; am = ch; }
        }
        ws[kk] = m;
        arg[b * k + kk] = am;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int kk = threadIdx.x; kk < k; kk += 256) m = fmaxf(m, ws[kk]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int kk = threadIdx.x; kk < k; kk += 256) s += expf(ws[kk] - m);
    s = block256_sum(s, red);
    for (int kk = threadIdx.x; kk < k; kk += 256) {
        const float a = expf(ws[kk] - m) / s;
        ws[kk] = a;
        w[b * k + kk] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c * k; i += 256) {
        const float v = xb[i];
        out[(size_t)b * c * k + i] = fmaxf(v + v * ws[i % k], 0.f);
    }
}
// mask = x + x w > 0;  dx = dout mask (1 + w);  dw[k] = sum_c dout mask x;  dm = w (dw - sum_k dw w);  dr[c][k] = dm[k] at c = arg[k], else 0
__global__ __launch_bounds__(256) void afa_attn_bwd_kernel(int c, int k, const float *__restrict__ dout, const float *__restrict__ x, const float *__restrict__ w,
                                                           const int *__restrict__ arg, float *__restrict__ dx, float *__restrict__ dr)
{
    extern __shared__ float ws[];                        // [k] dw, then dm
    __shared__ float red[4];
    const int b = blockIdx.x;
    const size_t base = (size_t)b * c * k;
    for (int kk = threadIdx.x; kk < k; kk += 256) {
        const float wk = w[b * k + kk];
        float s = 0.f;
        for (int ch = 0; ch < c; ++ch) {
            const float v = x[base + ch * k + kk];
            if (v + v * wk > 0.f) s += dout[base + ch * k + kk] * v;
        }
        ws[kk] = s;
    }
    __syncthreads();
    float dot = 0.f;
    for (int kk = threadIdx.x; kk < k; kk += 256) dot += ws[kk] * w[b * k + kk];
    dot = block256_sum(dot, red);
    for (int kk = threadIdx.x; kk < k; kk += 256) ws[kk] = w[b * k + kk] * (ws[kk] - dot);
    __syncthreads();
    for (int i = threadIdx.x; i < c * k; i += 256) {
        const int kk = i % k, ch = i / k;
        const float v = x[base + i], wk = w[b * k + kk];
        dx[base + i] = v + v * wk > 0.f ? dout[base + i] * (1.f + wk) : 0.f;
        if (dr) dr[base + i] = ch == arg[b * k + kk] ? ws[kk] : 0.f;
    }
}

// The same two for k <= 1024 with 1024 threads per cloud: thread t = (channel lane t / k, column t % k), contiguous reads, the lanes of a column
// combined in lane order (the one-thread-per-column walks above take 70 / 170 us at 18 x 256 x 84; these 10 / 15).
__device__ __forceinline__ float block1024_sum(float v, float *red)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i];
    return t;
}
__global__ __launch_bounds__(1024) void afa_attn_wide_kernel(int c, int k, const float *__restrict__ x, const float *__restrict__ r, float *__restrict__ out,
                                                             float *__restrict__ w, int *__restrict__ arg)
{
    __shared__ float rv[1024], ws[1024], red[16];
    __shared__ int ri[1024];
    const int b = blockIdx.x, lanes = 1024 / k, cl = threadIdx.x / k, kk = threadIdx.x % k;
    const float *rb = r + (size_t)b * c * k, *xb = x + (size_t)b * c * k;
    float m = -INFINITY;
    int am = 0x7fffffff;
    if (cl < lanes)
        for (int ch = cl; ch < c; ch += lanes) {
            const float v = rb[ch * k + kk];
            if (v > m) { m = v; am = ch; }
        }
    rv[threadIdx.x] = m;
    ri[threadIdx.x] = am;
    __syncthreads();
    float best = -INFINITY;
    if (threadIdx.x < k) {
        int bi = 0x7fffffff;
        for (int l = 0; l < lanes; ++l) {
            const float v = rv[l * k + threadIdx.x];
            const int i = ri[l * k + threadIdx.x];
            if (v > best || (v == best && i < bi)) { best = v; bi = i; }
        }
        arg[b * k + threadIdx.x] = bi == 0x7fffffff ? 0 : bi;
    }
    float mm = wave_max(best);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mm;
    __syncthreads();
    mm = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mm = fmaxf(mm, red[i]);
    const float e = threadIdx.x < k ? expf(best - mm) : 0.f;
    const float ssum = block1024_sum(e, red);
    if (threadIdx.x < k) {
        ws[threadIdx.x] = e / ssum;
        w[b * k + threadIdx.x] = e / ssum;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c * k; i += 1024) {
        const float v = xb[i];
        out[(size_t)b * c * k + i] = fmaxf(v + v * ws[i % k], 0.f);
    }
}
__global__ __launch_bounds__(1024) void afa_attn_wide_bwd_kernel(int c, int k, const float *__restrict__ dout, const float *__restrict__ x, const float *__restrict__ w,
                                                                 const int *__restrict__ arg, float *__restrict__ dx, float *__restrict__ dr)
{
    __shared__ float rv[1024], ws[1024], wsh[1024], red[16];
    __shared__ int as[1024];
    const int b = blockIdx.x, lanes = 1024 / k, cl = threadIdx.x / k, kk = threadIdx.x % k;
    const size_t base = (size_t)b * c * k;
    const float wk = w[b * k + kk];
    float s = 0.f;
    if (cl < lanes)
        for (int ch = cl; ch < c; ch += lanes) {
            const float v = x[base + ch * k + kk];
            if (v + v * wk > 0.f) s += dout[base + ch * k + kk] * v;
        }
    rv[threadIdx.x] = s;
    __syncthreads();
    float dw = 0.f;
    if (threadIdx.x < k)
        for (int l = 0; l < lanes; ++l) dw += rv[l * k + threadIdx.x];
    const float dot = block1024_sum(threadIdx.x < k ? dw * wk : 0.f, red);          // threadIdx.x < k: kk == threadIdx.x
    if (threadIdx.x < k) {
        ws[threadIdx.x] = wk * (dw - dot);
        wsh[threadIdx.x] = wk;
        as[threadIdx.x] = arg[b * k + threadIdx.x];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c * k; i += 1024) {
        const int q = i % k, ch = i / k;
        const float v = x[base + i], wq = wsh[q];
        dx[base + i] = v + v * wq > 0.f ? dout[base + i] * (1.f + wq) : 0.f;
        if (dr) dr[base + i] = ch == as[q] ? ws[q] : 0.f;
    }
}

}  // namespace

// act (B, K, N) = softmax over K of in; part (B, ceil(N / 256), K) = per-block sums over the points (pa_vlad_residual_normalize adds them up)
PA_API int pa_softmax_cols(int b, int k, int n, const float *in, float *act, float *part, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && k > 0 && n > 0 && in && act && part && b <= 65535 && k <= 4096, "pa_softmax_cols: bad arguments");
    const dim3 grid(pa_div_up(n, 256), b);
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: hipLaunchKernelGGL(softmax_cols_reg_kernel<1>, grid, dim3(256), 0, st, n, in, act, part); break;
        case 4: hipLaunchKernelGGL(softmax_cols_reg_kernel<4>, grid, dim3(256), 0, st, n, in, act, part); break;
        case 16: hipLaunchKernelGGL(softmax_cols_reg_kernel<16>, grid, dim3(256), 0, st, n, in, act, part); break;
        case 64: hipLaunchKernelGGL(softmax_cols_reg_kernel<64>, grid, dim3(256), 0, st, n, in, act, part); break;
        default: hipLaunchKernelGGL(softmax_cols_kernel, grid, dim3(256), 4 * k * sizeof(float), st, k, n, in, act, part);
    }
    PA_CHECK_LAUNCH("pa_softmax_cols");
    return PA_OK;
}

PA_API int pa_softmax_cols_backward(int b, int k, int n, const float *act, const float *dact, const float *dasum, float *dpre, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && k > 0 && n > 0 && act && dact && dpre && b <= 65535, "pa_softmax_cols_backward: bad arguments");
    const dim3 grid(pa_div_up(n, 256), b);
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: hipLaunchKernelGGL(softmax_cols_bwd_reg_kernel<1>, grid, dim3(256), 0, st, n, act, dact, dasum, dpre); break;
        case 4: hipLaunchKernelGGL(softmax_cols_bwd_reg_kernel<4>, grid, dim3(256), 0, st, n, act, dact, dasum, dpre); break;
        case 16: hipLaunchKernelGGL(softmax_cols_bwd_reg_kernel<16>, grid, dim3(256), 0, st, n, act, dact, dasum, dpre); break;
        case 64: hipLaunchKernelGGL(softmax_cols_bwd_reg_kernel<64>, grid, dim3(256), 0, st, n, act, dact, dasum, dpre); break;
        default: hipLaunchKernelGGL(softmax_cols_bwd_kernel, grid, dim3(256), 0, st, k, n, act, dact, dasum, dpre);
    }
    PA_CHECK_LAUNCH("pa_softmax_cols_backward");
    return PA_OK;
}

// out (B, C, K) = (raw - a_sum cw2) / max(||.||_C, 1e-12); a_sum (B, K) and nrm (B, K) are written for the backward pass
PA_API int pa_vlad_residual_normalize(int b, int c, int k, int nblk, const float *raw, const float *part, const float *cw2, float *out, float *asum, float *nrm,
                                      pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && c > 0 && k > 0 && nblk > 0 && raw && part && cw2 && out && asum && nrm && b <= 65535, "pa_vlad_residual_normalize: bad arguments");
    if (k <= 1024) hipLaunchKernelGGL(vlad_resnorm_slab_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, c, k, nblk, raw, part, cw2, out, asum, nrm);
    else hipLaunchKernelGGL(vlad_resnorm_kernel, dim3(k, b), dim3(256), 0, (hipStream_t)stream, c, k, nblk, raw, part, cw2, out, asum, nrm);
    PA_CHECK_LAUNCH("pa_vlad_residual_normalize");
    return PA_OK;
}

// dv (B, C, K) = gradient of (raw - a_sum cw2) (= draw), dasum (B, K), dcw2 (C, K) (NULL = not wanted)
PA_API int pa_vlad_residual_normalize_backward(int b, int c, int k, const float *dout, const float *out, const float *nrm, const float *asum, const float *cw2, float *dv,
                                               float *dasum, float *dcw2, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && c > 0 && k > 0 && dout && out && nrm && asum && cw2 && dv && dasum && b <= 65535, "pa_vlad_residual_normalize_backward: bad arguments");
    if (k <= 1024) hipLaunchKernelGGL(vlad_resnorm_slab_bwd_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, c, k, dout, out, nrm, cw2, dv, dasum);
    else hipLaunchKernelGGL(vlad_resnorm_bwd_kernel, dim3(k, b), dim3(256), 0, (hipStream_t)stream, c, k, dout, out, nrm, cw2, dv, dasum);
    if (dcw2) hipLaunchKernelGGL(vlad_resnorm_dcw2_kernel, dim3(pa_div_up(c * k, 256)), dim3(256), 0, (hipStream_t)stream, b, c * k, k, dv, asum, dcw2);
    PA_CHECK_LAUNCH("pa_vlad_residual_normalize_backward");
    return PA_OK;
}

// torch.nn.functional.normalize(x, dim = 1) of x (B, C, M) (M = 1: the rows of a (B, C) matrix); nrm (B, M) = the norms before the eps clamp
PA_API int pa_l2_normalize(int b, int c, int m, const float *x, float *out, float *nrm, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && c > 0 && m > 0 && x && out && nrm && (m == 1 || b <= 65535), "pa_l2_normalize: bad arguments");
    if (m == 1) hipLaunchKernelGGL(l2_rows_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, c, x, out, nrm);
    else hipLaunchKernelGGL(l2_dim1_kernel, dim3(pa_div_up(m, 32), b), dim3(256), 0, (hipStream_t)stream, c, m, x, out, nrm);
    PA_CHECK_LAUNCH("pa_l2_normalize");
    return PA_OK;
}

PA_API int pa_l2_normalize_backward(int b, int c, int m, const float *dout, const float *out, const float *nrm, float *dx, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && c > 0 && m > 0 && dout && out && nrm && dx && (m == 1 || b <= 65535), "pa_l2_normalize_backward: bad arguments");
    if (m == 1) hipLaunchKernelGGL(l2_rows_bwd_kernel, dim3(b), dim3(64), 0, (hipStream_t)stream, c, dout, out, nrm, dx);
    else hipLaunchKernelGGL(l2_dim1_bwd_kernel, dim3(pa_div_up(m, 32), b), dim3(256), 0, (hipStream_t)stream, c, m, dout, out, nrm, dx);
    PA_CHECK_LAUNCH("pa_l2_normalize_backward");
    return PA_OK;
}

// torch.nn.BatchNorm1d.forward in train mode on (R, F): batch statistics (biased variance) normalise, the running statistics take the unbiased
// variance with `momentum`, *num_batches_tracked += 1 (all three optional); mean / rstd (F each) are kept for the backward pass
PA_API int pa_bn_rows_train(int r, int f, const float *x, const float *gamma, const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                            long long *num_batches_tracked, float *out, float *mean, float *rstd, pa_stream_t stream)
{
    PA_REQUIRE(r > 0 && f > 0 && x && out && mean && rstd, "pa_bn_rows_train: bad arguments");
    hipLaunchKernelGGL(bn_rows_kernel, dim3(pa_div_up(f, 256)), dim3(256), 0, (hipStream_t)stream, r, f, x, gamma, beta, eps, momentum, running_mean, running_var,
                       num_batches_tracked, out, mean, rstd);
    PA_CHECK_LAUNCH("pa_bn_rows_train");
    return PA_OK;
}

PA_API int pa_bn_rows_backward(int r, int f, const float *dy, const float *x, const float *mean, const float *rstd, const float *gamma, float *dx, float *dgamma,
                               float *dbeta, pa_stream_t stream)
{
    PA_REQUIRE(r > 0 && f > 0 && dy && x && mean && rstd && dx, "pa_bn_rows_backward: bad arguments");
    hipLaunchKernelGGL(bn_rows_bwd_kernel, dim3(pa_div_up(f, 256)), dim3(256), 0, (hipStream_t)stream, r, f, dy, x, mean, rstd, gamma, dx, dgamma, dbeta);
    PA_CHECK_LAUNCH("pa_bn_rows_backward");
    return PA_OK;
}

// out (B, C, K) = relu(x + x w), w (B, K) = softmax over K of the channel maxima of r (B, C, K); arg (B, K) = the channels holding them
PA_API int pa_afa_attention(int b, int c, int k, const float *x, const float *r, float *out, float *w, int *arg, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && c > 0 && k > 0 && k <= 8192 && x && r && out && w && arg, "pa_afa_attention: bad arguments");
    if (k <= 1024) hipLaunchKernelGGL(afa_attn_wide_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, c, k, x, r, out, w, arg);
    else hipLaunchKernelGGL(afa_attn_kernel, dim3(b), dim3(256), k * sizeof(float), (hipStream_t)stream, c, k, x, r, out, w, arg);
    PA_CHECK_LAUNCH("pa_afa_attention");
    return PA_OK;
}

PA_API int pa_afa_attention_backward(int b, int c, int k, const float *dout, const float *x, const float *w, const int *arg, float *dx, float *dr, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && c > 0 && k > 0 && k <= 8192 && dout && x && w && arg && dx, "pa_afa_attention_backward: bad arguments");
    if (k <= 1024) hipLaunchKernelGGL(afa_attn_wide_bwd_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, c, k, dout, x, w, arg, dx, dr);
    else hipLaunchKernelGGL(afa_attn_bwd_kernel, dim3(b), dim3(256), k * sizeof(float), (hipStream_t)stream, c, k, dout, x, w, arg, dx, dr);
    PA_CHECK_LAUNCH("pa_afa_attention_backward");
    return PA_OK;
}
