// Descriptor losses of the training step as ONE launch: forward value AND the gradient with respect to every descriptor.
//
// Reference: losses/pointnetvlad_loss.py:9-15 (best_pos_distance), :18-45 (triplet_loss), :53-105 (quadruplet_loss), as train_one_epoch calls
// them (place_recognition/train_place_recognition.py:296-304) on the (query, positives, negatives, other negative) descriptors of a tuple:
//     d_pos[p] = |pos_p - q|^2,  positive = max_p d_pos (min_p with use_min)
//     first[n]  = max(m1 + positive - |neg_n - q|^2, 0),      second[n] = max(m2 + positive - |neg_n - other|^2, 0)      (quadruplet only)
//     per tuple: max over n (lazy) or the mean (quadruplet) / sum (triplet) over n; over the batch: the mean, or with ignore_zero_loss the sum
//     divided by the number of tuples whose term is > 1e-16 (+ 1e-16), each term separately.
// In torch that is ~25 elementwise / reduction kernels forward and ~40 backward on a few KB of data (the profile of the training step counted
// 463 such launches per step, a third of its GPU time).  Here one workgroup computes the 2 + 2 Nn distances (a wavefront per distance), one
// thread does the scalar logic, and every thread then writes its dimension of all gradients.  The gradient is that of the returned value;
// autograd scales it by the upstream gradient (patchaugnet_amd/losses.py).
#include "pa_common.h"

namespace {

constexpr int QL_MAXN = 64;       // negatives + positives per tuple the scalar stage holds in LDS

__global__ __launch_bounds__(256) void quadruplet_loss_kernel(int bsz, int P, int Nn, int D, const float *__restrict__ desc,   // (B, T = 1 + P + Nn + 1, D): q, pos.., neg.., other
                                                               float m1, float m2, int use_min, int lazy, int ignore_zero, int quad, int lazy_false_mean,
                                                               float *__restrict__ loss, float *__restrict__ grad)                // loss (1), grad (B, T, D)
{
    __shared__ float dist[2 * QL_MAXN + QL_MAXN];     // per tuple: d_pos[P], d_nq[Nn], d_no[Nn]
    __shared__ float coef[2 * QL_MAXN + 2];           // per tuple: g1[n], g2[n], G (gradient reaching `positive`), p* (as float)
    __shared__ float term[2][64];                     // per tuple b (<= 64): the two reduced terms
    __shared__ float wsc[2];                          // batch weights of the two terms
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = 1 + P + Nn + 1;
    // ---- pass 1: the reduced terms of every tuple
    for (int b = 0; b < bsz; ++b) {
        const float *base = desc + (size_t)b * T * D;
        const float *q = base, *oth = base + (size_t)(1 + P + Nn) * D;
        for (int j = wave; j < P + 2 * Nn; j += 4) {      // distance j: pos_j - q | neg_n - q | neg_n - other
            const float *a = j < P ? base + (size_t)(1 + j) * D : base + (size_t)(1 + P + (j - P) % Nn) * D;
            const float *c = (j < P + Nn) ? q : oth;
            float s = 0.f;
            for (int d = lane; d < D; d += 64) { const float t = a[d] - c[d]; s += t * t; }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
            if (lane == 0) dist[j] = s;
        }
        __syncthreads();
        if (tid == 0) {
            float positive = dist[0];
            for (int p = 1; p < P; ++p) positive = use_min ? fminf(positive, dist[p]) : fmaxf(positive, dist[p]);
            for (int t = 0; t < (quad ? 2 : 1); ++t) {
                const float m = t == 0 ? m1 : m2;
                float red = lazy ? -INFINITY : 0.f;
                for (int n = 0; n < Nn; ++n) {
                    const float h = fmaxf(m + positive - dist[P + t * Nn + n], 0.f);
                    red = lazy ? fmaxf(red, h) : red + h;
                }
                if (!lazy && lazy_false_mean) red /= (float)Nn;
                term[t][b] = red;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        float total = 0.f;
        for (int t = 0; t < (quad ? 2 : 1); ++t) {
            float s = 0.f, hard = 0.f;
            for (int b = 0; b < bsz; ++b) { s += term[t][b]; hard += term[t][b] > 1e-16f ? 1.f : 0.f; }
            const float w = ignore_zero ? 1.0f / (hard + 1e-16f) : 1.0f / (float)bsz;
            wsc[t] = w;
            total += s * w;
        }
        loss[0] = total;
    }
    __syncthreads();
    // ---- pass 2: gradients (distances recomputed per tuple: a few KB from cache)
    for (int b = 0; b < bsz; ++b) {
        const float *base = desc + (size_t)b * T * D;
        const float *q = base, *oth = base + (size_t)(1 + P + Nn) * D;
        float *gb = grad + (size_t)b * T * D;
        for (int j = wave; j < P + 2 * Nn; j += 4) {
            const float *a = j < P ? base + (size_t)(1 + j) * D : base + (size_t)(1 + P + (j - P) % Nn) * D;
            const float *c = (j < P + Nn) ? q : oth;
            float s = 0.f;
            for (int d = lane; d < D; d += 64) { const float t = a[d] - c[d]; s += t * t; }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
            if (lane == 0) dist[j] = s;
        }
        __syncthreads();
        if (tid == 0) {
            int ps = 0;
            float positive = dist[0];
            for (int p = 1; p < P; ++p)
                if (use_min ? dist[p] < positive : dist[p] > positive) { positive = dist[p]; ps = p; }     // first extremum, like torch.min / max
            float G = 0.f;
            for (int t = 0; t < 2; ++t) {
                float *g = coef + t * QL_MAXN;
                for (int n = 0; n < Nn; ++n) g[n] = 0.f;
                if (t == 1 && !quad) continue;
                const float m = t == 0 ? m1 : m2, w = wsc[t];
                if (lazy) {
                    int ns = 0;
                    float best = -INFINITY;
                    for (int n = 0; n < Nn; ++n) { const float h = fmaxf(m + positive - dist[P + t * Nn + n], 0.f); if (h > best) { best = h; ns = n; } }
                    if (best > 0.f) { g[ns] = w; G += w; }
                } else {
                    const float wn = lazy_false_mean ? w / (float)Nn : w;
                    for (int n = 0; n < Nn; ++n)
                        if (m + positive - dist[P + t * Nn + n] > 0.f) { g[n] = wn; G += wn; }
                }
            }
            coef[2 * QL_MAXN] = G;
            coef[2 * QL_MAXN + 1] = (float)ps;
        }
        __syncthreads();
        const float G = coef[2 * QL_MAXN];
        const int ps = (int)coef[2 * QL_MAXN + 1];
        for (int d = tid; d < D; d += 256) {
            const float qd = q[d], od = oth[d];
            const float dp = base[(size_t)(1 + ps) * D + d] - qd;           // pos* - q
            float gq = -2.f * G * dp, go = 0.f;
            for (int p = 0; p < P; ++p) gb[(size_t)(1 + p) * D + d] = p == ps ? 2.f * G * dp : 0.f;
            for (int n = 0; n < Nn; ++n) {
                const float nd = base[(size_t)(1 + P + n) * D + d];
                const float g1 = coef[n], g2 = coef[QL_MAXN + n];
                // first = m1 + positive - |neg - q|^2: d/dneg = -2 (neg - q), d/dq = +2 (neg - q); second likewise with `other`
                gb[(size_t)(1 + P + n) * D + d] = -2.f * (g1 * (nd - qd) + g2 * (nd - od));
                gq += 2.f * g1 * (nd - qd);
                go += 2.f * g2 * (nd - od);
            }
            gb[d] = gq;
            gb[(size_t)(1 + P + Nn) * D + d] = go;
        }
        __syncthreads();
    }
}

}  // namespace

// desc (b, 1 + p + nn + 1, d) fp32: per tuple the query, the p positives, the nn negatives, the other negative.  quad != 0: quadruplet_loss
// (mean over the negatives when not lazy), quad == 0: triplet_loss (sum over the negatives, m2 / `other` unused).  loss (1 float) = the value,
// grad (same shape as desc) = its gradient.  b <= 64, p, nn <= 64.
PA_API int pa_quadruplet_loss(int b, int p, int nn, int d, const float *desc, float m1, float m2, int use_min, int lazy, int ignore_zero, int quad,
                              float *loss, float *grad, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && b <= 64 && p > 0 && p <= QL_MAXN && nn > 0 && nn <= QL_MAXN && d > 0 && desc && loss && grad, "pa_quadruplet_loss: bad arguments (b, p, nn <= 64)");
    hipLaunchKernelGGL(quadruplet_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, b, p, nn, d, desc, m1, m2, use_min, lazy, ignore_zero, quad, quad ? 1 : 0,
                       loss, grad);
    PA_CHECK_LAUNCH("pa_quadruplet_loss");
    return PA_OK;
}
