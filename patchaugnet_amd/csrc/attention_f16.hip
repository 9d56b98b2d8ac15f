// fp16-operand form of the grouped self-attention (BASELINE.json configs[4], "fp16 MFMA path"; opt-in with model.mlp_dtype = "f16").
//
// Same function and the same two-pass plan as attention.hip (SA_Layer.forward, place_recognition/pptnet_origin/models/pptnet.py:261-282):
//     energy = Y^T Y;  attn = softmax(energy, dim=-1);  attn /= 1e-9 + attn.sum(dim=1);  x_r = V attn;  d = x - x_r
// but both contractions run on v_mfma_f32_16x16x32_f16 (fp32 accumulation; 16 x the rate of the fp32-operand MFMA):
//   * energy: Y rounded to fp16 -- or, SPLIT = true, carried as a (hi, lo) pair of fp16 values with the three significant products
//     hi.hi + hi.lo + lo.hi accumulated in fp32, which keeps ~21 bits of the logits (the soft-max exponentiates them, so a plain fp16
//     rounding of Y costs |e| * 2^-11 of ABSOLUTE error in the exponent);
//   * soft-max statistics, exp, the column sums and the final division in fp32;
//   * x_r = V p with V rounded to fp16 and fp32 accumulation.  The soft-max values are NOT rounded as they are: a column j that no row attends
//     to has values p_ij of 1e-8 and less, below the fp16 range, while the column re-normalisation divides by their (equally tiny) sum -- the
//     ratio is an ordinary weighted mean of V.  Pass 2 therefore keeps a running per-column maximum M_j of log2 p_ij (flash-attention style, over
//     the ROWS here), feeds p'_ij = p_ij / 2^M_j in (0, 1] to the MFMA, rescales the accumulators when M_j grows, and divides by
//     sum_i p'_ij + 1e-9 / 2^M_j at the end: the same quotient, with the largest term of every column equal to one.
// Under configs[4]'s contract (descriptor cosine >= 0.999 against the fp32 reference vectors, tests/test_gpu_f16.py); the fp32 kernel of
// attention.hip remains the parity path.
//
// Layouts.  pa_sa_attention_pack_f16 turns the linear launch's YV = [Y | V] (B, N, 2C) fp32 rows into
//     yh (B, Np, C)  fp16 rows (and yl, the residuals Y - yh, when SPLIT),        Np = N rounded up to 32, zero rows past N
//     vt (B, C, Np)  fp16, CHANNEL-major, the points of every block of 32 permuted: position 8 g + e holds point 4 g + e (e < 4) or
//                    16 + 4 g + (e - 4) (e >= 4) of the block.
// The permutation is what lets pass 2 feed the soft-max values back WITHOUT leaving registers: an energy tile's accumulator layout gives lane
// (g = l / 16, j = l % 16) the rows i = 4 g + r of a 16-row tile; two consecutive tiles are eight values = one B operand (k-slot 8 g + e) of the
// 16x16x32 MFMA, provided the A operand (V^T) enumerates its contraction index in the same order -- which the packed layout does with one
// 16-byte LDS read per fragment.
#include "pa_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float fexp16(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

template <int C>
struct Attn16Cfg {
    static constexpr int TJ = C <= 64 ? 128 : (C == 128 ? 64 : 32);   // points per LDS tile (multiple of 32)
    static constexpr int YS = C + 8;                                   // halfs per Y row: 16-byte fragment reads of 16 rows spread over all banks
    static constexpr int VS = TJ + 8;                                  // halfs per V^T row (one row per channel)
};

// grid (ceil(N / 64), B), 256 threads: wave w owns the 16 points j0 + 16 w ..; tiles of TJ points i stream through LDS.
template <int C, int PASS, bool SPLIT>
__global__ __launch_bounds__(256) void sa_attn16_kernel(int n, int np, const _Float16 *__restrict__ yh_all, const _Float16 *__restrict__ yl_all,
                                                         const _Float16 *__restrict__ vt_all, const float *__restrict__ x_all,
                                                         float *__restrict__ stats_all, float *__restrict__ d_all,
                                                         const _Float16 *__restrict__ wtp = nullptr, const float *__restrict__ bt = nullptr)
{
    // wtp != null (pass 2): the layer that follows the attention -- trans_conv + BatchNorm (folded) + ReLU + residual, pptnet.py:279-281 -- is applied to
    // the wave's 16 points HERE and d_all receives x + relu(W d + b) instead of d: one launch and one (rows x C) round trip less per level.  wtp:
    // pa_sa_attention_f16_pack_trans of the K-major folded weight (fp16, the k order this epilogue's registers have); bt: folded bias.
    using Cfg = Attn16Cfg<C>;
    constexpr int TJ = Cfg::TJ, YS = Cfg::YS, VS = Cfg::VS, KS = C / 32, CT = C / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16 *Yh = reinterpret_cast<_Float16 *>(smem_raw);                 // [TJ][YS]
    _Float16 *Yl = Yh + (SPLIT ? TJ * YS : 0);                             // [TJ][YS]   (SPLIT)
    _Float16 *Vt = Yl + TJ * YS;                                           // [C][VS]    (pass 2)
    float *Ms = reinterpret_cast<float *>(Vt + (PASS == 2 ? C * VS : 0)); // [TJ] log-sum-exp of row i   (pass 2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int b = blockIdx.y;
    const int j0 = blockIdx.x * 64 + wave * 16;
    const bool active = j0 < n;                                            // wave-uniform
    const _Float16 *yh = yh_all + (size_t)b * np * C;
    const _Float16 *yl = SPLIT ? yl_all + (size_t)b * np * C : nullptr;
    const _Float16 *vt = vt_all + (size_t)b * C * np;
    const float *stats = stats_all + (size_t)b * n * 2;

    // own points as the B operand: lane (g, j) holds Y[j0 + j][32 ks + 8 g + e], e = 0..7 (rows past n are zero rows of the packed buffer)
    half8 bh[KS], bl[SPLIT ? KS : 1];
    if (active) {
        const size_t off = (size_t)(j0 + li) * C + lg * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bh[ks] = *reinterpret_cast<const half8 *>(yh + off + ks * 32);
            if (SPLIT) bl[ks] = *reinterpret_cast<const half8 *>(yl + off + ks * 32);
        }
    }
    float m_run = -INFINITY, l_run = 0.f, s_run = 0.f;       // pass 1: running row max / sum; pass 2: m_run = M_j (log2 domain), s_run = sum of p'
    floatx4 o[PASS == 2 ? CT : 1];
#pragma unroll
    for (int ct = 0; ct < (PASS == 2 ? CT : 1); ++ct) o[ct] = (floatx4){0.f, 0.f, 0.f, 0.f};

    constexpr int YQ = C / 8;                        // 16-byte pieces per Y row
    constexpr int VQ = TJ / 8;                       // 16-byte pieces per V^T row of a tile
    // Tiles are fetched ONE TILE AHEAD into registers (the L2 round trip hides under the previous tile's MFMAs / exps: with fp16 MFMAs a tile is
    // only a few thousand cycles of work) and dropped into LDS between the two barriers.
    constexpr int NY = TJ * YQ / 256, NV = C * VQ / 256;
    static_assert(TJ * YQ % 256 == 0 && C * VQ % 256 == 0, "tile must split evenly over the workgroup");
    half8 py[NY], pl[SPLIT ? NY : 1], pv[PASS == 2 ? NV : 1];
    float pm = 0.f;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto fetch = [&](int t0) {
#pragma unroll
        for (int u = 0; u < NY; ++u) {
            const int q = tid + u * 256, r = q / YQ, part = q - r * YQ;
            const bool ok = t0 + r < np;
            py[u] = ok ? *reinterpret_cast<const half8 *>(yh + (size_t)(t0 + r) * C + part * 8) : zero8;
            if (SPLIT) pl[u] = ok ? *reinterpret_cast<const half8 *>(yl + (size_t)(t0 + r) * C + part * 8) : zero8;
        }
        if (PASS == 2) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int q = tid + u * 256, c = q / VQ, part = q - c * VQ;
                pv[u] = t0 + part * 8 < np ? *reinterpret_cast<const half8 *>(vt + (size_t)c * np + t0 + part * 8) : zero8;
            }
            if (tid < TJ) pm = t0 + tid < n ? stats[(size_t)(t0 + tid) * 2] : INFINITY;   // log-sum-exp of row i; +inf past the cloud: p = 0
        }
    };
    static_assert(TJ <= 256, "one thread per tile row for the statistics");
    fetch(0);
    for (int t0 = 0; t0 < np; t0 += TJ) {
        // ---- tile t0 .. t0 + TJ - 1: registers -> LDS, then request the next tile
#pragma unroll
        for (int u = 0; u < NY; ++u) {
            const int q = tid + u * 256, r = q / YQ, part = q - r * YQ;
            *reinterpret_cast<half8 *>(Yh + r * YS + part * 8) = py[u];
            if (SPLIT) *reinterpret_cast<half8 *>(Yl + r * YS + part * 8) = pl[u];
        }
        if (PASS == 2) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int q = tid + u * 256, c = q / VQ, part = q - c * VQ;
                *reinterpret_cast<half8 *>(Vt + c * VS + part * 8) = pv[u];
            }
            if (tid < TJ) Ms[tid] = pm;
        }
        __syncthreads();
        if (t0 + TJ < np) fetch(t0 + TJ);
        if (active) {
            const int nblk = (min(TJ, np - t0)) >> 5;                          // 32-point blocks in this tile (np is a multiple of 32)
            for (int ib = 0; ib < nblk; ++ib) {
                floatx4 acc[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    acc[h] = (floatx4){0.f, 0.f, 0.f, 0.f};
                    const _Float16 *ap = Yh + (ib * 32 + h * 16 + li) * YS + lg * 8;
                    const _Float16 *al = Yl + (ib * 32 + h * 16 + li) * YS + lg * 8;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const half8 a = *reinterpret_cast<const half8 *>(ap + ks * 32);
                        if (SPLIT) {   // small terms first
                            const half8 a2 = *reinterpret_cast<const half8 *>(al + ks * 32);
                            acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, bh[ks], acc[h], 0, 0, 0);
                            acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bl[ks], acc[h], 0, 0, 0);
                        }
                        acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh[ks], acc[h], 0, 0, 0);
                    }
                }
                // acc[h][r] = e(i = t0 + 32 ib + 16 h + 4 g + r, j = j0 + l % 16)
                if (PASS == 1) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ig = t0 + ib * 32 + h * 16 + lg * 4;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (ig + r >= n) acc[h][r] = -INFINITY;
                    }
                    const float mt = fmaxf(fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[0][2], acc[0][3])), fmaxf(fmaxf(acc[1][0], acc[1][1]), fmaxf(acc[1][2], acc[1][3])));
                    const float mn = fmaxf(m_run, mt);
                    if (mn > -INFINITY) {
                        const float s0 = (fexp16(acc[0][0] - mn) + fexp16(acc[0][1] - mn)) + (fexp16(acc[0][2] - mn) + fexp16(acc[0][3] - mn));
                        const float s1 = (fexp16(acc[1][0] - mn) + fexp16(acc[1][1] - mn)) + (fexp16(acc[1][2] - mn) + fexp16(acc[1][3] - mn));
                        l_run = l_run * fexp16(m_run - mn) + (s0 + s1);
                        m_run = mn;
                    }
                } else {
                    // t = log2 p_ij = (e_ij - lse_i) log2 e for this lane's eight rows of column j; the column maximum over the block's 32 rows
                    float t[8];
                    float tmax = -INFINITY;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 lse = *reinterpret_cast<const float4 *>(Ms + ib * 32 + h * 16 + lg * 4);
                        t[4 * h + 0] = (acc[h][0] - lse.x) * 1.4426950408889634f;
                        t[4 * h + 1] = (acc[h][1] - lse.y) * 1.4426950408889634f;
                        t[4 * h + 2] = (acc[h][2] - lse.z) * 1.4426950408889634f;
                        t[4 * h + 3] = (acc[h][3] - lse.w) * 1.4426950408889634f;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) tmax = fmaxf(tmax, t[e]);
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                    if (__any(tmax > m_run)) {                                 // wave-uniform: some column's maximum grew -> rescale what it has
                        const float mn = fmaxf(m_run, tmax);
                        const float sc = mn > -INFINITY ? __builtin_amdgcn_exp2f(m_run - mn) : 0.f;      // m_run = -inf: nothing accumulated yet
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) { o[ct][0] *= sc; o[ct][1] *= sc; o[ct][2] *= sc; o[ct][3] *= sc; }
                        s_run *= sc;
                        m_run = mn;
                    }
                    half8 pb;
                    float ps = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float pe = m_run > -INFINITY ? __builtin_amdgcn_exp2f(t[e] - m_run) : 0.f;
                        ps += pe;
                        pb[e] = (_Float16)pe;
                    }
                    s_run += ps;
                    // x_r^T[c][j] += sum over the block's 32 points, k-slot 8 g + e <-> point 4 g + e / 16 + 4 g + (e - 4): the packed V^T order
                    const _Float16 *vp = Vt + li * VS + ib * 32 + lg * 8;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const half8 a = *reinterpret_cast<const half8 *>(vp + ct * 16 * VS);
                        o[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb, o[ct], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (!active) return;

    if (PASS == 1) {
#pragma unroll
        for (int sft = 16; sft < 64; sft <<= 1) {
            const float mo = __shfl_xor(m_run, sft), lo = __shfl_xor(l_run, sft);
            const float mn = fmaxf(m_run, mo);
            l_run = mn > -INFINITY ? l_run * fexp16(m_run - mn) + lo * fexp16(mo - mn) : 0.f;
            m_run = mn;
        }
        if (lane < 16 && j0 + lane < n) {
            float *st = stats_all + ((size_t)b * n + j0 + lane) * 2;
            st[0] = m_run + __logf(l_run);                                     // log-sum-exp of the row: p_ij = exp(e_ij - st[0])
            st[1] = 1.0f / l_run;
        }
    } else {
        s_run += __shfl_xor(s_run, 16);
        s_run += __shfl_xor(s_run, 32);
        // pptnet.py:277 in the scaled domain: x_r = o 2^M / (1e-9 + s 2^M) = o / (1e-9 2^-M + s); 2^-M = +inf (a column of mass < 2^-126): x_r = 0
        const float den = 1e-9f * __builtin_amdgcn_exp2f(-m_run) + s_run;
        const bool valid = j0 + li < n;            // lanes past the cloud stay in the wave: the fused epilogue's MFMAs take their weight fragments
        const size_t row = (size_t)b * n + min(j0 + li, n - 1);
        const float *xr = x_all + row * C;
        float *dr = d_all + row * C;
        if (wtp == nullptr) {
            if (!valid) return;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {                                  // o[ct][r]: channel 16 ct + 4 g + r of point j = l % 16
                const int c = ct * 16 + lg * 4;
                const float4 xv = *reinterpret_cast<const float4 *>(xr + c);
                float4 d;
                d.x = xv.x - o[ct][0] / den;
                d.y = xv.y - o[ct][1] / den;
                d.z = xv.z - o[ct][2] / den;
                d.w = xv.w - o[ct][3] / den;
                *reinterpret_cast<float4 *>(dr + c) = d;
            }
            return;
        }
        // fused trans_conv: d (fp16) is the B operand as it sits in the registers -- k-step s takes the channel tiles 2 s and 2 s + 1, k-slot 8 g + e
        // = channel 32 s + 4 g + e (e < 4) or 32 s + 16 + 4 g + (e - 4); the packed weights enumerate the contraction in the same order.
        half8 db[C / 32];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float4 xv = *reinterpret_cast<const float4 *>(xr + ct * 16 + lg * 4);
            const int sidx = ct >> 1, eo = (ct & 1) * 4;
            db[sidx][eo + 0] = (_Float16)(xv.x - o[ct][0] / den);
            db[sidx][eo + 1] = (_Float16)(xv.y - o[ct][1] / den);
            db[sidx][eo + 2] = (_Float16)(xv.z - o[ct][2] / den);
            db[sidx][eo + 3] = (_Float16)(xv.w - o[ct][3] / den);
        }
        const half8 *wq = reinterpret_cast<const half8 *>(wtp) + lane;          // wtp[((cot * (C / 32) + s) * 64 + lane) * 8 + e]
#pragma unroll
        for (int cot = 0; cot < CT; ++cot) {
            floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sidx = 0; sidx < C / 32; ++sidx)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[(size_t)(cot * (C / 32) + sidx) * 64], db[sidx], acc, 0, 0, 0);
            const int c = cot * 16 + lg * 4;                                    // acc[r]: output channel 16 cot + 4 g + r of point j
            const float4 bias = *reinterpret_cast<const float4 *>(bt + c);
            const float4 xv = *reinterpret_cast<const float4 *>(xr + c);
            float4 y;
            y.x = xv.x + fmaxf(acc[0] + bias.x, 0.f);
            y.y = xv.y + fmaxf(acc[1] + bias.y, 0.f);
            y.z = xv.z + fmaxf(acc[2] + bias.z, 0.f);
            y.w = xv.w + fmaxf(acc[3] + bias.w, 0.f);
            if (valid) *reinterpret_cast<float4 *>(dr + c) = y;
        }
    }
}

// wtp[((cot * (c / 32) + s) * 64 + lane) * 8 + e] = (half) wt[k(s, lane / 16, e) * c + 16 cot + lane % 16],  k(s, g, e) = 32 s + 4 g + e (e < 4) or
// 32 s + 16 + 4 g + (e - 4): the A-operand fragments of the fused trans_conv epilogue (wt: K-major (c x c) fp32, BatchNorm folded)
__global__ void attn_pack_trans16_kernel(int c, const float *__restrict__ wt, _Float16 *__restrict__ wtp)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= c * c) return;
    const int e = t & 7, lane = (t >> 3) & 63, rest = t >> 9, ks = c / 32;
    const int sidx = rest % ks, cot = rest / ks, g = lane >> 4, m = lane & 15;
    const int k = e < 4 ? 32 * sidx + 4 * g + e : 32 * sidx + 16 + 4 * g + (e - 4);
    wtp[t] = (_Float16)wt[(size_t)k * c + 16 * cot + m];
}

// yv (B, N, 2C) fp32 -> yh / yl (B, Np, C) fp16 rows, vt (B, C, Np) fp16 channel-major with the 32-point block permutation; zero past N.
// grid (Np / 32, B), 256 threads: one 32-point block per workgroup, staged through LDS so that both outputs leave as 16-byte pieces.
template <int C>
__global__ __launch_bounds__(256) void attn_pack16_kernel(int n, int np, const float *__restrict__ yv_all, _Float16 *__restrict__ yh_all,
                                                           _Float16 *__restrict__ yl_all, _Float16 *__restrict__ vt_all)
{
    __shared__ _Float16 vs[C][40];                   // [channel][position in the block], 80-byte rows: 16-byte aligned pieces
    const int tid = threadIdx.x, b = blockIdx.y, i0 = blockIdx.x * 32;
    const float *yv = yv_all + (size_t)b * n * (2 * C);
    constexpr int Q = C / 4;                         // float4 pieces per half row
    for (int q = tid; q < 32 * Q; q += 256) {
        const int r = q / Q, part = q - r * Q, i = i0 + r;
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f), v = y;
        if (i < n) {
            y = *reinterpret_cast<const float4 *>(yv + (size_t)i * (2 * C) + part * 4);
            v = *reinterpret_cast<const float4 *>(yv + (size_t)i * (2 * C) + C + part * 4);
        }
        const half4 h = {(_Float16)y.x, (_Float16)y.y, (_Float16)y.z, (_Float16)y.w};
        *reinterpret_cast<half4 *>(yh_all + ((size_t)b * np + i) * C + part * 4) = h;
        if (yl_all) {
            const half4 l = {(_Float16)(y.x - (float)h[0]), (_Float16)(y.y - (float)h[1]), (_Float16)(y.z - (float)h[2]), (_Float16)(y.w - (float)h[3])};
            *reinterpret_cast<half4 *>(yl_all + ((size_t)b * np + i) * C + part * 4) = l;
        }
        const int pos = r < 16 ? 8 * (r >> 2) + (r & 3) : 8 * ((r - 16) >> 2) + 4 + (r & 3);
        vs[part * 4 + 0][pos] = (_Float16)v.x;
        vs[part * 4 + 1][pos] = (_Float16)v.y;
        vs[part * 4 + 2][pos] = (_Float16)v.z;
        vs[part * 4 + 3][pos] = (_Float16)v.w;
    }
    __syncthreads();
    for (int q = tid; q < C * 4; q += 256) {
        const int c = q >> 2, part = q & 3;
        *reinterpret_cast<half8 *>(vt_all + ((size_t)b * C + c) * np + i0 + part * 8) = *reinterpret_cast<const half8 *>(&vs[c][part * 8]);
    }
}

template <int C, bool SPLIT>
int launch_attn16(int b, int n, const float *yv, const float *x, _Float16 *scratch, float *stats, float *d, hipStream_t st, const _Float16 *wtp, const float *bt)
{
    using Cfg = Attn16Cfg<C>;
    const int np = (n + 31) & ~31;
    _Float16 *yh = scratch, *yl = SPLIT ? yh + (size_t)b * np * C : nullptr, *vt = yh + (size_t)(SPLIT ? 2 : 1) * b * np * C;
    hipLaunchKernelGGL((attn_pack16_kernel<C>), dim3(np / 32, b), dim3(256), 0, st, n, np, yv, yh, yl, vt);
    const size_t ybytes = (size_t)(SPLIT ? 2 : 1) * Cfg::TJ * Cfg::YS * 2;
    const size_t lds1 = ybytes, lds2 = ybytes + (size_t)C * Cfg::VS * 2 + Cfg::TJ * 4;
    const dim3 grid(pa_div_up(n, 64), b);
    if (lds2 > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_attn16_kernel<C, 2, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (lds1 > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_attn16_kernel<C, 1, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
    hipLaunchKernelGGL((sa_attn16_kernel<C, 1, SPLIT>), grid, dim3(256), lds1, st, n, np, yh, yl, vt, x, stats, d, (const _Float16 *)nullptr, (const float *)nullptr);
    hipLaunchKernelGGL((sa_attn16_kernel<C, 2, SPLIT>), grid, dim3(256), lds2, st, n, np, yh, yl, vt, x, stats, d, wtp, bt);
    return 0;
}

}  // namespace

// fp16 elements of scratch pa_sa_attention_f16 needs: yh [+ yl] + vt
PA_API long pa_sa_attention_f16_scratch_halfs(int b, int n, int c, int split) { return (long)(split ? 3 : 2) * b * ((n + 31) & ~31) * c; }

// d (b, n, c) = x - x_r as pa_sa_attention, both contractions on fp16 MFMA (fp32 accumulate, fp32 soft-max).  split != 0: the energy operands
// as (hi, lo) fp16 pairs (three products).  scratch: pa_sa_attention_f16_scratch_halfs fp16 elements, 16-byte aligned.  c in {64, 128, 256}.
static int attn16_dispatch(int b, int n, int c, int split, const float *yv, const float *x, void *scratch, float *stats, float *d, const _Float16 *wtp,
                           const float *bt, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && yv && x && scratch && stats && d, "pa_sa_attention_f16: bad arguments");
    PA_REQUIRE(b <= 65535 && ((uintptr_t)scratch & 15) == 0, "pa_sa_attention_f16: b=%d exceeds the grid limit or scratch is not 16-byte aligned", b);
    hipStream_t st = (hipStream_t)stream;
    _Float16 *sc = reinterpret_cast<_Float16 *>(scratch);
#define PA_ATTN16(CC) do { if (split) launch_attn16<CC, true>(b, n, yv, x, sc, stats, d, st, wtp, bt); else launch_attn16<CC, false>(b, n, yv, x, sc, stats, d, st, wtp, bt); } while (0)
    switch (c) {
        case 64: PA_ATTN16(64); break;
        case 128: PA_ATTN16(128); break;
        case 256: PA_ATTN16(256); break;
        default: pa_set_error("pa_sa_attention_f16: built for 64 / 128 / 256 channels, got %d", c); return PA_EUNSUPPORTED;
    }
#undef PA_ATTN16
    PA_CHECK_LAUNCH("pa_sa_attention_f16");
    return PA_OK;
}

PA_API int pa_sa_attention_f16(int b, int n, int c, int split, const float *yv, const float *x, void *scratch, float *stats, float *d, pa_stream_t stream)
{
    return attn16_dispatch(b, n, c, split, yv, x, scratch, stats, d, nullptr, nullptr, stream);
}

// The attention AND the layer behind it in one pass-2 launch: out (b, n, c) = x + relu(W (x - x_r) + bias), W = trans_conv with after_norm folded
// (pptnet.py:279-281), wtp = pa_sa_attention_f16_pack_trans(W K-major) (c * c fp16 elements), bt (c) fp32.
PA_API int pa_sa_attention_trans_f16(int b, int n, int c, int split, const float *yv, const float *x, void *scratch, float *stats, const void *wtp, const float *bt,
                                     float *out, pa_stream_t stream)
{
    PA_REQUIRE(wtp && bt && out, "pa_sa_attention_trans_f16: null weights / bias / output");
    return attn16_dispatch(b, n, c, split, yv, x, scratch, stats, out, reinterpret_cast<const _Float16 *>(wtp), bt, stream);
}

PA_API int pa_sa_attention_f16_pack_trans(int c, const float *wt, void *wtp, pa_stream_t stream)
{
    PA_REQUIRE(c > 0 && c % 32 == 0 && wt && wtp, "pa_sa_attention_f16_pack_trans: c=%d must be a multiple of 32", c);
    hipLaunchKernelGGL(attn_pack_trans16_kernel, dim3(pa_div_up((long)c * c, 256)), dim3(256), 0, (hipStream_t)stream, c, wt, reinterpret_cast<_Float16 *>(wtp));
    PA_CHECK_LAUNCH("pa_sa_attention_f16_pack_trans");
    return PA_OK;
}
