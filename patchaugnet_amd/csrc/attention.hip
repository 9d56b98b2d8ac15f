// Grouped self-attention of the pyramid point transformer (PPT-Net) as two MFMA passes, without the (B, gp, N, N) tensor.
//
// Reference: SA_Layer.forward, place_recognition/pptnet_origin/models/pptnet.py:261-282 (twin: GroupSALayer,
// place_recognition/patch_aug_net/models/loupe.py:69-114):
//     Y = q_conv(x) = k_conv(x)  (tied grouped 1x1 conv)         energy = sum_g Y_g^T Y_g = Y^T Y           (:264-274)
//     attn = softmax(energy, dim=-1);  attn = attn / (1e-9 + attn.sum(dim=1))                               (:276-277)
//     x_r = v_conv(x) @ attn;   x = x + relu(BN(trans_conv(x - x_r)))                                       (:278-281)
// The reference materialises (B, 8, N, N) per-group products (32 MB per submap at N = 1024), sums them, and runs soft-max,
// column sum, division and a second batched matmul as separate kernels.
//
// MI355X plan (activations point-major, like the rest of the engine).  A preceding linear launch produces
// YV = [Y | V] (B, N, 2C).  The energy matrix is symmetric, so "row i of the soft-max" needs only row statistics
// m_i = max_j e_ij and l_i = sum_j exp(e_ij - m_i):
//   pass 1  every wavefront owns 16 points j, streams all points i through LDS tiles and computes
//           e(i, j) = Y_i . Y_j with v_mfma_f32_16x16x4_f32 (own rows as the B operand, held in registers); by symmetry
//           the column statistics it accumulates ARE the row statistics of its own points -> stats (m, 1/l);
//   pass 2  the same e(i, j) tiles again, p = exp(e - m_i) / l_i on the accumulator layout; the accumulator registers
//           are fed straight back as the B operand of x_r^T += V^T p (the contraction index is permuted consistently on
//           both operands, so no LDS round trip), column sums alongside; epilogue divides by (1e-9 + sum_i p) and
//           writes d = x - x_r.
// The trans_conv + BatchNorm + ReLU + residual that follows is one pa_linear launch (mlp_chain.hip).
// HBM traffic per cloud: YV read N/64 times from L2 (it is 0.5 MB at N = 1024, C = 64), nothing N x N ever leaves the CU.
#include "pa_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

// exp on the transcendental unit: v_exp_f32(x * log2 e), ~2 ulp; both passes use the same function, so the soft-max rows stay
// normalised exactly as computed.  exp(-inf) = 0.
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

template <int C>
struct AttnCfg {
    static constexpr int TJ = C <= 128 ? 64 : (C == 256 ? 32 : 16);   // rows of Y / V per LDS tile
    static constexpr int YS = C + 4;                                  // 16-byte A-fragment reads: row l%16 at float 16q + 4(l/16) -> bank 4(row + g + 4q), conflict-free
    static constexpr int VS = C + 4;                                  // 16-byte reads of V[4g + r][64u + 4(l%16) ..]: 16 lanes of a phase cover 64 banks
};

template <int C, int PASS>
__global__ __launch_bounds__(256) void sa_attn_kernel(int n, const float *__restrict__ yv_all, const float *__restrict__ x_all,
                                                       float *__restrict__ stats_all, float *__restrict__ d_all,
                                                       const float *__restrict__ wt = nullptr, const float *__restrict__ bt = nullptr)
{
    // wt != null (pass 2): the layer behind the attention -- trans_conv + BatchNorm (folded) + ReLU + residual, pptnet.py:279-281 -- runs on the wave's 16
    // points in the epilogue and d_all receives x + relu(W d + b) instead of d (one launch and one (rows x C) round trip less per level).  wt: K-major
    // (C x C) fp32 with the BatchNorm folded, bt (C).  Exact fp32 MFMA; the contraction visits the channels in the order the registers hold them.
    using Cfg = AttnCfg<C>;
    constexpr int TJ = Cfg::TJ, YS = Cfg::YS, VS = Cfg::VS, KS = C / 4, CT = C / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ys = smem;                                  // [TJ][YS]
    float *Vs = Ys + TJ * YS;                          // [TJ][VS]      (pass 2)
    float *Ms = Vs + (PASS == 2 ? TJ * VS : 0);        // [TJ] m_i, [TJ] 1/l_i   (pass 2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int j0 = blockIdx.x * 64 + wave * 16;
    const bool active = j0 < n;                        // wave-uniform; lanes whose own point is >= n compute on a clamped row and store nothing
    const float *yv = yv_all + (size_t)b * n * (2 * C);
    const float *stats = stats_all + (size_t)b * n * 2;

    // Own points as the B operand.  The contraction index is permuted the same way on both operands so that a lane's four
    // consecutive k-steps read ONE float4: MFMA (q, s) contracts k = 16q + 4(l/16) + s, i.e. B[slot l/16][j = l%16] = Y[j0 + l%16][16q + 4(l/16) + s].
    float4 yb[KS / 4];
    if (active) {
        const float *p = yv + (size_t)min(j0 + (lane & 15), n - 1) * (2 * C) + (lane >> 4) * 4;
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) yb[q] = *reinterpret_cast<const float4 *>(p + q * 16);
    }
    float m_run = -INFINITY, l_run = 0.f, s_run = 0.f;
    floatx4 o[PASS == 2 ? CT : 1];
#pragma unroll
    for (int ct = 0; ct < (PASS == 2 ? CT : 1); ++ct) o[ct] = (floatx4){0.f, 0.f, 0.f, 0.f};

    // Tiles are fetched one step ahead into registers (global latency hides under the previous tile's MFMAs) and dropped into LDS
    // between the two barriers.
    constexpr int Q = C / 4, NPF = TJ * Q / 256;
    static_assert(TJ * Q % 256 == 0, "tile must split evenly over the workgroup");
    float4 py[NPF], pv[PASS == 2 ? NPF : 1];
    float pm = 0.f, pl = 0.f;
    auto fetch = [&](int t0) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int q = tid + u * 256, r = q / Q, part = q - r * Q;
            const bool ok = t0 + r < n;
            const float *src = yv + (size_t)(t0 + r) * (2 * C) + part * 4;
            py[u] = ok ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (PASS == 2) pv[u] = ok ? *reinterpret_cast<const float4 *>(src + C) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (PASS == 2 && tid < TJ) {
            const bool ok = t0 + tid < n;
            pm = ok ? stats[(size_t)(t0 + tid) * 2] : 0.f;
            pl = ok ? stats[(size_t)(t0 + tid) * 2 + 1] : 0.f;
        }
    };
    constexpr bool PF = C <= 256;          // C = 512 (N = 16 in PPT-Net: a single tile) has no registers to spare for the look-ahead
    if (PF) fetch(0);
    for (int t0 = 0; t0 < n; t0 += TJ) {
        if (!PF) fetch(t0);
        // ---- rows t0 .. t0+TJ-1 of Y (and V, m, 1/l): registers -> LDS ---------------------------------------------
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int q = tid + u * 256, r = q / Q, part = q - r * Q;
            *reinterpret_cast<float4 *>(Ys + r * YS + part * 4) = py[u];
            if (PASS == 2) *reinterpret_cast<float4 *>(Vs + r * VS + part * 4) = pv[u];
        }
        if (PASS == 2 && tid < TJ) {
            Ms[tid] = pm;
            Ms[TJ + tid] = pl;
        }
        __syncthreads();
        if (PF && t0 + TJ < n) fetch(t0 + TJ);
        if (active) {
            const int nit = (min(TJ, n - t0) + 15) >> 4;
            for (int it = 0; it < nit; ++it) {
                floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
                const float *ap = Ys + (it * 16 + (lane & 15)) * YS + (lane >> 4) * 4;
#pragma unroll
                for (int q = 0; q < KS / 4; ++q) {
                    const float4 a = *reinterpret_cast<const float4 *>(ap + q * 16);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, yb[q].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, yb[q].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, yb[q].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, yb[q].w, acc, 0, 0, 0);
                }
                // acc[r] = e(i = t0 + 16it + 4(l/16) + r, j = j0 + l%16)
                if (PASS == 1) {
                    const int ig = t0 + it * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ig + r >= n) acc[r] = -INFINITY;                   // rows past the cloud contribute exp(-inf) = 0
                    const float mt = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
                    const float mn = fmaxf(m_run, mt);
                    if (mn > -INFINITY) {
                        l_run = l_run * fexp(m_run - mn) + ((fexp(acc[0] - mn) + fexp(acc[1] - mn)) + (fexp(acc[2] - mn) + fexp(acc[3] - mn)));
                        m_run = mn;
                    }
                } else {
                    const int ib = it * 16 + (lane >> 4) * 4;
                    const float4 mi = *reinterpret_cast<const float4 *>(Ms + ib), li = *reinterpret_cast<const float4 *>(Ms + TJ + ib);
                    float p[4];
                    p[0] = fexp(acc[0] - mi.x) * li.x;
                    p[1] = fexp(acc[1] - mi.y) * li.y;
                    p[2] = fexp(acc[2] - mi.z) * li.z;
                    p[3] = fexp(acc[3] - mi.w) * li.w;
                    s_run += (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // A[row m = l%16][slot g = l/16] = V[i = 16it + 4g + r][chan(ct, m)], chan(ct, m) = 64(ct/4) + 4m + ct%4: one float4 per four ct
                        const float *vp = Vs + (ib + r) * VS + (lane & 15) * 4;
#pragma unroll
                        for (int u = 0; u < CT / 4; ++u) {
                            const float4 a = *reinterpret_cast<const float4 *>(vp + u * 64);
                            o[4 * u + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, p[r], o[4 * u + 0], 0, 0, 0);
                            o[4 * u + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, p[r], o[4 * u + 1], 0, 0, 0);
                            o[4 * u + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, p[r], o[4 * u + 2], 0, 0, 0);
                            o[4 * u + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, p[r], o[4 * u + 3], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (!active) return;

    if (PASS == 1) {
        // merge the four lane groups that hold different i for the same own point j = l%16
#pragma unroll
        for (int sft = 16; sft < 64; sft <<= 1) {
            const float mo = __shfl_xor(m_run, sft), lo = __shfl_xor(l_run, sft);
            const float mn = fmaxf(m_run, mo);
            l_run = mn > -INFINITY ? l_run * fexp(m_run - mn) + lo * fexp(mo - mn) : 0.f;
            m_run = mn;
        }
        if (lane < 16 && j0 + lane < n) {
            float *st = stats_all + ((size_t)b * n + j0 + lane) * 2;
            st[0] = m_run;
            st[1] = 1.0f / l_run;
        }
    } else {
        s_run += __shfl_xor(s_run, 16);
        s_run += __shfl_xor(s_run, 32);
        const float den = 1e-9f + s_run;                                       // pptnet.py:277
        const bool valid = j0 + (lane & 15) < n;   // lanes past the cloud stay in the wave: the fused epilogue's MFMAs take their weight fragments
        const size_t row = (size_t)b * n + min(j0 + (lane & 15), n - 1);
        const float *xr = x_all + row * C;
        float *dr = d_all + row * C;
#pragma unroll
        for (int u = 0; u < CT / 4; ++u) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int c = u * 64 + (lane >> 4) * 16 + rr * 4;              // o[4u + e][rr]: channel chan(4u + e, 4(l/16) + rr) = c + e, point j = l%16
                const float4 xv = *reinterpret_cast<const float4 *>(xr + c);
                float4 d;
                d.x = xv.x - o[4 * u + 0][rr] / den;
                d.y = xv.y - o[4 * u + 1][rr] / den;
                d.z = xv.z - o[4 * u + 2][rr] / den;
                d.w = xv.w - o[4 * u + 3][rr] / den;
                if (wt == nullptr) {
                    if (valid) *reinterpret_cast<float4 *>(dr + c) = d;
                } else {                                                        // kept in place: the B operands of the fused layer
                    o[4 * u + 0][rr] = d.x; o[4 * u + 1][rr] = d.y; o[4 * u + 2][rr] = d.z; o[4 * u + 3][rr] = d.w;
                }
            }
        }
        if (wt == nullptr) return;
        // out[j][co] = x[j][co] + relu(sum_c wt[c][co] d[j][c] + bt[co]): MFMA (u, rr, e) contracts the channels {64 u + 16 g + 4 rr + e : g = 0..3} -- a lane's
        // own register as the B operand, the weight row of the SAME channel as the A operand (row m = output channel 16 cot + m)
        const int lg = lane >> 4, lm = lane & 15;
        for (int cot = 0; cot < CT; ++cot) {
            floatx4 acc = (floatx4){0.f, 0.f, 0.f, 0.f};
            const float *wp = wt + (size_t)(16 * lg) * C + 16 * cot + lm;
#pragma unroll
            for (int u = 0; u < CT / 4; ++u)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[(size_t)(64 * u + 4 * rr + e) * C], o[4 * u + e][rr], acc, 0, 0, 0);
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

template <int C>
int launch_attn(int b, int n, const float *yv, const float *x, float *stats, float *d, hipStream_t st, const float *wt = nullptr, const float *bt = nullptr)
{
    using Cfg = AttnCfg<C>;
    const size_t lds1 = (size_t)Cfg::TJ * Cfg::YS * 4;
    const size_t lds2 = (size_t)(Cfg::TJ * Cfg::YS + Cfg::TJ * Cfg::VS + 2 * Cfg::TJ) * 4;
    const dim3 grid(pa_div_up(n, 64), b);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&sa_attn_kernel<C, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    hipLaunchKernelGGL((sa_attn_kernel<C, 1>), grid, dim3(256), lds1, st, n, yv, x, stats, d, (const float *)nullptr, (const float *)nullptr);
    hipLaunchKernelGGL((sa_attn_kernel<C, 2>), grid, dim3(256), lds2, st, n, yv, x, stats, d, wt, bt);
    return 0;
}

}  // namespace

static int attn_dispatch(int b, int n, int c, const float *yv, const float *x, float *stats, float *d, const float *wt, const float *bt, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && yv && x && stats && d, "pa_sa_attention: bad arguments");
    PA_REQUIRE(b <= 65535, "pa_sa_attention: b=%d exceeds the grid limit", b);
    hipStream_t st = (hipStream_t)stream;
    switch (c) {
        case 64: launch_attn<64>(b, n, yv, x, stats, d, st, wt, bt); break;
        case 128: launch_attn<128>(b, n, yv, x, stats, d, st, wt, bt); break;
        case 256: launch_attn<256>(b, n, yv, x, stats, d, st, wt, bt); break;
        case 512: launch_attn<512>(b, n, yv, x, stats, d, st, wt, bt); break;
        default: pa_set_error("pa_sa_attention: built for 64/128/256/512 channels (PPT-Net widths), got %d", c); return PA_EUNSUPPORTED;
    }
    PA_CHECK_LAUNCH("pa_sa_attention");
    return PA_OK;
}

PA_API int pa_sa_attention(int b, int n, int c, const float *yv, const float *x, float *stats, float *d, pa_stream_t stream)
{
    return attn_dispatch(b, n, c, yv, x, stats, d, nullptr, nullptr, stream);
}

// The attention AND the layer behind it in one pass-2 launch: out (b, n, c) = x + relu(W (x - x_r) + bias); wt: K-major (c x c) fp32 = trans_conv with
// after_norm folded (pptnet.py:279-281), bt (c).  Exact fp32 MFMA (the channels are contracted in the order the attention's registers hold them).
PA_API int pa_sa_attention_trans(int b, int n, int c, const float *yv, const float *x, float *stats, const float *wt, const float *bt, float *out, pa_stream_t stream)
{
    PA_REQUIRE(wt && bt && out, "pa_sa_attention_trans: null weights / bias / output");
    return attn_dispatch(b, n, c, yv, x, stats, out, wt, bt, stream);
}
