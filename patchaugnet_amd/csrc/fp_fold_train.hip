// Training-mode feature propagation with the FIRST dense layer folded through the interpolation (gfx950).
//
// Reference: the decoder level of patch_aug_net.py:350-362 -- interpolated = pointops.interpolation(known_feats, idx, weight) (b, C2, n);
// new = cat([interpolated, unknown_feats], 1) (b, C2 + C1, n); SharedMLP(new) whose first layer is conv1x1 (O x (C2 + C1)) + BatchNorm + ReLU
// (pt_util.py:16-41).  At the finest level n = 4096, m = 1024 known points, C2 = 256, C1 = 3 (raw coordinates): the layer costs three
// 18 x (256 x 4096 x 259) contractions per training step (forward, input gradient, weight gradient: 0.47 ms of a 5.7 ms step on the LDS-tiled
// GEMM, whose 64-row / 16-deep tiles also pay for the 259 that is not a multiple of anything), plus the 76 MB cat and its backward.
//
// Interpolation is linear and acts on the point axis, the 1x1 convolution on the channel axis: W [interp(F); S] = interp(W_a F) + W_b S.  So
//   forward   Z = W_a F on the m KNOWN points (a quarter of the columns: pa_tgemm_nn), then ONE pass here:
//             Y1[b,o,j] = sum_t w[b,j,t] Z[b,o,idx[b,j,t]] + sum_c W_b[o,c] S[b,c,j], with the BatchNorm statistics of Y1 in the epilogue;
//   backward  dY1 = BatchNorm/ReLU backward of (dAct1, Y1) formed on the fly (the LDS-tiled GEMM's operand transform), and in ONE pass here
//             G[b,o,i] = sum_{(j,t): idx[b,j,t] = i} w[b,j,t] dY1[b,o,j]   (the interpolation's transposed gather over inverted lists) and
//             dW_b[o,c] += sum_{b,j} dY1[b,o,j] S[b,c,j];  then dW_a = G F^T and dF = W_a^T G are contractions over m = 1024 columns.
// Same values as the unfolded form up to fp32 summation order (tests/test_gpu_train_ops.py against float64 autograd of the torch modules).
#include "pa_common.h"

namespace {

constexpr int FF_NT = 256;

// grid (ceil(n / (4 FF_NT)), O / CT, b); LDS: CT rows of Z (m floats each) + CT x 8 skip weights
template <int CT>
__global__ __launch_bounds__(FF_NT) void fp_fold_fwd_kernel(int O, int m, int n, int C1, const float *__restrict__ Z, const int *__restrict__ idx,
                                                            const float *__restrict__ w, const float *__restrict__ S, const float *__restrict__ Wb, int ldw,
                                                            float *__restrict__ out, double *__restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];      // [CT][m], then [CT][8] skip weights, then the reduction scratch
    float *wsk = rows + (size_t)CT * m;
    float *red = wsk + CT * 8;                                         // [FF_NT / 64][CT][2]
    const int b = blockIdx.z, o0 = blockIdx.y * CT, tid = threadIdx.x;
    {
        const float4 *s4 = reinterpret_cast<const float4 *>(Z + ((size_t)b * O + o0) * m);
        float4 *d4 = reinterpret_cast<float4 *>(rows);
        for (int i = tid; i < CT * (m >> 2); i += FF_NT) d4[i] = s4[i];
        if (tid < CT * 8) {
            const int r = tid >> 3, c = tid & 7;
            wsk[tid] = c < C1 ? Wb[(size_t)(o0 + r) * ldw + c] : 0.f;
        }
    }
    __syncthreads();
    float a1[CT], a2[CT];
#pragma unroll
    for (int r = 0; r < CT; ++r) a1[r] = a2[r] = 0.f;
    const int j0 = (blockIdx.x * FF_NT + tid) * 4;
    if (j0 < n) {                                                      // n % 4 == 0 (host): a group of four columns is in or out
        const int4 *ip = reinterpret_cast<const int4 *>(idx + ((size_t)b * n + j0) * 3);
        const float4 *wp = reinterpret_cast<const float4 *>(w + ((size_t)b * n + j0) * 3);
        const int4 ia = ip[0], ib = ip[1], ic = ip[2];
        const float4 wa = wp[0], wb = wp[1], wc = wp[2];
        const int id[12] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w, ic.x, ic.y, ic.z, ic.w};
        const float ww[12] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w, wc.x, wc.y, wc.z, wc.w};
        float4 sk[8];
#pragma unroll
        for (int c = 0; c < 8; ++c)
            sk[c] = c < C1 ? *reinterpret_cast<const float4 *>(S + ((size_t)b * C1 + c) * n + j0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < CT; ++r) {
            const float *row = rows + (size_t)r * m;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)       // the interpolation in the reference's order (interpolation_cuda_kernel.cu:194), then the skip term
                v[q] = ww[3 * q] * row[id[3 * q]] + ww[3 * q + 1] * row[id[3 * q + 1]] + ww[3 * q + 2] * row[id[3 * q + 2]];
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float wk = wsk[r * 8 + c];
                t[0] += wk * sk[c].x; t[1] += wk * sk[c].y; t[2] += wk * sk[c].z; t[3] += wk * sk[c].w;
            }
            const float4 y = make_float4(v[0] + t[0], v[1] + t[1], v[2] + t[2], v[3] + t[3]);
            *reinterpret_cast<float4 *>(out + ((size_t)b * O + o0 + r) * n + j0) = y;
            a1[r] = (y.x + y.y) + (y.z + y.w);
            a2[r] = (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
        }
    }
    if (stats) {
        // per-row sums over this workgroup's columns: wave reduction (fixed order), the waves meet in LDS, one fp64 atomic pair per row
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int r = 0; r < CT; ++r) {
            float s1 = a1[r], s2 = a2[r];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            if (lane == 0) { red[(wave * CT + r) * 2] = s1; red[(wave * CT + r) * 2 + 1] = s2; }
        }
        __syncthreads();
        if (tid < CT) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int wv = 0; wv < FF_NT / 64; ++wv) { s1 += (double)red[(wv * CT + tid) * 2]; s2 += (double)red[(wv * CT + tid) * 2 + 1]; }
            const unsigned slot = (blockIdx.x + blockIdx.z * gridDim.x) % PA_BN_STAT_SLOTS;
            double *st = stats + (size_t)slot * 2 * O;
            atomicAdd(st + o0 + tid, s1);
            atomicAdd(st + O + o0 + tid, s2);
        }
    }
}

// grid (O / 4, b), NT threads; LDS 4 n floats + reduction scratch.  The four channel rows of dY1 are built in LDS from (g, raw Y1, BatchNorm
// parameters), then (a) contracted with the skip rows for dW_b and (b) gathered over the inverted neighbour lists into G.
// CP: skip channels padded to 4 or 8 (compile time: the accumulators and their reduction cost CP, not 8)
template <int NT, int CP>
__global__ __launch_bounds__(NT) void fp_fold_bwd_kernel(int O, int n, int m, int C1, const float *__restrict__ g, const float *__restrict__ yraw,
                                                         const float *__restrict__ p, int relu, const float *__restrict__ S, const int *__restrict__ off_all,
                                                         const int2 *__restrict__ ent_all, float *__restrict__ G, float *__restrict__ dWb, int ldw)
{
    extern __shared__ __attribute__((aligned(16))) float rows[];      // [4][n], then [NT / 64][32] reduction scratch
    float *red = rows + (size_t)4 * n;
    const int b = blockIdx.y, c0 = blockIdx.x * 4, tid = threadIdx.x;
    {
        const float4 *g4 = reinterpret_cast<const float4 *>(g + ((size_t)b * O + c0) * n);
        const float4 *y4 = reinterpret_cast<const float4 *>(yraw + ((size_t)b * O + c0) * n);
        float4 *d4 = reinterpret_cast<float4 *>(rows);
        const int nq = n >> 2;
        for (int i = tid; i < 4 * nq; i += NT) {
            const int ch = c0 + i / nq;
            const float p0 = p[ch], p1 = p[(size_t)O + ch], p2 = p[(size_t)2 * O + ch], p3 = p[(size_t)3 * O + ch], p4 = p[(size_t)4 * O + ch],
                        p5 = p[(size_t)5 * O + ch], p6 = p[(size_t)6 * O + ch];
            const float4 gv = g4[i], yv = y4[i];
            auto tf = [&](float gg, float yy) {      // train_gemm.hip tf_apply<TF_BN_BWD_RELU / TF_BN_BWD>
                const float z = fmaf(yy, p0, p1);
                const float gm = (!relu || z > 0.f) ? gg : 0.f;
                const float xhat = (yy - p2) * p3;
                return (gm - p4 - xhat * p5) * p6;
            };
            d4[i] = make_float4(tf(gv.x, yv.x), tf(gv.y, yv.y), tf(gv.z, yv.z), tf(gv.w, yv.w));
        }
    }
    __syncthreads();
    // (a) dW_b[c0 + r][c] += sum_j dY1[r][j] S[b][c][j]
    if (C1 > 0) {
        float acc[4][CP];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < CP; ++c) acc[r][c] = 0.f;
        for (int j = tid; j < n; j += NT) {
            float sv[CP];
#pragma unroll
            for (int c = 0; c < CP; ++c) sv[c] = c < C1 ? S[((size_t)b * C1 + c) * n + j] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = rows[(size_t)r * n + j];
#pragma unroll
                for (int c = 0; c < CP; ++c) acc[r][c] += d * sv[c];
            }
        }
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                float v = acc[r][c];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
                if (lane == 0) red[wave * 32 + r * 8 + c] = v;      // (columns c >= CP of the scratch are never read: the adder below stops at C1 <= CP)
            }
        __syncthreads();
        if (tid < 32 && (tid & 7) < C1) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < NT / 64; ++wv) v += red[wv * 32 + tid];
            atomicAdd(dWb + (size_t)(c0 + (tid >> 3)) * ldw + (tid & 7), v);
        }
    }
    // (b) G[b][c0 + r][i] = sum over the list of known point i (csrc/gather.hip: interp_bwd_gather_kernel; four entries per round trip)
    const int *off = off_all + (size_t)b * (m + 1);
    const int2 *ent = ent_all + (size_t)b * n * 3;
    float *dst = G + ((size_t)b * O + c0) * m;
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
                    const float wv = __int_as_float(en[q].y);
                    a0 += wv * rows[en[q].x];
                    a1 += wv * rows[n + en[q].x];
                    a2 += wv * rows[2 * n + en[q].x];
                    a3 += wv * rows[3 * n + en[q].x];
                }
        }
        dst[j] = a0;
        dst[(size_t)m + j] = a1;
        dst[2 * (size_t)m + j] = a2;
        dst[3 * (size_t)m + j] = a3;
    }
}

bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

// Y1 (b, O, n) = interpolation(Z (b, O, m); idx, weight (b, n, 3)) + Wb (O x C1, row stride ldw) . S (b, C1, n); stats: PA_BN_STAT_SLOTS x 2*O doubles
// receiving (accumulating) the per-channel sum and sum of squares of Y1, or NULL.  O % 8 == 0, m % 4 == 0, n % 4 == 0, 0 <= C1 <= 8, m <= 4096.
PA_API int pa_fp_fold_forward(int b, int O, int m, int n, int C1, const float *Z, const int *idx, const float *weight, const float *S, const float *Wb, int ldw,
                              float *out, double *stats, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && O > 0 && m > 0 && n > 0 && Z && idx && weight && out, "pa_fp_fold_forward: bad arguments");
    PA_REQUIRE(C1 >= 0 && C1 <= 8 && (C1 == 0 || (S && Wb)), "pa_fp_fold_forward: 0 <= C1 <= 8 skip channels (got %d)", C1);
    PA_REQUIRE(O % 8 == 0 && m % 4 == 0 && n % 4 == 0 && m <= 4096, "pa_fp_fold_forward: O %% 8, m %% 4, n %% 4, m <= 4096 (O=%d m=%d n=%d)", O, m, n);
    PA_REQUIRE(al16(Z) && al16(idx) && al16(weight) && al16(out) && (C1 == 0 || al16(S)), "pa_fp_fold_forward: 16-byte aligned tensors");
    PA_REQUIRE(b <= 65535 && O / 8 <= 65535, "pa_fp_fold_forward: grid limits");
    constexpr int CT = 8;
    const size_t lds = ((size_t)CT * m + CT * 8 + (FF_NT / 64) * CT * 2) * sizeof(float);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fp_fold_fwd_kernel<CT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fp_fold_fwd_kernel<CT>, dim3(pa_div_up(n, 4 * FF_NT), O / CT, b), dim3(FF_NT), lds, (hipStream_t)stream, O, m, n, C1, Z, idx, weight, S, Wb,
                       ldw, out, stats);
    PA_CHECK_LAUNCH("pa_fp_fold_forward");
    return PA_OK;
}

// The first layer's backward behind the fold.  g (b, O, n): gradient of the layer's activation; yraw (b, O, n): its raw output; p: the layer's
// 7*O BatchNorm block after pa_bn_bwd_finalize; relu: the activation's mask.  lists: pa_interpolation_backward_lists(idx, weight) of this level.
// Writes G (b, O, m) = interpolation^T(dY1) and ADDS dY1 . S^T to dWb (O x C1, row stride ldw).  O % 4 == 0, n % 4 == 0, n <= 4096, m <= 8192.
PA_API int pa_fp_fold_backward(int b, int O, int n, int m, int C1, const float *g, const float *yraw, const float *p, int relu, const float *S,
                               const int *lists, float *G, float *dWb, int ldw, pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && O > 0 && m > 0 && n > 0 && g && yraw && p && lists && G, "pa_fp_fold_backward: bad arguments");
    PA_REQUIRE(C1 >= 0 && C1 <= 8 && (C1 == 0 || (S && dWb)), "pa_fp_fold_backward: 0 <= C1 <= 8 skip channels (got %d)", C1);
    PA_REQUIRE(O % 4 == 0 && n % 4 == 0 && n <= 4096 && m <= 8192, "pa_fp_fold_backward: O %% 4, n %% 4, n <= 4096, m <= 8192 (O=%d n=%d m=%d)", O, n, m);
    PA_REQUIRE(al16(g) && al16(yraw) && (reinterpret_cast<uintptr_t>(lists) & 7) == 0, "pa_fp_fold_backward: alignment");
    PA_REQUIRE(b <= 65535, "pa_fp_fold_backward: grid limits");
    const int *off = lists;
    const int2 *ent = reinterpret_cast<const int2 *>(lists + (((size_t)b * (m + 1) + 1) & ~(size_t)1));
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_BWD                                                                                                                                              \
    if (C1 <= 4) {                                                                                                                                              \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fp_fold_bwd_kernel<NT, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((fp_fold_bwd_kernel<NT, 4>), dim3(O / 4, b), dim3(NT), lds, st, O, n, m, C1, g, yraw, p, relu, S, off, ent, G, dWb, ldw);             \
    } else {                                                                                                                                                    \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fp_fold_bwd_kernel<NT, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((fp_fold_bwd_kernel<NT, 8>), dim3(O / 4, b), dim3(NT), lds, st, O, n, m, C1, g, yraw, p, relu, S, off, ent, G, dWb, ldw);             \
    }
    if (m >= 1024) {
        constexpr int NT = 1024;
        const size_t lds = ((size_t)4 * n + (NT / 64) * 32) * sizeof(float);
        LAUNCH_BWD
    } else {
        constexpr int NT = 256;
        const size_t lds = ((size_t)4 * n + (NT / 64) * 32) * sizeof(float);
        LAUNCH_BWD
    }
#undef LAUNCH_BWD
    PA_CHECK_LAUNCH("pa_fp_fold_backward");
    return PA_OK;
}
