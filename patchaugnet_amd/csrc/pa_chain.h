// Shared between the fp32 and fp16 shared-MLP chain kernels (mlp_chain.hip, mlp_chain_f16.hip): launch descriptor, tile
// synchronisation and the gather / interpolate PROLOGUE that builds the first layer's activation tile in LDS.
#pragma once
#include "pa_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct PaLayer {
    const float *wt;    // [kpad][n], K-major, rows >= k are zero
    const float *wp;    // optional fragment-major packing of the same matrix (pa_pack_weights), n % 64 == 0; null = use wt
    const _Float16 *wp16;  // fp16 path: fragment-major fp16 packing (pa_pack_weights_f16), K padded to k32
    int k32;               // fp16 path: K rounded up to 32
    const float *bias;  // [n]
    int kpad;           // multiple of 4
    int n;              // multiple of 16
    int ldw;            // row stride of wt in floats (= n, or the full width when this launch computes a column slice)
};

struct PaChain {
    int nlayers;
    PaLayer L[3];
    long rows;        // plain / FP: number of rows; SA: number of groups (B*m)
    int k0;           // true number of input channels
    int lds_stride;   // floats per activation row in LDS (max kpad over layer inputs + 2)
    int wave_floats;  // floats of LDS per wave (activation tile + prologue scratch)
    // MODE 0: plain rows
    const float *x;
    int ldx;
    // MODE 1: set-abstraction gather
    const float *xyz;        // (B, n_src, 3)
    const float *feat;       // (B, n_src, c_feat) point-major
    const int *center_idx;   // (B, m_ctr)
    const int *nbr_idx;      // (B, m_ctr, ns)
    int n_src, m_ctr, ns, c_feat;
    // MODE 2: feature-propagation interpolate + skip
    const float *known;  // (B, m_known, c2) point-major
    const int *idx3;     // (B, n_unknown, 3)
    const float *w3;     // (B, n_unknown, 3)
    const float *skip;   // (B, n_unknown, c1) point-major
    int n_unknown, m_known, c2, c1;
    // MODE 3: feature propagation with the first layer folded into the prologue (known = W1a-premultiplied features, see pa_fp_chain_premul)
    const float *wskip;  // (c1, c2) K-major: the first layer's weights for the skip channels, BatchNorm folded
    const float *bias0;  // (c2)
    float *out;
    int ldo;
    float *tap;                  // optional second output: the result of layer nlayers - 2 (row-major, ldtap), written when that layer finishes --
    int ldtap;                   // the chain then continues with one more layer on the tile that is still in LDS (fused pre-multiply)
    // last-layer epilogue (plain rows only): out = residual + act(acc + bias), act = ReLU when relu_last != 0 else identity
    int relu_last;
    const float *residual;   // (rows, ldr) or null
    int ldr;
    long long *dbg;          // profiling only: per-tile s_memtime stamps at phase boundaries (null in production)
    int xcd_remap;           // != 0: contiguous tile ranges per XCD (see chain_kernel)
    int ep_stride;           // > 0: the last layer's tile is staged through LDS (row stride ep_stride floats) and leaves as whole rows
    int fold0;               // MODE_FP only, != 0: `known` holds features already multiplied by the first layer's interpolated-part weights;
                             //   layer 0 then contracts only the c1 skip channels (tile columns c2 ..) and ADDS the interpolated term it finds in
                             //   columns 0 .. c2-1 before bias and ReLU (linearity of interpolation; pa_fp_chain_premul with c1 > 4)
    int vec_out;             // != 0: out (and residual) rows are 16-byte aligned -> 16-byte stores
    // Single-layer launches over few rows (pa_linear): gridDim.y column slices of slice_n outputs each, so that a 512-row layer fills
    // the chip instead of 32 workgroups streaming the whole weight matrix each.  L[0].n == slice_n; slice y shifts the packed weights,
    // bias, out and residual (packed layouts are column-group-major, so a slice is a contiguous range).
    int col_slices, slice_n;
    long wp_slice, wp16_slice;   // packed fp32 floats / fp16 halfs per slice
    int stagger;                 // eight-wave wave-private variant: waves 4..7 (the second wave of every SIMD) start this many x 8128 cycles late
    // Group window (persistent first-level kernel only, sa_tiny.hip): > 0 = this launch computes centres win_off .. win_off + win_len - 1 of
    // EVERY cloud (rows = clouds * win_len); the centre / neighbour / output rows keep their m_ctr-per-cloud layout.  pa_sa_group_window.
    int win_len, win_off;
};

namespace {

enum { MODE_PLAIN = 0, MODE_SA = 1, MODE_FP = 2, MODE_FPX = 3 };

// Column slice y = blockIdx.y of a single-layer launch: shifted copies of the layer descriptor and of the output / residual pointers.
// (The kernel argument itself is never written: a modified by-value argument would be demoted to scratch memory.)
__device__ __forceinline__ void pa_col_slice(const PaChain &a, PaLayer &L, float *&out, const float *&residual)
{
    if (a.col_slices > 1) {
        const int y = blockIdx.y;
        L.wt += y * a.slice_n;
        if (L.wp) L.wp += (size_t)y * a.wp_slice;
        if (L.wp16) L.wp16 += (size_t)y * a.wp16_slice;
        L.bias += y * a.slice_n;
        out += y * a.slice_n;
        if (residual) residual += y * a.slice_n;
    }
}

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// WPT = waves per tile: 1 = the wave owns its rows end to end (no workgroup barrier anywhere);
//                       4 = the workgroup's four waves share one tile and split every layer's COLUMNS, for
//                           problems with too few row tiles to fill 1024 SIMDs (two barriers per hidden layer).
template <int WPT>
__device__ __forceinline__ void tile_sync()
{
    if (WPT == 1) lds_fence();
    else __syncthreads();
}

// four consecutive activations to LDS (8-byte aligned destination)
__device__ __forceinline__ void pa_store4(float *p, float a, float b, float c, float d)
{
    float2 *q = reinterpret_cast<float2 *>(p);
    q[0] = make_float2(a, b);
    q[1] = make_float2(c, d);
}
__device__ __forceinline__ void pa_store4(_Float16 *p, float a, float b, float c, float d)
{
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    half4 h = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
    *reinterpret_cast<half4 *>(p) = h;
}

// Builds the R x k0pad activation tile of layer 0 (element type T, row stride `stride` elements) for tile `tile`.
// scratch: per-tile LDS scratch (indices / weights) right behind the tile.  The caller synchronises (tile_sync) afterwards.
template <typename T, int R, int MODE, bool POOLED, int WPT>
__device__ __forceinline__ void chain_prologue(T *act, float *scratch, const PaChain &a, long tile, int tid, int lane, int stride, int k0pad)
{
    constexpr int NTH = WPT * 64;
    // ---------------------------------------------------------------- prologue: build the A tile of layer 0
    if (MODE == MODE_PLAIN) {
        const long row0 = tile * R;
        if ((a.k0 & 3) == 0 && (a.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0) {
            // aligned rows: 16-byte loads, four activations per store (no per-element division)
            const int qpr = a.k0 >> 2;
            for (int q = tid; q < R * qpr; q += NTH) {
                const int r = q / qpr, part = q - r * qpr;
                const long row = row0 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < a.rows) v = *reinterpret_cast<const float4 *>(a.x + row * a.ldx + part * 4);
                pa_store4(act + r * stride + part * 4, v.x, v.y, v.z, v.w);
            }
            const int tail = k0pad - a.k0;   // fp16 path: K is padded to 32
            for (int q = tid; q < R * tail; q += NTH) {
                const int r = q / tail, ch = q - r * tail;
                act[r * stride + a.k0 + ch] = 0.f;
            }
        } else {
            for (int q = tid; q < R * k0pad; q += NTH) {
                const int r = q / k0pad, ch = q - r * k0pad;
                const long row = row0 + r;
                act[r * stride + ch] = (row < a.rows && ch < a.k0) ? a.x[row * a.ldx + ch] : 0.f;
            }
        }
    } else if (MODE == MODE_SA) {
        int *src = reinterpret_cast<int *>(scratch);  // [R] source point (global row), -1 = padding row
        int *ctr = src + R;                                    // [R] centre point (global row)
        for (int r = tid; r < R; r += NTH) {
            long gid;
            int s;
            if (POOLED) { gid = tile * 4 + (r & 3); s = r >> 2; if (s >= a.ns) s = 0; }
            else { const long grow = tile * R + r; gid = grow / a.ns; s = (int)(grow - gid * a.ns); }
            if (gid < a.rows) {
                const long b = gid / a.m_ctr;
                src[r] = (int)(b * a.n_src + a.nbr_idx[gid * a.ns + s]);
                ctr[r] = (int)(b * a.n_src + a.center_idx[gid]);
            } else {
                src[r] = -1;
                ctr[r] = 0;
            }
        }
        tile_sync<WPT>();
        for (int q = tid; q < R * 3; q += NTH) {  // centred coordinates -> channels 0..2 (pointops.py:562)
            const int r = q / 3, t = q - r * 3;
            const int s = src[r];
            act[r * stride + t] = s >= 0 ? a.xyz[(size_t)s * 3 + t] - a.xyz[(size_t)ctr[r] * 3 + t] : 0.f;
        }
        const int C = a.c_feat;
        if ((C & 3) == 0) {  // centred features -> channels 3..3+C (pointops.py:567-568), 16-byte loads
            const int qpr = C >> 2;
            const float4 *f4 = reinterpret_cast<const float4 *>(a.feat);
            // four items per trip, all eight 16-byte gathers (neighbour and centre rows) requested before the first use: a 16-row x 256-channel tile is
            // then one round trip to L2 per thread instead of four dependent ones (padding rows read row 0 and are zeroed)
            const int items = R * qpr;
            for (int q0 = tid; q0 < items; q0 += NTH * 4) {
                float4 pv[4], cv[4];
                int rr[4], pp[4], ss[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = min(q0 + u * NTH, items - 1);
                    rr[u] = q / qpr;
                    pp[u] = q - rr[u] * qpr;
                    ss[u] = src[rr[u]];
                    pv[u] = f4[(size_t)max(ss[u], 0) * qpr + pp[u]];
                    cv[u] = f4[(size_t)ctr[rr[u]] * qpr + pp[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (q0 + u * NTH >= items) break;
                    const bool live = ss[u] >= 0;
                    T *d = act + rr[u] * stride + 3 + pp[u] * 4;
                    d[0] = live ? pv[u].x - cv[u].x : 0.f; d[1] = live ? pv[u].y - cv[u].y : 0.f;
                    d[2] = live ? pv[u].z - cv[u].z : 0.f; d[3] = live ? pv[u].w - cv[u].w : 0.f;
                }
            }
        } else {
            for (int q = tid; q < R * C; q += NTH) {
                const int r = q / C, ch = q - r * C;
                const int s = src[r];
                act[r * stride + 3 + ch] = s >= 0 ? a.feat[(size_t)s * C + ch] - a.feat[(size_t)ctr[r] * C + ch] : 0.f;
            }
        }
        for (int q = tid; q < R * (k0pad - a.k0); q += NTH) {  // zero the K padding
            const int r = q / (k0pad - a.k0), ch = q - r * (k0pad - a.k0);
            act[r * stride + a.k0 + ch] = 0.f;
        }
    } else {  // MODE_FP / MODE_FPX
        int *nb = reinterpret_cast<int *>(scratch);  // [R][3] global rows of the three known neighbours
        float *wt = reinterpret_cast<float *>(nb + 3 * R);    // [R][3] interpolation weights
        float *sk = wt + 3 * R;                               // [R][4] skip channels (MODE_FPX)
        const long row0 = tile * R;
        for (int q = tid; q < R * 3; q += NTH) {
            const int r = q / 3;
            const long p = row0 + r;
            if (p < a.rows) {
                const long b = p / a.n_unknown;
                nb[q] = (int)(b * a.m_known + a.idx3[p * 3 + (q - r * 3)]);
                wt[q] = a.w3[p * 3 + (q - r * 3)];
            } else {
                nb[q] = 0;
                wt[q] = 0.f;
            }
        }
        const int C2 = a.c2, C1 = a.c1;
        if (MODE == MODE_FPX)
            for (int q = tid; q < R * 4; q += NTH) {
                const int r = q >> 2, t = q & 3;
                const long p = row0 + r;
                sk[q] = (p < a.rows && t < C1) ? a.skip[p * C1 + t] : 0.f;
            }
        tile_sync<WPT>();
        const int qpr = C2 >> 2;  // host guarantees c2 % 4 == 0
        const float4 *k4 = reinterpret_cast<const float4 *>(a.known);
        const int items = R * qpr;
        // PF items per trip with all 3 PF 16-byte gathers issued before the first use: the known features of a whole batch
        // (33 MB at fp0) live in the Infinity Cache, not in L2, and a wave-private tile has nobody else to hide that latency.  Shared tiles
        // (four waves per tile): eight items per trip -- the coarser levels' 16- and 32-row tiles are then ONE trip per thread instead of two
        // dependent ones (phase stamps: the prologue was 14.6 k of a tile's 68.6 k cycles at both of them), and the skip channels (MODE_FP) are
        // requested in the same flight.
        constexpr int PF = (WPT == 4 && !POOLED && MODE == MODE_FP) ? 8 : 4;
        // the skip channels of plain FP mode ride in the same flight when they are 16-byte rows: one float4 per (row, quad)
        const int C1q = (MODE == MODE_FP && (C1 & 3) == 0 && (reinterpret_cast<uintptr_t>(a.skip) & 15) == 0 && ((k0pad - C2) & 3) == 0 && (C2 & 3) == 0) ? (k0pad - C2) >> 2 : 0;
        constexpr int SK = 4;                                     // skip quads per thread in flight (R * C1q <= SK * NTH at the model's levels: 16 x 64 and 32 x 16 quads)
        float4 skv[SK];
        const bool skip_fast = C1q > 0 && R * C1q <= SK * NTH;
        if (skip_fast) {
#pragma unroll
            for (int u = 0; u < SK; ++u) {
                const int q = tid + u * NTH, r = q / C1q, part = q - r * C1q;
                const long p = row0 + r;
                skv[u] = (q < R * C1q && p < a.rows && part * 4 < C1) ? *reinterpret_cast<const float4 *>(a.skip + p * C1 + part * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (MODE == MODE_FPX && WPT == 1 && C2 == 256) {
            // Common shape (256-wide features, wave-private tile): lane l owns float4 column l of every row, so the bias and the
            // skip weights are loop invariants, row / column indices need no division, and addresses are 32-bit.  This part is
            // not bit-matched to anything (the first layer is already re-associated), so it uses explicit fmaf chains.
            const float4 bz = *reinterpret_cast<const float4 *>(a.bias0 + lane * 4);
            float4 wv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                wv[t] = t < C1 ? *reinterpret_cast<const float4 *>(a.wskip + (size_t)t * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            // eight rows per trip: 24 independent 16-byte gathers in flight per lane (the coarse features of a batch live in the Infinity
            // Cache / L2, ~1-2 us away; a wave-private tile has nobody else to hide that latency, so the trip count IS the prologue time:
            // 4 trips instead of 8 for a 32-row tile).  The accumulators are not live yet, so the 96 extra registers are free here.
            constexpr int U = (R % 8 == 0) ? 8 : 4;
            for (int r0 = 0; r0 < R; r0 += U) {
                float4 f[U][3];
                float w[U][3];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        w[u][t] = wt[(r0 + u) * 3 + t];
                        f[u][t] = k4[(unsigned)nb[(r0 + u) * 3 + t] * 64u + (unsigned)lane];
                    }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float *s4 = sk + (r0 + u) * 4;
                    const float sx = s4[0], sy = s4[1], sz = s4[2], sw = s4[3];
                    // fixed fmaf order (bias, skip channels, then the three interpolation terms): the generic path below uses the same
                    // chain, so the result does not depend on which tiling a batch size selects
                    float v[4] = {bz.x, bz.y, bz.z, bz.w};
                    const float sv[4] = {sx, sy, sz, sw};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        v[0] = fmaf(sv[t], wv[t].x, v[0]); v[1] = fmaf(sv[t], wv[t].y, v[1]);
                        v[2] = fmaf(sv[t], wv[t].z, v[2]); v[3] = fmaf(sv[t], wv[t].w, v[3]);
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        v[0] = fmaf(w[u][t], f[u][t].x, v[0]); v[1] = fmaf(w[u][t], f[u][t].y, v[1]);
                        v[2] = fmaf(w[u][t], f[u][t].z, v[2]); v[3] = fmaf(w[u][t], f[u][t].w, v[3]);
                    }
                    pa_store4(act + (r0 + u) * stride + lane * 4, fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
                }
            }
        } else
        for (int q0 = tid; q0 < items; q0 += NTH * PF) {
            float4 f[PF][3];
            float w[PF][3];
            int rr[PF], pp[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int q = min(q0 + u * NTH, items - 1);
                rr[u] = q / qpr;
                pp[u] = q - rr[u] * qpr;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    w[u][t] = wt[rr[u] * 3 + t];
                    f[u][t] = k4[(size_t)nb[rr[u] * 3 + t] * qpr + pp[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (q0 + u * NTH >= items) break;
                float v[4];
                if (MODE == MODE_FPX) {  // bias + skip . Wskip + interpolation, ReLU: this IS the first layer's output (linearity of interpolation)
                    const float4 bz = *reinterpret_cast<const float4 *>(a.bias0 + pp[u] * 4);
                    v[0] = bz.x; v[1] = bz.y; v[2] = bz.z; v[3] = bz.w;
                    for (int t = 0; t < C1; ++t) {
                        const float4 wv = *reinterpret_cast<const float4 *>(a.wskip + (size_t)t * C2 + pp[u] * 4);
                        const float xv = sk[rr[u] * 4 + t];
                        v[0] = fmaf(xv, wv.x, v[0]); v[1] = fmaf(xv, wv.y, v[1]); v[2] = fmaf(xv, wv.z, v[2]); v[3] = fmaf(xv, wv.w, v[3]);
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        v[0] = fmaf(w[u][t], f[u][t].x, v[0]); v[1] = fmaf(w[u][t], f[u][t].y, v[1]);
                        v[2] = fmaf(w[u][t], f[u][t].z, v[2]); v[3] = fmaf(w[u][t], f[u][t].w, v[3]);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
                } else {                 // interpolation_cuda_kernel.cu:194: (w0*p0 + w1*p1) + w2*p2
                    v[0] = w[u][0] * f[u][0].x + w[u][1] * f[u][1].x + w[u][2] * f[u][2].x;
                    v[1] = w[u][0] * f[u][0].y + w[u][1] * f[u][1].y + w[u][2] * f[u][2].y;
                    v[2] = w[u][0] * f[u][0].z + w[u][1] * f[u][1].z + w[u][2] * f[u][2].z;
                    v[3] = w[u][0] * f[u][0].w + w[u][1] * f[u][1].w + w[u][2] * f[u][2].w;
                }
                pa_store4(act + rr[u] * stride + pp[u] * 4, v[0], v[1], v[2], v[3]);
            }
        }
        if (MODE == MODE_FP && skip_fast) {
#pragma unroll
            for (int u = 0; u < SK; ++u) {
                const int q = tid + u * NTH, r = q / C1q, part = q - r * C1q;
                if (q < R * C1q) pa_store4(act + r * stride + C2 + part * 4, skv[u].x, skv[u].y, skv[u].z, skv[u].w);
            }
        } else if (MODE == MODE_FP) {
            const int tail = k0pad - C2;  // skip channels (patch_aug_net.py:359: cat([interpolated, skip])) + zero padding
            for (int q = tid; q < R * tail; q += NTH) {
                const int r = q / tail, ch = q - r * tail;
                const long p = row0 + r;
                act[r * stride + C2 + ch] = (p < a.rows && ch < C1) ? a.skip[p * C1 + ch] : 0.f;
            }
        }
    }
}

// ---- epilogues of the operand-swapped layout (weights = MFMA A operand): a lane holds channels 16ct + 4(l/16) + r of point 16rt + l%16
// last layer, plain: out = residual + act(acc + bias), 16-byte row segments (VEC: out / residual rows are 16-byte aligned)
template <int RT, int NC, bool VEC>
__device__ __forceinline__ void store_rows(float *__restrict__ out, int ldo, long row0, long rows, const PaLayer &L, int c0, int lane,
                                            floatx4 (&acc)[RT][NC], int relu, const float *__restrict__ residual, int ldr)
{
    const float floor_v = relu ? 0.f : -INFINITY;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        const int col = (c0 + ct) * 16 + (lane >> 4) * 4;
        const float4 bias = *reinterpret_cast<const float4 *>(L.bias + col);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const long row = row0 + rt * 16 + (lane & 15);
            if (row >= rows) continue;
            float4 v = make_float4(fmaxf(acc[rt][ct][0] + bias.x, floor_v), fmaxf(acc[rt][ct][1] + bias.y, floor_v),
                                   fmaxf(acc[rt][ct][2] + bias.z, floor_v), fmaxf(acc[rt][ct][3] + bias.w, floor_v));
            if (VEC) {
                if (residual) {
                    const float4 rr = *reinterpret_cast<const float4 *>(residual + row * ldr + col);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                *reinterpret_cast<float4 *>(out + row * ldo + col) = v;
            } else {
                float *o = out + row * ldo + col;
                const float *rr = residual ? residual + row * ldr + col : nullptr;
                o[0] = v.x + (rr ? rr[0] : 0.f); o[1] = v.y + (rr ? rr[1] : 0.f); o[2] = v.z + (rr ? rr[2] : 0.f); o[3] = v.w + (rr ? rr[3] : 0.f);
            }
        }
    }
}

// last layer, pooled: rows are neighbour-major (row = slot*4 + group), so a lane's point 16rt + l%16 belongs to group l%4: the max
// over a group's neighbours is a max across row tiles (registers) and across the lanes l%16 = g, g+4, g+8, g+12 (two DPP row
// rotations); then bias + ReLU (both monotone, so the order is exact) and one 16-byte store per group and channel quad.
template <int RT, int NC, bool VEC>
__device__ __forceinline__ void store_pooled(float *__restrict__ out, int ldo, long group0, long groups, const PaLayer &L, int c0, int lane,
                                              floatx4 (&acc)[RT][NC])
{
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        floatx4 m = acc[0][ct];
#pragma unroll
        for (int rt = 1; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m[r] = fmaxf(m[r], acc[rt][ct][r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            m[r] = fmaxf(m[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m[r]), 0x120 + 4, 0xf, 0xf, true)));   // row_ror:4
            m[r] = fmaxf(m[r], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m[r]), 0x120 + 8, 0xf, 0xf, true)));   // row_ror:8
        }
        const int g = lane & 15;
        if (g < 4 && group0 + g < groups) {
            const int col = (c0 + ct) * 16 + (lane >> 4) * 4;
            const float4 bias = *reinterpret_cast<const float4 *>(L.bias + col);
            const float4 v = make_float4(fmaxf(m[0] + bias.x, 0.f), fmaxf(m[1] + bias.y, 0.f), fmaxf(m[2] + bias.z, 0.f), fmaxf(m[3] + bias.w, 0.f));
            float *o = out + (group0 + g) * ldo + col;
            if (VEC) *reinterpret_cast<float4 *>(o) = v;
            else { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
        }
    }
}

}  // namespace
