// Fused shared-MLP chains on the MFMA pipe (gfx950): gather/interpolate prologue -> up to 3 x (1x1 conv + folded
// BatchNorm + ReLU) -> optional max over the neighbourhood, without ever materialising the (B, C, m, k) tensors.
//
// Replaces, for evaluation, the reference's unfused sequence
//   pointops.grouping x2 + subtract + cat   (libs/pointops/functions/pointops.py:559-570)
//   pt_util.SharedMLP = Conv2d 1x1 (no bias) + BatchNorm2d + ReLU, three separate modules per layer
//                                            (utils/model_util/pt_util.py:16-41, :98-152)
//   F.max_pool2d over the k neighbours       (place_recognition/patch_aug_net/models/patch_aug_net.py:236)
//   pointops.interpolation + cat             (patch_aug_net.py:354-359)
//
// Design (MI355X-first):
//   * activations are POINT-MAJOR (row = point, channels contiguous), so a neighbour gather is one contiguous row;
//   * one WAVEFRONT owns a tile of R = 16*RT rows end to end: its activation tile lives in a wave-private LDS
//     region, every layer is computed with v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fmaf chain) and written
//     back IN PLACE, so there is not a single workgroup barrier in the kernel;
//   * weights are BN-folded and stored K-major (Wt[k][n]); the B fragment of lane l is Wt[k0 + l/16][n0 + l%16],
//     four 64-byte segments straight from L1/L2 into a VGPR, double-buffered one k-step ahead;
//   * LDS row stride = Kpad + 2 floats: the A-fragment read (row l%16, k = k0 + l/16) then hits 32 distinct banks
//     per 32-lane group (conflict-free);
//   * pooled mode orders a wave's rows neighbour-major (row = slot*4 + group): with the 16x16 C/D layout
//     (col = l%16, row = 4*(l/16) + reg) the four accumulator registers of a lane are the four groups and the
//     max over neighbours is a plain v_max across row tiles plus two cross-lane steps -- the widest activation
//     (the last layer's output) never leaves registers.
#include <stdlib.h>
#include <string.h>

#include "pa_common.h"

#include "pa_chain.h"

namespace {

// One column chunk (NC tiles of 16 columns starting at tile c0) of one layer for the wave's RT row tiles.
//
// Operand roles: the WEIGHT fragment is passed as the MFMA's A operand and the activation fragment as its B operand, i.e. the
// instruction computes the transposed tile D[i = channel][j = point].  The fragment values are exactly those of the natural
// order (the A map (i = l%16, k = l/16) and the B map (k = l/16, j = l%16) coincide), but in the C/D layout a lane now holds FOUR
// CONSECUTIVE CHANNELS of ONE point (channel 16ct + 4(l/16) + r, point 16rt + l%16): hidden activations go back to LDS and
// results go to memory as 8/16-byte row segments instead of four scattered 4-byte words per accumulator.
//
// Operand ring: PD register sets; set u holds k-step ks+u and is refilled for k-step ks+u+PD right after its last use, so the
// weight fragments get PD k-steps of latency cover with no register-to-register copies.  PD = 2 when a k-step is >= 32 MFMAs,
// 4 for the column-split tilings whose k-steps are only 8-16 MFMAs.
//
// Weight layouts.  K-major (L.wt): lane l's fragment for column tile ct is Wt[4ks + l/16][16ct + l%16] -- one 4-byte load
// per column tile, 16 VMEM instructions per k-step at NC = 16.  Measured on MI355X every non-MFMA instruction in the stream
// costs the matrix pipe ~6-7 cycles (one wave per SIMD: nobody else fills the slot), and 18 of them per 32 MFMAs held the
// loop at 42-44 cycles per MFMA instead of 32.  Fragment-major packed weights (L.wp, pa_pack_weights):
//     wp[((cg * ksteps + ks) * 64 + l) * 4 + j] = Wt[4ks + l/16][64cg + 16j + l%16]
// put a lane's four fragments of a 64-column group in one 16-byte word: 4 dwordx4 loads per k-step, each wave-load one
// contiguous 1 KB segment (37 cycles per MFMA in the same loop).
// activation fragment x weight fragment; SWAP: the weight fragment is the MFMA's A operand (transposed tile, see gemm_chunk)
template <bool SWAP>
__device__ __forceinline__ floatx4 mfma_ab(float act_frag, float w_frag, floatx4 acc)
{
    return SWAP ? __builtin_amdgcn_mfma_f32_16x16x4f32(w_frag, act_frag, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(act_frag, w_frag, acc, 0, 0, 0);
}

template <int RT, int NC, int PD, bool SWAP>
__device__ __forceinline__ void gemm_chunk(const float *__restrict__ act, int stride, const PaLayer &L, int c0, int lane,
                                            floatx4 (&acc)[RT][NC])
{
    const int ksteps = L.kpad >> 2, last = ksteps - 1;
    const float *ap = act + (lane & 15) * stride + (lane >> 4);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) acc[rt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
    float bq[PD][NC], aq[PD][RT];
    int ks = 0;
    if ((NC % 4 == 0) && L.wp != nullptr) {
        constexpr int NQ = NC / 4 > 0 ? NC / 4 : 1;
        // Packed weights through a buffer descriptor: the per-lane part of the address (lane and 64-column group) is a loop-invariant
        // 32-bit VGPR offset, the k-step is a SCALAR offset, so a weight fetch costs one VMEM instruction and no VALU address math
        // (flat 64-bit addressing cost a v_lshl_add_u64 per load: ~1 extra vector instruction per 8 MFMAs in a stream where every
        // non-MFMA issue delays the matrix pipe).  The descriptor base must be provably wave-uniform: c0 depends on the wave id in the
        // shared-tile variants, which the compiler treats as divergent, hence the readfirstlane.
        const int c0u = __builtin_amdgcn_readfirstlane(c0);
        const float *wbase = L.wp + (size_t)(c0u >> 2) * ksteps * 256;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wbase), 0, 0x7fffffff, 0x00020000);
        unsigned voff[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) voff[q] = ((unsigned)q * (unsigned)ksteps * 64u + (unsigned)lane) * 16u;
        auto load_b = [&](int u, int k) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[q], k * 1024, 0);
                bq[u][4 * q] = __uint_as_float(v.x); bq[u][4 * q + 1] = __uint_as_float(v.y);
                bq[u][4 * q + 2] = __uint_as_float(v.z); bq[u][4 * q + 3] = __uint_as_float(v.w);
            }
        };
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int k = min(u, last);
            load_b(u, k);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) aq[u][rt] = ap[rt * 16 * stride + k * 4];
        }
        // main loop: every refill is in range, so no clamps -- LDS reads become base + immediate offset, weight loads base + scalar offset
        for (; ks + 2 * PD <= ksteps; ks += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int nx = ks + u + PD;
                float an[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) an[rt] = ap[rt * 16 * stride + nx * 4];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            acc[rt][4 * q + c] = mfma_ab<SWAP>(aq[u][rt], bq[u][4 * q + c], acc[rt][4 * q + c]);
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[q], nx * 1024, 0);
                    bq[u][4 * q] = __uint_as_float(v.x); bq[u][4 * q + 1] = __uint_as_float(v.y);
                    bq[u][4 * q + 2] = __uint_as_float(v.z); bq[u][4 * q + 3] = __uint_as_float(v.w);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) aq[u][rt] = an[rt];
            }
        }
        // last full ring pass: refills beyond the end are clamped to the last k-step (loaded, never used)
        for (; ks + PD <= ksteps; ks += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int nx = min(ks + u + PD, last);
                float an[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) an[rt] = ap[rt * 16 * stride + nx * 4];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            acc[rt][4 * q + c] = mfma_ab<SWAP>(aq[u][rt], bq[u][4 * q + c], acc[rt][4 * q + c]);
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[q], nx * 1024, 0);
                    bq[u][4 * q] = __uint_as_float(v.x); bq[u][4 * q + 1] = __uint_as_float(v.y);
                    bq[u][4 * q + 2] = __uint_as_float(v.z); bq[u][4 * q + 3] = __uint_as_float(v.w);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) aq[u][rt] = an[rt];
            }
        }
    } else {
        const float *wp = L.wt + (size_t)(lane >> 4) * L.ldw + c0 * 16 + (lane & 15);
        const size_t wstep = (size_t)4 * L.ldw;
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int k = min(u, last);
#pragma unroll
            for (int ct = 0; ct < NC; ++ct) bq[u][ct] = wp[(size_t)k * wstep + ct * 16];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) aq[u][rt] = ap[rt * 16 * stride + k * 4];
        }
        for (; ks + PD <= ksteps; ks += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int nx = min(ks + u + PD, last);
                const float *wn = wp + (size_t)nx * wstep;
                float an[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) an[rt] = ap[rt * 16 * stride + nx * 4];
#pragma unroll
                for (int ct = 0; ct < NC; ++ct) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        acc[rt][ct] = mfma_ab<SWAP>(aq[u][rt], bq[u][ct], acc[rt][ct]);
                    bq[u][ct] = wn[ct * 16];
                    __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) aq[u][rt] = an[rt];
            }
        }
    }
    // tail: the sets that still hold valid (not clamped) k-steps
#pragma unroll
    for (int u = 0; u < PD - 1; ++u) {
        if (ks + u < ksteps) {
#pragma unroll
            for (int ct = 0; ct < NC; ++ct)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc[rt][ct] = mfma_ab<SWAP>(aq[u][rt], bq[u][ct], acc[rt][ct]);
        }
    }
}

// ---- natural operand order (activation = A operand): a lane holds 4 ROWS x 1 column per accumulator.  Kept for the wave-private
// plain-row kernels, where hipcc 7.2 turns the operand-swapped loop's loop-carried vmcnt(7) into vmcnt(0) (the weight prefetch then
// no longer overlaps the MFMAs: 45 instead of 38 cycles per MFMA at fp0).
// hidden layer: bias + ReLU, written back in place as the next layer's A tile
template <int RT, int NC, bool ADD>
__device__ __forceinline__ void store_hidden_nat(float *act, int stride, const PaLayer &L, int c0, int lane, floatx4 (&acc)[RT][NC])
{
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        const int col = (c0 + ct) * 16 + (lane & 15);
        const float bias = L.bias[col];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rt * 16 + (lane >> 4) * 4 + r;
                float *d = act + row * stride + col;
                *d = fmaxf(acc[rt][ct][r] + bias + (ADD ? *d : 0.f), 0.f);
            }
    }
}

// last layer, plain: bias + ReLU to global memory (row-major, ldo)
template <int RT, int NC>
__device__ __forceinline__ void store_rows_nat(float *__restrict__ out, int ldo, long row0, long rows, const PaLayer &L, int c0, int lane,
                                            floatx4 (&acc)[RT][NC], int relu, const float *__restrict__ residual, int ldr)
{
    const float floor_v = relu ? 0.f : -INFINITY;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        const int col = (c0 + ct) * 16 + (lane & 15);
        const float bias = L.bias[col];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = row0 + rt * 16 + (lane >> 4) * 4 + r;
                if (row < rows) {
                    float v = fmaxf(acc[rt][ct][r] + bias, floor_v);
                    if (residual) v = residual[row * ldr + col] + v;
                    out[row * ldo + col] = v;
                }
            }
    }
}

// last layer, plain, via LDS: the accumulator layout gives a lane 4-byte pieces of 64-byte row segments (128 store
// instructions per lane for a 32 x 256 tile -- store-issue bound).  Writing the tile to the (now dead) activation region
// and reading it back row-major turns that into 16-byte stores of whole contiguous rows, 4x fewer and fully coalesced.
template <int RT, int NC>
__device__ __forceinline__ void stage_rows_lds_nat(float *act, int ostride, const PaLayer &L, int c0, int lane, floatx4 (&acc)[RT][NC], int relu)
{
    const float floor_v = relu ? 0.f : -INFINITY;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        const int col = (c0 + ct) * 16 + (lane & 15);
        const float bias = L.bias[col];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) act[(rt * 16 + (lane >> 4) * 4 + r) * ostride + col] = fmaxf(acc[rt][ct][r] + bias, floor_v);
    }
}

// hidden layer: bias + ReLU, written back in place as the next layer's activation tile (row stride is even: 8-byte stores)
template <int RT, int NC, bool ADD>
__device__ __forceinline__ void store_hidden(float *act, int stride, const PaLayer &L, int c0, int lane, floatx4 (&acc)[RT][NC])
{
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        const int col = (c0 + ct) * 16 + (lane >> 4) * 4;
        const float4 bias = *reinterpret_cast<const float4 *>(L.bias + col);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float2 *d = reinterpret_cast<float2 *>(act + (rt * 16 + (lane & 15)) * stride + col);
            float2 lo = make_float2(0.f, 0.f), hi = make_float2(0.f, 0.f);
            if (ADD) { lo = d[0]; hi = d[1]; }   // folded first layer: the interpolated term sits where the result goes
            d[0] = make_float2(fmaxf(acc[rt][ct][0] + bias.x + lo.x, 0.f), fmaxf(acc[rt][ct][1] + bias.y + lo.y, 0.f));
            d[1] = make_float2(fmaxf(acc[rt][ct][2] + bias.z + hi.x, 0.f), fmaxf(acc[rt][ct][3] + bias.w + hi.y, 0.f));
        }
    }
}

// last layer, plain, staged through LDS: the tile goes to the (now dead) activation region with 8-byte stores and comes back
// row-major, so that global memory sees whole contiguous rows (1 KB per wave-store at 256 columns) instead of 64-byte segments.
template <int RT, int NC>
__device__ __forceinline__ void stage_rows_lds(float *act, int ostride, const PaLayer &L, int c0, int lane, floatx4 (&acc)[RT][NC], int relu)
{
    const float floor_v = relu ? 0.f : -INFINITY;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        const int col = (c0 + ct) * 16 + (lane >> 4) * 4;
        const float4 bias = *reinterpret_cast<const float4 *>(L.bias + col);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float2 *d = reinterpret_cast<float2 *>(act + (rt * 16 + (lane & 15)) * ostride + col);
            d[0] = make_float2(fmaxf(acc[rt][ct][0] + bias.x, floor_v), fmaxf(acc[rt][ct][1] + bias.y, floor_v));
            d[1] = make_float2(fmaxf(acc[rt][ct][2] + bias.z, floor_v), fmaxf(acc[rt][ct][3] + bias.w, floor_v));
        }
    }
}

template <int R>
__device__ __forceinline__ void copy_rows_out(const float *act, int ostride, int n, float *__restrict__ out, int ldo, long row0, long rows,
                                              const float *__restrict__ residual, int ldr, int tid, int nth)
{
    const int qpr = n >> 2;
    for (int q = tid; q < R * qpr; q += nth) {
        const int r = q / qpr, part = q - r * qpr;
        const long row = row0 + r;
        if (row >= rows) continue;
        float4 v = *reinterpret_cast<const float4 *>(act + r * ostride + part * 4);
        if (residual) {
            const float4 rr = *reinterpret_cast<const float4 *>(residual + row * ldr + part * 4);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        *reinterpret_cast<float4 *>(out + row * ldo + part * 4) = v;
    }
}

// hidden tile (rows x n, LDS row stride even) -> global rows, 8-byte pieces (the activation stride is even, not a multiple of 4)
template <int R>
__device__ __forceinline__ void copy_rows_tap(const float *act, int stride, int n, float *__restrict__ out, int ldo, long row0, long rows, int tid, int nth)
{
    const int hpr = n >> 1;
    for (int q = tid; q < R * hpr; q += nth) {
        const int r = q / hpr, part = q - r * hpr;
        const long row = row0 + r;
        if (row >= rows) continue;
        *reinterpret_cast<float2 *>(out + row * ldo + part * 2) = *reinterpret_cast<const float2 *>(act + r * stride + part * 2);
    }
}

// last layer of an UNPOOLED set-abstraction tiling with the max over the neighbourhood folded in (operand-swapped layout: a lane holds channels
// col .. col + 3 of point 16 rt + l % 16).  The 16 points of a row tile belong to at most two groups of `ns` consecutive rows (ns >= 16): the max
// over each group's points is a masked DPP-row reduction, and lanes 0 / 1 of every DPP row fold the two results into out[group][channel] with
// an integer atomicMax on the float bit pattern -- exact and order-independent because the values are >= 0 after the ReLU (out is zero-filled by
// the launcher).  Replaces writing the (groups * ns, C) tensor + the rowgroup_max pass over it.
template <int RT, int NC>
__device__ __forceinline__ void store_group_max_atomic(float *__restrict__ out, int ldo, long row0, long total_rows, int ns, const PaLayer &L, int c0, int lane,
                                                         floatx4 (&acc)[RT][NC])
{
    const int i = lane & 15;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const long r0 = row0 + rt * 16;
        if (r0 >= total_rows) continue;                                  // wave-uniform
        const long gA = r0 / ns;
        const int split = (int)min((gA + 1) * (long)ns - r0, 16L);       // rows of this tile in group gA; the rest (if any) belong to gA + 1
        const bool valid = r0 + i < total_rows, inA = i < split;
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) {
            const int col = (c0 + ct) * 16 + (lane >> 4) * 4;
            const float4 bias = *reinterpret_cast<const float4 *>(L.bias + col);
            const float v[4] = {fmaxf(acc[rt][ct][0] + bias.x, 0.f), fmaxf(acc[rt][ct][1] + bias.y, 0.f), fmaxf(acc[rt][ct][2] + bias.z, 0.f),
                                fmaxf(acc[rt][ct][3] + bias.w, 0.f)};
            float ma[4], mb[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = (valid && inA) ? v[c] : 0.f, b = (valid && !inA) ? v[c] : 0.f;
                a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x120 + 8, 0xf, 0xf, true)));   // row_ror:8
                a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x120 + 4, 0xf, 0xf, true)));
                a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x120 + 2, 0xf, 0xf, true)));
                a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x120 + 1, 0xf, 0xf, true)));
                b = fmaxf(b, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x120 + 8, 0xf, 0xf, true)));
                b = fmaxf(b, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x120 + 4, 0xf, 0xf, true)));
                b = fmaxf(b, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x120 + 2, 0xf, 0xf, true)));
                b = fmaxf(b, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x120 + 1, 0xf, 0xf, true)));
                ma[c] = a;
                mb[c] = b;
            }
            if (i == 0) {
                int *o = reinterpret_cast<int *>(out + gA * ldo + col);
#pragma unroll
                for (int c = 0; c < 4; ++c) atomicMax(o + c, __float_as_int(ma[c]));
            } else if (i == 1 && split < 16 && r0 + split < total_rows) {
                int *o = reinterpret_cast<int *>(out + (gA + 1) * ldo + col);
#pragma unroll
                for (int c = 0; c < 4; ++c) atomicMax(o + c, __float_as_int(mb[c]));
            }
        }
    }
}

template <int RT, int NC, int MODE, bool POOLED, int WPT, bool APOOL = false>
__device__ __forceinline__ void run_layer_chunks(float *act, const PaChain &a, const PaLayer &L, float *out, const float *residual, int l, long tile,
                                                 int lane, int c_begin, int c_end)
{
#ifndef PA_RT1_SWAP
#define PA_RT1_SWAP 0
#endif
#ifndef PA_RT1_PD
#define PA_RT1_PD 2
#endif
    constexpr bool RT1 = RT == 1 && WPT == 1 && !POOLED;                        // the eight-wave 16-row variant (finest FP level)
    constexpr bool SWAP = POOLED || WPT > 1 || (RT1 && PA_RT1_SWAP);   // see the note above store_hidden_nat
    const bool last = (l == a.nlayers - 1);
    for (int c0 = c_begin; c0 < c_end; c0 += NC) {
        floatx4 acc[RT][NC];
        const bool fold = MODE == MODE_FP && l == 0 && a.fold0;     // layer 0 contracts the skip columns only and adds the interpolated term
        gemm_chunk<RT, NC, (RT1 ? PA_RT1_PD : WPT == 1 ? 2 : (RT * NC >= 16 ? 4 : 8)), SWAP>(fold ? act + a.c2 : act, a.lds_stride, L, c0, lane, acc);
        if (!last) {
            tile_sync<WPT>();  // every A read of this layer has landed before its rows are overwritten (single chunk per wave: host-checked)
            if (fold) {
                if (SWAP) store_hidden<RT, NC, true>(act, a.lds_stride, L, c0, lane, acc);
                else store_hidden_nat<RT, NC, true>(act, a.lds_stride, L, c0, lane, acc);
            } else if (SWAP) store_hidden<RT, NC, false>(act, a.lds_stride, L, c0, lane, acc);
            else store_hidden_nat<RT, NC, false>(act, a.lds_stride, L, c0, lane, acc);
        } else if (APOOL) {
            store_group_max_atomic<RT, NC>(out, a.ldo, tile * (RT * 16), a.rows * a.ns, a.ns, L, c0, lane, acc);
        } else if (POOLED) {
            if (a.vec_out) store_pooled<RT, NC, true>(out, a.ldo, tile * 4, a.rows, L, c0, lane, acc);
            else store_pooled<RT, NC, false>(out, a.ldo, tile * 4, a.rows, L, c0, lane, acc);
        } else if (!APOOL && a.ep_stride > 0) {   // host guarantees a single chunk per wave here
            tile_sync<WPT>();           // every A read of the last layer has landed: the activation tile is dead
            if (SWAP) stage_rows_lds<RT, NC>(act, a.ep_stride, L, c0, lane, acc, a.relu_last);
            else stage_rows_lds_nat<RT, NC>(act, a.ep_stride, L, c0, lane, acc, a.relu_last);
        } else {
            const long total_rows = (MODE == MODE_SA) ? a.rows * a.ns : a.rows;
            if (!SWAP) store_rows_nat<RT, NC>(out, a.ldo, tile * (RT * 16), total_rows, L, c0, lane, acc, a.relu_last, residual, a.ldr);
            else if (a.vec_out) store_rows<RT, NC, true>(out, a.ldo, tile * (RT * 16), total_rows, L, c0, lane, acc, a.relu_last, residual, a.ldr);
            else store_rows<RT, NC, false>(out, a.ldo, tile * (RT * 16), total_rows, L, c0, lane, acc, a.relu_last, residual, a.ldr);
        }
    }
    if (!last) {
        tile_sync<WPT>();
        if (!POOLED && a.tap && l == a.nlayers - 2) {       // second output: this layer's result, straight from the tile the next layer reads
            const long total_rows = (MODE == MODE_SA) ? a.rows * a.ns : a.rows;
            copy_rows_tap<RT * 16>(act, a.lds_stride, L.n, a.tap, a.ldtap, tile * (RT * 16), total_rows, WPT == 1 ? lane : (int)threadIdx.x, WPT * 64);
        }
    } else if (!POOLED && !APOOL && a.ep_stride > 0) {
        tile_sync<WPT>();
        const long total_rows = (MODE == MODE_SA) ? a.rows * a.ns : a.rows;
        copy_rows_out<RT * 16>(act, a.ep_stride, L.n, out, a.ldo, tile * (RT * 16), total_rows, residual, a.ldr,
                               WPT == 1 ? lane : (int)threadIdx.x, WPT * 64);
    }
}

// Pooled wave-private kernels (the set-abstraction levels) are gather-latency bound in their prologue: keep two waves per SIMD
// (<= 256 registers) there; the plain row kernels trade occupancy for their 128 accumulator registers.
// The 16-row wave-private variant (RT == 1, unpooled) runs EIGHT waves per workgroup, two per SIMD: while one wave of a SIMD gathers its
// next tile or stores its last one, the other keeps the matrix pipe busy.
template <int RT, int NCMAX, int MODE, bool POOLED, int WPT, bool APOOL = false>
__global__ __launch_bounds__((RT == 1 && WPT == 1 && !POOLED) ? 512 : 256, (POOLED && RT <= 5) || (RT == 1 && WPT == 1) ? 2 : 1) void chain_kernel(PaChain a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int R = RT * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware order: workgroup b is dispatched to XCD b % 8 (observed round-robin), and each XCD has its own 4 MB L2.  Giving
    // every XCD one CONTIGUOUS eighth of the tiles (= a few whole clouds) keeps the rows its gathers touch (1 MB of coarse features
    // per cloud at fp0) inside that L2 instead of spreading every cloud over all eight.  Purely a performance mapping.
    const long nblk = gridDim.x;
    const long blk = (a.xcd_remap && (nblk & 7) == 0) ? (long)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    const long tile = WPT == 1 ? blk * (blockDim.x >> 6) + wave : blk;
    const long total_rows = (MODE == MODE_SA) ? a.rows * a.ns : a.rows;
    const long ntiles = POOLED ? (a.rows + 3) / 4 : (total_rows + R - 1) / R;
    if (tile >= ntiles) return;  // WPT == 1: wave-uniform, and that variant has no workgroup barrier; WPT == 4: grid == ntiles
    float *act = smem + (WPT == 1 ? (size_t)wave * a.wave_floats : (size_t)0);
    const int tid = WPT == 1 ? lane : (int)threadIdx.x;  // prologue work is spread over the tile's owner(s)
    const int stride = a.lds_stride;
    const int k0pad = (MODE == MODE_FP && a.fold0) ? a.c2 + a.L[0].kpad : a.L[0].kpad;
#define PA_STAMP(i) do { if (a.dbg && tile < 512 && lane == 0 && (WPT == 1 || wave == 0)) a.dbg[tile * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
    PA_STAMP(0);
    // Two waves share each SIMD in the eight-wave variant and every tile costs the same, so without help both would sit in their
    // prologue / epilogue at the same moments and leave the matrix pipe idle together.  Delaying the second wave of each SIMD by about
    // half a tile once, at the start, puts the pair in anti-phase for the rest of the launch: one gathers or stores while the other
    // multiplies.  Pure scheduling; results unchanged.
    if (RT == 1 && WPT == 1 && !POOLED && wave >= 4)
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);

    chain_prologue<float, R, MODE, POOLED, WPT>(act, act + R * stride, a, tile, tid, lane, stride, k0pad);
    tile_sync<WPT>();
    PA_STAMP(1);

    // ---------------------------------------------------------------- layers
    for (int l = 0; l < a.nlayers; ++l) {
        PaLayer L = a.L[l];
        float *out = a.out;
        const float *residual = a.residual;
        if (MODE == MODE_PLAIN && WPT == 4) pa_col_slice(a, L, out, residual);
        const int nct = L.n >> 4;
        const int per = WPT == 1 ? nct : nct / WPT;          // column tiles this wave computes (host: nct % WPT == 0)
        const int cb = WPT == 1 ? 0 : wave * per, ce = cb + per;
        if (NCMAX >= 16 && per % 16 == 0) run_layer_chunks<RT, (NCMAX >= 16 ? 16 : NCMAX), MODE, POOLED, WPT, APOOL>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else if (NCMAX >= 8 && per % 8 == 0) run_layer_chunks<RT, (NCMAX >= 8 ? 8 : NCMAX), MODE, POOLED, WPT, APOOL>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else if (NCMAX >= 4 && per % 4 == 0) run_layer_chunks<RT, (NCMAX >= 4 ? 4 : NCMAX), MODE, POOLED, WPT, APOOL>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else if (per % 2 == 0) run_layer_chunks<RT, 2, MODE, POOLED, WPT, APOOL>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else run_layer_chunks<RT, 1, MODE, POOLED, WPT, APOOL>(act, a, L, out, residual, l, tile, lane, cb, ce);
        PA_STAMP(2 + l);
    }
#undef PA_STAMP
}

// max over groups of `ns` consecutive rows: out[g][c] = max_s in[g*ns + s][c]   (patch_aug_net.py:236 for the unfused SA level)
__global__ __launch_bounds__(256) void rowgroup_max_kernel(long groups, int ns, int c, const float *__restrict__ in, float *__restrict__ out)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= groups * c) return;
    const long g = t / c;
    const int ch = (int)(t - g * c);
    const float *p = in + (g * ns) * c + ch;
    float m = p[0];
    for (int s = 1; s < ns; ++s) m = fmaxf(m, p[(size_t)s * c]);
    out[t] = m;
}

template <int RT, int NCMAX, int MODE, bool POOLED, int WPT, bool APOOL = false>
int launch_chain(const PaChain &a, int waves_per_wg, long ntiles, hipStream_t st)
{
    const size_t lds = (size_t)(WPT == 1 ? waves_per_wg : 1) * a.wave_floats * 4;
    auto kern = chain_kernel<RT, NCMAX, MODE, POOLED, WPT, APOOL>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (WPT == 1) hipLaunchKernelGGL(kern, dim3(pa_div_up(ntiles, waves_per_wg)), dim3(64 * waves_per_wg), lds, st, a);
    else hipLaunchKernelGGL(kern, dim3(ntiles, a.col_slices > 1 ? a.col_slices : 1), dim3(256), lds, st, a);
    return 0;
}

template <int MODE>
void launch_rows(const PaChain &a, int rt, bool split, int wpw, long ntiles, hipStream_t st)
{
    if (!split && rt == 1) {
        if (MODE == MODE_FPX || MODE == MODE_FP || MODE == MODE_PLAIN) launch_chain<1, 16, MODE, false, 1>(a, wpw, ntiles, st);
        else launch_chain<2, 16, MODE, false, 1>(a, wpw, ntiles, st);
    } else if (!split) launch_chain<2, 16, MODE, false, 1>(a, wpw, ntiles, st);
    else if (rt == 2) launch_chain<2, 8, MODE, false, 4>(a, 4, ntiles, st);
    else launch_chain<1, 8, MODE, false, 4>(a, 4, ntiles, st);
}

}  // namespace

// Generic entry point.  mode: 0 plain rows, 1 set-abstraction gather, 2 feature-propagation interpolate.
// wt[l] is K-major (kpad[l] x n[l]) with BN folded in and zero rows beyond the true K; bias[l] has n[l] entries.
int pa_chain16_launch(PaChain &a, int mode, bool is_pooled, bool split, int RTv, int scratch_floats, hipStream_t st);   // mlp_chain_f16.hip
bool pa_sa_tiny_applies(const PaChain &a, int rt);                                                                      // sa_tiny.hip
int pa_sa_tiny_launch(const PaChain &a, int rt, long ntiles, hipStream_t st);
int pa_linear_lds_try(long rows, int k, int n, const float *x, int ldx, const float *wt, const float *bias, int relu, const float *residual, int ldr,
                      float *out, int ldo, hipStream_t st);                                                                    // linear_lds.hip

static int g_chain_tiny = -1;
// test / A/B switch for the persistent first-level kernel (sa_tiny.hip): 1 = wherever its shape applies, 0 = never, -1 = the default rule
PA_API void pa_chain_tiny_enable(int on) { g_chain_tiny = on; }

static long long *g_chain_dbg = nullptr;
// profiling hook (tools/chain_phases.py): device buffer of 512 x 8 int64 receiving s_memtime stamps of the next launches; NULL = off
PA_API void pa_chain_debug_buffer(long long *buf) { g_chain_dbg = buf; }

static int chain_dispatch(int mode, int pooled, int nlayers, const float *const *wt, const float *const *wpk, const float *const *bias, const int *kpad, const int *nout,
                          long rows, int k0,
                          const float *x, int ldx,
                          const float *xyz, const float *feat, const int *center_idx, const int *nbr_idx, int n_src, int m_ctr, int ns, int c_feat,
                          const float *known, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                          float *out, int ldo, int relu_last, const float *residual, int ldr, pa_stream_t stream,
                          const float *wskip = nullptr, const float *bias0 = nullptr, const void *const *wp16 = nullptr, int fold0 = 0, int col_slices = 1,
                          float *tap = nullptr, int ldtap = 0)
{
    PA_REQUIRE(nlayers >= 1 && nlayers <= 3, "pa_mlp_chain: nlayers=%d must be 1..3", nlayers);
    PA_REQUIRE(rows > 0 && k0 > 0 && out, "pa_mlp_chain: rows/k0 must be positive and out non-null");
    PaChain a;
    memset(&a, 0, sizeof(a));
    a.nlayers = nlayers;
    int kin = k0, maxk = 0;
    for (int l = 0; l < nlayers; ++l) {
        PA_REQUIRE(wt[l] && bias[l], "pa_mlp_chain: null weights for layer %d", l);
        PA_REQUIRE(kpad[l] % 4 == 0 && kpad[l] >= kin && kpad[l] < kin + 4, "pa_mlp_chain: layer %d kpad=%d must be k=%d rounded up to 4", l, kpad[l], kin);
        PA_REQUIRE(nout[l] % 16 == 0 && nout[l] > 0, "pa_mlp_chain: layer %d n=%d must be a positive multiple of 16", l, nout[l]);
        a.L[l].wt = wt[l]; a.L[l].bias = bias[l]; a.L[l].kpad = kpad[l]; a.L[l].n = nout[l]; a.L[l].ldw = nout[l];
        static const bool no_packed = getenv("PA_CHAIN_NO_PACKED") != nullptr;   // A/B knob
        a.L[l].wp = (!no_packed && wpk && wpk[l] && nout[l] % 64 == 0) ? wpk[l] : nullptr;
        a.L[l].wp16 = wp16 ? reinterpret_cast<const _Float16 *>(wp16[l]) : nullptr;
        if (kpad[l] > maxk) maxk = kpad[l];
        kin = nout[l];
    }
    if (fold0) {   // MODE_FP with a folded first layer: the tile holds [interpolated term (c2) | skip channels (kpad[0])]
        PA_REQUIRE(mode == MODE_FP && nlayers >= 2 && c2 == nout[0] && k0 == c1, "pa_fp_chain_premul: folded first layer needs >= 2 layers, c2 == nout[0]");
        if (c2 + kpad[0] > maxk) maxk = c2 + kpad[0];
    }
    a.fold0 = fold0;
    a.rows = rows; a.k0 = k0; a.lds_stride = maxk + 2;
    a.x = x; a.ldx = ldx;
    a.xyz = xyz; a.feat = feat; a.center_idx = center_idx; a.nbr_idx = nbr_idx; a.n_src = n_src; a.m_ctr = m_ctr; a.ns = ns; a.c_feat = c_feat;
    a.known = known; a.idx3 = idx3; a.w3 = w3; a.skip = skip; a.n_unknown = n_unknown; a.m_known = m_known; a.c2 = c2; a.c1 = c1;
    a.wskip = wskip; a.bias0 = bias0;
    a.out = out; a.ldo = ldo;
    a.tap = tap; a.ldtap = ldtap;
    PA_REQUIRE(tap == nullptr || (nlayers >= 2 && pooled == 0 && !wp16 && ldtap % 2 == 0 && ldtap >= nout[nlayers - 2] && nout[nlayers - 2] % 2 == 0 &&
                                  ((uintptr_t)tap & 7) == 0), "pa_mlp_chain: the tap output needs >= 2 fp32 layers, an unpooled chain and 8-byte aligned rows");
    a.relu_last = relu_last; a.residual = residual; a.ldr = ldr;
    a.dbg = g_chain_dbg;
    if (col_slices > 1) {   // pa_linear over few rows: nout[0] is the slice width
        PA_REQUIRE(nlayers == 1 && mode == MODE_PLAIN && nout[0] % 64 == 0 && (a.L[0].wp || a.L[0].wp16), "pa_linear: column slices need one packed layer");
        a.col_slices = col_slices; a.slice_n = nout[0];
        a.wp_slice = (long)nout[0] * kpad[0];
        a.L[0].ldw = nout[0] * col_slices;
    }
    static const bool no_xcd = getenv("PA_CHAIN_NO_XCD_REMAP") != nullptr;   // A/B knob
    a.xcd_remap = no_xcd ? 0 : 1;
    hipStream_t st = (hipStream_t)stream;

    // pooled = 2: the UNPOOLED shared-tile tiling with the max over the neighbourhood folded into the last layer's epilogue by atomics
    // (store_group_max_atomic): out is (groups, n_last) and is zero-filled here
    const bool atomic_pool = pooled == 2;
    const bool is_pooled = pooled != 0 && !atomic_pool;
    PA_REQUIRE(pooled == 0 || mode == MODE_SA, "pa_mlp_chain: pooled output needs mode 1 (set-abstraction gather)");
    PA_REQUIRE(!atomic_pool || (ns >= 16 && !wp16 && residual == nullptr && tap == nullptr), "pa_mlp_chain: the atomic pooled epilogue needs nsample >= 16, fp32, no residual / tap");
    const long total_rows = (mode == MODE_SA) ? rows * ns : rows;
    // Tiling choice.  Wave-private tiles (WPT = 1) need >= ~2 tiles per SIMD to hide latency; with fewer row tiles the
    // four waves of a workgroup share one tile and split the columns (WPT = 4), optionally with 16-row tiles.
    bool split = false;
    int RTv = is_pooled ? (ns + 3) / 4 : 2;
    bool can_split = true;
    for (int l = 0; l < nlayers; ++l) {  // every layer splits 4 ways; a hidden layer must stay one chunk (<= 8 tiles, power of two) per wave
        const int per = nout[l] / 64;
        can_split = can_split && (nout[l] % 64 == 0) && (l + 1 == nlayers || (per <= 8 && (per & (per - 1)) == 0));
    }
    static const long split_below = getenv("PA_CHAIN_SPLIT_BELOW") ? atol(getenv("PA_CHAIN_SPLIT_BELOW")) : 2048;   // tuning knob
    if (!is_pooled) {
        const long t32 = (total_rows + 31) / 32;
        static const long rt2_above = getenv("PA_CHAIN_SPLIT_RT2_ABOVE") ? atol(getenv("PA_CHAIN_SPLIT_RT2_ABOVE")) : 512;          // tuning knob
        if (can_split && t32 < split_below) { split = true; RTv = t32 >= rt2_above ? 2 : 1; }
        // Measured and dropped (round 1): 128-row tiles shared by four waves for all-256-wide chains (slower than wave-private tiles
        // once the weights are packed) and 16-row wave-private tiles with eight waves per workgroup (no gain over 32-row tiles).
    } else {
        const long tp = (rows + 3) / 4;
        if (can_split && tp < 2048 && RTv == 5) split = true;
    }
    // Finest FP level (and any other wave-private FPX launch): 16-row tiles with EIGHT waves per workgroup, two per SIMD -- with the lean
    // buffer-load k-loop the partner wave's MFMAs fill the slots one wave leaves while it gathers / stores (0.332 -> 0.307 ms at B = 32).
    static const bool fpx_rt2 = getenv("PA_CHAIN_FPX_RT2") != nullptr;   // A/B knob: the former 32-row, one-wave-per-SIMD tiling
    static const int stagger_env = getenv("PA_CHAIN_STAGGER") ? atoi(getenv("PA_CHAIN_STAGGER")) : 0;
    // The same tiling for the mid-sized plain / FP launches (32 768 rows at B = 32: fp1 and the pre-multiplies): 2048 wave tiles = one
    // eight-wave workgroup per CU in a single round, instead of 1024 four-wave shared-tile workgroups of which three fit a CU at a time.
    // Measured at B = 32: the plain 32 768-row pre-multiply 0.056 -> 0.050 ms; the FP-mode launches get SLOWER (fp1 0.072 -> 0.120 ms: their
    // 3-NN interpolation prologue weighs more against only K = 64 + 256 of MFMA work, and K = 320 leaves room for seven waves), so FP mode
    // keeps the shared-tile variant unless PA_CHAIN_RT1_FP is set.
    static const long rt1_min_rows = getenv("PA_CHAIN_RT1_MIN_ROWS") ? atol(getenv("PA_CHAIN_RT1_MIN_ROWS")) : 30000;
    static const bool rt1_fp = getenv("PA_CHAIN_RT1_FP") != nullptr;
    bool rt1 = !fpx_rt2 && mode == MODE_FPX && !split && !is_pooled && !wp16;
    if (!rt1 && rt1_min_rows > 0 && !is_pooled && !wp16 && col_slices <= 1 && (mode == MODE_PLAIN || (mode == MODE_FP && rt1_fp)) && total_rows >= rt1_min_rows) {
        bool ok = true;
        for (int l = 0; l + 1 < nlayers; ++l) ok = ok && nout[l] / 16 <= 16 && ((nout[l] / 16) & (nout[l] / 16 - 1)) == 0;
        if (ok) { rt1 = true; split = false; }
    }
    if (rt1) { RTv = 1; a.stagger = stagger_env; }
    PA_REQUIRE(col_slices <= 1 || split, "pa_linear: column slices are built for the shared-tile (few rows) variant");
    const int R = RTv * 16;
    const int ncmax = split ? 8 : (is_pooled ? 4 : 16);
    if (!split)
        for (int l = 0; l + 1 < nlayers; ++l)  // hidden layers are written back in place => must be a single column chunk
            PA_REQUIRE(nout[l] / 16 <= ncmax && ((nout[l] / 16) & (nout[l] / 16 - 1)) == 0,
                       "pa_mlp_chain: hidden layer %d with n=%d must be 16*2^j <= %d (single column chunk, written back in place)", l, nout[l], ncmax * 16);
    // LDS-staged epilogue: needs the last layer to be one chunk per wave (its A tile is dead when the results are staged), the staged
    // tile to fit the activation region (or 40 KB), and 16-byte aligned rows in the output / residual.
    int ep_floats = 0;
    if (!is_pooled && !atomic_pool) {
        const int nl = nout[nlayers - 1];
        const int per = split ? nl / 64 : nl / 16;
        const int nc = (ncmax >= 16 && per % 16 == 0) ? 16 : (ncmax >= 8 && per % 8 == 0) ? 8 : (ncmax >= 4 && per % 4 == 0) ? 4 : (per % 2 == 0) ? 2 : 1;
        const int ostride = nl + 4;   // multiple of 4: the row-major read-back uses 16-byte LDS loads
        static const bool ep_off = getenv("PA_CHAIN_NO_LDS_EPILOGUE") != nullptr;
        if (!ep_off && nc == per && (ostride <= a.lds_stride || (size_t)R * ostride * 4 <= 40 * 1024) && ldo % 4 == 0 && ((uintptr_t)out & 15) == 0 &&
            (residual == nullptr || (ldr % 4 == 0 && ((uintptr_t)residual & 15) == 0))) {
            a.ep_stride = ostride;
            ep_floats = R * ostride;
        }
    }
    a.vec_out = (ldo % 4 == 0 && ((uintptr_t)out & 15) == 0 && (residual == nullptr || (ldr % 4 == 0 && ((uintptr_t)residual & 15) == 0))) ? 1 : 0;
    int scratch = 0;
    if (mode == MODE_PLAIN) {
        PA_REQUIRE(x && ldx >= k0, "pa_mlp_chain: plain mode needs x and ldx >= k0");
    } else if (mode == MODE_SA) {
        PA_REQUIRE(xyz && feat && center_idx && nbr_idx && n_src > 0 && m_ctr > 0 && ns > 0 && c_feat > 0, "pa_mlp_chain: SA mode arguments");
        PA_REQUIRE(k0 == 3 + c_feat, "pa_mlp_chain: SA mode k0=%d must be 3 + c_feat=%d", k0, c_feat);
        PA_REQUIRE(rows % m_ctr == 0, "pa_mlp_chain: SA rows=%ld must be B*m", rows);
        PA_REQUIRE((rows / m_ctr) * (long)n_src < 2147483647L, "pa_mlp_chain: B*n_src overflows int32 row ids");
        scratch = 2 * R;
    } else if (mode == MODE_FP) {
        PA_REQUIRE(known && idx3 && w3 && n_unknown > 0 && m_known > 0 && c2 > 0 && c1 >= 0 && (c1 == 0 || skip), "pa_mlp_chain: FP mode arguments");
        PA_REQUIRE(c2 % 4 == 0 && (fold0 ? k0 == c1 : k0 == c2 + c1), "pa_mlp_chain: FP mode needs c2 %% 4 == 0 and k0 == c2 + c1");
        PA_REQUIRE(rows % n_unknown == 0, "pa_mlp_chain: FP rows=%ld must be B*n", rows);
        scratch = 6 * R;
    } else if (mode == MODE_FPX) {
        PA_REQUIRE(known && idx3 && w3 && skip && wskip && bias0 && n_unknown > 0 && m_known > 0, "pa_fp_chain_premul: null argument");
        PA_REQUIRE(c2 % 4 == 0 && c1 >= 1 && c1 <= 4 && k0 == c2, "pa_fp_chain_premul: needs c2 %% 4 == 0, 1 <= c1 <= 4 (c2=%d c1=%d)", c2, c1);
        PA_REQUIRE(rows % n_unknown == 0, "pa_fp_chain_premul: rows=%ld must be B*n", rows);
        scratch = 10 * R;
    } else {
        PA_REQUIRE(false, "pa_mlp_chain: unknown mode %d", mode);
    }
    if (wp16) {   // fp16-operand variant (mlp_chain_f16.hip): same tiling decisions, its own LDS geometry
        PA_REQUIRE(is_pooled || (split ? RTv != 8 : RTv == 2), "pa_mlp_chain(fp16): tiling variant not built");
        return pa_chain16_launch(a, mode, is_pooled, split, RTv, scratch, st);
    }
    a.wave_floats = ((R * a.lds_stride + scratch + 3) / 4) * 4;
    if (ep_floats > a.wave_floats) a.wave_floats = ep_floats;
    const size_t per_wave = (size_t)a.wave_floats * 4;
    PA_REQUIRE(per_wave <= 156 * 1024, "pa_mlp_chain: one tile needs %zu B of LDS (> 156 KiB); reduce K", per_wave);
    // waves per workgroup of the wave-private tilings.  Smaller workgroups leave LDS granules in which OTHER kernels' workgroups
    // (kNN: 70 KB, 3-NN, FPS) can become resident next to a chain workgroup when several streams are in flight.
    static const int wpw_env = getenv("PA_CHAIN_WPW") ? atoi(getenv("PA_CHAIN_WPW")) : 0;
    int wpw = wpw_env > 0 ? wpw_env : (rt1 ? 8 : 4);
    while (wpw > 1 && wpw * per_wave > 156 * 1024) wpw = rt1 ? wpw - 1 : wpw >> 1;
    const long ntiles = is_pooled ? (rows + 3) / 4 : (total_rows + R - 1) / R;

    // the finest set-abstraction level of both models (<= 8 -> 32 -> 32 -> 64): persistent workgroups with the weights resident in LDS and the
    // gather pipelined across tiles (sa_tiny.hip); bit-identical to the generic pooled kernel, so which one runs is a matter of speed only
    static const bool no_tiny = getenv("PA_CHAIN_NO_TINY") != nullptr;                                               // A/B knob
    static const long tiny_min = getenv("PA_CHAIN_TINY_MIN_TILES") ? atol(getenv("PA_CHAIN_TINY_MIN_TILES")) : 1024;  // tuning knob
    const bool tiny_on = g_chain_tiny < 0 ? (!no_tiny && ntiles >= tiny_min) : g_chain_tiny > 0;
    if (is_pooled && !split && tiny_on && pa_sa_tiny_applies(a, RTv)) {
        pa_sa_tiny_launch(a, RTv, ntiles, st);
    } else if (is_pooled) {
        if (split) launch_chain<5, 4, MODE_SA, true, 4>(a, 4, ntiles, st);
        else switch (RTv) {
            case 4: launch_chain<4, 4, MODE_SA, true, 1>(a, wpw, ntiles, st); break;
            case 5: launch_chain<5, 4, MODE_SA, true, 1>(a, wpw, ntiles, st); break;
            case 8: launch_chain<8, 4, MODE_SA, true, 1>(a, wpw, ntiles, st); break;
            default:
                pa_set_error("pa_mlp_chain: pooled tiling is built for nsample in (13..16], (17..20], (29..32]; got %d", ns);
                return PA_EUNSUPPORTED;
        }
    } else if (atomic_pool) {
        PA_REQUIRE(split && (RTv == 1 || RTv == 2), "pa_mlp_chain: the atomic pooled epilogue is built for the shared-tile tilings (rows=%ld)", total_rows);
        if (hipMemsetAsync(out, 0, (size_t)rows * ldo * sizeof(float), st) != hipSuccess) { pa_set_error("pa_mlp_chain: hipMemsetAsync failed"); return PA_EINVAL; }
        if (RTv == 2) launch_chain<2, 8, MODE_SA, false, 4, true>(a, 4, ntiles, st);
        else launch_chain<1, 8, MODE_SA, false, 4, true>(a, 4, ntiles, st);
    } else if (mode == MODE_PLAIN) launch_rows<MODE_PLAIN>(a, RTv, split, wpw, ntiles, st);
    else if (mode == MODE_SA) launch_rows<MODE_SA>(a, RTv, split, wpw, ntiles, st);
    else if (mode == MODE_FP) launch_rows<MODE_FP>(a, RTv, split, wpw, ntiles, st);
    else launch_rows<MODE_FPX>(a, RTv, split, wpw, ntiles, st);
    PA_CHECK_LAUNCH("pa_mlp_chain");
    return PA_OK;
}

PA_API int pa_mlp_chain(int mode, int pooled, int nlayers, const float *const *wt, const float *const *bias, const int *kpad, const int *nout,
                        long rows, int k0,
                        const float *x, int ldx,
                        const float *xyz, const float *feat, const int *center_idx, const int *nbr_idx, int n_src, int m_ctr, int ns, int c_feat,
                        const float *known, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                        float *out, int ldo, pa_stream_t stream)
{
    return chain_dispatch(mode, pooled, nlayers, wt, nullptr, bias, kpad, nout, rows, k0, x, ldx, xyz, feat, center_idx, nbr_idx, n_src, m_ctr, ns, c_feat,
                          known, idx3, w3, skip, n_unknown, m_known, c2, c1, out, ldo, 1, nullptr, 0, stream);
}

// Same, with fragment-major packed copies of the weights (wpk[l] from pa_pack_weights, or NULL for a layer to use wt[l]).
PA_API int pa_mlp_chain_packed(int mode, int pooled, int nlayers, const float *const *wt, const float *const *wpk, const float *const *bias,
                               const int *kpad, const int *nout, long rows, int k0,
                               const float *x, int ldx,
                               const float *xyz, const float *feat, const int *center_idx, const int *nbr_idx, int n_src, int m_ctr, int ns, int c_feat,
                               const float *known, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                               float *out, int ldo, pa_stream_t stream)
{
    return chain_dispatch(mode, pooled, nlayers, wt, wpk, bias, kpad, nout, rows, k0, x, ldx, xyz, feat, center_idx, nbr_idx, n_src, m_ctr, ns, c_feat,
                          known, idx3, w3, skip, n_unknown, m_known, c2, c1, out, ldo, 1, nullptr, 0, stream);
}

// Column slices for a single layer over few rows: with 16-row tiles a (512 x 512) @ (512 x 1024) layer is 32 workgroups, each streaming
// the whole 2 MB weight matrix through one CU; slicing the outputs over gridDim.y spreads that stream over the chip.
static int linear_col_slices(long rows, int n, bool packed)
{
    static const long target = getenv("PA_LINEAR_SLICE_BLOCKS") ? atol(getenv("PA_LINEAR_SLICE_BLOCKS")) : 512;
    static const int min_n = getenv("PA_LINEAR_SLICE_MIN") ? atoi(getenv("PA_LINEAR_SLICE_MIN")) : 128;
    const long t16 = (rows + 15) / 16;
    if (!packed || t16 >= 512 || n % 64 != 0) return 1;
    int s = 1;
    while (t16 * s < target && n % (2 * s) == 0 && (n / (2 * s)) % 64 == 0 && n / (2 * s) >= min_n) s *= 2;
    return s;
}

// One dense layer on point-major rows: out = residual + act(x Wt + bias); wt K-major (kpad x n), kpad = k rounded up to 4;
// wpk: optional packed copy of wt (NULL = none).
PA_API int pa_linear(long rows, int k, int n, const float *x, int ldx, const float *wt, const float *wpk, const float *bias, int relu,
                     const float *residual, int ldr, float *out, int ldo, pa_stream_t stream)
{
    PA_REQUIRE(residual == nullptr || ldr >= n, "pa_linear: residual row stride %d < n=%d", ldr, n);
    if (ldx >= k && ldo >= n && pa_linear_lds_try(rows, k, n, x, ldx, wt, bias, relu, residual, ldr, out, ldo, (hipStream_t)stream)) {
        PA_CHECK_LAUNCH("pa_linear (LDS-resident weights)");
        return 0;
    }
    const int kpad = (k + 3) / 4 * 4;
    static const bool no_packed = getenv("PA_CHAIN_NO_PACKED") != nullptr;
    const int slices = linear_col_slices(rows, n, wpk != nullptr && !no_packed);
    const int ns = n / slices;
    return chain_dispatch(0, 0, 1, &wt, &wpk, &bias, &kpad, &ns, rows, k, x, ldx, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0,
                          nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, out, ldo, relu, residual, ldr, stream, nullptr, nullptr, nullptr, 0, slices);
}

// Feature propagation with the first layer folded into the prologue.  Interpolation is linear, so
//     relu(W1 [interp(f); skip] + b1) = relu(interp(W1a f) + W1b skip + b1):
// g = known features already multiplied by W1a (one pa_linear over the B*m known rows instead of a 259-wide layer over the B*n
// unknown rows, n/m = 4 at fp0); this kernel interpolates g, adds the skip term on the VALU (c1 <= 4 channels: xyz) and the bias,
// applies ReLU, and runs the REMAINING layers (wt/wpk/bias/kpad/nout describe layers 2..) on the MFMA pipe.
// c1 <= 4 (the finest level: skip = xyz) is the case above.  c1 > 4 (coarser levels: skip = encoder features): the first layer stays in
// the chain as a c1-wide contraction over the skip channels whose output gets the interpolated term added; it is built here from wskip
// (c1 x c2 K-major, c1 % 4 == 0, optional packed copy) and bias0, in front of the caller's nlayers (<= 2) remaining layers.
static int fp_premul_dispatch(int nlayers, const float *const *wt, const float *const *wpk, const void *const *wp16, const float *const *bias,
                              const int *kpad, const int *nout, long rows, const float *g, const int *idx3, const float *w3, const float *skip,
                              int n_unknown, int m_known, int c2, int c1, const float *wskip, const float *wskip_p, const void *wskip16,
                              const float *bias0, float *out, int ldo, pa_stream_t stream, float *tap = nullptr, int ldtap = 0, int relu_last = 1)
{
    if (c1 <= 4)
        return chain_dispatch(MODE_FPX, 0, nlayers, wt, wpk, bias, kpad, nout, rows, c2, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0,
                              g, idx3, w3, skip, n_unknown, m_known, c2, c1, out, ldo, relu_last, nullptr, 0, stream, wskip, bias0, wp16, 0, 1, tap, ldtap);
    PA_REQUIRE(nlayers >= 1 && nlayers <= 2 && c1 % 4 == 0 && wskip && bias0, "pa_fp_chain_premul: c1=%d > 4 needs c1 %% 4 == 0 and at most 2 remaining layers", c1);
    const float *wt2[3] = {wskip, wt[0], nlayers > 1 ? wt[1] : nullptr};
    const float *wp2[3] = {wskip_p, wpk ? wpk[0] : nullptr, (wpk && nlayers > 1) ? wpk[1] : nullptr};
    const void *w162[3] = {wskip16, wp16 ? wp16[0] : nullptr, (wp16 && nlayers > 1) ? wp16[1] : nullptr};
    const float *b2[3] = {bias0, bias[0], nlayers > 1 ? bias[1] : nullptr};
    const int k2[3] = {c1, kpad[0], nlayers > 1 ? kpad[1] : 0}, n2[3] = {c2, nout[0], nlayers > 1 ? nout[1] : 0};
    return chain_dispatch(MODE_FP, 0, nlayers + 1, wt2, wp2, b2, k2, n2, rows, c1, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0,
                          g, idx3, w3, skip, n_unknown, m_known, c2, c1, out, ldo, relu_last, nullptr, 0, stream, nullptr, nullptr, wp16 ? w162 : nullptr, 1, 1,
                          tap, ldtap);
}

PA_API int pa_fp_chain_premul(int nlayers, const float *const *wt, const float *const *wpk, const float *const *bias, const int *kpad, const int *nout,
                              long rows, const float *g, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2,
                              int c1, const float *wskip, const float *wskip_p, const float *bias0, float *out, int ldo, pa_stream_t stream)
{
    return fp_premul_dispatch(nlayers, wt, wpk, nullptr, bias, kpad, nout, rows, g, idx3, w3, skip, n_unknown, m_known, c2, c1, wskip, wskip_p, nullptr,
                              bias0, out, ldo, stream);
}

// pa_fp_chain_premul with a fused tail: the chain's LAST layer (wt[nlayers - 1], relu_last = 0, zero bias) is the NEXT finer level's pre-multiply
// applied to this level's output, which is the result of layer nlayers - 2 and leaves through `tap` (ldtap) -- one launch instead of the
// chain + a pa_linear over the same rows, and the tile is contracted while it is still in LDS.
PA_API int pa_fp_chain_premul_tap(int nlayers, const float *const *wt, const float *const *wpk, const float *const *bias, const int *kpad, const int *nout,
                                  long rows, const float *g, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2,
                                  int c1, const float *wskip, const float *wskip_p, const float *bias0, float *out, int ldo, float *tap, int ldtap,
                                  int relu_last, pa_stream_t stream)
{
    PA_REQUIRE(tap, "pa_fp_chain_premul_tap: null tap");
    return fp_premul_dispatch(nlayers, wt, wpk, nullptr, bias, kpad, nout, rows, g, idx3, w3, skip, n_unknown, m_known, c2, c1, wskip, wskip_p, nullptr,
                              bias0, out, ldo, stream, tap, ldtap, relu_last);
}

// ---- fp16-operand variants (BASELINE configs[4], opt-in): same arguments, wp16[l] = pa_pack_weights_f16 of layer l -----------
PA_API int pa_mlp_chain_f16(int mode, int pooled, int nlayers, const float *const *wt, const void *const *wp16, const float *const *bias,
                            const int *kpad, const int *nout, long rows, int k0,
                            const float *x, int ldx,
                            const float *xyz, const float *feat, const int *center_idx, const int *nbr_idx, int n_src, int m_ctr, int ns, int c_feat,
                            const float *known, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                            float *out, int ldo, pa_stream_t stream)
{
    PA_REQUIRE(wp16, "pa_mlp_chain_f16: null wp16");
    return chain_dispatch(mode, pooled, nlayers, wt, nullptr, bias, kpad, nout, rows, k0, x, ldx, xyz, feat, center_idx, nbr_idx, n_src, m_ctr, ns, c_feat,
                          known, idx3, w3, skip, n_unknown, m_known, c2, c1, out, ldo, 1, nullptr, 0, stream, nullptr, nullptr, wp16);
}

PA_API int pa_linear_f16(long rows, int k, int n, const float *x, int ldx, const float *wt, const void *wp16, const float *bias, int relu,
                         const float *residual, int ldr, float *out, int ldo, pa_stream_t stream)
{
    PA_REQUIRE(wp16, "pa_linear_f16: null wp16");
    PA_REQUIRE(residual == nullptr || ldr >= n, "pa_linear_f16: residual row stride %d < n=%d", ldr, n);
    const int kpad = (k + 3) / 4 * 4;
    const int slices = linear_col_slices(rows, n, true);
    const int ns = n / slices;
    return chain_dispatch(0, 0, 1, &wt, nullptr, &bias, &kpad, &ns, rows, k, x, ldx, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0,
                          nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, out, ldo, relu, residual, ldr, stream, nullptr, nullptr, &wp16, 0, slices);
}

PA_API int pa_fp_chain_premul_f16(int nlayers, const float *const *wt, const void *const *wp16, const float *const *bias, const int *kpad,
                                  const int *nout, long rows, const float *g, const int *idx3, const float *w3, const float *skip, int n_unknown,
                                  int m_known, int c2, int c1, const float *wskip, const void *wskip16, const float *bias0, float *out, int ldo,
                                  pa_stream_t stream)
{
    PA_REQUIRE(wp16, "pa_fp_chain_premul_f16: null wp16");
    PA_REQUIRE(c1 <= 4 || wskip16, "pa_fp_chain_premul_f16: c1 > 4 needs the fp16 packing of wskip");
    return fp_premul_dispatch(nlayers, wt, nullptr, wp16, bias, kpad, nout, rows, g, idx3, w3, skip, n_unknown, m_known, c2, c1, wskip, nullptr, wskip16,
                              bias0, out, ldo, stream);
}

namespace {
// wp[((cg * ksteps + ks) * 64 + l) * 4 + j] = wt[(4ks + l/16) * n + 64cg + 16j + l%16]
__global__ void pack_weights_kernel(int kpad, int n, const float *__restrict__ wt, float *__restrict__ wp)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)kpad * n) return;
    const int j = (int)(t & 3), l = (int)((t >> 2) & 63);
    const long rest = t >> 8;
    const int ksteps = kpad >> 2;
    const int ks = (int)(rest % ksteps), cg = (int)(rest / ksteps);
    wp[t] = wt[(size_t)(4 * ks + (l >> 4)) * n + 64 * cg + 16 * j + (l & 15)];
}
}  // namespace

// Fragment-major packing of a K-major (kpad x n) weight matrix for the MFMA chain kernels; n % 64 == 0, kpad % 4 == 0.
PA_API int pa_pack_weights(int kpad, int n, const float *wt, float *wp, pa_stream_t stream)
{
    PA_REQUIRE(kpad > 0 && kpad % 4 == 0 && n > 0 && n % 64 == 0 && wt && wp, "pa_pack_weights: need kpad %% 4 == 0, n %% 64 == 0 (kpad=%d n=%d)", kpad, n);
    const long total = (long)kpad * n;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(pa_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, kpad, n, wt, wp);
    PA_CHECK_LAUNCH("pa_pack_weights");
    return PA_OK;
}

PA_API int pa_rowgroup_max(long groups, int ns, int c, const float *in, float *out, pa_stream_t stream)
{
    PA_REQUIRE(groups > 0 && ns > 0 && c > 0 && in && out, "pa_rowgroup_max: bad arguments");
    hipLaunchKernelGGL(rowgroup_max_kernel, dim3(pa_div_up(groups * c, 256)), dim3(256), 0, (hipStream_t)stream, groups, ns, c, in, out);
    PA_CHECK_LAUNCH("pa_rowgroup_max");
    return PA_OK;
}
