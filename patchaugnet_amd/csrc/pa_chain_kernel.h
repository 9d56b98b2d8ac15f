// Device templates of the fused shared-MLP chain kernels (see mlp_chain.hip for the design notes and the reference citations).
// Included by one translation unit per instantiation FAMILY (chain_wp_fpx.hip, chain_wp_rows.hip, chain_split_*.hip, chain_pooled.hip), so that
// a family can be edited, re-tuned or compiled with its own flags without touching the register allocation of the others; the host-side
// tiling choice lives in mlp_chain.hip and reaches the families through the pa_chain_launch_* functions declared at the end of this file.
#pragma once
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

template <int RT, int NC, int MODE, bool POOLED, int WPT, bool APOOL = false, int PDMAX = 8>
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
        gemm_chunk<RT, NC, (RT1 ? PA_RT1_PD : WPT == 1 ? 2 : (RT * NC >= 16 ? 4 : PDMAX)), SWAP>(fold ? act + a.c2 : act, a.lds_stride, L, c0, lane, acc);
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
#ifndef PA_CHAIN_MINB            // A/B builds (tools/build_variant.sh): -DPA_CHAIN_MINB=n forces n workgroups per CU for every kernel of one family's translation unit
#define PA_CHAIN_MINB ((POOLED && WPT == 4 && NCMAX == 2) ? 4 : (POOLED && RT <= 5) || (RT == 1 && WPT == 1) ? 2 : 1)
#endif
__global__ __launch_bounds__((RT == 1 && WPT == 1 && !POOLED) ? 512 : 256, PA_CHAIN_MINB) void chain_kernel(PaChain a)
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
        // the four-workgroups-per-CU pooled tiling (NCMAX == 2: <= 128 registers) keeps a 4-deep operand ring; every other tiling up to 8
#ifndef PA_CHAIN_PDMAX          // A/B builds: ring depth of the shared-tile tilings whose k-step is < 16 MFMAs
#define PA_CHAIN_PDMAX 8
#endif
        constexpr int PDM = (POOLED && WPT == 4 && NCMAX == 2) ? 4 : (RT == 1 ? PA_CHAIN_PDMAX : 8);
        if (NCMAX >= 16 && per % 16 == 0) run_layer_chunks<RT, (NCMAX >= 16 ? 16 : NCMAX), MODE, POOLED, WPT, APOOL, PDM>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else if (NCMAX >= 8 && per % 8 == 0) run_layer_chunks<RT, (NCMAX >= 8 ? 8 : NCMAX), MODE, POOLED, WPT, APOOL, PDM>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else if (NCMAX >= 4 && per % 4 == 0) run_layer_chunks<RT, (NCMAX >= 4 ? 4 : NCMAX), MODE, POOLED, WPT, APOOL, PDM>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else if (per % 2 == 0) run_layer_chunks<RT, 2, MODE, POOLED, WPT, APOOL, PDM>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else run_layer_chunks<RT, 1, MODE, POOLED, WPT, APOOL, PDM>(act, a, L, out, residual, l, tile, lane, cb, ce);
        PA_STAMP(2 + l);
    }
#undef PA_STAMP
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

}  // namespace

// ---- one launcher per translation unit (family); every argument is decided by chain_dispatch (mlp_chain.hip)
int pa_chain_launch_wp_fpx(const PaChain &a, int rt, int wpw, long ntiles, hipStream_t st);                  // chain_wp_fpx.hip: wave-private tiles, MODE_FPX (the finest FP level)
int pa_chain_launch_wp_rows(const PaChain &a, int mode, int rt, int wpw, long ntiles, hipStream_t st);       // chain_wp_rows.hip: wave-private tiles, plain / SA / FP
int pa_chain_launch_split_plain(const PaChain &a, int rt, long ntiles, hipStream_t st);                      // chain_split_plain.hip: shared tiles (four waves split the columns), plain rows (pa_linear)
int pa_chain_launch_split_fp(const PaChain &a, int mode, int rt, long ntiles, hipStream_t st);               // chain_split_fp.hip: shared tiles, FP / FPX
int pa_chain_launch_split_sa(const PaChain &a, int rt, bool atomic_pool, long ntiles, hipStream_t st);       // chain_split_sa.hip: shared tiles, SA gather (unpooled / atomic-max epilogue)
int pa_chain_launch_pooled(const PaChain &a, int rt, bool split, int wpw, long ntiles, hipStream_t st);      // chain_pooled.hip: pooled SA tilings; PA_EUNSUPPORTED for an unbuilt nsample range
