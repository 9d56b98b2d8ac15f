// Finest feature-propagation level (pa_fp_chain_premul with an xyz skip and two 256 -> 256 layers left) in HALF-K passes: 9 KB of LDS per wavefront.
//
// Same function and the same arithmetic as chain_kernel<1, 16, MODE_FPX, false, 1> (pa_chain_kernel.h): 3-NN interpolation of the pre-multiplied
// coarse features + skip . Wskip + bias, ReLU (patch_aug_net.py:350-362 with the first layer folded through the interpolation), then two
// 1x1 conv + folded BatchNorm + ReLU layers (pt_util.py:16-41) on v_mfma_f32_16x16x4_f32 with the contraction index ASCENDING -- bit-identical
// results, so which of the two kernels runs is a matter of speed only (tests/test_gpu_chain.py).
//
// Why: the dominant kernel of the step holds its 16 x 256 fp32 activation tile in a wave-private LDS region
// (16.5 KB), so eight wavefronts -- two per SIMD -- fill a CU's 160 KB and the matrix pipe idles whenever both waves of a SIMD are outside their MFMA
// streams (75.8 % busy, profiles/r06_pmc_mfma_util.txt).  A layer's contraction does not need the whole K at once: here the tile is a 16 x 128
// HALF (8.4 KB) that is refilled between the two halves of every contraction,
//     layer A:  prologue(channels 0..127) -> LDS -> 32 k-steps x 16 column tiles;  prologue(128..255) -> LDS -> 32 k-steps x 16 column tiles
//     layer B:  its input (layer A's output after bias + ReLU) STAYS IN REGISTERS (64) and is written to the LDS half tile twice per output half;
//               two passes of 8 column tiles (32 accumulator registers): 164 registers, three wavefronts per SIMD
// and twelve wavefronts in three four-wave workgroups share a CU.  k ascends inside every output element exactly as before (half 0, then half 1).
// What it buys (profiles/r06_fp0_forms_pmc.txt): alone on the chip the launch is as long as the tile kernel's (0.303-0.313 vs 0.305 ms, MFMA busy 0.77 both:
// the texture addresser that feeds the per-wave weight stream is 65-73 % busy in either form), inside the four-stream pipeline the step gains 0.4-1.3 %
// because a workgroup holds 36 KB of LDS instead of 135 KB.
#include <stdlib.h>
#include <type_traits>

#include "pa_common.h"

#include "pa_chain.h"

namespace {

constexpr int FX_STR = 132;                 // floats per LDS row of the R x 128 half tile: 4 (mod 64) -> the B-fragment read (row l % 16, k 4 ks + l / 16) is conflict-free; rows 16-byte aligned
__host__ __device__ constexpr int fx_wave_floats(int rt) { return 16 * rt * FX_STR + 16 * rt * 10; }      // half tile + neighbour rows (3), interpolation weights (3), skip channels (4) per point

// 32 k-steps (one K half, k-steps ks0 .. ks0 + 31 of the packed matrix) x NQ 64-column groups x RT row tiles: acc[rt][4 q + j] += W^T fragment x
// activation fragment.  Operand ring of two k-steps: the weight fragments (one 16-byte buffer load per column group: lane offset in a VGPR, k-step
// in the scalar offset) and the activation fragments (one 4-byte LDS read per row tile) of k-step ks + 2 are requested right behind the MFMAs of k-step ks.
template <int RT, int NQ>
__device__ __forceinline__ void fx_pass(const float *__restrict__ ap, const __amdgpu_buffer_rsrc_t rsrc, const unsigned (&voff)[NQ], int ks0, floatx4 (&acc)[RT][4 * NQ])
{
    u32x4 bq[2][NQ];
    float aq[2][RT];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) bq[u][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[q], (ks0 + u) * 1024, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) aq[u][rt] = ap[rt * 16 * FX_STR + 4 * u];
    }
#pragma unroll 1
    for (int ks = 0; ks < 32; ks += 2) {       // kept ROLLED: fully unrolled, hipcc 7.2 sinks every refill to its use (s_waitcnt vmcnt(0) per 4 MFMAs)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int nx = min(ks + u + 2, 31);
            float an[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) an[rt] = ap[rt * 16 * FX_STR + 4 * nx];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    acc[rt][4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(bq[u][q].x), aq[u][rt], acc[rt][4 * q + 0], 0, 0, 0);
                    acc[rt][4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(bq[u][q].y), aq[u][rt], acc[rt][4 * q + 1], 0, 0, 0);
                    acc[rt][4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(bq[u][q].z), aq[u][rt], acc[rt][4 * q + 2], 0, 0, 0);
                    acc[rt][4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(bq[u][q].w), aq[u][rt], acc[rt][4 * q + 3], 0, 0, 0);
                }
                bq[u][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[q], (ks0 + nx) * 1024, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) aq[u][rt] = an[rt];
        }
    }
}

// RT row tiles of 16 points per wavefront (a weight fragment fetched from L2 serves RT MFMAs: the 16-row form streams 512 KB of weights per tile,
// 14 TB/s at B = 32, and is bound by exactly that); W wavefronts per workgroup; MINW = wavefronts per SIMD the register allocation must allow
// (hipcc's second __launch_bounds__ argument); U1: rows per gather trip of the SECOND half prologue (layer A's accumulators are live there)
template <int RT, int W, int MINW, int U1>
__global__ __launch_bounds__(W * 64, MINW) void fpx32_kernel(PaChain a)
{
    constexpr int R = 16 * RT, WAVE_FLOATS = fx_wave_floats(RT);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    const long nblk = gridDim.x;
    const long blk = (a.xcd_remap && (nblk & 7) == 0) ? (long)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;      // contiguous tile ranges per XCD (chain_kernel)
    const long tile = blk * W + wave;
    const long ntiles = (a.rows + R - 1) / R;
    if (tile >= ntiles) return;                          // wave-uniform; the kernel has no workgroup barrier
    float *buf = smem + (size_t)wave * WAVE_FLOATS;
    int *nb = reinterpret_cast<int *>(buf + R * FX_STR);
    float *wt = reinterpret_cast<float *>(nb + 3 * R), *sk = wt + 3 * R;
    const long row0 = tile * R;
#ifndef FX_STAMP_TILES
#define FX_STAMP_TILES 512
#endif
#define FX_STAMP(i) do { if (a.dbg && tile < FX_STAMP_TILES && lane == 0) a.dbg[tile * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
    FX_STAMP(0);
    if (a.dbg && tile < FX_STAMP_TILES && lane == 0) a.dbg[tile * 8 + 5] = (long long)__builtin_amdgcn_s_memrealtime();      // constant 100 MHz beside the shader-clock stamps
    // ---- the tile's neighbour rows, weights and skip channels (pa_chain.h chain_prologue, MODE_FPX)
    for (int q = lane; q < 3 * R; q += 64) {
        const int r = q / 3;
        const long p = row0 + r;
        int nbv = 0;
        float wv = 0.f;
        if (p < a.rows) {
            nbv = (int)((p / a.n_unknown) * a.m_known + a.idx3[p * 3 + (q - r * 3)]);
            wv = a.w3[p * 3 + (q - r * 3)];
        }
        nb[q] = nbv;
        wt[q] = wv;
    }
    for (int q = lane; q < 4 * R; q += 64) {
        const int r = q >> 2, t = q & 3;
        const long p = row0 + r;
        sk[q] = (p < a.rows && t < a.c1) ? a.skip[p * a.c1 + t] : 0.f;
    }
    lds_fence();
    const float4 *k4 = reinterpret_cast<const float4 *>(a.known);
    const int c4 = lane & 31, rpar = lane >> 5;          // half prologue: lane owns float4 column c4 of the half, rows rpar, rpar + 2, ...
    // first-layer output (bias + skip . Wskip + interpolation, ReLU) of channels 128 kh + 4 c4 .. + 3 for the tile's rows -> the LDS half tile.
    // The fmaf chain is chain_prologue's (bias, the skip channels, then the three interpolation terms): identical values.
    auto prologue_half = [&](int kh, auto ucount) {
        constexpr int U = decltype(ucount)::value;
        const int chan = 128 * kh + 4 * c4;
        const float4 bz = *reinterpret_cast<const float4 *>(a.bias0 + chan);
        float4 wv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wv[t] = t < a.c1 ? *reinterpret_cast<const float4 *>(a.wskip + (size_t)t * 256 + chan) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int u0 = 0; u0 < R / 2; u0 += U) {
            float4 f[U][3];
            float w[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = rpar + 2 * (u0 + u);
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    w[u][t] = wt[r * 3 + t];
                    f[u][t] = k4[(unsigned)nb[r * 3 + t] * 64u + (unsigned)(32 * kh + c4)];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = rpar + 2 * (u0 + u);
                const float *s4 = sk + r * 4;
                const float sv[4] = {s4[0], s4[1], s4[2], s4[3]};
                float v[4] = {bz.x, bz.y, bz.z, bz.w};
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
                *reinterpret_cast<float4 *>(buf + r * FX_STR + 4 * c4) = make_float4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
            }
        }
    };
    const float *ap = buf + li * FX_STR + lq;             // activation fragment (MFMA B operand): point 16 rt + l % 16, channel 4 ks + l / 16 of the half
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.L[0].wp), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.L[1].wp), 0, 0x7fffffff, 0x00020000);

    // ---- layer A: 256 -> 256, all 16 column tiles, K in two halves through the LDS half tile
    floatx4 h2[RT][16];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) h2[rt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
    {
        unsigned voff[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) voff[q] = ((unsigned)q * 64u * 64u + (unsigned)lane) * 16u;
        prologue_half(0, std::integral_constant<int, 8>());
        lds_fence();
        FX_STAMP(1);
        fx_pass<RT, 4>(ap, rsA, voff, 0, h2);
        lds_fence();                                      // every fragment read of half 0 has landed before the tile is refilled
        prologue_half(1, std::integral_constant<int, U1>());
        lds_fence();
        fx_pass<RT, 4>(ap, rsA, voff, 32, h2);
    }
    // bias + ReLU in registers: lane (i, q) holds channels 16 ct + 4 q + r of point 16 rt + i
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) {
        const float4 bz = *reinterpret_cast<const float4 *>(a.L[0].bias + 16 * ct + 4 * lq);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            h2[rt][ct][0] = fmaxf(h2[rt][ct][0] + bz.x, 0.f); h2[rt][ct][1] = fmaxf(h2[rt][ct][1] + bz.y, 0.f);
            h2[rt][ct][2] = fmaxf(h2[rt][ct][2] + bz.z, 0.f); h2[rt][ct][3] = fmaxf(h2[rt][ct][3] + bz.w, 0.f);
        }
    }
    FX_STAMP(2);
    // ---- layer B: 256 -> 256 in two output halves of 8 column tiles; its input goes register -> LDS half tile -> B fragments, half by half
    const float floor_v = a.relu_last ? 0.f : -INFINITY;
#pragma unroll 1
    for (int nh = 0; nh < 2; ++nh) {
        floatx4 acc[RT][8];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[rt][c] = (floatx4){0.f, 0.f, 0.f, 0.f};
        unsigned voff[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) voff[q] = ((unsigned)(2 * nh + q) * 64u * 64u + (unsigned)lane) * 16u;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            lds_fence();                                  // the previous pass's fragment reads / the previous half's row copy have landed
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    *reinterpret_cast<float4 *>(buf + (rt * 16 + li) * FX_STR + 16 * c + 4 * lq) =
                        make_float4(h2[rt][8 * kh + c][0], h2[rt][8 * kh + c][1], h2[rt][8 * kh + c][2], h2[rt][8 * kh + c][3]);
            lds_fence();
            fx_pass<RT, 2>(ap, rsB, voff, 32 * kh, acc);
        }
        // output half: bias + activation, staged through the half tile, out as whole 512-byte row segments (16-byte stores)
        lds_fence();
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 bz = *reinterpret_cast<const float4 *>(a.L[1].bias + 128 * nh + 16 * c + 4 * lq);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                *reinterpret_cast<float4 *>(buf + (rt * 16 + li) * FX_STR + 16 * c + 4 * lq) =
                    make_float4(fmaxf(acc[rt][c][0] + bz.x, floor_v), fmaxf(acc[rt][c][1] + bz.y, floor_v), fmaxf(acc[rt][c][2] + bz.z, floor_v), fmaxf(acc[rt][c][3] + bz.w, floor_v));
        }
        lds_fence();
#pragma unroll
        for (int u = 0; u < R / 2; ++u) {
            const int r = rpar + 2 * u;
            const long row = row0 + r;
            if (row < a.rows)
                *reinterpret_cast<float4 *>(a.out + row * a.ldo + 128 * nh + 4 * c4) = *reinterpret_cast<const float4 *>(buf + r * FX_STR + 4 * c4);
        }
        FX_STAMP(3 + nh);
    }
    if (a.dbg && tile < FX_STAMP_TILES && lane == 0) a.dbg[tile * 8 + 6] = (long long)__builtin_amdgcn_s_memrealtime();
#undef FX_STAMP
}

int g_fpx32 = -1;

}  // namespace

// test / A/B switch: 1 = wherever the shape applies, 0 = never, -1 = the default rule (PA_CHAIN_NO_FPX32 in the environment turns it off)
PA_API void pa_chain_fpx32_enable(int on) { g_fpx32 = on; }

// Takes the launch when it is the finest level's shape: MODE_FPX, c2 = 256, exactly two packed 256 -> 256 layers, 16-byte aligned output rows, no
// residual / tap.  Returns 1 = launched, 0 = not applicable.
int pa_fpx32_try(const PaChain &a, hipStream_t st)
{
    static const bool off = getenv("PA_CHAIN_NO_FPX32") != nullptr;
    if (g_fpx32 == 0 || (g_fpx32 < 0 && off)) return 0;
    if (a.nlayers != 2 || a.c2 != 256 || a.c1 < 1 || a.c1 > 4 || a.L[0].kpad != 256 || a.L[0].n != 256 || a.L[1].kpad != 256 || a.L[1].n != 256 || !a.L[0].wp ||
        !a.L[1].wp || !a.vec_out || a.residual || a.tap || !a.known || !a.wskip || !a.bias0 || (reinterpret_cast<uintptr_t>(a.known) & 15) != 0)
        return 0;
    // Four-wave workgroups of 16-row wave tiles, three per CU (36 KB of LDS each: they start and finish at different times, so their gather / store
    // phases interleave by themselves, and another stream's workgroup fits beside two of them).  Measured and dropped:
    // one twelve-wave workgroup per CU 0.347 ms and two six-wave ones 0.43 ms against 0.305 ms (lock-step phases, the CU drains between workgroups);
    // 32-row wave tiles at two waves per SIMD (half the weight stream) 0.311-0.317 ms; eight-wave workgroups at 128 registers spill; TWO of these
    // workgroups per CU (an exact four rounds at batch 32 instead of 2.67): 0.311 vs 0.296 ms, pipeline 42.5 vs 42.8 k.
    // Where the launch's 0.76-0.78 of the nominal peak goes (tools/probes/fx_phases.py on a stamp build, profiles/r06_fp0_clock.txt): the shader clock under
    // this kernel is 2.16 GHz, not 2.4 (s_memtime against s_memrealtime over every tile: 0.90); a tile takes 211.6 k cycles against 196.6 k of pure MFMA
    // issue for its three co-resident waves (0.93); and 8192 tiles on 3072 wave slots are 2.67 rounds plus the ramp (0.91).
    const long ntiles = (a.rows + 15) / 16;
    const size_t lds = (size_t)4 * fx_wave_floats(1) * 4;
    auto kern = fpx32_kernel<1, 4, 3, 2>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)pa_div_up(ntiles, 4)), dim3(256), lds, st, a);
    return 1;
}
