// Finest feature-propagation level of the fp16 path (BASELINE.json configs[4]; patch_aug_net.py:350-362 / pptnet.py FP level 0 after
// the first layer went through the interpolation: pa_fp_chain_premul_f16 with c1 <= 4 and two remaining 256 -> 256 layers)
//     h1 = relu(interp(g) + skip . Wskip + b0);  h2 = relu(h1 W2 + b2);  out = relu(h2 W3 + b3)
// with the WEIGHTS SHARED THROUGH LDS and the activations in REGISTERS.
//
// Why: the wave-private LDS-tile kernel (chain16_kernel<2, 16, FPX>) streams the layer's 128 KB of fp16 fragments per 32-row tile and
// wave; PMC (profiles/r04_pptnet_f16_pmc_tcp.txt) has the vector L1 at its peak request rate and the MFMA pipe 11 % busy.  Here a
// workgroup's waves run the layers in lock-step: a slab of KSB k-steps x 16 column tiles goes global -> LDS ONCE per workgroup
// (global_load_lds_dwordx4, no registers), double-buffered, one barrier per slab; a wave reads each 16-byte fragment once for its two
// 16-row tiles.  L1 weight traffic per row falls by the number of waves sharing the slab.
//
// Activations never touch LDS.  With the weights as the MFMA's A operand a lane's accumulators acc[ct][i] are four consecutive output
// channels of point l%16 -- and two of them back to back are exactly one B operand (k slots 8 (l/16) .. + 7) of the next layer IF
// accumulator slot (ct, g = l/16, i) stands for channel
//     r(ct, g, i) = 32 (ct / 2) + 8 g + 4 (ct % 2) + i,
// because then operand slot (ks, g, e) = accumulator (2 ks + e / 4, g, e % 4) = channel 32 ks + 8 g + e: the contraction order the
// standard fp16 fragments (pa_pack_weights_f16) are packed in.  A layer produces its outputs in r-order when its A fragments have
// their 16 columns permuted accordingly; that permutation is applied by the global -> LDS copy (the per-lane SOURCE address of
// global_load_lds is free, the destination is lane-linear), so the kernel takes the SAME packed weights as chain16_kernel and every
// LDS fragment read is lane-linear (conflict-free).  The interpolation prologue builds h1 directly in r-order: for a column-tile pair
// a lane gathers 32 contiguous bytes per neighbour.
//
// The pre-multiplied features g can be fp16 (pa_fp_premul_g16 writes them: the same kernel body with the rows of the coarse level as
// the operand source, one layer, no bias / ReLU): the interpolation gathers are the kernel's largest memory phase (3 KB per row in
// fp32, L2 -> L1 at half-line granularity); in fp16 a lane's column-tile pair is ONE 16-byte load and the table of a cloud is 512 KB.
//
// Arithmetic: interpolation, skip term, biases, ReLU in fp32; operands rounded to fp16 (RNE); v_mfma_f32_16x16x32_f16 with fp32
// accumulation.  Unlike chain16_kernel the interpolated term is not rounded to fp16 before the skip term and bias are added, so the
// result is closer to the fp32 path, not bit-identical to chain16_kernel (tests/test_gpu_f16.py compares both against fp32).
#include <stdlib.h>

#include "pa_common.h"

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));

namespace {

struct Fpx16Args {
    long rows;
    const void *g;        // (b * m_known, 256) features already multiplied by the first layer's interpolated-part weights: fp32, or fp16 (G16)
    const int *idx3;      // (rows, 3) neighbour indices inside the cloud
    const float *w3;      // (rows, 3)
    const float *skip;    // (rows, c1)
    const float *wskip;   // (c1, 256) K-major
    const float *bias0;   // (256)
    const half8 *wq[2];   // pa_pack_weights_f16(256, 256) of the layers
    const float *b[2];
    float *out;
    _Float16 *out16;      // OUT16: the result as fp16 rows of 256 (descriptor-only callers: the map is only read by the fp16 NetVLAD kernel)
    int ldo, n_unknown, m_known, c1, xcd_remap;
    // PREMUL form: g16out[r][:] = fp16(x[r][:256] . W): rows = rows of x
    const float *x;
    int ldx;
    _Float16 *g16out;
    long long *dbg;   // profiling only (pa_chain_debug_buffer): cycle stamps of the first 512 wave tiles
};

__device__ __forceinline__ int r_ofs(int ct, int g) { return 32 * (ct >> 1) + 8 * g + 4 * (ct & 1); }

// NL layers of 256 -> 256 on 32-row wave tiles.  PREMUL: operand = rows of x, one layer, fp16 output without bias / ReLU.
template <int WAVES, int KSB, int NL, bool PREMUL, bool G16, bool DBG, bool OUT16 = false>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 1 : 2) void fpx16_kernel(Fpx16Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fpx16_lds[];
    constexpr int NS = 8 / KSB;             // slabs per layer (K = 256 = 8 k-steps)
    constexpr int SLAB = KSB * 16 * 1024;   // bytes: KSB k-steps x 16 column tiles x 64 lanes x 16 B
    constexpr int PER = KSB * 16 / WAVES;   // 1 KB pieces per wave and slab
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mq = lane & 15, g = lane >> 4;
    float *cst = reinterpret_cast<float *>(fpx16_lds + 2 * SLAB);   // wskip[4][256], bias0[256], b2[256], b3[256]
    const long nblk = gridDim.x;
    const long blk = (a.xcd_remap && (nblk & 7) == 0) ? (long)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    // profiling build only: branch-free stamps (every lane writes the same LDS word), copied out at the end
    unsigned *stamps = reinterpret_cast<unsigned *>(fpx16_lds + 2 * SLAB + 7 * 1024);
#define FPX16_STAMP(i) do { if (DBG) stamps[wave * 8 + (i)] = (unsigned)__builtin_readcyclecounter(); } while (0)
    FPX16_STAMP(0);

    // global -> LDS copy of slab s (the layers numbered through: 0 .. NL NS - 1) into buffer s & 1.  Piece p = (k-step, column tile ct);
    // lane (mq, g) of the piece fetches the chunk the r-ordered fragment needs: standard tile 2 (ct / 2) + (mq >= 8), standard lane
    // 16 g + 8 ((mq / 4) & 1) + 4 (ct % 2) + mq % 4.
    const unsigned lane_src = ((mq >> 3) * 512 + g * 16 + ((mq >> 2) & 1) * 8 + (mq & 3)) * 16u;   // bytes (8 k-steps x 64 lanes x 16 B per tile)
    auto fetch = [&](int s) {
        const char *src = reinterpret_cast<const char *>(a.wq[s / NS]);   // uniform base + uniform piece offset (SGPRs) + one VGPR
        const int ks0 = (s % NS) * KSB;
        unsigned char *dst = fpx16_lds + (s & 1) * SLAB;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int p = u * WAVES + wave, ksl = p >> 4, ct = p & 15;
            const char *piece = src + (size_t)((((ct >> 1) * 16 + ks0 + ksl) * 64 + 4 * (ct & 1)) * 16);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(piece + lane_src),
                                             (void __attribute__((address_space(3))) *)(dst + p * 1024), 16, 0, 0);
        }
    };

    const long row0 = (blk * WAVES + wave) * 32 + mq;
    half8 h[2][8];
    if constexpr (PREMUL) {
        // ---- operand = the rows themselves: lane (mq, g) needs x[row][32 ks + 8 g .. + 7] ---------------------------------------------
        const float *xp[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const long row = row0 + rt * 16, rowc = row < a.rows ? row : a.rows - 1;
            xp[rt] = a.x + (size_t)rowc * a.ldx + 8 * g;
        }
        fetch(0);
        fetch(1);
#pragma unroll
        for (int q = 0; q < 8; q += 4) {   // 16 loads in flight
            __builtin_amdgcn_sched_barrier(0);
            float4 f[2][4][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    f[rt][u][0] = *reinterpret_cast<const float4 *>(xp[rt] + 32 * (q + u));
                    f[rt][u][1] = *reinterpret_cast<const float4 *>(xp[rt] + 32 * (q + u) + 4);
                }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 lo = f[rt][u][0], hi = f[rt][u][1];
                    h[rt][q + u] = (half8){(_Float16)lo.x, (_Float16)lo.y, (_Float16)lo.z, (_Float16)lo.w, (_Float16)hi.x, (_Float16)hi.y, (_Float16)hi.z, (_Float16)hi.w};
                }
        }
        FPX16_STAMP(1);
    } else {
        // ---- h1 in registers: interpolation + skip term + bias, ReLU, fp16 ----------------------------------------------------------
        // the index loads head the dependent chain (index -> row address -> gather): they go out first, the weight copies behind them
        int nb[2][3];
        float wj[2][3], sv[2][4];
        long cloud[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const long row = row0 + rt * 16, rowc = row < a.rows ? row : a.rows - 1;
            cloud[rt] = rowc / a.n_unknown;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                nb[rt][t] = a.idx3[rowc * 3 + t];
                wj[rt][t] = a.w3[rowc * 3 + t];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) sv[rt][t] = t < a.c1 ? a.skip[rowc * a.c1 + t] : 0.f;
        }
        // the constants follow (their LDS stores wait for these loads only), then the weight copies
        constexpr int CPT = (7 * 256 + WAVES * 64 - 1) / (WAVES * 64);
        float cv[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int t = tid + u * WAVES * 64;
            const float *src = t < 1024 ? a.wskip + ((t >> 8) < a.c1 ? t : 0) : t < 1280 ? a.bias0 + (t - 1024) : t < 1536 ? a.b[0] + (t - 1280) : a.b[1] + (t < 1792 ? t - 1536 : 0);
            cv[u] = *src;
            if (t < 1024 && (t >> 8) >= a.c1) cv[u] = 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(0);
        fetch(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int t = tid + u * WAVES * 64;
            if (t < 7 * 256) cst[t] = cv[u];
        }
        // one (column-tile pair, both row tiles) unit: h[rt][p] from its gathered values
        auto finish = [&](int p, const float (&f)[2][3][8]) {
            const int c = 32 * p + 8 * g;
            float4 bz[2], wv[4][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                bz[e] = *reinterpret_cast<const float4 *>(cst + 1024 + c + 4 * e);
#pragma unroll
                for (int t = 0; t < 4; ++t) wv[t][e] = *reinterpret_cast<const float4 *>(cst + t * 256 + c + 4 * e);
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float v[8] = {bz[0].x, bz[0].y, bz[0].z, bz[0].w, bz[1].x, bz[1].y, bz[1].z, bz[1].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float w8[8] = {wv[t][0].x, wv[t][0].y, wv[t][0].z, wv[t][0].w, wv[t][1].x, wv[t][1].y, wv[t][1].z, wv[t][1].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(sv[rt][t], w8[e], v[e]);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(wj[rt][t], f[rt][t][e], v[e]);
#pragma unroll
                for (int e = 0; e < 8; ++e) h[rt][p][e] = (_Float16)fmaxf(v[e], 0.f);
            }
        };
        if constexpr (G16) {
            const _Float16 *gp[2][3];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int t = 0; t < 3; ++t) gp[rt][t] = reinterpret_cast<const _Float16 *>(a.g) + (size_t)(cloud[rt] * a.m_known + nb[rt][t]) * 256 + 8 * g;
            constexpr int PB = 2;   // pairs per batch: 6 PB 16-byte gathers in flight, the next batch issued before this one is consumed
            half8 f16[2][2][PB][3];
            auto issue = [&](int bq, int slot) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int u = 0; u < PB; ++u)
#pragma unroll
                        for (int t = 0; t < 3; ++t) f16[slot][rt][u][t] = *reinterpret_cast<const half8 *>(gp[rt][t] + 32 * (bq * PB + u));
            };
            issue(0, 0);
            __syncthreads();   // cst visible
            FPX16_STAMP(1);
#pragma unroll
            for (int bq = 0; bq < 8 / PB; ++bq) {
                __builtin_amdgcn_sched_barrier(0);
                if (bq + 1 < 8 / PB) issue(bq + 1, (bq + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    float f[2][3][8];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int t = 0; t < 3; ++t)
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[rt][t][e] = (float)f16[bq & 1][rt][u][t][e];
                    finish(bq * PB + u, f);
                }
            }
        } else {
            const float *gp[2][3];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int t = 0; t < 3; ++t) gp[rt][t] = reinterpret_cast<const float *>(a.g) + (size_t)(cloud[rt] * a.m_known + nb[rt][t]) * 256 + 8 * g;
            __syncthreads();   // cst visible
            FPX16_STAMP(1);
#pragma unroll
            for (int q = 0; q < 8; q += 2) {   // 24 16-byte gathers in flight
                __builtin_amdgcn_sched_barrier(0);
                float4 f4[2][2][3][2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            f4[rt][u][t][0] = *reinterpret_cast<const float4 *>(gp[rt][t] + 32 * (q + u));
                            f4[rt][u][t][1] = *reinterpret_cast<const float4 *>(gp[rt][t] + 32 * (q + u) + 4);
                        }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float f[2][3][8];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            const float4 lo = f4[rt][u][t][0], hi = f4[rt][u][t][1];
                            f[rt][t][0] = lo.x; f[rt][t][1] = lo.y; f[rt][t][2] = lo.z; f[rt][t][3] = lo.w;
                            f[rt][t][4] = hi.x; f[rt][t][5] = hi.y; f[rt][t][6] = hi.z; f[rt][t][7] = hi.w;
                        }
                    finish(q + u, f);
                }
            }
        }
    }

    FPX16_STAMP(2);
    // ---- the layers: one barrier per slab ----------------------------------------------------------------------------------------------
    floatx4 acc[2][16];
    // one base register per buffer, opaque to the compiler: every fragment read is base + a 16-bit immediate (it otherwise materialises
    // an address register per read of the second 64 KB buffer)
    unsigned off0 = lane * 16u, off1 = SLAB + lane * 16u;
    asm volatile("" : "+v"(off0), "+v"(off1));
    const half8 *buf0 = reinterpret_cast<const half8 *>(fpx16_lds + off0), *buf1 = reinterpret_cast<const half8 *>(fpx16_lds + off1);
#pragma unroll
    for (int s = 0; s < NL * NS; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of slab s have landed ...
        __syncthreads();                                    // ... everybody's have, and everybody is done with the other buffer
        if (s == 0) FPX16_STAMP(3);
        if (s >= 1 && s + 1 < NL * NS) fetch(s + 1);
        if (s % NS == 0) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 16; ++ct) acc[rt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
        }
        const half8 *buf = (s & 1) ? buf1 : buf0;
#pragma unroll
        for (int ksl = 0; ksl < KSB; ++ksl) {
            const int ks = (s % NS) * KSB + ksl;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) {
                const half8 w = buf[(ksl * 16 + ct) * 64];
                acc[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, h[0][ks], acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, h[1][ks], acc[1][ct], 0, 0, 0);
            }
            // fragment reads run four ahead of the MFMAs that consume them (all sixteen hoisted = 64 registers = spills)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        if (s == NS - 1) FPX16_STAMP(4);
        if (NL == 2 && s == NS - 1) {   // h2 = relu(acc + b2), straight from the accumulators
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) {
                const float4 bz = *reinterpret_cast<const float4 *>(cst + 1280 + r_ofs(ct, g));
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    h[rt][ct >> 1][4 * (ct & 1) + 0] = (_Float16)fmaxf(acc[rt][ct][0] + bz.x, 0.f);
                    h[rt][ct >> 1][4 * (ct & 1) + 1] = (_Float16)fmaxf(acc[rt][ct][1] + bz.y, 0.f);
                    h[rt][ct >> 1][4 * (ct & 1) + 2] = (_Float16)fmaxf(acc[rt][ct][2] + bz.z, 0.f);
                    h[rt][ct >> 1][4 * (ct & 1) + 3] = (_Float16)fmaxf(acc[rt][ct][3] + bz.w, 0.f);
                }
            }
        }
    }

    FPX16_STAMP(5);
    if constexpr (PREMUL) {
        // ---- g16 = fp16(acc): a column-tile pair = 16 contiguous bytes per lane -----------------------------------------------------------
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const long row = row0 + rt * 16;
            if (row < a.rows) {
                _Float16 *o = a.g16out + (size_t)row * 256 + 8 * g;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const floatx4 lo = acc[rt][2 * p], hi = acc[rt][2 * p + 1];
                    *reinterpret_cast<half8 *>(o + 32 * p) =
                        (half8){(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3], (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2], (_Float16)hi[3]};
                }
            }
        }
    } else if constexpr (OUT16) {
        // ---- out16 = fp16(relu(acc + b3)): a column-tile pair = eight consecutive channels = ONE 16-byte store (half the store volume) ----------
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const long row = row0 + rt * 16;
            if (row < a.rows) {
                _Float16 *o = a.out16 + (size_t)row * 256 + 8 * g;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const float4 b0 = *reinterpret_cast<const float4 *>(cst + 1536 + r_ofs(2 * p, g)), b1 = *reinterpret_cast<const float4 *>(cst + 1536 + r_ofs(2 * p + 1, g));
                    const floatx4 lo = acc[rt][2 * p], hi = acc[rt][2 * p + 1];
                    *reinterpret_cast<half8 *>(o + 32 * p) =
                        (half8){(_Float16)fmaxf(lo[0] + b0.x, 0.f), (_Float16)fmaxf(lo[1] + b0.y, 0.f), (_Float16)fmaxf(lo[2] + b0.z, 0.f), (_Float16)fmaxf(lo[3] + b0.w, 0.f),
                                (_Float16)fmaxf(hi[0] + b1.x, 0.f), (_Float16)fmaxf(hi[1] + b1.y, 0.f), (_Float16)fmaxf(hi[2] + b1.z, 0.f), (_Float16)fmaxf(hi[3] + b1.w, 0.f)};
                }
            }
        }
    } else {
        // ---- out = relu(acc + b3): 16 bytes per lane, a column-tile pair = 32 contiguous bytes ------------------------------------------
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const long row = row0 + rt * 16;
            if (row < a.rows) {
                float *o = a.out + (size_t)row * a.ldo;
#pragma unroll
                for (int ct = 0; ct < 16; ++ct) {
                    const int c = r_ofs(ct, g);
                    const float4 bz = *reinterpret_cast<const float4 *>(cst + 1536 + c);
                    *reinterpret_cast<float4 *>(o + c) = make_float4(fmaxf(acc[rt][ct][0] + bz.x, 0.f), fmaxf(acc[rt][ct][1] + bz.y, 0.f),
                                                                     fmaxf(acc[rt][ct][2] + bz.z, 0.f), fmaxf(acc[rt][ct][3] + bz.w, 0.f));
                }
            }
        }
    }
    if (DBG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FPX16_STAMP(6);
        if (blk * WAVES + wave < 512 && lane < 7) a.dbg[(blk * WAVES + wave) * 8 + lane] = (long long)stamps[wave * 8 + lane];
    }
#undef FPX16_STAMP
}

template <int WAVES, int KSB, int NL, bool PREMUL, bool G16, bool OUT16 = false>
void launch_fpx16(const Fpx16Args &a, hipStream_t st)
{
    const size_t lds = (size_t)2 * KSB * 16 * 1024 + 7 * 1024 + (a.dbg ? 256 : 0);
    auto kern = (a.dbg && !OUT16) ? fpx16_kernel<WAVES, KSB, NL, PREMUL, G16, true, false> : fpx16_kernel<WAVES, KSB, NL, PREMUL, G16, false, OUT16>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(pa_div_up(a.rows, WAVES * 32)), dim3(WAVES * 64), lds, st, a);
}

int g_fpx16_mode = -1;

int fpx16_mode()
{
    static const int env_mode = getenv("PA_FPX16_LDS") ? atoi(getenv("PA_FPX16_LDS")) : 4;   // 0 = off, 4 = four-wave workgroups, 8 = eight-wave
    return g_fpx16_mode >= 0 ? g_fpx16_mode : env_mode;
}

bool fpx16_shape(int nlayers, const void *const *wp16, const int *kpad, const int *nout, int c2, int c1, const float *out, int ldo, long rows, int n_unknown)
{
    if (nlayers != 2 || c2 != 256 || c1 < 1 || c1 > 4) return false;
    for (int l = 0; l < 2; ++l)
        if (kpad[l] != 256 || nout[l] != 256 || !wp16[l]) return false;
    return ldo % 4 == 0 && ((uintptr_t)out & 15) == 0 && n_unknown > 0 && rows % n_unknown == 0;
}

int fpx16_chain(bool g16, const void *const *wp16, const float *const *bias, long rows, const void *g, const int *idx3, const float *w3, const float *skip,
                int n_unknown, int m_known, int c1, const float *wskip, const float *bias0, float *out, int ldo, long long *dbg, hipStream_t st,
                _Float16 *out16 = nullptr)
{
    Fpx16Args a = {};
    a.rows = rows; a.g = g; a.idx3 = idx3; a.w3 = w3; a.skip = skip; a.wskip = wskip; a.bias0 = bias0;
    a.wq[0] = reinterpret_cast<const half8 *>(wp16[0]); a.wq[1] = reinterpret_cast<const half8 *>(wp16[1]);
    a.b[0] = bias[0]; a.b[1] = bias[1]; a.out = out; a.ldo = ldo; a.n_unknown = n_unknown; a.m_known = m_known; a.c1 = c1; a.dbg = dbg;
    static const bool no_xcd = getenv("PA_CHAIN_NO_XCD_REMAP") != nullptr;
    a.xcd_remap = no_xcd ? 0 : 1;
    const bool w8 = fpx16_mode() == 8;
    if (out16) {               // fp16 table in, fp16 map out (four-wave workgroups: the shipped form)
        a.out16 = out16; a.dbg = nullptr;
        launch_fpx16<4, 2, 2, false, true, true>(a, st);
        PA_CHECK_LAUNCH("pa_fp_chain_premul_g16h");
        return PA_OK;
    }
    if (g16) { if (w8) launch_fpx16<8, 4, 2, false, true>(a, st); else launch_fpx16<4, 2, 2, false, true>(a, st); }
    else     { if (w8) launch_fpx16<8, 4, 2, false, false>(a, st); else launch_fpx16<4, 2, 2, false, false>(a, st); }
    PA_CHECK_LAUNCH("pa_fp_chain_premul_f16(lds)");
    return PA_OK;
}

}  // namespace

long long *pa_chain_dbg_ptr();   // mlp_chain.hip

PA_API void pa_fpx16_enable(int mode) { g_fpx16_mode = mode; }

// Shape test + launch for pa_fp_chain_premul_f16 (mlp_chain.hip): returns 1 when this kernel took the call, 0 when the shape is not its
// own (the caller falls through to chain16_kernel), < 0 / hipError on a launch error.
int pa_fpx16_try(int nlayers, const void *const *wp16, const float *const *bias, const int *kpad, const int *nout, long rows, const float *g,
                 const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1, const float *wskip,
                 const float *bias0, float *out, int ldo, long long *dbg, hipStream_t st)
{
    if (fpx16_mode() == 0 || !fpx16_shape(nlayers, wp16, kpad, nout, c2, c1, out, ldo, rows, n_unknown) || ((uintptr_t)g & 15) != 0) return 0;
    const int rc = fpx16_chain(false, wp16, bias, rows, g, idx3, w3, skip, n_unknown, m_known, c1, wskip, bias0, out, ldo, dbg, st);
    return rc == PA_OK ? 1 : (rc < 0 ? rc : -rc);
}

// g16[r][:] = fp16(x[r][:256] . W) for the (256 x 256) first-layer slice W (wp16 = pa_pack_weights_f16(256, 256, W)): the pre-multiply of
// the finest level with an fp16 table for pa_fp_chain_premul_g16's gathers.
PA_API int pa_fp_premul_g16(long rows, const float *x, int ldx, const void *wp16, void *g16, pa_stream_t stream)
{
    PA_REQUIRE(rows > 0 && x && wp16 && g16 && ldx >= 256 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)g16 & 15) == 0,
               "pa_fp_premul_g16: needs 256-wide 16-byte aligned rows");
    Fpx16Args a = {};
    a.rows = rows; a.x = x; a.ldx = ldx; a.wq[0] = reinterpret_cast<const half8 *>(wp16); a.g16out = reinterpret_cast<_Float16 *>(g16);
    a.xcd_remap = 0;
    launch_fpx16<4, 2, 1, true, false>(a, (hipStream_t)stream);
    PA_CHECK_LAUNCH("pa_fp_premul_g16");
    return PA_OK;
}

// pa_fp_chain_premul_f16 with the pre-multiplied features in fp16 (pa_fp_premul_g16).  Only the finest level's shape: c2 = 256, 1 <= c1 <= 4,
// two remaining 256 -> 256 layers; PA_EUNSUPPORTED otherwise (the caller then uses pa_linear_f16 + pa_fp_chain_premul_f16).
PA_API int pa_fp_chain_premul_g16(int nlayers, const void *const *wp16, const float *const *bias, const int *kpad, const int *nout, long rows,
                                  const void *g16, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                                  const float *wskip, const float *bias0, float *out, int ldo, pa_stream_t stream)
{
    PA_REQUIRE(wp16 && bias && kpad && nout && rows > 0 && g16 && idx3 && w3 && skip && wskip && bias0 && out, "pa_fp_chain_premul_g16: null argument");
    if (!fpx16_shape(nlayers, wp16, kpad, nout, c2, c1, out, ldo, rows, n_unknown) || ((uintptr_t)g16 & 15) != 0) {
        pa_set_error("pa_fp_chain_premul_g16: only c2 = 256, 1 <= c1 <= 4 and two 256 -> 256 layers (got nlayers=%d c2=%d c1=%d)", nlayers, c2, c1);
        return PA_EUNSUPPORTED;
    }
    return fpx16_chain(true, wp16, bias, rows, g16, idx3, w3, skip, n_unknown, m_known, c1, wskip, bias0, out, ldo, pa_chain_dbg_ptr(), (hipStream_t)stream);
}

// pa_fp_chain_premul_g16 with the level's output ALSO in fp16: out16 (rows, 256) halfs.  For callers that hand the map to the fp16 NetVLAD kernel
// only (descriptor extraction without feature maps: pa_netvlad_pyramid_f16h): the 134 MB fp32 store and its re-read become 67 MB each.
PA_API int pa_fp_chain_premul_g16h(int nlayers, const void *const *wp16, const float *const *bias, const int *kpad, const int *nout, long rows,
                                   const void *g16, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                                   const float *wskip, const float *bias0, void *out16, pa_stream_t stream)
{
    PA_REQUIRE(wp16 && bias && kpad && nout && rows > 0 && g16 && idx3 && w3 && skip && wskip && bias0 && out16, "pa_fp_chain_premul_g16h: null argument");
    PA_REQUIRE(((uintptr_t)out16 & 15) == 0, "pa_fp_chain_premul_g16h: out16 must be 16-byte aligned");
    if (!fpx16_shape(nlayers, wp16, kpad, nout, c2, c1, reinterpret_cast<const float *>(out16), 256, rows, n_unknown) || ((uintptr_t)g16 & 15) != 0) {
        pa_set_error("pa_fp_chain_premul_g16h: only c2 = 256, 1 <= c1 <= 4 and two 256 -> 256 layers (got nlayers=%d c2=%d c1=%d)", nlayers, c2, c1);
        return PA_EUNSUPPORTED;
    }
    return fpx16_chain(true, wp16, bias, rows, g16, idx3, w3, skip, n_unknown, m_known, c1, wskip, bias0, nullptr, 256, nullptr, (hipStream_t)stream,
                       reinterpret_cast<_Float16 *>(out16));
}
