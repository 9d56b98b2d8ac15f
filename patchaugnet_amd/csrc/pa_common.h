// Shared device/host helpers for libpatchaugnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <cmath>

#include "pa_internal.h"

#define PA_API extern "C" __attribute__((visibility("default")))

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- error reporting ---------------------------------------------------------------------------
void pa_set_error(const char *fmt, ...);

#define PA_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            pa_set_error(__VA_ARGS__);   \
            return PA_EINVAL;            \
        }                                \
    } while (0)

// Called right after a kernel launch: reports launch-configuration errors without synchronising.
#define PA_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            pa_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return (int)e__;                                                         \
        }                                                                            \
    } while (0)

static inline int pa_div_up(long a, long b) { return (int)((a + b - 1) / b); }

/* dst[0 .. words) = value (32-bit words), or `rows` runs of `words` words `pitch_words` apart: a KERNEL launch (csrc/abi.hip).  Used instead of
 * hipMemset*Async everywhere a launch sequence may be captured into a hipGraph: memset NODES of a captured graph were not reliably ordered in
 * front of the kernel nodes that accumulate into the buffer (ROCm 7.2, MI355X: the patch-Chamfer gradient of train.GraphedTrainer came out as
 * |g| = 1e20 .. inf in a fraction of the replays).  Kernel nodes are. */
int pa_fill32(void *dst, unsigned value, size_t words, hipStream_t st);
int pa_fill32_2d(void *dst, size_t pitch_words, unsigned value, size_t words, size_t rows, hipStream_t st);

// libs/pointops/src/cuda_utils.h:15-18 -- the reference's block-size rule; it fixes the FPS
// tie-break order, so it is computed with the same double-precision log ratio.
static inline int pa_opt_n_threads(int work_size)
{
    const int pow_2 = (int)(std::log(static_cast<double>(work_size)) / std::log(2.0));
    int t = 1 << pow_2;
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return t;
}

// ---- wavefront helpers (wave64) ------------------------------------------------------------------
#define PA_DPP_ROW_SHR(n) (0x110 + (n))
#define PA_DPP_WAVE_SHR1 0x138
#define PA_DPP_ROW_BCAST15 0x142
#define PA_DPP_ROW_BCAST31 0x143

__device__ __forceinline__ u64 pa_make_key(float d, u32 lo) { return ((u64)__float_as_uint(d) << 32) | lo; }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 pa_dpp_u64(u64 v)
{
    const u32 lo = __builtin_amdgcn_update_dpp(0u, (u32)v, CTRL, ROW_MASK, 0xf, true);
    const u32 hi = __builtin_amdgcn_update_dpp(0u, (u32)(v >> 32), CTRL, ROW_MASK, 0xf, true);
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 pa_max_u64(u64 a, u64 b) { return a > b ? a : b; }

// A pair of floats pinned to a register pair, opaque to the compiler: packed fp32 arithmetic on it is the PLAIN instruction form (v_pk_add_f32 / v_pk_mul_f32 without
// op_sel / op_sel_hi / neg_lo / neg_hi).  The library issues no packed fp32 instruction with operand modifiers: on gfx950 (MI355X, ROCm 7.2) those return wrong values
// -- the low half of the result, at rates up to 1e-3 of the lane-operations -- while another wave of the SIMD issues v_mfma_f32_16x16x32_{f16,bf16}
// instructions, which is what every fp16 chain / attention / NetVLAD kernel of the "f16" mode does; the plain form and scalar fp32 are not affected (measured:
// tools/probes/pk_f32_fault_repro.hip (standalone), pk_f32_victim.hip + corun_stress.hip, profiles/r05_pk_f32_modifier_fault.txt; it is the half-select modifiers op_sel / op_sel_hi that fail, the
// negation modifiers alone measured clean -- the rule bans both, the test cannot tell a safe modifier from an unsafe one on the next toolchain).  Two rules keep the forms out: the library is built with
// -fno-slp-vectorize (the compiler's own pairing of scalar fp32 code is where most of them came from), and hand-written pair arithmetic negates / broadcasts into a
// pa_pk_plain() pair first.  tests/test_abi.py disassembles the library and fails on any v_pk_*_f32 that carries a modifier.
typedef float pa_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pa_f2 pa_pk_plain(pa_f2 v)
{
    asm("" : "+v"(v));
    return v;
}

// max of a 32-bit value over the 64 lanes, returned broadcast.  One instruction per step: the DPP move folds into the v_max_u32.
__device__ __forceinline__ u32 pa_wave_max_u32(u32 v)
{
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, PA_DPP_ROW_SHR(1), 0xf, 0xf, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, PA_DPP_ROW_SHR(2), 0xf, 0xf, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, PA_DPP_ROW_SHR(4), 0xf, 0xf, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, PA_DPP_ROW_SHR(8), 0xf, 0xf, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, PA_DPP_ROW_BCAST15, 0xa, 0xf, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, PA_DPP_ROW_BCAST31, 0xc, 0xf, true));
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

// max of a (hi, lo) key over the wavefront as two 32-bit reductions (hi first, then lo among the lanes that hold the largest hi):
// 12 dependent vector instructions instead of ~30 for the 64-bit DPP form below -- for kernels whose round IS this chain (FPS).
__device__ __forceinline__ u64 pa_wave_max_key2(u64 v)
{
    const u32 hi = (u32)(v >> 32), lo = (u32)v;
    const u32 H = pa_wave_max_u32(hi);
    const u32 L = pa_wave_max_u32(hi == H ? lo : 0u);
    return ((u64)H << 32) | L;
}

// max over the 64 lanes of a wavefront; result valid in lane 63 and returned broadcast (SGPR pair).
__device__ __forceinline__ u64 pa_wave_max_u64(u64 v)
{
    v = pa_max_u64(v, pa_dpp_u64<PA_DPP_ROW_SHR(1), 0xf>(v));
    v = pa_max_u64(v, pa_dpp_u64<PA_DPP_ROW_SHR(2), 0xf>(v));
    v = pa_max_u64(v, pa_dpp_u64<PA_DPP_ROW_SHR(4), 0xf>(v));
    v = pa_max_u64(v, pa_dpp_u64<PA_DPP_ROW_SHR(8), 0xf>(v));
    v = pa_max_u64(v, pa_dpp_u64<PA_DPP_ROW_BCAST15, 0xa>(v));
    v = pa_max_u64(v, pa_dpp_u64<PA_DPP_ROW_BCAST31, 0xc>(v));
    const u32 lo = __builtin_amdgcn_readlane((u32)v, 63);
    const u32 hi = __builtin_amdgcn_readlane((u32)(v >> 32), 63);
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 pa_readlane_u64(u64 v, int lane)
{
    const u32 lo = __builtin_amdgcn_readlane((u32)v, lane);
    const u32 hi = __builtin_amdgcn_readlane((u32)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}

// squared distance in the arithmetic contract of SURVEY.md section 8: fp32, (dx*dx + dy*dy) + dz*dz,
// no FMA contraction (the library is built with -ffp-contract=off).
__device__ __forceinline__ float pa_sqdist(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return dx * dx + dy * dy + dz * dz;
}
