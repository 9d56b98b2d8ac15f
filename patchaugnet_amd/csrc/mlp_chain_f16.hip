// fp16-operand variant of the fused shared-MLP chain kernels (BASELINE.json configs[4]: "fp16 MFMA MLP path").
//
// Same fusion, same prologues (pa_chain.h) and same fp32 outputs as mlp_chain.hip; what changes is the arithmetic of the dense
// layers: activations live in LDS as fp16, the BatchNorm-folded weights are pre-packed as fp16 fragments, and every layer is
// v_mfma_f32_16x16x32_f16 (fp32 accumulation, 8x the K per instruction and 1/2 the issue time of the fp32-input MFMA: 16x its
// rate).  Bias, ReLU, pooling and the value written to memory stay fp32.  Opt-in: descriptors then agree with the fp32 path to
// cosine >= 0.999 instead of max-abs 1e-4 (arithmetic contract).
//
// Fragments (lane l, g = l / 16, e = 0..7): activation A-map element = act[row 16rt + l%16][k = 32ks + 8g + e] -- one 16-byte LDS
// read; weight element = W[k = 32ks + 8g + e][col 16ct + l%16], packed by pa_pack_weights_f16 as
//     wp16[(((ct * ksteps + ks) * 64 + l) * 8 + e]
// -- one 16-byte global load.  As in the fp32 kernel's pooled/split variants the WEIGHT fragment is the MFMA's A operand, so a
// lane ends up with four consecutive channels of one point: hidden activations return to LDS as one 8-byte store (4 halfs).
#include <stdlib.h>
#include <string.h>

#include "pa_chain.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

template <int RT, int NC>
__device__ __forceinline__ void gemm16(const _Float16 *__restrict__ act, int stride, const PaLayer &L, int c0, int lane, floatx4 (&acc)[RT][NC])
{
    const int ksteps = L.k32 >> 5, last = ksteps - 1;
    const _Float16 *ap = act + (lane & 15) * stride + (lane >> 4) * 8;
    const half8 *wq = reinterpret_cast<const half8 *>(L.wp16) + (size_t)c0 * ksteps * 64 + lane;
    const size_t tstep = (size_t)ksteps * 64;   // half8s between consecutive column tiles
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) acc[rt][ct] = (floatx4){0.f, 0.f, 0.f, 0.f};
    // two operand sets, each refilled for k-step ks+2 right after its last use (same scheme as the fp32 kernel)
    half8 b0[NC], b1[NC], a0[RT], a1[RT];
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) { b0[ct] = wq[ct * tstep]; b1[ct] = wq[ct * tstep + (size_t)min(1, last) * 64]; }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        a0[rt] = *reinterpret_cast<const half8 *>(ap + rt * 16 * stride);
        a1[rt] = *reinterpret_cast<const half8 *>(ap + rt * 16 * stride + min(1, last) * 32);
    }
#define PA16_BLOCK(A, B, NX)                                                                                                  \
    {                                                                                                                         \
        const int nx_ = min((NX), last);                                                                                      \
        half8 an_[RT];                                                                                                        \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) an_[rt] = *reinterpret_cast<const half8 *>(ap + rt * 16 * stride + nx_ * 32); \
        _Pragma("unroll") for (int ct = 0; ct < NC; ++ct) {                                                                   \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                                                 \
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(B[ct], A[rt], acc[rt][ct], 0, 0, 0);                     \
            B[ct] = wq[ct * tstep + (size_t)nx_ * 64];                                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);                                                               \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                                \
        }                                                                                                                     \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) A[rt] = an_[rt];                                                    \
    }
    int ks = 0;
    for (; ks + 2 <= ksteps; ks += 2) {
        PA16_BLOCK(a0, b0, ks + 2)
        PA16_BLOCK(a1, b1, ks + 3)
    }
#undef PA16_BLOCK
    if (ks < ksteps) {
#pragma unroll
        for (int ct = 0; ct < NC; ++ct)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0[ct], a0[rt], acc[rt][ct], 0, 0, 0);
    }
}

// hidden layer: bias + ReLU in fp32, rounded to fp16, written back in place (four consecutive channels = one 8-byte store)
template <int RT, int NC, bool ADD>
__device__ __forceinline__ void store_hidden16(_Float16 *act, int stride, const PaLayer &L, int c0, int lane, floatx4 (&acc)[RT][NC])
{
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) {
        const int col = (c0 + ct) * 16 + (lane >> 4) * 4;
        const float4 bias = *reinterpret_cast<const float4 *>(L.bias + col);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            _Float16 *d = act + (rt * 16 + (lane & 15)) * stride + col;
            float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
            if (ADD) {   // folded first layer: the interpolated term sits where the result goes
                typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                const half4 h = *reinterpret_cast<const half4 *>(d);
                e0 = (float)h[0]; e1 = (float)h[1]; e2 = (float)h[2]; e3 = (float)h[3];
            }
            pa_store4(d, fmaxf(acc[rt][ct][0] + bias.x + e0, 0.f), fmaxf(acc[rt][ct][1] + bias.y + e1, 0.f),
                      fmaxf(acc[rt][ct][2] + bias.z + e2, 0.f), fmaxf(acc[rt][ct][3] + bias.w + e3, 0.f));
        }
    }
}

template <int RT, int NC, int MODE, bool POOLED, int WPT>
__device__ __forceinline__ void run_layer16(_Float16 *act, const PaChain &a, const PaLayer &L, float *out, const float *residual, int l, long tile,
                                            int lane, int c_begin, int c_end)
{
    const bool last = (l == a.nlayers - 1);
    for (int c0 = c_begin; c0 < c_end; c0 += NC) {
        floatx4 acc[RT][NC];
        const bool fold = MODE == MODE_FP && l == 0 && a.fold0;
        gemm16<RT, NC>(fold ? act + a.c2 : act, a.lds_stride, L, c0, lane, acc);
        if (!last) {
            tile_sync<WPT>();  // every read of this layer's input has landed before its rows are overwritten (single chunk per wave: host-checked)
            if (fold) store_hidden16<RT, NC, true>(act, a.lds_stride, L, c0, lane, acc);
            else store_hidden16<RT, NC, false>(act, a.lds_stride, L, c0, lane, acc);
        } else if (POOLED) {
            if (a.vec_out) store_pooled<RT, NC, true>(out, a.ldo, tile * 4, a.rows, L, c0, lane, acc);
            else store_pooled<RT, NC, false>(out, a.ldo, tile * 4, a.rows, L, c0, lane, acc);
        } else {
            const long total_rows = (MODE == MODE_SA) ? a.rows * a.ns : a.rows;
            if (a.vec_out) store_rows<RT, NC, true>(out, a.ldo, tile * (RT * 16), total_rows, L, c0, lane, acc, a.relu_last, residual, a.ldr);
            else store_rows<RT, NC, false>(out, a.ldo, tile * (RT * 16), total_rows, L, c0, lane, acc, a.relu_last, residual, a.ldr);
        }
    }
    if (!last) tile_sync<WPT>();
}

template <int RT, int NCMAX, int MODE, bool POOLED, int WPT>
__global__ __launch_bounds__(256, (POOLED && RT <= 5) ? 2 : 1) void chain16_kernel(PaChain a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    constexpr int R = RT * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nblk = gridDim.x;
    const long blk = (a.xcd_remap && (nblk & 7) == 0) ? (long)(blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    const long tile = WPT == 1 ? blk * (blockDim.x >> 6) + wave : blk;
    const long total_rows = (MODE == MODE_SA) ? a.rows * a.ns : a.rows;
    const long ntiles = POOLED ? (a.rows + 3) / 4 : (total_rows + R - 1) / R;
    if (tile >= ntiles) return;
    _Float16 *act = reinterpret_cast<_Float16 *>(smem16 + (WPT == 1 ? (size_t)wave * a.wave_floats * 4 : (size_t)0));
    const int tid = WPT == 1 ? lane : (int)threadIdx.x;
    const int stride = a.lds_stride;
    float *scratch = reinterpret_cast<float *>(act + (size_t)R * stride);
    chain_prologue<_Float16, R, MODE, POOLED, WPT>(act, scratch, a, tile, tid, lane, stride, (MODE == MODE_FP && a.fold0) ? a.c2 + a.L[0].k32 : a.L[0].k32);
    tile_sync<WPT>();
    for (int l = 0; l < a.nlayers; ++l) {
        PaLayer L = a.L[l];
        float *out = a.out;
        const float *residual = a.residual;
        if (MODE == MODE_PLAIN && WPT == 4) pa_col_slice(a, L, out, residual);
        const int nct = L.n >> 4;
        const int per = WPT == 1 ? nct : nct / WPT;
        const int cb = WPT == 1 ? 0 : wave * per, ce = cb + per;
        if (NCMAX >= 16 && per % 16 == 0) run_layer16<RT, (NCMAX >= 16 ? 16 : NCMAX), MODE, POOLED, WPT>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else if (NCMAX >= 8 && per % 8 == 0) run_layer16<RT, (NCMAX >= 8 ? 8 : NCMAX), MODE, POOLED, WPT>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else if (NCMAX >= 4 && per % 4 == 0) run_layer16<RT, (NCMAX >= 4 ? 4 : NCMAX), MODE, POOLED, WPT>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else if (per % 2 == 0) run_layer16<RT, 2, MODE, POOLED, WPT>(act, a, L, out, residual, l, tile, lane, cb, ce);
        else run_layer16<RT, 1, MODE, POOLED, WPT>(act, a, L, out, residual, l, tile, lane, cb, ce);
    }
}

template <int RT, int NCMAX, int MODE, bool POOLED, int WPT>
void launch16(const PaChain &a, int waves_per_wg, long ntiles, hipStream_t st)
{
    const size_t lds = (size_t)(WPT == 1 ? waves_per_wg : 1) * a.wave_floats * 4;
    auto kern = chain16_kernel<RT, NCMAX, MODE, POOLED, WPT>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (WPT == 1) hipLaunchKernelGGL(kern, dim3(pa_div_up(ntiles, waves_per_wg)), dim3(64 * waves_per_wg), lds, st, a);
    else hipLaunchKernelGGL(kern, dim3(ntiles, a.col_slices > 1 ? a.col_slices : 1), dim3(256), lds, st, a);
}

template <int MODE>
void launch16_rows(const PaChain &a, int rt, bool split, int wpw, long ntiles, hipStream_t st)
{
    if (!split) launch16<2, 16, MODE, false, 1>(a, wpw, ntiles, st);
    else if (rt == 2) launch16<2, 8, MODE, false, 4>(a, 4, ntiles, st);
    else launch16<1, 8, MODE, false, 4>(a, 4, ntiles, st);
}

// wp16[(((ct * ksteps + ks) * 64 + l) * 8 + e] = (half) wt[(32 ks + 8 (l/16) + e) * n + 16 ct + l%16], zero where k >= kpad
__global__ void pack_weights_f16_kernel(int kpad, int n, const float *__restrict__ wt, _Float16 *__restrict__ wp)
{
    const int ksteps = (kpad + 31) >> 5;
    const long total = (long)(n >> 4) * ksteps * 512;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int e = (int)(t & 7), l = (int)((t >> 3) & 63);
    const long rest = t >> 9;
    const int ks = (int)(rest % ksteps), ct = (int)(rest / ksteps);
    const int k = 32 * ks + 8 * (l >> 4) + e;
    wp[t] = k < kpad ? (_Float16)wt[(size_t)k * n + 16 * ct + (l & 15)] : (_Float16)0.f;
}

}  // namespace

// Called by chain_dispatch (mlp_chain.hip) once the tiling (pooled / split / row tiles) is chosen; fills the fp16 LDS geometry.
int pa_chain16_launch(PaChain &a, int mode, bool is_pooled, bool split, int RTv, int scratch_floats, hipStream_t st)
{
    int maxk = 0;
    for (int l = 0; l < a.nlayers; ++l) {
        PA_REQUIRE(a.L[l].wp16 != nullptr, "fp16 chain: layer %d has no fp16 weights", l);
        a.L[l].k32 = (a.L[l].kpad + 31) & ~31;
        if (a.L[l].k32 > maxk) maxk = a.L[l].k32;
        PA_REQUIRE(l + 1 == a.nlayers || a.L[l].n % 32 == 0, "fp16 chain: hidden width %d must be a multiple of 32", a.L[l].n);
    }
    PA_REQUIRE((mode != MODE_FPX && !a.fold0) || a.c2 % 32 == 0, "fp16 chain: pre-multiplied width %d must be a multiple of 32", a.c2);
    if (a.fold0 && a.c2 + a.L[0].k32 > maxk) maxk = a.c2 + a.L[0].k32;
    if (a.col_slices > 1) a.wp16_slice = (long)a.slice_n * a.L[0].k32;
    const int R = RTv * 16;
    a.lds_stride = maxk + 8;                                   // halfs; row pitch is a multiple of 16 bytes, 16 rows hit all 64 banks
    const size_t tile_bytes = (size_t)R * a.lds_stride * 2;
    a.wave_floats = (int)((tile_bytes + (size_t)scratch_floats * 4 + 15) / 16 * 4);   // per-tile LDS in 4-byte units, 16-byte granules
    a.ep_stride = 0;
    const size_t per_wave = (size_t)a.wave_floats * 4;
    PA_REQUIRE(per_wave <= 156 * 1024, "fp16 chain: one tile needs %zu B of LDS", per_wave);
    int wpw = 4;
    while (wpw > 1 && wpw * per_wave > 156 * 1024) wpw >>= 1;
    const long total_rows = (mode == MODE_SA) ? a.rows * a.ns : a.rows;
    const long ntiles = is_pooled ? (a.rows + 3) / 4 : (total_rows + R - 1) / R;
    if (is_pooled) {
        if (split) launch16<5, 4, MODE_SA, true, 4>(a, 4, ntiles, st);
        else switch (RTv) {
            case 4: launch16<4, 4, MODE_SA, true, 1>(a, wpw, ntiles, st); break;
            case 5: launch16<5, 4, MODE_SA, true, 1>(a, wpw, ntiles, st); break;
            case 8: launch16<8, 4, MODE_SA, true, 1>(a, wpw, ntiles, st); break;
            default: pa_set_error("fp16 chain: pooled tiling is built for nsample in (13..16], (17..20], (29..32]"); return PA_EUNSUPPORTED;
        }
    } else if (mode == MODE_PLAIN) launch16_rows<MODE_PLAIN>(a, RTv, split, wpw, ntiles, st);
    else if (mode == MODE_SA) launch16_rows<MODE_SA>(a, RTv, split, wpw, ntiles, st);
    else if (mode == MODE_FP) launch16_rows<MODE_FP>(a, RTv, split, wpw, ntiles, st);
    else launch16_rows<MODE_FPX>(a, RTv, split, wpw, ntiles, st);
    PA_CHECK_LAUNCH("pa_mlp_chain(fp16)");
    return PA_OK;
}

PA_API long pa_pack_weights_f16_halfs(int kpad, int n) { return (long)(n / 16) * ((kpad + 31) / 32) * 512; }

// fp16 fragment packing of a K-major (kpad x n) fp32 weight matrix for the fp16 chain kernels; n % 16 == 0.
PA_API int pa_pack_weights_f16(int kpad, int n, const float *wt, void *wp16, pa_stream_t stream)
{
    PA_REQUIRE(kpad > 0 && n > 0 && n % 16 == 0 && wt && wp16, "pa_pack_weights_f16: need n %% 16 == 0 (kpad=%d n=%d)", kpad, n);
    const long total = pa_pack_weights_f16_halfs(kpad, n);
    hipLaunchKernelGGL(pack_weights_f16_kernel, dim3(pa_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, kpad, n, wt,
                       reinterpret_cast<_Float16 *>(wp16));
    PA_CHECK_LAUNCH("pa_pack_weights_f16");
    return PA_OK;
}
