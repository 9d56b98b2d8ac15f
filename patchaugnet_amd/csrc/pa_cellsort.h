// Spatial sort of one cloud inside a workgroup (shared by the pruned kNN and 3-NN kernels): counting sort into 512 Morton-ordered
// cells in LDS, so that each run of 64 consecutive sorted points (a "chunk") is spatially compact, plus each chunk's bounding box.
#pragma once
#include "pa_common.h"

namespace {

constexpr int KG_CELLS = 512;

#ifdef KG_STAMPS     // phase stamps of the sort (variant builds only: tools/build_variant.sh NAME -DKG_STAMPS knn_quad.hip; read with pa_kg_stamps_read)
__device__ long long kg_stamps[16];
#define KG_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) kg_stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define KG_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ u32 kg_spread3(u32 v)  // 3 bits -> every third bit
{
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4);
}

__device__ __forceinline__ u64 kg_shfl_xor_u64(u64 v, int m)
{
    const u32 lo = (u32)__shfl_xor((int)(u32)v, m), hi = (u32)__shfl_xor((int)(u32)(v >> 32), m);
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 kg_wave_min_u64(u64 v) { return ~pa_wave_max_u64(~v); }

// min / max of a float over the wavefront, returned broadcast.  DPP row shifts and row broadcasts (one VALU instruction per step; the xor-shuffle
// butterfly this replaces is an LDS crossbar round trip per step -- six reductions of six steps were 3.8 k of the sort's 23 k cycles).  Lanes a
// step has no source for keep `ident` (bound_ctrl off), so the identity never has to be a bit pattern of zeros.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float kg_dpp_f(float v, float ident)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float kg_wave_min_f(float v)
{
    v = fminf(v, kg_dpp_f<PA_DPP_ROW_SHR(1), 0xf>(v, INFINITY));
    v = fminf(v, kg_dpp_f<PA_DPP_ROW_SHR(2), 0xf>(v, INFINITY));
    v = fminf(v, kg_dpp_f<PA_DPP_ROW_SHR(4), 0xf>(v, INFINITY));
    v = fminf(v, kg_dpp_f<PA_DPP_ROW_SHR(8), 0xf>(v, INFINITY));
    v = fminf(v, kg_dpp_f<PA_DPP_ROW_BCAST15, 0xa>(v, INFINITY));
    v = fminf(v, kg_dpp_f<PA_DPP_ROW_BCAST31, 0xc>(v, INFINITY));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float kg_wave_max_f(float v)
{
    v = fmaxf(v, kg_dpp_f<PA_DPP_ROW_SHR(1), 0xf>(v, -INFINITY));
    v = fmaxf(v, kg_dpp_f<PA_DPP_ROW_SHR(2), 0xf>(v, -INFINITY));
    v = fmaxf(v, kg_dpp_f<PA_DPP_ROW_SHR(4), 0xf>(v, -INFINITY));
    v = fmaxf(v, kg_dpp_f<PA_DPP_ROW_SHR(8), 0xf>(v, -INFINITY));
    v = fmaxf(v, kg_dpp_f<PA_DPP_ROW_BCAST15, 0xa>(v, -INFINITY));
    v = fmaxf(v, kg_dpp_f<PA_DPP_ROW_BCAST31, 0xc>(v, -INFINITY));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// LDS floats needed behind the sorted points (16 n bytes): boxes [64][8], cell counters [KG_CELLS + 1], cross-wave scratch [16][6]
constexpr int KG_AUX_FLOATS = 64 * 8 + (KG_CELLS + 1) + 16 * 6;

// sorted[0 .. n_valid) = finite points in Morton-cell order as (x, y, z, original index bits); [n_valid, n) = the non-finite ones.
// box[c*8 + 0..5] = lo.xyz, hi.xyz of chunk c.  Returns n_valid; *nchunks_out = ceil(n_valid / 64) (<= 64: n <= 4096).
// All NT threads of the workgroup must call; ends with a barrier.
// ROWMAJOR: cells are numbered (cz * 8 + cy) * 8 + cx instead of in Morton order, so the cells cx-1 .. cx+1 of one (cy, cz) row are ONE
// contiguous range of the sorted array (the per-lane query kernel, knn_lane.hip); grid_out (6 floats, LDS): lo.xyz and scale.xyz of the
// cell function cell_a(p) = clamp((int)((p_a - lo_a) * scale_a), 0, 7).  After the call cnt[c] = END of cell c in `sorted` (c < 512).
template <int PTS, int NT, bool ROWMAJOR = false>
__device__ __forceinline__ int cell_sort_cloud(int n, const float *__restrict__ xyz, float4 *sorted, float *box, int *cnt, float *red,
                                               int *nchunks_out, float *grid_out = nullptr)
{
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- 1. cloud bounding box over the finite points
    KG_STAMP(0);
    float px[PTS], py[PTS], pz[PTS];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    // (16-byte loads of the cloud into LDS and a pick-up from there -- a third of the cache-line requests -- measured no different: the phase is the
    // latency of the first touch at launch, not the address path)
#pragma unroll
    for (int u = 0; u < PTS; ++u) {
        const int i = tid + u * NT;
        if (i < n) { px[u] = xyz[i * 3 + 0]; py[u] = xyz[i * 3 + 1]; pz[u] = xyz[i * 3 + 2]; }
    }
#pragma unroll
    for (int u = 0; u < PTS; ++u) {
        const int i = tid + u * NT;
        if (i < n) {
            if (isfinite(px[u]) && isfinite(py[u]) && isfinite(pz[u])) {
                lo[0] = fminf(lo[0], px[u]); hi[0] = fmaxf(hi[0], px[u]);
                lo[1] = fminf(lo[1], py[u]); hi[1] = fmaxf(hi[1], py[u]);
                lo[2] = fminf(lo[2], pz[u]); hi[2] = fmaxf(hi[2], pz[u]);
            }
        }
    }
    KG_STAMP(1);
    for (int c = tid; c <= KG_CELLS; c += NT) cnt[c] = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        lo[t] = kg_wave_min_f(lo[t]);
        hi[t] = kg_wave_max_f(hi[t]);
        if (lane == 0) { red[wave * 6 + t] = lo[t]; red[wave * 6 + 3 + t] = hi[t]; }
    }
    __syncthreads();
    float scale[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        lo[t] = red[t];
        hi[t] = red[3 + t];
#pragma unroll
        for (int w = 1; w < NW; ++w) { lo[t] = fminf(lo[t], red[w * 6 + t]); hi[t] = fmaxf(hi[t], red[w * 6 + 3 + t]); }
        const float ext = hi[t] - lo[t];
        scale[t] = (ext > 0.f && isfinite(ext)) ? 8.0f / ext : 0.f;
    }
    if (grid_out && tid == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t) { grid_out[t] = lo[t]; grid_out[3 + t] = scale[t]; }
    }
    // ---- 2. Morton cell of every point, histogram
    KG_STAMP(2);
    int cell[PTS];
#pragma unroll
    for (int u = 0; u < PTS; ++u) {
        const int i = tid + u * NT;
        cell[u] = -1;
        if (i < n) {
            cell[u] = KG_CELLS;                                                  // non-finite points: last bin, never a candidate
            if (isfinite(px[u]) && isfinite(py[u]) && isfinite(pz[u])) {
                const u32 cx = (u32)min(max((int)((px[u] - lo[0]) * scale[0]), 0), 7);
                const u32 cy = (u32)min(max((int)((py[u] - lo[1]) * scale[1]), 0), 7);
                const u32 cz = (u32)min(max((int)((pz[u] - lo[2]) * scale[2]), 0), 7);
                cell[u] = ROWMAJOR ? (int)((cz * 8u + cy) * 8u + cx) : (int)(kg_spread3(cx) | (kg_spread3(cy) << 1) | (kg_spread3(cz) << 2));
            }
            atomicAdd(&cnt[cell[u]], 1);
        }
    }
    __syncthreads();
    // ---- 3. exclusive scan of the 513 bins (wave 0, 9 bins per lane), scatter
    KG_STAMP(3);
    if (wave == 0) {
        int v[9], sum = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int c = lane * 9 + t;
            v[t] = c <= KG_CELLS ? cnt[c] : 0;
            sum += v[t];
        }
        int incl = sum;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int o = __shfl_up(incl, s);
            if (lane >= s) incl += o;
        }
        int run = incl - sum;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int c = lane * 9 + t;
            if (c <= KG_CELLS) cnt[c] = run;
            run += v[t];
        }
    }
    __syncthreads();
    const int n_valid = cnt[KG_CELLS];                                           // start of the non-finite bin == number of finite points
    __syncthreads();
    KG_STAMP(4);
#pragma unroll
    for (int u = 0; u < PTS; ++u) {
        const int i = tid + u * NT;
        if (i < n) {   // non-finite points (bin KG_CELLS) land behind the n_valid finite ones: never candidates, but still listed
            const int pos = atomicAdd(&cnt[cell[u]], 1);
            sorted[pos] = make_float4(px[u], py[u], pz[u], __int_as_float(i));
        }
    }
    __syncthreads();
    // ---- 4. chunk bounding boxes: four threads per chunk, 16 points each, combined inside the quad
    KG_STAMP(5);
    const int nchunks = (n_valid + 63) >> 6;
    *nchunks_out = nchunks;
    if (!ROWMAJOR) {            // only the chunk-pruned kernel (knn.hip) reads the boxes; the cell-grid kernels walk cells (5.5 k of the sort's 23 k cycles)
        const int c = tid >> 2, part = tid & 3;
        float bl[3] = {INFINITY, INFINITY, INFINITY}, bh[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (c < nchunks) {
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int i = c * 64 + part * 16 + j;
                if (i < n_valid) {
                    const float4 p = sorted[i];
                    bl[0] = fminf(bl[0], p.x); bh[0] = fmaxf(bh[0], p.x);
                    bl[1] = fminf(bl[1], p.y); bh[1] = fmaxf(bh[1], p.y);
                    bl[2] = fminf(bl[2], p.z); bh[2] = fmaxf(bh[2], p.z);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            bl[t] = fminf(bl[t], __shfl_xor(bl[t], 1)); bh[t] = fmaxf(bh[t], __shfl_xor(bh[t], 1));
            bl[t] = fminf(bl[t], __shfl_xor(bl[t], 2)); bh[t] = fmaxf(bh[t], __shfl_xor(bh[t], 2));
        }
        if (c < nchunks && part == 0) {
            float *bx = box + c * 8;
            bx[0] = bl[0]; bx[1] = bl[1]; bx[2] = bl[2]; bx[3] = bh[0]; bx[4] = bh[1]; bx[5] = bh[2];
        }
        __syncthreads();
    }
    KG_STAMP(6);

    return n_valid;
}

}  // namespace
