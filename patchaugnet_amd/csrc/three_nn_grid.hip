// K9, grid form: three nearest neighbours of MANY queries in a mid-sized cloud (the finest feature-propagation level: 4096 unknown
// points against 1024 known ones per cloud), one lane per query over an 8 x 8 x 8 cell grid.
//
// Reference semantics (libs/pointops/src/interpolation/interpolation_cuda_kernel.cu:134-176): the three best by
// strict '<' in scan order => ascending (d2, index); d2 = (ux-x)*(ux-x) + (uy-y)*(uy-y) + (uz-z)*(uz-z) in fp32 without contraction;
// an empty slot is (index 0, +inf).
//
// three_nn_kernel (three_nn.hip) scans all m known points for every query: ~21 instructions per pair, 0.084 ms for 32 x 4096 x 1024.  Here
// the known cloud is counting-sorted into 512 cells in LDS (row-major cell order: the three cells cx-1..cx+1 of a row are one contiguous
// range) and a lane walks the 3 x 3 x 3 neighbourhood of its query's cell -- ~54 candidates instead of 1024 -- keeping the three smallest
// 64-bit (d2 bits, index) keys, so the (d2, index) order holds whatever the visiting order.  Shells are added until the third distance is
// provably smaller than anything unvisited.  Unlike the kNN at the set-abstraction level (knn_lane.hip: 512 query waves, measured slower),
// there are 131 072 queries per batch here = 2048 wavefronts, and a query's chain is ~25x shorter.
//
// Stop rule (exact).  Cells: c_a(p) = clamp((int)f_a(p), 0, 7), f_a(p) = fl(fl(p_a - lo_a) * scale_a), monotone in p_a, 0 <= f_a(p) <= 8
// for cloud points.  After the cells within Chebyshev distance R of the query's (clamped) cell are done, an unvisited point lies beyond
// one of the six faces of that block: beyond the +a face f_a(p) >= c_a + R + 1, beyond the -a face f_a(p) < c_a - R; a face outside the
// grid has nothing behind it.  So |f_a(p) - f_a(q)| >= gap with gap = (c_a + R + 1) - f_a(q) resp. f_a(q) - (c_a - R) -- for ANY query
// position, inside the cloud's box or not.  f carries a relative rounding error of 2 ulp; for |f_a(q)| < 1000 that is < 2.4e-4 absolute,
// hence |p_a - q_a| >= (gap - 1e-3) / scale_a in real numbers and the fp32 distance is at least that squared times (1 - 6 ulp).  The
// kernel stops when d3 < ((gap - 1e-3) * 0.999 / scale_a)^2 for every face (strict <: equal distances are never cut off); a query
// farther out than |f| = 1000 or with non-finite coordinates never stops early.
#include <stdlib.h>

#include "pa_common.h"
#include "pa_cellsort.h"

namespace {

constexpr u64 TG_INF0 = ((u64)0x7F800000u) << 32;
constexpr int TG_AUX_FLOATS = KG_AUX_FLOATS + 8;

template <int PTS, int NT, bool WEIGHTS>
__global__ __launch_bounds__(NT) void three_nn_grid_kernel(int n, int m, int q_per_block, const float *__restrict__ unknown_all, const float *__restrict__ known_all,
                                                            float *__restrict__ out_all, int *__restrict__ idx_all)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *sorted = reinterpret_cast<float4 *>(smem);
    float *box = smem + 4 * (size_t)m;
    int *cnt = reinterpret_cast<int *>(box + 64 * 8);
    float *red = reinterpret_cast<float *>(cnt + KG_CELLS + 1);
    float *grid = red + 16 * 6;
    int *qcnt = reinterpret_cast<int *>(grid + 8);
    unsigned short *qorder = reinterpret_cast<unsigned short *>(qcnt + KG_CELLS + 1);
    const int b = blockIdx.y, tid = threadIdx.x;
    const float *known = known_all + (size_t)b * m * 3;
    const float *unknown = unknown_all + (size_t)b * n * 3;
    int nchunks;
    cell_sort_cloud<PTS, NT, true>(m, known, sorted, box, cnt, red, &nchunks, grid);
    const float lo0 = grid[0], lo1 = grid[1], lo2 = grid[2], sc0 = grid[3], sc1 = grid[4], sc2 = grid[5];

    // this workgroup's queries in cell order (neighbouring lanes walk neighbouring rows)
    const int q_begin = blockIdx.x * q_per_block, q_end = min(q_begin + q_per_block, n);
    {
        for (int c = tid; c <= KG_CELLS; c += NT) qcnt[c] = 0;
        __syncthreads();
        auto qcell = [&](int qi) {
            const float *qp = unknown + (size_t)qi * 3;
            const int cx = min(max((int)((qp[0] - lo0) * sc0), 0), 7), cy = min(max((int)((qp[1] - lo1) * sc1), 0), 7), cz = min(max((int)((qp[2] - lo2) * sc2), 0), 7);
            return (cz * 8 + cy) * 8 + cx;
        };
        for (int qi = q_begin + tid; qi < q_end; qi += NT) atomicAdd(&qcnt[qcell(qi)], 1);
        __syncthreads();
        if (tid < 64) {
            int v[9], sum = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) { const int c = tid * 9 + t; v[t] = c <= KG_CELLS ? qcnt[c] : 0; sum += v[t]; }
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (tid >= o) incl += u; }
            int run = incl - sum;
#pragma unroll
            for (int t = 0; t < 9; ++t) { const int c = tid * 9 + t; if (c <= KG_CELLS) qcnt[c] = run; run += v[t]; }
        }
        __syncthreads();
        for (int qi = q_begin + tid; qi < q_end; qi += NT) qorder[atomicAdd(&qcnt[qcell(qi)], 1)] = (unsigned short)(qi - q_begin);
        __syncthreads();
    }

    for (int qs = tid; qs < q_end - q_begin; qs += NT) {
        const int q = q_begin + qorder[qs];
        const float *qp = unknown + (size_t)q * 3;
        const float ux = qp[0], uy = qp[1], uz = qp[2];
        const float f0 = (ux - lo0) * sc0, f1 = (uy - lo1) * sc1, f2 = (uz - lo2) * sc2;
        const bool tame = fabsf(f0) < 1000.f && fabsf(f1) < 1000.f && fabsf(f2) < 1000.f;   // false for NaN / inf
        const int cx = min(max((int)f0, 0), 7), cy = min(max((int)f1, 0), 7), cz = min(max((int)f2, 0), 7);
        u64 k1 = TG_INF0, k2 = TG_INF0, k3 = TG_INF0;
        auto visit = [&](int pos) {
            const float4 p = sorted[pos];
            const float d = (ux - p.x) * (ux - p.x) + (uy - p.y) * (uy - p.y) + (uz - p.z) * (uz - p.z);   // :155
            const u64 key = pa_make_key(d, (u32)__float_as_int(p.w));
            const bool c1 = key < k1, c2 = key < k2, c3 = key < k3;     // +inf / NaN distances are >= TG_INF0: never admitted
            k3 = c2 ? k2 : (c3 ? key : k3);
            k2 = c1 ? k1 : (c2 ? key : k2);
            k1 = c1 ? key : k1;
        };
        auto scan = [&](int r_from, int r_to) {
            for (int dz = -r_to; dz <= r_to; ++dz) {
                const int z = cz + dz;
                if (z < 0 || z > 7) continue;
                for (int dy = -r_to; dy <= r_to; ++dy) {
                    const int y = cy + dy;
                    if (y < 0 || y > 7) continue;
                    const int rowbase = (z * 8 + y) * 8;
                    const int xl = max(cx - r_to, 0), xh = min(cx + r_to, 7);
                    const bool interior = r_from > 0 && abs(dz) < r_from && abs(dy) < r_from;   // the middle of the row was scanned before
                    {
                        const int c0 = rowbase + xl, c1 = rowbase + (interior ? min(cx - r_from, xh) : xh);
                        if (c1 >= c0) {
                            const int beg = c0 ? cnt[c0 - 1] : 0, end = cnt[c1];
                            for (int pos = beg; pos < end; ++pos) visit(pos);
                        }
                    }
                    if (interior) {
                        const int c0 = rowbase + max(cx + r_from, xl), c1 = rowbase + xh;
                        if (c1 >= c0) {
                            const int beg = c0 ? cnt[c0 - 1] : 0, end = cnt[c1];
                            for (int pos = beg; pos < end; ++pos) visit(pos);
                        }
                    }
                }
            }
        };
        auto outside_bound = [&](int R) {
            float best = INFINITY;
            auto face = [&](float f, int c, float sc) {
                if (!(sc > 0.f)) return;
                const float inv = 0.999f / sc;
                if (c + R + 1 <= 7) { const float g = fmaxf((float)(c + R + 1) - f - 1e-3f, 0.f) * inv; best = fminf(best, g * g); }
                if (c - R - 1 >= 0) { const float g = fmaxf(f - (float)(c - R) - 1e-3f, 0.f) * inv; best = fminf(best, g * g); }
            };
            face(f0, cx, sc0); face(f1, cy, sc1); face(f2, cz, sc2);
            return best;
        };
        int R = 1;
        scan(0, 1);
        while (R < 7) {
            if (tame && k3 < TG_INF0 && __uint_as_float((u32)(k3 >> 32)) < outside_bound(R)) break;
            ++R;
            scan(R, R);
        }
        const float b1 = __uint_as_float((u32)(k1 >> 32)), b2 = __uint_as_float((u32)(k2 >> 32)), b3 = __uint_as_float((u32)(k3 >> 32));
        float *od = out_all + ((size_t)b * n + q) * 3;
        int *oi = idx_all + ((size_t)b * n + q) * 3;
        if (WEIGHTS) {  // patch_aug_net.py:350-353: d = sqrt(d2); r = 1/(d + 1e-8); w = r / ((r0 + r1) + r2)  (as three_nn_kernel<true>)
            const float r1 = 1.0f / (sqrtf(b1) + 1e-8f), r2 = 1.0f / (sqrtf(b2) + 1e-8f), r3 = 1.0f / (sqrtf(b3) + 1e-8f);
            const float norm = (r1 + r2) + r3;
            od[0] = r1 / norm; od[1] = r2 / norm; od[2] = r3 / norm;
        } else {
            od[0] = b1; od[1] = b2; od[2] = b3;
        }
        oi[0] = (int)(u32)k1; oi[1] = (int)(u32)k2; oi[2] = (int)(u32)k3;
    }
}

#ifndef TG_QPB
#define TG_QPB 256
#endif
#ifndef TG_NT
#define TG_NT 256
#endif
template <int PTS, bool WEIGHTS>
void launch_tg(int b, int n, int m, const float *unknown, const float *known, float *out, int *idx, hipStream_t st)
{
    constexpr int NT = TG_NT;
    const int qpb = TG_QPB;   // measured at (32, 4096, 1024): 1024 -> 52 us, 512 -> 40 us, 256 -> 30 us
    const size_t lds = (size_t)m * 16 + (size_t)TG_AUX_FLOATS * 4 + (size_t)(KG_CELLS + 1) * 4 + (size_t)qpb * 2;
    auto kern = three_nn_grid_kernel<PTS, NT, WEIGHTS>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(pa_div_up(n, qpb), b), dim3(NT), lds, st, n, m, qpb, unknown, known, out, idx);
}

}  // namespace

static int g_tg_on = -1;
// A/B and test switch: 0 = always the brute-force scan (three_nn.hip), 1 = the grid kernel where it applies (default; PA_TNN_NO_GRID=1 turns it off)
PA_API void pa_three_nn_grid_enable(int on) { g_tg_on = on ? 1 : 0; }

// 1 when the grid kernel took the call.  weights != 0: inverse-distance weights instead of squared distances (pa_three_nn_weights).
int pa_three_nn_grid_try(int b, int n, int m, const float *unknown, const float *known, float *out, int *idx, int weights, hipStream_t st)
{
    if (g_tg_on < 0) g_tg_on = getenv("PA_TNN_NO_GRID") != nullptr ? 0 : 1;
    // pays when the cloud is big enough for 512 cells to prune and there are enough queries to amortise the per-workgroup sort
    if (!g_tg_on || m < 512 || m > 4096 || n < 1024) return 0;
    if (m <= 1024) { if (weights) launch_tg<1024 / TG_NT, true>(b, n, m, unknown, known, out, idx, st); else launch_tg<1024 / TG_NT, false>(b, n, m, unknown, known, out, idx, st); }
    else { if (weights) launch_tg<4096 / TG_NT, true>(b, n, m, unknown, known, out, idx, st); else launch_tg<4096 / TG_NT, false>(b, n, m, unknown, known, out, idx, st); }
    return 1;
}
