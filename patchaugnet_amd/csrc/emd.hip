// Earth mover's distance by the auction algorithm -- forward (assignment + squared distances) and backward.
//
// Replaces libs/emd_module/emd_cuda.cu:228-282 (emd_cuda_forward: `iters` rounds of the seven kernels clear /
// calc_unass_cnt / calc_unass_cnt_sum / calc_unass_idx / Bid / GetMax / Assign, then CalcDist) and :284-317
// (NmDistanceGradKernel).  The reference pays 7 launches per round (7 168 launches at iters = 1024) and re-reads xyz2
// and the prices from global memory in every Bid block.
//
// MI355X plan: ONE persistent workgroup (16 wavefronts) per cloud pair runs every round on chip.
//   * xyz2 (SoA) and the prices live in LDS for the whole launch (16 n bytes, n <= 8192), the list of unassigned
//     points is rebuilt each round with ballot/popcount prefix sums into LDS (ascending order);
//   * Bid: one wavefront per unassigned point, lanes stride the objects (conflict-free LDS reads), the
//     (best, best index, second best) triple is merged across the wavefront with xor-shuffles;
//   * GetMax / Assign are per-point passes separated by workgroup barriers; nothing leaves the CU between rounds,
//     and a cloud whose points are all assigned stops early (later rounds are no-ops in the reference too).
//
// Determinism.  The reference's result depends on thread timing (GetMax/Assign "last writer wins", :181-215) and on a
// data-dependent split of the object range in Bid (:108-118).  This kernel resolves those free choices exactly like
// the CPU oracle (oracle/pointops_oracle.c, oracle_emd_forward): lowest object index among equal values, highest
// bidder index among increments that match the maximum within 1e-6, evictions seen only by the next round, and on
// the last round bidders are applied in ascending order.  Arithmetic follows the source: value =
// (float)(3.0 - (double)sqrtf(d2) - (double)price) (the literal 3.0 is a double, :146), increment =
// (best - better) + eps in fp32.
#include "pa_common.h"

namespace {

constexpr int EMD_THREADS = 1024;
constexpr int EMD_WAVES = EMD_THREADS / 64;

struct Bid3 {
    float best, better;
    int best_i;
};

__device__ __forceinline__ Bid3 bid_merge(Bid3 a, float ob, float obt, int oi)
{
    // order statistics of the union: best = max (lowest index among equals), better = second largest of the multiset
    const bool other_wins = (ob > a.best) || (ob == a.best && oi < a.best_i);
    const float lo = other_wins ? a.best : ob;
    Bid3 r;
    r.best = other_wins ? ob : a.best;
    r.best_i = other_wins ? oi : a.best_i;
    r.better = fmaxf(fmaxf(a.better, obt), lo);
    return r;
}

__device__ __forceinline__ float ld_agent_f(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent_i(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent_f(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent_i(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// emd_cuda.cu:10-20 -- float atomicMax by compare-and-swap (replace only when val > stored)
__device__ __forceinline__ void atomic_max_f(float *addr, float val)
{
    int old = __float_as_int(ld_agent_f(addr));
    while (val > __int_as_float(old)) {
        const int seen = atomicCAS((int *)addr, old, __float_as_int(val));
        if (seen == old) break;
        old = seen;
    }
}

__global__ __launch_bounds__(EMD_THREADS) void emd_auction_kernel(int n, const float *__restrict__ xyz1_all, const float *__restrict__ xyz2_all,
                                                                  float *dist_all, int *assignment_all, float *price_all, int *assignment_inv_all,
                                                                  int *bid_all, float *bid_inc_all, float *max_inc_all, int *max_idx_all, float eps,
                                                                  int iters)
{
    extern __shared__ float lds[];
    float *x2 = lds, *y2 = lds + n, *z2 = lds + 2 * n, *pr = lds + 3 * n;
    unsigned short *unass = (unsigned short *)(lds + 4 * n);
    __shared__ int wcnt[EMD_WAVES];

    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t off = (size_t)i * n;
    const float *xyz1 = xyz1_all + off * 3, *xyz2 = xyz2_all + off * 3;
    int *ass = assignment_all + off, *ass_inv = assignment_inv_all + off, *bid = bid_all + off, *max_idx = max_idx_all + off;
    float *price = price_all + off, *binc = bid_inc_all + off, *minc = max_inc_all + off, *dist = dist_all + off;

    for (int k = tid; k < n; k += EMD_THREADS) {
        x2[k] = xyz2[k * 3 + 0];
        y2[k] = xyz2[k * 3 + 1];
        z2[k] = xyz2[k * 3 + 2];
        pr[k] = price[k];
    }
    __syncthreads();

    for (int it = 0; it < iters; ++it) {
        const bool last = (it == iters - 1);
        // ---- list the unassigned points in ascending order (calc_unass_cnt / _sum / _idx, :30-93) -------------
        int U = 0;
        for (int c = 0; c < n; c += EMD_THREADS) {
            const int j = c + tid;
            const bool un = ld_agent_i(ass + j) == -1;
            const u64 mask = __ballot(un);
            if (lane == 0) wcnt[wave] = __popcll(mask);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < EMD_WAVES; ++w) {
                const int cw = wcnt[w];
                before += (w < wave) ? cw : 0;
                total += cw;
            }
            if (un) unass[U + before + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)j;
            U += total;
            __syncthreads();
        }
        if (U == 0) break;   // every later round would find nothing to do

        // ---- Bid (:95-179): one wavefront per unassigned point ----------------------------------------------
        for (int u = wave; u < U; u += EMD_WAVES) {
            const int j = unass[u];
            const float x1 = xyz1[j * 3 + 0], y1 = xyz1[j * 3 + 1], z1 = xyz1[j * 3 + 2];
            Bid3 b = {-1e9f, -1e9f, -1};
            for (int k = lane; k < n; k += 64) {
                const float dx = x2[k] - x1, dy = y2[k] - y1, dz = z2[k] - z1;
                const float d = (float)(3.0 - (double)sqrtf(dx * dx + dy * dy + dz * dz) - (double)pr[k]);
                if (d > b.best) {
                    b.better = b.best;
                    b.best = d;
                    b.best_i = k;
                } else if (d > b.better) {
                    b.better = d;
                }
            }
#pragma unroll
            for (int s = 1; s < 64; s <<= 1) {
                const float ob = __shfl_xor(b.best, s), obt = __shfl_xor(b.better, s);
                const int oi = __shfl_xor(b.best_i, s);
                b = bid_merge(b, ob, obt, oi);
            }
            if (lane == 0) {
                const float inc = b.best - b.better + eps;
                bid[j] = b.best_i;
                binc[j] = inc;
                if (b.best_i >= 0) atomic_max_f(minc + b.best_i, inc);
            }
        }
        __syncthreads();

        // ---- GetMax (:181-194): the highest-indexed bidder whose increment matches the maximum within 1e-6 ---
        for (int pass = 0; pass < 2; ++pass) {
            for (int u = tid; u < U; u += EMD_THREADS) {
                const int j = unass[u];
                const int bid_id = bid[j];
                if (bid_id < 0) continue;
                const double bi = (double)binc[j], mx = (double)ld_agent_f(minc + bid_id);
                if (bi - 1e-6 <= mx && mx <= bi + 1e-6) {
                    if (pass == 0) st_agent_i(max_idx + bid_id, -1);
                    else atomicMax(max_idx + bid_id, j);
                }
            }
            __syncthreads();
        }

        // ---- Assign (:196-215) --------------------------------------------------------------------------------
        if (!last) {
            for (int u = tid; u < U; u += EMD_THREADS) {
                const int j = unass[u];
                const int bid_id = bid[j];
                if (bid_id < 0 || ld_agent_i(max_idx + bid_id) != j) continue;
                const int prev = ld_agent_i(ass_inv + bid_id);
                if (prev != -1) st_agent_i(ass + prev, -1);
                st_agent_i(ass_inv + bid_id, j);
                st_agent_i(ass + j, bid_id);
                pr[bid_id] += binc[j];
                st_agent_f(minc + bid_id, -1e9f);
            }
        } else if (tid == 0) {
            // last round: every unassigned point takes the object it bid for, applied in ascending point order
            for (int u = 0; u < U; ++u) {
                const int j = unass[u];
                const int bid_id = bid[j];
                st_agent_i(ass + j, bid_id);
                if (bid_id < 0) continue;
                st_agent_i(ass_inv + bid_id, j);
                pr[bid_id] += binc[j];
                st_agent_f(minc + bid_id, -1e9f);
            }
        }
        __syncthreads();
    }

    // ---- CalcDist (:217-226) + the prices back to the caller's state tensor -------------------------------------
    for (int j = tid; j < n; j += EMD_THREADS) {
        price[j] = pr[j];
        const int k = ld_agent_i(ass + j);
        float d = 0.f;
        if (k >= 0) {
            const float dx = xyz1[j * 3 + 0] - x2[k], dy = xyz1[j * 3 + 1] - y2[k], dz = xyz1[j * 3 + 2] - z2[k];
            d = dx * dx + dy * dy + dz * dz;
        }
        dist[j] = d;
    }
}

// NmDistanceGradKernel (:284-300): grad_xyz[i,j] += 2 g (xyz1[i,j] - xyz2[i,idx[i,j]]); one thread owns one point.
__global__ void emd_backward_kernel(long total, int n, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                    const float *__restrict__ grad_dist, const int *__restrict__ idx, float *grad_xyz)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const long i = t / n;
    const int j2 = idx[t];
    const float g = grad_dist[t] * 2;
    const float *p = xyz1 + t * 3, *q = xyz2 + (i * n + j2) * 3;
    float *o = grad_xyz + t * 3;
    o[0] += g * (p[0] - q[0]);
    o[1] += g * (p[1] - q[1]);
    o[2] += g * (p[2] - q[2]);
}

}  // namespace

PA_API int pa_emd_forward(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *assignment, float *price,
                          int *assignment_inv, int *bid, float *bid_increments, float *max_increments, int *max_idx, float eps, int iters,
                          pa_stream_t stream)
{
    // emd_cuda.cu:236-249 -- the reference's shape rules (it returns -1 for these)
    if (n != m) { pa_set_error("pa_emd_forward: the two point clouds must have the same size (n=%d, m=%d)", n, m); return PA_EUNSUPPORTED; }
    if (b > 512) { pa_set_error("pa_emd_forward: batch size %d > 512", b); return PA_EUNSUPPORTED; }
    if (n % 1024 != 0 || n <= 0) { pa_set_error("pa_emd_forward: n=%d is not a positive multiple of 1024", n); return PA_EUNSUPPORTED; }
    if (n > 8192) { pa_set_error("pa_emd_forward: n=%d > 8192 does not fit the on-chip auction state", n); return PA_EUNSUPPORTED; }
    PA_REQUIRE(b > 0 && iters >= 1, "pa_emd_forward: need b >= 1 and iters >= 1 (b=%d, iters=%d)", b, iters);
    PA_REQUIRE(xyz1 && xyz2 && dist && assignment && price && assignment_inv && bid && bid_increments && max_increments && max_idx,
               "pa_emd_forward: null pointer");
    const size_t lds_bytes = (size_t)n * 16 + (size_t)n * 2;
    {   // the opt-in is a per-DEVICE attribute of the function: set it on every call (a process may drive several GPUs)
        hipError_t e = hipFuncSetAttribute((const void *)emd_auction_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e != hipSuccess) { pa_set_error("pa_emd_forward: cannot raise the LDS limit: %s", hipGetErrorString(e)); return (int)e; }
    }
    hipLaunchKernelGGL(emd_auction_kernel, dim3(b), dim3(EMD_THREADS), lds_bytes, (hipStream_t)stream, n, xyz1, xyz2, dist, assignment, price,
                       assignment_inv, bid, bid_increments, max_increments, max_idx, eps, iters);
    PA_CHECK_LAUNCH("pa_emd_forward");
    return PA_OK;
}

PA_API int pa_emd_backward(int b, int n, const float *xyz1, const float *xyz2, float *grad_xyz, const float *grad_dist, const int *idx,
                           pa_stream_t stream)
{
    PA_REQUIRE(b > 0 && n > 0 && xyz1 && xyz2 && grad_xyz && grad_dist && idx, "pa_emd_backward: bad arguments");
    const long total = (long)b * n;
    hipLaunchKernelGGL(emd_backward_kernel, dim3(pa_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, total, n, xyz1, xyz2, grad_dist, idx,
                       grad_xyz);
    PA_CHECK_LAUNCH("pa_emd_backward");
    return PA_OK;
}
