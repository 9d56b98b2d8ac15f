// K1 -- furthest point sampling for gfx950.
//
// Reference semantics: libs/pointops/src/sampling/sampling_cuda_kernel.cu:59-168.
// Round j: temp[k] = fminf(d2(p_k, p_old), temp[k]) for every k; the next sample is the argmax of temp under
// the total order the reference's strided scan + stride-halving LDS tree induce:
//     (temp larger first, then bitrev_{log2 bs}(k mod bs) smaller first, then k smaller first),  bs = opt_n_threads(n).
//
// MI355X design (not the reference's): the reference round-trips temp through global memory and runs a
// 10-barrier LDS tree per round.  Here one workgroup owns one cloud, every lane keeps its points AND their
// running minima in VGPRs for the whole launch, and the order above is folded into one unsigned 64-bit key
//     key = float_bits(temp) << 32 | ~rank(k),   rank(k) = bitrev(k mod bs) << qbits | (k / bs)
// so that a plain unsigned max is the reference's argmax (temp >= 0, so its bit pattern is monotone).
// The wave maximum is taken with DPP row shifts / broadcasts (no LDS), waves meet through one double-buffered
// LDS slot array with ONE barrier per round, and the winner's coordinates come from an LDS copy of the cloud.
#include <stdlib.h>

#include "pa_common.h"


namespace {

struct FpsOrder {
    int log2bs;  // log2 of the reference's block size for this n
    int qbits;   // bits needed for k / bs
};

__device__ __forceinline__ u32 fps_lowkey(int k, FpsOrder o)
{
    const u32 r = o.log2bs ? (__brev((u32)k & ((1u << o.log2bs) - 1u)) >> (32 - o.log2bs)) : 0u;
    const u32 rank = (r << o.qbits) | ((u32)k >> o.log2bs);
    return ~rank;
}

__device__ __forceinline__ int fps_decode(u32 lowkey, FpsOrder o)
{
    const u32 rank = ~lowkey;
    const u32 q = rank & ((1u << o.qbits) - 1u);
    const u32 rb = rank >> o.qbits;
    const u32 r = o.log2bs ? (__brev(rb) >> (32 - o.log2bs)) : 0u;
    return (int)((q << o.log2bs) | r);
}

// fminf(d, t) for a d that comes out of arithmetic and a running minimum t: v_min_f32 returns the other operand for a quiet NaN exactly as fminf does; the
// compiler's form first quiets a possible SIGNALLING NaN in t (v_max_f32 t, t -- one more instruction per point and round), which only a caller-supplied
// temp buffer could hold: the launcher's load canonicalises those once instead.
__device__ __forceinline__ float fps_min(float d, float t)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(d), "v"(t));
    return r;
}

// value pinned to vector registers (an empty asm with a "+v" operand is opaque to the uniformity analysis)
__device__ __forceinline__ u64 fps_in_vgpr(u64 v)
{
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    asm("" : "+v"(lo), "+v"(hi));          // not volatile: the four slot reads stay one batch of loads
    return ((u64)hi << 32) | lo;
}

// Register-resident path: NT threads, PPT points per lane, n <= NT*PPT.
// LDSXYZ = true: cloud copy in LDS ((x, y, z, -) by rank, 16 B/point: 64 KB at n = 4096), the winner's coordinates are read from it after the barrier.
// LDSXYZ = false (opt-in, see fps_lds_xyz below): no cloud copy.  Each wave fetches ITS winner's coordinates from the owning lane's registers (uniform register index
//   + v_readlane) and publishes (key, x, y, z) in its slot before the round's single barrier; after it every thread takes the slot
//   with the largest key.  LDS: 256 bytes, so the 0.7 ms launch no longer keeps LDS-heavy kernels of other streams (the 132 KB
//   chain workgroups, the 80 KB NetVLAD ones) off the CUs it runs on; the post-barrier critical path loses one LDS round trip.
struct FpsSlot { u64 key; float x, y, z; float pad[3]; };   // 32 bytes

// CPW: clouds per WORKGROUP (1, 2 or 3): cloud c of a workgroup is the threads [c NT, (c + 1) NT) with their own LDS copy and slot array; the
// one barrier of a round is shared (every cloud runs the same m - 1 rounds).  Not faster per cloud -- the point is WHERE the launch sits: a
// batch's sampling chain keeps its CUs for 0.7 ms while other streams' dense kernels run, and a chain workgroup with 132 KB of LDS cannot share
// a CU with a 48 KB sampling workgroup: with one cloud per workgroup a batch of 32 takes 32 of the 256 CUs away from them, with three 11.
template <int NT, int PPT, bool LDSXYZ, int CPW = 1>
__global__ __launch_bounds__(NT * CPW) void fps_reg_kernel(int n, int m, FpsOrder ord, const float *__restrict__ xyz_all,
                                                       float *__restrict__ temp_all, int *__restrict__ idx_all, float *__restrict__ new_xyz_all,
                                                       int j_begin = 0, int j_end = -1, int nclouds = 0)
{
    // [j_begin, j_end): the samples this launch produces (default: all m).  A launch that starts at j_begin > 0 RESUMES: the running minima come
    // from temp (which the previous launch wrote back) and the last selected point from idx[j_begin - 1] -- the sampling order is prefix-stable
    // (sampling_cuda_kernel.cu:59-168: round j only needs the minima after round j - 1), so chunked launches give the same samples bit for bit
    // and whatever consumes the first samples can start while the later ones are still being drawn (engine latency mode).
    if (j_end < 0) j_end = m;
    constexpr int NW = NT / 64;
    constexpr int LOG_NT = NT == 64 ? 6 : NT == 128 ? 7 : NT == 256 ? 8 : NT == 512 ? 9 : 10;
    static_assert((1 << LOG_NT) == NT, "NT must be a power of two");
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int sub = CPW > 1 ? (int)threadIdx.x / NT : 0;                  // this thread's cloud inside the workgroup
    // The LDS copy of the cloud is indexed by RANK (the key's low half is ~rank): the winner's coordinates are one 16-byte read at (~low) * 16 straight
    // from the selected key -- the decode of the point index (not, shift, bit reverse, shift, mask, shift, or) leaves the round's serial chain and only
    // thread 0 runs it, for the index it stores.  Ranks without a point (n below a multiple of the reference's block size) are never read.
    const int nrank = 1 << (ord.log2bs + ord.qbits);
    const int per_cloud = 4 * nrank + 2 * NW * 2;                         // floats of LDS per cloud: (x, y, z, -) by rank + [2][NW] 64-bit slots
    float *smem = smem_all + (CPW > 1 ? sub * per_cloud : 0);
    float4 *sp = reinterpret_cast<float4 *>(smem);
    u64 *slots = reinterpret_cast<u64 *>(smem + 4 * nrank);              // [2][NW]
    FpsSlot *rslots = reinterpret_cast<FpsSlot *>(smem);                  // LDSXYZ = false: [2][NW]

    const int tid = CPW > 1 ? (int)threadIdx.x - sub * NT : (int)threadIdx.x;
    const int b_raw = CPW > 1 ? (int)blockIdx.x * CPW + sub : (int)blockIdx.x;
    const bool ghost = CPW > 1 && b_raw >= nclouds;                       // padding cloud of the last workgroup: computes on the last real cloud, writes nothing
    const int b = ghost ? nclouds - 1 : b_raw;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    float *temp = temp_all ? temp_all + (size_t)b * n : nullptr;   // nullptr: start from 1e10 (pointops.py:21), do not write back
    int *idxs = idx_all + (size_t)b * m;
    float *nxyz = new_xyz_all ? new_xyz_all + (size_t)b * m * 3 : nullptr;   // optional fused gather of the sampled coordinates

    // coordinates as vector values: a register set indexed by a wave-uniform p is then one s_set_gpr_idx_on + v_mov (LDSXYZ = false)
    typedef float vecp __attribute__((ext_vector_type(PPT)));
    vecp px, py, pz;
    float t[PPT];
    u32 low[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const int k = tid + p * NT;
        if (k < n) {
            px[p] = xyz[k * 3 + 0];
            py[p] = xyz[k * 3 + 1];
            pz[p] = xyz[k * 3 + 2];
            t[p] = temp ? __builtin_canonicalizef(temp[k]) : 1e10f;     // quiets a signalling NaN once (see fps_min)
            low[p] = fps_lowkey(k, ord);
            if (LDSXYZ) sp[~low[p]] = make_float4(px[p], py[p], pz[p], 0.f);
        } else {  // padding: key 0 never beats a real point (real low keys are >= 1)
            px[p] = py[p] = pz[p] = 0.f;
            t[p] = 0.f;
            low[p] = 0u;
        }
    }
    const int first = j_begin > 0 ? idxs[j_begin - 1] : 0;       // written by the previous launch of the chain (same stream)
    if (tid == 0 && j_begin == 0 && !ghost) idxs[0] = 0;
    float ox, oy, oz;
    if (LDSXYZ) {
        if (NW > 1 && tid < 3) slots[tid] = 0;
        __syncthreads();
        const float4 o4 = sp[~fps_lowkey(first, ord)];
        ox = o4.x; oy = o4.y; oz = o4.z;
    } else {
        ox = xyz[first * 3]; oy = xyz[first * 3 + 1]; oz = xyz[first * 3 + 2];
    }
    if (tid == 0 && nxyz && j_begin == 0 && !ghost) { nxyz[0] = ox; nxyz[1] = oy; nxyz[2] = oz; }

    // The round's arithmetic on PAIRS of points (PAIRED): the packed fp32 instructions of gfx950 (v_pk_add_f32 / v_pk_mul_f32: two IEEE operations per lane and
    // issue slot, each rounded exactly like its scalar form, so d is the same bits) take two points' dx, dy, dz, their squares and the two sums in eight
    // instructions; the compiler's own packing of the scalar loop pairs (z, x) of ONE point and leaves y scalar: six per point, plus a v_max t, t in front
    // of every fminf (quieting a signalling NaN the running minimum can never be).  Ten instructions per pair against sixteen: the round is ~60 % VALU
    // issue at one wave per SIMD (the rest is the DPP maximum and three LDS round trips).
    constexpr bool PAIRED = LDSXYZ && PPT % 2 == 0 && PPT >= 2;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 qx[PAIRED ? PPT / 2 : 1], qy[PAIRED ? PPT / 2 : 1], qz[PAIRED ? PPT / 2 : 1];
    if (PAIRED) {
#pragma unroll
        for (int h = 0; h < PPT / 2; ++h) {
            qx[h] = (f2){px[2 * h], px[2 * h + 1]};
            qy[h] = (f2){py[2 * h], py[2 * h + 1]};
            qz[h] = (f2){pz[2 * h], pz[2 * h + 1]};
        }
    }
    int slot_i = 0;                                            // this round's selection slot (three rotate, see below)
    for (int j = max(j_begin, 1); j < j_end; ++j) {
        // (a two-pass form -- the largest minimum with three-operand maxima, then the largest low key among the points that have it: 48 instead of
        // ~110 instructions -- was measured SLOWER, 0.755 vs 0.69 us per round: the second pass cannot start before the first ends, while the running
        // 64-bit maximum overlaps the distance arithmetic)
        u64 best = 0;
        if (PAIRED) {
            // (-o, -o) as materialised register pairs, the subtraction as a plain packed add of them (x - o and x + (-o) are the same IEEE operation): the
            // compiler's own form folds the broadcast and the negation into operand modifiers of v_pk_add_f32 (op_sel / neg_lo / neg_hi), and that form
            // returns wrong low halves on gfx950 while a neighbouring wave of the SIMD issues 16x16x32 MFMAs (every fp16 chain kernel
            // does): pa_common.h, pa_pk_plain; tools/probes/pk_f32_victim.hip is the self-contained reproduction.  Three v_xor and
            // three v_mov per round: 484 -> 500 us per first-level launch.  (Measured and not kept: the lane's points negated once and the selected point
            // read as ready-made (x, x) pairs by three ds_read2_b32 with both offsets on one word -- no VALU on the chain, but 554 us.)
            const f2 n2x = pa_pk_plain((f2){-ox, -ox}), n2y = pa_pk_plain((f2){-oy, -oy}), n2z = pa_pk_plain((f2){-oz, -oz});
#pragma unroll
            for (int h = 0; h < PPT / 2; ++h) {
                const f2 dx = qx[h] + n2x, dy = qy[h] + n2y, dz = qz[h] + n2z;
                const f2 d = dx * dx + dy * dy + dz * dz;     // sampling_cuda_kernel.cu:93, two points
                t[2 * h] = fps_min(d.x, t[2 * h]);            // :94
                t[2 * h + 1] = fps_min(d.y, t[2 * h + 1]);
                best = pa_max_u64(best, pa_make_key(t[2 * h], low[2 * h]));
                best = pa_max_u64(best, pa_make_key(t[2 * h + 1], low[2 * h + 1]));
            }
        } else {
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                const float dx = px[p] - ox, dy = py[p] - oy, dz = pz[p] - oz;
                const float d = dx * dx + dy * dy + dz * dz;  // sampling_cuda_kernel.cu:93
                t[p] = fminf(d, t[p]);                        // :94
                best = pa_max_u64(best, pa_make_key(t[p], low[p]));
            }
        }
        u64 g;
        int old;
        if (LDSXYZ) {
            if (NW > 1) {
                // Selection across the workgroup by the LDS atomic unit: ONE wave-level reduction (the largest minimum, six DPP steps), then the lanes
                // that hold it -- one per wave unless minima tie exactly -- send their whole key to ds_max_u64 on this round's slot; the unit orders equal
                // minima by the low half and the four waves against each other, and after the barrier the slot IS the selected key.  Against the form it
                // replaces (second DPP reduction for the low half, a slot per wave, four slot reads and three 64-bit compare / select steps in every
                // wave) the round loses ~30 dependent instructions: 0.58 -> see profiles/r04_ab_log.txt.  Three slots rotate: round j's is cleared
                // during round j - 1 (thread 0, before that round's barrier; its last readers passed the barrier of round j - 2 ... j - 1 before).
                const u32 hi = (u32)(best >> 32);
                const u32 H = pa_wave_max_u32(hi);
                u64 *cur = slots + slot_i;
                slot_i = slot_i == 2 ? 0 : slot_i + 1;
                if (tid == 0) slots[slot_i] = 0;
                // (as inline assembly: hipcc's atomic optimizer rewrites __hip_atomic_fetch_max into a scalar loop over the active lanes -- fourteen
                // instructions and two branches to save an LDS operation that one lane issues anyway; the wait belongs to the assembly too, the
                // compiler's counter bookkeeping does not see the operation)
                if (hi == H) asm volatile("ds_max_u64 %0, %1" : : "v"((u32)(uintptr_t)cur), "v"(best) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
                __syncthreads();
                g = fps_in_vgpr(*cur);
            } else {
                g = pa_wave_max_key2(best);
            }
            const float4 o4 = sp[~(u32)g];
            ox = o4.x; oy = o4.y; oz = o4.z;
            old = 0;
            if (tid == 0) old = fps_decode((u32)g, ord);
        } else {
            // this wave's winner: point kw = owner thread + p * NT lives in register set p of lane kw % 64 of THIS wave (g is wave-uniform)
            g = pa_wave_max_key2(best);
            const int kw = fps_decode((u32)g, ord);
            const int pw = __builtin_amdgcn_readfirstlane(kw >> LOG_NT), lw = __builtin_amdgcn_readfirstlane(kw & 63);
            const int pi = pw < PPT ? pw : 0;                             // a wave of padding lanes only decodes garbage (its key never wins)
            float cx = px[pi], cy = py[pi], cz = pz[pi];
            cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), lw));
            cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), lw));
            cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cz), lw));
            if (NW > 1) {
                FpsSlot *s = rslots + (j & 1) * NW;
                if ((tid & 63) == 0) { s[tid >> 6].key = g; s[tid >> 6].x = cx; s[tid >> 6].y = cy; s[tid >> 6].z = cz; }
                __syncthreads();
                u64 gk[NW];
                float gx[NW], gy[NW], gz[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) { gk[w] = s[w].key; gx[w] = s[w].x; gy[w] = s[w].y; gz[w] = s[w].z; }
                g = gk[0]; cx = gx[0]; cy = gy[0]; cz = gz[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    const bool up = gk[w] > g;
                    g = up ? gk[w] : g; cx = up ? gx[w] : cx; cy = up ? gy[w] : cy; cz = up ? gz[w] : cz;
                }
            }
            old = fps_decode((u32)g, ord);
            ox = cx; oy = cy; oz = cz;
        }
        if (tid == 0 && !ghost) {
            idxs[j] = old;
            if (nxyz) { nxyz[j * 3 + 0] = ox; nxyz[j * 3 + 1] = oy; nxyz[j * 3 + 2] = oz; }
        }
    }
    if (temp && !ghost) {
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            const int k = tid + p * NT;
            if (k < n) temp[k] = t[p];
        }
    }
}

// Measured and not kept (round 2): block pruning.  With the cloud counting-sorted into Morton cells a register set of a wave is a compact
// block of 64 points with a bounding sphere (c, r); every running minimum is <= M (the minimum of the point just selected), so a block with
// |o - c| >= (r + sqrt(M)) * 1.0005 cannot change and its nine instructions per point can be skipped -- exact (bit-identical samples on every
// test cloud), and only 20-25 % of the blocks are touched per round on uniform and plane-like clouds.  But a block is ONE point per lane:
// the skip is a wave-uniform branch around nine VALU instructions, sixteen of them per round, and a taken s_cbranch costs about as much as
// the instructions it jumps over; the per-lane key maximum over all sixteen sets (48 instructions) cannot be skipped.  Result at n = 4096,
// m = 1024: 0.95 us per round against 0.70 (973 vs 721 us per launch).  Skipping in groups of four sets would recover ~10 % at best.

// Any-n path: temp stays in global memory (as in the reference), same key order, 1024 threads.
__global__ __launch_bounds__(1024) void fps_stream_kernel(int n, int m, FpsOrder ord, const float *__restrict__ xyz_all,
                                                            float *__restrict__ temp_all, int *__restrict__ idx_all)
{
    constexpr int NT = 1024, NW = NT / 64;
    __shared__ u64 slots[2 * NW];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *xyz = xyz_all + (size_t)b * n * 3;
    float *temp = temp_all + (size_t)b * n;
    int *idxs = idx_all + (size_t)b * m;
    if (tid == 0) idxs[0] = 0;
    int old = 0;
    for (int j = 1; j < m; ++j) {
        const float ox = xyz[old * 3 + 0], oy = xyz[old * 3 + 1], oz = xyz[old * 3 + 2];
        u64 best = 0;
        for (int k = tid; k < n; k += NT) {
            const float dx = xyz[k * 3 + 0] - ox, dy = xyz[k * 3 + 1] - oy, dz = xyz[k * 3 + 2] - oz;
            const float d = dx * dx + dy * dy + dz * dz;
            const float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            best = pa_max_u64(best, pa_make_key(d2, fps_lowkey(k, ord)));
        }
        u64 g = pa_wave_max_u64(best);
        u64 *s = slots + (j & 1) * NW;
        if ((tid & 63) == 0) s[tid >> 6] = g;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; ++w) g = pa_max_u64(g, s[w]);
        old = fps_decode((u32)g, ord);
        if (tid == 0) idxs[j] = old;
    }
}

// The launcher instantiates ONE form: cloud copy in LDS, one cloud per workgroup.  Forms that were built, bit-exact and slower (numbers in
// DESIGN.md's appendix; the kernel template keeps their branches compiled out: LDSXYZ = false, CPW > 1): the 256-byte-LDS form with the winner's
// coordinates from the owner's registers (the round 0.70 -> 0.86 us), two / three clouds per workgroup (0.76 -> 0.97 / 1.28 ms per launch),
// 512 / 1024 threads per cloud (re-measured in round 5 on the ds_max_u64 selection: 485 -> 462 / 491 us per launch, no change in the pipeline rate).

template <int NT, int PPT>
int launch_reg(int b, int n, int m, FpsOrder ord, const float *xyz, float *temp, int *idx, float *new_xyz, hipStream_t st, int j_begin = 0, int j_end = -1)
{
    size_t lds = ((size_t)16 << (ord.log2bs + ord.qbits)) + 2 * (NT / 64) * 8;      // the cloud by rank (see the kernel) + the slot pairs
    // LDS RESERVE of the long sampling chains (round 6).  A first-level launch is ~1000 strictly serial rounds on one CU per cloud; every other stream's
    // workgroup that becomes resident beside it takes issue slots and LDS bandwidth from those rounds (0.502 -> 0.550 us per launch inside the four-stream
    // pipeline), and the step follows the chain's length.  Asking for 128 KB instead of the 64 KB the cloud copy needs keeps all but the smallest
    // workgroups (<= 32 KB) off the sampling CUs: + 1.0 / + 1.8 / + 0.6 % of the pipeline's rate at 104 / 128 / 156 KB (profiles/r06_ab_log.txt).  Only
    // for chains of >= 512 rounds over clouds that fill a workgroup's registers (the first level of both models); PA_FPS_LDS_RESERVE overrides (bytes, 0 = off).
    static const long reserve = getenv("PA_FPS_LDS_RESERVE") ? atol(getenv("PA_FPS_LDS_RESERVE")) : 128 * 1024;
    // ... and only for launches of at most one workgroup per CU (b <= 256: a batch's own sampling, the look-ahead pipeline's groups of <= 8 batches).  A
    // launch over MORE clouds than CUs (the look-ahead pipeline's default group: 16 batches = 512 clouds) keeps the 64 KB of the copy: two sampling
    // workgroups share a CU and the group is ONE round of workgroups instead of two -- half the CU-time per cloud, and the chain's length is hidden a
    // group ahead: 42.9 k against 42.0-42.4 k submaps/s with the reserve (profiles/r06_ab_log.txt).  PA_FPS_LDS_RESERVE_ALWAYS = A/B knob.
    static const bool always = getenv("PA_FPS_LDS_RESERVE_ALWAYS") != nullptr;
    if (b <= 256 || always)
    if (m >= 512 && n >= 2048 && lds < (size_t)reserve && reserve <= 160 * 1024 - 2048) lds = (size_t)reserve;
    if (lds > 48 * 1024)  // opt in to the large-LDS carve-out (gfx950: 160 KiB per CU)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&fps_reg_kernel<NT, PPT, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((fps_reg_kernel<NT, PPT, true>), dim3(b), dim3(NT), lds, st, n, m, ord, xyz, temp, idx, new_xyz, j_begin, j_end, b);
    return 0;
}

}  // namespace

static int fps_dispatch(int b, int n, int m, const float *xyz, float *temp, int *idx, float *new_xyz, pa_stream_t stream, int j_begin = 0, int j_end = -1)
{
    PA_REQUIRE(b > 0 && n > 0, "pa_furthestsampling: b=%d n=%d must be positive", b, n);
    PA_REQUIRE(xyz && idx, "pa_furthestsampling: null pointer");
    if (m <= 0) return PA_OK;  // sampling_cuda_kernel.cu:61-62
    hipStream_t st = (hipStream_t)stream;
    const int bs = pa_opt_n_threads(n);
    FpsOrder ord;
    ord.log2bs = 0;
    while ((1 << ord.log2bs) < bs) ++ord.log2bs;
    const int Q = (n + bs - 1) / bs;
    ord.qbits = 0;
    while ((1 << ord.qbits) < Q) ++ord.qbits;

    if (n <= 64) launch_reg<64, 1>(b, n, m, ord, xyz, temp, idx, new_xyz, st, j_begin, j_end);
    else if (n <= 128) launch_reg<64, 2>(b, n, m, ord, xyz, temp, idx, new_xyz, st, j_begin, j_end);
    else if (n <= 256) launch_reg<64, 4>(b, n, m, ord, xyz, temp, idx, new_xyz, st, j_begin, j_end);
    else if (n <= 512) launch_reg<64, 8>(b, n, m, ord, xyz, temp, idx, new_xyz, st, j_begin, j_end);
    else if (n <= 1024) launch_reg<256, 4>(b, n, m, ord, xyz, temp, idx, new_xyz, st, j_begin, j_end);
    else if (n <= 2048) launch_reg<256, 8>(b, n, m, ord, xyz, temp, idx, new_xyz, st, j_begin, j_end);
    else if (n <= 4096) {
        // 512 threads (two waves per SIMD, eight points per lane) since round 6: 0.46 vs 0.50 ms per first-level launch; together with the LDS reserve
        // of launch_reg + 1.2 % of the four-stream rate (alone + 0.3 %; round 5 measured "no change" without the reserve).  PA_FPS_NT256 = the former form.
        static const bool nt256 = getenv("PA_FPS_NT256") != nullptr;
        if (nt256) launch_reg<256, 16>(b, n, m, ord, xyz, temp, idx, new_xyz, st, j_begin, j_end);
        else launch_reg<512, 8>(b, n, m, ord, xyz, temp, idx, new_xyz, st, j_begin, j_end);
    }
    else if (n <= 8192) launch_reg<256, 32>(b, n, m, ord, xyz, temp, idx, new_xyz, st, j_begin, j_end);
    else {
        PA_REQUIRE(j_begin == 0 && j_end < 0, "pa_furthestsampling_range: clouds above 8192 points are sampled in one launch");
        PA_REQUIRE(temp && !new_xyz, "pa_furthestsampling: clouds above 8192 points need the caller's temp buffer and no fused gather");
        hipLaunchKernelGGL(fps_stream_kernel, dim3(b), dim3(1024), 0, st, n, m, ord, xyz, temp, idx);
    }
    PA_CHECK_LAUNCH("pa_furthestsampling");
    return PA_OK;
}


PA_API int pa_furthestsampling(int b, int n, int m, const float *xyz, float *temp, int *idx, pa_stream_t stream)
{
    PA_REQUIRE(temp, "pa_furthestsampling: null temp");
    return fps_dispatch(b, n, m, xyz, temp, idx, nullptr, stream);
}

// Fused form used by the inference engine: running minima start at 1e10 and stay in registers (no temp tensor),
// and the sampled coordinates new_xyz (b, m, 3) are written alongside idx (replaces the gathering call of
// patch_aug_net.py:222-225).  n <= 8192.
PA_API int pa_furthestsampling_gather(int b, int n, int m, const float *xyz, int *idx, float *new_xyz, pa_stream_t stream)
{
    PA_REQUIRE(new_xyz, "pa_furthestsampling_gather: null new_xyz");
    PA_REQUIRE(n <= 8192, "pa_furthestsampling_gather: n=%d exceeds the register-resident limit 8192", n);
    return fps_dispatch(b, n, m, xyz, nullptr, idx, new_xyz, stream);
}

// Samples [j_begin, j_end) of the m-sample sequence (n <= 8192): the first launch of a chain (j_begin = 0) starts the running minima temp (b, n) at
// 1e10 itself -- the caller need not fill it -- every launch writes them back, a later launch resumes from them and from idx[j_begin - 1].  The
// chain of launches (same stream, ascending ranges covering [0, m)) produces exactly pa_furthestsampling_gather's idx / new_xyz.
PA_API int pa_furthestsampling_range(int b, int n, int m, int j_begin, int j_end, const float *xyz, float *temp, int *idx, float *new_xyz, pa_stream_t stream)
{
    PA_REQUIRE(temp && n <= 8192, "pa_furthestsampling_range: needs the running-minima buffer and n <= 8192 (n=%d)", n);
    PA_REQUIRE(j_begin >= 0 && j_begin < j_end && j_end <= m, "pa_furthestsampling_range: bad range [%d, %d) of %d", j_begin, j_end, m);
    if (j_begin == 0 && pa_fill32(temp, 0x501502f9u /* 1e10f */, (size_t)b * n, (hipStream_t)stream) != PA_OK) {      // a kernel, not a memset node (pa_common.h)
        pa_set_error("pa_furthestsampling_range: could not initialise temp");
        return PA_EINVAL;
    }
    return fps_dispatch(b, n, m, xyz, temp, idx, new_xyz, stream, j_begin, j_end);
}
