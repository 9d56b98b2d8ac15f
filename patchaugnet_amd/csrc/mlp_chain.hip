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
#include "pa_chain_kernel.h"

namespace {

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

}  // namespace

// Generic entry point.  mode: 0 plain rows, 1 set-abstraction gather, 2 feature-propagation interpolate.
// wt[l] is K-major (kpad[l] x n[l]) with BN folded in and zero rows beyond the true K; bias[l] has n[l] entries.
int pa_chain16_launch(PaChain &a, int mode, bool is_pooled, bool split, int RTv, int scratch_floats, hipStream_t st);   // mlp_chain_f16.hip
int pa_fpx16_try(int nlayers, const void *const *wp16, const float *const *bias, const int *kpad, const int *nout, long rows, const float *g, const int *idx3,
                 const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1, const float *wskip, const float *bias0, float *out, int ldo,
                 long long *dbg, hipStream_t st);   // fpx_f16.hip
bool pa_sa_tiny_applies(const PaChain &a, int rt);                                                                      // sa_tiny.hip
int pa_sa_tiny_launch(const PaChain &a, int rt, long ntiles, hipStream_t st);
bool pa_sa_mid_applies(const PaChain &a, int rt);                                                                       // sa_mid.hip
int pa_sa_mid_launch(const PaChain &a, int rt, long ntiles, hipStream_t st);
int pa_fpx32_try(const PaChain &a, hipStream_t st);                                                                      // fpx_f32.hip
int pa_linear_lds_try(long rows, int k, int n, const float *x, int ldx, const float *wt, const float *bias, int relu, const float *residual, int ldr,
                      float *out, int ldo, hipStream_t st);                                                                    // linear_lds.hip

static int g_chain_tiny = -1;
// test / A/B switch for the persistent first-level kernel (sa_tiny.hip): 1 = wherever its shape applies, 0 = never, -1 = the default rule
PA_API void pa_chain_tiny_enable(int on) { g_chain_tiny = on; }
static int g_chain_mid = -1;
// same for the LDS-resident second-level kernel (sa_mid.hip); equal to the generic kernels up to the order of the fp32 additions, not bit for bit
PA_API void pa_chain_mid_enable(int on) { g_chain_mid = on; }

// pa_sa_group_window: applies to the NEXT pa_mlp_chain* call of this thread (mode 1, pooled), then resets
static thread_local int g_win_len = 0, g_win_off = 0;
PA_API int pa_sa_group_window(int win_len, int win_off)
{
    PA_REQUIRE(win_len >= 0 && win_off >= 0, "pa_sa_group_window: negative window");
    g_win_len = win_len; g_win_off = win_off;
    return PA_OK;
}

static long long *g_chain_dbg = nullptr;
// profiling hook (tools/chain_phases.py): device buffer of 512 x 8 int64 receiving s_memtime stamps of the next launches; NULL = off
PA_API void pa_chain_debug_buffer(long long *buf) { g_chain_dbg = buf; }
long long *pa_chain_dbg_ptr() { return g_chain_dbg; }

static int chain_dispatch(int mode, int pooled, int nlayers, const float *const *wt, const float *const *wpk, const float *const *bias, const int *kpad, const int *nout,
                          long rows, int k0,
                          const float *x, int ldx,
                          const float *xyz, const float *feat, const int *center_idx, const int *nbr_idx, int n_src, int m_ctr, int ns, int c_feat,
                          const float *known, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                          float *out, int ldo, int relu_last, const float *residual, int ldr, pa_stream_t stream,
                          const float *wskip = nullptr, const float *bias0 = nullptr, const void *const *wp16 = nullptr, int fold0 = 0, int col_slices = 1,
                          float *tap = nullptr, int ldtap = 0)
{
    const int win_len = g_win_len, win_off = g_win_off;      // pa_sa_group_window: consumed by this call whatever its outcome
    g_win_len = g_win_off = 0;
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
        // MODE_FP (interpolate + skip in the prologue) keeps the shared-tile tiling at EVERY size: its wave-private form is 2.3 x slower (fp1 at batch 64:
        // 282 vs 126 us, batch 256: 1150 vs 461 us -- the 3-NN gather prologue of a 32-row wave-private tile has nothing to hide under), and the former
        // 2048-tile threshold sat between batch 32 and batch 64 of the 1024-point level (profiles/r06_stage_scaling.txt)
        if (can_split && (t32 < split_below || mode == MODE_FP)) { split = true; RTv = t32 >= rt2_above ? 2 : 1; }
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
        PA_REQUIRE(rows % (win_len > 0 ? win_len : m_ctr) == 0, "pa_mlp_chain: SA rows=%ld must be B*m (B * window with pa_sa_group_window)", rows);
        PA_REQUIRE((rows / (win_len > 0 ? win_len : m_ctr)) * (long)n_src < 2147483647L, "pa_mlp_chain: B*n_src overflows int32 row ids");
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

    // the finest set-abstraction level of both models (<= 8 -> 32 -> 32 -> 64): persistent workgroups with the weights and the activations in
    // registers and the gather pipelined across tiles (sa_tiny.hip); equal to the generic pooled kernel up to the order of the fp32 additions
    static const bool no_tiny = getenv("PA_CHAIN_NO_TINY") != nullptr;                                               // A/B knob
    // (round 6: the register-chained kernel contracts in another order than the generic one, so the choice must not depend on the batch size: every
    // launch of this shape takes it)
    static const long tiny_min = getenv("PA_CHAIN_TINY_MIN_TILES") ? atol(getenv("PA_CHAIN_TINY_MIN_TILES")) : 1;  // test knob
    const bool tiny_on = g_chain_tiny < 0 ? (!no_tiny && ntiles >= tiny_min) : g_chain_tiny > 0;
    // the second set-abstraction level (67 -> 64 -> 64 -> n2): weights resident in LDS, activations in registers (sa_mid.hip)
    // Default rule (a function of the layer shapes only, never of the batch size: the kernel contracts in another order than the generic one):
    // last layer <= 128 wide (PPT-Net's level: 68 KB of LDS, 51 vs 68 us per launch, step + 1.3 %).  At 256 (PatchAugNet's: 100 KB, one workgroup
    // per CU) the launch alone is 37 vs 48 us, but inside the four-stream pipeline the step gets 0.3-1 % SLOWER (a workgroup that needs most of a
    // CU's LDS starts only where everything else has drained), so that shape stays on the shared-tile kernel (profiles/r04_ab_log.txt).
    static const bool no_mid = getenv("PA_CHAIN_NO_MID") != nullptr;                                                 // A/B knob
    static const int mid_max_n2 = getenv("PA_CHAIN_MID_MAX_N2") ? atoi(getenv("PA_CHAIN_MID_MAX_N2")) : 128;          // A/B knob
    const bool mid_on = g_chain_mid < 0 ? (!no_mid && nout[nlayers - 1] <= mid_max_n2) : g_chain_mid > 0;
    if (win_len > 0) {   // rows = clouds * win_len groups; only the persistent first-level kernel takes windows
        PA_REQUIRE(mode == MODE_SA && is_pooled && !split && pa_sa_tiny_applies(a, RTv) && win_off + win_len <= m_ctr && rows % win_len == 0,
                   "pa_sa_group_window: the next launch must be the pooled first-level chain with a window inside its %d centres", m_ctr);
        a.win_len = win_len; a.win_off = win_off;
        pa_sa_tiny_launch(a, RTv, ntiles, st);
    } else if (is_pooled && !split && tiny_on && pa_sa_tiny_applies(a, RTv)) {
        pa_sa_tiny_launch(a, RTv, ntiles, st);
    } else if (is_pooled && mid_on && pa_sa_mid_applies(a, RTv)) {
        if (pa_sa_mid_launch(a, RTv, ntiles, st) != PA_OK) return PA_EINVAL;
    } else if (is_pooled) {
        if (pa_chain_launch_pooled(a, RTv, split, wpw, ntiles, st) != 0) {
            pa_set_error("pa_mlp_chain: pooled tiling is built for nsample in (13..16], (17..20], (29..32]; got %d", ns);
            return PA_EUNSUPPORTED;
        }
    } else if (atomic_pool) {
        PA_REQUIRE(split && (RTv == 1 || RTv == 2), "pa_mlp_chain: the atomic pooled epilogue is built for the shared-tile tilings (rows=%ld)", total_rows);
        if (pa_fill32(out, 0u, (size_t)rows * ldo, st) != PA_OK) { pa_set_error("pa_mlp_chain: zero fill failed"); return PA_EINVAL; }      // a kernel, not a memset node (pa_common.h)
        pa_chain_launch_split_sa(a, RTv, true, ntiles, st);
    } else if (split) {
        if (mode == MODE_PLAIN) pa_chain_launch_split_plain(a, RTv, ntiles, st);
        else if (mode == MODE_SA) pa_chain_launch_split_sa(a, RTv, false, ntiles, st);
        else pa_chain_launch_split_fp(a, mode, RTv, ntiles, st);
    } else if (mode == MODE_FPX && rt1 && pa_fpx32_try(a, st)) {
        // the finest level's shape in half-K passes (fpx_f32.hip): same bits as the wave-private tile kernel below, three / four waves per SIMD
    } else if (mode == MODE_FPX) pa_chain_launch_wp_fpx(a, RTv, wpw, ntiles, st);
    else pa_chain_launch_wp_rows(a, mode, RTv, wpw, ntiles, st);
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
    PA_REQUIRE(nlayers >= 1 && nlayers <= 3 && wt && bias && kpad && nout && rows > 0 && g && idx3 && w3 && skip && wskip && bias0 && out,
               "pa_fp_chain_premul_f16: bad arguments");
    // the finest level's shape (xyz skip, two 256 -> 256 layers left): weights shared through LDS, activations in registers (fpx_f16.hip)
    const int took = pa_fpx16_try(nlayers, wp16, bias, kpad, nout, rows, g, idx3, w3, skip, n_unknown, m_known, c2, c1, wskip, bias0, out, ldo,
                                  g_chain_dbg, (hipStream_t)stream);
    if (took != 0) return took > 0 ? PA_OK : took;
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
