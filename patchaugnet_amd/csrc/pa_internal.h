/* Private declarations of libpatchaugnet_hip.so -- NOT part of the drop-in boundary (include/patchaugnet_hip.h is).
 *
 * Two kinds of symbols live here:
 *   1. test / profiling hooks of the PRODUCT library: switches that force one of two shipping kernels for the same op (both bit-identical;
 *      the tests A/B them) and the cycle-stamp buffers of tools/chain_phases.py / tools/knn_phases.py;
 *   2. (under PA_EXPERIMENTAL) the entry points and switches of measured-slower kernel variants.  Those are compiled ONLY into the test-only
 *      library libpatchaugnet_hip_exp.so (csrc/Makefile: every source that mentions PA_EXPERIMENTAL is built a second time with
 *      -DPA_EXPERIMENTAL, fpx_reg.hip and knn_lane.hip only there); `nm -D libpatchaugnet_hip.so` shows none of them
 *      (tests/test_abi.py::test_product_library_has_no_experimental_symbols).
 * Environment knobs (PA_CHAIN_*, PA_TGEMM_*, PA_KNN_*, ...) are tuning aids of the A/B tooling under tools/; they are documented where they
 * are read (grep getenv csrc/) and are not an interface.
 */
#ifndef PA_INTERNAL_H
#define PA_INTERNAL_H
#include "../../include/patchaugnet_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- 1. test / profiling hooks of the product library ------------------------------------------------------------------------------- */
/* Profiling hook: when set to a device buffer of 512 x 8 int64, the chain kernels store cycle-counter stamps at their phase
 * boundaries (tile start, prologue done, each layer done) for the first 512 tiles.  NULL (the default) turns it off. */
void pa_chain_debug_buffer(long long *buf);
void pa_knn_debug_buffer(long long *buf);   /* same for the pruned kNN kernel: 6 int64 (prologue cycles, query cycles, chunks visited, insertions, sort cycles, queries per wave) */

/* 1 = wherever the kernel's shape rules hold, 0 = never, -1 = the default rule / environment:
 *   pa_knn_quad_enable        pa_knnquery at 2048..4096 source points on the four-lanes-per-query cell-grid kernel (csrc/knn_quad.hip); 0 forces the
 *                             wave-per-query kernels
 *   pa_three_nn_grid_enable   pa_nearestneighbor / pa_three_nn_weights on the cell-grid kernel (csrc/three_nn_grid.hip); 0 forces the brute-force scan
 *   pa_chain_tiny_enable      the persistent first-set-abstraction kernel (csrc/sa_tiny.hip)
 *   pa_chain_mid_enable       the LDS-resident second-set-abstraction kernel (csrc/sa_mid.hip; equal to the generic kernel up to the order of the
 *                             fp32 additions, not bit for bit)
 *   pa_linear_lds_enable      pa_linear at k = 256 on LDS-resident weights (csrc/linear_lds.hip)
 *   pa_emd_persistent_enable  pa_emd_forward as one persistent workgroup per cloud instead of one launch per round (0 / 1)
 *   pa_fpx16_enable           pa_fp_chain_premul_f16 at the finest level's shape on LDS-shared weights (csrc/fpx_f16.hip): 8 / 4 = waves per
 *                             workgroup, 0 = the wave-private LDS-tile kernel (fp16 path: both within fp16 rounding of the fp32 kernel, not
 *                             the same bits)
 *   pa_tgemm_cm_enable        pa_tgemm_nn's aligned shapes on LDS-resident weights (csrc/train_gemm_cm.hip): 1 = wherever the shape rules hold
 *                             (also below the default's minimum tile count), 0 = never (the LDS-tiled kernel)
 * Either setting of the index / gather / inference switches gives the same bits (tests/test_gpu_ops.py, test_gpu_chain.py, test_gpu_losses.py);
 * pa_chain_mid_enable, pa_fpx16_enable and pa_tgemm_cm_enable select kernels that agree to fp32 / fp16 rounding (a different summation
 * order of the statistics or the contraction), as their lines above say. */
void pa_knn_quad_enable(int on);
void pa_three_nn_grid_enable(int on);
void pa_chain_tiny_enable(int on);
void pa_chain_mid_enable(int on);
void pa_linear_lds_enable(int on);
void pa_emd_persistent_enable(int on);
void pa_fpx16_enable(int mode);
void pa_tgemm_cm_enable(int on);

/* ---- 2. measured-slower variants: libpatchaugnet_hip_exp.so only --------------------------------------------------------------------- */
#ifdef PA_EXPERIMENTAL
/* Same with a fused tail: the last layer (wt[nlayers - 1], zero bias, relu_last = 0) is the next finer level's pre-multiply applied to this
 * level's output; that output (the result of layer nlayers - 2) leaves through tap (ldtap), the pre-multiplied rows through out. */
int pa_fp_chain_premul_tap(int nlayers, const float *const *wt, const float *const *wpk, const float *const *bias, const int *kpad, const int *nout,
                           long rows, const float *g, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2,
                           int c1, const float *wskip, const float *wskip_p, const float *bias0, float *out, int ldo, float *tap, int ldtap,
                           int relu_last, pa_stream_t stream);

/* EXPERIMENTAL, not used by default.  Finest feature-propagation level with register-resident activations (fpx_reg.hip):
 * pa_fp_chain_premul for 1 <= c1 <= 4 and exactly
 * two remaining 256 -> 256 layers (patch_aug_net.py:350-362 at the 4096-point level), computed with operand-swapped MFMAs so that a
 * layer's accumulators are the next layer's B operand (no activation tile in LDS; weights through a shared LDS stage; 16-point waves).  g (b*m_known, 256);
 * wp2 / wp3: the two layers' K-major (256 x 256) weights in the k-permuted packing
 *     wp[((q*16 + ot)*64 + l)*4 + s] = Wt[16q + 4(l/16) + s][16 ot + l%16],  q, ot in 0..15, l in 0..63, s in 0..3.
 * Contracts the channels in a different order than pa_fp_chain_premul: same values to fp32 rounding, not the same bits. */
int pa_fpx256(long rows, const float *g, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c1,
              const float *wskip, const float *bias0, const float *wp2, const float *b2, const float *wp3, const float *b3,
              float *out, int ldo, pa_stream_t stream);

/* one lane per query over the cell grid (csrc/knn_lane.hip; 0.287 vs 0.149 ms at the model's size) */
void pa_knn_lane_enable(int on);
/* register-resident FPS without the LDS copy of the cloud (fps.hip; the round grows 0.70 -> 0.86 us) */
void pa_fps_reg_xyz_enable(int on);
/* pa_tgemm_nn's wave-private kernel (train_gemm.hip; a tie with the LDS-tiled kernel) */
void pa_tgemm_wave_enable(int on);
#endif /* PA_EXPERIMENTAL */

#ifdef __cplusplus
}
#endif
#endif /* PA_INTERNAL_H */
