/* Private declarations of libpatchaugnet_hip.so -- NOT part of the drop-in boundary (include/patchaugnet_hip.h is).
 *
 * What lives here: test / profiling hooks of the library -- switches that force one of two shipping kernels for the same op (the tests A/B
 * them) and the cycle-stamp buffers of tools/chain_phases.py / tools/knn_phases.py.  Kernel variants that lost their A/B are deleted from the
 * sources (DESIGN.md appendix lists them with their numbers); there is no second library.
 * Environment knobs (PA_CHAIN_*, PA_TGEMM_*, PA_KNN_*, ...) are tuning aids of the A/B tooling under tools/; they are documented where they
 * are read (grep getenv csrc/) and are not an interface.
 */
#ifndef PA_INTERNAL_H
#define PA_INTERNAL_H
#include "../../include/patchaugnet_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Profiling hook: when set to a device buffer of 512 x 8 int64, the chain kernels store cycle-counter stamps at their phase
 * boundaries (tile start, prologue done, each layer done) for the first 512 tiles.  NULL (the default) turns it off. */
void pa_chain_debug_buffer(long long *buf);
void pa_knn_debug_buffer(long long *buf);   /* same for the pruned kNN kernel: 6 int64 (prologue cycles, query cycles, chunks visited, insertions, sort cycles, queries per wave) */

/* 1 = wherever the kernel's shape rules hold, 0 = never, -1 = the default rule / environment:
 *   pa_knn_quad_enable        pa_knnquery at 1024..4096 source points and >= 128 queries on the four-lanes-per-query cell-grid kernel (csrc/knn_quad.hip); 0 forces the
 *                             wave-per-query kernels
 *   pa_three_nn_grid_enable   pa_nearestneighbor / pa_three_nn_weights on the cell-grid kernel (csrc/three_nn_grid.hip); 0 forces the brute-force scan
 *   pa_chain_tiny_enable      the persistent first-set-abstraction kernel (csrc/sa_tiny.hip; since round 6 register-chained: equal to the generic
 *                             kernel up to the order of the fp32 additions, not bit for bit)
 *   pa_chain_fpx32_enable     the finest feature-propagation level in half-K passes, 9 KB of LDS per wavefront (csrc/fpx_f32.hip); 0 forces the
 *                             16-row wave-private tile kernel of pa_chain_kernel.h (same bits)
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
void pa_chain_fpx32_enable(int on);
void pa_linear_lds_enable(int on);
void pa_emd_persistent_enable(int on);
void pa_fpx16_enable(int mode);
void pa_tgemm_cm_enable(int on);

#ifdef __cplusplus
}
#endif
#endif /* PA_INTERNAL_H */
