/*
 * patchaugnet_hip.h -- C ABI of libpatchaugnet_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for PatchAugNet's descriptor-extraction hot path.  Plain
 * C: raw device pointers, ints, floats and an opaque stream handle; no torch,
 * no C++ types.  Two groups of entry points:
 *
 *  1. pa_* -- the boundary proper.  Every function enqueues work on `stream`
 *     (a hipStream_t; NULL = the null stream), never synchronises, never
 *     allocates, never retains a pointer, and returns 0 on success, a negative
 *     PA_E* code for bad arguments, or a positive hipError_t for a failed launch
 *     (pa_last_error() gives the text).  All tensors are contiguous; float = fp32,
 *     int = int32 unless stated.  Layouts are the reference's (citations are
 *     file:line under the reference tree, WHU-USI3DV/PatchAugNet).
 *
 *  2. the reference's own extern "C" launcher names (furthestsampling_cuda_launcher,
 *     knnquery_cuda_launcher, ...) with the reference's exact signatures, so
 *     the reference's torch binding layer (the *_cuda.cpp files under libs/pointops/src/<op>/)
 *     can link against this library unchanged.  They forward to group 1; where
 *     the original has no stream argument they use the null stream exactly as
 *     the original does.  "cuda" in these names is the
 *     reference's spelling of an import name, not a CUDA dependency.
 */
#ifndef PATCHAUGNET_HIP_H
#define PATCHAUGNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *pa_stream_t; /* hipStream_t */

#define PA_OK 0
#define PA_EINVAL (-1)      /* bad size / null pointer */
#define PA_EUNSUPPORTED (-2) /* shape outside what the kernel supports (documented per function) */

int pa_abi_version(void);
const char *pa_last_error(void); /* thread-local text of the last non-zero return */

/* ---- K1: furthest point sampling ------------------------------------------------------------
 * replaces furthestsampling_cuda_launcher  (libs/pointops/src/sampling/sampling_cuda_kernel.h:17,
 * kernel sampling_cuda_kernel.cu:59-168).  xyz (b,n,3); temp (b,n) caller-filled (1e10) and left
 * holding the final running minima, as the reference leaves it; idx (b,m) int32.  Bit-exact,
 * including the reference's tie-break order (larger min-distance, then lower bit-reversed
 * k mod block, then lower k). */
int pa_furthestsampling(int b, int n, int m, const float *xyz, float *temp, int *idx, pa_stream_t stream);

/* Engine form: running minima start at 1e10 and never leave registers (no temp tensor) and the sampled coordinates
 * new_xyz (b, m, 3) are written next to idx (fuses the gathering call of patch_aug_net.py:222-225).  n <= 8192. */
int pa_furthestsampling_gather(int b, int n, int m, const float *xyz, int *idx, float *new_xyz, pa_stream_t stream);
/* Samples [j_begin, j_end) of the same m-sample sequence (n <= 8192): the running minima live in temp (b, n) between launches -- the launch with
 * j_begin = 0 initialises them, every launch writes them back, a later one resumes from them and from idx[j_begin - 1].  Launches over ascending
 * ranges covering [0, m) on one stream produce pa_furthestsampling_gather's idx / new_xyz bit for bit (the reference's order is prefix-stable,
 * sampling_cuda_kernel.cu:59-168), so consumers of the first samples can start early (the engine's latency mode). */
int pa_furthestsampling_range(int b, int n, int m, int j_begin, int j_end, const float *xyz, float *temp, int *idx, float *new_xyz, pa_stream_t stream);

/* ---- K2/K3: gathering  (sampling_cuda_kernel.h:15-16, .cu:6-36) -----------------------------
 * forward: out[b,c,j] = points[b,c,idx[b,j]];  backward: grad_points[b,c,idx[b,j]] += grad_out[b,c,j]
 * (grad_points must be zeroed by the caller, libs/pointops/functions/pointops.py:52). */
int pa_gathering_forward(int b, int c, int n, int m, const float *points, const int *idx, float *out, pa_stream_t stream);
int pa_gathering_backward(int b, int c, int n, int m, const float *grad_out, const int *idx, float *grad_points, pa_stream_t stream);

/* ---- K4: kNN query  (knnquery_cuda_kernel.h:14, .cu:6-50) ------------------------------------
 * xyz (b,n,3), new_xyz (b,m,3) -> idx (b,m,nsample) int32, dist2 (b,m,nsample) fp32, ascending
 * (d2, index).  Slots that cannot be filled (nsample > n, non-finite distances) hold index 0 and
 * +inf like the reference.  Any nsample >= 1 is accepted (the reference silently overflows above
 * 200). */
int pa_knnquery(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx, float *dist2, pa_stream_t stream);
/* pa_knnquery for queries q0 .. q0 + mq - 1 of every cloud (buffers keep their m-query stride; same results for those rows).  PA_EUNSUPPORTED when
 * the level's shape (n, m, nsample) does not run the cell-grid kernel that takes windows: answer the whole level with pa_knnquery then. */
int pa_knnquery_window(int b, int n, int m, int nsample, int q0, int mq, const float *xyz, const float *new_xyz, int *idx, float *dist2, pa_stream_t stream);
/* The source cloud's counting sort as a launch of its own: cells = pa_cloud_cellsort_floats(b, n) floats (16-byte aligned), n <= 4096.  A caller
 * that knows the cloud before the queries (the engine: the input cloud vs the centres its sampling chain is still drawing) sorts early;
 * pa_knnquery_presorted then answers like pa_knnquery, bit for bit, without every workgroup repeating the sort.  PA_EUNSUPPORTED when the level's
 * shape does not run the cell-grid kernel (1024..4096 source points, >= 128 queries, nsample 16 / 20 / 32). */
long pa_cloud_cellsort_floats(int b, int n);
int pa_cloud_cellsort(int b, int n, const float *xyz, float *cells, pa_stream_t stream);
int pa_knnquery_presorted(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, const float *cells, int *idx, float *dist2, pa_stream_t stream);

/* ---- K5/K6/K8: grouping  (grouping_cuda_kernel.h:16-19, .cu:28-46, :60-74; grouping_int .cu:33-49)
 * forward: out[b,c,j,s] = points[b,c,idx[b,j,s]]
 * backward (here and for gathering / interpolation): ACCUMULATES into grad_points, which the caller zero-fills as the reference's
 * wrappers do; like the reference's atomicAdd, the order of the float additions is not fixed. */
int pa_grouping_forward(int b, int c, int n, int m, int nsample, const float *points, const int *idx, float *out, pa_stream_t stream);
int pa_grouping_backward(int b, int c, int n, int m, int nsample, const float *grad_out, const int *idx, float *grad_points, pa_stream_t stream);
int pa_grouping_int_forward(int b, int c, int n, int m, int nsample, const int64_t *points, const int *idx, int64_t *out, pa_stream_t stream);

/* ---- K9/K10/K11: three-NN + interpolation  (interpolation_cuda_kernel.h:18-23, .cu:90-114, :134-195)
 * nearestneighbor: unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) SQUARED distances, idx (b,n,3).
 * interpolation forward: out[b,c,j] = (w0*p[i0] + w1*p[i1]) + w2*p[i2], points (b,c,m), idx/weight (b,n,3). */
int pa_nearestneighbor(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx, pa_stream_t stream);
int pa_interpolation_forward(int b, int c, int m, int n, const float *points, const int *idx, const float *weight, float *out, pa_stream_t stream);
int pa_interpolation_backward(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight, float *grad_points, pa_stream_t stream);
/* The same gradient without atomics (csrc/gather.hip): the (point, neighbour) references are counting-sorted by target once per launch, then
 * every output element is a plain sum over its list.  Equal to pa_interpolation_backward up to the order of the float sums; n <= 4096,
 * m <= 8192; scratch: pa_interpolation_backward_scratch_ints(b, n, m) ints, 8-byte aligned.  grad_out_batch_stride: floats between consecutive
 * clouds of grad_out (0 = c n; larger for a channel slice of a wider contiguous tensor, read in place). */
long pa_interpolation_backward_scratch_ints(int b, int n, int m);
int pa_interpolation_backward_gather(int b, int c, int n, int m, const float *grad_out, long grad_out_batch_stride, const int *idx, const float *weight,
                                     float *grad_points, int *scratch, pa_stream_t stream);
/* The list inversion of pa_interpolation_backward_gather alone (it depends on idx / weight only): scratch then serves any number of
 * pa_interpolation_backward_gather calls with idx = weight = NULL for the same (b, n, m). */
int pa_interpolation_backward_lists(int b, int n, int m, const int *idx, const float *weight, int *scratch, pa_stream_t stream);

/* Training-mode feature propagation with the first 1x1 convolution folded through the interpolation (csrc/fp_fold_train.hip; the module of
 * patch_aug_net.py:350-362 in train() mode).  W [interp(F); S] = interp(W_a F) + W_b S:
 *   forward   Y1 (b,O,n) = interpolation(Z (b,O,m) = W_a F; idx, weight) + Wb (O x C1, row stride ldw) . S (b,C1,n), 0 <= C1 <= 8; stats:
 *             PA_BN_STAT_SLOTS x 2*O doubles receiving (accumulating) the per-channel sum and sum of squares of Y1, or NULL;
 *   backward  g (b,O,n) gradient of the layer's activation, yraw its raw output, p the layer's 7*O BatchNorm block after pa_bn_bwd_finalize,
 *             relu its activation mask, lists = pa_interpolation_backward_lists(idx, weight): writes G (b,O,m) = interpolation^T(dY1) and ADDS
 *             dY1 . S^T to dWb (O x C1, row stride ldw). */
int pa_fp_fold_forward(int b, int O, int m, int n, int C1, const float *Z, const int *idx, const float *weight, const float *S, const float *Wb, int ldw,
                       float *out, double *stats, pa_stream_t stream);
int pa_fp_fold_backward(int b, int O, int n, int m, int C1, const float *g, const float *yraw, const float *p, int relu, const float *S,
                        const int *lists, float *G, float *dWb, int ldw, pa_stream_t stream);

/* EdgeConv-style grouping of a set-abstraction level as one op each way (csrc/group_edge.hip; pointops.py:559-570 under autograd):
 *   forward   out (b,3+c,m,k): channels 0..2 = grouped_xyz (b,3,m,k) as given, channel 3 + i = features[b,i,idx[b,j,s]] - features[b,i,center_idx[b,j]];
 *   backward  dfeatures (b,c,n) WRITTEN = scatter of grad_out[:, 3:] over idx minus, at every centre, the sum over its k neighbours. */
int pa_group_edge_forward(int b, int c, int n, int m, int k, const float *features, const int *center_idx, const int *idx, const float *grouped_xyz,
                          float *out, pa_stream_t stream);
int pa_group_edge_backward(int b, int c, int n, int m, int k, const float *grad_out, const int *center_idx, const int *idx, float *dfeatures,
                           pa_stream_t stream);
/* The coordinate-only part of the same grouping (pointops.py:559-562) in one launch: o_grouped (b,3,m,k) = xyz[b,idx[b,j,s],:] (NULL = not wanted),
 * centred (b,3*reps,m,k) = that minus new_xyz[b,j,:], written reps (1|2) times along the channel axis -- reps = 2 is the first level's whole grouped
 * input, whose features are the coordinates.  xyz (b,n,3), new_xyz (b,m,3), idx (b,m,k). */
int pa_group_xyz(int b, int n, int m, int k, int reps, const float *xyz, const float *new_xyz, const int *idx, float *o_grouped, float *centred,
                 pa_stream_t stream);
/* out (b,count) = table (b,n) indexed by idx (b,count), int32: the level-local centre / neighbour indices mapped to input-cloud indices
 * (torch.gather in patch_aug_net.py:169-177). */
int pa_compose_indices(int b, int n, long count, const int *table, const int *idx, int *out, pa_stream_t stream);

/* K9 with the FP module's inverse-distance weights fused in (patch_aug_net.py:350-353): weight (b,n,3), idx (b,n,3). */
int pa_three_nn_weights(int b, int n, int m, const float *unknown, const float *known, float *weight, int *idx, pa_stream_t stream);

/* ---- K13: ball query  (ballquery_cuda_kernel.h:17, .cu:47-80); idx caller-zeroed --------------*/
int pa_ballquery(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx, pa_stream_t stream);

/* ---- K14/K15: featuredistribute / featuregather  (featuredistribute_cuda_kernel.h:15-17) ------*/
int pa_featuredistribute(int b, int n, int m, const float *max_xyz, const float *xyz, int *distribute_idx, pa_stream_t stream);
int pa_featuregather_forward(int b, int n, int m, int c, const float *max_feature, const int *distribute_idx, float *distribute_feature, pa_stream_t stream);
int pa_featuregather_backward(int b, int n, int m, int c, const float *grad_distribute_feature, const int *distribute_idx, float *grad_max_feature, pa_stream_t stream);

/* ---- K16: label statistics  (labelstat_cuda_kernel.h:20-27) -----------------------------------*/
int pa_labelstat_and_ballquery(int b, int n, int m, float radius, int nsample, int nclass, const float *new_xyz, const float *xyz,
                               const int *label_stat, int *idx, int *new_label_stat, pa_stream_t stream);
int pa_labelstat_ballrange(int b, int n, int m, float radius, int nclass, const float *new_xyz, const float *xyz,
                           const int *label_stat, int *new_label_stat, pa_stream_t stream);
int pa_labelstat_idx(int b, int n, int m, int nsample, int nclass, const int *label_stat, const int *idx, int *new_label_stat, pa_stream_t stream);

/* ---- Chamfer distance  (libs/chamfer_dist/chamfer_cuda.cpp:12-39, chamfer.cu:15-229) ----------
 * forward: xyz1 (B,n,3), xyz2 (B,m,3) -> dist1 (B,n), dist2 (B,m) squared NN distances, idx1, idx2.
 * backward: grad_xyz1/grad_xyz2 are zero-filled by the call, then accumulated (both directions). */
int pa_chamfer_forward(int B, int n, int m, const float *xyz1, const float *xyz2, float *dist1, float *dist2, int *idx1, int *idx2, pa_stream_t stream);
int pa_chamfer_backward(int B, int n, int m, const float *xyz1, const float *xyz2, const int *idx1, const int *idx2,
                        const float *grad_dist1, const float *grad_dist2, float *grad_xyz1, float *grad_xyz2, pa_stream_t stream);
/* ChamferDistanceL1 (libs/chamfer_dist/__init__.py:79-84) in one call each way: forward also writes loss[0] =
 * (mean(sqrt(dist1)) + mean(sqrt(dist2))) / 2 (partial: scratch of B * (ceil(n/256) + ceil(m/256)) doubles); backward takes the distances and the scalar gout[0] (device) instead of per-point gradients. */
int pa_chamfer_l1_forward(int B, int n, int m, const float *xyz1, const float *xyz2, float *dist1, float *dist2, int *idx1, int *idx2, float *loss,
                          double *partial, pa_stream_t stream);
int pa_chamfer_l1_backward(int B, int n, int m, const float *xyz1, const float *xyz2, const int *idx1, const int *idx2, const float *dist1,
                           const float *dist2, const float *gout, float *grad_xyz1, float *grad_xyz2, pa_stream_t stream);

/* ---- Earth mover's distance, auction algorithm  (libs/emd_module/emd.cpp:6-30, emd_cuda.cu:228-317) ------------
 * forward: xyz1, xyz2 (b,n,3) with n == m, n % 1024 == 0, b <= 512 (the reference's rules, emd_cuda.cu:236-249; other
 * shapes return PA_EUNSUPPORTED where the reference returns -1) and n <= 8192.  State tensors are the caller's, initialised
 * as libs/emd_module/emd_module.py:42-53 does: assignment, assignment_inv = -1; price, max_increments = 0; bid,
 * bid_increments, max_idx any.  Runs `iters` auction rounds (eps = minimum bid increment) and writes dist (b,n) =
 * |xyz1[j] - xyz2[assignment[j]]|^2.  The reference's thread-timing-dependent choices are fixed deterministically (lowest
 * object index among equal values; highest bidder index among matching increments), exactly as oracle_emd_forward.
 * The reference's unass_idx / unass_cnt / unass_cnt_sum / cnt_tmp scratch tensors are not needed (the list of unassigned
 * points lives in LDS).
 * backward: grad_xyz (b,n,3) += 2 grad_dist (xyz1 - xyz2[idx]); gradient w.r.t. xyz1 only, like the reference. */
int pa_emd_forward(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *assignment, float *price,
                   int *assignment_inv, int *bid, float *bid_increments, float *max_increments, int *max_idx, float eps, int iters,
                   pa_stream_t stream);
int pa_emd_backward(int b, int n, const float *xyz1, const float *xyz2, float *grad_xyz, const float *grad_dist, const int *idx,
                    pa_stream_t stream);

/* ---- Generic-dimension brute-force kNN  (libs/KNN_CUDA/knn_cuda/csrc/cuda/knn.cpp:23-56, knn.cu:232-269)
 * ref (dim,nr), query (dim,nq) -> dist (k,nq) fp32 L2 (sqrt applied), ind (k,nq) int64 1-BASED, order (dist asc, row asc). */
int pa_knn_generic(const float *ref, int nr, const float *query, int nq, int dim, int k, float *dist, int64_t *ind, pa_stream_t stream);
/* kNN over a per-query candidate list (the hard-negative refresh, datasets/scene_dataset.py:1101-1113): out (nq, k) int64 = for every query the
 * k entries of ITS list cand[q][0..L) (row indices into ref_rows, -1 = padding) nearest to it, nearest first, ties to the earlier list position
 * (what pa_knn_generic returns on the gathered rows, bit for bit); -1 where a list has fewer than k live entries.  Rows row-major; k <= 64. */
int pa_knn_candidates(const float *ref_rows, int dim, const float *q_rows, int nq, const int64_t *cand, int L, int k, int64_t *out, pa_stream_t stream);

/* ---- Fused shared-MLP chains (MFMA, fp32)  -- evaluation-time replacement of the unfused
 * grouping + subtract + cat (libs/pointops/functions/pointops.py:559-570), SharedMLP = Conv2d 1x1 + BatchNorm2d + ReLU
 * (utils/model_util/pt_util.py:16-41), max over the neighbourhood (place_recognition/patch_aug_net/models/patch_aug_net.py:236)
 * and interpolation + cat (patch_aug_net.py:354-359).  Activations are POINT-MAJOR (rows = points, channels contiguous).
 *   mode 0: rows of x (rows x k0, row stride ldx)
 *   mode 1: set-abstraction rows [xyz[nbr]-xyz[ctr] (3), feat[nbr]-feat[ctr] (c_feat)], rows = B*m_ctr groups of ns neighbours;
 *           pooled != 0 writes the max over each group's ns rows (out: groups x n_last), else all groups*ns rows
 *   mode 2: feature-propagation rows [(w0*f[i0] + w1*f[i1]) + w2*f[i2] (c2), skip (c1)], rows = B*n_unknown
 * wt[l]: device pointer, K-major (kpad[l] x nout[l]) weights with BatchNorm folded in, rows >= K zero; bias[l]: nout[l] floats.
 * wt / bias / kpad / nout themselves are HOST arrays of nlayers entries.  Every layer applies ReLU. */
int pa_mlp_chain(int mode, int pooled, int nlayers, const float *const *wt, const float *const *bias, const int *kpad, const int *nout,
                 long rows, int k0,
                 const float *x, int ldx,
                 const float *xyz, const float *feat, const int *center_idx, const int *nbr_idx, int n_src, int m_ctr, int ns, int c_feat,
                 const float *known, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                 float *out, int ldo, pa_stream_t stream);
/* Feature propagation with the first layer folded into the prologue (interpolation is linear):
 *   relu(W1 [interp(f); skip] + b1) = relu(interp(W1a f) + W1b skip + b1).
 * g (b*m_known, c2) = known features already multiplied by W1a (pa_linear, no bias, no ReLU); skip (b*n_unknown, c1), 1 <= c1 <= 4
 * (the xyz channels of the finest level, patch_aug_net.py:359); wskip (c1 x c2) K-major and bias0 (c2): W1b and b1 with
 * BatchNorm folded.  wt/wpk/bias/kpad/nout describe the REMAINING nlayers layers (input width c2).  Same result as mode 2 of
 * pa_mlp_chain up to fp32 summation order; (n_unknown/m_known) x fewer first-layer FLOPs. */
int pa_fp_chain_premul(int nlayers, const float *const *wt, const float *const *wpk, const float *const *bias, const int *kpad, const int *nout,
                       long rows, const float *g, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2,
                       int c1, const float *wskip, const float *wskip_p, const float *bias0, float *out, int ldo, pa_stream_t stream);
/*   c1 > 4 (coarser levels, skip = encoder features, c1 % 4 == 0, nlayers <= 2): the first layer stays in the chain as a c1-wide
 *   contraction over the skip channels (wskip, optional packed copy wskip_p) whose output gets the interpolated term added. */

/* ---- fp16-operand variants of the chain kernels (opt-in; BASELINE.json configs[4] "fp16 MFMA MLP path") ---------------------
 * Same fusion and fp32 inputs / outputs; inside the kernel activations are held as fp16 in LDS, weights are fp16 fragments
 * (pa_pack_weights_f16: wp16[(((ct*ceil(kpad/32) + ks)*64 + l)*8 + e] = wt[(32ks + 8(l/16) + e)*n + 16ct + l%16], zero past kpad;
 * pa_pack_weights_f16_halfs = number of fp16 elements) and every layer is v_mfma_f32_16x16x32_f16 with fp32 accumulation.
 * Hidden widths must be multiples of 32.  Results agree with the fp32 path to fp16 rounding (cosine >= 0.999 on descriptors). */
long pa_pack_weights_f16_halfs(int kpad, int n);
int pa_pack_weights_f16(int kpad, int n, const float *wt, void *wp16, pa_stream_t stream);
int pa_mlp_chain_f16(int mode, int pooled, int nlayers, const float *const *wt, const void *const *wp16, const float *const *bias,
                     const int *kpad, const int *nout, long rows, int k0,
                     const float *x, int ldx,
                     const float *xyz, const float *feat, const int *center_idx, const int *nbr_idx, int n_src, int m_ctr, int ns, int c_feat,
                     const float *known, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                     float *out, int ldo, pa_stream_t stream);
int pa_linear_f16(long rows, int k, int n, const float *x, int ldx, const float *wt, const void *wp16, const float *bias, int relu,
                  const float *residual, int ldr, float *out, int ldo, pa_stream_t stream);
int pa_fp_chain_premul_f16(int nlayers, const float *const *wt, const void *const *wp16, const float *const *bias, const int *kpad,
                           const int *nout, long rows, const float *g, const int *idx3, const float *w3, const float *skip, int n_unknown,
                           int m_known, int c2, int c1, const float *wskip, const void *wskip16, const float *bias0, float *out, int ldo,
                           pa_stream_t stream);   /* wskip16: pa_pack_weights_f16 of wskip, needed when c1 > 4 */

/* The finest level of the fp16 path with the pre-multiplied features as an fp16 table (csrc/fpx_f16.hip: weights shared through LDS by a
 * workgroup's waves, activations in registers; the interpolation gathers read 512 B instead of 1 KB per neighbour).
 *   pa_fp_premul_g16        g16[r][:] = fp16(x[r][:256] . W), W the (256 x 256) interpolated-part slice of the first layer, wp16 its
 *                           pa_pack_weights_f16 packing; x rows 16-byte aligned, ldx >= 256.
 *   pa_fp_chain_premul_g16  pa_fp_chain_premul_f16 with g16 in place of g.  Only c2 = 256, 1 <= c1 <= 4 and two remaining 256 -> 256 layers
 *                           (patch_aug_net.py:350-362 / pptnet.py FP level 0); PA_EUNSUPPORTED otherwise. */
int pa_fp_premul_g16(long rows, const float *x, int ldx, const void *wp16, void *g16, pa_stream_t stream);
int pa_fp_chain_premul_g16(int nlayers, const void *const *wp16, const float *const *bias, const int *kpad, const int *nout, long rows,
                           const void *g16, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                           const float *wskip, const float *bias0, float *out, int ldo, pa_stream_t stream);
/* pa_fp_chain_premul_g16 with the level's output in fp16: out16 (rows, 256) halfs, 16-byte aligned (for pa_netvlad_pyramid_f16h). */
int pa_fp_chain_premul_g16h(int nlayers, const void *const *wp16, const float *const *bias, const int *kpad, const int *nout, long rows,
                            const void *g16, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                            const float *wskip, const float *bias0, void *out16, pa_stream_t stream);

/* OPT-IN (model.mlp_dtype = "f32x3"; never the default path): pa_fp_chain_premul at the finest level's shape (c2 = 256, 1 <= c1 <= 4, two remaining
 * 256 -> 256 layers) with every product of the two dense layers evaluated from (hi, lo) fp16 operand pairs -- hi(a) hi(w) + lo(a) hi(w) +
 * hi(a) lo(w) on the fp16 MFMA with fp32 accumulation: ~2^-21 relative per product (fp32 MFMA: 2^-24; TF32: 2^-11) at 3/16 of the fp32 MFMA's
 * issue time (csrc/fpx_f32x3.hip).  g fp32 as for pa_fp_chain_premul.  wp16x3[l]: 131 072 halfs = pa_pack_weights_f16(256, 256) of
 * hi(W_l 2^s_l), then of lo(W_l 2^s_l), s_l a per-layer power-of-two scale that keeps lo(W) out of fp16's subnormal range; inv_scale[l] = 2^-s_l
 * (host array).  PA_EUNSUPPORTED for any other shape. */
int pa_linear_x3(long rows, const float *x, int ldx, const void *wp16x3, float inv_scale, float *out, int ldo, pa_stream_t stream);   /* the level's pre-multiply: out = x[:, :256] . W, one layer, same arithmetic */
int pa_fp_chain_premul_x3(int nlayers, const void *const *wp16x3, const float *inv_scale, const float *const *bias, long rows, const float *g,
                          const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1, const float *wskip,
                          const float *bias0, float *out, int ldo, pa_stream_t stream);

/* One dense layer on point-major rows (the chain kernel's plain mode with a selectable epilogue):
 * out[r][:] = residual[r][:] + act(x[r][:k] . Wt + bias), act = ReLU if relu != 0 else identity; residual may be NULL.
 * wt K-major (kpad x n), kpad = k rounded up to 4 with zero rows, n % 16 == 0; wpk: optional packed copy (below) or NULL. */
int pa_linear(long rows, int k, int n, const float *x, int ldx, const float *wt, const float *wpk, const float *bias, int relu,
              const float *residual, int ldr, float *out, int ldo, pa_stream_t stream);

/* Fragment-major packing of a K-major (kpad x n) weight matrix, n % 64 == 0:
 *   wp[((cg * (kpad/4) + ks) * 64 + l) * 4 + j] = wt[(4 ks + l/16) * n + 64 cg + 16 j + l%16]
 * i.e. the four v_mfma_f32_16x16x4_f32 B fragments a lane needs for a 64-column group sit in one 16-byte word, so the chain
 * kernels fetch weights with 4 instead of 16 loads per k-step.  pa_mlp_chain_packed = pa_mlp_chain with one packed copy per
 * layer (wpk[l] may be NULL: that layer then reads wt[l]); pa_linear takes the packed copy as `wpk` (or NULL). */
int pa_pack_weights(int kpad, int n, const float *wt, float *wp, pa_stream_t stream);
int pa_mlp_chain_packed(int mode, int pooled, int nlayers, const float *const *wt, const float *const *wpk, const float *const *bias,
                        const int *kpad, const int *nout, long rows, int k0,
                        const float *x, int ldx,
                        const float *xyz, const float *feat, const int *center_idx, const int *nbr_idx, int n_src, int m_ctr, int ns, int c_feat,
                        const float *known, const int *idx3, const float *w3, const float *skip, int n_unknown, int m_known, int c2, int c1,
                        float *out, int ldo, pa_stream_t stream);
/* The NEXT pa_mlp_chain* call of this thread (mode 1, pooled, the persistent first-level kernel's shape) computes only centres win_off ..
 * win_off + win_len - 1 of every cloud: rows = clouds * win_len; centre / neighbour / output buffers keep their m_ctr-per-cloud layout.  The window
 * is consumed by that call.  (First level of the engine's latency mode: its chain runs on the samples of a finished sampling chunk.) */
int pa_sa_group_window(int win_len, int win_off);

/* ---- PPT-Net grouped self-attention core  (place_recognition/pptnet_origin/models/pptnet.py:261-282; twin GroupSALayer,
 * place_recognition/patch_aug_net/models/loupe.py:69-114).  yv (b, n, 2c) point-major = [Y | V] with Y = q_conv(x) = k_conv(x)
 * (tied grouped conv expanded to a dense block-diagonal weight) and V = v_conv(x) + bias; x (b, n, c).
 * Writes d (b, n, c) = x - x_r, x_r = V^T attn, attn = row soft-max of Y Y^T re-normalised by (1e-9 + column sums).
 * stats: scratch of 2*b*n floats (row max and 1/row-sum).  Any n >= 1; c in {64, 128, 256, 512}. */
int pa_sa_attention(int b, int n, int c, const float *yv, const float *x, float *stats, float *d, pa_stream_t stream);
/* The same with the layer behind the attention in the second pass's epilogue: out (b, n, c) = x + relu(W (x - x_r) + bias), wt K-major (c x c) fp32 =
 * trans_conv with after_norm folded (pptnet.py:279-281), bt (c): one launch and one (rows x c) round trip less per level. */
int pa_sa_attention_trans(int b, int n, int c, const float *yv, const float *x, float *stats, const float *wt, const float *bt, float *out, pa_stream_t stream);
/* The same op with both contractions on fp16 MFMA (fp32 accumulation, fp32 soft-max statistics; csrc/attention_f16.hip) -- the fp16 path of
 * BASELINE.json configs[4], under its cosine >= 0.999 contract; c in {64, 128, 256}.  split != 0 carries the energy operands as (hi, lo) fp16
 * pairs (three products: ~21 bits of the logits).  scratch: pa_sa_attention_f16_scratch_halfs(b, n, c, split) fp16 elements, 16-byte aligned. */
long pa_sa_attention_f16_scratch_halfs(int b, int n, int c, int split);
int pa_sa_attention_f16(int b, int n, int c, int split, const float *yv, const float *x, void *scratch, float *stats, float *d, pa_stream_t stream);
/* ... and with the layer behind the attention in the same launch: out (b, n, c) = x + relu(W (x - x_r) + bias), W = trans_conv with after_norm folded
 * (pptnet.py:279-281).  wtp: pa_sa_attention_f16_pack_trans of the K-major (c x c) fp32 weight (c * c fp16 elements); bt (c) fp32. */
int pa_sa_attention_f16_pack_trans(int c, const float *wt, void *wtp, pa_stream_t stream);
int pa_sa_attention_trans_f16(int b, int n, int c, int split, const float *yv, const float *x, void *scratch, float *stats, const void *wtp, const float *bt,
                              float *out, pa_stream_t stream);


/* out[g][c] = max over s < ns of in[g*ns + s][c]  (rows of c floats) */
int pa_rowgroup_max(long groups, int ns, int c, const float *in, float *out, pa_stream_t stream);

/* ---- NetVLAD pyramid + adaptive feature aggregator (MFMA, fp32; evaluation mode) ------------------------------
 * pa_netvlad: one pyramid scale of place_recognition/patch_aug_net/models/loupe.py:191-222.  x (b, n, 256) POINT-MAJOR;
 *   wc_t: cluster_weights with BatchNorm1d folded in, K-major (256 x kp), kp = k rounded up to 16, padding columns zero;
 *   bias: folded BatchNorm shift (kp);  w2: cluster_weights2[0] (256 x k);  scratch: pa_netvlad_scratch_floats(b, n, k) floats;
 *   writes the intra-normalised VLAD block to out[b][c][koff + j], j < k, with ldo floats per (b, c) row, i.e. straight into the
 *   concatenated (b, 256, sum k) tensor of loupe.py:301-303.
 * pa_afa: MLPAttentionLayer + AdaptiveFeatureAggregator (loupe.py:24-41, :57-66).  v (b, 256, ktot); watt (256, 256) row-major
 *   (out, in) of the single attention conv; fc_wt K-major (256*ktot, nout); fc_bias (nout); scale/shift: BatchNorm1d (eval) folded
 *   to y*scale + shift; l2norm != 0 applies F.normalize; scratch: pa_afa_scratch_floats(b, 256, ktot, nout) floats; desc (b, nout). */
long pa_netvlad_scratch_floats(int b, int n, int k);
int pa_netvlad(int b, int n, int c, int k, const float *x, const float *wc_t, const float *bias, const float *w2, float *scratch,
               float *out, int ldo, int koff, pa_stream_t stream);
/* Cluster-major variants used by the fused engine: pa_netvlad_rows writes out[b][koff + j][c] ((b, ktot, 256): one contiguous
 * 1 KB row per cluster); pa_afa_rows consumes that layout: the attention logits become one dense-layer launch on b*ktot rows
 * (watt_t = the attention conv as a K-major (in, out) matrix, watt_p its packed copy or NULL, zero_bias = 256 zeros), and the FC
 * weight must have its ROWS ORDERED k*256 + c (the reference's nn.Linear weight is c*ktot + k: permute once on the host).
 * scratch: pa_afa_rows_scratch_floats(b, 256, ktot, nout) floats. */
/* wc_p of pa_netvlad_rows: NULL, or (k > 48 only) the fragment-ordered copy of wc_t (256, 64) that pa_netvlad_pack_weights writes
 * (256 * 64 floats; the assignment GEMM contracts channels in a permuted order that lets it read its LDS tile 16 bytes at a time). */
int pa_netvlad_pack_weights(int c, int kp, const float *wc_t, float *wc_p, pa_stream_t stream);
int pa_netvlad_rows(int b, int n, int c, int k, const float *x, const float *wc_t, const float *wc_p, const float *bias, const float *w2,
                    float *scratch, float *out, int ktot, int koff, pa_stream_t stream);   /* wc_p: pa_pack_weights(256, 64, wc_t) when k > 48, else NULL */
long pa_afa_rows_scratch_floats(int b, int c, int ktot, int nout);
int pa_afa_rows(int b, int c, int ktot, int nout, const float *vt, const float *watt_t, const float *watt_p, const float *zero_bias,
                const float *fc_wt, const float *fc_bias, const float *scale, const float *shift, int l2norm, float *scratch,
                float *desc, pa_stream_t stream);
/* The whole pyramid in one call (what SpatialPyramidNetVLAD.forward does scale by scale, loupe.py:284-303): arrays of nscales (<= 4)
 * entries, coarse to fine; x[s] (b, n[s], 256) point-major; wc_t / wc_p / bias / w2 / scratch per scale as for pa_netvlad_rows
 * (wc_p may be NULL, wc_p[s] is used only where k[s] > 48).  out (b, ktot, 256) cluster-major rows, ktot = sum k[s].  One accumulate
 * launch per scale and ONE finalize launch for all of them; bit-identical to per-scale pa_netvlad_rows.
 * phases: bit 0 = the launches of the <= 16-cluster scales, bit 1 = the other scales' launches, bit 2 = finalize (7 = all); x[s] may be
 * NULL for scales the requested phases do not read, so the coarse scales can be issued as soon as their feature maps exist.
 * pa_afa_fused: the APFA head (loupe.py:24-41, :57-66) on those rows in two launches.  Because 1 + w[k] > 0, relu(x + x*w[k]) =
 * (1 + w[k]) relu(x): the per-cluster partial products relu(v[k]) . Wfc[k] are computed once, beside the attention logits, and the
 * soft-max over clusters only scales them.  Arguments as pa_afa_rows (no packed / zero-bias operands); nout % 64 == 0;
 * scratch: pa_afa_fused_scratch_floats(b, ktot, nout) floats.  Same function as pa_afa_rows up to fp32 re-association of the FC sum. */
int pa_netvlad_pyramid(int b, int nscales, const int *n, const int *k, const float *const *x, const float *const *wc_t, const float *const *wc_p,
                       const float *const *bias, const float *const *w2, float *const *scratch, float *out, int phases, pa_stream_t stream);
/* The same for the model's fp16 path: at the scales with 49..64 clusters (wc16[s] = 32 768 halfs: pa_pack_weights_f16(256, 64) of wc_t[s],
 * then of wc_t[s] - fp16(wc_t[s]); NULL elsewhere) the assignment GEMM and the aggregation run on the 16-bit MFMAs with fp32 accumulation.  The
 * soft-max logits keep fp32 accuracy ((hi, lo) fp16 operand pairs, three products); the aggregation takes features and soft-assignments
 * rounded to bf16 (fp32's exponent range: the masses of clusters nobody is assigned to are below fp16's); logits bias, soft-max, a_sum, residual, normalisation in fp32.  Same scratch layout and finalize as pa_netvlad_pyramid. */
int pa_netvlad_pyramid_f16(int b, int nscales, const int *n, const int *k, const float *const *x, const float *const *wc_t, const float *const *wc_p,
                           const void *const *wc16, const float *const *bias, const float *const *w2, float *const *scratch, float *out, int phases,
                           pa_stream_t stream);
/* The same where the feature maps of the scales flagged in x16_mask (bit s) arrive as fp16 rows of 256 halfs (x[s] points at halfs; only scales
 * that run the fp16 kernel).  Producer: pa_fp_chain_premul_g16h.  For descriptor-only extraction on the fp16 path: the finest map is written
 * and read once in half the bytes. */
int pa_netvlad_pyramid_f16h(int b, int nscales, const int *n, const int *k, const float *const *x, const float *const *wc_t, const float *const *wc_p,
                            const void *const *wc16, const float *const *bias, const float *const *w2, float *const *scratch, float *out, int phases,
                            int x16_mask, pa_stream_t stream);
long pa_afa_fused_scratch_floats(int b, int ktot, int nout);
int pa_afa_fused(int b, int c, int ktot, int nout, const float *vt, const float *watt_t, const float *fc_wt, const float *fc_bias,
                 const float *scale, const float *shift, int l2norm, float *scratch, float *desc, pa_stream_t stream);
long pa_afa_scratch_floats(int b, int c, int ktot, int nout);
int pa_afa(int b, int c, int ktot, int nout, const float *v, const float *watt, const float *fc_wt, const float *fc_bias,
           const float *scale, const float *shift, int l2norm, float *scratch, float *desc, pa_stream_t stream);

/* Split-K fully connected layer with folded BatchNorm1d, for the aggregation heads
 * (PPT-Net: hidden_weights + bn2, then GatingContext -- place_recognition/pptnet_origin/models/loupe.py:98-105, :108-137):
 * out (b, nout) = g(BN(y (b, kdim) . fc_wt + fc_bias)), fc_wt K-major (kdim x nout); g = identity, or gate_x * sigmoid(.) when
 * gate_x (b, nout) is given; l2norm != 0 applies F.normalize last.  fc_bias and gate_x may be NULL.
 * scratch: pa_fc_scratch_floats(b, kdim, nout) floats. */
long pa_fc_scratch_floats(int b, int kdim, int nout);
int pa_fc(int b, int kdim, int nout, const float *y, const float *fc_wt, const float *fc_bias, const float *scale, const float *shift,
          int l2norm, const float *gate_x, float *scratch, float *out, pa_stream_t stream);

/* Max-pool head (SpatialPyramidNetVLAD aggregation_type 3, place_recognition/patch_aug_net/models/loupe.py:304-306):
 * out[b][c] = max over the ktot clusters of vt[b][k][c] (cluster-major rows from pa_netvlad_rows), then F.normalize when l2norm != 0.
 * c must be 256. */
int pa_vlad_maxpool(int b, int ktot, int c, const float *vt, int l2norm, float *out, pa_stream_t stream);

/* ---- Training-mode dense path (MFMA, fp32): building blocks of the hand-written forward / backward of SharedMLP in train() mode
 * (utils/model_util/pt_util.py:16-41, :98-152: conv 1x1 -> BatchNorm with batch statistics -> ReLU) and of PointNetDecoder
 * (place_recognition/patch_aug_net/models/pointnet_autoencoder.py:85-111), as the training step drives them
 * (place_recognition/train_place_recognition.py:142-169, :386-392).  Activations are CHANNEL-MAJOR (B, C, P) like the reference's.
 * Per-channel parameter block p: 7 rows of nch floats -- 0 scale = gamma*rstd, 1 shift = beta - mean*scale, 2 mean, 3 rstd (pa_bn_finalize),
 * 4 mean(mask g), 5 mean(mask g * xhat), 6 gamma*rstd (pa_bn_bwd_finalize).
 * per_batch_stats != 0 (PointNetDecoder: every related cloud is its own BatchNorm batch, patch_aug_net.py:83-98): one statistics /
 * parameter block PER BATCH ENTRY, laid out back to back (stats: batch x PA_BN_STAT_SLOTS x 2*nch, p: batch x 7*nch, sums: batch x 2*nch);
 * pa_bn_finalize / pa_bn_bwd_finalize then take groups = batch, update the running statistics group after group and add the groups'
 * parameter gradients up.
 * pa_tgemm_nn: C_b (M x N) = [beta*C_b +] act(A_b (M x K) . f(B_b) (K x N) + bias[m]); B, C n-contiguous; A(m,k) = A[m*lda + k] when
 *   a_kcontig else A[k*lda + m]; sAb = 0 shares A over the batch.  f acts per k (the channel): bmode 0 identity, 1 relu(x*p0 + p1),
 *   2 / 3 the BatchNorm(+ReLU mask for 2) input gradient built from B = gradient w.r.t. the activation and baux = raw layer output.
 *   act 0 none / 1 tanh / 2 squared distance C = max(bias[m] + colv[n] - 2 A.B, 0) (retrieval, pa_knn_mfma_select).  stats: PA_BN_STAT_SLOTS replicas of 2*M doubles receiving (accumulating) the per-row sum and sum of squares
 *   of the stored values (a workgroup adds to one replica; pa_bn_finalize sums them), or NULL.
 * pa_tgemm_kk: C (M x N) += sum over batch and k of fA(A_b)(m,k) * fB(B_b)(n,k), both operands k-contiguous (weight gradients: k runs
 *   over the points); amode 0 / 2 / 3 per row m, bmode 0 / 1 per row n; partial tiles are combined with fp32 atomics, so C must be
 *   zero-filled (or hold the value to add to); per_batch != 0 accumulates into C_b = C + b*sCb instead of summing over the batch. */
#define PA_BN_STAT_SLOTS 32
int pa_tgemm_nn(int batch, int M, int N, int K, const float *A, long sAb, int lda, int a_kcontig,
                const float *B, long sBb, int ldb, int bmode, const float *baux, const float *bp,
                float *C, long sCb, int ldc, int beta, const float *bias, const float *colv, int act, double *stats, int per_batch_stats, pa_stream_t stream);
int pa_tgemm_kk(int batch, int M, int N, long K, const float *A, long sAb, int lda, int amode, const float *aaux, const float *ap,
                const float *B, long sBb, int ldb, int bmode, const float *bp,
                float *C, long sCb, int ldc, int per_batch, int per_batch_stats, pa_stream_t stream);
/* pa_tgemm_kk (shared C) through `reps` zero-filled replicas of the output (scratch: reps x M x N floats) and a second launch that adds them to
 * C: for small outputs contracted over very long k, where the split-K partial tiles otherwise queue their atomics on a few hundred addresses. */
int pa_tgemm_kk_rep(int batch, int M, int N, long K, const float *A, long sAb, int lda, int amode, const float *aaux, const float *ap,
                    const float *B, long sBb, int ldb, int bmode, const float *bp, float *C, int ldc, float *scratch, int reps, pa_stream_t stream);

/* Adam over a list of fp32 tensors (csrc/adam.hip; torch.optim.Adam's arithmetic: amsgrad / maximize off, weight_decay as L2): pa_adam_tick
 * advances the device step counter (step[0] += 1), pa_adam_step updates ntensors contiguous tensors -- HOST arrays of device pointers
 * (parameter, gradient, exp_avg, exp_avg_sq) and element counts -- in ceil(ntensors / 84) launches.  step and lr are DEVICE scalars: no host
 * value that changes from step to step enters the arithmetic, so a captured hipGraph replays it and a learning-rate schedule
 * (train_place_recognition.py:531-568) reaches the graph by rewriting lr[0]. */
int pa_adam_tick(float *step, pa_stream_t stream);
int pa_adam_step(int ntensors, float *const *p, const float *const *g, float *const *m, float *const *v, const long *numel, const float *step,
                 const float *lr, float beta1, float beta2, float eps, float weight_decay, pa_stream_t stream);
/* BatchNorm (training): statistics -> rows 0..3 of p, running statistics updated in place (momentum, unbiased variance) when given;
 * *num_batches_tracked += groups when given (torch.nn.BatchNorm's int64 counter, one forward pass per statistics group). */
int pa_bn_finalize(int nch, int groups, double count, const double *stats, const float *gamma, const float *beta, float eps, float momentum,
                   float *running_mean, float *running_var, float *p, long long *num_batches_tracked, pa_stream_t stream);
/* BatchNorm (evaluation): the parameter block from the RUNNING statistics (rows 0..3; rows 4, 5 zero, row 6 = scale), `groups` identical copies --
 * the module path in eval() mode runs on the same GEMM kernels, forward and backward (the input gradient then has no batch-statistics terms:
 * pa_bn_bwd_finalize with count = +inf leaves rows 4, 5 at zero and still returns dgamma / dbeta). */
int pa_bn_eval_params(int nch, int groups, const float *gamma, const float *beta, const float *running_mean, const float *running_var, float eps,
                      float *p, pa_stream_t stream);
/* sums (2*C doubles, zero-filled) += per-channel sum of mask(g) and of mask(g)*xhat over g, y (B, C, P); relu != 0: mask = BN(y) > 0. */
int pa_bn_bwd_reduce(int B, int C, long P, const float *g, const float *y, const float *p, int relu, double *sums, int per_batch_stats, pa_stream_t stream);
/* pa_tgemm_nn for an input-gradient contraction (bmode 2 / 3, A shared by the batch, C contiguous (batch, M, N)) fused with the NEXT (earlier)
 * layer's pa_bn_bwd_reduce(batch, M, N, C, ynext, pnext, relu_next, sums_next): where the LDS-resident-weights kernel takes the shape the
 * sums ride on the contraction's epilogue, otherwise the two launches run one after the other.  Same results either way (fp32 summation order). */
int pa_tgemm_nn_bnred(int batch, int M, int N, int K, const float *A, int lda, int a_kcontig, const float *B, long sBb, int ldb, int bmode,
                      const float *baux, const float *bp, float *C, long sCb, int ldc, const float *ynext, const float *pnext, int relu_next,
                      double *sums_next, pa_stream_t stream);
/* rows 4..6 of p from the sums; dgamma / dbeta (nch floats) written when given. */
int pa_bn_bwd_finalize(int nch, int groups, double count, const double *sums, float *p, float *dgamma, float *dbeta, pa_stream_t stream);
/* out = [relu](y*scale + shift) over (B, C, P); pool > 0: max over groups of `pool` consecutive points (patch_aug_net.py:236) ->
 * out (B, C, P/pool) and arg (int8 winning slot, first maximum).  pa_maxpool_bwd scatters a pooled gradient back: rows = B*C. */
int pa_bn_apply(int B, int C, long P, int pool, int relu, const float *y, const float *p, float *out, signed char *arg, int per_batch_stats, pa_stream_t stream);
int pa_maxpool_bwd(int rows, long Pout, int pool, const float *gp, const signed char *arg, float *g, pa_stream_t stream);
/* pa_maxpool_bwd plus the pooled layer's pa_bn_bwd_reduce in one launch: the BatchNorm-backward sums (2 x C doubles, accumulated) come from the
 * Pout pooled gradients and the raw outputs y (B, C, Pout * pool) at their arg-max positions -- the dense (gradient, output) pair is not read. */
int pa_maxpool_bwd_bnred(int B, int C, long Pout, int pool, const float *gp, const signed char *arg, float *g, const float *y, const float *p, int relu,
                         double *sums, pa_stream_t stream);

/* ---- Descriptor losses of the training step in one launch (csrc/losses.hip; losses/pointnetvlad_loss.py:18-45, :53-105): value and gradient.
 * desc (b, 1 + p + nn + 1, d): per tuple the query, p positives, nn negatives, the other negative.  quad != 0: quadruplet_loss, else
 * triplet_loss (m2 and the last row unused).  loss: 1 float; grad: same shape as desc, the gradient of the value.  b, p, nn <= 64. */
int pa_quadruplet_loss(int b, int p, int nn, int d, const float *desc, float m1, float m2, int use_min, int lazy, int ignore_zero, int quad,
                       float *loss, float *grad, pa_stream_t stream);

/* ---- Grouped self-attention in training / autograd mode (csrc/attention_train.hip; pptnet.py:261-282): the part between the two GEMMs.
 * pa_attn_softmax_renorm: energy (b, n, n) -> A = softmax_rows(energy) / (1e-9 + column sums) IN PLACE; colsum (b, n) receives the
 * denominators.  pa_attn_softmax_renorm_backward: grad (b, n, n) holds dL/dA on entry and dL/dEnergy on return.
 * scratch: pa_attn_train_scratch_floats(b, n) floats for the forward call, that + b*n for the backward call.  Deterministic. */
long pa_attn_train_scratch_floats(int b, int n);
int pa_attn_softmax_renorm(int b, int n, float *energy, float *colsum, float *scratch, pa_stream_t stream);
int pa_attn_softmax_renorm_backward(int b, int n, const float *attn, const float *colsum, float *grad, float *scratch, pa_stream_t stream);

/* ---- The aggregation heads' small non-GEMM steps in training / autograd mode, one launch each way (csrc/train_glue.hip); all deterministic.
 * NetVLAD (patch_aug_net/models/loupe.py:196-222): pa_softmax_cols: act (b, k, n) = softmax over the k clusters of every point of `in`;
 *   part (b, ceil(n / 256), k) = per-256-point partial sums of act over the points.  _backward: dpre = act (g - sum_k g act), g = dact + dasum[b][k]
 *   (dasum NULL = no gradient reached the cluster sums).
 * pa_vlad_residual_normalize: out (b, c, k) = v / max(||v||_c, 1e-12), v = raw - a_sum cw2 (cw2 (c, k) = cluster_weights2), a_sum (b, k) = the nblk
 *   partials of `part` added up; a_sum and nrm (b, k: the norms before the clamp) are outputs kept for _backward, which returns dv (= draw), dasum (b, k)
 *   and dcw2 (c, k; NULL = not wanted).
 * pa_l2_normalize: torch.nn.functional.normalize(x, dim = 1) of x (b, c, m) (m = 1: the rows of a matrix); nrm (b, m) = the norms before the clamp.
 * pa_bn_rows_train: torch.nn.BatchNorm1d in train mode over the rows of x (r, f): batch statistics, running statistics updated with the unbiased
 *   variance and *num_batches_tracked += 1 when given; mean / rstd (f) kept for pa_bn_rows_backward (dx, dgamma, dbeta; the last two may be NULL).
 * pa_afa_attention (patch_aug_net/models/loupe.py:8-41): w (b, k) = softmax over k of max over c of r (b, c, k); out = relu(x + x w); arg (b, k) = the
 *   channel of each maximum.  _backward: dx, and dr (the soft-max gradient at the arg channels, 0 elsewhere; NULL = not wanted). */
int pa_softmax_cols(int b, int k, int n, const float *in, float *act, float *part, pa_stream_t stream);
int pa_softmax_cols_backward(int b, int k, int n, const float *act, const float *dact, const float *dasum, float *dpre, pa_stream_t stream);
int pa_vlad_residual_normalize(int b, int c, int k, int nblk, const float *raw, const float *part, const float *cw2, float *out, float *asum, float *nrm,
                               pa_stream_t stream);
int pa_vlad_residual_normalize_backward(int b, int c, int k, const float *dout, const float *out, const float *nrm, const float *asum, const float *cw2, float *dv,
                                        float *dasum, float *dcw2, pa_stream_t stream);
int pa_l2_normalize(int b, int c, int m, const float *x, float *out, float *nrm, pa_stream_t stream);
int pa_l2_normalize_backward(int b, int c, int m, const float *dout, const float *out, const float *nrm, float *dx, pa_stream_t stream);
int pa_bn_rows_train(int r, int f, const float *x, const float *gamma, const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                     long long *num_batches_tracked, float *out, float *mean, float *rstd, pa_stream_t stream);
int pa_bn_rows_backward(int r, int f, const float *dy, const float *x, const float *mean, const float *rstd, const float *gamma, float *dx, float *dgamma,
                        float *dbeta, pa_stream_t stream);
int pa_afa_attention(int b, int c, int k, const float *x, const float *r, float *out, float *w, int *arg, pa_stream_t stream);
int pa_afa_attention_backward(int b, int c, int k, const float *dout, const float *x, const float *w, const int *arg, float *dx, float *dr, pa_stream_t stream);

/* ---- Patch overlap-pair selection of the training step's contrastive patch-feature term (the Python loops of train_one_epoch,
 * place_recognition/train_place_recognition.py:308-372), csrc/patch_pairs.hip.  Records in CSR form: idx1 (nrec), near_off / far_off
 * (nrec + 1), near / far = original point indices; center_m / center_n = the m0 FPS centre indices (original indices < npoints) of the
 * query / positive cloud.  Pass 1 fills scratch_inv (2*npoints ints) and counts (nrec: triplets per record, 0 = record dropped); pass 2
 * takes offsets = exclusive prefix sum of counts and writes the triplets (positions in the centre lists: query in m, positive / negative
 * in n; positives ascending per record like np.where(np.isin(...)), negatives uniform with replacement from a counter hash of seed). */
int pa_patch_pairs_count(int nrec, const int *idx1, const int *near_off, const int *near_v, const int *far_off, const int *far_v, int npoints, int m0,
                         const int *center_m, const int *center_n, int *scratch_inv, int *counts, pa_stream_t stream);
int pa_patch_pairs_fill(int nrec, const int *idx1, const int *near_off, const int *near_v, const int *far_off, const int *far_v, int npoints, int m0,
                        const int *scratch_inv, unsigned long long seed, const int *offsets, int *out_idx1, int *out_pos2, int *out_neg2, pa_stream_t stream);

/* ---- Retrieval kNN at database scale (csrc/knn_mfma.hip): the recall harness' brute-force search (datasets/scene_dataset.py:1016-1099,
 * KNN_CUDA knn.cu:232-269) with the distance matrix on MFMA and an exact re-rank: columns equal pa_knn_generic's bit for bit.
 * a (nq_blk x lda) = pa_tgemm_nn(act = 2) output for a block of queries: a[q][r] = max(|q|^2 + |r|^2 - 2 q.r, 0); ref_rows (nr, dim),
 * query_rows (nq_blk, dim) row-major; qnorm (nq_blk); rnorm_max: device scalar max |r|^2.  dist / ind: (k, nq_total) KNN_CUDA layout,
 * 1-based int64 indices, columns q0.. written; flags[q0 + q] = 1: candidate overflow, column NOT written -- rerun through pa_knn_generic. */
int pa_knn_mfma_select(const float *a, long lda, int nq_blk, int nr, int dim, int k, const float *ref_rows, const float *query_rows,
                       const float *qnorm, const float *rnorm_max, int q0, int nq_total, float *dist, int64_t *ind, int *flags, pa_stream_t stream);

/* ---- the reference's launcher names (group 2) -------------------------------------------------*/
void furthestsampling_cuda_launcher(int b, int n, int m, const float *dataset, float *temp, int *idxs);
void gathering_forward_cuda_launcher(int b, int c, int n, int m, const float *points, const int *idx, float *out);
void gathering_backward_cuda_launcher(int b, int c, int n, int m, const float *grad_out, const int *idx, float *grad_points);
void knnquery_cuda_launcher(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx, float *dist2, pa_stream_t stream);
void grouping_forward_cuda_launcher(int b, int c, int n, int m, int nsample, const float *points, const int *idx, float *out);
void grouping_forward_cuda_launcher_fast(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out);
void grouping_backward_cuda_launcher(int b, int c, int n, int m, int nsample, const float *grad_out, const int *idx, float *grad_points);
void grouping_int_forward_cuda_launcher(int b, int c, int n, int m, int nsample, const long int *points, const int *idx, long int *out);
void grouping_int_forward_cuda_launcher_fast(int b, int c, int n, int npoints, int nsample, const long int *points, const int *idx, long int *out);
void nearestneighbor_cuda_launcher(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx);
void nearestneighbor_cuda_launcher_fast(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx);
void interpolation_forward_cuda_launcher(int b, int c, int m, int n, const float *points, const int *idx, const float *weight, float *out);
void interpolation_forward_cuda_launcher_fast(int b, int c, int m, int n, const float *points, const int *idx, const float *weight, float *out);
/* the reference DECLARES this one as (b, n, c, m) (interpolation_cuda_kernel.h:23) but its only caller passes (b, c, n, m)
 * (interpolation_cuda.cpp:62): the declared NAMES n/c are swapped there; positions are what link, and the names here say what they carry */
void interpolation_backward_cuda_launcher(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight, float *grad_points);
void ballquery_cuda_launcher(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx);
void ballquery_cuda_launcher_fast(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx, pa_stream_t stream);
void featuredistribute_cuda_launcher(int b, int n, int m, const float *max_xyz, const float *xyz, int *distribute_idx, pa_stream_t stream);
void featuregather_forward_cuda_launcher(int b, int n, int m, int c, const float *max_feature, const int *distribute_idx, float *distribute_feature, pa_stream_t stream);
void featuregather_backward_cuda_launcher(int b, int n, int m, int c, const float *grad_distribute_feature, const int *distribute_idx, float *grad_max_feature, pa_stream_t stream);
void labelstat_and_ballquery_cuda_launcher_fast(int b, int n, int m, float radius, int nsample, int nclass, const float *new_xyz, const float *xyz,
                                                const int *label_stat, int *idx, int *new_label_stat, pa_stream_t stream);
void labelstat_ballrange_cuda_launcher_fast(int b, int n, int m, float radius, int nclass, const float *new_xyz, const float *xyz,
                                            const int *label_stat, int *new_label_stat, pa_stream_t stream);
void labelstat_idx_cuda_launcher_fast(int b, int n, int m, int nsample, int nclass, const int *label_stat, const int *idx, int *new_label_stat, pa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PATCHAUGNET_HIP_H */
