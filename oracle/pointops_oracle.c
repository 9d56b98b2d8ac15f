/*
 * oracle/pointops_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded-per-cloud restatement of the point-set kernels on
 * PatchAugNet's descriptor-extraction hot path.  It exists only so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have a CPU
 * checker for the HIP kernels in patchaugnet_amd/csrc/.  Nothing under
 * patchaugnet_amd/ may import, link or call it.
 *
 * Parity status: the reference's native code is CUDA (.cu) and cannot be
 * compiled or run in this image (no nvcc, no NVIDIA device), and the reference
 * ships no golden vectors for these ops (its only native test,
 * libs/KNN_CUDA/tests/test_knn_cuda.py, needs a GPU).  This file therefore
 * follows each .cu source READ AS C under the arithmetic contract of SURVEY.md
 * section 8: IEEE fp32, operations in source order, NO FMA contraction (build
 * with -ffp-contract=off).  It is pinned (a) op-by-op against independent
 * numpy/brute-force statements in tests/test_oracle_ops.py and (b) end to end
 * by running the reference's own Python model classes (imported from
 * /root/reference in the build container only, see oracle/gen_golden.py) on top
 * of these ops and committing the resulting vectors under tests/golden/.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* libs/pointops/src/cuda_utils.h:15-18  opt_n_threads: largest power of two
 * <= work_size, clamped to [1, 1024]; computed with the same double log ratio. */
ORACLE_API int oracle_opt_n_threads(int work_size)
{
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int t = 1 << pow_2;
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return t;
}

/* squared distance, three products summed left to right, fp32, no contraction */
static inline float sqdist3(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return dx * dx + dy * dy + dz * dz;
}

/* ------------------------------------------------------------------------- *
 * K1  furthest point sampling
 * libs/pointops/src/sampling/sampling_cuda_kernel.cu:59-168 (kernel),
 * :48-54 (__update), :170-... (launcher picks block_size = opt_n_threads(n)).
 * Literal simulation of the thread-strided scan followed by the stride-halving
 * shared-memory tree, so the tie-break falls out of the structure instead of
 * being asserted.  temp is caller-filled (1e10, libs/pointops/functions/pointops.py:21).
 * ------------------------------------------------------------------------- */
ORACLE_API void oracle_furthestsampling(int b, int n, int m, const float *dataset, float *temp, int *idxs)
{
    if (m <= 0) return;
    const int bs = oracle_opt_n_threads(n);
#pragma omp parallel for schedule(dynamic, 1)
    for (int bi = 0; bi < b; ++bi) {
        float *dists = (float *)malloc(sizeof(float) * (size_t)bs);
        int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
        const float *pts = dataset + (size_t)bi * n * 3;
        float *tmp = temp + (size_t)bi * n;
        int *out = idxs + (size_t)bi * m;
        int old = 0;
        out[0] = old;                                           /* :72-74 */
        for (int j = 1; j < m; ++j) {
            const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
            for (int tid = 0; tid < bs; ++tid) {                /* one simulated thread */
                int besti = 0;
                float best = -1.0f;                             /* :79-80 */
                for (int k = tid; k < n; k += bs) {             /* :84 */
                    const float x2 = pts[k * 3 + 0], y2 = pts[k * 3 + 1], z2 = pts[k * 3 + 2];
                    const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1); /* :93 */
                    const float d2 = fminf(d, tmp[k]);          /* :94  CUDA min(float,float) */
                    tmp[k] = d2;
                    besti = d2 > best ? k : besti;              /* :96-97 */
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            for (int stride = bs / 2; stride >= 1; stride >>= 1) {   /* :102-161 */
                for (int tid = 0; tid < stride; ++tid) {
                    const float v1 = dists[tid], v2 = dists[tid + stride];   /* __update :48-54 */
                    const int i1 = dists_i[tid], i2 = dists_i[tid + stride];
                    dists[tid] = fmaxf(v1, v2);
                    dists_i[tid] = v2 > v1 ? i2 : i1;
                }
            }
            old = dists_i[0];                                   /* :164-166 */
            out[j] = old;
        }
        free(dists);
        free(dists_i);
    }
}

/* K2  gathering forward: out[b,c,j] = points[b,c,idx[b,j]]
 * libs/pointops/src/sampling/sampling_cuda_kernel.cu:6-19 */
ORACLE_API void oracle_gathering_forward(int b, int c, int n, int m, const float *points, const int *idx, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < b; ++i)
        for (int l = 0; l < c; ++l)
            for (int j = 0; j < m; ++j) {
                const int a = idx[(size_t)i * m + j];
                out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
            }
}

/* K3  gathering backward (scatter-add; reference uses atomicAdd so the fp32
 * summation order is unspecified there; here: ascending j).
 * libs/pointops/src/sampling/sampling_cuda_kernel.cu:23-36 */
ORACLE_API void oracle_gathering_backward(int b, int c, int n, int m, const float *grad_out, const int *idx, float *grad_points)
{
    for (int i = 0; i < b; ++i)
        for (int l = 0; l < c; ++l)
            for (int j = 0; j < m; ++j) {
                const int a = idx[(size_t)i * m + j];
                grad_points[((size_t)i * c + l) * n + a] += grad_out[((size_t)i * c + l) * m + j];
            }
}

/* K4  kNN query.  libs/pointops/src/knnquery/knnquery_cuda_kernel.cu:6-50.
 * best[] is double in the reference (:21) holding fp32 distances; kept double. */
ORACLE_API void oracle_knnquery(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx, float *dist2)
{
#pragma omp parallel
    {
    double *best = (double *)malloc(sizeof(double) * (size_t)nsample);
    int *besti = (int *)malloc(sizeof(int) * (size_t)nsample);
#pragma omp for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi) {
        for (int pt = 0; pt < m; ++pt) {
            const float *src = xyz + (size_t)bi * n * 3;
            const float *q = new_xyz + ((size_t)bi * m + pt) * 3;
            const float new_x = q[0], new_y = q[1], new_z = q[2];
            for (int i = 0; i < nsample; ++i) { best[i] = 1e40; besti[i] = 0; }       /* :23-26 */
            for (int k = 0; k < n; ++k) {
                const float x = src[k * 3 + 0], y = src[k * 3 + 1], z = src[k * 3 + 2];
                const float d2 = (new_x - x) * (new_x - x) + (new_y - y) * (new_y - y) + (new_z - z) * (new_z - z); /* :31 */
                for (int j = 0; j < nsample; ++j) {
                    if (d2 < best[j]) {                                                /* :33 strict */
                        for (int i = nsample - 1; i > j; --i) { best[i] = best[i - 1]; besti[i] = besti[i - 1]; }
                        best[j] = d2;
                        besti[j] = k;
                        break;
                    }
                }
            }
            int *oi = idx + ((size_t)bi * m + pt) * nsample;
            float *od = dist2 + ((size_t)bi * m + pt) * nsample;
            for (int i = 0; i < nsample; ++i) { oi[i] = besti[i]; od[i] = (float)best[i]; }   /* :44-47 */
        }
    }
    free(best);
    free(besti);
    }
}

/* K5  grouping forward: out[b,c,j,s] = points[b,c,idx[b,j,s]]
 * libs/pointops/src/grouping/grouping_cuda_kernel.cu:60-74 */
ORACLE_API void oracle_grouping_forward(int b, int c, int n, int m, int nsample, const float *points, const int *idx, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int l = 0; l < c; ++l) {
            const float *p = points + ((size_t)bi * c + l) * n;
            float *o = out + ((size_t)bi * c + l) * m * nsample;
            const int *id = idx + (size_t)bi * m * nsample;
            for (int t = 0; t < m * nsample; ++t) o[t] = p[id[t]];
        }
}

/* K6  grouping backward.  libs/pointops/src/grouping/grouping_cuda_kernel.cu:28-46
 * (atomicAdd in the reference; here ascending (j, s)). */
ORACLE_API void oracle_grouping_backward(int b, int c, int n, int m, int nsample, const float *grad_out, const int *idx, float *grad_points)
{
    for (int bi = 0; bi < b; ++bi)
        for (int l = 0; l < c; ++l) {
            float *g = grad_points + ((size_t)bi * c + l) * n;
            const float *go = grad_out + ((size_t)bi * c + l) * m * nsample;
            const int *id = idx + (size_t)bi * m * nsample;
            for (int t = 0; t < m * nsample; ++t) g[id[t]] += go[t];
        }
}

/* K8  grouping of an int64 payload.
 * libs/pointops/src/grouping_int/grouping_int_cuda_kernel.cu:33-49 */
ORACLE_API void oracle_grouping_int_forward(int b, int c, int n, int m, int nsample, const int64_t *points, const int *idx, int64_t *out)
{
    for (int bi = 0; bi < b; ++bi)
        for (int l = 0; l < c; ++l) {
            const int64_t *p = points + ((size_t)bi * c + l) * n;
            int64_t *o = out + ((size_t)bi * c + l) * m * nsample;
            const int *id = idx + (size_t)bi * m * nsample;
            for (int t = 0; t < m * nsample; ++t) o[t] = p[id[t]];
        }
}

/* K9  three nearest neighbours.
 * libs/pointops/src/interpolation/interpolation_cuda_kernel.cu:134-176 */
ORACLE_API void oracle_nearestneighbor(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi) {
        for (int pt = 0; pt < n; ++pt) {
            const float *kn = known + (size_t)bi * m * 3;
            const float *u = unknown + ((size_t)bi * n + pt) * 3;
            const float ux = u[0], uy = u[1], uz = u[2];
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;                          /* :149 */
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                const float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
                const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);   /* :155 */
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = k;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = k;
                } else if (d < best3) {
                    best3 = d; besti3 = k;
                }
            }
            float *od = dist2 + ((size_t)bi * n + pt) * 3;
            int *oi = idx + ((size_t)bi * n + pt) * 3;
            od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
            oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
        }
    }
}

/* K10  three-point weighted interpolation, order (w0*p0 + w1*p1) + w2*p2.
 * libs/pointops/src/interpolation/interpolation_cuda_kernel.cu:181-195 */
ORACLE_API void oracle_interpolation_forward(int b, int c, int m, int n, const float *points, const int *idx, const float *weight, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int bi = 0; bi < b; ++bi)
        for (int l = 0; l < c; ++l) {
            const float *p = points + ((size_t)bi * c + l) * m;
            float *o = out + ((size_t)bi * c + l) * n;
            for (int pt = 0; pt < n; ++pt) {
                const float *w = weight + ((size_t)bi * n + pt) * 3;
                const int *id = idx + ((size_t)bi * n + pt) * 3;
                o[pt] = w[0] * p[id[0]] + w[1] * p[id[1]] + w[2] * p[id[2]];          /* :194 */
            }
        }
}

/* K11  interpolation backward.
 * libs/pointops/src/interpolation/interpolation_cuda_kernel.cu:90-114 */
ORACLE_API void oracle_interpolation_backward(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight, float *grad_points)
{
    for (int bi = 0; bi < b; ++bi)
        for (int l = 0; l < c; ++l) {
            float *g = grad_points + ((size_t)bi * c + l) * m;
            const float *go = grad_out + ((size_t)bi * c + l) * n;
            for (int j = 0; j < n; ++j) {
                const float *w = weight + ((size_t)bi * n + j) * 3;
                const int *id = idx + ((size_t)bi * n + j) * 3;
                g[id[0]] += go[j] * w[0];
                g[id[1]] += go[j] * w[1];
                g[id[2]] += go[j] * w[2];
            }
        }
}

/* K13  ball query.  libs/pointops/src/ballquery/ballquery_cuda_kernel.cu:47-80.
 * idx is caller-zeroed (libs/pointops/functions/pointops.py:189). */
ORACLE_API void oracle_ballquery(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx)
{
    const float radius2 = radius * radius;                                            /* :56 */
    for (int bi = 0; bi < b; ++bi) {
        const float *src = xyz + (size_t)bi * n * 3;
        for (int pt = 0; pt < m; ++pt) {
            const float *q = new_xyz + ((size_t)bi * m + pt) * 3;
            const float new_x = q[0], new_y = q[1], new_z = q[2];
            int *o = idx + ((size_t)bi * m + pt) * nsample;
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                const float x = src[k * 3 + 0], y = src[k * 3 + 1], z = src[k * 3 + 2];
                const float d2 = (new_x - x) * (new_x - x) + (new_y - y) * (new_y - y) + (new_z - z) * (new_z - z);
                if (d2 < radius2) {
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) o[l] = k;                   /* :68-72 */
                    o[cnt] = k;
                    ++cnt;
                    if (cnt >= nsample) break;
                }
            }
        }
    }
}

/* K14  nearest "max" point per point.
 * libs/pointops/src/featuredistribute/featuredistribute_cuda_kernel.cu:4-30 */
ORACLE_API void oracle_featuredistribute(int b, int n, int m, const float *max_xyz, const float *xyz, int *distribute_idx)
{
    for (int bi = 0; bi < b; ++bi)
        for (int pt = 0; pt < m; ++pt) {
            const float *p = xyz + ((size_t)bi * m + pt) * 3;
            const float x = p[0], y = p[1], z = p[2];
            float min_dist2 = 100000;                                                 /* :17-18 */
            int min_dist_idx = -1;
            for (int k = 0; k < n; ++k) {
                const float *mx = max_xyz + ((size_t)bi * n + k) * 3;
                const float d2 = (mx[0] - x) * (mx[0] - x) + (mx[1] - y) * (mx[1] - y) + (mx[2] - z) * (mx[2] - z);
                if (d2 < min_dist2) { min_dist_idx = k; min_dist2 = d2; }
            }
            distribute_idx[(size_t)bi * m + pt] = min_dist_idx;
        }
}

/* K15  featuregather forward / backward.
 * libs/pointops/src/featuredistribute/featuredistribute_cuda_kernel.cu:53-65, :89-101 */
ORACLE_API void oracle_featuregather_forward(int b, int n, int m, int c, const float *max_feature, const int *distribute_idx, float *distribute_feature)
{
    oracle_gathering_forward(b, c, n, m, max_feature, distribute_idx, distribute_feature);
}
ORACLE_API void oracle_featuregather_backward(int b, int n, int m, int c, const float *grad_distribute_feature, const int *distribute_idx, float *grad_max_feature)
{
    oracle_gathering_backward(b, c, n, m, grad_distribute_feature, distribute_idx, grad_max_feature);
}

/* K16  label statistics.
 * libs/pointops/src/labelstat/labelstat_cuda_kernel.cu:6-49, :74-105, :131-151 */
ORACLE_API void oracle_labelstat_and_ballquery(int b, int n, int m, float radius, int nsample, int nclass,
                                               const float *new_xyz, const float *xyz, const int *label_stat, int *idx, int *new_label_stat)
{
    const float radius2 = radius * radius;
    for (int bi = 0; bi < b; ++bi)
        for (int pt = 0; pt < m; ++pt) {
            const float *q = new_xyz + ((size_t)bi * m + pt) * 3;
            const float new_x = q[0], new_y = q[1], new_z = q[2];
            int *o = idx + ((size_t)bi * m + pt) * nsample;
            int *ls = new_label_stat + ((size_t)bi * m + pt) * nclass;
            for (int i = 0; i < nclass; ++i) ls[i] = 0;
            int cnt = 0;
            for (int k = 0; k < n; ++k) {
                const float *p = xyz + ((size_t)bi * n + k) * 3;
                const float d2 = (new_x - p[0]) * (new_x - p[0]) + (new_y - p[1]) * (new_y - p[1]) + (new_z - p[2]) * (new_z - p[2]);
                if (d2 < radius2) {
                    for (int i = 0; i < nclass; ++i) ls[i] += label_stat[((size_t)bi * n + k) * nclass + i];
                    if (cnt == 0)
                        for (int l = 0; l < nsample; ++l) o[l] = k;
                    o[cnt] = k;
                    ++cnt;
                    if (cnt >= nsample) break;
                }
            }
        }
}
ORACLE_API void oracle_labelstat_ballrange(int b, int n, int m, float radius, int nclass,
                                           const float *new_xyz, const float *xyz, const int *label_stat, int *new_label_stat)
{
    const float radius2 = radius * radius;
    for (int bi = 0; bi < b; ++bi)
        for (int pt = 0; pt < m; ++pt) {
            const float *q = new_xyz + ((size_t)bi * m + pt) * 3;
            const float new_x = q[0], new_y = q[1], new_z = q[2];
            int *ls = new_label_stat + ((size_t)bi * m + pt) * nclass;
            for (int i = 0; i < nclass; ++i) ls[i] = 0;
            for (int k = 0; k < n; ++k) {
                const float *p = xyz + ((size_t)bi * n + k) * 3;
                const float d2 = (new_x - p[0]) * (new_x - p[0]) + (new_y - p[1]) * (new_y - p[1]) + (new_z - p[2]) * (new_z - p[2]);
                if (d2 < radius2)
                    for (int i = 0; i < nclass; ++i) ls[i] += label_stat[((size_t)bi * n + k) * nclass + i];
            }
        }
}
ORACLE_API void oracle_labelstat_idx(int b, int n, int m, int nsample, int nclass, const int *label_stat, const int *idx, int *new_label_stat)
{
    for (int bi = 0; bi < b; ++bi)
        for (int pt = 0; pt < m; ++pt) {
            int *ls = new_label_stat + ((size_t)bi * m + pt) * nclass;
            const int *id = idx + ((size_t)bi * m + pt) * nsample;
            for (int i = 0; i < nclass; ++i) ls[i] = 0;
            for (int k = 0; k < nsample; ++k)
                for (int i = 0; i < nclass; ++i) ls[i] += label_stat[((size_t)bi * n + id[k]) * nclass + i];
        }
}

/* ------------------------------------------------------------------------- *
 * C1  Chamfer forward, one direction: for each p in xyz1 the nearest q in xyz2.
 * libs/chamfer_dist/chamfer.cu:15-145.  Tiles of 512 points of xyz2; inside a
 * tile the first element is taken unconditionally (k == 0) and later ones by
 * strict '<'; across tiles the running result is replaced on strict '>' (:137).
 * ------------------------------------------------------------------------- */
static void chamfer_one_direction(int batch_size, int n, const float *xyz1, int m, const float *xyz2, float *dist, int *indexes)
{
    const int batch = 512;
    for (int i = 0; i < batch_size; ++i)
        for (int k2 = 0; k2 < m; k2 += batch) {
            const int end_k = (m < k2 + batch ? m : k2 + batch) - k2;
            const float *buf = xyz2 + ((size_t)i * m + k2) * 3;
            for (int j = 0; j < n; ++j) {
                const float x1 = xyz1[((size_t)i * n + j) * 3 + 0];
                const float y1 = xyz1[((size_t)i * n + j) * 3 + 1];
                const float z1 = xyz1[((size_t)i * n + j) * 3 + 2];
                float best_dist = 0;
                int best_dist_index = 0;
                for (int k = 0; k < end_k; ++k) {
                    const float x2 = buf[k * 3 + 0] - x1;
                    const float y2 = buf[k * 3 + 1] - y1;
                    const float z2 = buf[k * 3 + 2] - z1;
                    const float d = x2 * x2 + y2 * y2 + z2 * z2;                      /* :42-45 */
                    if (k == 0 || d < best_dist) { best_dist = d; best_dist_index = k + k2; }
                }
                if (k2 == 0 || dist[(size_t)i * n + j] > best_dist) {                /* :137 */
                    dist[(size_t)i * n + j] = best_dist;
                    indexes[(size_t)i * n + j] = best_dist_index;
                }
            }
        }
}
/* libs/chamfer_dist/chamfer.cu:147-171 */
ORACLE_API void oracle_chamfer_forward(int batch_size, int n, int m, const float *xyz1, const float *xyz2,
                                       float *dist1, float *dist2, int *idx1, int *idx2)
{
    chamfer_one_direction(batch_size, n, xyz1, m, xyz2, dist1, idx1);
    chamfer_one_direction(batch_size, m, xyz2, n, xyz1, dist2, idx2);
}

/* C2  Chamfer backward.  libs/chamfer_dist/chamfer.cu:173-229 (atomics there;
 * ascending j here).  grad_xyz1/grad_xyz2 are zero-initialised by this call. */
static void chamfer_grad_one_direction(int b, int n, const float *xyz1, int m, const float *xyz2,
                                       const float *grad_dist1, const int *idx1, float *grad_xyz1, float *grad_xyz2)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float x1 = xyz1[((size_t)i * n + j) * 3 + 0];
            const float y1 = xyz1[((size_t)i * n + j) * 3 + 1];
            const float z1 = xyz1[((size_t)i * n + j) * 3 + 2];
            const int j2 = idx1[(size_t)i * n + j];
            const float x2 = xyz2[((size_t)i * m + j2) * 3 + 0];
            const float y2 = xyz2[((size_t)i * m + j2) * 3 + 1];
            const float z2 = xyz2[((size_t)i * m + j2) * 3 + 2];
            const float g = grad_dist1[(size_t)i * n + j] * 2;
            grad_xyz1[((size_t)i * n + j) * 3 + 0] += g * (x1 - x2);
            grad_xyz1[((size_t)i * n + j) * 3 + 1] += g * (y1 - y2);
            grad_xyz1[((size_t)i * n + j) * 3 + 2] += g * (z1 - z2);
            grad_xyz2[((size_t)i * m + j2) * 3 + 0] += -(g * (x1 - x2));
            grad_xyz2[((size_t)i * m + j2) * 3 + 1] += -(g * (y1 - y2));
            grad_xyz2[((size_t)i * m + j2) * 3 + 2] += -(g * (z1 - z2));
        }
}
ORACLE_API void oracle_chamfer_backward(int b, int n, int m, const float *xyz1, const float *xyz2, const int *idx1, const int *idx2,
                                        const float *grad_dist1, const float *grad_dist2, float *grad_xyz1, float *grad_xyz2)
{
    memset(grad_xyz1, 0, sizeof(float) * (size_t)b * n * 3);
    memset(grad_xyz2, 0, sizeof(float) * (size_t)b * m * 3);
    chamfer_grad_one_direction(b, n, xyz1, m, xyz2, grad_dist1, idx1, grad_xyz1, grad_xyz2);
    chamfer_grad_one_direction(b, m, xyz2, n, xyz1, grad_dist2, idx2, grad_xyz2, grad_xyz1);
}

/* ------------------------------------------------------------------------- *
 * N1-N3  KNN_CUDA brute-force kNN in generic dimension.
 * libs/KNN_CUDA/knn_cuda/csrc/cuda/knn.cu:29-93 (distance matrix, accumulated
 * in dimension order, zero padding adds +0), :105-167 (per-column insertion:
 * strict '<' against the current k-th, inserted before the first strictly
 * greater entry => order (dist asc, row asc)), :178-183 (sqrt), 1-based int64
 * row ids (:124, :146, :161).  ref is (dim, nr), query is (dim, nq).
 * ------------------------------------------------------------------------- */
ORACLE_API void oracle_knn_generic(const float *ref, int nr, const float *query, int nq, int dim, int k, float *dist_out, int64_t *ind_out)
{
    float *col = (float *)malloc(sizeof(float) * (size_t)nr);
    float *bd = (float *)malloc(sizeof(float) * (size_t)k);
    int64_t *bi = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
    for (int q = 0; q < nq; ++q) {
        for (int r = 0; r < nr; ++r) {
            float ssd = 0;
            for (int d = 0; d < dim; ++d) {
                const float tmp = ref[(size_t)d * nr + r] - query[(size_t)d * nq + q];   /* :80-83 */
                ssd += tmp * tmp;
            }
            col[r] = ssd;
        }
        int filled = 0;
        for (int r = 0; r < nr; ++r) {
            const float cd = col[r];
            if (filled < k) {
                int pos = filled;
                for (int a = 0; a < filled; ++a) if (bd[a] > cd) { pos = a; break; }
                for (int j = filled; j > pos; --j) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; }
                bd[pos] = cd; bi[pos] = r + 1;
                ++filled;
            } else if (cd < bd[k - 1]) {
                int pos = k - 1;
                for (int a = 0; a < k - 1; ++a) if (bd[a] > cd) { pos = a; break; }
                for (int j = k - 1; j > pos; --j) { bd[j] = bd[j - 1]; bi[j] = bi[j - 1]; }
                bd[pos] = cd; bi[pos] = r + 1;
            }
        }
        for (int j = 0; j < k; ++j) {
            dist_out[(size_t)j * nq + q] = sqrtf(bd[j]);
            ind_out[(size_t)j * nq + q] = bi[j];
        }
    }
    free(col); free(bd); free(bi);
}

/* ------------------------------------------------------------------------- *
 * E1-E8  auction-algorithm EMD forward.  libs/emd_module/emd_cuda.cu:228-282.
 * The reference's GetMax/Assign kernels race (last writer wins, :181-215) and
 * its Bid kernel splits the object range over a data-dependent number of
 * threads (:108-118, :165-173), so its assignment is not reproducible bit for
 * bit.  This restatement fixes the free choices deterministically:
 *   - among equal-valued objects the LOWEST index wins the bid;
 *   - among bidders whose increment matches the maximum within 1e-6 the
 *     HIGHEST point index wins (the "last writer" under in-order execution).
 * Parity with it is therefore checked on the mean distance and on structural
 * properties, not on the assignment vector.  Returns 1 / -1 like :236-249.
 * State tensors follow libs/emd_module/emd_module.py:42-53 (caller-initialised).
 * ------------------------------------------------------------------------- */
ORACLE_API int oracle_emd_forward(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *assignment,
                                  float *price, int *assignment_inv, int *bid, float *bid_increments, float *max_increments,
                                  int *max_idx, float eps, int iters)
{
    if (n != m) return -1;
    if (b > 512) return -1;
    if (n % 1024 != 0) return -1;
    for (int it = 0; it < iters; ++it) {
        const int last = (it == iters - 1);
        for (int i = 0; i < b; ++i) {
            int *ass = assignment + (size_t)i * n;
            int *ass_inv = assignment_inv + (size_t)i * n;
            float *pr = price + (size_t)i * n;
            int *bd = bid + (size_t)i * n;
            float *binc = bid_increments + (size_t)i * n;
            float *minc = max_increments + (size_t)i * n;
            int *mi = max_idx + (size_t)i * n;
            /* Bid (:95-179) for every unassigned point */
            for (int j = 0; j < n; ++j) {
                if (ass[j] != -1) continue;
                const float x1 = xyz1[((size_t)i * n + j) * 3 + 0];
                const float y1 = xyz1[((size_t)i * n + j) * 3 + 1];
                const float z1 = xyz1[((size_t)i * n + j) * 3 + 2];
                float best = -1e9f, better = -1e9f;
                int best_i = -1;
                for (int k = 0; k < n; ++k) {
                    const float x2 = xyz2[((size_t)i * n + k) * 3 + 0] - x1;
                    const float y2 = xyz2[((size_t)i * n + k) * 3 + 1] - y1;
                    const float z2 = xyz2[((size_t)i * n + k) * 3 + 2] - z1;
                    /* :146  "3.0 - sqrtf(..) - price": 3.0 is a double literal, so the chain is
                     * evaluated in double and rounded to float on assignment */
                    const float dd = (float)(3.0 - (double)sqrtf(x2 * x2 + y2 * y2 + z2 * z2) - (double)pr[k]);
                    if (dd > best) { better = best; best = dd; best_i = k; }
                    else if (dd > better) { better = dd; }
                }
                bd[j] = best_i;
                binc[j] = best - better + eps;
                if (binc[j] > minc[best_i]) minc[best_i] = binc[j];                   /* atomicMax :10-20, :176 */
            }
            /* GetMax (:181-194) */
            for (int j = 0; j < n; ++j) {
                if (ass[j] != -1) continue;
                const int bid_id = bd[j];
                const float bid_inc = binc[j];
                const float max_inc = minc[bid_id];
                if (bid_inc - 1e-6 <= max_inc && max_inc <= bid_inc + 1e-6) mi[bid_id] = j;
            }
            /* Assign (:196-215); snapshot of "unassigned" taken first because the
             * kernel tests assignment[j] == -1 concurrently for all j */
            for (int j = 0; j < n; ++j) {
                if (ass[j] != -1) continue;
                const int bid_id = bd[j];
                if (last || mi[bid_id] == j) {
                    const float bid_inc = binc[j];
                    const int ai = ass_inv[bid_id];
                    if (!last && ai != -1) ass[ai] = -2;  /* evicted this round: marked, released below */
                    ass_inv[bid_id] = j;
                    ass[j] = bid_id;
                    pr[bid_id] += bid_inc;
                    minc[bid_id] = -1e9f;
                }
            }
            for (int j = 0; j < n; ++j) if (ass[j] == -2) ass[j] = -1;
        }
    }
    for (int i = 0; i < b; ++i)                                                       /* CalcDist :217-226 */
        for (int j = 0; j < n; ++j) {
            const int k = assignment[(size_t)i * n + j];
            const float dx = xyz1[((size_t)i * n + j) * 3 + 0] - xyz2[((size_t)i * n + k) * 3 + 0];
            const float dy = xyz1[((size_t)i * n + j) * 3 + 1] - xyz2[((size_t)i * n + k) * 3 + 1];
            const float dz = xyz1[((size_t)i * n + j) * 3 + 2] - xyz2[((size_t)i * n + k) * 3 + 2];
            dist[(size_t)i * n + j] = dx * dx + dy * dy + dz * dz;
        }
    return 1;
}

/* E9  EMD backward (gradient w.r.t. xyz1 only).  libs/emd_module/emd_cuda.cu:284-300 */
ORACLE_API void oracle_emd_backward(int b, int n, const float *xyz1, const float *xyz2, const float *grad_dist, const int *idx, float *grad_xyz)
{
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const int j2 = idx[(size_t)i * n + j];
            const float g = grad_dist[(size_t)i * n + j] * 2;
            for (int t = 0; t < 3; ++t)
                grad_xyz[((size_t)i * n + j) * 3 + t] += g * (xyz1[((size_t)i * n + j) * 3 + t] - xyz2[((size_t)i * n + j2) * 3 + t]);
        }
}
