// ABI plumbing for libpatchaugnet_hip.so: error text, version, and the reference's own launcher names.
#include <stdarg.h>
#include <string.h>

#include "pa_common.h"

static thread_local char g_err[512] = "";

void pa_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

PA_API const char *pa_last_error(void) { return g_err; }

// ---- stream-ordered fills as kernel launches (pa_common.h: why not hipMemset*Async)
namespace {
__global__ __launch_bounds__(256) void pa_fill32_kernel(unsigned *__restrict__ dst, size_t pitch_words, unsigned value, size_t words, size_t rows)
{
    const size_t total = words * rows;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / words, c = i - r * words;
        dst[r * pitch_words + c] = value;
    }
}
}  // namespace

int pa_fill32_2d(void *dst, size_t pitch_words, unsigned value, size_t words, size_t rows, hipStream_t st)
{
    if (!dst || words == 0 || rows == 0) return PA_OK;
    const size_t blocks = (words * rows + 255) / 256;
    hipLaunchKernelGGL(pa_fill32_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, (unsigned *)dst, pitch_words, value, words, rows);
    return hipGetLastError() == hipSuccess ? PA_OK : PA_EINVAL;
}

int pa_fill32(void *dst, unsigned value, size_t words, hipStream_t st) { return pa_fill32_2d(dst, words, value, words, 1, st); }
PA_API int pa_abi_version(void) { return 1; }

// ---- group 2: the reference's launcher symbols --------------------------------------------------------------
// The originals report a failed launch with fprintf(stderr) + exit(-1) from inside the library (SURVEY.md
// section 9.6); these print the same kind of message but return to the caller, who can read pa_last_error().
static void report(int rc, const char *name)
{
    if (rc != PA_OK) fprintf(stderr, "%s failed (%d): %s\n", name, rc, pa_last_error());
}

#define FWD(name, call) report(call, name)

PA_API void furthestsampling_cuda_launcher(int b, int n, int m, const float *dataset, float *temp, int *idxs)
{ FWD("furthestsampling_cuda_launcher", pa_furthestsampling(b, n, m, dataset, temp, idxs, nullptr)); }

PA_API void gathering_forward_cuda_launcher(int b, int c, int n, int m, const float *points, const int *idx, float *out)
{ FWD("gathering_forward_cuda_launcher", pa_gathering_forward(b, c, n, m, points, idx, out, nullptr)); }

PA_API void gathering_backward_cuda_launcher(int b, int c, int n, int m, const float *grad_out, const int *idx, float *grad_points)
{ FWD("gathering_backward_cuda_launcher", pa_gathering_backward(b, c, n, m, grad_out, idx, grad_points, nullptr)); }

PA_API void knnquery_cuda_launcher(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, int *idx, float *dist2, pa_stream_t stream)
{ FWD("knnquery_cuda_launcher", pa_knnquery(b, n, m, nsample, xyz, new_xyz, idx, dist2, stream)); }

PA_API void grouping_forward_cuda_launcher(int b, int c, int n, int m, int nsample, const float *points, const int *idx, float *out)
{ FWD("grouping_forward_cuda_launcher", pa_grouping_forward(b, c, n, m, nsample, points, idx, out, nullptr)); }

PA_API void grouping_forward_cuda_launcher_fast(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out)
{ FWD("grouping_forward_cuda_launcher_fast", pa_grouping_forward(b, c, n, npoints, nsample, points, idx, out, nullptr)); }

PA_API void grouping_backward_cuda_launcher(int b, int c, int n, int m, int nsample, const float *grad_out, const int *idx, float *grad_points)
{ FWD("grouping_backward_cuda_launcher", pa_grouping_backward(b, c, n, m, nsample, grad_out, idx, grad_points, nullptr)); }

PA_API void grouping_int_forward_cuda_launcher(int b, int c, int n, int m, int nsample, const long int *points, const int *idx, long int *out)
{ FWD("grouping_int_forward_cuda_launcher", pa_grouping_int_forward(b, c, n, m, nsample, (const int64_t *)points, idx, (int64_t *)out, nullptr)); }

PA_API void grouping_int_forward_cuda_launcher_fast(int b, int c, int n, int npoints, int nsample, const long int *points, const int *idx, long int *out)
{ FWD("grouping_int_forward_cuda_launcher_fast", pa_grouping_int_forward(b, c, n, npoints, nsample, (const int64_t *)points, idx, (int64_t *)out, nullptr)); }

PA_API void nearestneighbor_cuda_launcher(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx)
{ FWD("nearestneighbor_cuda_launcher", pa_nearestneighbor(b, n, m, unknown, known, dist2, idx, nullptr)); }

PA_API void nearestneighbor_cuda_launcher_fast(int b, int n, int m, const float *unknown, const float *known, float *dist2, int *idx)
{ FWD("nearestneighbor_cuda_launcher_fast", pa_nearestneighbor(b, n, m, unknown, known, dist2, idx, nullptr)); }

PA_API void interpolation_forward_cuda_launcher(int b, int c, int m, int n, const float *points, const int *idx, const float *weight, float *out)
{ FWD("interpolation_forward_cuda_launcher", pa_interpolation_forward(b, c, m, n, points, idx, weight, out, nullptr)); }

PA_API void interpolation_forward_cuda_launcher_fast(int b, int c, int m, int n, const float *points, const int *idx, const float *weight, float *out)
{ FWD("interpolation_forward_cuda_launcher_fast", pa_interpolation_forward(b, c, m, n, points, idx, weight, out, nullptr)); }

// NB the reference declares this one as (b, n, c, m) but passes (b, c, n, m) from interpolation_cuda.cpp, i.e. its
// parameter NAMES n/c are swapped; positions are what matter and are kept.
PA_API void interpolation_backward_cuda_launcher(int b, int c, int n, int m, const float *grad_out, const int *idx, const float *weight, float *grad_points)
{ FWD("interpolation_backward_cuda_launcher", pa_interpolation_backward(b, c, n, m, grad_out, idx, weight, grad_points, nullptr)); }

PA_API void ballquery_cuda_launcher(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx)
{ FWD("ballquery_cuda_launcher", pa_ballquery(b, n, m, radius, nsample, new_xyz, xyz, idx, nullptr)); }

PA_API void ballquery_cuda_launcher_fast(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx, pa_stream_t stream)
{ FWD("ballquery_cuda_launcher_fast", pa_ballquery(b, n, m, radius, nsample, new_xyz, xyz, idx, stream)); }

PA_API void featuredistribute_cuda_launcher(int b, int n, int m, const float *max_xyz, const float *xyz, int *distribute_idx, pa_stream_t stream)
{ FWD("featuredistribute_cuda_launcher", pa_featuredistribute(b, n, m, max_xyz, xyz, distribute_idx, stream)); }

PA_API void featuregather_forward_cuda_launcher(int b, int n, int m, int c, const float *max_feature, const int *distribute_idx, float *distribute_feature, pa_stream_t stream)
{ FWD("featuregather_forward_cuda_launcher", pa_featuregather_forward(b, n, m, c, max_feature, distribute_idx, distribute_feature, stream)); }

PA_API void featuregather_backward_cuda_launcher(int b, int n, int m, int c, const float *grad_distribute_feature, const int *distribute_idx, float *grad_max_feature, pa_stream_t stream)
{ FWD("featuregather_backward_cuda_launcher", pa_featuregather_backward(b, n, m, c, grad_distribute_feature, distribute_idx, grad_max_feature, stream)); }

PA_API void labelstat_and_ballquery_cuda_launcher_fast(int b, int n, int m, float radius, int nsample, int nclass, const float *new_xyz, const float *xyz,
                                                       const int *label_stat, int *idx, int *new_label_stat, pa_stream_t stream)
{ FWD("labelstat_and_ballquery_cuda_launcher_fast", pa_labelstat_and_ballquery(b, n, m, radius, nsample, nclass, new_xyz, xyz, label_stat, idx, new_label_stat, stream)); }

PA_API void labelstat_ballrange_cuda_launcher_fast(int b, int n, int m, float radius, int nclass, const float *new_xyz, const float *xyz,
                                                   const int *label_stat, int *new_label_stat, pa_stream_t stream)
{ FWD("labelstat_ballrange_cuda_launcher_fast", pa_labelstat_ballrange(b, n, m, radius, nclass, new_xyz, xyz, label_stat, new_label_stat, stream)); }

PA_API void labelstat_idx_cuda_launcher_fast(int b, int n, int m, int nsample, int nclass, const int *label_stat, const int *idx, int *new_label_stat, pa_stream_t stream)
{ FWD("labelstat_idx_cuda_launcher_fast", pa_labelstat_idx(b, n, m, nsample, nclass, label_stat, idx, new_label_stat, stream)); }
