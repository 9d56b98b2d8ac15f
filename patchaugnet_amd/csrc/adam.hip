// Adam over a LIST of parameter tensors in one or two launches (gfx950): the optimizer step of the training loop
// (train_place_recognition.py:386-392: torch.optim.Adam over the model's 166 tensors, 13.5 M parameters).
//
// torch's fused multi-tensor Adam is three launches of 46-63 us at this model (profiles/r05_train_step_per_replay.csv: 162 us per step, 2.3 TB/s
// for 377 MB of parameter / gradient / moment traffic).  Here the tensor list travels in the kernel ARGUMENTS (<= 84 tensors per launch: pointers
// by value, like torch's, so a captured hipGraph bakes in exactly the addresses of the step it captured), a workgroup owns 4096 consecutive
// elements of one tensor and streams them with 16-byte accesses; the step counter is a device scalar advanced by a one-thread launch in front
// (capturable: no host value enters the arithmetic).  Arithmetic = torch.optim.Adam (amsgrad off, maximize off; weight_decay as L2 added to the
// gradient), in torch's capturable operation order:
//     m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "pa_common.h"

namespace {

constexpr int AD_MAXT = 84, AD_EPB = 4096;      // tensors per launch (the argument block stays under 4 KB: 84 x 36 + 85 x 4 bytes); elements per workgroup

struct AdamList {
    float *p[AD_MAXT];
    const float *g[AD_MAXT];
    float *m[AD_MAXT];
    float *v[AD_MAXT];
    int first_block[AD_MAXT + 1];                // workgroups [first_block[t], first_block[t + 1]) belong to tensor t
    int numel[AD_MAXT];
    int count;
};

__global__ void adam_tick_kernel(float *step) { step[0] += 1.0f; }

__global__ __launch_bounds__(256) void adam_kernel(AdamList L, const float *__restrict__ step, const float *__restrict__ lrp, float b1, float b2, float eps, float wd)
{
    // the tensor of this workgroup: binary search over <= 85 block offsets (wave-uniform)
    int lo = 0, hi = L.count;
    const int blk = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (L.first_block[mid] <= blk) lo = mid; else hi = mid;
    }
    const int t = lo;
    const long n = L.numel[t];
    const long base = (long)(blk - L.first_block[t]) * AD_EPB;
    float *__restrict__ p = L.p[t];
    const float *__restrict__ g = L.g[t];
    float *__restrict__ m = L.m[t];
    float *__restrict__ v = L.v[t];
    const float tt = step[0], lr = lrp[0];
    const float bc1 = 1.0f - powf(b1, tt), bc2 = 1.0f - powf(b2, tt);
    const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    auto upd = [&](float &pp, float gg, float &mm, float &vv) {
        if (wd != 0.f) gg = fmaf(wd, pp, gg);
        mm = fmaf(omb1, gg - mm, mm);                                   // torch: exp_avg.lerp_(grad, 1 - beta1)
        vv = fmaf(omb2 * gg, gg, b2 * vv);                              // torch: exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp -= step_size * (mm / denom);                                 // torch: param.addcdiv_(exp_avg, denom, value = -step_size)
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
#pragma unroll
    for (int it = 0; it < AD_EPB / 1024; ++it) {
        const long i = base + (long)it * 1024 + threadIdx.x * 4;
        if (vec && i + 3 < n) {
            float4 pv = *reinterpret_cast<float4 *>(p + i), mv = *reinterpret_cast<float4 *>(m + i), vv = *reinterpret_cast<float4 *>(v + i);
            const float4 gv = *reinterpret_cast<const float4 *>(g + i);
            upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
            *reinterpret_cast<float4 *>(p + i) = pv; *reinterpret_cast<float4 *>(m + i) = mv; *reinterpret_cast<float4 *>(v + i) = vv;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < n) { float pp = p[i + e], mm = m[i + e], vv = v[i + e]; upd(pp, g[i + e], mm, vv); p[i + e] = pp; m[i + e] = mm; v[i + e] = vv; }
        }
    }
}

}  // namespace

// step[0] += 1 (one thread): the first launch of an optimizer step; pa_adam_step then reads it
PA_API int pa_adam_tick(float *step, pa_stream_t stream)
{
    PA_REQUIRE(step, "pa_adam_tick: null step");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    PA_CHECK_LAUNCH("pa_adam_tick");
    return PA_OK;
}

// One Adam update of ntensors fp32 tensors (HOST arrays of device pointers p / g / m / v and element counts numel; contiguous tensors); step: device
// scalar holding t >= 1 (pa_adam_tick advances it); lr: DEVICE scalar too -- a learning-rate schedule (train_place_recognition.py:531-568) then
// reaches a captured hipGraph of the step by rewriting that scalar, not by re-capturing.  ceil(ntensors / 84) launches.
PA_API int pa_adam_step(int ntensors, float *const *p, const float *const *g, float *const *m, float *const *v, const long *numel, const float *step,
                        const float *lr, float beta1, float beta2, float eps, float weight_decay, pa_stream_t stream)
{
    PA_REQUIRE(ntensors > 0 && p && g && m && v && numel && step && lr, "pa_adam_step: bad arguments");
    for (int t0 = 0; t0 < ntensors; t0 += AD_MAXT) {
        AdamList L;
        const int cnt = ntensors - t0 < AD_MAXT ? ntensors - t0 : AD_MAXT;
        int blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            PA_REQUIRE(p[t0 + i] && g[t0 + i] && m[t0 + i] && v[t0 + i] && numel[t0 + i] > 0 && numel[t0 + i] < 2147483647L, "pa_adam_step: tensor %d: null pointer, empty or too large", t0 + i);
            L.p[i] = p[t0 + i]; L.g[i] = g[t0 + i]; L.m[i] = m[t0 + i]; L.v[i] = v[t0 + i]; L.numel[i] = (int)numel[t0 + i];
            L.first_block[i] = blocks;
            blocks += (int)((numel[t0 + i] + AD_EPB - 1) / AD_EPB);
        }
        for (int i = cnt; i <= AD_MAXT; ++i) L.first_block[i] = blocks;
        for (int i = cnt; i < AD_MAXT; ++i) { L.p[i] = nullptr; L.g[i] = nullptr; L.m[i] = nullptr; L.v[i] = nullptr; L.numel[i] = 0; }
        L.count = cnt;
        hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L, step, lr, beta1, beta2, eps, weight_decay);
    }
    PA_CHECK_LAUNCH("pa_adam_step");
    return PA_OK;
}
