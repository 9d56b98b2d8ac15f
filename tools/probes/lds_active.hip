// The selection plumbing of fps_reg_kernel (patchaugnet_amd/csrc/fps.hip) without the geometry: 256 threads, 64 KiB + 64 B of dynamic LDS, per round one DPP wave
// maximum, ds_max_u64 on a rotating slot AT OFFSET 65536, s_barrier, read back, a 16-byte read of a table entry chosen by the key.  The winner of every round is
// known in closed form, so every thread checks every round; a workgroup reports its mismatches and its HW_REG_LDS_ALLOC (where in the CU's LDS it was placed).
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC -I patchaugnet_amd/csrc tools/probes/lds_active.hip -o tools/probes/lds_active.so
#include <hip/hip_runtime.h>
#include "pa_common.h"
__global__ __launch_bounds__(256) void lds_active_kernel(int rounds, int slot_words_offset, unsigned *bad, unsigned *alloc, unsigned long long *example)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4 *table = reinterpret_cast<float4 *>(smem);
    u64 *slots = reinterpret_cast<u64 *>(smem + slot_words_offset);
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) table[i] = make_float4((float)i, (float)(i * 3), (float)(i ^ 0x55), 0.f);
    if (tid < 3) slots[tid] = 0;
    __syncthreads();
    unsigned nbad = 0;
    int slot_i = 0;
    for (int r = 1; r <= rounds; ++r) {
        const int winner = (r * 37) & 255;
        const u32 hi = tid == winner ? 0x40000000u + r : 0x3f000000u + ((tid * 131 + r * 7) & 0xffff);
        const u64 best = ((u64)hi << 32) | (u32)(tid * 16 + (r & 15) + 1);
        const u32 H = pa_wave_max_u32(hi);
        u64 *cur = slots + slot_i;
        slot_i = slot_i == 2 ? 0 : slot_i + 1;
        if (tid == 0) slots[slot_i] = 0;
        if (hi == H) asm volatile("ds_max_u64 %0, %1" : : "v"((u32)(uintptr_t)cur), "v"(best) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
        __syncthreads();
        const u64 g = *cur;
        const u64 want = ((u64)(0x40000000u + r) << 32) | (u32)(winner * 16 + (r & 15) + 1);
        const float4 e = table[(u32)g & 4095];
        const bool ok = g == want && e.x == (float)((u32)g & 4095) && e.z == (float)(((u32)g & 4095) ^ 0x55);
        if (!ok) { if (!nbad) { example[blockIdx.x * 2] = g; example[blockIdx.x * 2 + 1] = want; } ++nbad; }
    }
    if (nbad) atomicAdd(&bad[blockIdx.x], nbad);
    if (tid == 0) alloc[blockIdx.x] = __builtin_amdgcn_s_getreg((6 /* HW_REG_LDS_ALLOC */) | (0 << 6) | (31 << 11));
}
extern "C" int lds_active_launch(int blocks, int rounds, int slots_at_top, unsigned *bad, unsigned *alloc, unsigned long long *example, void *stream)
{
    const int bytes = 65536 + 64;
    hipFuncSetAttribute((const void *)lds_active_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(lds_active_kernel, dim3(blocks), dim3(256), bytes, (hipStream_t)stream, rounds, slots_at_top ? 16384 : 16384, bad, alloc, example);
    return (int)hipGetLastError();
}
