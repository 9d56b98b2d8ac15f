// Co-runners for the sampling kernel (tools/probes/corun_stress.py): which kind of neighbour on the CU changes fps_reg_kernel's result?
// mode 0: fp16 MFMA in registers; 1: fp32 MFMA in registers; 2: LDS read-modify-write hammer on its own 32 KB; 3: packed fp32 VALU; 4: global memory stream;
// 5: v_cvt_pk_f16_f32; 6: v_accvgpr_write / read; 7: fp16 MFMA accumulating in AGPRs; 8: fp16 MFMA + v_cvt_pk_f16_f32 + v_pk_add_f32 interleaved (the chain16 mix);
// 9: bf16 MFMA 16x16x32; 10: fp16 MFMA 32x32x16; 11: fp16 MFMA 16x16x16 (the pre-gfx950 shape);
// 12: fp16 MFMA x32 + v_pk_add_f32; 13: fp16 MFMA x32 + v_cvt_pk_f16_f32; 14: fp32 MFMA + v_pk_add_f32; 15: fp16 MFMA 16x16x16 + v_pk_add_f32; 16: fp16 MFMA x32 + v_add_f32; 17: fp16 MFMA x32 + v_pk_mul_f32;
// 18: fp16 MFMA x32 + 2 x v_cvt_f16_f32 + v_pack_b32_f16; 19: ... + v_cvt_pkrtz_f16_f32; 20: ... + v_cvt_f16_f32 and its SDWA WORD_1 form; 21: fp16 MFMA 16x16x16 + v_cvt_pk_f16_f32;
// 22: fp32 MFMA + v_cvt_pk_f16_f32; 23: bf16 MFMA x32 + v_cvt_pk_bf16_f32; 24: v_cvt_pk_f16_f32 + v_pk_add_f32, no MFMA;
// 25: odd workgroups fp16 MFMA x32 only, even workgroups v_cvt_pk_f16_f32 only; 26-29: fp16 MFMA x32 + v_cvt_f32_f16 / v_cvt_f32_u32 / v_exp_f32 / v_mul+v_fma; 30: as 13 with the MFMA drained (s_nop) around the conversion.
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/corun_stress.hip -o tools/probes/corun_stress.so
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void corun_kernel(int mode, long spin_ticks, float *sink, float *stream_buf, long stream_n)
{
    extern __shared__ float lds[];
    const long t0 = wall_clock64();
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    float2v p = {1.0f + threadIdx.x * 1e-3f, 2.0f}, q = {0.999f, 1.001f};
    float fa = 0.5f + threadIdx.x * 1e-3f, fb = 0.25f;
    if (mode == 2) for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    long it = 0;
    while (wall_clock64() - t0 < spin_ticks) {
        if (mode == 0) { for (int r = 0; r < 64; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); }
        else if (mode == 1) { for (int r = 0; r < 64; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0); }
        else if (mode == 2) { for (int r = 0; r < 64; ++r) { const int i = (threadIdx.x * 33 + r * 257 + (int)it) & 8191; lds[i] = lds[(i + 4099) & 8191] * 1.0001f + 1.f; } }
        else if (mode == 3) { for (int r = 0; r < 256; ++r) { p = p * q + q; } }
        else if (mode == 5) { for (int r = 0; r < 128; ++r) { unsigned o; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); } }
        else if (mode == 6) { for (int r = 0; r < 128; ++r) { float o; asm volatile("v_accvgpr_write_b32 a0, %1\n s_nop 1\n v_accvgpr_read_b32 %0, a0" : "=v"(o) : "v"(fa) : "a0"); fa = o; } }
        else if (mode == 7) { for (int r = 0; r < 64; ++r) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b)); }
        else if (mode == 8) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); unsigned o; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); p = p + q; } }
        else if (mode == 9) { typedef __bf16 bf8 __attribute__((ext_vector_type(8))); bf8 ba, bb; for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(float)a[i]; bb[i] = (__bf16)(float)b[i]; } for (int r = 0; r < 64; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc, 0, 0, 0); }
        else if (mode == 10) { typedef float floatx16 __attribute__((ext_vector_type(16))); floatx16 c16 = {}; for (int r = 0; r < 32; ++r) c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c16, 0, 0, 0); acc[0] += c16[3]; }
        else if (mode == 11) { typedef _Float16 half4 __attribute__((ext_vector_type(4))); half4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]}; for (int r = 0; r < 64; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0); }
        else if (mode == 12) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q)); } }
        else if (mode == 13) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); unsigned o; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); } }
        else if (mode == 14) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q)); } }
        else if (mode == 15) { typedef _Float16 half4 __attribute__((ext_vector_type(4))); half4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]}; for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q)); } }
        else if (mode == 16) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(fb)); } }
        else if (mode == 17) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q)); } }
        else if (mode == 18) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); unsigned o, o2; asm volatile("v_cvt_f16_f32 %0, %2\n v_cvt_f16_f32 %1, %3\n v_pack_b32_f16 %0, %0, %1" : "=&v"(o), "=&v"(o2) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); } }
        else if (mode == 19) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); unsigned o; asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); } }
        else if (mode == 20) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); unsigned o; asm volatile("v_cvt_f16_f32 %0, %1\n s_nop 0\n v_cvt_f16_f32_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "=&v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); } }
        else if (mode == 21) { typedef _Float16 half4 __attribute__((ext_vector_type(4))); half4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]}; for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0); unsigned o; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); } }
        else if (mode == 22) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0); unsigned o; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); } }
        else if (mode == 23) { typedef __bf16 bf8 __attribute__((ext_vector_type(8))); bf8 ba, bb; for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(float)a[i]; bb[i] = (__bf16)(float)b[i]; } for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc, 0, 0, 0); unsigned o; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); } }
        else if (mode == 24) { for (int r = 0; r < 64; ++r) { unsigned o; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q)); } }
        else if (mode == 25) { if (blockIdx.x & 1) { for (int r = 0; r < 64; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); } else { for (int r = 0; r < 128; ++r) { unsigned o; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); } } }
        else if (mode == 26) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); float o; asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(o) : "v"(fa)); fa += o * 1e-30f; } }
        else if (mode == 27) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); float o; asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(o) : "v"(fa)); fa += o * 1e-30f; } }
        else if (mode == 28) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); float o; asm volatile("v_exp_f32 %0, %1" : "=v"(o) : "v"(fb)); fa += o * 1e-30f; } }
        else if (mode == 29) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); float o; asm volatile("v_mul_f32 %0, %1, %1\n v_fma_f32 %0, %1, %1, %0" : "=&v"(o) : "v"(fb)); fa += o * 1e-30f; } }
        else if (mode == 30) { for (int r = 0; r < 32; ++r) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0); asm volatile("s_nop 15\n s_nop 7"); unsigned o; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); asm volatile("s_nop 7"); } }
        else { for (int r = 0; r < 16; ++r) { const long i = (((long)blockIdx.x * 256 + threadIdx.x) * 16 + r + it * 4096) % stream_n; stream_buf[i] = stream_buf[(i + 1234567) % stream_n] + 1.f; } }
        ++it;
    }
    if (acc[0] + p[0] + fa + (mode == 2 ? lds[threadIdx.x] : 0.f) == 1.2345f) sink[0] = acc[1];
}
extern "C" int corun_launch(int mode, int blocks, long spin_us, float *sink, float *stream_buf, long stream_n, void *stream)
{
    hipLaunchKernelGGL(corun_kernel, dim3(blocks), dim3(256), 32768, (hipStream_t)stream, mode, spin_us * 100, sink, stream_buf, stream_n);
    return (int)hipGetLastError();
}
