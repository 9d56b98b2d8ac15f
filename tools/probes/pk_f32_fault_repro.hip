// Standalone reproduction (no Python, no library) of the gfx950 fault behind csrc/pa_common.h's pa_pk_plain rule:
//   v_pk_add_f32 with the half-select operand modifiers (op_sel / op_sel_hi) returns wrong values while another wave of the SIMD issues
//   v_mfma_f32_16x16x32_f16 (alone or interleaved with ordinary VALU instructions).
// Two streams: the NEIGHBOUR kernel (1024 workgroups, ~3 ms) and the VICTIM kernel (1024 workgroups), which computes d = (x-o)^2 + (y-o')^2 + (z-o'')^2 for two points
// at a time with packed instructions in one of four operand forms and with scalar instructions on the same operands in the same lane, and counts disagreements.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/pk_f32_fault_repro.hip -o tools/probes/pk_f32_fault_repro.bin && tools/probes/pk_f32_fault_repro.bin
// Measured on MI355X / ROCm 7.2 (profiles/r05_pk_f32_fault_repro.txt): forms "op_sel + neg" and "op_sel only" disagree in 1e4 .. 1e6 of 1.9e10 lane-rounds beside
// neighbours A, B and C (any wave issuing 16x16x32 MFMAs), "neg only" and "plain" in none; no form disagrees beside neighbour D (no MFMA) or alone.
// (tools/probes/pk_f32_victim.hip + corun_stress.hip are the fuller versions: 16x16x16 and fp32 MFMAs as neighbours are clean there.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void neighbour(int mode, long ticks, float *sink)
{
    const long t0 = wall_clock64();
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    float fa = 0.5f + threadIdx.x * 1e-3f, fb = 0.25f;
    while (wall_clock64() - t0 < ticks) {
        for (int r = 0; r < 32; ++r) {
            if (mode != 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            if (mode == 0) asm volatile("s_nop 15\n s_nop 7");                                      // A: the MFMA drained before the VALU instruction
            if (mode != 2) { unsigned o; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(o) : "v"(fa), "v"(fb)); fa += __uint_as_float(o & 1u); }
            if (mode == 0) asm volatile("s_nop 7");
        }
    }
    if (acc[0] + fa == 1.2345f) sink[0] = acc[1];
}

template <int FORM>
__global__ __launch_bounds__(256) void victim(int rounds, unsigned *bad)
{
    constexpr int NP = 8;
    f2 qx[NP], qy[NP], qz[NP];
    unsigned s = blockIdx.x * 7919u + threadIdx.x * 104729u + 1u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.f / 16777216.f) * 50.f - 25.f; };
    for (int h = 0; h < NP; ++h) { qx[h] = (f2){rnd(), rnd()}; qy[h] = (f2){rnd(), rnd()}; qz[h] = (f2){rnd(), rnd()}; }
    unsigned nbad = 0;
    for (int r = 0; r < rounds; ++r) {
        f2 oxy = (f2){rnd(), rnd()}, oz = (f2){rnd(), 0.f};
        f2 bx = (f2){oxy.x, oxy.x}, by = (f2){oxy.y, oxy.y}, bz = (f2){oz.x, oz.x};                  // broadcast by moves
        f2 nxy = (f2){-oxy.x, -oxy.y}, nz = (f2){-oz.x, 0.f};                                       // negated by moves
        f2 nbx = (f2){-oxy.x, -oxy.x}, nby = (f2){-oxy.y, -oxy.y}, nbz = (f2){-oz.x, -oz.x};        // both
        asm volatile("" : "+v"(bx), "+v"(by), "+v"(bz), "+v"(nxy), "+v"(nz), "+v"(nbx), "+v"(nby), "+v"(nbz));
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            f2 dx, dy, dz, d;
            if (FORM == 0) {
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(qx[h]), "v"(oxy));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(qy[h]), "v"(oxy));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dz) : "v"(qz[h]), "v"(oz));
            } else if (FORM == 1) {
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(dx) : "v"(qx[h]), "v"(nxy));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(dy) : "v"(qy[h]), "v"(nxy));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(dz) : "v"(qz[h]), "v"(nz));
            } else if (FORM == 2) {
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(qx[h]), "v"(bx));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(qy[h]), "v"(by));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dz) : "v"(qz[h]), "v"(bz));
            } else {
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(dx) : "v"(qx[h]), "v"(nbx));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(dy) : "v"(qy[h]), "v"(nby));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(dz) : "v"(qz[h]), "v"(nbz));
            }
            asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(dx));
            asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(dy));
            asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(dz));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(dx), "v"(dy));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d) : "v"(dz));
            float e[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float sx, sy, sz, sd;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(sx) : "v"(qx[h][c]), "v"(oxy.x));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(sy) : "v"(qy[h][c]), "v"(oxy.y));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(sz) : "v"(qz[h][c]), "v"(oz.x));
                asm volatile("v_mul_f32 %0, %0, %0" : "+v"(sx));
                asm volatile("v_mul_f32 %0, %0, %0" : "+v"(sy));
                asm volatile("v_mul_f32 %0, %0, %0" : "+v"(sz));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(sd) : "v"(sx), "v"(sy));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sd) : "v"(sz));
                e[c] = sd;
            }
            if (__float_as_uint(e[0]) != __float_as_uint(d.x) || __float_as_uint(e[1]) != __float_as_uint(d.y)) ++nbad;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main()
{
    hipStream_t sa, sb;
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    unsigned *bad; float *sink;
    CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&sink, 16));
    const char *nb[5] = {"A: 16x16x32 MFMA, drained, + v_cvt_pk_f16_f32", "B: 16x16x32 MFMA + v_cvt_pk_f16_f32", "C: 16x16x32 MFMA only", "D: v_cvt_pk_f16_f32 only", "none"};
    const char *vf[4] = {"op_sel + neg", "op_sel only", "neg only", "plain"};
    for (int m = 0; m < 5; ++m)
        for (int f = 0; f < 4; ++f) {
            CHECK(hipMemset(bad, 0, 4));
            CHECK(hipDeviceSynchronize());
            for (int t = 0; t < 6; ++t) {
                if (m < 4) hipLaunchKernelGGL(neighbour, dim3(1024), dim3(256), 32768, sa, m, 3000L * 100, sink);
                if (f == 0) hipLaunchKernelGGL(victim<0>, dim3(1024), dim3(256), 1024, sb, 1500, bad);
                else if (f == 1) hipLaunchKernelGGL(victim<1>, dim3(1024), dim3(256), 1024, sb, 1500, bad);
                else if (f == 2) hipLaunchKernelGGL(victim<2>, dim3(1024), dim3(256), 1024, sb, 1500, bad);
                else hipLaunchKernelGGL(victim<3>, dim3(1024), dim3(256), 1024, sb, 1500, bad);
                CHECK(hipDeviceSynchronize());
            }
            unsigned h = 0;
            CHECK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            printf("neighbour %-48s victim %-13s: packed != scalar in %10u of %.3g lane-rounds\n", nb[m], vf[f], h, 6.0 * 1024 * 256 * 1500 * 8);
        }
    return 0;
}
