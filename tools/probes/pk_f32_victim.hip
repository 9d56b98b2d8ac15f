// Packed fp32 arithmetic as the sampling kernel issues it (v_pk_add_f32 with op_sel / op_sel_hi broadcast and neg modifiers, v_pk_mul_f32; fps.hip PAIRED) against the
// same arithmetic in scalar instructions, on the same operands, in the same lane: d = (x - o)^2 + (y - o')^2 + (z - o'')^2 for two points at a time.  Both forms are IEEE
// and must give the same bits; every lane counts the rounds where they do not.  Run beside pa_linear_f16 / pa_linear on another stream (tools/probes/pk_f32_victim.py).
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/pk_f32_victim.hip -o tools/probes/pk_f32_victim.so
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NP = 8;   // pairs per lane
template <int FORM>
__global__ __launch_bounds__(256) void pk_victim_kernel(int rounds, unsigned lds_words, unsigned *bad, float *example)
{
    extern __shared__ float pad[];       // only to make the workgroup as large in LDS as the sampling kernel's (co-residency pattern)
    if (lds_words) pad[threadIdx.x] = 0.f;
    f2 qx[NP], qy[NP], qz[NP];
    unsigned s = blockIdx.x * 7919u + threadIdx.x * 104729u + 1u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.f / 16777216.f) * 50.f - 25.f; };
#pragma unroll
    for (int h = 0; h < NP; ++h) { qx[h] = (f2){rnd(), rnd()}; qy[h] = (f2){rnd(), rnd()}; qz[h] = (f2){rnd(), rnd()}; }
    unsigned nbad = 0;
    for (int r = 0; r < rounds; ++r) {
        f2 oxy = (f2){rnd(), rnd()};     // (ox, oy) in one pair, oz in the low half of another: the broadcast forms pick halves
        f2 oz_ = (f2){rnd(), 0.f};
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            f2 dx, dy, dz, d;
            if (FORM == 0) {
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(qx[h]), "v"(oxy));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(qy[h]), "v"(oxy));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dz) : "v"(qz[h]), "v"(oz_));
            } else if (FORM == 2) {   // negation modifiers only: the broadcast is done beforehand
                f2 bx = (f2){oxy.x, oxy.x}, by = (f2){oxy.y, oxy.y}, bz = (f2){oz_.x, oz_.x};
                asm volatile("" : "+v"(bx), "+v"(by), "+v"(bz));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(qx[h]), "v"(bx));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(qy[h]), "v"(by));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dz) : "v"(qz[h]), "v"(bz));
            } else if (FORM == 3) {   // broadcast modifiers only: the negation is done beforehand
                f2 nxy = (f2){-oxy.x, -oxy.y}, nz_ = (f2){-oz_.x, 0.f};
                asm volatile("" : "+v"(nxy), "+v"(nz_));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(dx) : "v"(qx[h]), "v"(nxy));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(dy) : "v"(qy[h]), "v"(nxy));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(dz) : "v"(qz[h]), "v"(nz_));
            } else {   // no operand modifiers: the broadcast and the negation are done beforehand with moves
                f2 nx = (f2){-oxy.x, -oxy.x}, ny = (f2){-oxy.y, -oxy.y}, nz = (f2){-oz_.x, -oz_.x};
                asm volatile("" : "+v"(nx), "+v"(ny), "+v"(nz));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(dx) : "v"(qx[h]), "v"(nx));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(dy) : "v"(qy[h]), "v"(ny));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(dz) : "v"(qz[h]), "v"(nz));
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
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(sz) : "v"(qz[h][c]), "v"(oz_.x));
                asm volatile("v_mul_f32 %0, %0, %0" : "+v"(sx));
                asm volatile("v_mul_f32 %0, %0, %0" : "+v"(sy));
                asm volatile("v_mul_f32 %0, %0, %0" : "+v"(sz));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(sd) : "v"(sx), "v"(sy));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(sd) : "v"(sz));
                e[c] = sd;
            }
            if (__float_as_uint(e[0]) != __float_as_uint(d.x) || __float_as_uint(e[1]) != __float_as_uint(d.y)) {
                if (!nbad) { example[0] = d.x; example[1] = e[0]; example[2] = d.y; example[3] = e[1]; example[4] = qx[h].x; example[5] = oxy.x; example[6] = (float)r; example[7] = (float)h; }
                ++nbad;
            }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}
// v_fma_mix_f32 with op_sel / op_sel_hi on its fp16 operand (how fpx16_kernel interpolates its fp16 table) against v_cvt_f32_f16 + v_fma_f32 on the same values.
__global__ __launch_bounds__(256) void mix_victim_kernel(int rounds, unsigned *bad, float *example)
{
    unsigned s = blockIdx.x * 7919u + threadIdx.x * 104729u + 1u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.f / 16777216.f) * 8.f - 4.f; };
    unsigned h[NP];
    float acc0[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const _Float16 a = (_Float16)rnd(), b = (_Float16)rnd();
        h[i] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
        acc0[i] = rnd();
    }
    unsigned nbad = 0;
    for (int r = 0; r < rounds; ++r) {
        const float w = rnd();
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float mlo, mhi, tlo, thi, elo, ehi;
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(mlo) : "v"(w), "v"(h[i]), "v"(acc0[i]));
            asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(mhi) : "v"(w), "v"(h[i]), "v"(acc0[i]));
            unsigned hs;
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(tlo) : "v"(h[i]));
            asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(hs) : "v"(h[i]));
            asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(thi) : "v"(hs));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(elo) : "v"(w), "v"(tlo), "v"(acc0[i]));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ehi) : "v"(w), "v"(thi), "v"(acc0[i]));
            if (__float_as_uint(mlo) != __float_as_uint(elo) || __float_as_uint(mhi) != __float_as_uint(ehi)) {
                if (!nbad) { example[0] = mlo; example[1] = elo; example[2] = mhi; example[3] = ehi; example[4] = w; example[5] = acc0[i]; example[6] = (float)r; example[7] = (float)i; }
                ++nbad;
            }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}
extern "C" int pk_victim_launch(int form, int blocks, int rounds, int lds_bytes, unsigned *bad, float *example, void *stream)
{
    (void)hipFuncSetAttribute((const void *)pk_victim_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute((const void *)pk_victim_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute((const void *)pk_victim_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute((const void *)pk_victim_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (form == 4) hipLaunchKernelGGL(mix_victim_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rounds, bad, example);
    else if (form == 2) hipLaunchKernelGGL(pk_victim_kernel<2>, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, rounds, (unsigned)lds_bytes / 4, bad, example);
    else if (form == 3) hipLaunchKernelGGL(pk_victim_kernel<3>, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, rounds, (unsigned)lds_bytes / 4, bad, example);
    else if (form) hipLaunchKernelGGL(pk_victim_kernel<1>, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, rounds, (unsigned)lds_bytes / 4, bad, example);
    else hipLaunchKernelGGL(pk_victim_kernel<0>, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, rounds, (unsigned)lds_bytes / 4, bad, example);
    return (int)hipGetLastError();
}
