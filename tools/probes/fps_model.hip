// Issue / latency constants behind the sampling round's latency model (bench.py -> roofline_latency):
// what ONE wave per SIMD (the sampling kernel's occupancy: 256 threads per cloud, one cloud per CU) pays per instruction class, in shader
// cycles (s_memtime), on the instruction classes the round of fps_reg_kernel<256,16> is made of -- and the same with two waves per SIMD
// (512 threads) to see which classes a second wave would overlap.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fno-vectorize tools/probes/fps_model.hip -o tools/probes/fps_model.bin && tools/probes/fps_model.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

enum { K_ADD_IND, K_ADD_DEP, K_PKADD_IND, K_PKMUL_DEP, K_PKMIX, K_MIN_IND, K_KEY64, K_DPP, K_ATOMIC, K_BARRIER, K_LDS64, K_LDS96, K_TAIL, K_ROUND, K_COUNT };
static const char *NAMES[K_COUNT] = {"v_add_f32 x16 independent", "v_add_f32 x16 dependent", "v_pk_add_f32 x16 independent", "v_pk_mul_f32 x16 dependent",
                                     "pair block: 3 pk_add + 3 pk_mul + 2 pk_add + 2 v_min + 2 x (cmp_u64 + 2 cndmask) [16 instr] + per round 3 v_xor + 3 v_mov", "v_min_f32 x16 independent",
                                     "v_cmp_gt_u64 + 2 v_cndmask (dependent key maximum) x8  [24 instr]", "wave max: 6 x (v_max_u32_dpp + s_nop 1) + v_readlane",
                                     "ds_max_u64 (one lane) + s_waitcnt lgkmcnt(0)", "s_barrier (all waves of the workgroup)", "ds_read_b64 + wait (dependent address)",
                                     "ds_read_b96 + wait (dependent address)", "tail: ds_max_u64 + wait + s_barrier + ds_read_b64 + wait + not/lshl_add + ds_read_b96 + wait",
                                     "whole synthetic round: 3 v_xor + 3 v_mov + 8 pair blocks + wave max + tail"};
static const int INSTR[K_COUNT] = {16, 16, 16, 16, 16 + 6, 16, 24, 13, 2, 1, 2, 2, 9, 6 + 8 * 16 + 13 + 9};

__device__ __forceinline__ u64 now() { return __builtin_amdgcn_s_memtime(); }

template <int KIND>
__global__ void probe(int iters, u64 *out, float seed)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + tid * 1e-3f + i;
    f2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = (f2){seed + i, seed + tid + i};
    f2 o = (f2){seed, seed * 0.5f};
    u32 klo = tid, khi = __float_as_uint(seed + tid);
    u32 blo = 0, bhi = 0;
    float tmin[16];
    for (int i = 0; i < 16; ++i) tmin[i] = 1e10f;
    for (int i = tid; i < 4096 * 4 + 64; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    u32 addr = 4096 * 16;        // slot
    u32 slotv = 0;
    const u64 t0 = now();
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_ADD_IND) {
            asm volatile(REP4("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[4]));
        } else if (KIND == K_ADD_DEP) {
            asm volatile(REP16("v_add_f32 %0, %0, %1\n") : "+v"(a[0]) : "v"(a[4]));
        } else if (KIND == K_PKADD_IND) {
            asm volatile(REP4("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(o));
        } else if (KIND == K_PKMUL_DEP) {
            asm volatile(REP16("v_pk_mul_f32 %0, %0, %1\n") : "+v"(p[0]) : "v"(o));
        } else if (KIND == K_MIN_IND) {
            asm volatile(REP4("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4\n")
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[4]));
        } else if (KIND == K_KEY64) {
            u64 best = ((u64)bhi << 32) | blo;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                u64 key = ((u64)(khi + r) << 32) | klo;
                asm volatile("" : "+v"(key));
                best = best > key ? best : key;
            }
            blo = (u32)best; bhi = (u32)(best >> 32);
        }
        if (KIND == K_PKMIX || KIND == K_ROUND) {
            // the kernel's pair block (fps.hip, PAIRED): two points' distances, minima and the running 64-bit key maximum
            u64 best = 0;
            asm volatile("" : "+v"(o));
            // the shipped form (round 5): (-o, -o) pairs materialised once per round (3 v_xor + 3 v_mov), the subtractions as plain v_pk_add_f32 -- no operand
            // modifiers on packed fp32 (csrc/pa_common.h, pa_pk_plain)
            f2 nx = (f2){-o.x, -o.x}, ny = (f2){-o.y, -o.y}, nz = (f2){-o.x, -o.x};
            asm("" : "+v"(nx), "+v"(ny), "+v"(nz));
#pragma unroll
            for (int h = 0; h < (KIND == K_ROUND ? 8 : 1); ++h) {
                const f2 dx = p[h & 7] + nx, dy = p[(h + 1) & 7] + ny, dz = p[(h + 2) & 7] + nz;
                const f2 d = dx * dx + dy * dy + dz * dz;
                float m0, m1;
                asm("v_min_f32 %0, %1, %2" : "=v"(m0) : "v"(d.x), "v"(tmin[2 * h]));
                asm("v_min_f32 %0, %1, %2" : "=v"(m1) : "v"(d.y), "v"(tmin[2 * h + 1]));
                tmin[2 * h] = m0; tmin[2 * h + 1] = m1;
                const u64 k0 = ((u64)__float_as_uint(m0) << 32) | (klo + 2 * h), k1 = ((u64)__float_as_uint(m1) << 32) | (klo + 2 * h + 1);
                best = best > k0 ? best : k0;
                best = best > k1 ? best : k1;
            }
            blo = (u32)best; bhi = (u32)(best >> 32);
        }
        if (KIND == K_DPP || KIND == K_ROUND) {
            u32 v = bhi, s;
            asm volatile("s_nop 1\n"
                         "v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                         "v_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                         "v_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                         "v_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                         "v_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
                         "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 0\n"
                         "v_readlane_b32 %1, %0, 63\n"
                         : "+v"(v), "=s"(s));
            bhi = (bhi & 0xffff0000u) | (s & 0xffffu);
        }
        if (KIND == K_ATOMIC) {
            u64 best = ((u64)bhi << 32) | blo;
            if ((tid & 63) == 0) asm volatile("ds_max_u64 %0, %1" : : "v"(addr), "v"(best) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
        } else if (KIND == K_BARRIER) {
            asm volatile("s_barrier" : : : "memory");
        } else if (KIND == K_LDS64) {
            u64 g;
            asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(g) : "v"(addr + (slotv & 8)) : "memory");
            slotv = (u32)g;
        } else if (KIND == K_LDS96) {
            float x, y, z;
            typedef float f3 __attribute__((ext_vector_type(3)));
            f3 c;
            asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(c) : "v"((slotv & 0xff0u)) : "memory");
            x = c.x; y = c.y; z = c.z;
            slotv = __float_as_uint(x + y + z);
        }
        if (KIND == K_TAIL || KIND == K_ROUND) {
            u64 best = ((u64)bhi << 32) | blo;
            if ((tid & 63) == 0) asm volatile("ds_max_u64 %0, %1" : : "v"(addr), "v"(best) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" : : : "memory");
            u64 g;
            asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(g) : "v"(addr) : "memory");
            u32 r = ~(u32)g;
            r = (r & 4095u) << 4;
            typedef float f3 __attribute__((ext_vector_type(3)));
            f3 c;
            asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(r) : "memory");
            o = (f2){c.x + c.y, c.z};
            if (tid == 0) { lds[4096 * 4] = 0.f; lds[4096 * 4 + 1] = 0.f; }
        }
    }
    const u64 t1 = now();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    for (int i = 0; i < 16; ++i) s += tmin[i];
    s += o.x + o.y + (float)blo + (float)bhi + (float)slotv;
    if ((tid & 63) == 0) out[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
    if (s == 123.456f) out[0] = 0;
}

template <int KIND>
void run(int nt, u64 *dout, int blocks)
{
    const int iters = 2000;
    const size_t lds = (4096 * 4 + 64) * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    probe<KIND><<<blocks, nt, lds>>>(100, dout, 1.25f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<KIND><<<blocks, nt, lds>>>(iters, dout, 1.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    u64 h[16 * 32];
    hipMemcpy(h, dout, sizeof(u64) * 16 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0; int nw = nt / 64;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < nw; ++w) cyc += (double)h[b * 16 + w];
    cyc /= blocks * nw;
    printf("  %4d thr  %8.1f cyc/iter  %6.2f cyc/instr  wall %7.1f ns/iter (clock %.2f GHz)   %s\n", nt, cyc / iters, cyc / iters / INSTR[KIND], ms * 1e6 / iters,
           cyc / (ms * 1e6), NAMES[KIND]);
}

int main()
{
    u64 *dout; hipMalloc(&dout, sizeof(u64) * 16 * 64);
    for (int nt : {256, 512}) {
        printf("%d threads per workgroup (%d wave(s) per SIMD), 32 workgroups:\n", nt, nt / 256);
        run<K_ADD_IND>(nt, dout, 32); run<K_ADD_DEP>(nt, dout, 32); run<K_PKADD_IND>(nt, dout, 32); run<K_PKMUL_DEP>(nt, dout, 32); run<K_MIN_IND>(nt, dout, 32);
        run<K_PKMIX>(nt, dout, 32); run<K_KEY64>(nt, dout, 32); run<K_DPP>(nt, dout, 32); run<K_ATOMIC>(nt, dout, 32); run<K_BARRIER>(nt, dout, 32);
        run<K_LDS64>(nt, dout, 32); run<K_LDS96>(nt, dout, 32); run<K_TAIL>(nt, dout, 32); run<K_ROUND>(nt, dout, 32);
    }
    return 0;
}
