// Micro-benchmark: issue cost, in SHADER CYCLES per wave64 instruction per SIMD, of the VALU instructions the svmc
// stepping kernels are made of -- measured in the kernel with s_memtime (tick = shader cycle, MI355X_MICROARCH.md) and
// s_memrealtime (100 MHz, chip-wide), so the answer does not depend on what the clock does during the run.
//
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates > profiles/rNN_valu_rates.txt
//
// Every test is 256-thread blocks (4 waves = one per SIMD) with the dynamic LDS sized so that exactly W blocks fit a CU:
// W waves per SIMD, W in {1, 2, 4, 8}.  A wave issues ITERS x 16 instructions of the tested stream on 8 independent
// accumulators (dependency distance 8 instructions; the 16 sit in ONE asm block, so the compiler puts nothing between
// them), brackets them with s_memtime and s_memrealtime, and lane 0 stores both.  Readings per test:
//   cyc(span) = (last wave's end - first wave's start on the chip-wide 100 MHz counter) x the shader clock the same
//               waves measured (sum of s_memtime ticks / sum of s_memrealtime ticks) / (ITERS * 16 * W): the SIMD's cost
//               per instruction whatever the arbitration between its waves does -- THE number;
//   cyc(wave) = the slowest wave's own s_memtime ticks / (ITERS * 16 * W).
// The SIMD's identity (HW_ID: se, sh, cu, simd + XCC_ID) is recorded and the number of waves per SIMD is checked.
// CHAIN variants run the 16 instructions on ONE accumulator (latency).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#define ITERS 2048
#define HW_REG_HW_ID 4
#define HW_REG_XCC_ID 20

struct Rec {
    uint64_t ticks, real, real_start, real_end;
    uint32_t hw_id, xcc;
};

__device__ __forceinline__ void stamp(Rec *out, uint64_t t0, uint64_t r0)
{
    const uint64_t t1 = __builtin_readcyclecounter();      // s_memtime
    const uint64_t r1 = wall_clock64();                    // s_memrealtime, 100 MHz
    if ((threadIdx.x & 63) == 0) {
        Rec r;
        r.ticks = t1 - t0;
        r.real = r1 - r0;
        r.real_start = r0;
        r.real_end = r1;
        r.hw_id = __builtin_amdgcn_s_getreg((HW_REG_HW_ID) | (0 << 6) | (31 << 11));
        r.xcc = __builtin_amdgcn_s_getreg((HW_REG_XCC_ID) | (0 << 6) | (31 << 11));
        out[blockIdx.x * 4 + (threadIdx.x >> 6)] = r;
    }
}

__global__ __launch_bounds__(256) void k_v_fma_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_fma_f64 %6, %6, %8, %9\n\t"
                     "v_fma_f64 %7, %7, %8, %9\n\t"
                     "v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_fma_f64 %6, %6, %8, %9\n\t"
                     "v_fma_f64 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_mul_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_mul_f64 %0, %0, %8\n\t"
                     "v_mul_f64 %1, %1, %8\n\t"
                     "v_mul_f64 %2, %2, %8\n\t"
                     "v_mul_f64 %3, %3, %8\n\t"
                     "v_mul_f64 %4, %4, %8\n\t"
                     "v_mul_f64 %5, %5, %8\n\t"
                     "v_mul_f64 %6, %6, %8\n\t"
                     "v_mul_f64 %7, %7, %8\n\t"
                     "v_mul_f64 %0, %0, %8\n\t"
                     "v_mul_f64 %1, %1, %8\n\t"
                     "v_mul_f64 %2, %2, %8\n\t"
                     "v_mul_f64 %3, %3, %8\n\t"
                     "v_mul_f64 %4, %4, %8\n\t"
                     "v_mul_f64 %5, %5, %8\n\t"
                     "v_mul_f64 %6, %6, %8\n\t"
                     "v_mul_f64 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_add_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_add_f64 %0, %0, %8\n\t"
                     "v_add_f64 %1, %1, %8\n\t"
                     "v_add_f64 %2, %2, %8\n\t"
                     "v_add_f64 %3, %3, %8\n\t"
                     "v_add_f64 %4, %4, %8\n\t"
                     "v_add_f64 %5, %5, %8\n\t"
                     "v_add_f64 %6, %6, %8\n\t"
                     "v_add_f64 %7, %7, %8\n\t"
                     "v_add_f64 %0, %0, %8\n\t"
                     "v_add_f64 %1, %1, %8\n\t"
                     "v_add_f64 %2, %2, %8\n\t"
                     "v_add_f64 %3, %3, %8\n\t"
                     "v_add_f64 %4, %4, %8\n\t"
                     "v_add_f64 %5, %5, %8\n\t"
                     "v_add_f64 %6, %6, %8\n\t"
                     "v_add_f64 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_max_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_max_f64 %0, %0, %8\n\t"
                     "v_max_f64 %1, %1, %8\n\t"
                     "v_max_f64 %2, %2, %8\n\t"
                     "v_max_f64 %3, %3, %8\n\t"
                     "v_max_f64 %4, %4, %8\n\t"
                     "v_max_f64 %5, %5, %8\n\t"
                     "v_max_f64 %6, %6, %8\n\t"
                     "v_max_f64 %7, %7, %8\n\t"
                     "v_max_f64 %0, %0, %8\n\t"
                     "v_max_f64 %1, %1, %8\n\t"
                     "v_max_f64 %2, %2, %8\n\t"
                     "v_max_f64 %3, %3, %8\n\t"
                     "v_max_f64 %4, %4, %8\n\t"
                     "v_max_f64 %5, %5, %8\n\t"
                     "v_max_f64 %6, %6, %8\n\t"
                     "v_max_f64 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_ldexp_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_ldexp_f64 %0, %0, 1\n\t"
                     "v_ldexp_f64 %1, %1, 1\n\t"
                     "v_ldexp_f64 %2, %2, 1\n\t"
                     "v_ldexp_f64 %3, %3, 1\n\t"
                     "v_ldexp_f64 %4, %4, 1\n\t"
                     "v_ldexp_f64 %5, %5, 1\n\t"
                     "v_ldexp_f64 %6, %6, 1\n\t"
                     "v_ldexp_f64 %7, %7, 1\n\t"
                     "v_ldexp_f64 %0, %0, 1\n\t"
                     "v_ldexp_f64 %1, %1, 1\n\t"
                     "v_ldexp_f64 %2, %2, 1\n\t"
                     "v_ldexp_f64 %3, %3, 1\n\t"
                     "v_ldexp_f64 %4, %4, 1\n\t"
                     "v_ldexp_f64 %5, %5, 1\n\t"
                     "v_ldexp_f64 %6, %6, 1\n\t"
                     "v_ldexp_f64 %7, %7, 1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_rndne_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_rndne_f64 %0, %0\n\t"
                     "v_rndne_f64 %1, %1\n\t"
                     "v_rndne_f64 %2, %2\n\t"
                     "v_rndne_f64 %3, %3\n\t"
                     "v_rndne_f64 %4, %4\n\t"
                     "v_rndne_f64 %5, %5\n\t"
                     "v_rndne_f64 %6, %6\n\t"
                     "v_rndne_f64 %7, %7\n\t"
                     "v_rndne_f64 %0, %0\n\t"
                     "v_rndne_f64 %1, %1\n\t"
                     "v_rndne_f64 %2, %2\n\t"
                     "v_rndne_f64 %3, %3\n\t"
                     "v_rndne_f64 %4, %4\n\t"
                     "v_rndne_f64 %5, %5\n\t"
                     "v_rndne_f64 %6, %6\n\t"
                     "v_rndne_f64 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_mov_b64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_mov_b64 %0, %8\n\t"
                     "v_mov_b64 %1, %8\n\t"
                     "v_mov_b64 %2, %8\n\t"
                     "v_mov_b64 %3, %8\n\t"
                     "v_mov_b64 %4, %8\n\t"
                     "v_mov_b64 %5, %8\n\t"
                     "v_mov_b64 %6, %8\n\t"
                     "v_mov_b64 %7, %8\n\t"
                     "v_mov_b64 %0, %8\n\t"
                     "v_mov_b64 %1, %8\n\t"
                     "v_mov_b64 %2, %8\n\t"
                     "v_mov_b64 %3, %8\n\t"
                     "v_mov_b64 %4, %8\n\t"
                     "v_mov_b64 %5, %8\n\t"
                     "v_mov_b64 %6, %8\n\t"
                     "v_mov_b64 %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_cmp_gt_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cmp_gt_f64 vcc, %0, %8\n\t"
                     "v_cmp_gt_f64 vcc, %1, %8\n\t"
                     "v_cmp_gt_f64 vcc, %2, %8\n\t"
                     "v_cmp_gt_f64 vcc, %3, %8\n\t"
                     "v_cmp_gt_f64 vcc, %4, %8\n\t"
                     "v_cmp_gt_f64 vcc, %5, %8\n\t"
                     "v_cmp_gt_f64 vcc, %6, %8\n\t"
                     "v_cmp_gt_f64 vcc, %7, %8\n\t"
                     "v_cmp_gt_f64 vcc, %0, %8\n\t"
                     "v_cmp_gt_f64 vcc, %1, %8\n\t"
                     "v_cmp_gt_f64 vcc, %2, %8\n\t"
                     "v_cmp_gt_f64 vcc, %3, %8\n\t"
                     "v_cmp_gt_f64 vcc, %4, %8\n\t"
                     "v_cmp_gt_f64 vcc, %5, %8\n\t"
                     "v_cmp_gt_f64 vcc, %6, %8\n\t"
                     "v_cmp_gt_f64 vcc, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_rcp_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %1, %1\n\t"
                     "v_rcp_f64 %2, %2\n\t"
                     "v_rcp_f64 %3, %3\n\t"
                     "v_rcp_f64 %4, %4\n\t"
                     "v_rcp_f64 %5, %5\n\t"
                     "v_rcp_f64 %6, %6\n\t"
                     "v_rcp_f64 %7, %7\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %1, %1\n\t"
                     "v_rcp_f64 %2, %2\n\t"
                     "v_rcp_f64 %3, %3\n\t"
                     "v_rcp_f64 %4, %4\n\t"
                     "v_rcp_f64 %5, %5\n\t"
                     "v_rcp_f64 %6, %6\n\t"
                     "v_rcp_f64 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_rsq_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_rsq_f64 %0, %0\n\t"
                     "v_rsq_f64 %1, %1\n\t"
                     "v_rsq_f64 %2, %2\n\t"
                     "v_rsq_f64 %3, %3\n\t"
                     "v_rsq_f64 %4, %4\n\t"
                     "v_rsq_f64 %5, %5\n\t"
                     "v_rsq_f64 %6, %6\n\t"
                     "v_rsq_f64 %7, %7\n\t"
                     "v_rsq_f64 %0, %0\n\t"
                     "v_rsq_f64 %1, %1\n\t"
                     "v_rsq_f64 %2, %2\n\t"
                     "v_rsq_f64 %3, %3\n\t"
                     "v_rsq_f64 %4, %4\n\t"
                     "v_rsq_f64 %5, %5\n\t"
                     "v_rsq_f64 %6, %6\n\t"
                     "v_rsq_f64 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_sqrt_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_sqrt_f64 %0, %0\n\t"
                     "v_sqrt_f64 %1, %1\n\t"
                     "v_sqrt_f64 %2, %2\n\t"
                     "v_sqrt_f64 %3, %3\n\t"
                     "v_sqrt_f64 %4, %4\n\t"
                     "v_sqrt_f64 %5, %5\n\t"
                     "v_sqrt_f64 %6, %6\n\t"
                     "v_sqrt_f64 %7, %7\n\t"
                     "v_sqrt_f64 %0, %0\n\t"
                     "v_sqrt_f64 %1, %1\n\t"
                     "v_sqrt_f64 %2, %2\n\t"
                     "v_sqrt_f64 %3, %3\n\t"
                     "v_sqrt_f64 %4, %4\n\t"
                     "v_sqrt_f64 %5, %5\n\t"
                     "v_sqrt_f64 %6, %6\n\t"
                     "v_sqrt_f64 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_frexp_mant_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_frexp_mant_f64 %0, %0\n\t"
                     "v_frexp_mant_f64 %1, %1\n\t"
                     "v_frexp_mant_f64 %2, %2\n\t"
                     "v_frexp_mant_f64 %3, %3\n\t"
                     "v_frexp_mant_f64 %4, %4\n\t"
                     "v_frexp_mant_f64 %5, %5\n\t"
                     "v_frexp_mant_f64 %6, %6\n\t"
                     "v_frexp_mant_f64 %7, %7\n\t"
                     "v_frexp_mant_f64 %0, %0\n\t"
                     "v_frexp_mant_f64 %1, %1\n\t"
                     "v_frexp_mant_f64 %2, %2\n\t"
                     "v_frexp_mant_f64 %3, %3\n\t"
                     "v_frexp_mant_f64 %4, %4\n\t"
                     "v_frexp_mant_f64 %5, %5\n\t"
                     "v_frexp_mant_f64 %6, %6\n\t"
                     "v_frexp_mant_f64 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_mad_u64_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint64_t a0 = (uint64_t)(seed + threadIdx.x) + (uint64_t)0; uint64_t a1 = (uint64_t)(seed + threadIdx.x) + (uint64_t)1; uint64_t a2 = (uint64_t)(seed + threadIdx.x) + (uint64_t)2; uint64_t a3 = (uint64_t)(seed + threadIdx.x) + (uint64_t)3; uint64_t a4 = (uint64_t)(seed + threadIdx.x) + (uint64_t)4; uint64_t a5 = (uint64_t)(seed + threadIdx.x) + (uint64_t)5; uint64_t a6 = (uint64_t)(seed + threadIdx.x) + (uint64_t)6; uint64_t a7 = (uint64_t)(seed + threadIdx.x) + (uint64_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
                     "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
                     "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\t"
                     "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\t"
                     "v_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                     "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\t"
                     "v_mad_u64_u32 %7, vcc, %8, %9, %7\n\t"
                     "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
                     "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
                     "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\t"
                     "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                     "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\t"
                     "v_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                     "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\t"
                     "v_mad_u64_u32 %7, vcc, %8, %9, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_lshrrev_b64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint64_t a0 = (uint64_t)(seed + threadIdx.x) + (uint64_t)0; uint64_t a1 = (uint64_t)(seed + threadIdx.x) + (uint64_t)1; uint64_t a2 = (uint64_t)(seed + threadIdx.x) + (uint64_t)2; uint64_t a3 = (uint64_t)(seed + threadIdx.x) + (uint64_t)3; uint64_t a4 = (uint64_t)(seed + threadIdx.x) + (uint64_t)4; uint64_t a5 = (uint64_t)(seed + threadIdx.x) + (uint64_t)5; uint64_t a6 = (uint64_t)(seed + threadIdx.x) + (uint64_t)6; uint64_t a7 = (uint64_t)(seed + threadIdx.x) + (uint64_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_lshrrev_b64 %0, 3, %0\n\t"
                     "v_lshrrev_b64 %1, 3, %1\n\t"
                     "v_lshrrev_b64 %2, 3, %2\n\t"
                     "v_lshrrev_b64 %3, 3, %3\n\t"
                     "v_lshrrev_b64 %4, 3, %4\n\t"
                     "v_lshrrev_b64 %5, 3, %5\n\t"
                     "v_lshrrev_b64 %6, 3, %6\n\t"
                     "v_lshrrev_b64 %7, 3, %7\n\t"
                     "v_lshrrev_b64 %0, 3, %0\n\t"
                     "v_lshrrev_b64 %1, 3, %1\n\t"
                     "v_lshrrev_b64 %2, 3, %2\n\t"
                     "v_lshrrev_b64 %3, 3, %3\n\t"
                     "v_lshrrev_b64 %4, 3, %4\n\t"
                     "v_lshrrev_b64 %5, 3, %5\n\t"
                     "v_lshrrev_b64 %6, 3, %6\n\t"
                     "v_lshrrev_b64 %7, 3, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_lshlrev_b64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint64_t a0 = (uint64_t)(seed + threadIdx.x) + (uint64_t)0; uint64_t a1 = (uint64_t)(seed + threadIdx.x) + (uint64_t)1; uint64_t a2 = (uint64_t)(seed + threadIdx.x) + (uint64_t)2; uint64_t a3 = (uint64_t)(seed + threadIdx.x) + (uint64_t)3; uint64_t a4 = (uint64_t)(seed + threadIdx.x) + (uint64_t)4; uint64_t a5 = (uint64_t)(seed + threadIdx.x) + (uint64_t)5; uint64_t a6 = (uint64_t)(seed + threadIdx.x) + (uint64_t)6; uint64_t a7 = (uint64_t)(seed + threadIdx.x) + (uint64_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_lshlrev_b64 %0, 3, %0\n\t"
                     "v_lshlrev_b64 %1, 3, %1\n\t"
                     "v_lshlrev_b64 %2, 3, %2\n\t"
                     "v_lshlrev_b64 %3, 3, %3\n\t"
                     "v_lshlrev_b64 %4, 3, %4\n\t"
                     "v_lshlrev_b64 %5, 3, %5\n\t"
                     "v_lshlrev_b64 %6, 3, %6\n\t"
                     "v_lshlrev_b64 %7, 3, %7\n\t"
                     "v_lshlrev_b64 %0, 3, %0\n\t"
                     "v_lshlrev_b64 %1, 3, %1\n\t"
                     "v_lshlrev_b64 %2, 3, %2\n\t"
                     "v_lshlrev_b64 %3, 3, %3\n\t"
                     "v_lshlrev_b64 %4, 3, %4\n\t"
                     "v_lshlrev_b64 %5, 3, %5\n\t"
                     "v_lshlrev_b64 %6, 3, %6\n\t"
                     "v_lshlrev_b64 %7, 3, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_mul_lo_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_mul_lo_u32 %0, %0, %8\n\t"
                     "v_mul_lo_u32 %1, %1, %8\n\t"
                     "v_mul_lo_u32 %2, %2, %8\n\t"
                     "v_mul_lo_u32 %3, %3, %8\n\t"
                     "v_mul_lo_u32 %4, %4, %8\n\t"
                     "v_mul_lo_u32 %5, %5, %8\n\t"
                     "v_mul_lo_u32 %6, %6, %8\n\t"
                     "v_mul_lo_u32 %7, %7, %8\n\t"
                     "v_mul_lo_u32 %0, %0, %8\n\t"
                     "v_mul_lo_u32 %1, %1, %8\n\t"
                     "v_mul_lo_u32 %2, %2, %8\n\t"
                     "v_mul_lo_u32 %3, %3, %8\n\t"
                     "v_mul_lo_u32 %4, %4, %8\n\t"
                     "v_mul_lo_u32 %5, %5, %8\n\t"
                     "v_mul_lo_u32 %6, %6, %8\n\t"
                     "v_mul_lo_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_mul_hi_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_mul_hi_u32 %0, %0, %8\n\t"
                     "v_mul_hi_u32 %1, %1, %8\n\t"
                     "v_mul_hi_u32 %2, %2, %8\n\t"
                     "v_mul_hi_u32 %3, %3, %8\n\t"
                     "v_mul_hi_u32 %4, %4, %8\n\t"
                     "v_mul_hi_u32 %5, %5, %8\n\t"
                     "v_mul_hi_u32 %6, %6, %8\n\t"
                     "v_mul_hi_u32 %7, %7, %8\n\t"
                     "v_mul_hi_u32 %0, %0, %8\n\t"
                     "v_mul_hi_u32 %1, %1, %8\n\t"
                     "v_mul_hi_u32 %2, %2, %8\n\t"
                     "v_mul_hi_u32 %3, %3, %8\n\t"
                     "v_mul_hi_u32 %4, %4, %8\n\t"
                     "v_mul_hi_u32 %5, %5, %8\n\t"
                     "v_mul_hi_u32 %6, %6, %8\n\t"
                     "v_mul_hi_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_mul_u32_u24(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_mul_u32_u24 %0, %0, %8\n\t"
                     "v_mul_u32_u24 %1, %1, %8\n\t"
                     "v_mul_u32_u24 %2, %2, %8\n\t"
                     "v_mul_u32_u24 %3, %3, %8\n\t"
                     "v_mul_u32_u24 %4, %4, %8\n\t"
                     "v_mul_u32_u24 %5, %5, %8\n\t"
                     "v_mul_u32_u24 %6, %6, %8\n\t"
                     "v_mul_u32_u24 %7, %7, %8\n\t"
                     "v_mul_u32_u24 %0, %0, %8\n\t"
                     "v_mul_u32_u24 %1, %1, %8\n\t"
                     "v_mul_u32_u24 %2, %2, %8\n\t"
                     "v_mul_u32_u24 %3, %3, %8\n\t"
                     "v_mul_u32_u24 %4, %4, %8\n\t"
                     "v_mul_u32_u24 %5, %5, %8\n\t"
                     "v_mul_u32_u24 %6, %6, %8\n\t"
                     "v_mul_u32_u24 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_mad_u32_u24(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n\t"
                     "v_mad_u32_u24 %1, %1, %8, %9\n\t"
                     "v_mad_u32_u24 %2, %2, %8, %9\n\t"
                     "v_mad_u32_u24 %3, %3, %8, %9\n\t"
                     "v_mad_u32_u24 %4, %4, %8, %9\n\t"
                     "v_mad_u32_u24 %5, %5, %8, %9\n\t"
                     "v_mad_u32_u24 %6, %6, %8, %9\n\t"
                     "v_mad_u32_u24 %7, %7, %8, %9\n\t"
                     "v_mad_u32_u24 %0, %0, %8, %9\n\t"
                     "v_mad_u32_u24 %1, %1, %8, %9\n\t"
                     "v_mad_u32_u24 %2, %2, %8, %9\n\t"
                     "v_mad_u32_u24 %3, %3, %8, %9\n\t"
                     "v_mad_u32_u24 %4, %4, %8, %9\n\t"
                     "v_mad_u32_u24 %5, %5, %8, %9\n\t"
                     "v_mad_u32_u24 %6, %6, %8, %9\n\t"
                     "v_mad_u32_u24 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_bitop3_b32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %7, %7, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\n\t"
                     "v_bitop3_b32 %7, %7, %8, %9 bitop3:0x96"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_xor_b32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_xor_b32 %0, %0, %8\n\t"
                     "v_xor_b32 %1, %1, %8\n\t"
                     "v_xor_b32 %2, %2, %8\n\t"
                     "v_xor_b32 %3, %3, %8\n\t"
                     "v_xor_b32 %4, %4, %8\n\t"
                     "v_xor_b32 %5, %5, %8\n\t"
                     "v_xor_b32 %6, %6, %8\n\t"
                     "v_xor_b32 %7, %7, %8\n\t"
                     "v_xor_b32 %0, %0, %8\n\t"
                     "v_xor_b32 %1, %1, %8\n\t"
                     "v_xor_b32 %2, %2, %8\n\t"
                     "v_xor_b32 %3, %3, %8\n\t"
                     "v_xor_b32 %4, %4, %8\n\t"
                     "v_xor_b32 %5, %5, %8\n\t"
                     "v_xor_b32 %6, %6, %8\n\t"
                     "v_xor_b32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_and_b32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_and_b32 %0, %0, %8\n\t"
                     "v_and_b32 %1, %1, %8\n\t"
                     "v_and_b32 %2, %2, %8\n\t"
                     "v_and_b32 %3, %3, %8\n\t"
                     "v_and_b32 %4, %4, %8\n\t"
                     "v_and_b32 %5, %5, %8\n\t"
                     "v_and_b32 %6, %6, %8\n\t"
                     "v_and_b32 %7, %7, %8\n\t"
                     "v_and_b32 %0, %0, %8\n\t"
                     "v_and_b32 %1, %1, %8\n\t"
                     "v_and_b32 %2, %2, %8\n\t"
                     "v_and_b32 %3, %3, %8\n\t"
                     "v_and_b32 %4, %4, %8\n\t"
                     "v_and_b32 %5, %5, %8\n\t"
                     "v_and_b32 %6, %6, %8\n\t"
                     "v_and_b32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_add_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_add_u32 %0, %0, %8\n\t"
                     "v_add_u32 %1, %1, %8\n\t"
                     "v_add_u32 %2, %2, %8\n\t"
                     "v_add_u32 %3, %3, %8\n\t"
                     "v_add_u32 %4, %4, %8\n\t"
                     "v_add_u32 %5, %5, %8\n\t"
                     "v_add_u32 %6, %6, %8\n\t"
                     "v_add_u32 %7, %7, %8\n\t"
                     "v_add_u32 %0, %0, %8\n\t"
                     "v_add_u32 %1, %1, %8\n\t"
                     "v_add_u32 %2, %2, %8\n\t"
                     "v_add_u32 %3, %3, %8\n\t"
                     "v_add_u32 %4, %4, %8\n\t"
                     "v_add_u32 %5, %5, %8\n\t"
                     "v_add_u32 %6, %6, %8\n\t"
                     "v_add_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_add3_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_add3_u32 %0, %0, %8, %9\n\t"
                     "v_add3_u32 %1, %1, %8, %9\n\t"
                     "v_add3_u32 %2, %2, %8, %9\n\t"
                     "v_add3_u32 %3, %3, %8, %9\n\t"
                     "v_add3_u32 %4, %4, %8, %9\n\t"
                     "v_add3_u32 %5, %5, %8, %9\n\t"
                     "v_add3_u32 %6, %6, %8, %9\n\t"
                     "v_add3_u32 %7, %7, %8, %9\n\t"
                     "v_add3_u32 %0, %0, %8, %9\n\t"
                     "v_add3_u32 %1, %1, %8, %9\n\t"
                     "v_add3_u32 %2, %2, %8, %9\n\t"
                     "v_add3_u32 %3, %3, %8, %9\n\t"
                     "v_add3_u32 %4, %4, %8, %9\n\t"
                     "v_add3_u32 %5, %5, %8, %9\n\t"
                     "v_add3_u32 %6, %6, %8, %9\n\t"
                     "v_add3_u32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_add_co_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_add_co_u32 %0, vcc, %0, %8\n\t"
                     "v_add_co_u32 %1, vcc, %1, %8\n\t"
                     "v_add_co_u32 %2, vcc, %2, %8\n\t"
                     "v_add_co_u32 %3, vcc, %3, %8\n\t"
                     "v_add_co_u32 %4, vcc, %4, %8\n\t"
                     "v_add_co_u32 %5, vcc, %5, %8\n\t"
                     "v_add_co_u32 %6, vcc, %6, %8\n\t"
                     "v_add_co_u32 %7, vcc, %7, %8\n\t"
                     "v_add_co_u32 %0, vcc, %0, %8\n\t"
                     "v_add_co_u32 %1, vcc, %1, %8\n\t"
                     "v_add_co_u32 %2, vcc, %2, %8\n\t"
                     "v_add_co_u32 %3, vcc, %3, %8\n\t"
                     "v_add_co_u32 %4, vcc, %4, %8\n\t"
                     "v_add_co_u32 %5, vcc, %5, %8\n\t"
                     "v_add_co_u32 %6, vcc, %6, %8\n\t"
                     "v_add_co_u32 %7, vcc, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_lshrrev_b32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_lshrrev_b32 %0, 3, %0\n\t"
                     "v_lshrrev_b32 %1, 3, %1\n\t"
                     "v_lshrrev_b32 %2, 3, %2\n\t"
                     "v_lshrrev_b32 %3, 3, %3\n\t"
                     "v_lshrrev_b32 %4, 3, %4\n\t"
                     "v_lshrrev_b32 %5, 3, %5\n\t"
                     "v_lshrrev_b32 %6, 3, %6\n\t"
                     "v_lshrrev_b32 %7, 3, %7\n\t"
                     "v_lshrrev_b32 %0, 3, %0\n\t"
                     "v_lshrrev_b32 %1, 3, %1\n\t"
                     "v_lshrrev_b32 %2, 3, %2\n\t"
                     "v_lshrrev_b32 %3, 3, %3\n\t"
                     "v_lshrrev_b32 %4, 3, %4\n\t"
                     "v_lshrrev_b32 %5, 3, %5\n\t"
                     "v_lshrrev_b32 %6, 3, %6\n\t"
                     "v_lshrrev_b32 %7, 3, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_lshl_add_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_lshl_add_u32 %0, %0, 3, %8\n\t"
                     "v_lshl_add_u32 %1, %1, 3, %8\n\t"
                     "v_lshl_add_u32 %2, %2, 3, %8\n\t"
                     "v_lshl_add_u32 %3, %3, 3, %8\n\t"
                     "v_lshl_add_u32 %4, %4, 3, %8\n\t"
                     "v_lshl_add_u32 %5, %5, 3, %8\n\t"
                     "v_lshl_add_u32 %6, %6, 3, %8\n\t"
                     "v_lshl_add_u32 %7, %7, 3, %8\n\t"
                     "v_lshl_add_u32 %0, %0, 3, %8\n\t"
                     "v_lshl_add_u32 %1, %1, 3, %8\n\t"
                     "v_lshl_add_u32 %2, %2, 3, %8\n\t"
                     "v_lshl_add_u32 %3, %3, 3, %8\n\t"
                     "v_lshl_add_u32 %4, %4, 3, %8\n\t"
                     "v_lshl_add_u32 %5, %5, 3, %8\n\t"
                     "v_lshl_add_u32 %6, %6, 3, %8\n\t"
                     "v_lshl_add_u32 %7, %7, 3, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_lshl_or_b32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_lshl_or_b32 %0, %0, 3, %8\n\t"
                     "v_lshl_or_b32 %1, %1, 3, %8\n\t"
                     "v_lshl_or_b32 %2, %2, 3, %8\n\t"
                     "v_lshl_or_b32 %3, %3, 3, %8\n\t"
                     "v_lshl_or_b32 %4, %4, 3, %8\n\t"
                     "v_lshl_or_b32 %5, %5, 3, %8\n\t"
                     "v_lshl_or_b32 %6, %6, 3, %8\n\t"
                     "v_lshl_or_b32 %7, %7, 3, %8\n\t"
                     "v_lshl_or_b32 %0, %0, 3, %8\n\t"
                     "v_lshl_or_b32 %1, %1, 3, %8\n\t"
                     "v_lshl_or_b32 %2, %2, 3, %8\n\t"
                     "v_lshl_or_b32 %3, %3, 3, %8\n\t"
                     "v_lshl_or_b32 %4, %4, 3, %8\n\t"
                     "v_lshl_or_b32 %5, %5, 3, %8\n\t"
                     "v_lshl_or_b32 %6, %6, 3, %8\n\t"
                     "v_lshl_or_b32 %7, %7, 3, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_alignbit_b32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_alignbit_b32 %0, %0, %8, 7\n\t"
                     "v_alignbit_b32 %1, %1, %8, 7\n\t"
                     "v_alignbit_b32 %2, %2, %8, 7\n\t"
                     "v_alignbit_b32 %3, %3, %8, 7\n\t"
                     "v_alignbit_b32 %4, %4, %8, 7\n\t"
                     "v_alignbit_b32 %5, %5, %8, 7\n\t"
                     "v_alignbit_b32 %6, %6, %8, 7\n\t"
                     "v_alignbit_b32 %7, %7, %8, 7\n\t"
                     "v_alignbit_b32 %0, %0, %8, 7\n\t"
                     "v_alignbit_b32 %1, %1, %8, 7\n\t"
                     "v_alignbit_b32 %2, %2, %8, 7\n\t"
                     "v_alignbit_b32 %3, %3, %8, 7\n\t"
                     "v_alignbit_b32 %4, %4, %8, 7\n\t"
                     "v_alignbit_b32 %5, %5, %8, 7\n\t"
                     "v_alignbit_b32 %6, %6, %8, 7\n\t"
                     "v_alignbit_b32 %7, %7, %8, 7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_bfe_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_bfe_u32 %0, %0, 3, 9\n\t"
                     "v_bfe_u32 %1, %1, 3, 9\n\t"
                     "v_bfe_u32 %2, %2, 3, 9\n\t"
                     "v_bfe_u32 %3, %3, 3, 9\n\t"
                     "v_bfe_u32 %4, %4, 3, 9\n\t"
                     "v_bfe_u32 %5, %5, 3, 9\n\t"
                     "v_bfe_u32 %6, %6, 3, 9\n\t"
                     "v_bfe_u32 %7, %7, 3, 9\n\t"
                     "v_bfe_u32 %0, %0, 3, 9\n\t"
                     "v_bfe_u32 %1, %1, 3, 9\n\t"
                     "v_bfe_u32 %2, %2, 3, 9\n\t"
                     "v_bfe_u32 %3, %3, 3, 9\n\t"
                     "v_bfe_u32 %4, %4, 3, 9\n\t"
                     "v_bfe_u32 %5, %5, 3, 9\n\t"
                     "v_bfe_u32 %6, %6, 3, 9\n\t"
                     "v_bfe_u32 %7, %7, 3, 9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_perm_b32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_perm_b32 %0, %0, %8, %9\n\t"
                     "v_perm_b32 %1, %1, %8, %9\n\t"
                     "v_perm_b32 %2, %2, %8, %9\n\t"
                     "v_perm_b32 %3, %3, %8, %9\n\t"
                     "v_perm_b32 %4, %4, %8, %9\n\t"
                     "v_perm_b32 %5, %5, %8, %9\n\t"
                     "v_perm_b32 %6, %6, %8, %9\n\t"
                     "v_perm_b32 %7, %7, %8, %9\n\t"
                     "v_perm_b32 %0, %0, %8, %9\n\t"
                     "v_perm_b32 %1, %1, %8, %9\n\t"
                     "v_perm_b32 %2, %2, %8, %9\n\t"
                     "v_perm_b32 %3, %3, %8, %9\n\t"
                     "v_perm_b32 %4, %4, %8, %9\n\t"
                     "v_perm_b32 %5, %5, %8, %9\n\t"
                     "v_perm_b32 %6, %6, %8, %9\n\t"
                     "v_perm_b32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_and_or_b32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_and_or_b32 %0, %0, %8, %9\n\t"
                     "v_and_or_b32 %1, %1, %8, %9\n\t"
                     "v_and_or_b32 %2, %2, %8, %9\n\t"
                     "v_and_or_b32 %3, %3, %8, %9\n\t"
                     "v_and_or_b32 %4, %4, %8, %9\n\t"
                     "v_and_or_b32 %5, %5, %8, %9\n\t"
                     "v_and_or_b32 %6, %6, %8, %9\n\t"
                     "v_and_or_b32 %7, %7, %8, %9\n\t"
                     "v_and_or_b32 %0, %0, %8, %9\n\t"
                     "v_and_or_b32 %1, %1, %8, %9\n\t"
                     "v_and_or_b32 %2, %2, %8, %9\n\t"
                     "v_and_or_b32 %3, %3, %8, %9\n\t"
                     "v_and_or_b32 %4, %4, %8, %9\n\t"
                     "v_and_or_b32 %5, %5, %8, %9\n\t"
                     "v_and_or_b32 %6, %6, %8, %9\n\t"
                     "v_and_or_b32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_xad_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_xad_u32 %0, %0, %8, %9\n\t"
                     "v_xad_u32 %1, %1, %8, %9\n\t"
                     "v_xad_u32 %2, %2, %8, %9\n\t"
                     "v_xad_u32 %3, %3, %8, %9\n\t"
                     "v_xad_u32 %4, %4, %8, %9\n\t"
                     "v_xad_u32 %5, %5, %8, %9\n\t"
                     "v_xad_u32 %6, %6, %8, %9\n\t"
                     "v_xad_u32 %7, %7, %8, %9\n\t"
                     "v_xad_u32 %0, %0, %8, %9\n\t"
                     "v_xad_u32 %1, %1, %8, %9\n\t"
                     "v_xad_u32 %2, %2, %8, %9\n\t"
                     "v_xad_u32 %3, %3, %8, %9\n\t"
                     "v_xad_u32 %4, %4, %8, %9\n\t"
                     "v_xad_u32 %5, %5, %8, %9\n\t"
                     "v_xad_u32 %6, %6, %8, %9\n\t"
                     "v_xad_u32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_mov_b32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_mov_b32 %0, %8\n\t"
                     "v_mov_b32 %1, %8\n\t"
                     "v_mov_b32 %2, %8\n\t"
                     "v_mov_b32 %3, %8\n\t"
                     "v_mov_b32 %4, %8\n\t"
                     "v_mov_b32 %5, %8\n\t"
                     "v_mov_b32 %6, %8\n\t"
                     "v_mov_b32 %7, %8\n\t"
                     "v_mov_b32 %0, %8\n\t"
                     "v_mov_b32 %1, %8\n\t"
                     "v_mov_b32 %2, %8\n\t"
                     "v_mov_b32 %3, %8\n\t"
                     "v_mov_b32 %4, %8\n\t"
                     "v_mov_b32 %5, %8\n\t"
                     "v_mov_b32 %6, %8\n\t"
                     "v_mov_b32 %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_cndmask_b32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\t"
                     "v_cndmask_b32 %1, %1, %8, vcc\n\t"
                     "v_cndmask_b32 %2, %2, %8, vcc\n\t"
                     "v_cndmask_b32 %3, %3, %8, vcc\n\t"
                     "v_cndmask_b32 %4, %4, %8, vcc\n\t"
                     "v_cndmask_b32 %5, %5, %8, vcc\n\t"
                     "v_cndmask_b32 %6, %6, %8, vcc\n\t"
                     "v_cndmask_b32 %7, %7, %8, vcc\n\t"
                     "v_cndmask_b32 %0, %0, %8, vcc\n\t"
                     "v_cndmask_b32 %1, %1, %8, vcc\n\t"
                     "v_cndmask_b32 %2, %2, %8, vcc\n\t"
                     "v_cndmask_b32 %3, %3, %8, vcc\n\t"
                     "v_cndmask_b32 %4, %4, %8, vcc\n\t"
                     "v_cndmask_b32 %5, %5, %8, vcc\n\t"
                     "v_cndmask_b32 %6, %6, %8, vcc\n\t"
                     "v_cndmask_b32 %7, %7, %8, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_max_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_max_u32 %0, %0, %8\n\t"
                     "v_max_u32 %1, %1, %8\n\t"
                     "v_max_u32 %2, %2, %8\n\t"
                     "v_max_u32 %3, %3, %8\n\t"
                     "v_max_u32 %4, %4, %8\n\t"
                     "v_max_u32 %5, %5, %8\n\t"
                     "v_max_u32 %6, %6, %8\n\t"
                     "v_max_u32 %7, %7, %8\n\t"
                     "v_max_u32 %0, %0, %8\n\t"
                     "v_max_u32 %1, %1, %8\n\t"
                     "v_max_u32 %2, %2, %8\n\t"
                     "v_max_u32 %3, %3, %8\n\t"
                     "v_max_u32 %4, %4, %8\n\t"
                     "v_max_u32 %5, %5, %8\n\t"
                     "v_max_u32 %6, %6, %8\n\t"
                     "v_max_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_sub_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_sub_u32 %0, %0, %8\n\t"
                     "v_sub_u32 %1, %1, %8\n\t"
                     "v_sub_u32 %2, %2, %8\n\t"
                     "v_sub_u32 %3, %3, %8\n\t"
                     "v_sub_u32 %4, %4, %8\n\t"
                     "v_sub_u32 %5, %5, %8\n\t"
                     "v_sub_u32 %6, %6, %8\n\t"
                     "v_sub_u32 %7, %7, %8\n\t"
                     "v_sub_u32 %0, %0, %8\n\t"
                     "v_sub_u32 %1, %1, %8\n\t"
                     "v_sub_u32 %2, %2, %8\n\t"
                     "v_sub_u32 %3, %3, %8\n\t"
                     "v_sub_u32 %4, %4, %8\n\t"
                     "v_sub_u32 %5, %5, %8\n\t"
                     "v_sub_u32 %6, %6, %8\n\t"
                     "v_sub_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_fma_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0; float a1 = (float)(seed + threadIdx.x) + (float)1; float a2 = (float)(seed + threadIdx.x) + (float)2; float a3 = (float)(seed + threadIdx.x) + (float)3; float a4 = (float)(seed + threadIdx.x) + (float)4; float a5 = (float)(seed + threadIdx.x) + (float)5; float a6 = (float)(seed + threadIdx.x) + (float)6; float a7 = (float)(seed + threadIdx.x) + (float)7;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n\t"
                     "v_fma_f32 %1, %1, %8, %9\n\t"
                     "v_fma_f32 %2, %2, %8, %9\n\t"
                     "v_fma_f32 %3, %3, %8, %9\n\t"
                     "v_fma_f32 %4, %4, %8, %9\n\t"
                     "v_fma_f32 %5, %5, %8, %9\n\t"
                     "v_fma_f32 %6, %6, %8, %9\n\t"
                     "v_fma_f32 %7, %7, %8, %9\n\t"
                     "v_fma_f32 %0, %0, %8, %9\n\t"
                     "v_fma_f32 %1, %1, %8, %9\n\t"
                     "v_fma_f32 %2, %2, %8, %9\n\t"
                     "v_fma_f32 %3, %3, %8, %9\n\t"
                     "v_fma_f32 %4, %4, %8, %9\n\t"
                     "v_fma_f32 %5, %5, %8, %9\n\t"
                     "v_fma_f32 %6, %6, %8, %9\n\t"
                     "v_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_fmac_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0; float a1 = (float)(seed + threadIdx.x) + (float)1; float a2 = (float)(seed + threadIdx.x) + (float)2; float a3 = (float)(seed + threadIdx.x) + (float)3; float a4 = (float)(seed + threadIdx.x) + (float)4; float a5 = (float)(seed + threadIdx.x) + (float)5; float a6 = (float)(seed + threadIdx.x) + (float)6; float a7 = (float)(seed + threadIdx.x) + (float)7;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fmac_f32 %0, %8, %9\n\t"
                     "v_fmac_f32 %1, %8, %9\n\t"
                     "v_fmac_f32 %2, %8, %9\n\t"
                     "v_fmac_f32 %3, %8, %9\n\t"
                     "v_fmac_f32 %4, %8, %9\n\t"
                     "v_fmac_f32 %5, %8, %9\n\t"
                     "v_fmac_f32 %6, %8, %9\n\t"
                     "v_fmac_f32 %7, %8, %9\n\t"
                     "v_fmac_f32 %0, %8, %9\n\t"
                     "v_fmac_f32 %1, %8, %9\n\t"
                     "v_fmac_f32 %2, %8, %9\n\t"
                     "v_fmac_f32 %3, %8, %9\n\t"
                     "v_fmac_f32 %4, %8, %9\n\t"
                     "v_fmac_f32 %5, %8, %9\n\t"
                     "v_fmac_f32 %6, %8, %9\n\t"
                     "v_fmac_f32 %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_mul_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0; float a1 = (float)(seed + threadIdx.x) + (float)1; float a2 = (float)(seed + threadIdx.x) + (float)2; float a3 = (float)(seed + threadIdx.x) + (float)3; float a4 = (float)(seed + threadIdx.x) + (float)4; float a5 = (float)(seed + threadIdx.x) + (float)5; float a6 = (float)(seed + threadIdx.x) + (float)6; float a7 = (float)(seed + threadIdx.x) + (float)7;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_mul_f32 %0, %0, %8\n\t"
                     "v_mul_f32 %1, %1, %8\n\t"
                     "v_mul_f32 %2, %2, %8\n\t"
                     "v_mul_f32 %3, %3, %8\n\t"
                     "v_mul_f32 %4, %4, %8\n\t"
                     "v_mul_f32 %5, %5, %8\n\t"
                     "v_mul_f32 %6, %6, %8\n\t"
                     "v_mul_f32 %7, %7, %8\n\t"
                     "v_mul_f32 %0, %0, %8\n\t"
                     "v_mul_f32 %1, %1, %8\n\t"
                     "v_mul_f32 %2, %2, %8\n\t"
                     "v_mul_f32 %3, %3, %8\n\t"
                     "v_mul_f32 %4, %4, %8\n\t"
                     "v_mul_f32 %5, %5, %8\n\t"
                     "v_mul_f32 %6, %6, %8\n\t"
                     "v_mul_f32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_add_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0; float a1 = (float)(seed + threadIdx.x) + (float)1; float a2 = (float)(seed + threadIdx.x) + (float)2; float a3 = (float)(seed + threadIdx.x) + (float)3; float a4 = (float)(seed + threadIdx.x) + (float)4; float a5 = (float)(seed + threadIdx.x) + (float)5; float a6 = (float)(seed + threadIdx.x) + (float)6; float a7 = (float)(seed + threadIdx.x) + (float)7;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_add_f32 %0, %0, %8\n\t"
                     "v_add_f32 %1, %1, %8\n\t"
                     "v_add_f32 %2, %2, %8\n\t"
                     "v_add_f32 %3, %3, %8\n\t"
                     "v_add_f32 %4, %4, %8\n\t"
                     "v_add_f32 %5, %5, %8\n\t"
                     "v_add_f32 %6, %6, %8\n\t"
                     "v_add_f32 %7, %7, %8\n\t"
                     "v_add_f32 %0, %0, %8\n\t"
                     "v_add_f32 %1, %1, %8\n\t"
                     "v_add_f32 %2, %2, %8\n\t"
                     "v_add_f32 %3, %3, %8\n\t"
                     "v_add_f32 %4, %4, %8\n\t"
                     "v_add_f32 %5, %5, %8\n\t"
                     "v_add_f32 %6, %6, %8\n\t"
                     "v_add_f32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_fmac_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fmac_f64 %0, %8, %9\n\t"
                     "v_fmac_f64 %1, %8, %9\n\t"
                     "v_fmac_f64 %2, %8, %9\n\t"
                     "v_fmac_f64 %3, %8, %9\n\t"
                     "v_fmac_f64 %4, %8, %9\n\t"
                     "v_fmac_f64 %5, %8, %9\n\t"
                     "v_fmac_f64 %6, %8, %9\n\t"
                     "v_fmac_f64 %7, %8, %9\n\t"
                     "v_fmac_f64 %0, %8, %9\n\t"
                     "v_fmac_f64 %1, %8, %9\n\t"
                     "v_fmac_f64 %2, %8, %9\n\t"
                     "v_fmac_f64 %3, %8, %9\n\t"
                     "v_fmac_f64 %4, %8, %9\n\t"
                     "v_fmac_f64 %5, %8, %9\n\t"
                     "v_fmac_f64 %6, %8, %9\n\t"
                     "v_fmac_f64 %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_pk_fma_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n\t"
                     "v_pk_fma_f32 %1, %1, %8, %9\n\t"
                     "v_pk_fma_f32 %2, %2, %8, %9\n\t"
                     "v_pk_fma_f32 %3, %3, %8, %9\n\t"
                     "v_pk_fma_f32 %4, %4, %8, %9\n\t"
                     "v_pk_fma_f32 %5, %5, %8, %9\n\t"
                     "v_pk_fma_f32 %6, %6, %8, %9\n\t"
                     "v_pk_fma_f32 %7, %7, %8, %9\n\t"
                     "v_pk_fma_f32 %0, %0, %8, %9\n\t"
                     "v_pk_fma_f32 %1, %1, %8, %9\n\t"
                     "v_pk_fma_f32 %2, %2, %8, %9\n\t"
                     "v_pk_fma_f32 %3, %3, %8, %9\n\t"
                     "v_pk_fma_f32 %4, %4, %8, %9\n\t"
                     "v_pk_fma_f32 %5, %5, %8, %9\n\t"
                     "v_pk_fma_f32 %6, %6, %8, %9\n\t"
                     "v_pk_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_pk_mul_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_pk_mul_f32 %0, %0, %8\n\t"
                     "v_pk_mul_f32 %1, %1, %8\n\t"
                     "v_pk_mul_f32 %2, %2, %8\n\t"
                     "v_pk_mul_f32 %3, %3, %8\n\t"
                     "v_pk_mul_f32 %4, %4, %8\n\t"
                     "v_pk_mul_f32 %5, %5, %8\n\t"
                     "v_pk_mul_f32 %6, %6, %8\n\t"
                     "v_pk_mul_f32 %7, %7, %8\n\t"
                     "v_pk_mul_f32 %0, %0, %8\n\t"
                     "v_pk_mul_f32 %1, %1, %8\n\t"
                     "v_pk_mul_f32 %2, %2, %8\n\t"
                     "v_pk_mul_f32 %3, %3, %8\n\t"
                     "v_pk_mul_f32 %4, %4, %8\n\t"
                     "v_pk_mul_f32 %5, %5, %8\n\t"
                     "v_pk_mul_f32 %6, %6, %8\n\t"
                     "v_pk_mul_f32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_pk_add_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_pk_add_f32 %0, %0, %8\n\t"
                     "v_pk_add_f32 %1, %1, %8\n\t"
                     "v_pk_add_f32 %2, %2, %8\n\t"
                     "v_pk_add_f32 %3, %3, %8\n\t"
                     "v_pk_add_f32 %4, %4, %8\n\t"
                     "v_pk_add_f32 %5, %5, %8\n\t"
                     "v_pk_add_f32 %6, %6, %8\n\t"
                     "v_pk_add_f32 %7, %7, %8\n\t"
                     "v_pk_add_f32 %0, %0, %8\n\t"
                     "v_pk_add_f32 %1, %1, %8\n\t"
                     "v_pk_add_f32 %2, %2, %8\n\t"
                     "v_pk_add_f32 %3, %3, %8\n\t"
                     "v_pk_add_f32 %4, %4, %8\n\t"
                     "v_pk_add_f32 %5, %5, %8\n\t"
                     "v_pk_add_f32 %6, %6, %8\n\t"
                     "v_pk_add_f32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_cvt_f32_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cvt_f32_u32 %0, %0\n\t"
                     "v_cvt_f32_u32 %1, %1\n\t"
                     "v_cvt_f32_u32 %2, %2\n\t"
                     "v_cvt_f32_u32 %3, %3\n\t"
                     "v_cvt_f32_u32 %4, %4\n\t"
                     "v_cvt_f32_u32 %5, %5\n\t"
                     "v_cvt_f32_u32 %6, %6\n\t"
                     "v_cvt_f32_u32 %7, %7\n\t"
                     "v_cvt_f32_u32 %0, %0\n\t"
                     "v_cvt_f32_u32 %1, %1\n\t"
                     "v_cvt_f32_u32 %2, %2\n\t"
                     "v_cvt_f32_u32 %3, %3\n\t"
                     "v_cvt_f32_u32 %4, %4\n\t"
                     "v_cvt_f32_u32 %5, %5\n\t"
                     "v_cvt_f32_u32 %6, %6\n\t"
                     "v_cvt_f32_u32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_cvt_u32_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cvt_u32_f32 %0, %0\n\t"
                     "v_cvt_u32_f32 %1, %1\n\t"
                     "v_cvt_u32_f32 %2, %2\n\t"
                     "v_cvt_u32_f32 %3, %3\n\t"
                     "v_cvt_u32_f32 %4, %4\n\t"
                     "v_cvt_u32_f32 %5, %5\n\t"
                     "v_cvt_u32_f32 %6, %6\n\t"
                     "v_cvt_u32_f32 %7, %7\n\t"
                     "v_cvt_u32_f32 %0, %0\n\t"
                     "v_cvt_u32_f32 %1, %1\n\t"
                     "v_cvt_u32_f32 %2, %2\n\t"
                     "v_cvt_u32_f32 %3, %3\n\t"
                     "v_cvt_u32_f32 %4, %4\n\t"
                     "v_cvt_u32_f32 %5, %5\n\t"
                     "v_cvt_u32_f32 %6, %6\n\t"
                     "v_cvt_u32_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_exp_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0; float a1 = (float)(seed + threadIdx.x) + (float)1; float a2 = (float)(seed + threadIdx.x) + (float)2; float a3 = (float)(seed + threadIdx.x) + (float)3; float a4 = (float)(seed + threadIdx.x) + (float)4; float a5 = (float)(seed + threadIdx.x) + (float)5; float a6 = (float)(seed + threadIdx.x) + (float)6; float a7 = (float)(seed + threadIdx.x) + (float)7;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_exp_f32 %0, %0\n\t"
                     "v_exp_f32 %1, %1\n\t"
                     "v_exp_f32 %2, %2\n\t"
                     "v_exp_f32 %3, %3\n\t"
                     "v_exp_f32 %4, %4\n\t"
                     "v_exp_f32 %5, %5\n\t"
                     "v_exp_f32 %6, %6\n\t"
                     "v_exp_f32 %7, %7\n\t"
                     "v_exp_f32 %0, %0\n\t"
                     "v_exp_f32 %1, %1\n\t"
                     "v_exp_f32 %2, %2\n\t"
                     "v_exp_f32 %3, %3\n\t"
                     "v_exp_f32 %4, %4\n\t"
                     "v_exp_f32 %5, %5\n\t"
                     "v_exp_f32 %6, %6\n\t"
                     "v_exp_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_log_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0; float a1 = (float)(seed + threadIdx.x) + (float)1; float a2 = (float)(seed + threadIdx.x) + (float)2; float a3 = (float)(seed + threadIdx.x) + (float)3; float a4 = (float)(seed + threadIdx.x) + (float)4; float a5 = (float)(seed + threadIdx.x) + (float)5; float a6 = (float)(seed + threadIdx.x) + (float)6; float a7 = (float)(seed + threadIdx.x) + (float)7;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_log_f32 %0, %0\n\t"
                     "v_log_f32 %1, %1\n\t"
                     "v_log_f32 %2, %2\n\t"
                     "v_log_f32 %3, %3\n\t"
                     "v_log_f32 %4, %4\n\t"
                     "v_log_f32 %5, %5\n\t"
                     "v_log_f32 %6, %6\n\t"
                     "v_log_f32 %7, %7\n\t"
                     "v_log_f32 %0, %0\n\t"
                     "v_log_f32 %1, %1\n\t"
                     "v_log_f32 %2, %2\n\t"
                     "v_log_f32 %3, %3\n\t"
                     "v_log_f32 %4, %4\n\t"
                     "v_log_f32 %5, %5\n\t"
                     "v_log_f32 %6, %6\n\t"
                     "v_log_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_rcp_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0; float a1 = (float)(seed + threadIdx.x) + (float)1; float a2 = (float)(seed + threadIdx.x) + (float)2; float a3 = (float)(seed + threadIdx.x) + (float)3; float a4 = (float)(seed + threadIdx.x) + (float)4; float a5 = (float)(seed + threadIdx.x) + (float)5; float a6 = (float)(seed + threadIdx.x) + (float)6; float a7 = (float)(seed + threadIdx.x) + (float)7;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_rcp_f32 %0, %0\n\t"
                     "v_rcp_f32 %1, %1\n\t"
                     "v_rcp_f32 %2, %2\n\t"
                     "v_rcp_f32 %3, %3\n\t"
                     "v_rcp_f32 %4, %4\n\t"
                     "v_rcp_f32 %5, %5\n\t"
                     "v_rcp_f32 %6, %6\n\t"
                     "v_rcp_f32 %7, %7\n\t"
                     "v_rcp_f32 %0, %0\n\t"
                     "v_rcp_f32 %1, %1\n\t"
                     "v_rcp_f32 %2, %2\n\t"
                     "v_rcp_f32 %3, %3\n\t"
                     "v_rcp_f32 %4, %4\n\t"
                     "v_rcp_f32 %5, %5\n\t"
                     "v_rcp_f32 %6, %6\n\t"
                     "v_rcp_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_rsq_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0; float a1 = (float)(seed + threadIdx.x) + (float)1; float a2 = (float)(seed + threadIdx.x) + (float)2; float a3 = (float)(seed + threadIdx.x) + (float)3; float a4 = (float)(seed + threadIdx.x) + (float)4; float a5 = (float)(seed + threadIdx.x) + (float)5; float a6 = (float)(seed + threadIdx.x) + (float)6; float a7 = (float)(seed + threadIdx.x) + (float)7;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_rsq_f32 %0, %0\n\t"
                     "v_rsq_f32 %1, %1\n\t"
                     "v_rsq_f32 %2, %2\n\t"
                     "v_rsq_f32 %3, %3\n\t"
                     "v_rsq_f32 %4, %4\n\t"
                     "v_rsq_f32 %5, %5\n\t"
                     "v_rsq_f32 %6, %6\n\t"
                     "v_rsq_f32 %7, %7\n\t"
                     "v_rsq_f32 %0, %0\n\t"
                     "v_rsq_f32 %1, %1\n\t"
                     "v_rsq_f32 %2, %2\n\t"
                     "v_rsq_f32 %3, %3\n\t"
                     "v_rsq_f32 %4, %4\n\t"
                     "v_rsq_f32 %5, %5\n\t"
                     "v_rsq_f32 %6, %6\n\t"
                     "v_rsq_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_sqrt_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0; float a1 = (float)(seed + threadIdx.x) + (float)1; float a2 = (float)(seed + threadIdx.x) + (float)2; float a3 = (float)(seed + threadIdx.x) + (float)3; float a4 = (float)(seed + threadIdx.x) + (float)4; float a5 = (float)(seed + threadIdx.x) + (float)5; float a6 = (float)(seed + threadIdx.x) + (float)6; float a7 = (float)(seed + threadIdx.x) + (float)7;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_sqrt_f32 %0, %0\n\t"
                     "v_sqrt_f32 %1, %1\n\t"
                     "v_sqrt_f32 %2, %2\n\t"
                     "v_sqrt_f32 %3, %3\n\t"
                     "v_sqrt_f32 %4, %4\n\t"
                     "v_sqrt_f32 %5, %5\n\t"
                     "v_sqrt_f32 %6, %6\n\t"
                     "v_sqrt_f32 %7, %7\n\t"
                     "v_sqrt_f32 %0, %0\n\t"
                     "v_sqrt_f32 %1, %1\n\t"
                     "v_sqrt_f32 %2, %2\n\t"
                     "v_sqrt_f32 %3, %3\n\t"
                     "v_sqrt_f32 %4, %4\n\t"
                     "v_sqrt_f32 %5, %5\n\t"
                     "v_sqrt_f32 %6, %6\n\t"
                     "v_sqrt_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_sin_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0; float a1 = (float)(seed + threadIdx.x) + (float)1; float a2 = (float)(seed + threadIdx.x) + (float)2; float a3 = (float)(seed + threadIdx.x) + (float)3; float a4 = (float)(seed + threadIdx.x) + (float)4; float a5 = (float)(seed + threadIdx.x) + (float)5; float a6 = (float)(seed + threadIdx.x) + (float)6; float a7 = (float)(seed + threadIdx.x) + (float)7;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_sin_f32 %0, %0\n\t"
                     "v_sin_f32 %1, %1\n\t"
                     "v_sin_f32 %2, %2\n\t"
                     "v_sin_f32 %3, %3\n\t"
                     "v_sin_f32 %4, %4\n\t"
                     "v_sin_f32 %5, %5\n\t"
                     "v_sin_f32 %6, %6\n\t"
                     "v_sin_f32 %7, %7\n\t"
                     "v_sin_f32 %0, %0\n\t"
                     "v_sin_f32 %1, %1\n\t"
                     "v_sin_f32 %2, %2\n\t"
                     "v_sin_f32 %3, %3\n\t"
                     "v_sin_f32 %4, %4\n\t"
                     "v_sin_f32 %5, %5\n\t"
                     "v_sin_f32 %6, %6\n\t"
                     "v_sin_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_cos_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0; float a1 = (float)(seed + threadIdx.x) + (float)1; float a2 = (float)(seed + threadIdx.x) + (float)2; float a3 = (float)(seed + threadIdx.x) + (float)3; float a4 = (float)(seed + threadIdx.x) + (float)4; float a5 = (float)(seed + threadIdx.x) + (float)5; float a6 = (float)(seed + threadIdx.x) + (float)6; float a7 = (float)(seed + threadIdx.x) + (float)7;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cos_f32 %0, %0\n\t"
                     "v_cos_f32 %1, %1\n\t"
                     "v_cos_f32 %2, %2\n\t"
                     "v_cos_f32 %3, %3\n\t"
                     "v_cos_f32 %4, %4\n\t"
                     "v_cos_f32 %5, %5\n\t"
                     "v_cos_f32 %6, %6\n\t"
                     "v_cos_f32 %7, %7\n\t"
                     "v_cos_f32 %0, %0\n\t"
                     "v_cos_f32 %1, %1\n\t"
                     "v_cos_f32 %2, %2\n\t"
                     "v_cos_f32 %3, %3\n\t"
                     "v_cos_f32 %4, %4\n\t"
                     "v_cos_f32 %5, %5\n\t"
                     "v_cos_f32 %6, %6\n\t"
                     "v_cos_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void c_v_fma_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2\n\t"
                     "v_fma_f64 %0, %0, %1, %2"
                     : "+v"(a0) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void c_v_add_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1\n\t"
                     "v_add_u32 %0, %0, %1"
                     : "+v"(a0) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void c_v_fma_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float a0 = (float)(seed + threadIdx.x) + (float)0;
    float b = (float)(1.0000001f), c = (float)(0.5f);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2\n\t"
                     "v_fma_f32 %0, %0, %1, %2"
                     : "+v"(a0) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void c_v_rcp_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0;
    double b = (double)(1.0000001), c = (double)(0.5);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0\n\t"
                     "v_rcp_f64 %0, %0"
                     : "+v"(a0) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void c_v_mad_u64_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint64_t a0 = (uint64_t)(seed + threadIdx.x) + (uint64_t)0;
    uint32_t b = (uint32_t)(0xD2511F53u), c = (uint32_t)(12345u);
    asm volatile("" : "+v"(b), "+v"(c));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0"
                     : "+v"(a0) : "v"(b), "v"(c) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m11_add_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7; uint32_t i0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t i1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t i2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t i3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t i4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t i5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t i6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t i7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    double b = (double)1.0000001, c = (double)0.5;
    uint32_t ib = (uint32_t)0xD2511F53u, ic = (uint32_t)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %16, %17\n\t"
                     "v_add_u32 %8, %8, %18\n\t"
                     "v_fma_f64 %1, %1, %16, %17\n\t"
                     "v_add_u32 %9, %9, %18\n\t"
                     "v_fma_f64 %2, %2, %16, %17\n\t"
                     "v_add_u32 %10, %10, %18\n\t"
                     "v_fma_f64 %3, %3, %16, %17\n\t"
                     "v_add_u32 %11, %11, %18\n\t"
                     "v_fma_f64 %4, %4, %16, %17\n\t"
                     "v_add_u32 %12, %12, %18\n\t"
                     "v_fma_f64 %5, %5, %16, %17\n\t"
                     "v_add_u32 %13, %13, %18\n\t"
                     "v_fma_f64 %6, %6, %16, %17\n\t"
                     "v_add_u32 %14, %14, %18\n\t"
                     "v_fma_f64 %7, %7, %16, %17\n\t"
                     "v_add_u32 %15, %15, %18"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)i0 + (double)i1 + (double)i2 + (double)i3 + (double)i4 + (double)i5 + (double)i6 + (double)i7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m31_add_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; uint32_t i0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t i1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1;
    double b = (double)1.0000001, c = (double)0.5;
    uint32_t ib = (uint32_t)0xD2511F53u, ic = (uint32_t)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_add_u32 %6, %6, %10\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_add_u32 %7, %7, %10\n\t"
                     "v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_add_u32 %6, %6, %10\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_add_u32 %7, %7, %10"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(i0), "+v"(i1) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)i0 + (double)i1 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m11_bitop3(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7; uint32_t i0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t i1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t i2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t i3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t i4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t i5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t i6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t i7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    double b = (double)1.0000001, c = (double)0.5;
    uint32_t ib = (uint32_t)0xD2511F53u, ic = (uint32_t)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %16, %17\n\t"
                     "v_bitop3_b32 %8, %8, %18, %19 bitop3:0x96\n\t"
                     "v_fma_f64 %1, %1, %16, %17\n\t"
                     "v_bitop3_b32 %9, %9, %18, %19 bitop3:0x96\n\t"
                     "v_fma_f64 %2, %2, %16, %17\n\t"
                     "v_bitop3_b32 %10, %10, %18, %19 bitop3:0x96\n\t"
                     "v_fma_f64 %3, %3, %16, %17\n\t"
                     "v_bitop3_b32 %11, %11, %18, %19 bitop3:0x96\n\t"
                     "v_fma_f64 %4, %4, %16, %17\n\t"
                     "v_bitop3_b32 %12, %12, %18, %19 bitop3:0x96\n\t"
                     "v_fma_f64 %5, %5, %16, %17\n\t"
                     "v_bitop3_b32 %13, %13, %18, %19 bitop3:0x96\n\t"
                     "v_fma_f64 %6, %6, %16, %17\n\t"
                     "v_bitop3_b32 %14, %14, %18, %19 bitop3:0x96\n\t"
                     "v_fma_f64 %7, %7, %16, %17\n\t"
                     "v_bitop3_b32 %15, %15, %18, %19 bitop3:0x96"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)i0 + (double)i1 + (double)i2 + (double)i3 + (double)i4 + (double)i5 + (double)i6 + (double)i7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m31_bitop3(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; uint32_t i0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t i1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1;
    double b = (double)1.0000001, c = (double)0.5;
    uint32_t ib = (uint32_t)0xD2511F53u, ic = (uint32_t)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_bitop3_b32 %6, %6, %10, %11 bitop3:0x96\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_bitop3_b32 %7, %7, %10, %11 bitop3:0x96\n\t"
                     "v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_bitop3_b32 %6, %6, %10, %11 bitop3:0x96\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_bitop3_b32 %7, %7, %10, %11 bitop3:0x96"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(i0), "+v"(i1) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)i0 + (double)i1 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m11_add3(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7; uint32_t i0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t i1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t i2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t i3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t i4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t i5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t i6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t i7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7;
    double b = (double)1.0000001, c = (double)0.5;
    uint32_t ib = (uint32_t)0xD2511F53u, ic = (uint32_t)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %16, %17\n\t"
                     "v_add3_u32 %8, %8, %18, %19\n\t"
                     "v_fma_f64 %1, %1, %16, %17\n\t"
                     "v_add3_u32 %9, %9, %18, %19\n\t"
                     "v_fma_f64 %2, %2, %16, %17\n\t"
                     "v_add3_u32 %10, %10, %18, %19\n\t"
                     "v_fma_f64 %3, %3, %16, %17\n\t"
                     "v_add3_u32 %11, %11, %18, %19\n\t"
                     "v_fma_f64 %4, %4, %16, %17\n\t"
                     "v_add3_u32 %12, %12, %18, %19\n\t"
                     "v_fma_f64 %5, %5, %16, %17\n\t"
                     "v_add3_u32 %13, %13, %18, %19\n\t"
                     "v_fma_f64 %6, %6, %16, %17\n\t"
                     "v_add3_u32 %14, %14, %18, %19\n\t"
                     "v_fma_f64 %7, %7, %16, %17\n\t"
                     "v_add3_u32 %15, %15, %18, %19"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)i0 + (double)i1 + (double)i2 + (double)i3 + (double)i4 + (double)i5 + (double)i6 + (double)i7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m31_add3(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; uint32_t i0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t i1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1;
    double b = (double)1.0000001, c = (double)0.5;
    uint32_t ib = (uint32_t)0xD2511F53u, ic = (uint32_t)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_add3_u32 %6, %6, %10, %11\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_add3_u32 %7, %7, %10, %11\n\t"
                     "v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_add3_u32 %6, %6, %10, %11\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_add3_u32 %7, %7, %10, %11"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(i0), "+v"(i1) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)i0 + (double)i1 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m11_fma_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7; float i0 = (float)(seed + threadIdx.x) + (float)0; float i1 = (float)(seed + threadIdx.x) + (float)1; float i2 = (float)(seed + threadIdx.x) + (float)2; float i3 = (float)(seed + threadIdx.x) + (float)3; float i4 = (float)(seed + threadIdx.x) + (float)4; float i5 = (float)(seed + threadIdx.x) + (float)5; float i6 = (float)(seed + threadIdx.x) + (float)6; float i7 = (float)(seed + threadIdx.x) + (float)7;
    double b = (double)1.0000001, c = (double)0.5;
    float ib = (float)0xD2511F53u, ic = (float)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %16, %17\n\t"
                     "v_fma_f32 %8, %8, %18, %19\n\t"
                     "v_fma_f64 %1, %1, %16, %17\n\t"
                     "v_fma_f32 %9, %9, %18, %19\n\t"
                     "v_fma_f64 %2, %2, %16, %17\n\t"
                     "v_fma_f32 %10, %10, %18, %19\n\t"
                     "v_fma_f64 %3, %3, %16, %17\n\t"
                     "v_fma_f32 %11, %11, %18, %19\n\t"
                     "v_fma_f64 %4, %4, %16, %17\n\t"
                     "v_fma_f32 %12, %12, %18, %19\n\t"
                     "v_fma_f64 %5, %5, %16, %17\n\t"
                     "v_fma_f32 %13, %13, %18, %19\n\t"
                     "v_fma_f64 %6, %6, %16, %17\n\t"
                     "v_fma_f32 %14, %14, %18, %19\n\t"
                     "v_fma_f64 %7, %7, %16, %17\n\t"
                     "v_fma_f32 %15, %15, %18, %19"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)i0 + (double)i1 + (double)i2 + (double)i3 + (double)i4 + (double)i5 + (double)i6 + (double)i7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m31_fma_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; float i0 = (float)(seed + threadIdx.x) + (float)0; float i1 = (float)(seed + threadIdx.x) + (float)1;
    double b = (double)1.0000001, c = (double)0.5;
    float ib = (float)0xD2511F53u, ic = (float)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_fma_f32 %6, %6, %10, %11\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_fma_f32 %7, %7, %10, %11\n\t"
                     "v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_fma_f32 %6, %6, %10, %11\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_fma_f32 %7, %7, %10, %11"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(i0), "+v"(i1) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)i0 + (double)i1 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m11_pk_fma_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7; double i0 = (double)(seed + threadIdx.x) + (double)0; double i1 = (double)(seed + threadIdx.x) + (double)1; double i2 = (double)(seed + threadIdx.x) + (double)2; double i3 = (double)(seed + threadIdx.x) + (double)3; double i4 = (double)(seed + threadIdx.x) + (double)4; double i5 = (double)(seed + threadIdx.x) + (double)5; double i6 = (double)(seed + threadIdx.x) + (double)6; double i7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)1.0000001, c = (double)0.5;
    double ib = (double)0xD2511F53u, ic = (double)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %16, %17\n\t"
                     "v_pk_fma_f32 %8, %8, %18, %19\n\t"
                     "v_fma_f64 %1, %1, %16, %17\n\t"
                     "v_pk_fma_f32 %9, %9, %18, %19\n\t"
                     "v_fma_f64 %2, %2, %16, %17\n\t"
                     "v_pk_fma_f32 %10, %10, %18, %19\n\t"
                     "v_fma_f64 %3, %3, %16, %17\n\t"
                     "v_pk_fma_f32 %11, %11, %18, %19\n\t"
                     "v_fma_f64 %4, %4, %16, %17\n\t"
                     "v_pk_fma_f32 %12, %12, %18, %19\n\t"
                     "v_fma_f64 %5, %5, %16, %17\n\t"
                     "v_pk_fma_f32 %13, %13, %18, %19\n\t"
                     "v_fma_f64 %6, %6, %16, %17\n\t"
                     "v_pk_fma_f32 %14, %14, %18, %19\n\t"
                     "v_fma_f64 %7, %7, %16, %17\n\t"
                     "v_pk_fma_f32 %15, %15, %18, %19"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)i0 + (double)i1 + (double)i2 + (double)i3 + (double)i4 + (double)i5 + (double)i6 + (double)i7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m31_pk_fma_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double i0 = (double)(seed + threadIdx.x) + (double)0; double i1 = (double)(seed + threadIdx.x) + (double)1;
    double b = (double)1.0000001, c = (double)0.5;
    double ib = (double)0xD2511F53u, ic = (double)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_pk_fma_f32 %6, %6, %10, %11\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_pk_fma_f32 %7, %7, %10, %11\n\t"
                     "v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_pk_fma_f32 %6, %6, %10, %11\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_pk_fma_f32 %7, %7, %10, %11"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(i0), "+v"(i1) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)i0 + (double)i1 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m11_mad_u64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7; uint64_t i0 = (uint64_t)(seed + threadIdx.x) + (uint64_t)0; uint64_t i1 = (uint64_t)(seed + threadIdx.x) + (uint64_t)1; uint64_t i2 = (uint64_t)(seed + threadIdx.x) + (uint64_t)2; uint64_t i3 = (uint64_t)(seed + threadIdx.x) + (uint64_t)3; uint64_t i4 = (uint64_t)(seed + threadIdx.x) + (uint64_t)4; uint64_t i5 = (uint64_t)(seed + threadIdx.x) + (uint64_t)5; uint64_t i6 = (uint64_t)(seed + threadIdx.x) + (uint64_t)6; uint64_t i7 = (uint64_t)(seed + threadIdx.x) + (uint64_t)7;
    double b = (double)1.0000001, c = (double)0.5;
    uint32_t ib = (uint32_t)0xD2511F53u, ic = (uint32_t)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %16, %17\n\t"
                     "v_mad_u64_u32 %8, vcc, %18, %19, %8\n\t"
                     "v_fma_f64 %1, %1, %16, %17\n\t"
                     "v_mad_u64_u32 %9, vcc, %18, %19, %9\n\t"
                     "v_fma_f64 %2, %2, %16, %17\n\t"
                     "v_mad_u64_u32 %10, vcc, %18, %19, %10\n\t"
                     "v_fma_f64 %3, %3, %16, %17\n\t"
                     "v_mad_u64_u32 %11, vcc, %18, %19, %11\n\t"
                     "v_fma_f64 %4, %4, %16, %17\n\t"
                     "v_mad_u64_u32 %12, vcc, %18, %19, %12\n\t"
                     "v_fma_f64 %5, %5, %16, %17\n\t"
                     "v_mad_u64_u32 %13, vcc, %18, %19, %13\n\t"
                     "v_fma_f64 %6, %6, %16, %17\n\t"
                     "v_mad_u64_u32 %14, vcc, %18, %19, %14\n\t"
                     "v_fma_f64 %7, %7, %16, %17\n\t"
                     "v_mad_u64_u32 %15, vcc, %18, %19, %15"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)i0 + (double)i1 + (double)i2 + (double)i3 + (double)i4 + (double)i5 + (double)i6 + (double)i7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m31_mad_u64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; uint64_t i0 = (uint64_t)(seed + threadIdx.x) + (uint64_t)0; uint64_t i1 = (uint64_t)(seed + threadIdx.x) + (uint64_t)1;
    double b = (double)1.0000001, c = (double)0.5;
    uint32_t ib = (uint32_t)0xD2511F53u, ic = (uint32_t)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_mad_u64_u32 %6, vcc, %10, %11, %6\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_mad_u64_u32 %7, vcc, %10, %11, %7\n\t"
                     "v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_mad_u64_u32 %6, vcc, %10, %11, %6\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_mad_u64_u32 %7, vcc, %10, %11, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(i0), "+v"(i1) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)i0 + (double)i1 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m11_rcp_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7; double i0 = (double)(seed + threadIdx.x) + (double)0; double i1 = (double)(seed + threadIdx.x) + (double)1; double i2 = (double)(seed + threadIdx.x) + (double)2; double i3 = (double)(seed + threadIdx.x) + (double)3; double i4 = (double)(seed + threadIdx.x) + (double)4; double i5 = (double)(seed + threadIdx.x) + (double)5; double i6 = (double)(seed + threadIdx.x) + (double)6; double i7 = (double)(seed + threadIdx.x) + (double)7;
    double b = (double)1.0000001, c = (double)0.5;
    double ib = (double)0xD2511F53u, ic = (double)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %16, %17\n\t"
                     "v_rcp_f64 %8, %8\n\t"
                     "v_fma_f64 %1, %1, %16, %17\n\t"
                     "v_rcp_f64 %9, %9\n\t"
                     "v_fma_f64 %2, %2, %16, %17\n\t"
                     "v_rcp_f64 %10, %10\n\t"
                     "v_fma_f64 %3, %3, %16, %17\n\t"
                     "v_rcp_f64 %11, %11\n\t"
                     "v_fma_f64 %4, %4, %16, %17\n\t"
                     "v_rcp_f64 %12, %12\n\t"
                     "v_fma_f64 %5, %5, %16, %17\n\t"
                     "v_rcp_f64 %13, %13\n\t"
                     "v_fma_f64 %6, %6, %16, %17\n\t"
                     "v_rcp_f64 %14, %14\n\t"
                     "v_fma_f64 %7, %7, %16, %17\n\t"
                     "v_rcp_f64 %15, %15"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)i0 + (double)i1 + (double)i2 + (double)i3 + (double)i4 + (double)i5 + (double)i6 + (double)i7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m31_rcp_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double i0 = (double)(seed + threadIdx.x) + (double)0; double i1 = (double)(seed + threadIdx.x) + (double)1;
    double b = (double)1.0000001, c = (double)0.5;
    double ib = (double)0xD2511F53u, ic = (double)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_rcp_f64 %6, %6\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_rcp_f64 %7, %7\n\t"
                     "v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_rcp_f64 %6, %6\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_rcp_f64 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(i0), "+v"(i1) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)i0 + (double)i1 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m11_exp_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; double a6 = (double)(seed + threadIdx.x) + (double)6; double a7 = (double)(seed + threadIdx.x) + (double)7; float i0 = (float)(seed + threadIdx.x) + (float)0; float i1 = (float)(seed + threadIdx.x) + (float)1; float i2 = (float)(seed + threadIdx.x) + (float)2; float i3 = (float)(seed + threadIdx.x) + (float)3; float i4 = (float)(seed + threadIdx.x) + (float)4; float i5 = (float)(seed + threadIdx.x) + (float)5; float i6 = (float)(seed + threadIdx.x) + (float)6; float i7 = (float)(seed + threadIdx.x) + (float)7;
    double b = (double)1.0000001, c = (double)0.5;
    float ib = (float)0xD2511F53u, ic = (float)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %16, %17\n\t"
                     "v_exp_f32 %8, %8\n\t"
                     "v_fma_f64 %1, %1, %16, %17\n\t"
                     "v_exp_f32 %9, %9\n\t"
                     "v_fma_f64 %2, %2, %16, %17\n\t"
                     "v_exp_f32 %10, %10\n\t"
                     "v_fma_f64 %3, %3, %16, %17\n\t"
                     "v_exp_f32 %11, %11\n\t"
                     "v_fma_f64 %4, %4, %16, %17\n\t"
                     "v_exp_f32 %12, %12\n\t"
                     "v_fma_f64 %5, %5, %16, %17\n\t"
                     "v_exp_f32 %13, %13\n\t"
                     "v_fma_f64 %6, %6, %16, %17\n\t"
                     "v_exp_f32 %14, %14\n\t"
                     "v_fma_f64 %7, %7, %16, %17\n\t"
                     "v_exp_f32 %15, %15"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)i0 + (double)i1 + (double)i2 + (double)i3 + (double)i4 + (double)i5 + (double)i6 + (double)i7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m31_exp_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = (double)(seed + threadIdx.x) + (double)0; double a1 = (double)(seed + threadIdx.x) + (double)1; double a2 = (double)(seed + threadIdx.x) + (double)2; double a3 = (double)(seed + threadIdx.x) + (double)3; double a4 = (double)(seed + threadIdx.x) + (double)4; double a5 = (double)(seed + threadIdx.x) + (double)5; float i0 = (float)(seed + threadIdx.x) + (float)0; float i1 = (float)(seed + threadIdx.x) + (float)1;
    double b = (double)1.0000001, c = (double)0.5;
    float ib = (float)0xD2511F53u, ic = (float)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_exp_f32 %6, %6\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_exp_f32 %7, %7\n\t"
                     "v_fma_f64 %0, %0, %8, %9\n\t"
                     "v_fma_f64 %1, %1, %8, %9\n\t"
                     "v_fma_f64 %2, %2, %8, %9\n\t"
                     "v_exp_f32 %6, %6\n\t"
                     "v_fma_f64 %3, %3, %8, %9\n\t"
                     "v_fma_f64 %4, %4, %8, %9\n\t"
                     "v_fma_f64 %5, %5, %8, %9\n\t"
                     "v_exp_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(i0), "+v"(i1) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)i0 + (double)i1 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m11_add_u32_fma_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7; float i0 = (float)(seed + threadIdx.x) + (float)0; float i1 = (float)(seed + threadIdx.x) + (float)1; float i2 = (float)(seed + threadIdx.x) + (float)2; float i3 = (float)(seed + threadIdx.x) + (float)3; float i4 = (float)(seed + threadIdx.x) + (float)4; float i5 = (float)(seed + threadIdx.x) + (float)5; float i6 = (float)(seed + threadIdx.x) + (float)6; float i7 = (float)(seed + threadIdx.x) + (float)7;
    uint32_t b = (uint32_t)1.0000001, c = (uint32_t)0.5;
    float ib = (float)0xD2511F53u, ic = (float)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_add_u32 %0, %0, %16\n\t"
                     "v_fma_f32 %8, %8, %18, %19\n\t"
                     "v_add_u32 %1, %1, %16\n\t"
                     "v_fma_f32 %9, %9, %18, %19\n\t"
                     "v_add_u32 %2, %2, %16\n\t"
                     "v_fma_f32 %10, %10, %18, %19\n\t"
                     "v_add_u32 %3, %3, %16\n\t"
                     "v_fma_f32 %11, %11, %18, %19\n\t"
                     "v_add_u32 %4, %4, %16\n\t"
                     "v_fma_f32 %12, %12, %18, %19\n\t"
                     "v_add_u32 %5, %5, %16\n\t"
                     "v_fma_f32 %13, %13, %18, %19\n\t"
                     "v_add_u32 %6, %6, %16\n\t"
                     "v_fma_f32 %14, %14, %18, %19\n\t"
                     "v_add_u32 %7, %7, %16\n\t"
                     "v_fma_f32 %15, %15, %18, %19"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)i0 + (double)i1 + (double)i2 + (double)i3 + (double)i4 + (double)i5 + (double)i6 + (double)i7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void m11_add_u32_exp_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t a0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t a1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t a2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t a3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3; uint32_t a4 = (uint32_t)(seed + threadIdx.x) + (uint32_t)4; uint32_t a5 = (uint32_t)(seed + threadIdx.x) + (uint32_t)5; uint32_t a6 = (uint32_t)(seed + threadIdx.x) + (uint32_t)6; uint32_t a7 = (uint32_t)(seed + threadIdx.x) + (uint32_t)7; float i0 = (float)(seed + threadIdx.x) + (float)0; float i1 = (float)(seed + threadIdx.x) + (float)1; float i2 = (float)(seed + threadIdx.x) + (float)2; float i3 = (float)(seed + threadIdx.x) + (float)3; float i4 = (float)(seed + threadIdx.x) + (float)4; float i5 = (float)(seed + threadIdx.x) + (float)5; float i6 = (float)(seed + threadIdx.x) + (float)6; float i7 = (float)(seed + threadIdx.x) + (float)7;
    uint32_t b = (uint32_t)1.0000001, c = (uint32_t)0.5;
    float ib = (float)0xD2511F53u, ic = (float)12345u;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(ib), "+v"(ic));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_add_u32 %0, %0, %16\n\t"
                     "v_exp_f32 %8, %8\n\t"
                     "v_add_u32 %1, %1, %16\n\t"
                     "v_exp_f32 %9, %9\n\t"
                     "v_add_u32 %2, %2, %16\n\t"
                     "v_exp_f32 %10, %10\n\t"
                     "v_add_u32 %3, %3, %16\n\t"
                     "v_exp_f32 %11, %11\n\t"
                     "v_add_u32 %4, %4, %16\n\t"
                     "v_exp_f32 %12, %12\n\t"
                     "v_add_u32 %5, %5, %16\n\t"
                     "v_exp_f32 %13, %13\n\t"
                     "v_add_u32 %6, %6, %16\n\t"
                     "v_exp_f32 %14, %14\n\t"
                     "v_add_u32 %7, %7, %16\n\t"
                     "v_exp_f32 %15, %15"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b), "v"(c), "v"(ib), "v"(ic) : "vcc");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)a0 + (double)a1 + (double)a2 + (double)a3 + (double)a4 + (double)a5 + (double)a6 + (double)a7 + (double)i0 + (double)i1 + (double)i2 + (double)i3 + (double)i4 + (double)i5 + (double)i6 + (double)i7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_cvt_f64_u32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double d0 = 0; double d1 = 0; double d2 = 0; double d3 = 0; double d4 = 0; double d5 = 0; double d6 = 0; double d7 = 0; uint32_t s0 = (uint32_t)(seed + threadIdx.x) + (uint32_t)0; uint32_t s1 = (uint32_t)(seed + threadIdx.x) + (uint32_t)1; uint32_t s2 = (uint32_t)(seed + threadIdx.x) + (uint32_t)2; uint32_t s3 = (uint32_t)(seed + threadIdx.x) + (uint32_t)3;
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cvt_f64_u32 %0, %8\n\t"
                     "v_cvt_f64_u32 %1, %9\n\t"
                     "v_cvt_f64_u32 %2, %10\n\t"
                     "v_cvt_f64_u32 %3, %11\n\t"
                     "v_cvt_f64_u32 %4, %8\n\t"
                     "v_cvt_f64_u32 %5, %9\n\t"
                     "v_cvt_f64_u32 %6, %10\n\t"
                     "v_cvt_f64_u32 %7, %11\n\t"
                     "v_cvt_f64_u32 %0, %8\n\t"
                     "v_cvt_f64_u32 %1, %9\n\t"
                     "v_cvt_f64_u32 %2, %10\n\t"
                     "v_cvt_f64_u32 %3, %11\n\t"
                     "v_cvt_f64_u32 %4, %8\n\t"
                     "v_cvt_f64_u32 %5, %9\n\t"
                     "v_cvt_f64_u32 %6, %10\n\t"
                     "v_cvt_f64_u32 %7, %11"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)d0 + (double)d1 + (double)d2 + (double)d3 + (double)d4 + (double)d5 + (double)d6 + (double)d7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_cvt_f64_i32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double d0 = 0; double d1 = 0; double d2 = 0; double d3 = 0; double d4 = 0; double d5 = 0; double d6 = 0; double d7 = 0; int32_t s0 = (int32_t)(seed + threadIdx.x) + (int32_t)0; int32_t s1 = (int32_t)(seed + threadIdx.x) + (int32_t)1; int32_t s2 = (int32_t)(seed + threadIdx.x) + (int32_t)2; int32_t s3 = (int32_t)(seed + threadIdx.x) + (int32_t)3;
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cvt_f64_i32 %0, %8\n\t"
                     "v_cvt_f64_i32 %1, %9\n\t"
                     "v_cvt_f64_i32 %2, %10\n\t"
                     "v_cvt_f64_i32 %3, %11\n\t"
                     "v_cvt_f64_i32 %4, %8\n\t"
                     "v_cvt_f64_i32 %5, %9\n\t"
                     "v_cvt_f64_i32 %6, %10\n\t"
                     "v_cvt_f64_i32 %7, %11\n\t"
                     "v_cvt_f64_i32 %0, %8\n\t"
                     "v_cvt_f64_i32 %1, %9\n\t"
                     "v_cvt_f64_i32 %2, %10\n\t"
                     "v_cvt_f64_i32 %3, %11\n\t"
                     "v_cvt_f64_i32 %4, %8\n\t"
                     "v_cvt_f64_i32 %5, %9\n\t"
                     "v_cvt_f64_i32 %6, %10\n\t"
                     "v_cvt_f64_i32 %7, %11"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)d0 + (double)d1 + (double)d2 + (double)d3 + (double)d4 + (double)d5 + (double)d6 + (double)d7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_cvt_f64_f32(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double d0 = 0; double d1 = 0; double d2 = 0; double d3 = 0; double d4 = 0; double d5 = 0; double d6 = 0; double d7 = 0; float s0 = (float)(seed + threadIdx.x) + (float)0; float s1 = (float)(seed + threadIdx.x) + (float)1; float s2 = (float)(seed + threadIdx.x) + (float)2; float s3 = (float)(seed + threadIdx.x) + (float)3;
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cvt_f64_f32 %0, %8\n\t"
                     "v_cvt_f64_f32 %1, %9\n\t"
                     "v_cvt_f64_f32 %2, %10\n\t"
                     "v_cvt_f64_f32 %3, %11\n\t"
                     "v_cvt_f64_f32 %4, %8\n\t"
                     "v_cvt_f64_f32 %5, %9\n\t"
                     "v_cvt_f64_f32 %6, %10\n\t"
                     "v_cvt_f64_f32 %7, %11\n\t"
                     "v_cvt_f64_f32 %0, %8\n\t"
                     "v_cvt_f64_f32 %1, %9\n\t"
                     "v_cvt_f64_f32 %2, %10\n\t"
                     "v_cvt_f64_f32 %3, %11\n\t"
                     "v_cvt_f64_f32 %4, %8\n\t"
                     "v_cvt_f64_f32 %5, %9\n\t"
                     "v_cvt_f64_f32 %6, %10\n\t"
                     "v_cvt_f64_f32 %7, %11"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)d0 + (double)d1 + (double)d2 + (double)d3 + (double)d4 + (double)d5 + (double)d6 + (double)d7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_cvt_f32_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    float d0 = 0; float d1 = 0; float d2 = 0; float d3 = 0; float d4 = 0; float d5 = 0; float d6 = 0; float d7 = 0; double s0 = (double)(seed + threadIdx.x) + (double)0; double s1 = (double)(seed + threadIdx.x) + (double)1; double s2 = (double)(seed + threadIdx.x) + (double)2; double s3 = (double)(seed + threadIdx.x) + (double)3;
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cvt_f32_f64 %0, %8\n\t"
                     "v_cvt_f32_f64 %1, %9\n\t"
                     "v_cvt_f32_f64 %2, %10\n\t"
                     "v_cvt_f32_f64 %3, %11\n\t"
                     "v_cvt_f32_f64 %4, %8\n\t"
                     "v_cvt_f32_f64 %5, %9\n\t"
                     "v_cvt_f32_f64 %6, %10\n\t"
                     "v_cvt_f32_f64 %7, %11\n\t"
                     "v_cvt_f32_f64 %0, %8\n\t"
                     "v_cvt_f32_f64 %1, %9\n\t"
                     "v_cvt_f32_f64 %2, %10\n\t"
                     "v_cvt_f32_f64 %3, %11\n\t"
                     "v_cvt_f32_f64 %4, %8\n\t"
                     "v_cvt_f32_f64 %5, %9\n\t"
                     "v_cvt_f32_f64 %6, %10\n\t"
                     "v_cvt_f32_f64 %7, %11"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)d0 + (double)d1 + (double)d2 + (double)d3 + (double)d4 + (double)d5 + (double)d6 + (double)d7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_cvt_u32_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    uint32_t d0 = 0; uint32_t d1 = 0; uint32_t d2 = 0; uint32_t d3 = 0; uint32_t d4 = 0; uint32_t d5 = 0; uint32_t d6 = 0; uint32_t d7 = 0; double s0 = (double)(seed + threadIdx.x) + (double)0; double s1 = (double)(seed + threadIdx.x) + (double)1; double s2 = (double)(seed + threadIdx.x) + (double)2; double s3 = (double)(seed + threadIdx.x) + (double)3;
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cvt_u32_f64 %0, %8\n\t"
                     "v_cvt_u32_f64 %1, %9\n\t"
                     "v_cvt_u32_f64 %2, %10\n\t"
                     "v_cvt_u32_f64 %3, %11\n\t"
                     "v_cvt_u32_f64 %4, %8\n\t"
                     "v_cvt_u32_f64 %5, %9\n\t"
                     "v_cvt_u32_f64 %6, %10\n\t"
                     "v_cvt_u32_f64 %7, %11\n\t"
                     "v_cvt_u32_f64 %0, %8\n\t"
                     "v_cvt_u32_f64 %1, %9\n\t"
                     "v_cvt_u32_f64 %2, %10\n\t"
                     "v_cvt_u32_f64 %3, %11\n\t"
                     "v_cvt_u32_f64 %4, %8\n\t"
                     "v_cvt_u32_f64 %5, %9\n\t"
                     "v_cvt_u32_f64 %6, %10\n\t"
                     "v_cvt_u32_f64 %7, %11"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)d0 + (double)d1 + (double)d2 + (double)d3 + (double)d4 + (double)d5 + (double)d6 + (double)d7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_cvt_i32_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    int32_t d0 = 0; int32_t d1 = 0; int32_t d2 = 0; int32_t d3 = 0; int32_t d4 = 0; int32_t d5 = 0; int32_t d6 = 0; int32_t d7 = 0; double s0 = (double)(seed + threadIdx.x) + (double)0; double s1 = (double)(seed + threadIdx.x) + (double)1; double s2 = (double)(seed + threadIdx.x) + (double)2; double s3 = (double)(seed + threadIdx.x) + (double)3;
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cvt_i32_f64 %0, %8\n\t"
                     "v_cvt_i32_f64 %1, %9\n\t"
                     "v_cvt_i32_f64 %2, %10\n\t"
                     "v_cvt_i32_f64 %3, %11\n\t"
                     "v_cvt_i32_f64 %4, %8\n\t"
                     "v_cvt_i32_f64 %5, %9\n\t"
                     "v_cvt_i32_f64 %6, %10\n\t"
                     "v_cvt_i32_f64 %7, %11\n\t"
                     "v_cvt_i32_f64 %0, %8\n\t"
                     "v_cvt_i32_f64 %1, %9\n\t"
                     "v_cvt_i32_f64 %2, %10\n\t"
                     "v_cvt_i32_f64 %3, %11\n\t"
                     "v_cvt_i32_f64 %4, %8\n\t"
                     "v_cvt_i32_f64 %5, %9\n\t"
                     "v_cvt_i32_f64 %6, %10\n\t"
                     "v_cvt_i32_f64 %7, %11"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)d0 + (double)d1 + (double)d2 + (double)d3 + (double)d4 + (double)d5 + (double)d6 + (double)d7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_v_frexp_exp_i32_f64(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    int32_t d0 = 0; int32_t d1 = 0; int32_t d2 = 0; int32_t d3 = 0; int32_t d4 = 0; int32_t d5 = 0; int32_t d6 = 0; int32_t d7 = 0; double s0 = (double)(seed + threadIdx.x) + (double)0; double s1 = (double)(seed + threadIdx.x) + (double)1; double s2 = (double)(seed + threadIdx.x) + (double)2; double s3 = (double)(seed + threadIdx.x) + (double)3;
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_frexp_exp_i32_f64 %0, %8\n\t"
                     "v_frexp_exp_i32_f64 %1, %9\n\t"
                     "v_frexp_exp_i32_f64 %2, %10\n\t"
                     "v_frexp_exp_i32_f64 %3, %11\n\t"
                     "v_frexp_exp_i32_f64 %4, %8\n\t"
                     "v_frexp_exp_i32_f64 %5, %9\n\t"
                     "v_frexp_exp_i32_f64 %6, %10\n\t"
                     "v_frexp_exp_i32_f64 %7, %11\n\t"
                     "v_frexp_exp_i32_f64 %0, %8\n\t"
                     "v_frexp_exp_i32_f64 %1, %9\n\t"
                     "v_frexp_exp_i32_f64 %2, %10\n\t"
                     "v_frexp_exp_i32_f64 %3, %11\n\t"
                     "v_frexp_exp_i32_f64 %4, %8\n\t"
                     "v_frexp_exp_i32_f64 %5, %9\n\t"
                     "v_frexp_exp_i32_f64 %6, %10\n\t"
                     "v_frexp_exp_i32_f64 %7, %11"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = (double)d0 + (double)d1 + (double)d2 + (double)d3 + (double)d4 + (double)d5 + (double)d6 + (double)d7 + (double)lds_pad[0];
}

__global__ __launch_bounds__(256) void k_fma_f64_with_ds(Rec *out, double *sink, double seed)
{
    extern __shared__ char lds_pad[];
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6;
    double b = 1.0000001, c = 0.5, l0 = 0, l1 = 0;
    uint32_t addr = (threadIdx.x & 255) * 8;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(addr));
    __syncthreads();
    const uint64_t r0 = wall_clock64();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("ds_read_b64 %7, %11\n\t"
                     "v_fma_f64 %0, %0, %9, %10\n\tv_fma_f64 %1, %1, %9, %10\n\tv_fma_f64 %2, %2, %9, %10\n\t"
                     "v_fma_f64 %3, %3, %9, %10\n\tv_fma_f64 %4, %4, %9, %10\n\tv_fma_f64 %5, %5, %9, %10\n\t"
                     "v_fma_f64 %6, %6, %9, %10\n\t"
                     "ds_read_b64 %8, %11 offset:2048\n\t"
                     "v_fma_f64 %0, %0, %9, %10\n\tv_fma_f64 %1, %1, %9, %10\n\tv_fma_f64 %2, %2, %9, %10\n\t"
                     "v_fma_f64 %3, %3, %9, %10\n\tv_fma_f64 %4, %4, %9, %10\n\tv_fma_f64 %5, %5, %9, %10\n\t"
                     "v_fma_f64 %6, %6, %9, %10\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "=&v"(l0), "=&v"(l1)
                     : "v"(b), "v"(c), "v"(addr) : "memory");
    }
    stamp(out, t0, r0);
    if (seed == -1.0) sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + l0 + l1;
}

typedef void (*kern_t)(Rec *, double *, double);

struct Result {
    double cyc_span, cyc_wave, mhz;
    int simds, wmin, wmax;
};

static Result run(kern_t fn, int waves_per_simd, int cus, Rec *d_out, double *d_sink, std::vector<Rec> &host)
{
    const int blocks = cus * waves_per_simd;
    // dynamic LDS so that exactly `waves_per_simd` 256-thread blocks fit one CU (160 KiB): one wave per SIMD per block
    const size_t lds = (size_t)(160 * 1024 / waves_per_simd) - (waves_per_simd > 1 ? 512 : 0);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {                       // first launch warms the clock and the code cache
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, 0, d_out, d_sink, 1.0);
        (void)hipDeviceSynchronize();
    }
    host.resize((size_t)blocks * 4);
    (void)hipMemcpy(host.data(), d_out, sizeof(Rec) * host.size(), hipMemcpyDeviceToHost);
    std::map<uint64_t, int> per_simd;
    double ticks = 0, real = 0, tmax = 0;
    uint64_t first = ~0ull, last = 0;
    for (auto &r : host) {
        tmax = std::max(tmax, (double)r.ticks);
        per_simd[((uint64_t)r.xcc << 32) | (r.hw_id & 0xFF30u)]++;    // HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13]
        ticks += (double)r.ticks;
        real += (double)r.real;
        first = std::min(first, r.real_start);
        last = std::max(last, r.real_end);
    }
    const double n_inst = (double)ITERS * 16 * waves_per_simd;
    Result res;
    res.mhz = ticks / real * 100.0;
    res.cyc_wave = tmax / n_inst;
    res.cyc_span = (double)(last - first) * (ticks / real) / n_inst;
    res.simds = (int)per_simd.size();
    res.wmin = 1 << 30;
    res.wmax = 0;
    for (auto &kv : per_simd) {
        res.wmin = std::min(res.wmin, kv.second);
        res.wmax = std::max(res.wmax, kv.second);
    }
    return res;
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# device %s  CUs %d  nominal clock %d MHz;  ITERS %d x 16 instructions per wave, one asm block per 16\n",
           prop.gcnArchName, cus, prop.clockRate / 1000, ITERS);
    printf("# cyc(span) = chip-wide first-start..last-end x measured shader clock / (instructions per wave x W): cycles per wave64\n"
           "#             instruction per SIMD;  cyc(wave) = slowest wave's own s_memtime ticks / same;  MHz = s_memtime / s_memrealtime\n"
           "# flag: ok = every one of the %d SIMDs held exactly W waves\n", cus * 4);
    Rec *d_out;
    double *d_sink;
    (void)hipMalloc(&d_out, sizeof(Rec) * cus * 8 * 4);
    (void)hipMalloc(&d_sink, sizeof(double) * 256);
    struct K {
        const char *name;
        kern_t fn;
    };
    std::vector<K> ks = {
        {"v_fma_f64", k_v_fma_f64},
        {"v_mul_f64", k_v_mul_f64},
        {"v_add_f64", k_v_add_f64},
        {"v_max_f64", k_v_max_f64},
        {"v_ldexp_f64", k_v_ldexp_f64},
        {"v_rndne_f64", k_v_rndne_f64},
        {"v_mov_b64", k_v_mov_b64},
        {"v_cmp_gt_f64", k_v_cmp_gt_f64},
        {"v_rcp_f64", k_v_rcp_f64},
        {"v_rsq_f64", k_v_rsq_f64},
        {"v_sqrt_f64", k_v_sqrt_f64},
        {"v_frexp_mant_f64", k_v_frexp_mant_f64},
        {"v_mad_u64_u32", k_v_mad_u64_u32},
        {"v_lshrrev_b64", k_v_lshrrev_b64},
        {"v_lshlrev_b64", k_v_lshlrev_b64},
        {"v_mul_lo_u32", k_v_mul_lo_u32},
        {"v_mul_hi_u32", k_v_mul_hi_u32},
        {"v_mul_u32_u24", k_v_mul_u32_u24},
        {"v_mad_u32_u24", k_v_mad_u32_u24},
        {"v_bitop3_b32 (xor3)", k_v_bitop3_b32},
        {"v_xor_b32", k_v_xor_b32},
        {"v_and_b32", k_v_and_b32},
        {"v_add_u32", k_v_add_u32},
        {"v_add3_u32", k_v_add3_u32},
        {"v_add_co_u32", k_v_add_co_u32},
        {"v_lshrrev_b32", k_v_lshrrev_b32},
        {"v_lshl_add_u32", k_v_lshl_add_u32},
        {"v_lshl_or_b32", k_v_lshl_or_b32},
        {"v_alignbit_b32", k_v_alignbit_b32},
        {"v_bfe_u32", k_v_bfe_u32},
        {"v_perm_b32", k_v_perm_b32},
        {"v_and_or_b32", k_v_and_or_b32},
        {"v_xad_u32", k_v_xad_u32},
        {"v_mov_b32", k_v_mov_b32},
        {"v_cndmask_b32 (vcc)", k_v_cndmask_b32},
        {"v_max_u32", k_v_max_u32},
        {"v_sub_u32", k_v_sub_u32},
        {"v_fma_f32", k_v_fma_f32},
        {"v_fmac_f32 (VOP2)", k_v_fmac_f32},
        {"v_mul_f32", k_v_mul_f32},
        {"v_add_f32", k_v_add_f32},
        {"v_fmac_f64 (VOP2)", k_v_fmac_f64},
        {"v_pk_fma_f32", k_v_pk_fma_f32},
        {"v_pk_mul_f32", k_v_pk_mul_f32},
        {"v_pk_add_f32", k_v_pk_add_f32},
        {"v_cvt_f32_u32", k_v_cvt_f32_u32},
        {"v_cvt_u32_f32", k_v_cvt_u32_f32},
        {"v_exp_f32", k_v_exp_f32},
        {"v_log_f32", k_v_log_f32},
        {"v_rcp_f32", k_v_rcp_f32},
        {"v_rsq_f32", k_v_rsq_f32},
        {"v_sqrt_f32", k_v_sqrt_f32},
        {"v_sin_f32", k_v_sin_f32},
        {"v_cos_f32", k_v_cos_f32},
        {"CHAIN v_fma_f64 (dependent)", c_v_fma_f64},
        {"CHAIN v_add_u32 (dependent)", c_v_add_u32},
        {"CHAIN v_fma_f32 (dependent)", c_v_fma_f32},
        {"CHAIN v_rcp_f64 (dependent)", c_v_rcp_f64},
        {"CHAIN v_mad_u64_u32 (dependent)", c_v_mad_u64_u32},
        {"MIX 1:1 v_fma_f64 | v_add_u32", m11_add_u32},
        {"MIX 3:1 v_fma_f64 | v_add_u32", m31_add_u32},
        {"MIX 1:1 v_fma_f64 | v_bitop3_b32", m11_bitop3},
        {"MIX 3:1 v_fma_f64 | v_bitop3_b32", m31_bitop3},
        {"MIX 1:1 v_fma_f64 | v_add3_u32", m11_add3},
        {"MIX 3:1 v_fma_f64 | v_add3_u32", m31_add3},
        {"MIX 1:1 v_fma_f64 | v_fma_f32", m11_fma_f32},
        {"MIX 3:1 v_fma_f64 | v_fma_f32", m31_fma_f32},
        {"MIX 1:1 v_fma_f64 | v_pk_fma_f32", m11_pk_fma_f32},
        {"MIX 3:1 v_fma_f64 | v_pk_fma_f32", m31_pk_fma_f32},
        {"MIX 1:1 v_fma_f64 | v_mad_u64_u32", m11_mad_u64},
        {"MIX 3:1 v_fma_f64 | v_mad_u64_u32", m31_mad_u64},
        {"MIX 1:1 v_fma_f64 | v_rcp_f64", m11_rcp_f64},
        {"MIX 3:1 v_fma_f64 | v_rcp_f64", m31_rcp_f64},
        {"MIX 1:1 v_fma_f64 | v_exp_f32", m11_exp_f32},
        {"MIX 3:1 v_fma_f64 | v_exp_f32", m31_exp_f32},
        {"MIX 1:1 v_add_u32 | v_fma_f32", m11_add_u32_fma_f32},
        {"MIX 1:1 v_add_u32 | v_exp_f32", m11_add_u32_exp_f32},
        {"v_cvt_f64_u32", k_v_cvt_f64_u32},
        {"v_cvt_f64_i32", k_v_cvt_f64_i32},
        {"v_cvt_f64_f32", k_v_cvt_f64_f32},
        {"v_cvt_f32_f64", k_v_cvt_f32_f64},
        {"v_cvt_u32_f64", k_v_cvt_u32_f64},
        {"v_cvt_i32_f64", k_v_cvt_i32_f64},
        {"v_frexp_exp_i32_f64", k_v_frexp_exp_i32_f64},
        {"14 v_fma_f64 + 2 ds_read_b64", k_fma_f64_with_ds},
    };
    std::vector<Rec> host;
    printf("%-36s", "instruction stream");
    for (int w : {1, 2, 4, 8}) printf(" | W=%d span   wave   MHz   ", w);
    printf("\n");
    for (auto &k : ks) {
        printf("%-36s", k.name);
        for (int w : {1, 2, 4, 8}) {
            Result r = run(k.fn, w, cus, d_out, d_sink, host);
            const bool even = r.wmin == w && r.wmax == w && r.simds == cus * 4;
            printf(" | %7.3f %7.3f %5.0f %s", r.cyc_span, r.cyc_wave, r.mhz, even ? "ok " : "UNEVEN");
            if (!even) printf("[%d simds, %d..%d]", r.simds, r.wmin, r.wmax);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
