// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU instructions the
// svmc stepping kernels are made of.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITERS 4096
#define UNROLL 16

#define DEF_KERNEL_F64(NAME, ASM)                                                                   \
    __global__ __launch_bounds__(256) void NAME(double *out, double seed)                           \
    {                                                                                               \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
               a6 = a0 + 6, a7 = a0 + 7;                                                            \
        double b = 1.0000001, c = 0.5;                                                              \
        for (int i = 0; i < ITERS; ++i) {                                                           \
            _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {                                \
                asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c) : "vcc");                                    \
            }                                                                                       \
        }                                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                \
    }

#define DEF_KERNEL_U32(NAME, ASM)                                                                   \
    __global__ __launch_bounds__(256) void NAME(double *out, double seed)                           \
    {                                                                                               \
        uint32_t a0 = (uint32_t)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, \
                 a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                             \
        uint32_t b = 0xD2511F53u, c = 12345u;                                                       \
        for (int i = 0; i < ITERS; ++i) {                                                           \
            _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {                                \
                asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c) : "vcc");                                    \
                asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c) : "vcc");                                    \
            }                                                                                       \
        }                                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);      \
    }

#define DEF_KERNEL_U64(NAME, ASM)                                                                   \
    __global__ __launch_bounds__(256) void NAME(double *out, double seed)                           \
    {                                                                                               \
        uint64_t a0 = (uint64_t)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, \
                 a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                             \
        uint32_t b = 0xD2511F53u, c = 12345u;                                                       \
        for (int i = 0; i < ITERS; ++i) {                                                           \
            _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) {                                \
                asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c) : "vcc");                              \
                asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c) : "vcc");                              \
                asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c) : "vcc");                              \
                asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c) : "vcc");                              \
                asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c) : "vcc");                              \
                asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c) : "vcc");                              \
                asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c) : "vcc");                              \
                asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c) : "vcc");                              \
            }                                                                                       \
        }                                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);      \
    }

DEF_KERNEL_F64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
DEF_KERNEL_F64(k_add_f64, "v_add_f64 %0, %0, %1")
DEF_KERNEL_F64(k_mul_f64, "v_mul_f64 %0, %0, %1")
DEF_KERNEL_F64(k_rcp_f64, "v_rcp_f64 %0, %0")
DEF_KERNEL_F64(k_rsq_f64, "v_rsq_f64 %0, %0")
DEF_KERNEL_F64(k_sqrt_f64, "v_sqrt_f64 %0, %0")
DEF_KERNEL_F64(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1")
DEF_KERNEL_F64(k_rndne_f64, "v_rndne_f64 %0, %0")
DEF_KERNEL_F64(k_fract_f64, "v_fract_f64 %0, %0")
DEF_KERNEL_F64(k_frexp_mant_f64, "v_frexp_mant_f64 %0, %0")
DEF_KERNEL_F64(k_mov_b64, "v_mov_b64 %0, %1")
DEF_KERNEL_F64(k_cmp_cnd_f64, "v_cmp_gt_f64 vcc, %0, %1")
DEF_KERNEL_U32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL_U32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
DEF_KERNEL_U32(k_xor_b32, "v_xor_b32 %0, %0, %1")
DEF_KERNEL_U32(k_xor3_b32, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEF_KERNEL_U32(k_mov_b32, "v_mov_b32 %0, %1")
DEF_KERNEL_U32(k_add_u32, "v_add_u32 %0, %0, %1")
DEF_KERNEL_U32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL_U32(k_exp_f32, "v_exp_f32 %0, %0")
DEF_KERNEL_U32(k_log_f32, "v_log_f32 %0, %0")
DEF_KERNEL_U32(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL_U64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
DEF_KERNEL_U64(k_lshrrev_b64, "v_lshrrev_b64 %0, 3, %0")

typedef void (*kern_t)(double *, double);

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;  // Hz
    printf("device %s  CUs %d  clock %.0f MHz\n", prop.gcnArchName, cus, clk / 1e6);
    const int waves_per_simd = 8;
    const int blocks = cus * waves_per_simd;  // 256-thread blocks = 4 waves = one per SIMD
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    struct K { const char *name; kern_t fn; int per_asm; };
    std::vector<K> ks = {
        {"v_fma_f64", k_fma_f64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1},
        {"v_rcp_f64", k_rcp_f64, 1}, {"v_rsq_f64", k_rsq_f64, 1}, {"v_sqrt_f64", k_sqrt_f64, 1},
        {"v_ldexp_f64", k_ldexp_f64, 1}, {"v_rndne_f64", k_rndne_f64, 1}, {"v_fract_f64", k_fract_f64, 1},
        {"v_frexp_mant_f64", k_frexp_mant_f64, 1}, {"v_mov_b64", k_mov_b64, 1},
        {"v_cmp_gt_f64", k_cmp_cnd_f64, 1},
        {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1}, {"v_xor_b32", k_xor_b32, 1},
        {"v_bitop3_b32(xor3)", k_xor3_b32, 1}, {"v_mov_b32", k_mov_b32, 1}, {"v_add_u32", k_add_u32, 1},
        {"v_fma_f32", k_fma_f32, 1}, {"v_exp_f32", k_exp_f32, 1}, {"v_log_f32", k_log_f32, 1},
        {"v_cndmask_b32", k_cndmask_b32, 1}, {"v_mad_u64_u32", k_mad_u64_u32, 1},
        {"v_lshrrev_b64", k_lshrrev_b64, 1},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 3;
        // per SIMD: waves_per_simd waves x ITERS*UNROLL instructions
        const double inst_per_simd = (double)waves_per_simd * ITERS * UNROLL;
        const double cyc = ms * 1e-3 * clk / inst_per_simd;
        printf("%-28s %8.3f ms   %6.2f cycles / wave-instruction / SIMD (at %.0f MHz nominal)\n", k.name, ms, cyc, clk / 1e6);
    }
    return 0;
}
