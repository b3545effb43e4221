// Round 6 micro-benchmark: what a two-cycle (32-bit) VALU instruction costs inside a stream of four-cycle (fp64) ones, by how the
// two kinds are ARRANGED in each wave's stream.  profiles/r03_valu_rates.txt has the alternating mixes (a v_add_u32 costs ~3.7
// cycles, not 2.2, beside v_fma_f64); this asks whether runs of 32-bit instructions pair up across the waves of a SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 tools/r06/issue_mix.hip -o /tmp/issue_mix && /tmp/issue_mix
//
// 256-thread blocks (one wave per SIMD), dynamic LDS sized so that W blocks fit a CU, 256 x W blocks.  A wave runs ITERS x 16
// instructions (one asm block, independent accumulators) between s_memtime / s_memrealtime reads; printed: chip-wide span x measured
// shader clock per instruction per wave-slot, i.e. SIMD cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITERS 4096
#define F(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n\t"
#define I(n) "v_add_u32 %" #n ", %" #n ", %10\n\t"
#define B(n) "v_bitop3_b32 %" #n ", %" #n ", %10, %11 bitop3:0x96\n\t"
#define M(n) "v_mad_u64_u32 %" #n ", vcc, %10, %11, %" #n "\n\t"

#define KERNEL(name, BODY)                                                                                                  \
    __global__ __launch_bounds__(256) void name(uint64_t *out, double *sink, double seed)                                    \
    {                                                                                                                        \
        extern __shared__ char lds_pad[];                                                                                    \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;                                               \
        uint32_t i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;                                                    \
        double b = 1.0000001, c = 0.5;                                                                                       \
        uint32_t d = 0x9E3779B9u, e = 12345u;                                                                                \
        asm volatile("" : "+v"(b), "+v"(c), "+v"(d), "+v"(e));                                                               \
        __syncthreads();                                                                                                     \
        const uint64_t r0 = wall_clock64();                                                                                  \
        const uint64_t t0 = __builtin_readcyclecounter();                                                                    \
        for (int i = 0; i < ITERS; ++i) {                                                                                    \
            asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3)               \
                         : "v"(b), "v"(c), "v"(d), "v"(e) : "vcc");                                                          \
        }                                                                                                                    \
        const uint64_t t1 = __builtin_readcyclecounter();                                                                    \
        const uint64_t r1 = wall_clock64();                                                                                  \
        if ((threadIdx.x & 63) == 0) {                                                                                       \
            uint64_t *o = out + 3 * (blockIdx.x * 4 + (threadIdx.x >> 6));                                                   \
            o[0] = t1 - t0;                                                                                                  \
            o[1] = r0;                                                                                                       \
            o[2] = r1;                                                                                                       \
        }                                                                                                                    \
        if (seed == -1.0) sink[threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3 + lds_pad[0];                            \
    }

// 8 fp64 + 8 int32 per block, arranged four ways
KERNEL(alt_1_1, F(0) I(4) F(1) I(5) F(2) I(6) F(3) I(7) F(0) I(4) F(1) I(5) F(2) I(6) F(3) I(7))
KERNEL(pairs_2_2, F(0) F(1) I(4) I(5) F(2) F(3) I(6) I(7) F(0) F(1) I(4) I(5) F(2) F(3) I(6) I(7))
KERNEL(runs_4_4, F(0) F(1) F(2) F(3) I(4) I(5) I(6) I(7) F(0) F(1) F(2) F(3) I(4) I(5) I(6) I(7))
KERNEL(runs_8_8, F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) I(4) I(5) I(6) I(7) I(4) I(5) I(6) I(7))
// 12 fp64 + 4 int32
KERNEL(alt_3_1, F(0) F(1) F(2) I(4) F(3) F(0) F(1) I(5) F(2) F(3) F(0) I(6) F(1) F(2) F(3) I(7))
KERNEL(runs_12_4, F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) I(4) I(5) I(6) I(7))
// the Philox round's shape: two 64-bit multiplies, two three-operand xors (x4)
KERNEL(philox_like, F(0) F(1) B(4) B(5) F(2) F(3) B(6) B(7) F(0) F(1) B(4) B(5) F(2) F(3) B(6) B(7))
// the LogSV loop's proportions (per two steps 67 fp64-class : 29 32-bit): 11 : 5 per block, scattered or in one run
KERNEL(real_scattered, F(0) F(1) I(4) F(2) F(3) I(5) F(0) F(1) I(6) F(2) F(3) I(7) F(0) F(1) I(4) F(2))
KERNEL(real_pairs, F(0) F(1) F(2) F(3) I(4) I(5) F(0) F(1) F(2) F(3) I(6) I(7) F(0) F(1) F(2) I(4))
KERNEL(real_run, F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) F(0) F(1) F(2) I(4) I(5) I(6) I(7) I(4))
KERNEL(all_f, F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3))
KERNEL(all_i, I(4) I(5) I(6) I(7) I(4) I(5) I(6) I(7) I(4) I(5) I(6) I(7) I(4) I(5) I(6) I(7))

struct Test {
    const char *name;
    void (*k)(uint64_t *, double *, double);
    int n_f, n_i;
};

int main()
{
    const Test tests[] = {{"all fp64", all_f, 16, 0},           {"all int32", all_i, 0, 16},
                          {"alternating 1:1", alt_1_1, 8, 8},   {"pairs 2:2", pairs_2_2, 8, 8},
                          {"runs 4:4", runs_4_4, 8, 8},         {"runs 8:8", runs_8_8, 8, 8},
                          {"alternating 3:1", alt_3_1, 12, 4},  {"runs 12:4", runs_12_4, 12, 4},
                          {"philox-like 2 fp64 : 2 bitop3", philox_like, 8, 8},
                          {"11:5 scattered", real_scattered, 11, 5},
                          {"11:5 in pairs", real_pairs, 11, 5},
                          {"11:5 one run", real_run, 11, 5}};
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint64_t *out;
    double *sink;
    hipMalloc(&out, sizeof(uint64_t) * cus * 8 * 4 * 3);
    hipMalloc(&sink, 4096);
    printf("# %s, %d CUs; SIMD cycles per instruction (first start .. last end on the 100 MHz counter x measured shader clock / (ITERS x 16 x W)); in brackets: what a\n"
           "# 32-bit instruction costs if an fp64 one costs what 'all fp64' shows at the same W\n", prop.gcnArchName, cus);
    printf("%-34s | %-18s | %-18s | %-18s | %-18s\n", "stream", "W=1", "W=2", "W=4", "W=8");
    double f_cost[4] = {0, 0, 0, 0};
    for (const Test &t : tests) {
        printf("%-34s", t.name);
        int wi = 0;
        for (int W : {1, 2, 4, 8}) {
            const size_t lds = (size_t)(160 * 1024 / W) - 2048;
            hipFuncSetAttribute((const void *)t.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const int blocks = cus * W;
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), lds, 0, out, sink, 1.0);
                hipDeviceSynchronize();
            }
            std::vector<uint64_t> h(blocks * 4 * 3);
            hipMemcpy(h.data(), out, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost);
            // span: first start .. last end on the chip-wide 100 MHz counter, times the shader clock the waves themselves measured
            uint64_t first = ~0ull, last = 0;
            double ticks = 0, real = 0;
            for (int w = 0; w < blocks * 4; ++w) {
                first = std::min(first, h[3 * w + 1]);
                last = std::max(last, h[3 * w + 2]);
                ticks += (double)h[3 * w];
                real += (double)(h[3 * w + 2] - h[3 * w + 1]);
            }
            const double cyc = (double)(last - first) * (ticks / real) / ((double)ITERS * 16.0 * W);
            if (t.n_i == 0) f_cost[wi] = cyc;
            if (t.n_f > 0 && t.n_i > 0)
                printf(" | %6.3f  [%5.2f]   ", cyc, (cyc * 16.0 - f_cost[wi] * t.n_f) / t.n_i);
            else
                printf(" | %6.3f            ", cyc);
            ++wi;
        }
        printf("\n");
    }
    return 0;
}
