// Experiment: how much does the placement of the stepping loop in instruction memory matter?  The same kernel body is
// instantiated with PAD extra 4-byte s_nop's ahead of the time loop (functions start 256-byte aligned), timed on the
// same box.  Motivation: two builds of libsvmc with an instruction-for-instruction identical loop differed by 7 %.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "svmc_models.h"
#include "svmc_rng.h"
using namespace svmc;

template <int PAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_sgpr(72)))
void k(double *x, double *sigma, double *qvar, size_t n, int nb, LogsvFast c, uint64_t seed)
{
    __shared__ RngTablesLds s_tab;
    const RngTables tab = stage_rng_tables(s_tab);
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    double xv = x[p], s = sigma[p], q = qvar[p], L = log(s), s2 = s * s;
    asm volatile(".rept %0\n\ts_nop 0\n\t.endr" ::"n"(PAD));
    for (int t = 0; t < nb; ++t) {
        double z0, z1;
        draw_normals(seed, 0, p, t, tab, z0, z1);
        logsv_step_fast(c, xv, L, s, s2, q, z0, z1);
    }
    x[p] = xv; sigma[p] = s; qvar[p] = q;
}

template <int PAD> void run(double *x, double *s, double *q, size_t n, int nb, LogsvFast c, const std::vector<double> &h)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 7; ++r) {
        (void)hipMemset(x, 0, n * 8); (void)hipMemset(q, 0, n * 8); (void)hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<PAD>, dim3(n / 256), dim3(256), 0, 0, x, s, q, n, nb, c, 42ull);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) { sum += ms / 5; best = ms < best ? ms : best; }
    }
    printf("pad %2d x 4 B : avg %.3f ms  best %.3f ms\n", PAD, sum, best);
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t n = 1 << 20; const int nb = 1024;
    double *x, *s, *q;
    (void)hipMalloc(&x, n * 8); (void)hipMalloc(&s, n * 8); (void)hipMalloc(&q, n * 8);
    std::vector<double> h(n, 0.8376);
    LogsvFast c = make_logsv_fast(make_logsv_consts(1.0 / 1024, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1));
    run<0>(x, s, q, n, nb, c, h); run<1>(x, s, q, n, nb, c, h); run<2>(x, s, q, n, nb, c, h); run<3>(x, s, q, n, nb, c, h);
    run<4>(x, s, q, n, nb, c, h); run<5>(x, s, q, n, nb, c, h); run<6>(x, s, q, n, nb, c, h); run<7>(x, s, q, n, nb, c, h);
    run<8>(x, s, q, n, nb, c, h); run<9>(x, s, q, n, nb, c, h); run<10>(x, s, q, n, nb, c, h); run<11>(x, s, q, n, nb, c, h);
    run<12>(x, s, q, n, nb, c, h); run<13>(x, s, q, n, nb, c, h); run<14>(x, s, q, n, nb, c, h); run<15>(x, s, q, n, nb, c, h);
    run<0>(x, s, q, n, nb, c, h);
    return 0;
}
