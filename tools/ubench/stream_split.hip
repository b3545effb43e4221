// Experiment: hide the ramp-down of the stepping launch by splitting it into S path-ranges on S streams, each
// running G time-segments back to back (state round-trips through HBM between segments).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "svmc_models.h"
#include "svmc_rng.h"
using namespace svmc;

template <bool PRIO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_sgpr(72)))
void k(double *x, double *sigma, double *qvar, size_t n, int t_begin, int t_end, LogsvFast c, uint64_t seed, uint64_t path0)
{
    __shared__ RngTablesLds s_tab;
    const RngTables tab = stage_rng_tables(s_tab);
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    double xv = x[p], s = sigma[p], q = qvar[p], L = log(s), s2 = s * s;
    const int quarter = (t_end - t_begin + 3) >> 2;
    int stage = 0, next_stage_t = t_begin;
    for (int t = t_begin; t < t_end; ++t) {
        if (PRIO && t == next_stage_t) {                   // the product kernel's least-progress-first priorities
            switch (stage++) {
            case 0: __builtin_amdgcn_s_setprio(3); break;
            case 1: __builtin_amdgcn_s_setprio(2); break;
            case 2: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
            }
            next_stage_t += quarter;
        }
        double z0, z1;
        draw_normals(seed, 0, path0 + p, t, tab, z0, z1);
        logsv_step_fast(c, xv, L, s, s2, q, z0, z1);
    }
    x[p] = xv; sigma[p] = s; qvar[p] = q;
}

int main()
{
    const size_t n = 1 << 20; const int nb = 1024;
    double *x, *s, *q;
    hipMalloc(&x, n * 8); hipMalloc(&s, n * 8); hipMalloc(&q, n * 8);
    std::vector<double> h(n, 0.8376);
    LogsvFast c = make_logsv_fast(make_logsv_consts(1.0 / 1024, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1));
    hipStream_t st[8]; for (auto &v : st) hipStreamCreateWithFlags(&v, hipStreamNonBlocking);
    hipEvent_t e0, e1, done[8]; hipEventCreate(&e0); hipEventCreate(&e1); for (auto &d : done) hipEventCreateWithFlags(&d, hipEventDisableTiming);
    const int Ss[] = {1, 2, 2, 2, 4, 4, 2, 1}, Gs[] = {1, 1, 4, 8, 4, 8, 16, 8};
    for (int cfg = 0; cfg < 8; ++cfg) {
        const int S = Ss[cfg], G = Gs[cfg];
        float best = 1e9, sum = 0;
        for (int r = 0; r < 5; ++r) {
            hipMemset(x, 0, n * 8); hipMemset(q, 0, n * 8); hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int i = 0; i < S; ++i) hipStreamWaitEvent(st[i], e0, 0);
            const size_t m = n / S;
            for (int g = 0; g < G; ++g)
                for (int i = 0; i < S; ++i)
                    hipLaunchKernelGGL(k<true>, dim3(m / 256), dim3(256), 0, st[i], x + i * m, s + i * m, q + i * m, m, g * nb / G, (g + 1) * nb / G, c, 42ull, (uint64_t)(i * m));
            for (int i = 0; i < S; ++i) { hipEventRecord(done[i], st[i]); hipStreamWaitEvent(0, done[i], 0); }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r) { best = ms < best ? ms : best; sum += ms / 4; }
        }
        double chk; hipMemcpy(&chk, x + 12345, 8, hipMemcpyDeviceToHost);
        printf("streams %d x segments %2d : avg %.3f ms  best %.3f ms   x[12345] = %.15g\n", S, G, sum, best, chk);
    }
    return 0;
}
