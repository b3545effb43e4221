// Where does the launch of the LogSV stepping kernel lose time?  Every wave records wall_clock64() when it starts
// and when it finishes; the host prints the concurrency profile (how many waves are resident over time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include "svmc_models.h"
#include "svmc_rng.h"
using namespace svmc;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k(double *x, double *sigma, double *qvar, size_t n, int nb_steps, LogsvFast c, uint64_t seed, uint64_t *t0, uint64_t *t1)
{
    __shared__ RngTablesLds s_tab;
    const RngTables tab = stage_rng_tables(s_tab);
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t start = wall_clock64();
    double xv = x[p], s = sigma[p], q = qvar[p], L = log(s), s2 = s * s;
    for (int t = 0; t < nb_steps; ++t) {
        double z0, z1;
        draw_normals(seed, 0, p, t, tab, z0, z1);
        logsv_step_fast(c, xv, L, s, s2, q, z0, z1);
    }
    x[p] = xv; sigma[p] = s; qvar[p] = q;
    if ((threadIdx.x & 63) == 0) { t0[p >> 6] = start; t1[p >> 6] = wall_clock64(); }
}

__global__ __launch_bounds__(256)
void k2(double *x, double *sigma, double *qvar, size_t n, int nb_steps, LogsvFast c, uint64_t seed, uint64_t *t0, uint64_t *t1)
{
    __shared__ RngTablesLds s_tab;
    const RngTables tab = stage_rng_tables(s_tab);
    const size_t half = n / 2;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;      // p < half
    const size_t pb = p + half;
    const uint64_t start = wall_clock64();
    double xa = x[p], sa = sigma[p], qa = qvar[p], La = log(sa), s2a = sa * sa;
    double xb = x[pb], sb = sigma[pb], qb = qvar[pb], Lb = log(sb), s2b = sb * sb;
    for (int t = 0; t < nb_steps; ++t) {
        double z0, z1, y0, y1;
        draw_normals(seed, 0, p, t, tab, z0, z1);
        draw_normals(seed, 0, pb, t, tab, y0, y1);
        logsv_step_fast(c, xa, La, sa, s2a, qa, z0, z1);
        logsv_step_fast(c, xb, Lb, sb, s2b, qb, y0, y1);
    }
    x[p] = xa; sigma[p] = sa; qvar[p] = qa;
    x[pb] = xb; sigma[pb] = sb; qvar[pb] = qb;
    if ((threadIdx.x & 63) == 0) { t0[p >> 6] = start; t1[p >> 6] = wall_clock64(); }
}

int main()
{
    const size_t n = 1 << 20; const int nb = 1024; const size_t nw = n / 64;
    double *x, *s, *q; uint64_t *t0, *t1;
    hipMalloc(&x, n * 8); hipMalloc(&s, n * 8); hipMalloc(&q, n * 8); hipMalloc(&t0, nw * 8); hipMalloc(&t1, nw * 8);
    hipMemset(x, 0, n * 8); hipMemset(q, 0, n * 8);
    std::vector<double> h(n, 0.8376); hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice);
    LogsvFast c = make_logsv_fast(make_logsv_consts(1.0 / 1024, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1));
    for (int r = 0; r < 3; ++r) {
        hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, s, q, n, nb, c, 42ull, t0, t1);
        hipDeviceSynchronize();
    }
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms1 = 0, ms2 = 0;
        for (int r = 0; r < 4; ++r) {
            hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice);
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, s, q, n, nb, c, 42ull, t0, t1); hipEventRecord(e1); hipEventSynchronize(e1);
            float m; hipEventElapsedTime(&m, e0, e1); if (r) ms1 += m / 3;
            hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice);
            hipEventRecord(e0); hipLaunchKernelGGL(k2, dim3(n / 512), dim3(256), 0, 0, x, s, q, n, nb, c, 42ull, t0, t1); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&m, e0, e1); if (r) ms2 += m / 3;
        }
        printf("1 path/lane: %.3f ms    2 paths/lane: %.3f ms\n", ms1, ms2);
        hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, s, q, n, nb, c, 42ull, t0, t1);
        hipDeviceSynchronize();
    }
    std::vector<uint64_t> a(nw), b(nw);
    hipMemcpy(a.data(), t0, nw * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), t1, nw * 8, hipMemcpyDeviceToHost);
    const uint64_t lo = *std::min_element(a.begin(), a.end()), hi = *std::max_element(b.begin(), b.end());
    const double span = double(hi - lo);                 // 100 MHz ticks
    printf("kernel span %.3f ms (wall_clock64 at 100 MHz)\n", span / 1e5);
    // wave lifetimes
    std::vector<double> life(nw);
    for (size_t i = 0; i < nw; ++i) life[i] = double(b[i] - a[i]) / 1e5;
    std::sort(life.begin(), life.end());
    printf("wave lifetime ms: min %.3f  p10 %.3f  median %.3f  p90 %.3f  max %.3f\n", life[0], life[nw / 10], life[nw / 2], life[nw * 9 / 10], life[nw - 1]);
    // concurrency over 40 time bins
    const int B = 40; std::vector<double> occ(B, 0.0);
    for (size_t i = 0; i < nw; ++i) {
        const double s0 = double(a[i] - lo) / span * B, s1 = double(b[i] - lo) / span * B;
        for (int j = (int)s0; j <= (int)s1 && j < B; ++j) { double l = std::max(s0, (double)j), r = std::min(s1, (double)j + 1); if (r > l) occ[j] += r - l; }
    }
    printf("resident waves per SIMD over time (40 bins):\n");
    for (int j = 0; j < B; ++j) printf("%.2f ", occ[j] / 1024.0);
    printf("\n");
    // start-time distribution of the last 10%% started waves
    std::vector<double> st(nw); for (size_t i = 0; i < nw; ++i) st[i] = double(a[i] - lo) / span; std::sort(st.begin(), st.end());
    printf("wave start quantiles (fraction of span): p50 %.3f p75 %.3f p90 %.3f p99 %.3f max %.3f\n", st[nw / 2], st[nw * 3 / 4], st[nw * 9 / 10], st[nw * 99 / 100], st[nw - 1]);
    return 0;
}
