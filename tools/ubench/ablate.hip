// Ablation of the LogSV on-device-RNG step: where do the cycles go?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "svmc_models.h"
#include "svmc_rng.h"
using namespace svmc;

template <int MODE>
__global__ __launch_bounds__(256) void k(double *x, double *sigma, double *qvar, size_t n, int nb_steps, LogsvConsts c,
                                         uint64_t seed)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    double xv = x[p], s = sigma[p], q = qvar[p], L = log(s);
    for (int t = 0; t < nb_steps; ++t) {
        double w0, w1;
        if (MODE == 0) {  // full
            draw_normals(seed, 0, p, t, w0, w1);
            logsv_step(c, xv, L, s, q, c.sdt * w0, c.sdt * w1);
        } else if (MODE == 1) {  // step only
            w0 = 1e-3 * (double)(t & 7); w1 = -w0;
            logsv_step(c, xv, L, s, q, c.sdt * w0, c.sdt * w1);
        } else if (MODE == 2) {  // philox only
            uint32_t r[4];
            philox4x32_10((uint32_t)p, 0, t, 0, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            xv += u52(r[0], r[1]); q += u52(r[2], r[3]);
        } else if (MODE == 3) {  // philox + box-muller
            draw_normals(seed, 0, p, t, w0, w1);
            xv += w0; q += w1;
        } else if (MODE == 4) {  // philox + log + sqrt
            uint32_t r[4];
            philox4x32_10((uint32_t)p, 0, t, 0, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            xv += sqrt(-2.0 * log(u52(r[0], r[1]))); q += u52(r[2], r[3]);
        } else if (MODE == 5) {  // philox + sincospi
            uint32_t r[4];
            philox4x32_10((uint32_t)p, 0, t, 0, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            double sn, cs; sincospi(2.0 * u52(r[2], r[3]), &sn, &cs);
            xv += sn + u52(r[0], r[1]); q += cs;
        } else if (MODE == 6) {  // exp only
            L += 1e-3; s = exp(L); xv += s;
        } else if (MODE == 7) {  // div only
            s = c.k1theta / (s + 1.0); xv += s;
        }
    }
    x[p] = xv; sigma[p] = s; qvar[p] = q;
}

template <int MODE> float run(double *x, double *s, double *q, size_t n, int nb, LogsvConsts c)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(n / 256), dim3(256), 0, 0, x, s, q, n, nb, c, 42ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(n / 256), dim3(256), 0, 0, x, s, q, n, nb, c, 42ull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main()
{
    const size_t n = 1 << 20; const int nb = 1024;
    double *x, *s, *q;
    hipMalloc(&x, n * 8); hipMalloc(&s, n * 8); hipMalloc(&q, n * 8);
    hipMemset(x, 0, n * 8); hipMemset(q, 0, n * 8);
    double *h = (double *)malloc(n * 8); for (size_t i = 0; i < n; ++i) h[i] = 0.8376;
    hipMemcpy(s, h, n * 8, hipMemcpyHostToDevice);
    LogsvConsts c = make_logsv_consts(1.0 / 1024, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1);
    const char *names[] = {"full", "step only", "philox only", "philox+boxmuller", "philox+log+sqrt", "philox+sincospi", "exp only", "div only"};
    float t[8];
    t[0] = run<0>(x, s, q, n, nb, c); hipMemcpy(s, h, n * 8, hipMemcpyHostToDevice);
    t[1] = run<1>(x, s, q, n, nb, c); hipMemcpy(s, h, n * 8, hipMemcpyHostToDevice);
    t[2] = run<2>(x, s, q, n, nb, c);
    t[3] = run<3>(x, s, q, n, nb, c);
    t[4] = run<4>(x, s, q, n, nb, c);
    t[5] = run<5>(x, s, q, n, nb, c);
    t[6] = run<6>(x, s, q, n, nb, c);
    t[7] = run<7>(x, s, q, n, nb, c);
    for (int i = 0; i < 8; ++i)
        printf("%-20s %8.3f ms  %7.1f nominal cycles/wave-step  %.3e path-steps/s\n", names[i], t[i],
               t[i] * 1e-3 * 2.4e9 / (16384.0 * nb / 1024.0), (double)n * nb / (t[i] * 1e-3));
    return 0;
}
