// Ablation of the LogSV on-device-RNG step: where do the cycles go?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "svmc_models.h"
#include "svmc_rng.h"
using namespace svmc;

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k(double *x, double *sigma,
                                                                                    double *qvar, size_t n,
                                                                                    int nb_steps, LogsvFast c,
                                                                                    uint64_t seed)
{
    __shared__ RngTablesLds s_tab;
    const RngTables tab = stage_rng_tables(s_tab);
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    double xv = x[p], s = sigma[p], q = qvar[p], L = log(s), s2 = s * s;
    for (int t = 0; t < nb_steps; ++t) {
        double w0, w1;
        uint32_t r[4];
        if (MODE == 0) {  // the shipped step: draw + logsv_step_fast
            draw_normals(seed, 0, p, t, tab, w0, w1);
            logsv_step_fast(c, xv, L, s, s2, q, w0, w1);
        } else if (MODE == 1) {  // step only
            w0 = 1e-3 * (double)(t & 7); w1 = -w0;
            logsv_step_fast(c, xv, L, s, s2, q, w0, w1);
        } else if (MODE == 2) {  // philox + mantissa conversion only
            philox_draw(seed, 0, p, t, r);
            xv += mantissa_1_2(r[0], r[1]); q += mantissa_1_2(r[2], r[3]);
        } else if (MODE == 3) {  // philox + box-muller
            draw_normals(seed, 0, p, t, tab, w0, w1);
            xv += w0; q += w1;
        } else if (MODE == 4) {  // philox + table log + sqrt
            philox_draw(seed, 0, p, t, r);
            const double e = neg_log_tab<-32>(static_cast<double>(r[0]) + 0.5, tab.log);      // stream v2's radius
            xv += sqrt_pos_1g(e); q += mantissa_1_2(r[2], r[3]);
        } else if (MODE == 5) {  // philox + the table-assisted direction
            philox_draw(seed, 0, p, t, r);
            double sn, cs; cossin_circle_tab32(r[1], tab.circle, cs, sn);                    // stream v2's direction
            xv += sn + mantissa_1_2(r[0], r[1]); q += cs;
        } else if (MODE == 6) {  // exp only
            L += 1e-3; s = exp_fast(L); xv += s;
        } else if (MODE == 7) {  // reciprocal only
            s = rcp_fast(s + 1.0); xv += s;
        }
    }
    x[p] = xv; sigma[p] = s; qvar[p] = q;
}

template <int MODE> float run(double *x, double *s, double *q, size_t n, int nb, LogsvFast c)
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
    const LogsvFast c = make_logsv_fast(make_logsv_consts(1.0 / 1024, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1));
    const char *names[] = {"full", "step only", "philox only", "philox+boxmuller", "philox+log+sqrt", "philox+sincos", "exp only", "rcp only"};
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
