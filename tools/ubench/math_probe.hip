// Device probe: accuracy of the v_rcp_f64 / v_rsq_f64 seeds and of svmc_math.h vs the device libm, and
// the issue cost of each function (nominal cycles per wave-call at 8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
#include "svmc_rng.h"      // svmc_math.h + the tables of the draw
using namespace svmc;

__global__ void acc_kernel(const double *x, double *out, int n, int what)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i], r = 0;
    switch (what) {
    case 0: r = fabs(__builtin_amdgcn_rcp(v) * v - 1.0); break;                       // rcp seed rel err
    case 1: { double y = __builtin_amdgcn_rsq(v); r = fabs(y * y * v - 1.0) * 0.5; } break;  // rsq seed rel err
    case 2: { double a = exp_fast(v), b = exp(v); r = fabs(a - b) / fabs(b); } break;
    case 3: { double a = neg_log(v), b = -log(v); r = fabs(a - b) / fabs(b); } break;
    case 4: { double a = sqrt_pos(v), b = sqrt(v); r = fabs(a - b) / fabs(b); } break;
    case 5: { double a = rcp_fast(v), b = 1.0 / v; r = fabs(a - b) / fabs(b); } break;
    case 6: {       // stream v2's direction from a 32-bit angle word against OCML sincospi, v in [0, 1) turns
        __shared__ CircleTabEntry tab[256];
        for (unsigned k = threadIdx.x; k < 256u; k += blockDim.x) tab[k] = g_circle_table[k];
        __syncthreads();
        const uint32_t w = static_cast<uint32_t>(v * 4294967296.0);
        double a, b, s2, c2;
        cossin_circle_tab32(w, tab, a, b);
        sincospi(2.0 * ((static_cast<double>(w) + 0.5) * 0x1.0p-32), &s2, &c2);
        r = fmax(fabs(a - 1.4142135623730951 * c2), fabs(b - 1.4142135623730951 * s2));
    } break;
    case 7: { double y = __builtin_amdgcn_sqrt(v); r = fabs(y * y / v - 1.0) * 0.5; } break;      // v_sqrt_f64 rel err
    }
    out[i] = r;
}

template <int WHAT>
__global__ __launch_bounds__(256) void time_kernel(double *out, double seed, int iters)
{
    double a = seed + 1e-3 * threadIdx.x, acc = 0;
    for (int i = 0; i < iters; ++i) {
        double v = a + 1e-6 * i;
        if (WHAT == 0) acc += exp(v);
        if (WHAT == 1) acc += exp_fast(v);
        if (WHAT == 2) acc += log(v);
        if (WHAT == 3) acc += neg_log(v);
        if (WHAT == 4) acc += sqrt(v);
        if (WHAT == 5) acc += sqrt_pos(v);
        if (WHAT == 6) acc += 1.0 / v;
        if (WHAT == 7) acc += rcp_fast(v);
        if (WHAT == 8) { double s, c; sincospi(v, &s, &c); acc += s + c; }
        if (WHAT == 9) {
            __shared__ CircleTabEntry tab[256];
            if (i == 0) {
                for (unsigned k = threadIdx.x; k < 256u; k += blockDim.x) tab[k] = g_circle_table[k];
                __syncthreads();
            }
            double s, c;
            cossin_circle_tab32(static_cast<uint32_t>(v * 1e6), tab, c, s);
            acc += s + c;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int W> float timeit(double *out, int blocks, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(time_kernel<W>, dim3(blocks), dim3(256), 0, 0, out, 0.3, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(time_kernel<W>, dim3(blocks), dim3(256), 0, 0, out, 0.3, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main()
{
    const int n = 1 << 22;
    std::vector<double> h(n), r(n);
    std::mt19937_64 g(1);
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 8);
    struct T { const char *name; int what; double lo, hi; bool logscale; };
    std::vector<T> tests = {
        {"rcp seed rel err, x in [0.5,4]", 0, 0.5, 4, false}, {"rsq seed rel err, x in [1e-6,80]", 1, 1e-6, 80, true},
        {"v_sqrt_f64 rel err", 7, 1e-6, 80, true},
        {"exp_fast vs OCML exp rel, x in [-20,20]", 2, -20, 20, false}, {"neg_log vs OCML rel, u in (0,1)", 3, 1e-16, 1, true},
        {"sqrt_pos vs OCML rel", 4, 1e-12, 80, true}, {"rcp_fast vs IEEE div rel", 5, 0.01, 100, true},
        {"cossin_circle_tab32 vs OCML sincospi abs", 6, 0.0, 0.999999, false}};
    for (auto &t : tests) {
        for (int i = 0; i < n; ++i) {
            double u = (g() >> 11) * 0x1.0p-53;
            h[i] = t.logscale ? std::exp(std::log(t.lo) + u * (std::log(t.hi) - std::log(t.lo))) : t.lo + u * (t.hi - t.lo);
        }
        hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(acc_kernel, dim3(n / 256), dim3(256), 0, 0, dx, dout, n, t.what);
        hipMemcpy(r.data(), dout, n * 8, hipMemcpyDeviceToHost);
        double mx = 0; for (int i = 0; i < n; ++i) mx = std::fmax(mx, r[i]);
        printf("%-45s max %.3e  (= 2^%.1f)\n", t.name, mx, std::log2(mx));
    }
    const int blocks = 256 * 8, iters = 4096;
    const char *names[] = {"OCML exp", "exp_fast", "OCML log", "neg_log", "OCML sqrt", "sqrt_pos", "IEEE div", "rcp_fast", "OCML sincospi", "cossin_circle_tab32"};
    float ms[10] = {timeit<0>(dout, blocks, iters), timeit<1>(dout, blocks, iters), timeit<2>(dout, blocks, iters), timeit<3>(dout, blocks, iters),
                    timeit<4>(dout, blocks, iters), timeit<5>(dout, blocks, iters), timeit<6>(dout, blocks, iters), timeit<7>(dout, blocks, iters),
                    timeit<8>(dout, blocks, iters), timeit<9>(dout, blocks, iters)};
    for (int i = 0; i < 10; ++i) printf("%-16s %7.3f ms  %7.1f nominal cycles / wave-call\n", names[i], ms[i], ms[i] * 1e-3 * 2.4e9 / (8.0 * iters));
    return 0;
}
