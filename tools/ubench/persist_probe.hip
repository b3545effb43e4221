// Experiment: a persistent stepping kernel.  2048 resident blocks pull (time-segment, path-block) tasks from an atomic
// counter in segment-major order; a task waits on the completion flag of the same path-block's previous segment
// (always handed out earlier, so always running or done: no deadlock), the state round-trips through HBM between
// segments.  Compared with the plain launch (one block per path-block, whole time range) for time and bit-equality.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <cstdint>
#include <vector>
#include "svmc_models.h"
#include "svmc_rng.h"
using namespace svmc;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_plain(double *x, double *sigma, double *qvar, size_t n, int nb, LogsvFast c, uint64_t seed)
{
    __shared__ RngTablesLds s_tab;
    const RngTables tab = stage_rng_tables(s_tab);
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    double xv = x[p], s = sigma[p], q = qvar[p], L = log(s), s2 = s * s;
    for (int t = 0; t < nb; ++t) {
        double z0, z1;
        draw_normals(seed, 0, p, t, tab, z0, z1);
        logsv_step_fast(c, xv, L, s, s2, q, z0, z1);
    }
    x[p] = xv; sigma[p] = s; qvar[p] = q;
}

// agent-coherent (sc1) accesses: visible across XCDs without the L2 write-back / invalidate an agent-scope fence costs
__device__ __forceinline__ double ld_agent(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// One atomic per wave, issued by lane 0 with the exec mask narrowed INSIDE one asm block.  Written as
// `if (lane == 0) atomicAdd(...)` + readfirstlane the loop hung: with a second lane-0-only statement at the end of the
// loop body the structurizer routed lanes 1..63 to the readfirstlane without lane 0, they read their own 0 and never
// saw the end of the queue.
__device__ __forceinline__ unsigned wave_grab(unsigned *counter)
{
    unsigned ret, one = 1u;
    unsigned long long saved;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b64 exec, 1\n\t"
                 "global_atomic_add %[r], %[a], %[o], off sc0\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [r] "=&v"(ret), [sv] "=&s"(saved)
                 : [a] "v"(counter), [o] "v"(one)
                 : "memory");
    return __builtin_amdgcn_readfirstlane(ret);
}

// ctl[0] = task counter, ctl[1 + s * W + w] = done flag of task (segment s, wave-block w of 64 paths).
// Tasks are per WAVE: no barrier inside the loop (a block-level version needs __syncthreads() after a lane-0-only
// atomic, and the structurizer duplicated that barrier into a divergent path: deadlock).
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_persist(double *x, double *sigma, double *qvar, size_t n, int nb, int seg, LogsvFast c, uint64_t seed,
               unsigned *ctl, unsigned long long max_spin)
{
    __shared__ RngTablesLds s_tab;
    const RngTables tab = stage_rng_tables(s_tab);
    const unsigned W = (unsigned)((n + 63) / 64), S = (unsigned)((nb + seg - 1) / seg), T = W * S;
    const unsigned lane = threadIdx.x & 63;
    for (;;) {
        const unsigned task = wave_grab(&ctl[0]);
        if (task >= T) break;
        const unsigned sidx = task / W, w = task - sidx * W;
        if ((MODE & 1) && sidx > 0) {
            unsigned long long spins = 0;
            while (__hip_atomic_load(&ctl[1 + (sidx - 1) * W + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > max_spin) break;              // safety net for the experiment: never hang the box
            }
            if (spins && lane == 0) { atomicAdd(&ctl[T + 1], 1u); atomicAdd(&ctl[T + 2], (unsigned)spins); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // keeps the state loads after the flag loop
        const size_t p = (size_t)w * 64 + lane;
        if (p < n) {
            double xv = ld_agent(x + p), s = ld_agent(sigma + p), q = ld_agent(qvar + p), L = log(s), s2 = s * s;
            const int t1 = ((int)(sidx + 1) * seg < nb) ? (int)(sidx + 1) * seg : nb;
            for (int t = (int)sidx * seg; t < t1; ++t) {
                if ((MODE & 4) && (t & 31) == 0) {          // fairness: the further behind the queue head, the higher
                    const unsigned lag = __builtin_amdgcn_readfirstlane(
                        __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - task;
                    const unsigned gen = gridDim.x * 4u;    // resident waves
                    if (lag > gen + gen / 2) __builtin_amdgcn_s_setprio(3);
                    else if (lag > gen + gen / 8) __builtin_amdgcn_s_setprio(2);
                    else if (lag > gen - gen / 4) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
                double z0, z1;
                draw_normals(seed, 0, p, t, tab, z0, z1);
                logsv_step_fast(c, xv, L, s, s2, q, z0, z1);
            }
            st_agent(x + p, xv); st_agent(sigma + p, s); st_agent(qvar + p, q);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // s_waitcnt vmcnt(0): the sc1 stores are acknowledged
        if (MODE & 2)                                               // every lane stores the same word: one write
            __hip_atomic_store(&ctl[1 + sidx * W + w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t n = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 20); const int nb = 1024;
    const unsigned long long max_spin = argc > 2 ? strtoull(argv[2], nullptr, 10) : 20000ull;
    double *x, *s, *q; unsigned *ctl;
    hipMalloc(&x, n * 8); hipMalloc(&s, n * 8); hipMalloc(&q, n * 8);
    const size_t ctl_n = 4 + 64 * (n / 64);
    hipMalloc(&ctl, ctl_n * 4);
    std::vector<double> h(n, 0.8376), ref(n), got(n);
    LogsvFast c = make_logsv_fast(make_logsv_consts(1.0 / 1024, 1.0413, 3.1844, 3.058, 0.1514, 1.8458, 1.0, 1));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto reset = [&] { hipMemset(x, 0, n * 8); hipMemset(q, 0, n * 8); hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice); hipDeviceSynchronize(); };
    float sum = 0, best = 1e9;
    for (int r = 0; r < 5; ++r) {
        reset();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_plain, dim3(n / 256), dim3(256), 0, 0, x, s, q, n, nb, c, 42ull);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r) { sum += ms / 4; best = ms < best ? ms : best; }
    }
    hipMemcpy(ref.data(), x, n * 8, hipMemcpyDeviceToHost);
    printf("plain launch                      : avg %.3f ms  best %.3f ms\n", sum, best);
    auto run = [&](auto kernel, const char *what, int grid, int seg) {
        sum = 0; best = 1e9;
        for (int r = 0; r < 4; ++r) {
            reset();
            hipEventRecord(e0, 0);
            hipMemsetAsync(ctl, 0, ctl_n * 4, 0);
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, x, s, q, n, nb, seg, c, 42ull, ctl, max_spin);
            hipEventRecord(e1, 0);
            for (int w = 0; hipEventQuery(e1) != hipSuccess; ++w) {      // watchdog: give up after ~3 s
                if (w > 3000) { printf("%s grid %d seg %d: HUNG\n", what, grid, seg); _exit(3); }
                usleep(1000);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r) { sum += ms / 3; best = ms < best ? ms : best; }
        }
        hipMemcpy(got.data(), x, n * 8, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += (got[i] != ref[i]);
        unsigned st[2]; const size_t Tt = (n / 64) * ((nb + seg - 1) / seg);
        hipMemcpy(st, ctl + Tt + 1, 8, hipMemcpyDeviceToHost);
        printf("%s grid %4d, segment %4d : avg %.3f ms  best %.3f ms   mismatches %zu  waited tasks %u spins %u\n", what, grid, seg, sum, best, bad, st[0], st[1]);
    };
    run(k_persist<0>, "no flags (1 segment)   ", 2048, 1024);
    run(k_persist<2>, "publish only (1 segment)", 2048, 1024);
    run(k_persist<2>, "publish only            ", 2048, 256);
    const int segs[] = {512, 256, 128, 64};
    for (int seg : segs) run(k_persist<3>, "publish + wait          ", 2048, seg);
    for (int seg : segs) run(k_persist<7>, "publish + wait + fair   ", 2048, seg);
    run(k_persist<4>, "fair only (1 segment)   ", 2048, 1024);
    return 0;
}
