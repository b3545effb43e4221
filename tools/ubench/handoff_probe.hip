// Micro-benchmark: handing a wave's state to ANOTHER wave (possibly on another XCD) through device memory inside one
// launch -- the synchronisation skeleton of the dynamic-units generator experiment (commit e53a51a, profiles/r03_launch_tail.txt)
// without the stepping.
//
//   hipcc --offload-arch=gfx950 -O3 handoff_probe.hip -o handoff_probe && ./handoff_probe
//
// 1024 blocks x 512 threads (= every wave slot of the chip); units (group, segment) are handed out by an atomic ticket
// counter, segment-major; a unit waits for the flag of its group, loads three doubles per lane, runs WORK dependent FMAs,
// stores them back and publishes the flag.  The final value of every lane is known in closed form, so stale reads show.
// MODE 0: plain loads / stores + agent-scope release / acquire FENCES (buffer_wbl2 sc1 / buffer_inv sc1)
// MODE 1: no fences: every state access is an agent-scope relaxed atomic (sc1), ordered before the flag by s_waitcnt
// MODE 2: as 1, groups pinned to the XCD that ran their first segment's queue (tickets per XCC_ID)
// MODE 3: baseline -- the same work as static units (wave w runs units w, w + 8192, ...), no tickets, no hand-off
// The host watches the launch: after 5 s it prints the progress counters from a side stream and exits.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(2); } } while (0)
#define HW_REG_XCC_ID 20

struct Args {
    unsigned long long *ctl;    // [0..7]: tickets (per XCD in MODE 2, stride 32), [256 ...): flags
    double *a, *b, *c;
    unsigned long long *rec;    // [8192][4]: xcc, units, first start, last end (100 MHz)
    uint32_t groups, segments, work;
};

constexpr int FLAGS_AT = 512;

template <int MODE>
__global__ __launch_bounds__(512) void handoff_kernel(Args A)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t xcc = 0, nq = 1;
    if (MODE == 2) {
        xcc = __builtin_amdgcn_s_getreg((HW_REG_XCC_ID) | (0 << 6) | ((4 - 1) << 11)) & 7u;
        nq = 8;
    }
    const uint32_t my_groups = (A.groups - xcc + nq - 1) / nq;          // groups g = xcc + nq * j
    const uint32_t total = my_groups * A.segments;
    const uint32_t wave = blockIdx.x * 8u + (threadIdx.x >> 6);
    const unsigned long long t_start = wall_clock64();
    unsigned long long units = 0;
    const uint32_t my_xcc = __builtin_amdgcn_s_getreg((HW_REG_XCC_ID) | (0 << 6) | ((4 - 1) << 11)) & 7u;
    for (uint32_t it = 0;; ++it) {
        if (lane == 0u) {
            A.rec[4 * wave + 0] = my_xcc; A.rec[4 * wave + 1] = units; A.rec[4 * wave + 2] = t_start; A.rec[4 * wave + 3] = wall_clock64();
        }
        ++units;
        if (MODE == 3) {                                   // baseline: static units, no tickets, no hand-off
            const uint32_t w = blockIdx.x * 8u + (threadIdx.x >> 6), u = w + it * 8192u;
            if (u >= A.groups * A.segments) break;
            double x = 1.0, y = 0.0;
            for (uint32_t i = 0; i < A.work; ++i) y = __builtin_fma(x, 1.0, y);
            const size_t p = static_cast<size_t>(u % A.groups) * 64u + lane;
            A.a[p] = x; A.b[p] = y * A.segments; A.c[p] = static_cast<double>(p);
            continue;
        }
        // one atomic by lane 0 with exec narrowed in the asm: `if (lane == 0) atomicAdd` + readfirstlane is mis-threaded by the
        // optimiser (the wave never leaves unit 0), an add by all lanes is left as 64 serialised atomics per ticket
        unsigned long long old, saved;
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add_x2 %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(old), "=&s"(saved) : "v"(0u), "v"(1ull), "s"(A.ctl + 32 * xcc) : "memory");
        const uint32_t u = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(old));
        if (u >= total) break;
        const uint32_t k = u / my_groups, j = u - k * my_groups, g = xcc + nq * j;
        const size_t p = static_cast<size_t>(g) * 64u + lane;
        if (k != 0u) {
            while (__hip_atomic_load(A.ctl + FLAGS_AT + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != k) __builtin_amdgcn_s_sleep(32);
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        double x, y, z;
        if (k == 0u) {
            x = 1.0; y = 0.0; z = static_cast<double>(p);
        } else if (MODE == 0) {
            x = A.a[p]; y = A.b[p]; z = A.c[p];
        } else {
            x = __hip_atomic_load(A.a + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            y = __hip_atomic_load(A.b + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            z = __hip_atomic_load(A.c + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (uint32_t i = 0; i < A.work; ++i) y = __builtin_fma(x, 1.0, y);       // y += 1 per iteration, dependent chain
        x = x + 0.0;
        if (MODE == 0) {
            A.a[p] = x; A.b[p] = y; A.c[p] = z;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        } else {
            __hip_atomic_store(A.a + p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.b + p, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.c + p, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // s_waitcnt vmcnt(0): the stores are acknowledged
        }
        if (lane == 0u) {
            __hip_atomic_store(A.ctl + FLAGS_AT + g, static_cast<unsigned long long>(k + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(A.ctl + 300, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // units done
        }
    }
}

template <int MODE>
static void run(uint32_t groups, uint32_t segments, uint32_t work)
{
    const size_t n = static_cast<size_t>(groups) * 64;
    Args A;
    const size_t ctl_words = FLAGS_AT + groups;
    CHECK(hipMalloc(reinterpret_cast<void **>(&A.ctl), ctl_words * 8));
    CHECK(hipMalloc(reinterpret_cast<void **>(&A.a), n * 8));
    CHECK(hipMalloc(reinterpret_cast<void **>(&A.b), n * 8));
    CHECK(hipMalloc(reinterpret_cast<void **>(&A.c), n * 8));
    CHECK(hipMalloc(reinterpret_cast<void **>(&A.rec), 4 * 8192 * 8));
    A.groups = groups; A.segments = segments; A.work = work;
    hipStream_t main_s, side;
    CHECK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemsetAsync(A.ctl, 0, ctl_words * 8, main_s));
        CHECK(hipEventRecord(e0, main_s));
        hipLaunchKernelGGL(handoff_kernel<MODE>, dim3(1024), dim3(512), 0, main_s, A);
        CHECK(hipEventRecord(e1, main_s));
        for (int i = 0; hipStreamQuery(main_s) != hipSuccess; ++i) {
            usleep(1000);
            if (i > 5000) {
                unsigned long long h[304];
                CHECK(hipMemcpyAsync(h, A.ctl, sizeof(h), hipMemcpyDeviceToHost, side));
                CHECK(hipStreamSynchronize(side));
                printf("MODE %d groups %u segments %u: STUCK after 5 s: tickets %llu units done %llu of %u\n", MODE, groups, segments,
                       h[0], h[300], groups * segments);
                fflush(stdout);
                _exit(3);
            }
        }
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    {
        std::vector<unsigned long long> r(4 * 8192);
        CHECK(hipMemcpy(r.data(), A.rec, r.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0, per_xcc[8] = {}, hist[16] = {}, waves_xcc[8] = {}, end_xcc[8] = {};
        for (int w = 0; w < 8192; ++w) {
            t0 = r[4 * w + 2] < t0 ? r[4 * w + 2] : t0;
            t1 = r[4 * w + 3] > t1 ? r[4 * w + 3] : t1;
        }
        for (int w = 0; w < 8192; ++w) {
            per_xcc[r[4 * w] & 7] += r[4 * w + 1];
            waves_xcc[r[4 * w] & 7]++;
            hist[r[4 * w + 1] < 15 ? r[4 * w + 1] : 15]++;
            if (r[4 * w + 3] - t0 > end_xcc[r[4 * w] & 7]) end_xcc[r[4 * w] & 7] = r[4 * w + 3] - t0;
        }
        printf("   span %.1f us; units per XCD:", (t1 - t0) / 100.0);
        for (int x = 0; x < 8; ++x) printf(" %llu(%llu w, end %.0f us)", per_xcc[x], waves_xcc[x], end_xcc[x] / 100.0);
        printf("\n   waves by units done:");
        for (int h = 0; h < 16; ++h) if (hist[h]) printf(" %d:%llu", h, hist[h]);
        printf("\n");
    }
    std::vector<double> a(n), b(n), c(n);
    CHECK(hipMemcpy(a.data(), A.a, n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), A.b, n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(c.data(), A.c, n * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t p = 0; p < n; ++p)
        if (a[p] != 1.0 || b[p] != static_cast<double>(work) * segments || c[p] != static_cast<double>(p)) ++bad;
    printf("MODE %d groups %u segments %u work %u: %.3f ms  (%.2f us per unit-slot), wrong lanes %zu\n", MODE, groups, segments, work,
           best, best * 1e3 * 8192.0 / (static_cast<double>(groups) * segments), bad);
    fflush(stdout);
    CHECK(hipFree(A.ctl)); CHECK(hipFree(A.a)); CHECK(hipFree(A.b)); CHECK(hipFree(A.c));
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t work = argc > 2 ? atoi(argv[2]) : 4000;
    for (uint32_t segments : {1u, 4u}) {
        if (mode == 0) run<0>(16384, segments, work);
        if (mode == 1) run<1>(16384, segments, work);
        if (mode == 2) run<2>(16384, segments, work);
        if (mode == 3) run<3>(16384, segments, work);
    }
    return 0;
}
