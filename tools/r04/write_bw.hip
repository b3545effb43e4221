// write_bw.hip -- what bounds logsv_vol_paths_kernel's 8.6 GB of stores?  Pure-store kernels, no arithmetic, on the kernel's own
// output shape ([nb + 1][n] doubles, n = 2^20, nb = 1024):
//   memset        hipMemsetAsync of the buffer (the runtime's fill kernel)
//   linear16[nt]  every thread stores 16 B, consecutive threads consecutive addresses, grid-stride: the ideal streaming write
//   walk8[nt]     the vol-paths pattern: a wave owns 64 columns and walks the rows, 8 B per lane per row (512 B per wave-store)
//   walk16[nt]    the same with 16-byte stores: lanes 0-31 row t, lanes 32-63 row t + 1 (round 4's kernel)
//   walkblk[nt]   1024-thread blocks walking in lockstep (a barrier per two rows): 8 KB contiguous per row at a time
//   walk8x4       8 B per lane, four rows' stores issued back to back (deeper store queue per wave)
// hipcc --offload-arch=gfx950 -O3 tools/r04/write_bw.hip -o tools/r04/write_bw && tools/r04/write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT, class T> __device__ __forceinline__ void st(T *p, T v)
{
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <int POL, int BLK> __global__ __launch_bounds__(BLK) void walk16pol(double *out, size_t ld, int rows)
{
    const unsigned lane = threadIdx.x & 63u;
    const size_t p0 = (size_t)blockIdx.x * BLK + threadIdx.x - lane;
    double *o = out + ld * (lane >> 5) + p0 + 2u * (lane & 31u);
    d2 v = {1.0 + lane, 2.0};
    for (int t = 0; t + 2 <= rows; t += 2) {
        if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o), "v"(v) : "memory");
        else if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(o), "v"(v) : "memory");
        else if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(o), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(o), "v"(v) : "memory");
        o += 2 * ld; v.x += 1.0;
    }
}

// the supplied-brownians pattern without arithmetic: walk-read one array, walk-write the other (8 B per lane per row each way)
template <int BLK> __global__ __launch_bounds__(BLK) void walkcopy(double *out, const double *in, size_t ld, int rows)
{
    const size_t p = (size_t)blockIdx.x * BLK + threadIdx.x;
    double *o = out + p;
    const double *w = in + p;
    double a[4], b[4];
    for (int u = 0; u < 4; ++u) a[u] = w[(size_t)u * ld];
    int t = 0;
    for (; t + 8 <= rows; t += 4) {
        for (int u = 0; u < 4; ++u) b[u] = w[(size_t)(t + 4 + u) * ld];
        for (int u = 0; u < 4; ++u) { *o = a[u] + 1.0; o += ld; }
        for (int u = 0; u < 4; ++u) a[u] = b[u];
    }
    for (int u = 0; u < 4; ++u) { *o = a[u] + 1.0; o += ld; }
}

template <bool NT> __global__ __launch_bounds__(256) void linear16(d2 *out, size_t n16)
{
    const d2 v = {1.0, 2.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) st<NT>(out + i, v);
}

template <bool NT, int BLK> __global__ __launch_bounds__(BLK) void walk8(double *out, size_t ld, int rows)
{
    const size_t p = (size_t)blockIdx.x * BLK + threadIdx.x;
    double *o = out + p;
    double v = 1.0 + threadIdx.x;
    for (int t = 0; t < rows; ++t) { st<NT>(o, v); o += ld; v += 1.0; }
}

template <bool NT, int BLK> __global__ __launch_bounds__(BLK) void walk8x4(double *out, size_t ld, int rows)
{
    const size_t p = (size_t)blockIdx.x * BLK + threadIdx.x;
    double *o = out + p;
    double v = 1.0 + threadIdx.x;
    for (int t = 0; t + 4 <= rows; t += 4) {
        st<NT>(o, v); st<NT>(o + ld, v + 1.0); st<NT>(o + 2 * ld, v + 2.0); st<NT>(o + 3 * ld, v + 3.0);
        o += 4 * ld; v += 4.0;
    }
}

template <bool NT, int BLK> __global__ __launch_bounds__(BLK) void walk16(double *out, size_t ld, int rows)
{
    const unsigned lane = threadIdx.x & 63u;
    const size_t p0 = (size_t)blockIdx.x * BLK + threadIdx.x - lane;
    double *o = out + ld * (lane >> 5) + p0 + 2u * (lane & 31u);
    d2 v = {1.0 + lane, 2.0};
    for (int t = 0; t + 2 <= rows; t += 2) { st<NT>((d2 *)o, v); o += 2 * ld; v.x += 1.0; }
}

template <bool NT> __global__ __launch_bounds__(1024) void walkblk(double *out, size_t ld, int rows)
{
    const unsigned lane = threadIdx.x & 63u;
    const size_t p0 = (size_t)blockIdx.x * 1024 + threadIdx.x - lane;
    double *o = out + ld * (lane >> 5) + p0 + 2u * (lane & 31u);
    d2 v = {1.0 + lane, 2.0};
    for (int t = 0; t + 2 <= rows; t += 2) { __syncthreads(); st<NT>((d2 *)o, v); o += 2 * ld; v.x += 1.0; }
}

// row-major sweep: the whole grid writes row t before row t + 1 as far as the hardware keeps the blocks in step -- here forced:
// block b writes columns [b 1024, (b+1) 1024) of rows [t0, t0 + RB) then moves on; a persistent grid of `resident` blocks
// covers the columns in strips: strip s = all columns of the resident set ... (what an ideal lockstep would give)
template <bool NT> __global__ __launch_bounds__(1024) void sweep(double *out, size_t ld, int rows, size_t n)
{
    // grid = resident blocks; each block loops over column strips of width gridDim.x * 1024
    const unsigned lane = threadIdx.x & 63u;
    d2 v = {1.0 + lane, 2.0};
    for (size_t c0 = (size_t)blockIdx.x * 1024; c0 < n; c0 += (size_t)gridDim.x * 1024) {
        const size_t p0 = c0 + threadIdx.x - lane;
        double *o = out + ld * (lane >> 5) + p0 + 2u * (lane & 31u);
        for (int t = 0; t + 2 <= rows; t += 2) { st<NT>((d2 *)o, v); o += 2 * ld; v.x += 1.0; }
    }
}

int main()
{
    const size_t n = 1 << 20;
    const int nb = 1024, rows = nb + 1;
    const size_t bytes = (size_t)rows * n * 8;
    double *out;
    CHECK(hipMalloc(&out, bytes + 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double wbytes, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        CHECK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0.f;
        for (int i = 0; i < 5; ++i) {
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        CHECK(hipGetLastError());
        printf("{\"kernel\": \"%s\", \"mean_ms\": %.4f, \"min_ms\": %.4f, \"TBps_mean\": %.3f, \"TBps_best\": %.3f}\n", name, sum / 5, best,
               wbytes / (sum / 5) / 1e9, wbytes / best / 1e9);
    };
    timeit("memset", bytes, [&] { CHECK(hipMemsetAsync(out, 0, bytes, 0)); });
    timeit("linear16", bytes, [&] { linear16<false><<<256 * 32, 256>>>((d2 *)out, bytes / 16); });
    timeit("linear16_nt", bytes, [&] { linear16<true><<<256 * 32, 256>>>((d2 *)out, bytes / 16); });
    const double wb = (double)nb * n * 8;
    timeit("walk8_b256", wb, [&] { walk8<false, 256><<<n / 256, 256>>>(out, n, nb); });
    timeit("walk8_b256_nt", wb, [&] { walk8<true, 256><<<n / 256, 256>>>(out, n, nb); });
    timeit("walk8_b1024", wb, [&] { walk8<false, 1024><<<n / 1024, 1024>>>(out, n, nb); });
    timeit("walk8x4_b256", wb, [&] { walk8x4<false, 256><<<n / 256, 256>>>(out, n, nb); });
    timeit("walk16_b256", wb, [&] { walk16<false, 256><<<n / 256, 256>>>(out, n, nb); });
    timeit("walk16_b256_nt", wb, [&] { walk16<true, 256><<<n / 256, 256>>>(out, n, nb); });
    timeit("walk16_b1024", wb, [&] { walk16<false, 1024><<<n / 1024, 1024>>>(out, n, nb); });
    timeit("walk16_sc1", wb, [&] { walk16pol<2, 256><<<n / 256, 256>>>(out, n, nb); });
    timeit("walk16_sc0sc1", wb, [&] { walk16pol<3, 256><<<n / 256, 256>>>(out, n, nb); });
    timeit("walk16_sc1nt", wb, [&] { walk16pol<4, 256><<<n / 256, 256>>>(out, n, nb); });
    timeit("walk16_sc0", wb, [&] { walk16pol<1, 256><<<n / 256, 256>>>(out, n, nb); });
    {
        double *in;
        CHECK(hipMalloc(&in, bytes));
        CHECK(hipMemsetAsync(in, 0, bytes, 0));
        timeit("walkcopy_b256 (read + write bytes)", 2 * wb, [&] { walkcopy<256><<<n / 256, 256>>>(out, in, n, nb); });
        timeit("memcpy_d2d (read + write bytes)", 2.0 * bytes, [&] { CHECK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); });
        CHECK(hipFree(in));
    }
    timeit("walkblk", wb, [&] { walkblk<false><<<n / 1024, 1024>>>(out, n, nb); });
    timeit("walkblk_nt", wb, [&] { walkblk<true><<<n / 1024, 1024>>>(out, n, nb); });
    timeit("sweep_512blocks", wb, [&] { sweep<false><<<512, 1024>>>(out, n, nb, n); });
    timeit("sweep_256blocks", wb, [&] { sweep<false><<<256, 1024>>>(out, n, nb, n); });
    timeit("sweep_1024blocks", wb, [&] { sweep<false><<<1024, 1024>>>(out, n, nb, n); });
    return 0;
}
