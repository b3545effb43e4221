// stream_battery.hip -- a long-sequence battery for random stream version 3 ON THE DEVICE, over the product's own counter
// layout (csrc/svmc_rng.h: key = seed, ctr = (path_lo, path_hi, call index, stream | call_id << 8), Philox4x32-R).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Istochvolmodels_amd/csrc -Iinclude [-DSVMC_PHILOX_ROUNDS=10] \
//         tools/r04/stream_battery.hip -o tools/r04/stream_battery_r7
//   tools/r04/stream_battery_r7 <log2 paths> <log2 calls per path> <seed> <out.bin>
//
// Each lane owns a path (its global id = blockIdx * 256 + thread) and walks the calls 0 .. 2^C - 1 of stream 0, call id 0 --
// exactly the words a generator's time loop consumes (one call = the four words of two time steps).  Histograms (64-bit
// counts, accumulated with 32-bit global atomics into REPLICAS picked by block so that one L2 line is not hammered by the
// whole chip), written raw to <out.bin>; tools/r04/stream_battery.py turns them into statistics and p-values:
//   [0..8)   chi-square of the 16-bit halves: word position w in 0..3 x {high, low} half, 2^16 bins each
//   [8]      pair (r3 of call c, r0 of call c + 1) of one path -- ACROSS THE CALL BOUNDARY -- top 8 bits x top 8 bits
//   [9]      the same pair on the LOW 8 bits of both words
//   [10]     pair (r1, r2) inside a call (the two time steps a call serves), top 8 x top 8 bits
//   [11]     pair (path p, path p + 1) at the same call, word r0, top 8 x top 8 bits (adjacent lanes of a wave)
//   [12]     the same on word r3, low 8 x low 8 bits
//   [13]     pair (word r0 of call c, word r0 of call c + 1): lag one CALL, top 8 x top 8 bits
//   gaps     the gap test on the top 4 bits of every word in sequence order (r0 r1 r2 r3 of call 0, r0 .. of call 1, ...):
//            gap lengths between visits of [0, 1/16), bins 0 .. 126 and >= 127
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "svmc_rng.h"

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

constexpr int N_HIST = 14, BINS = 1 << 16, REPLICAS = 64, GAP_BINS = 128;

__global__ __launch_bounds__(256) void battery_kernel(uint32_t *hist /* [REPLICAS][N_HIST][BINS] */, unsigned long long *gaps /* [GAP_BINS] */,
                                                      int log2_calls, uint64_t seed)
{
    __shared__ unsigned int s_gaps[GAP_BINS];
    for (int i = threadIdx.x; i < GAP_BINS; i += 256) s_gaps[i] = 0u;
    __syncthreads();
    uint32_t *h = hist + static_cast<size_t>(blockIdx.x % REPLICAS) * N_HIST * BINS;
    const uint64_t path = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t n_calls = 1u << log2_calls;
    uint32_t prev_r3 = 0u, prev_r0 = 0u;
    unsigned gap = 0u;
    bool seen = false;                                    // the gap before the first visit is not a gap between visits
    for (uint32_t c = 0; c < n_calls; ++c) {
        uint32_t r[4];
        svmc::philox_draw(seed, 0u, path, c, r);          // stream 0, call id 0: the generators' layout
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            atomicAdd(h + (2 * w) * BINS + (r[w] >> 16), 1u);
            atomicAdd(h + (2 * w + 1) * BINS + (r[w] & 0xFFFFu), 1u);
            if ((r[w] >> 28) == 0u) {
                if (seen) atomicAdd(&s_gaps[gap < GAP_BINS - 1 ? gap : GAP_BINS - 1], 1u);
                seen = true;
                gap = 0u;
            } else {
                ++gap;
            }
        }
        if (c > 0u) {
            atomicAdd(h + 8 * BINS + (((prev_r3 >> 24) << 8) | (r[0] >> 24)), 1u);
            atomicAdd(h + 9 * BINS + (((prev_r3 & 0xFFu) << 8) | (r[0] & 0xFFu)), 1u);
            atomicAdd(h + 13 * BINS + (((prev_r0 >> 24) << 8) | (r[0] >> 24)), 1u);
        }
        atomicAdd(h + 10 * BINS + (((r[1] >> 24) << 8) | (r[2] >> 24)), 1u);
        const uint32_t n0 = __shfl_down(r[0], 1, 64), n3 = __shfl_down(r[3], 1, 64);      // path p + 1 (lanes 0..62)
        if (lane < 63u) {
            atomicAdd(h + 11 * BINS + (((r[0] >> 24) << 8) | (n0 >> 24)), 1u);
            atomicAdd(h + 12 * BINS + (((r[3] & 0xFFu) << 8) | (n3 & 0xFFu)), 1u);
        }
        prev_r3 = r[3];
        prev_r0 = r[0];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < GAP_BINS; i += 256) atomicAdd(gaps + i, static_cast<unsigned long long>(s_gaps[i]));
}

int main(int argc, char **argv)
{
    if (argc < 5) { printf("usage: %s <log2 paths> <log2 calls> <seed> <out.bin>\n", argv[0]); return 2; }
    const int lp = atoi(argv[1]), lc = atoi(argv[2]);
    const uint64_t seed = strtoull(argv[3], nullptr, 0);
    const size_t n_hist = static_cast<size_t>(REPLICAS) * N_HIST * BINS;
    uint32_t *hist;
    unsigned long long *gaps;
    CHECK(hipMalloc(&hist, n_hist * sizeof(uint32_t)));
    CHECK(hipMalloc(&gaps, GAP_BINS * sizeof(unsigned long long)));
    CHECK(hipMemset(hist, 0, n_hist * sizeof(uint32_t)));
    CHECK(hipMemset(gaps, 0, GAP_BINS * sizeof(unsigned long long)));
    // a replica's bin may not overflow 32 bits: expected count per bin and replica = paths * calls / (REPLICAS * 2^16)
    if (lp + lc - 6 - 16 >= 31) { printf("too many words for 32-bit replica counters\n"); return 2; }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    battery_kernel<<<dim3(1u << (lp - 8)), dim3(256)>>>(hist, gaps, lc, seed);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint32_t> hh(n_hist);
    std::vector<unsigned long long> hg(GAP_BINS);
    CHECK(hipMemcpy(hh.data(), hist, n_hist * sizeof(uint32_t), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hg.data(), gaps, GAP_BINS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> sum(static_cast<size_t>(N_HIST) * BINS, 0ull);
    for (int rpl = 0; rpl < REPLICAS; ++rpl)
        for (size_t i = 0; i < static_cast<size_t>(N_HIST) * BINS; ++i) sum[i] += hh[static_cast<size_t>(rpl) * N_HIST * BINS + i];
    FILE *f = fopen(argv[4], "wb");
    if (!f) { printf("cannot open %s\n", argv[4]); return 1; }
    const unsigned long long header[6] = {static_cast<unsigned long long>(svmc::PHILOX_ROUNDS), static_cast<unsigned long long>(lp),
                                          static_cast<unsigned long long>(lc), seed, N_HIST, GAP_BINS};
    fwrite(header, sizeof(unsigned long long), 6, f);
    fwrite(sum.data(), sizeof(unsigned long long), sum.size(), f);
    fwrite(hg.data(), sizeof(unsigned long long), hg.size(), f);
    fclose(f);
    printf("{\"philox_rounds\": %d, \"log2_paths\": %d, \"log2_calls\": %d, \"words\": %.0f, \"kernel_ms\": %.1f}\n", svmc::PHILOX_ROUNDS, lp, lc,
           4.0 * static_cast<double>(1ull << (lp + lc)), ms);
    return 0;
}
