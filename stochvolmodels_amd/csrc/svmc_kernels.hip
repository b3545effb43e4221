// svmc_kernels.hip -- gfx950 kernels and compute entry points of the C ABI (include/svmc.h).
//
// Mapping: one wavefront lane per Monte Carlo path.  A path's state (x, L = ln sigma, sigma, I) lives in
// VGPRs for the whole slice; HBM is touched once per slice per path (24 B read + 24 B write) in the
// on-device-RNG kernels, and 16 B per path-step (the two supplied normals, coalesced 512 B per wave
// per load) in the streamed-randoms kernels.  No MFMA: the work is elementwise fp64 VALU +
// transcendentals (DESIGN.md "Rooflines").
#include "svmc_internal.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include "svmc_black.h"
#include "svmc_models.h"
#include "svmc_rng.h"

namespace svmc {

constexpr int BLOCK = 256;             // 4 waves of 64 lanes
#ifndef SVMC_MAX_REDUCE_GRID
#define SVMC_MAX_REDUCE_GRID 1024
#endif
constexpr int MAX_REDUCE_GRID = SVMC_MAX_REDUCE_GRID;  // 256 CUs x 4 blocks: cap for grid-stride reductions

// block size of the on-device-RNG generators: a block stages the draw's tables in LDS once for all of its waves
#ifndef SVMC_RNG_BLOCK
#define SVMC_RNG_BLOCK 512
#endif
constexpr int RNG_BLOCK = SVMC_RNG_BLOCK;
static constexpr int rng_block() { return RNG_BLOCK; }
#ifndef SVMC_RNG_SGPRS
#define SVMC_RNG_SGPRS 72              // SGPR budget of the LogSV stepping kernels (tools/ubench/ab_kernels.py sweeps it)
#endif
static inline unsigned rng_grid(size_t n) { return static_cast<unsigned>((n + rng_block() - 1) / rng_block()); }

static inline unsigned grid_for(size_t n) { return static_cast<unsigned>((n + BLOCK - 1) / BLOCK); }

// In-kernel clock probe of the on-device-RNG LogSV generators (svmc_clock_probe_arm / _read; bench.py
// roofline.clock_mhz_in_kernel): measurement plumbing, OFF unless the calling host thread armed it.  The stepping kernels
// take a nullable `probe` argument -- null in every product launch: one scalar test, nothing else.  When a thread has armed
// the probe, ITS launches pass that thread's own 8-word device buffer and thread 0 of the launch's FIRST and LAST block stamp
// s_memtime (tick = shader cycle) and s_memrealtime (100 MHz, chip-wide) at kernel entry and again after the time loop:
// (t_exit - t_entry) / (r_exit - r_entry) x 100 MHz is the shader clock that wave saw while it stepped -- measured in the
// timed launch itself, where the SMU's sysfs sensor (absent on some boxes of the pool) lags and averages.  No global: two
// engines on two streams (tests/test_gpu_parity.py::test_two_engines_on_their_own_streams) never share stamps, and a
// thread that did not arm the probe never writes any (round 4 kept one __device__ array every launch of three kernels
// wrote: a race between engines and bench-only code on the hot path).
// Layout: [which = 0 first block | 1 last block][t_entry, r_entry, t_exit, r_exit]; the latest armed launch wins.
__device__ __forceinline__ void clock_probe_stamp(uint64_t *probe, int at_exit)
{
    if (probe == nullptr) return;                                                    // wave-uniform (a kernel argument)
    const bool first = blockIdx.x == 0u, last = blockIdx.x == gridDim.x - 1u;        // wave-uniform
    if (first || last) {
        const uint64_t t = __builtin_readcyclecounter(), r = wall_clock64();
        if (threadIdx.x == 0u) {
            uint64_t *o = probe + (first ? 0 : 4) + 2 * at_exit;
            o[0] = t;
            o[1] = r;
        }
    }
}

// the calling host thread's armed probe buffer (device memory of the device that was current at svmc_clock_probe_arm), or null
static uint64_t *&armed_probe()
{
    thread_local uint64_t *p = nullptr;
    return p;
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void fill_state_kernel(double *__restrict__ x, double *__restrict__ vol,
                                                           double *__restrict__ qvar, size_t n, double x0,
                                                           double vol0, double qvar0)
{
    const size_t p = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
    if (p < n) {
        x[p] = x0;
        vol[p] = vol0;
        qvar[p] = qvar0;
    }
}

__global__ __launch_bounds__(RNG_BLOCK) void fill_normals_kernel(double *__restrict__ W0, double *__restrict__ W1,
                                                             size_t ldw, size_t n, int nb_steps, uint64_t seed,
                                                             uint32_t c3, uint64_t path_offset,
                                                             uint32_t step_offset)
{
    __shared__ RngTablesLds s_tab;
    const RngTables tab = stage_rng_tables(s_tab);
    const size_t p = static_cast<size_t>(blockIdx.x) * RNG_BLOCK + threadIdx.x;
    if (p >= n) return;
    const PhiloxLane lane = philox_prepare(seed, c3, path_offset + p);
    size_t row = p;
    rng_time_loop(lane, step_offset, nb_steps, tab, [&](double w0, double w1) {
        W0[row] = w0;
        W1[row] = w1;
        row += ldw;
    });
}

__global__ __launch_bounds__(BLOCK) void fill_uniforms_kernel(double *__restrict__ U, size_t ldw, size_t n,
                                                              int nb_steps, uint64_t seed, uint32_t c3,
                                                              uint64_t path_offset, uint32_t step_offset)
{
    const size_t p = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
    if (p >= n) return;
    const uint64_t gp = path_offset + p;
    for (int t = 0; t < nb_steps; ++t)
        U[static_cast<size_t>(t) * ldw + p] = draw_uniform(seed, c3, gp, step_offset + static_cast<uint32_t>(t));
}

// ---------------------------------------------------------------------------------------------------
// Deterministic block reductions: wave shuffles in a fixed tree, one LDS exchange, ONE barrier for any number of
// values (the first version paid two barriers per value: 96 per payoff block).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// One halving level of the multi-value butterfly: lanes with (lane & MASK) set keep the upper half of v[0..2H),
// the others the lower half, and add the partner lane's copy of the half they keep.
template <int H, int MASK>
__device__ __forceinline__ void butterfly_halve(const double *v, double *w, int lane)
{
    const bool hi = (lane & MASK) != 0;
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const double mine = hi ? v[H + j] : v[j];
        const double other = hi ? v[j] : v[H + j];
        w[j] = mine + __shfl_xor(other, MASK, 64);
    }
}

// values a block sum works on: NV rounded up to a multiple of 8 from 8 on (the halving tree below wants it; zeros fill up)
constexpr int block_sum_padded(int nv) { return nv >= 8 ? (nv + 7) / 8 * 8 : nv; }

// every thread of the block calls; thread j < n_out ends up with the block total of v[j] and hands it to store(j, total)
template <int NV, class Store>
__device__ __forceinline__ void block_sum_apply(double (&v)[NV], double *lds /* [4 * block_sum_padded(NV)] */, int n_out,
                                                Store &&store)
{
    constexpr int NP = block_sum_padded(NV);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    if constexpr (NP % 8 == 0) {
        // NP values over 64 lanes in NP/2 + NP/4 + NP/8 + 3*NP/8 shuffles instead of 6*NP: three halving levels
        // (xor 32, 16, 8) leave every lane with NP/8 partial sums of the index block its bits 5..3 select, three
        // plain levels (xor 4, 2, 1) finish them.  Fixed tree, hence deterministic.
        double vp[NP], a[NP / 2], b[NP / 4], c[NP / 8];
#pragma unroll
        for (int j = 0; j < NP; ++j) vp[j] = (j < NV) ? v[j] : 0.0;
        butterfly_halve<NP / 2, 32>(vp, a, lane);
        butterfly_halve<NP / 4, 16>(a, b, lane);
        butterfly_halve<NP / 8, 8>(b, c, lane);
#pragma unroll
        for (int j = 0; j < NP / 8; ++j) {
            c[j] += __shfl_xor(c[j], 4, 64);
            c[j] += __shfl_xor(c[j], 2, 64);
            c[j] += __shfl_xor(c[j], 1, 64);
        }
        if ((lane & 7) == 0) {
            const int off = ((lane & 32) ? NP / 2 : 0) + ((lane & 16) ? NP / 4 : 0) + ((lane & 8) ? NP / 8 : 0);
#pragma unroll
            for (int j = 0; j < NP / 8; ++j) lds[wave * NP + off + j] = c[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = wave_sum(v[j]);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < NV; ++j) lds[wave * NP + j] = v[j];
        }
    }
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < n_out) {
        const int j = threadIdx.x;
        double t = lds[j];
        for (int w = 1; w < n_waves; ++w) t += lds[w * NP + j];      // fixed order: wave 0, 1, 2, 3
        store(j, t);
    }
}

// thread j < n_out writes the block total of v[j] to out[j]
template <int NV>
__device__ __forceinline__ void block_sum_store(double (&v)[NV], double *lds /* [4 * block_sum_padded(NV)] */, double *out, int n_out)
{
    block_sum_apply<NV>(v, lds, n_out, [&](int j, double t) { out[j] = t; });
}

// The sum of one column, the ONE order of additions every column reduction of the library uses (so that a sum does not depend on
// which kernel formed it).  256 (virtual) threads: thread t adds rows t, t + 256, ... into four accumulators -- sixteen rows at a
// time into a[k mod 4] while sixteen remain, then four at a time into a[0..3], then the last three or fewer into a[0] --, forms
// (a0 + a1) + (a2 + a3), each wave adds its 64 lanes in a fixed shuffle tree and the four wave totals are added in wave order.
// A row is one 8-byte read at a stride of ld -- latency, not bandwidth: what a reduction costs is how many DEPENDENT round trips
// it makes, so the loads are batched (sixteen per trip in the long loop; a column of up to NR x 256 rows has ALL its loads in
// flight before the first addition -- round 6: the spot columns of a 10^5-path launch were 1 + 3 dependent round trips, 4.7 us of
// a 5 us kernel) and the additions keep the order above: the same bits whichever route.
template <int NR>
__device__ __forceinline__ void column_rows_load(const double *__restrict__ partials, unsigned r, unsigned n_rows, size_t ld, double (&t)[NR])
{
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const unsigned rk = r + static_cast<unsigned>(k) * BLOCK;
        t[k] = (rk < n_rows) ? partials[static_cast<size_t>(rk) * ld] : 0.0;
    }
}
// ... for n_rows <= NR x 256 (NR = 4 or 8: the sixteen-row loop never runs): the additions of the loops, on the preloaded rows
template <int NR>
__device__ __forceinline__ double column_rows_add(const double (&t)[NR], unsigned r, unsigned n_rows)
{
    static_assert(NR == 4 || NR == 8, "one or two trips of the four-row loop");
    const unsigned cnt = (n_rows > r) ? (n_rows - r + BLOCK - 1) / BLOCK : 0u;        // rows of this thread
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    unsigned k = 0;
    if (cnt >= 4u) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += t[u];
        k = 4;
        if constexpr (NR == 8) {
            if (cnt >= 8u) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] += t[4 + u];
                k = 8;
            }
        }
    }
#pragma unroll
    for (int kk = 0; kk < NR; ++kk)
        if (static_cast<unsigned>(kk) >= k && static_cast<unsigned>(kk) < cnt) a[0] += t[kk];
    return (a[0] + a[1]) + (a[2] + a[3]);
}
// any n_rows: the loops themselves
__device__ __forceinline__ double column_rows_sum(const double *__restrict__ partials, unsigned r, unsigned n_rows, size_t ld)
{
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (; r + 15 * BLOCK < n_rows; r += 16 * BLOCK) {
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = partials[static_cast<size_t>(r + u * BLOCK) * ld];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] += t[4 * k + u];
        }
    }
    for (; r + 3 * BLOCK < n_rows; r += 4 * BLOCK) {
        double t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = partials[static_cast<size_t>(r + u * BLOCK) * ld];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += t[u];
    }
    for (; r < n_rows; r += BLOCK) a[0] += partials[static_cast<size_t>(r) * ld];
    return (a[0] + a[1]) + (a[2] + a[3]);
}

// by a 256-thread block: every thread calls; thread 0 returns the total (the others a partial)
__device__ __forceinline__ double block_column_sum(const double *__restrict__ partials, unsigned n_rows, size_t ld, double *lds /* [4] */)
{
    double v[1];
    if (n_rows <= 8u * BLOCK) {                             // (block-uniform)
        double t[8];
        column_rows_load<8>(partials, threadIdx.x, n_rows, ld, t);
        v[0] = column_rows_add<8>(t, threadIdx.x, n_rows);
    } else {
        v[0] = column_rows_sum(partials, threadIdx.x, n_rows, ld);
    }
    double total = 0.0;
    block_sum_apply<1>(v, lds, 1, [&](int, double t) { total = t; });
    return total;
}

// two columns by one block with the loads of BOTH in flight together (the payoff kernel's spot sums: [sum F exp(x), count])
__device__ __forceinline__ void block_column_sum2(const double *__restrict__ p0, const double *__restrict__ p1, unsigned n_rows, double *lds,
                                                  double &s0, double &s1)
{
    double v0[1], v1[1];
    if (n_rows <= 8u * BLOCK) {
        double t0[8], t1[8];
        column_rows_load<8>(p0, threadIdx.x, n_rows, 1, t0);
        column_rows_load<8>(p1, threadIdx.x, n_rows, 1, t1);
        v0[0] = column_rows_add<8>(t0, threadIdx.x, n_rows);
        v1[0] = column_rows_add<8>(t1, threadIdx.x, n_rows);
    } else {
        v0[0] = column_rows_sum(p0, threadIdx.x, n_rows, 1);
        v1[0] = column_rows_sum(p1, threadIdx.x, n_rows, 1);
    }
    s0 = s1 = 0.0;
    block_sum_apply<1>(v0, lds, 1, [&](int, double t) { s0 = t; });
    __syncthreads();                                        // thread 0 has read the wave totals: lds may be written again
    block_sum_apply<1>(v1, lds, 1, [&](int, double t) { s1 = t; });
}

// ---------------------------------------------------------------------------------------------------
// LogSV generators (pricers/logsv_pricer.py:950-1047)
// ---------------------------------------------------------------------------------------------------
// Least-progress-first wave priority.  The SIMD arbiter picks by priority, then age; with equal priorities the
// oldest wave always wins, younger waves starve and each launch ends in a long ramp-down where lone, late waves run
// at ~60 % of the saturated rate (tools/ubench/tail_probe.hip).  Dropping a wave's own priority as it progresses
// through the time loop (3 -> 0 at the quarter points) lets waves that are behind catch up: residency stays at 8
// waves per SIMD and the ramp-down shrinks from ~30 % to ~10 % of the launch.
__device__ __forceinline__ void progress_priority(int stage)
{
    switch (stage) {
    case 0: __builtin_amdgcn_s_setprio(3); break;
    case 1: __builtin_amdgcn_s_setprio(2); break;
    case 2: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(0); break;
    }
}

// Optional slice epilogue fused into the stepping kernels: the terminal x (and qvar) is also written to the
// per-expiry snapshot the payoff pass reads, and [sum F*exp(x), count] (utils/mc_payoffs.py:61-62) goes out as one row PER
// WAVE -- partials[column][wave], wave = global thread index / 64 -- which reduce_columns_kernel adds up in row order: one
// launch and one pass over x less per expiry.  Rows per wave, not per block: the sum's order of additions is then the same
// whatever block size a kernel runs (the one-slice generators run 512-thread blocks, the whole-chain kernel 1024, the
// streamed ones 256, and their results must agree to the bit), and the epilogue needs no LDS and no barrier.
struct SliceOut {
    double *x_snap;     // nullable
    double *q_snap;     // nullable
    double *partials;   // nullable: this slice's two COLUMNS, [2][rows] -- column-major, so that the reduction reads them coalesced
    double forward;
    size_t rows = 0;    // column stride of `partials`: wave_rows(n) of the launch
};


// Start state of a generator launch: read from x / vol / qvar (uniform = 0), or the same three constants for every path --
// what a chain pricing starts from (x0 = 0, sigma0 | v0, qvar0 = 0: pricers/logsv_pricer.py:823-826, heston_pricer.py:303-305).
// The svmc_*_rng_from entry points use it: no fill launch (11 us + the write-back of its 24 bytes per path at the kernel
// boundary) and no 24-byte read per path ahead of the stepping.
struct StateInit {
    int uniform = 0;
    double x0 = 0.0, vol0 = 0.0, qvar0 = 0.0;
};

__device__ __forceinline__ void slice_epilogue(const SliceOut &so, size_t p, bool active, double xv, double q)
{
    if (active) {
        if (so.x_snap != nullptr) so.x_snap[p] = xv;
        if (so.q_snap != nullptr) so.q_snap[p] = q;
    }
    if (so.partials != nullptr) {
        const double sp = so.forward * exp_full(xv);       // full-range exp: x = +-inf must give inf / 0   :61
        const bool ok = active && (sp == sp);                                                   // nanmean :62
        const double v0 = wave_sum(ok ? sp : 0.0), v1 = wave_sum(ok ? 1.0 : 0.0);
        if ((threadIdx.x & 63u) == 0u && (p >> 6) < so.rows) {        // a launch's last block may hold waves past the last path
            so.partials[p >> 6] = v0;
            so.partials[so.rows + (p >> 6)] = v1;
        }
    }
}

__global__ __launch_bounds__(RNG_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_sgpr(SVMC_RNG_SGPRS))) void logsv_rng_kernel(double *__restrict__ x, double *__restrict__ sigma,
                                                          double *__restrict__ qvar, size_t n, int nb_steps,
                                                          LogsvFast c, uint64_t seed, uint32_t c3,
                                                          uint64_t path_offset, uint32_t step_offset, SliceOut so,
                                                          StateInit init, uint64_t *probe)
{
    __shared__ RngTablesLds s_tab;
    __shared__ double s_exp[256];
    const RngTables tab = stage_tables(s_tab, s_exp);
    clock_probe_stamp(probe, 0);
    const auto exp_of = [&](double v) { return exp2u_tab(v, s_exp); };     // L is carried in units of ln2/256
    const size_t p = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool active = p < n;
#ifdef SVMC_TAIL_PROBE                                     // tools/r03/tail_probe.py: per-wave start / end stamps instead of the
    const uint64_t probe_t0 = wall_clock64();              // qvar snapshot (100 MHz s_memrealtime)
#endif
    double xv = 0.0, s = 1.0, q = 0.0;
    if (active) {
        if (init.uniform) {                                // wave-uniform
            xv = init.x0;
            s = init.vol0;
            q = init.qvar0;
        } else {
            xv = x[p];
            s = sigma[p];
            q = qvar[p];
        }
        double L = log_state(s) * LOG_UNITS_PER_NAT;                                                  // :1039
        double s2 = square_rn(s), acc = 0.0, xacc = 0.0;
        const double s2_start = s2;
        const PhiloxLane lane = philox_prepare(seed, c3, path_offset + p);
        const int quarter = (nb_steps + 3) >> 2;
        int stage = 0, next_stage_t = 0;
        rng_time_loop(
            lane, step_offset, nb_steps, tab,
            [&](double z0, double z1) { logsv_step_acc(c, xacc, L, s, s2, acc, z0, z1, exp_of); },
            [&](int t) {
                if (t >= next_stage_t) {                   // wave-uniform
                    progress_priority(stage++);
                    next_stage_t += quarter;
                }
            });
        logsv_fold_acc(c, xv, q, xacc, acc, s2_start, square_rn(s));
        x[p] = xv;
        sigma[p] = s;
        qvar[p] = q;
    }
    clock_probe_stamp(probe, 1);
#ifdef SVMC_TAIL_PROBE
    if ((threadIdx.x & 63u) == 0u && so.q_snap != nullptr) {
        so.q_snap[2 * (p >> 6)] = static_cast<double>(probe_t0);
        so.q_snap[2 * (p >> 6) + 1] = static_cast<double>(wall_clock64());
    }
    so.q_snap = nullptr;
#endif
    slice_epilogue(so, p, active, xv, q);
}

// logsv_rng_kernel for a launch of a few waves per SIMD (few_waves_launch(): up to seven; the reference's default 10^5 paths are
// 1.5).  Statement for statement the kernel above -- the same bits -- but compiled for latency instead of residency: TB-thread
// blocks (256 in the product: every CU gets work, and four blocks' tables fit a CU's LDS), the register budget of WAVES waves per
// SIMD, and the time loop in form LOOP (svmc_rng.h):
//   GEN_LOOP_PIPE   the product: the step in two halves around its exp-table read, the next pair's cubics, the next Philox call
//                   and the issue of the pair after that between them -- a wave overlaps its own LDS round trips with its own
//                   arithmetic, and the waves of a CU cannot fall into a common LDS phase
//   GEN_LOOP_FEW    round 5's form (all eight reads of a call, then the cubics), GEN_LOOP_AHEAD / GEN_LOOP_PAIR: the alternatives
//                   the round-6 sweep measured against it (LOGSV_LAT_VARIANTS below)
constexpr int FEW_BLOCK = 256;
// the time loop of a LogSV generator in form LOOP: the pipelined form takes the step in its two halves
template <int LOOP>
__device__ __forceinline__ void logsv_gen_time_loop(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab, const LogsvFast &c,
                                                    double &xacc, double &L, double &s, double &acc, const double *exp_table)
{
    if constexpr (LOOP == GEN_LOOP_PIPE) {
        LogsvStepInFlight h;
        const Exp2uTailV k = exp2u_tail_consts();
        rng_time_loop_pipelined(
            lane, step0, nb, tab, [&](double z0, double z1) { logsv_step_acc_front(c, xacc, L, s, z0, z1, exp_table, h); },
            [&]() { logsv_step_acc_mid(h, k); }, [&]() { logsv_step_acc_back(s, acc, h); });
    } else {
        double s2_unused = 0.0;
        gen_time_loop<LOOP>(lane, step0, nb, tab, [&](double z0, double z1) {
            logsv_step_acc(c, xacc, L, s, s2_unused, acc, z0, z1, [&](double v) { return exp2u_tab(v, exp_table); });
        });
    }
}

template <int LOOP, int WAVES, int TB>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void logsv_rng_lat_kernel(
    double *__restrict__ x, double *__restrict__ sigma, double *__restrict__ qvar, size_t n, int nb_steps, LogsvFast c,
    uint64_t seed, uint32_t c3, uint64_t path_offset, uint32_t step_offset, SliceOut so, StateInit init, uint64_t *probe)
{
    __shared__ RngTablesLds s_tab;
    __shared__ double s_exp[256];
    const RngTables tab = stage_tables(s_tab, s_exp);
    clock_probe_stamp(probe, 0);
    const size_t p = static_cast<size_t>(blockIdx.x) * TB + threadIdx.x;
    const bool active = p < n;
    double xv = 0.0, s = 1.0, q = 0.0;
    if (active) {
        if (init.uniform) {
            xv = init.x0;
            s = init.vol0;
            q = init.qvar0;
        } else {
            xv = x[p];
            s = sigma[p];
            q = qvar[p];
        }
        double L = log_state(s) * LOG_UNITS_PER_NAT;                                                  // :1039
        double acc = 0.0, xacc = 0.0;
        const double s2_start = square_rn(s);
        const PhiloxLane lane = philox_prepare(seed, c3, path_offset + p);
        logsv_gen_time_loop<LOOP>(lane, step_offset, nb_steps, tab, c, xacc, L, s, acc, s_exp);
        logsv_fold_acc(c, xv, q, xacc, acc, s2_start, square_rn(s));
        x[p] = xv;
        sigma[p] = s;
        qvar[p] = q;
    }
    clock_probe_stamp(probe, 1);
    slice_epilogue(so, p, active, xv, q);
}

// Which form of a generator a launch runs is decided by how many waves per SIMD it puts on the device, not by a path constant:
//   up to lat_waves_per_simd() (7: 458752 paths on an MI355X, 256 CUs x 4 SIMDs)   the few-waves form: 256-thread blocks, the
//                                                                                  pipelined time loop, a 128-register budget
//   above                                                                          the full-launch kernels (eight waves per SIMD)
// Round 6 (tools/r06/mid_waves_sweep.py -> profiles/r06_mid_waves_sweep.json; wall time of logsv_mc_chain_pricer, 4 x 13 chain x
// 364 steps, pipelined form / round 5's batched form / full-launch kernels): 2^16 paths 0.125 / 0.140 / 0.211 ms, 10^5 0.151 /
// 0.165 / 0.212, 2 x 10^5 0.221 / 0.264 / 0.219, 4 x 10^5 0.331 / 0.411 / 0.344, 2^19 0.371 / 0.459 / 0.362.  From two waves per
// SIMD on EVERY form runs at 250-275 cycles per wave-step -- C2's rate: the loop is bound by VALU issue and the LDS pipe together,
// not by latency -- and a launch takes as long as its fullest SIMD: 200 000 paths are 3.05 waves per SIMD on average but four on
// the fullest, so they cost what 262 144 cost.  What the pipelined form buys is the latency-bound end (one or two waves per SIMD,
// the reference's default 10^5 paths) and the block shape in between (256-thread blocks spread 6.1 waves per SIMD evenly where
// 1024-thread blocks leave a third of the CUs with eight).
// SVMC_FEW_WAVES_MAX_PATHS overrides the limit as a path count (0: the full-launch kernels always); SVMC_GEN_VARIANT = index into
// LOGSV_LAT_VARIANTS / HESTON_LAT_VARIANTS (-1: full-launch) forces one compiled form for every launch (measurement only).
static size_t device_simds()
{
    // per device of the calling thread's current context; the attribute query costs microseconds, so it is cached per device
    static std::atomic<size_t> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1024;
    size_t v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        v = static_cast<size_t>(cus) * 4;
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

constexpr int LAT_WAVES_PER_SIMD = 7;

// true: the launch runs the few-waves form of its generator
static bool few_waves_launch(size_t n_path)
{
    static const long long env_max = [] {
        const char *e = getenv("SVMC_FEW_WAVES_MAX_PATHS");
        return e ? static_cast<long long>(strtoull(e, nullptr, 10)) : -1ll;
    }();
    const size_t limit = env_max >= 0 ? static_cast<size_t>(env_max) : static_cast<size_t>(LAT_WAVES_PER_SIMD) * 64 * device_simds();
    return n_path <= limit;
}

static int forced_gen_variant()
{
    static const int forced = [] {
        const char *e = getenv("SVMC_GEN_VARIANT");
        return e ? atoi(e) : -2;
    }();
    return forced;
}

// All expiries of a chain in ONE stepping launch: the slice loop runs inside the kernel, each slice with its own
// constants (dt, vol backbone) and its own epilogue (snapshot row i, spot partials column pair i).  Eight 128-step
// launches each pay their own ramp-up and ramp-down (C4: 8 x 1.03 ms against 7.44 ms for one 1024-step launch);
// here the chain pays one.  L and sigma^2 are re-derived from sigma at every slice start exactly as a fresh launch
// would, so the results are the bits of the slice-by-slice path.
constexpr int MAX_CHAIN_SLICES = 16;
struct ChainSlices {
    LogsvFast c[MAX_CHAIN_SLICES];
    double forward[MAX_CHAIN_SLICES];
    int nb_steps[MAX_CHAIN_SLICES];
    int m, total_steps;
};

#ifndef SVMC_CHAIN_WAVES
#define SVMC_CHAIN_WAVES 8, 8          // A/B hook: residency of the whole-chain kernel (7, 8 lifts the 64-VGPR cap)
#endif
// The whole-chain kernel runs 1024-thread blocks: two blocks per CU share the CU's LDS, which leaves room -- beside the
// draw's 32 KB table -- to PARK each path's x and qvar (16 KB per block) while the time loop runs.  They are dead inside
// the loop (the loop advances the accumulators, logsv_fold_acc folds them in at the slice's end) but live across it, and
// in registers they pushed the kernel over the 64 VGPRs that eight waves per SIMD allow: the round-2 kernel spilled 68
// bytes per lane to scratch at every slice boundary (2.4 x its algorithmic HBM traffic).  Now: no scratch.
#ifndef SVMC_CHAIN_BLOCK
#define SVMC_CHAIN_BLOCK 1024
#endif
#ifndef SVMC_CHAIN_SGPRS
#define SVMC_CHAIN_SGPRS 80            // the slice loop's scalars on top of the stepping loop's; 80 still admits 8 waves per SIMD
#endif
constexpr int CHAIN_BLOCK = SVMC_CHAIN_BLOCK;
static inline unsigned chain_grid(size_t n) { return static_cast<unsigned>((n + CHAIN_BLOCK - 1) / CHAIN_BLOCK); }

__global__ __launch_bounds__(CHAIN_BLOCK) __attribute__((amdgpu_waves_per_eu(SVMC_CHAIN_WAVES), amdgpu_num_sgpr(SVMC_CHAIN_SGPRS))) void logsv_chain_rng_kernel(
    double *__restrict__ x, double *__restrict__ sigma, double *__restrict__ qvar, size_t n, ChainSlices cs, uint64_t seed,
    uint32_t c3, uint64_t path_offset, uint32_t step_offset, double *__restrict__ x_snap, double *__restrict__ q_snap,
    double *__restrict__ partials, StateInit init, uint64_t *probe)
{
    __shared__ RngTablesLds s_tab;
    __shared__ double s_exp[256];
    __shared__ double s_park[2 * CHAIN_BLOCK];             // [0, B): x, [B, 2B): qvar -- lane t owns elements t and B + t
    const RngTables tab = stage_tables(s_tab, s_exp);
    clock_probe_stamp(probe, 0);
    const auto exp_of = [&](double v) { return exp2u_tab(v, s_exp); };     // L is carried in units of ln2/256
    // the path index is re-derived from threadIdx.x wherever it is needed (opaque to CSE): one VGPR across the time loop
    // instead of the 64-bit index and the addresses formed from it
    const auto path_index = [&]() {
        uint32_t t = threadIdx.x;
        asm volatile("" : "+v"(t));
        return static_cast<size_t>(blockIdx.x) * CHAIN_BLOCK + t;
    };
    const bool active = path_index() < n;
    double s = 1.0;
    {
        double xv = 0.0, q = 0.0;
        if (init.uniform) {                                // wave-uniform
            xv = init.x0;
            s = init.vol0;
            q = init.qvar0;
        } else if (active) {
            const size_t p = path_index();
            xv = x[p];
            s = sigma[p];
            q = qvar[p];
        }
        s_park[threadIdx.x] = xv;
        s_park[CHAIN_BLOCK + threadIdx.x] = q;
    }
    const PhiloxLane lane = philox_prepare(seed, c3, path_offset + path_index());
    const int quarter = (cs.total_steps + 3) >> 2;
    int stage = 0, next_stage_t = 0, tg = 0;
    for (int i = 0; i < cs.m; ++i) {
        const int nb = cs.nb_steps[i];
        double xv = 0.0, q = 0.0;
        {   // every lane steps, the lanes past the last path on a dummy state: inside `if (active)` the progress counters
            // would be divergent values (vector registers, exec-masked branches in the time loop)
            const LogsvFast c = cs.c[i];
            double L = log_state(s) * LOG_UNITS_PER_NAT;                                              // :1039
            double s2 = square_rn(s), acc = 0.0, xacc = 0.0;
            const double s2_start = s2;
            rng_time_loop(
                lane, step_offset + static_cast<uint32_t>(tg), nb, tab,
                [&](double z0, double z1) { logsv_step_acc(c, xacc, L, s, s2, acc, z0, z1, exp_of); },
                [&](int t) {
                    if (tg + t >= next_stage_t) {          // wave-uniform
                        progress_priority(stage++);
                        next_stage_t += quarter;
                    }
                });
            xv = s_park[threadIdx.x];                      // a lane reads back what it alone wrote: no barrier needed
            q = s_park[CHAIN_BLOCK + threadIdx.x];
            logsv_fold_acc(c, xv, q, xacc, acc, s2_start, square_rn(s));
            s_park[threadIdx.x] = xv;
            s_park[CHAIN_BLOCK + threadIdx.x] = q;
        }
        tg += nb;
        const SliceOut so = {x_snap + static_cast<size_t>(i) * n, q_snap ? q_snap + static_cast<size_t>(i) * n : nullptr,
                             partials + 2 * static_cast<size_t>(i) * ((n + 63) >> 6), cs.forward[i], (n + 63) >> 6};
        slice_epilogue(so, path_index(), active, xv, q);
    }
    if (active) {
        const size_t p = path_index();
        x[p] = s_park[threadIdx.x];
        sigma[p] = s;
        qvar[p] = s_park[CHAIN_BLOCK + threadIdx.x];
    }
    clock_probe_stamp(probe, 1);
}

// logsv_chain_rng_kernel for a launch of a few waves per SIMD (see logsv_rng_lat_kernel): the same statements, the same bits
template <int LOOP, int WAVES, int TB>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void logsv_chain_rng_lat_kernel(
    double *__restrict__ x, double *__restrict__ sigma, double *__restrict__ qvar, size_t n, ChainSlices cs, uint64_t seed,
    uint32_t c3, uint64_t path_offset, uint32_t step_offset, double *__restrict__ x_snap, double *__restrict__ q_snap,
    double *__restrict__ partials, StateInit init, uint64_t *probe)
{
    __shared__ RngTablesLds s_tab;
    __shared__ double s_exp[256];
    const RngTables tab = stage_tables(s_tab, s_exp);
    clock_probe_stamp(probe, 0);
    const size_t p = static_cast<size_t>(blockIdx.x) * TB + threadIdx.x;
    const bool active = p < n;
    double xv = 0.0, s = 1.0, q = 0.0;                     // (a few waves per SIMD: x and qvar stay in registers)
    if (init.uniform) {
        xv = init.x0;
        s = init.vol0;
        q = init.qvar0;
    } else if (active) {
        xv = x[p];
        s = sigma[p];
        q = qvar[p];
    }
    const PhiloxLane lane = philox_prepare(seed, c3, path_offset + p);
    int tg = 0;
    for (int i = 0; i < cs.m; ++i) {
        const int nb = cs.nb_steps[i];
        const LogsvFast c = cs.c[i];
        double L = log_state(s) * LOG_UNITS_PER_NAT;                                                  // :1039
        double acc = 0.0, xacc = 0.0;
        const double s2_start = square_rn(s);
        logsv_gen_time_loop<LOOP>(lane, step_offset + static_cast<uint32_t>(tg), nb, tab, c, xacc, L, s, acc, s_exp);
        logsv_fold_acc(c, xv, q, xacc, acc, s2_start, square_rn(s));
        tg += nb;
        const SliceOut so = {x_snap + static_cast<size_t>(i) * n, q_snap ? q_snap + static_cast<size_t>(i) * n : nullptr,
                             partials + 2 * static_cast<size_t>(i) * ((n + 63) >> 6), cs.forward[i], (n + 63) >> 6};
        slice_epilogue(so, p, active, xv, q);
    }
    if (active) {
        x[p] = xv;
        sigma[p] = s;
        qvar[p] = q;
    }
    clock_probe_stamp(probe, 1);
}

// The compiled few-waves forms of the LogSV generators.  Entry 0 is the product form; the rest are the alternatives the round-6
// sweep measured against it (tools/r06/mid_waves_sweep.py selects one for EVERY launch with SVMC_GEN_VARIANT = index).
using LogsvSliceKernel = void (*)(double *, double *, double *, size_t, int, LogsvFast, uint64_t, uint32_t, uint64_t, uint32_t, SliceOut,
                                  StateInit, uint64_t *);
using LogsvChainKernel = void (*)(double *, double *, double *, size_t, ChainSlices, uint64_t, uint32_t, uint64_t, uint32_t, double *,
                                  double *, double *, StateInit, uint64_t *);
struct LogsvLatVariant {
    LogsvSliceKernel slice;
    LogsvChainKernel chain;
    int block;
};
#define SVMC_LOGSV_LAT(LOOP, WAVES, TB) {logsv_rng_lat_kernel<LOOP, WAVES, TB>, logsv_chain_rng_lat_kernel<LOOP, WAVES, TB>, TB}
static const LogsvLatVariant LOGSV_LAT_VARIANTS[] = {
    SVMC_LOGSV_LAT(GEN_LOOP_PIPE, 4, 256),     // 0: the product form
    SVMC_LOGSV_LAT(GEN_LOOP_FEW, 2, 256),      // 1: round 5's few-waves form (all eight reads of a call, then the cubics)
    SVMC_LOGSV_LAT(GEN_LOOP_AHEAD, 4, 256),    // 2: the next call's reads under this call's two steps
    SVMC_LOGSV_LAT(GEN_LOOP_PAIR, 4, 256),     // 3: the next step's reads under this step
};
#undef SVMC_LOGSV_LAT
constexpr int N_LOGSV_LAT_VARIANTS = static_cast<int>(sizeof(LOGSV_LAT_VARIANTS) / sizeof(LOGSV_LAT_VARIANTS[0]));

// -> the entry of LOGSV_LAT_VARIANTS a launch of n_path paths runs, or null: the full-launch kernels
static const LogsvLatVariant *logsv_lat_variant(size_t n_path)
{
    const int forced = forced_gen_variant();
    if (forced == -1) return nullptr;
    if (forced >= 0) return &LOGSV_LAT_VARIANTS[forced < N_LOGSV_LAT_VARIANTS ? forced : 0];
    return few_waves_launch(n_path) ? &LOGSV_LAT_VARIANTS[0] : nullptr;
}

static inline unsigned lat_grid(size_t n, int block = FEW_BLOCK) { return static_cast<unsigned>((n + block - 1) / block); }

// Streamed-randoms time loop: HBM-bound (8 B per supplied random per path-step).  Software-pipelined by hand:
// the NARR*U loads of the next U steps are issued before the current U steps are computed, so every wave keeps
// NARR*U x 512 B in flight (hipcc does not unroll a loop that contains inline asm, and one load per array in
// flight leaves the memory system latency-bound: 5.3 -> 5.7 TB/s for LogSV at U = 4; U = 8, 16 give no more).
constexpr int STREAM_U = 4;
#ifndef SVMC_ROUGH_STREAM_U
#define SVMC_ROUGH_STREAM_U 4          // A/B hook: prefetch depth of the rough kernel's streamed normals
#endif

template <int NARR, int U = STREAM_U, class Step>
__device__ __forceinline__ void streamed_time_loop(const double *const (&w)[NARR], size_t ldw, int nb_steps,
                                                   Step &&step)
{
    double a[NARR][U], b[NARR][U];
    int t = 0;
    if (nb_steps >= U) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < NARR; ++k) a[k][u] = w[k][static_cast<size_t>(u) * ldw];
        for (; t + 2 * U <= nb_steps; t += U) {
#pragma unroll
            for (int u = 0; u < U; ++u)                         // prefetch steps t+U .. t+2U-1
#pragma unroll
                for (int k = 0; k < NARR; ++k) b[k][u] = w[k][static_cast<size_t>(t + U + u) * ldw];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                double v[NARR];
#pragma unroll
                for (int k = 0; k < NARR; ++k) v[k] = a[k][u];
                step(v);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < NARR; ++k) a[k][u] = b[k][u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double v[NARR];
#pragma unroll
            for (int k = 0; k < NARR; ++k) v[k] = a[k][u];
            step(v);
        }
        t += U;
    }
    for (; t < nb_steps; ++t) {
        double v[NARR];
#pragma unroll
        for (int k = 0; k < NARR; ++k) v[k] = w[k][static_cast<size_t>(t) * ldw];
        step(v);
    }
}

__device__ __forceinline__ void logsv_w_body(double *__restrict__ x, double *__restrict__ sigma,
                                             double *__restrict__ qvar, size_t n, int nb_steps, const LogsvConsts &c,
                                             const double *__restrict__ W0, const double *__restrict__ W1, size_t ldw,
                                             const SliceOut &so)
{
    const size_t p = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
    const bool active = p < n;
    double xv = 0.0, q = 0.0;
    if (active) {
        xv = x[p];
        q = qvar[p];
        double s = sigma[p];
        double L = log(s);
        const double *const w[2] = {W0 + p, W1 + p};
        streamed_time_loop<2>(w, ldw, nb_steps, [&](const double(&v)[2]) {
            logsv_step(c, xv, L, s, q, c.sdt * v[0], c.sdt * v[1]);                               // :1028-1030
        });
        x[p] = xv;
        sigma[p] = s;
        qvar[p] = q;
    }
    slice_epilogue(so, p, active, xv, q);
}

__global__ __launch_bounds__(BLOCK) void logsv_w_kernel(double *__restrict__ x, double *__restrict__ sigma,
                                                        double *__restrict__ qvar, size_t n, int nb_steps,
                                                        LogsvConsts c, const double *__restrict__ W0,
                                                        const double *__restrict__ W1, size_t ldw, SliceOut so)
{
    logsv_w_body(x, sigma, qvar, n, nb_steps, c, W0, W1, ldw, so);
}

// The same kernel with its model constants read from device memory: every launch argument is then fixed for a given
// chain and set of randoms, so the launch can sit in a captured hipGraph that is replayed for each new parameter set
// (svmc_chain.hip) -- only the small constants block is rewritten between replays.
__global__ __launch_bounds__(BLOCK) void logsv_w_indirect_kernel(double *__restrict__ x, double *__restrict__ sigma,
                                                                 double *__restrict__ qvar, size_t n, int nb_steps,
                                                                 const LogsvConsts *__restrict__ consts,
                                                                 const double *__restrict__ W0,
                                                                 const double *__restrict__ W1, size_t ldw, SliceOut so)
{
    const LogsvConsts c = *consts;                      // wave-uniform: scalar loads
    logsv_w_body(x, sigma, qvar, n, nb_steps, c, W0, W1, ldw, so);
}

// All expiries of a chain on resident fixed randoms in ONE launch, state initialised in the kernel: what the graph of
// svmc_logsv_chain_price_fixed replays per calibration iterate.  An objective evaluation on a 4 x 13 chain with 10^5 paths
// is a dozen microsecond-scale kernels; one launch per expiry (+ its reduce) and a separate fill were most of them.
// Per slice the arithmetic is logsv_w_body's (L re-derived from sigma at the slice start), so the bits equal the
// slice-by-slice launches.  `init` != nullptr: start every path from (0, *init, 0) instead of reading the state.
static_assert(MAX_FUSED_SLICES == MAX_CHAIN_SLICES, "svmc_internal.h and the chain kernels must agree");
struct ChainWSlices {
    const double *W0[MAX_CHAIN_SLICES], *W1[MAX_CHAIN_SLICES];
    double forward[MAX_CHAIN_SLICES];
    int nb_steps[MAX_CHAIN_SLICES];
    int m;
};

__global__ __launch_bounds__(BLOCK) void logsv_chain_w_indirect_kernel(double *__restrict__ x, double *__restrict__ sigma,
                                                                       double *__restrict__ qvar, size_t n, ChainWSlices cs,
                                                                       const LogsvConsts *__restrict__ consts,
                                                                       const double *__restrict__ init, size_t ldw,
                                                                       double *__restrict__ x_snap, double *__restrict__ q_snap,
                                                                       double *__restrict__ partials)
{
    const size_t p = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
    const bool active = p < n;
    double xv = 0.0, s = 1.0, q = 0.0;
    if (active) {
        if (init != nullptr) {
            s = *init;
        } else {
            xv = x[p];
            s = sigma[p];
            q = qvar[p];
        }
    }
    for (int i = 0; i < cs.m; ++i) {
        if (active) {
            const LogsvConsts c = consts[i];                    // wave-uniform: scalar loads
            double L = log(s);
            const double *const w[2] = {cs.W0[i] + p, cs.W1[i] + p};
            streamed_time_loop<2>(w, ldw, cs.nb_steps[i], [&](const double(&v)[2]) {
                logsv_step(c, xv, L, s, q, c.sdt * v[0], c.sdt * v[1]);                           // :1028-1030
            });
        }
        const SliceOut so = {x_snap + static_cast<size_t>(i) * n, q_snap ? q_snap + static_cast<size_t>(i) * n : nullptr,
                             partials + 2 * static_cast<size_t>(i) * ((n + 63) >> 6), cs.forward[i], (n + 63) >> 6};
        slice_epilogue(so, p, active, xv, q);
    }
    if (active) {
        x[p] = xv;
        sigma[p] = s;
        qvar[p] = q;
    }
}

// The same chain for P PARAMETER SETS in one launch, the resident randoms read ONCE: every lane carries the P states of
// its path and steps them all on each pair of normals it loads.  The single-set launch moves 16 B per path-step for some
// 45 instructions of arithmetic and is HBM-bound on a calibration-sized path set (5.2 TB/s); the base point of an SLSQP
// iterate and its finite-difference neighbours share their randoms, so P sets cost one pass over them.  Per set the
// arithmetic is logsv_chain_w_indirect_kernel's, statement for statement (L re-derived from sigma at every slice start;
// the same step, the same epilogue), hence the same bits as P single-set launches.  The model constants of the
// (slice, set) pairs sit in LDS: P x 13 doubles would not fit the scalar registers, and in vector registers they would
// cost the residency the loads need -- a broadcast ds_read per use runs beside the VALU stream.
// consts: [m][P] LogsvConsts, init: [P] initial volatilities, x_snap / q_snap rows and partial column pairs: set-major,
// (set s, slice i) at s m + i.
constexpr int MAX_CHAIN_SETS = 8;
static_assert(MAX_FUSED_SETS == MAX_CHAIN_SETS, "svmc_internal.h and the chain kernels must agree");
template <int P>
__global__ __launch_bounds__(BLOCK) void logsv_chain_w_sets_kernel(size_t n, ChainWSlices cs,
                                                                   const LogsvConsts *__restrict__ consts,
                                                                   const double *__restrict__ init, size_t ldw,
                                                                   double *__restrict__ x_snap, double *__restrict__ q_snap,
                                                                   double *__restrict__ partials)
{
    __shared__ LogsvConsts s_c[P];
    const size_t p = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
    const bool active = p < n;
    double xv[P], sg[P], q[P], L[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
        xv[s] = 0.0;
        sg[s] = init[s];
        q[s] = 0.0;
    }
    for (int i = 0; i < cs.m; ++i) {
        __syncthreads();                                   // everybody is done with the previous slice's constants
        {
            constexpr int ND = P * static_cast<int>(sizeof(LogsvConsts) / sizeof(double));
            const double *src = reinterpret_cast<const double *>(consts + static_cast<size_t>(i) * P);
            double *dst = reinterpret_cast<double *>(s_c);
            for (int j = threadIdx.x; j < ND; j += BLOCK) dst[j] = src[j];
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int s = 0; s < P; ++s) L[s] = log(sg[s]);
            const double *const w[2] = {cs.W0[i] + p, cs.W1[i] + p};
            streamed_time_loop<2>(w, ldw, cs.nb_steps[i], [&](const double(&v)[2]) {
                // the constants are RE-READ from LDS at every step, not hoisted into 26 P registers: an opaque zero in the
                // index (not an opaque pointer -- that would turn the reads into flat loads, which wait on the global
                // loads in flight as well and undo the prefetch)
                unsigned zero = 0;
                asm volatile("" : "+v"(zero));
#pragma unroll
                for (int s = 0; s < P; ++s) {
                    const LogsvConsts c = s_c[s + zero];
                    logsv_step(c, xv[s], L[s], sg[s], q[s], c.sdt * v[0], c.sdt * v[1]);          // :1028-1030
                }
            });
        }
#pragma unroll
        for (int s = 0; s < P; ++s) {
            const int row = s * cs.m + i;
            const SliceOut so = {x_snap + static_cast<size_t>(row) * n, q_snap ? q_snap + static_cast<size_t>(row) * n : nullptr,
                                 partials + 2 * static_cast<size_t>(row) * ((n + 63) >> 6), cs.forward[i], (n + 63) >> 6};
            slice_epilogue(so, p, active, xv[s], q[s]);
        }
    }
}

// The calibration chain WITHOUT resident randoms: P parameter sets stepped on randoms that are REGENERATED in registers
// from (seed, call_id) on every evaluation.  A counter-based stream is already "frozen by its seed": the fixed randoms of
// an MC calibration (pricers/logsv_pricer.py:244-265, :520-527 draw them once with RandomState and keep the arrays) need
// not exist anywhere -- the round-4 route drew 582 MB into HBM for 10^5 x 364 and streamed them back at every objective
// evaluation (5.3 TB/s, the kernel's floor).  Here a lane draws its path's normals once per step (one Philox call per
// two steps, the inversion per word: 28 of the single-set generator's 49 instructions per step) and advances ALL P states
// on them; the P chains of a lane are independent, so their exp-table round trips overlap and one wave per SIMD -- what a
// calibration-sized path set gives this chip -- keeps its SIMD busy.  No HBM term inside the time loop at all.
// Per set the arithmetic is logsv_chain_rng_kernel's, statement for statement (logsv_step_acc on the constants in log
// units, L re-derived with log_state at every slice start, logsv_fold_acc, slice_epilogue): set q of the launch is
// logsv_mc_chain_pricer(seed, call_id) with set q's parameters, bit for bit.  The constants of the (slice, set) pairs sit
// in LDS and are re-read at every trip (as logsv_chain_w_sets_kernel: 5 P doubles would not fit the scalar registers and
// in vector registers they would be spilled); x and qvar stay in registers (they are dead inside the loop, but a
// calibration-sized launch runs one or two waves per SIMD: there are registers to spare).
// consts: [m][P] LogsvFast in log units, init: [P] initial volatilities; snapshot rows and partial column pairs set-major.
struct ChainRngSetsSlices {
    double forward[MAX_CHAIN_SLICES];
    int nb_steps[MAX_CHAIN_SLICES];
    int m;
};
constexpr int LOGSV_FAST_DOUBLES = static_cast<int>(sizeof(LogsvFast) / sizeof(double));

//
// Latency, not issue, bounds a calibration-sized launch: 10^5 paths are 1563 waves for 1024 SIMDs, one or two per SIMD, and a
// step is a dependent chain of about twenty fp64 operations around an LDS round trip.  Hence (profiles/r05_frozen_*.txt):
//   * the register allocator is told the launch runs two waves per SIMD (amdgpu_waves_per_eu(2, 2): 256 registers) --
//     left at its default it schedules for eight, reuses a handful of registers for every LDS read and waits for each read
//     before issuing the next (185 full lgkmcnt waits per trip at P = 8);
//   * the step runs piece by piece across the P states (logsv_step_acc_sets), all P exp-table reads in flight together;
//   * an interior Philox call turns all four of its words into normals before the first of its two steps (eight table reads
//     in flight), the edge calls of a slice that starts or ends on an odd step take the one half they own -- which
//     (path, step) sees which word is rng_time_loop's rule, unchanged;
//   * the (slice, set) constants are plain loads from the block's LDS copy: loop-invariant, the compiler keeps them in
//     registers across the time loop where the budget allows and re-reads them where it does not.
// The launch shape (block size, one block per CU) was measured NOT to matter: the dispatcher spreads 391 blocks evenly.
#ifndef SVMC_FROZEN_DRAW_DEFAULT
#define SVMC_FROZEN_DRAW_DEFAULT(P) 4      // which form of the draw the P-set kernel compiles (see the comment in its time loop)
#endif
template <int P>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
void logsv_chain_rng_sets_kernel(size_t n, ChainRngSetsSlices cs, const LogsvFast *__restrict__ consts,
                                 const double *__restrict__ init, int p_total, int s0, uint64_t seed, uint32_t c3,
                                 uint64_t path_offset, uint32_t step_offset, double *__restrict__ x_snap,
                                 double *__restrict__ q_snap, double *__restrict__ partials, uint64_t *probe)
{
    // this launch advances the sets s0 .. s0 + P - 1 of the call's p_total (eight sets run as two launches of four: eight
    // states per lane need more than the 256 registers two waves per SIMD leave)
    __shared__ RngTablesLds s_tab;
    __shared__ double s_exp[256];
    __shared__ LogsvFast s_c[P];
    // six and more sets: x and qvar -- dead inside the time loop, live across it -- are parked in LDS (lane t owns its own
    // 2 P words, no barrier), as the whole-chain generator does: with them the eight-set kernel needed 266 registers, one
    // wave per SIMD, and a 10^5-path launch then runs in two rounds
    constexpr bool PARK = P >= 6;
    __shared__ double s_park[PARK ? 2 * P * BLOCK : 1];
    const RngTables tab = stage_tables(s_tab, s_exp);
    clock_probe_stamp(probe, 0);
    const size_t p = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
    const bool active = p < n;
    // (spreading the sets over groups of blocks instead -- more waves of fewer states each -- was measured and is slower at
    // every set count: two sets 0.221 -> 0.289 ms, four 0.302 -> 0.363, six 0.386 -> 0.439; an extra state in a lane costs
    // 40 us, an extra wave of one state 107)
    double xv[P], sg[P], q[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
        xv[s] = 0.0;
        sg[s] = init[s0 + s];
        q[s] = 0.0;
        if constexpr (PARK) {
            s_park[(2 * s) * BLOCK + threadIdx.x] = 0.0;
            s_park[(2 * s + 1) * BLOCK + threadIdx.x] = 0.0;
        }
    }
    // every lane steps (the lanes past the last path on a real path's counter: their results are never stored)
    const PhiloxLane lane = philox_prepare(seed, c3, path_offset + p);
    uint32_t tg = step_offset;
    for (int i = 0; i < cs.m; ++i) {
        __syncthreads();                                   // everybody is done with the previous slice's constants
        {
            constexpr int ND = P * LOGSV_FAST_DOUBLES;
            const double *src = reinterpret_cast<const double *>(consts + static_cast<size_t>(i) * p_total + s0);
            double *dst = reinterpret_cast<double *>(s_c);
            for (int j = threadIdx.x; j < ND; j += BLOCK) dst[j] = src[j];
        }
        __syncthreads();
        const int nb = cs.nb_steps[i];
        double L[P], acc[P], xacc[P], s2_start[P], k1[P], k2[P], k3[P], kb[P], ke[P];
#pragma unroll
        for (int s = 0; s < P; ++s) {
            L[s] = log_state(sg[s]) * LOG_UNITS_PER_NAT;                                              // :1039
            s2_start[s] = square_rn(sg[s]);
            acc[s] = 0.0;
            xacc[s] = 0.0;
            k1[s] = s_c[s].c1;
            k2[s] = s_c[s].c2;
            k3[s] = s_c[s].c3;
            kb[s] = s_c[s].bs;
            ke[s] = s_c[s].es;
        }
        const auto step = [&](double z0, double z1) { logsv_step_acc_sets<P>(k1, k2, k3, kb, ke, xacc, L, sg, acc, z0, z1, s_exp); };
#ifdef SVMC_FROZEN_DRAW
        constexpr int DRAW = SVMC_FROZEN_DRAW;
#else
        constexpr int DRAW = SVMC_FROZEN_DRAW_DEFAULT(P);
#endif
        if constexpr (DRAW == 4) {
            // round 6: the generators' pipelined loop -- the step in two halves around its P exp-table reads, the next pair's
            // cubics, the next Philox call and the issue of the pair after that between them (rng_time_loop_pipelined)
            LogsvSetsInFlight<P> h;
            if constexpr (P <= 6) {
                // the tails in the middle region on coefficients in vector registers (seven sets have no registers left for them)
                const Exp2uTailV tail_k = exp2u_tail_consts();
                rng_time_loop_pipelined(
                    lane, tg, nb, tab,
                    [&](double z0, double z1) { logsv_step_acc_sets_front<P>(k1, k2, k3, kb, ke, xacc, L, sg, z0, z1, s_exp, h); },
                    [&]() { logsv_step_acc_sets_mid<P>(h, tail_k); }, [&]() { logsv_step_acc_sets_back<P, true>(sg, acc, h); });
            } else {
                rng_time_loop_pipelined(
                    lane, tg, nb, tab,
                    [&](double z0, double z1) { logsv_step_acc_sets_front<P>(k1, k2, k3, kb, ke, xacc, L, sg, z0, z1, s_exp, h); },
                    [&]() { logsv_step_acc_sets_back<P>(sg, acc, h); });
            }
        } else if (nb > 0) {
            // rng_time_loop's rule -- call c serves the steps 2c (words 0, 1) and 2c + 1 (words 2, 3) -- as an odd-start half
            // call, the full calls, an even-end half call
            const uint32_t first = tg, last = tg + static_cast<uint32_t>(nb) - 1u;
            uint32_t c = first >> 1, r[4];
            double a0, a1, b0, b1;
            if (first & 1u) {
                philox_draw(lane, c, r);
                normals_from_words(r[2], r[3], tab, b0, b1);
                step(b0, b1);
                ++c;
            }
            // How the draw's table reads are scheduled against the steps decides a launch of one or two waves per SIMD, where no
            // other wave hides an LDS round trip (10^5 paths x 364 steps, wall time of an evaluation, one / seven sets):
            //   0  word by word, as the full-occupancy generators do -- the compiler reads, waits, evaluates four times over:
            //      0.180 / 0.427 ms;
            //   1  all eight reads of a call in flight, then the cubics (draw_issue / draw_finish): 0.172 / 0.419;
            //   3  call c + 1's words, indices and READS issued before call c's two steps, its cubics after them -- the draw's
            //      round trip hides behind the steps (it does not depend on the state); the last trip draws a call nobody
            //      uses rather than branch inside the trip (with the branch, form 2, the trip is 8 us slower than form 0):
            //      0.175 / 0.410.
            //   4  (round 6) the step itself in two halves around its exp-table reads with the draw's pieces between them:
            //      see above.
            // SVMC_FROZEN_DRAW forces one form (tools/ubench A/B builds); round 5 ran form 1 for one set, form 3 for several.
            const uint32_t c_end = (last + 1u) >> 1;
            if constexpr (DRAW == 0) {
                for (; c < c_end; ++c) {
                    philox_draw(lane, c, r);
                    normals_from_words(r[0], r[1], tab, a0, a1);
                    normals_from_words(r[2], r[3], tab, b0, b1);
                    step(a0, a1);
                    step(b0, b1);
                }
            } else if constexpr (DRAW == 1) {
                for (; c < c_end; ++c) {
                    DrawInFlight d;
                    double z[4];
                    philox_draw(lane, c, r);
                    draw_issue(r, tab, d);
                    draw_finish(d, z);
                    step(z[0], z[1]);
                    step(z[2], z[3]);
                }
            } else if (c < c_end) {
                DrawInFlight d;
                philox_draw(lane, c, r);
                draw_issue(r, tab, d);
                for (; c < c_end; ++c) {
                    double z[4];
                    draw_finish(d, z);
                    if constexpr (DRAW == 2) {
                        if (c + 1u < c_end) {
                            philox_draw(lane, c + 1u, r);
                            draw_issue(r, tab, d);
                        }
                    } else {
                        philox_draw(lane, c + 1u, r);
                        draw_issue(r, tab, d);
                    }
                    step(z[0], z[1]);
                    step(z[2], z[3]);
                }
            }
            if (!(last & 1u)) {
                philox_draw(lane, last >> 1, r);
                normals_from_words(r[0], r[1], tab, a0, a1);
                step(a0, a1);
            }
        }
        tg += static_cast<uint32_t>(nb);
#pragma unroll
        for (int s = 0; s < P; ++s) {
            const LogsvFast c = s_c[s];
            if constexpr (PARK) {
                xv[s] = s_park[(2 * s) * BLOCK + threadIdx.x];
                q[s] = s_park[(2 * s + 1) * BLOCK + threadIdx.x];
            }
            logsv_fold_acc(c, xv[s], q[s], xacc[s], acc[s], s2_start[s], square_rn(sg[s]));
            if constexpr (PARK) {
                s_park[(2 * s) * BLOCK + threadIdx.x] = xv[s];
                s_park[(2 * s + 1) * BLOCK + threadIdx.x] = q[s];
            }
            const int row = (s0 + s) * cs.m + i;
            const SliceOut so = {x_snap + static_cast<size_t>(row) * n, q_snap ? q_snap + static_cast<size_t>(row) * n : nullptr,
                                 partials + 2 * static_cast<size_t>(row) * ((n + 63) >> 6), cs.forward[i], (n + 63) >> 6};
            slice_epilogue(so, p, active, xv[s], q[s]);
        }
    }
    clock_probe_stamp(probe, 1);
}

__global__ __launch_bounds__(BLOCK) void fill_state_indirect_kernel(double *__restrict__ x, double *__restrict__ vol,
                                                                    double *__restrict__ qvar, size_t n,
                                                                    const double *__restrict__ vol0)
{
    const size_t p = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
    if (p < n) {
        x[p] = 0.0;
        vol[p] = *vol0;
        qvar[p] = 0.0;
    }
}

// Volatility paths on the full grid (pricers/logsv_pricer.py:930-945): 8 B written per path-step.
//
// Round 4 (profiles/r04_vol_paths.json, tools/r04/write_bw.hip, tools/ubench/ab_vol_paths.py; 2^20 paths x 1024 steps, 8.6 GB):
//   * what the chip does with this store pattern and NO arithmetic -- a wave owns 64 columns and walks the rows, 512 B per
//     wave-store -- is 5.7 TB/s (hipMemset: 6.3; 8- and 16-byte stores, plain / nt / sc1 policies all within 3 %, nt the
//     slowest); a copy in the same pattern (supplied brownians: 8 B read + 8 B written per path-step) runs at 4.84 TB/s of
//     total traffic, as does hipMemcpy -- the SUPPLIED instantiation sits on that roof (3.5-3.6 ms for 17.2 GB).
//   * the DRAWING instantiation is POWER-bound, not store-bound: its arithmetic alone (stores disabled) takes 1.27 ms at a
//     measured 2.0 GHz, its stores alone 1.5 ms -- but with both the shader clock inside the kernel falls to 1.45-1.5 GHz
//     (clock_probe_stamp: s_memtime against s_memrealtime) and the launch takes 1.95 ms = 4.4 TB/s.  Sixteen-byte stores
//     through a wave-private LDS transpose (half the store instructions) measured 4-6 % SLOWER -- the extra LDS traffic costs
//     more power than the store issue it saves -- and were removed again; so were non-temporal and write-through policies.
//   * what did help is a SHORTER step, the generators' form: ln sigma carried in units of ln2/256 so that exp2u_tab's
//     reduction is exact, the drift regrouped into four FMAs on host-scaled constants (logsv_step_acc's), rcp_1n for the
//     1/sigma term (it enters L at 1e-3 of its size): 48 -> 33 VALU instructions per path-step, 2.10 -> 1.95 ms on one box.
//     Identical in exact arithmetic to :942; rounding differs at the 1e-16 level per step (the tests hold both instantiations
//     against the reference's golden paths at 1e-12).
//   * supplied brownians are loaded a group of four steps ahead.
struct VolPathConsts {
    double c1;      // kappa1 theta dt K          (K = 256 / ln2: L is carried in units of ln2 / 256)
    double c2;      // (adj - kappa2) dt K
    double c3;      // (kappa2 theta - kappa1 - vartheta^2 / 2) dt K
    double cz;      // vartheta K for supplied (already scaled) increments, vartheta sqrt(dt) K for the kernel's own N(0,1)
    double L0;      // ln(v0) K
};

#ifndef SVMC_VOLPATHS_RNG_BLOCK
#define SVMC_VOLPATHS_RNG_BLOCK 1024   // drawing instantiation: two blocks per CU share one copy each of the draw's table
#endif
constexpr int VOLPATHS_RNG_BLOCK = SVMC_VOLPATHS_RNG_BLOCK;
#ifndef SVMC_VOLPATHS_VARIANT
#define SVMC_VOLPATHS_VARIANT 0        // measurement builds only (tools/r06/ab_vol_paths.sh): 1 = next call's reads ahead, 2 = four steps, four stores
#endif
#ifndef SVMC_VOLPATHS_PROBE
#define SVMC_VOLPATHS_PROBE 0          // measurement builds only (tools/ubench/ab_vol_paths.py): 1 = no stores (the store stays in
#endif                                 // the code behind a test that never holds), 2 = every store lands on row 1 (L2-resident)

__device__ __forceinline__ void vol_paths_store(double *ptr, double v)
{
#if SVMC_VOLPATHS_PROBE == 1
    if (v == -1.2345e300) *ptr = v;
#else
    *ptr = v;
#endif
}

template <bool RNG>
__global__ __launch_bounds__(RNG ? VOLPATHS_RNG_BLOCK : BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8)))
void logsv_vol_paths_kernel(double *__restrict__ sigma_t, size_t ld, size_t n, int nb_steps, double v0, VolPathConsts c,
                            const double *__restrict__ brownians, size_t ldb, uint64_t seed, uint32_t c3, uint64_t path_offset,
                            uint64_t *probe)
{
    constexpr int TB = RNG ? VOLPATHS_RNG_BLOCK : BLOCK;
    __shared__ RngTablesLdsIf<RNG> s_tab;                  // the draw's table only where the kernel draws
    __shared__ double s_exp[256];
    const RngTables tab = stage_tables_if(s_tab, s_exp);
    clock_probe_stamp(probe, 0);
    const size_t p = static_cast<size_t>(blockIdx.x) * TB + threadIdx.x;
    if (p < n) {
        double s = v0, L = c.L0;
        vol_paths_store(sigma_t + p, s);                                                        // :937
        double *out = sigma_t + ld + p;                    // row t + 1 of this path
        const auto step = [&](double w) {                  // w: N(0,1) (RNG) or the scaled increment sqrt(dt) N(0,1)   :925
            const double y = rcp_1n(s);
            L = fma(c.c2, s, L);                                                                // :942, regrouped
            L = fma(c.c1, y, L);
            L = L + c.c3;
            L = fma(c.cz, w, L);
            s = exp2u_tab(L, s_exp);                                                            // :943
            vol_paths_store(out, s);                                                            // :944
#if SVMC_VOLPATHS_PROBE != 2
            out += ld;
#endif
        };
        int t = 0;
        if (RNG) {
            // one Brownian per step: normal t is the inversion of word t & 3 of call t >> 2 (stream 2) -- a Philox call
            // serves four steps, so the loop runs call by call with no per-step selects
            const PhiloxLane pl = philox_prepare(seed, c3 | 2u, path_offset + p);
            uint32_t r[4];
            double a0, a1, b0, b1;
#if SVMC_VOLPATHS_VARIANT == 1     // measurement: the NEXT call's table reads in flight under this call's four steps (the same bits)
            if (t + 4 <= nb_steps) {
                DrawInFlight d;
                philox_draw(pl, 0u, r);
                draw_issue(r, tab, d);
                for (; t + 4 <= nb_steps; t += 4) {
                    double z[4];
                    draw_finish(d, z);
                    philox_draw(pl, static_cast<uint32_t>((t >> 2) + 1), r);      // the last trip draws a call nobody uses
                    draw_issue(r, tab, d);
                    step(z[0]);
                    step(z[1]);
                    step(z[2]);
                    step(z[3]);
                }
            }
#elif SVMC_VOLPATHS_VARIANT == 2   // measurement: four steps into registers, then their four stores back to back (the same bits)
            for (; t + 4 <= nb_steps; t += 4) {
                philox_draw(pl, static_cast<uint32_t>(t >> 2), r);
                normals_from_words(r[0], r[1], tab, a0, a1);
                normals_from_words(r[2], r[3], tab, b0, b1);
                double sv[4];
                const double zz[4] = {a0, a1, b0, b1};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double y = rcp_1n(s);
                    L = fma(c.c2, s, L);
                    L = fma(c.c1, y, L);
                    L = L + c.c3;
                    L = fma(c.cz, zz[u], L);
                    s = exp2u_tab(L, s_exp);
                    sv[u] = s;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    vol_paths_store(out, sv[u]);
                    out += ld;
                }
            }
#else
            for (; t + 4 <= nb_steps; t += 4) {
                philox_draw(pl, static_cast<uint32_t>(t >> 2), r);
                normals_from_words(r[0], r[1], tab, a0, a1);
                normals_from_words(r[2], r[3], tab, b0, b1);
                step(a0);
                step(a1);
                step(b0);
                step(b1);
            }
#endif
            if (t < nb_steps) {                            // the last, partial call (wave-uniform)
                philox_draw(pl, static_cast<uint32_t>(t >> 2), r);
                normals_from_words(r[0], r[1], tab, a0, a1);
                step(a0);
                if (t + 1 < nb_steps) step(a1);
                if (t + 2 < nb_steps) {
                    normals_from_words(r[2], r[3], tab, b0, b1);
                    step(b0);
                }
            }
        } else {
            const double *w = brownians + p;
            double a[4], b[4];
            if (nb_steps >= 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = w[static_cast<size_t>(u) * ldb];
                for (; t + 8 <= nb_steps; t += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) b[u] = w[static_cast<size_t>(t + 4 + u) * ldb];   // the next group, in flight
#pragma unroll
                    for (int u = 0; u < 4; ++u) step(a[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = b[u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) step(a[u]);
                t += 4;
            }
            for (; t < nb_steps; ++t) step(w[static_cast<size_t>(t) * ldb]);
        }
    }
    clock_probe_stamp(probe, 1);
}

// ---- what the callers of simulate_vol_paths do with the array, done where the array is -------------------------------
// The reference's users of the [nb_steps + 1][nb_path] volatility paths reduce them over the PATH axis at every time step
// (papers/logsv_model_with_quadratic_drift/moments_vol_qvar.py:48 np.mean / np.std of (sigma_t - theta)^k along axis 1, :98
// of sigma_t and of the expanding time average of sigma_t^2): 8.6 GB over PCIe to keep 1025 x 8 numbers.  Two kernels on the
// resident array instead, both HBM-bound single passes:
//   row_power_sums_kernel      per row t: sums over the paths of (a[t][p] - center)^j, j = 1 .. 2 K  (mean and variance of the
//                              first K central-ish moments); one block per (row, segment), deterministic tree
//   expanding_mean_sq_kernel   per path p: out[t][p] = mean of a[u][p]^2 over u <= t  (pandas expanding().mean() of the squares)
constexpr int ROW_MOMENTS_MAX = 4;
constexpr int ROW_SEGMENTS = 4;        // blocks per row: 1025 rows x 4 fill the chip; their partial sums are added in order

// Round 6: both kernels read 16 bytes per lane and keep eight (row sums) / four (expanding mean) such loads in flight -- 32 KB
// per block, several blocks per CU -- where round 5's read 8 bytes four deep and stood at 5.0 / 4.5 TB/s (0.63 / 0.56 of the
// HBM peak; logsv_w_kernel, the same kind of pass, reaches 5.9).  The expanding mean also lost its fp64 DIVIDE per element
// (about forty instructions: 1.1 ms of issue for 2^30 elements beside 2.9 ms of memory time): the divisor of a row is the
// same for every path, so each block forms the reciprocals 1 / (t + 1) of a chunk of rows once, in LDS, by the correctly
// rounded division, and an element costs one multiplication -- within an ulp of the quotient.
constexpr int ROW_SUM_U = 8;           // 16-byte loads in flight per lane

template <int K>
__global__ __launch_bounds__(BLOCK) void row_power_sums_kernel(const double *__restrict__ a, size_t ld, size_t n_cols, double center,
                                                               double *__restrict__ partials /* [rows][ROW_SEGMENTS][2K] */)
{
    __shared__ double lds[4 * block_sum_padded(2 * K)];
    const size_t row = blockIdx.x, seg = blockIdx.y;
    const double *__restrict__ src = a + row * ld;
    size_t lo = n_cols * seg / ROW_SEGMENTS;
    const size_t hi = n_cols * (seg + 1) / ROW_SEGMENTS;
    double acc[2 * K];
#pragma unroll
    for (int j = 0; j < 2 * K; ++j) acc[j] = 0.0;
    const auto add = [&](double v) {
        const double d = v - center;
        double pw = d;
#pragma unroll
        for (int j = 0; j < 2 * K; ++j) {
            acc[j] += pw;
            pw *= d;
        }
    };
    // a segment that does not start on a 16-byte boundary gives its first element to thread 0 (block-uniform test)
    if (lo < hi && (reinterpret_cast<uintptr_t>(src + lo) & 15u) != 0u) {
        if (threadIdx.x == 0) add(src[lo]);
        ++lo;
    }
    const size_t pairs = (hi - lo) >> 1;                   // double2 elements of the aligned body
    const double2 *__restrict__ src2 = reinterpret_cast<const double2 *>(src + lo);
    size_t i = threadIdx.x;
    for (; i + (ROW_SUM_U - 1) * BLOCK < pairs; i += ROW_SUM_U * BLOCK) {
        double2 v[ROW_SUM_U];
#pragma unroll
        for (int u = 0; u < ROW_SUM_U; ++u) v[u] = src2[i + u * BLOCK];
#pragma unroll
        for (int u = 0; u < ROW_SUM_U; ++u) {
            add(v[u].x);
            add(v[u].y);
        }
    }
    for (; i < pairs; i += BLOCK) {
        const double2 v = src2[i];
        add(v.x);
        add(v.y);
    }
    if (((hi - lo) & 1u) != 0u && threadIdx.x == 0) add(src[hi - 1]);
    block_sum_store<2 * K>(acc, lds, partials + (row * ROW_SEGMENTS + seg) * (2 * K), 2 * K);
}

// out[t][p] = mean of a[u][p]^2 over u <= t.  PAIR: a lane owns the two adjacent columns 2 i, 2 i + 1 and moves 16 bytes per
// access (the caller checks that both arrays and leading dimensions allow it); otherwise one column, 8 bytes.
constexpr int EXPANDING_CHUNK = 1024;  // rows whose reciprocals a block holds in LDS at a time
template <bool PAIR>
__global__ __launch_bounds__(BLOCK) void expanding_mean_sq_kernel(const double *__restrict__ a, size_t ld, size_t n_rows, size_t n_cols,
                                                                  double *__restrict__ out, size_t ldo)
{
    __shared__ double s_inv[EXPANDING_CHUNK];
    using Vec = typename std::conditional<PAIR, double2, double>::type;
    constexpr size_t W = PAIR ? 2 : 1;
    const size_t p = (static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x) * W;
    const bool active = p < n_cols;                        // (PAIR: n_cols is even)
    double q0 = 0.0, q1 = 0.0;
    for (size_t t0 = 0; t0 < n_rows; t0 += EXPANDING_CHUNK) {
        const size_t nt = (n_rows - t0 < static_cast<size_t>(EXPANDING_CHUNK)) ? n_rows - t0 : EXPANDING_CHUNK;
        __syncthreads();                                   // everybody is done with the previous chunk's reciprocals
        for (size_t j = threadIdx.x; j < nt; j += BLOCK) s_inv[j] = 1.0 / static_cast<double>(t0 + j + 1);
        __syncthreads();
        if (!active) continue;
        const auto one = [&](const Vec &v, size_t j) {
            Vec o;
            if constexpr (PAIR) {
                q0 = fma(v.x, v.x, q0);
                q1 = fma(v.y, v.y, q1);
                o.x = q0 * s_inv[j];
                o.y = q1 * s_inv[j];
            } else {
                q0 = fma(v, v, q0);
                o = q0 * s_inv[j];
            }
            *reinterpret_cast<Vec *>(out + (t0 + j) * ldo + p) = o;
        };
        size_t j = 0;
        for (; j + 4 <= nt; j += 4) {                      // four rows in flight
            Vec v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const Vec *>(a + (t0 + j + u) * ld + p);
#pragma unroll
            for (int u = 0; u < 4; ++u) one(v[u], j + u);
        }
        for (; j < nt; ++j) one(*reinterpret_cast<const Vec *>(a + (t0 + j) * ld + p), j);
    }
}

// ---------------------------------------------------------------------------------------------------
// Rough LogSV, Markovian lift with N <= 3 factors (pricers/rough_logsv/split_simulation.py:86-128, 228-356):
// Strang splitting D(h/2) S(h) D(h/2) per step -- RK4 on the factor drift, exact lognormal step of the weighted
// factor sum -- then the log-spot / quadratic-variance update.  One lane per path, the N factors in registers.
// ---------------------------------------------------------------------------------------------------
struct RoughConsts {
    double nodes[3], w[3], v0[3], wlam[3];
    double theta, kappa1, kappa2, rho, rho_comp, volvol, inv_volvol, h, inv_h, sqrt_h, wsum, w_inv, volvol_w, w_lam_v0;
    double ito, volvol_w_sqrt_h;   // -0.5 (volvol wsum)^2 h,  volvol wsum sqrt(h)
};

template <int N>
__device__ __forceinline__ void rough_rk4_drift(const RoughConsts &c, const double (&z0)[N], double h, double (&zh)[N])
{
    double s1[N], s2[N], s3[N], s4[N], zt[N], zw, g;
    zw = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) zw += c.w[i] * z0[i];
    g = (c.kappa1 + c.kappa2 * zw) * (c.theta - zw);
#pragma unroll
    for (int i = 0; i < N; ++i) s1[i] = -c.nodes[i] * (z0[i] - c.v0[i]) + g;                         // :101-103
    zw = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        zt[i] = z0[i] + 0.5 * h * s1[i];
        zw += c.w[i] * zt[i];
    }
    g = (c.kappa1 + c.kappa2 * zw) * (c.theta - zw);
#pragma unroll
    for (int i = 0; i < N; ++i) s2[i] = -c.nodes[i] * (zt[i] - c.v0[i]) + g;                         // :106-109
    zw = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        zt[i] = z0[i] + 0.5 * h * s2[i];
        zw += c.w[i] * zt[i];
    }
    g = (c.kappa1 + c.kappa2 * zw) * (c.theta - zw);
#pragma unroll
    for (int i = 0; i < N; ++i) s3[i] = -c.nodes[i] * (zt[i] - c.v0[i]) + g;                         // :112-115
    zw = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        zt[i] = z0[i] + h * s3[i];
        zw += c.w[i] * zt[i];
    }
    g = (c.kappa1 + c.kappa2 * zw) * (c.theta - zw);
#pragma unroll
    for (int i = 0; i < N; ++i) s4[i] = -c.nodes[i] * (zt[i] - c.v0[i]) + g;                         // :118-121
#pragma unroll
    for (int i = 0; i < N; ++i) zh[i] = z0[i] + (h / 6.0) * (s1[i] + 2.0 * s2[i] + 2.0 * s3[i] + s4[i]);
}

template <int N, class Exp>
__device__ __forceinline__ void rough_step(const RoughConsts &c, double (&v)[N], double &ls, double &y, double z0, double z1,
                                           Exp &&exp_of)
{
    double d[N], sn[N], vh[N];
    rough_rk4_drift<N>(c, v, 0.5 * c.h, d);                                                      // :273
    double yw = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) yw += c.w[i] * d[i];
    const double Yh = yw * exp_of(c.ito + c.volvol_w_sqrt_h * z0);   // :238-239
    const double Q = c.w_inv * (Yh - yw);
#pragma unroll
    for (int i = 0; i < N; ++i) sn[i] = d[i] + Q;
    rough_rk4_drift<N>(c, sn, 0.5 * c.h, vh);                                                    // :275
    double volw_h = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) volw_h += c.w[i] * vh[i];
    if (!(volw_h > 0.0)) {                                                                       // NaN or <= 0, :299-300
        volw_h = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            vh[i] = 1e-6;
            volw_h += c.w[i] * vh[i];
        }
    }
    double vw = 0.0, w_lam_vol = 0.0, w_lam_vol_h = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        vw += c.w[i] * v[i];
        w_lam_vol += c.wlam[i] * v[i];
        w_lam_vol_h += c.wlam[i] * vh[i];
    }
    const double sq_vw = vw * vw, sq_vhw = volw_h * volw_h;
    const double term1 = c.inv_volvol * (((volw_h - vw) * c.inv_h + 0.5 * w_lam_vol + 0.5 * w_lam_vol_h - c.w_lam_v0) * c.w_inv
                                         - c.kappa1 * c.theta + (c.kappa1 - c.kappa2 * c.theta) * (0.5 * vw + 0.5 * volw_h)
                                         + c.kappa2 * (0.5 * sq_vw + 0.5 * sq_vhw)) * c.h;       // :319-321
    const double term2 = 0.5 * c.h * sq_vw + 0.5 * c.h * sq_vhw;
    ls = ls - 0.5 * term2 + c.rho * term1 + c.rho_comp * sqrt_pos(term2) * z1;                         // :324
    y = y + 0.5 * c.h * (vw * vw + volw_h * volw_h);                                               // :326
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = vh[i];
}

// ALL expiries of a rough-LogSV chain in ONE launch: the reference re-simulates every expiry from time 0 on its own step
// (pricers/logsv_pricer.py:1206-1216), so the expiries are INDEPENDENT simulations -- blockIdx.y picks one.  A calibration-
// sized path set (4 000-10 000 paths: 63-157 waves) fills a sixteenth of the chip, and m such launches one after the other
// each pay their own latency-bound time loop; side by side they cost the longest one.  Per expiry the arithmetic is
// the same on the same constants (the step-dependent ones travel per expiry).  Snapshot row i / m + i and partial column pair i
// belong to expiry i; the state arrays receive the LAST expiry's terminal state, as the expiry-by-expiry loop leaves them.
// This is the ONLY rough kernel: a single expiry is grid.y = 1, and a continuation from the resident state (from_origin = 0,
// the draw's steps starting at step_offset) is the same launch reading the state first -- so one launch per expiry, all
// expiries in one launch and a run cut into continued halves execute the same code and agree to the bit by construction.
// RNG = false: Z0/Z1 supplied (the reference's only interface for this model); true: counter-based draw.
struct RoughExpiries {
    double h[MAX_CHAIN_SLICES], inv_h[MAX_CHAIN_SLICES], sqrt_h[MAX_CHAIN_SLICES], ito[MAX_CHAIN_SLICES],
        volvol_w_sqrt_h[MAX_CHAIN_SLICES], forward[MAX_CHAIN_SLICES];
    int nb_steps[MAX_CHAIN_SLICES];
    int m;
};

template <int N, bool RNG>
__global__ __launch_bounds__(RNG ? RNG_BLOCK : BLOCK) void rough_logsv_expiries_kernel(
    double *__restrict__ log_s, double *__restrict__ vol, double *__restrict__ yq, size_t n, RoughConsts c, RoughExpiries e,
    const double *__restrict__ Z0, const double *__restrict__ Z1, size_t ldw, uint64_t seed, uint32_t c3, uint64_t path_offset,
    uint32_t step_offset, int from_origin, double *__restrict__ x_snap, double *__restrict__ q_snap, double *__restrict__ partials)
{
    __shared__ RngTablesLdsIf<RNG> s_tab;                  // the draw's table only where the kernel draws
    __shared__ double s_exp[256];
    const RngTables tab = stage_tables_if(s_tab, s_exp);
    const auto exp_of = [&](double a) { return exp_tab(a, s_exp); };
    const int i = blockIdx.y;                              // the expiry: block-uniform
    c.h = e.h[i];
    c.inv_h = e.inv_h[i];
    c.sqrt_h = e.sqrt_h[i];
    c.ito = e.ito[i];
    c.volvol_w_sqrt_h = e.volvol_w_sqrt_h[i];
    const int nb_steps = e.nb_steps[i];
    const size_t p = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool active = p < n;
    double ls = 0.0, y = 0.0;
    if (active) {
        double v[N];
        if (from_origin) {                                 // (log_s, v, y) = (0, v0, 0)                   :1206-1208
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = c.v0[k];
        } else {                                           // continue from the resident state (single-expiry launches)
            ls = log_s[p];
            y = yq[p];
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = vol[static_cast<size_t>(k) * n + p];
        }
        if (RNG) {
            const PhiloxLane lane = philox_prepare(seed, c3, path_offset + p);
            rng_time_loop(lane, step_offset, nb_steps, tab, [&](double z0, double z1) { rough_step<N>(c, v, ls, y, z0, z1, exp_of); });
        } else {
            const double *const w[2] = {Z0 + p, Z1 + p};
            streamed_time_loop<2, SVMC_ROUGH_STREAM_U>(w, ldw, nb_steps,
                                                       [&](const double(&z)[2]) { rough_step<N>(c, v, ls, y, z[0], z[1], exp_of); });
        }
        if (i == e.m - 1) {
#pragma unroll
            for (int k = 0; k < N; ++k) vol[static_cast<size_t>(k) * n + p] = v[k];
            log_s[p] = ls;
            yq[p] = y;
        }
    }
    const SliceOut so = {x_snap + static_cast<size_t>(i) * n, q_snap ? q_snap + static_cast<size_t>(i) * n : nullptr,
                         partials + 2 * static_cast<size_t>(i) * ((n + 63) >> 6), e.forward[i], (n + 63) >> 6};
    slice_epilogue(so, p, active, ls, y);
}

// ---------------------------------------------------------------------------------------------------
// Heston generators (pricers/heston_pricer.py:334-381; QE is new)
// ---------------------------------------------------------------------------------------------------
#ifndef SVMC_HESTON_ATTR
#define SVMC_HESTON_ATTR               // A/B hook (tools/ubench/build_variants.sh): e.g. __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
// LOOP: the form of the time loop (svmc_rng.h GenLoop): the full-launch kernels run GEN_LOOP_FULL, the few-waves ones GEN_LOOP_PAIR
// kernel-template scheme ids: the C ABI's two (SVMC_HESTON_EULER_FLOOR = 0, SVMC_HESTON_QE = 1) and QE specialised at compile
// time for parameter sets that never leave the quadratic branch and keep the martingale correction defined (QeConsts::quad_only
// && e_below_one; heston_qe_step<true>) -- the host picks it, the caller never sees it
constexpr int HESTON_QE_QUAD = 2;
constexpr bool heston_is_qe(int scheme) { return scheme == SVMC_HESTON_QE || scheme == HESTON_QE_QUAD; }
static inline int heston_kernel_scheme(int scheme, const QeConsts &qc)
{
    return (scheme == SVMC_HESTON_QE && qc.quad_only && qc.e_below_one) ? HESTON_QE_QUAD : scheme;
}

template <int LOOP, class Step>
__device__ __forceinline__ void heston_time_loop(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab, Step &&step)
{
    gen_time_loop<LOOP>(lane, step0, nb, tab, step);
}

template <int SCHEME, int LOOP>
__device__ __forceinline__ void heston_rng_body(double *__restrict__ x, double *__restrict__ var, double *__restrict__ qvar, size_t n,
                                                int nb_steps, HestonConsts c, QeConsts qc, uint64_t seed, uint32_t c3,
                                                uint64_t path_offset, uint32_t step_offset, SliceOut so, StateInit init)
{
    __shared__ RngTablesLds s_tab;
    __shared__ LogTabEntry s_log[heston_is_qe(SCHEME) ? 512 : 1];
    RngTables tab;
    if constexpr (heston_is_qe(SCHEME)) tab = stage_rng_log_tables(s_tab, s_log);
    else tab = stage_rng_tables(s_tab);
    const size_t p = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool active = p < n;
    double xv = 0.0, v = 1.0, q = 0.0;
    if (active) {
        if (init.uniform) {                                // wave-uniform
            xv = init.x0;
            v = init.vol0;
            q = init.qvar0;
        } else {
            xv = x[p];
            v = var[p];
            q = qvar[p];
        }
        const PhiloxLane lane = philox_prepare(seed, heston_is_qe(SCHEME) ? (c3 | 4u) : c3, path_offset + p);
        const HestonEulerFast ef = make_heston_euler_fast(c);
        double xacc = 0.0, vacc = 0.0;
        if constexpr (SCHEME == HESTON_QE_QUAD) {
            double vsum = 0.0, ksum = 0.0;
            const double v_first = v;
            const QeVec qv = make_qe_vec(qc);
            heston_time_loop<LOOP>(lane, step_offset, nb_steps, tab, [&](double w0, double w1) {
                heston_qe_step<true>(qc, qv, tab.log, xv, v, vsum, ksum, w0, w1, []() { return 0.0; });
            });
            heston_qe_fold(qc, xv, q, vsum, ksum, v_first, v);
        } else if constexpr (SCHEME == SVMC_HESTON_QE) {
            const PhiloxLane lane_u = philox_prepare(seed, c3 | 5u, path_offset + p);
            QeUniforms uc;
            uint32_t step = step_offset;
            double vsum = 0.0, ksum = 0.0;
            const double v_first = v;
            const QeVec qv = make_qe_vec(qc);
            heston_time_loop<LOOP>(lane, step_offset, nb_steps, tab, [&](double w0, double w1) {
                heston_qe_step(qc, qv, tab.log, xv, v, vsum, ksum, w0, w1, [&]() { return qe_uniform(lane_u, step, uc); });
                ++step;
            });
            heston_qe_fold(qc, xv, q, vsum, ksum, v_first, v);
        } else {
            v = heston_euler_guard_zero(v);
            heston_time_loop<LOOP>(lane, step_offset, nb_steps, tab,
                                  [&](double w0, double w1) { heston_euler_step_acc(ef, xacc, v, vacc, w0, w1); });
            heston_fold_acc(ef, xv, q, v, xacc, vacc);
        }
        x[p] = xv;
        var[p] = v;
        qvar[p] = q;
    }
    slice_epilogue(so, p, active, xv, q);
}

template <int SCHEME>
__global__ __launch_bounds__(RNG_BLOCK) SVMC_HESTON_ATTR void heston_rng_kernel(double *__restrict__ x, double *__restrict__ var,
                                                           double *__restrict__ qvar, size_t n, int nb_steps,
                                                           HestonConsts c, QeConsts qc, uint64_t seed,
                                                           uint32_t c3, uint64_t path_offset,
                                                           uint32_t step_offset, SliceOut so, StateInit init)
{
    heston_rng_body<SCHEME, GEN_LOOP_FULL>(x, var, qvar, n, nb_steps, c, qc, seed, c3, path_offset, step_offset, so, init);
}

// the same generator for a launch of a few waves per SIMD (few_waves_launch(); the reference's default is 10^5 paths): the
// statements -- and the bits -- of heston_rng_kernel, compiled for latency (see logsv_rng_lat_kernel)
template <int SCHEME, int LOOP, int WAVES, int TB>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void heston_rng_lat_kernel(
    double *__restrict__ x, double *__restrict__ var, double *__restrict__ qvar, size_t n, int nb_steps, HestonConsts c, QeConsts qc,
    uint64_t seed, uint32_t c3, uint64_t path_offset, uint32_t step_offset, SliceOut so, StateInit init)
{
    heston_rng_body<SCHEME, LOOP>(x, var, qvar, n, nb_steps, c, qc, seed, c3, path_offset, step_offset, so, init);
}

// whole-chain variant, as logsv_chain_rng_kernel: the slice loop inside the kernel, one launch tail per chain
struct HestonChainSlices {
    HestonConsts c[MAX_CHAIN_SLICES];
    QeConsts qc[MAX_CHAIN_SLICES];
    double forward[MAX_CHAIN_SLICES];
    int nb_steps[MAX_CHAIN_SLICES];
    int m;
};

template <int SCHEME, int LOOP>
__device__ __forceinline__ void heston_chain_rng_body(double *__restrict__ x, double *__restrict__ var, double *__restrict__ qvar,
                                                      size_t n, const HestonChainSlices &cs, uint64_t seed, uint32_t c3,
                                                      uint64_t path_offset, uint32_t step_offset, double *__restrict__ x_snap,
                                                      double *__restrict__ q_snap, double *__restrict__ partials,
                                                      const StateInit &init)
{
    __shared__ RngTablesLds s_tab;
    __shared__ LogTabEntry s_log[heston_is_qe(SCHEME) ? 512 : 1];
    RngTables tab;
    if constexpr (heston_is_qe(SCHEME)) tab = stage_rng_log_tables(s_tab, s_log);
    else tab = stage_rng_tables(s_tab);
    const size_t p = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool active = p < n;
    double xv = 0.0, v = 1.0, q = 0.0;
    if (active) {
        if (init.uniform) {                                // wave-uniform
            xv = init.x0;
            v = init.vol0;
            q = init.qvar0;
        } else {
            xv = x[p];
            v = var[p];
            q = qvar[p];
        }
    }
    const PhiloxLane lane = philox_prepare(seed, heston_is_qe(SCHEME) ? (c3 | 4u) : c3, path_offset + p);
    const PhiloxLane lane_u = philox_prepare(seed, c3 | 5u, path_offset + p);      // QE's uniforms (dead in the other schemes)
    QeUniforms uc;
    uint32_t step = step_offset;
    for (int i = 0; i < cs.m; ++i) {
        const int nb = cs.nb_steps[i];
        if (active) {
            const HestonConsts c = cs.c[i];
            const QeConsts qc = cs.qc[i];
            const HestonEulerFast ef = make_heston_euler_fast(c);
            double xacc = 0.0, vacc = 0.0;
            if constexpr (SCHEME == HESTON_QE_QUAD) {
                double vsum = 0.0, ksum = 0.0;
                const double v_first = v;
                const QeVec qv = make_qe_vec(qc);
                heston_time_loop<LOOP>(lane, step, nb, tab, [&](double w0, double w1) {
                    heston_qe_step<true>(qc, qv, tab.log, xv, v, vsum, ksum, w0, w1, []() { return 0.0; });
                });
                heston_qe_fold(qc, xv, q, vsum, ksum, v_first, v);
            } else if constexpr (SCHEME == SVMC_HESTON_QE) {
                double vsum = 0.0, ksum = 0.0;
                const double v_first = v;
                uint32_t st = step;
                const QeVec qv = make_qe_vec(qc);
                heston_time_loop<LOOP>(lane, step, nb, tab, [&](double w0, double w1) {
                    heston_qe_step(qc, qv, tab.log, xv, v, vsum, ksum, w0, w1, [&]() { return qe_uniform(lane_u, st, uc); });
                    ++st;
                });
                heston_qe_fold(qc, xv, q, vsum, ksum, v_first, v);
            } else {
                v = heston_euler_guard_zero(v);
                heston_time_loop<LOOP>(lane, step, nb, tab,
                                      [&](double w0, double w1) { heston_euler_step_acc(ef, xacc, v, vacc, w0, w1); });
                heston_fold_acc(ef, xv, q, v, xacc, vacc);
            }
        }
        step += static_cast<uint32_t>(nb);
        const SliceOut so = {x_snap + static_cast<size_t>(i) * n, q_snap ? q_snap + static_cast<size_t>(i) * n : nullptr,
                             partials + 2 * static_cast<size_t>(i) * ((n + 63) >> 6), cs.forward[i], (n + 63) >> 6};
        slice_epilogue(so, p, active, xv, q);
    }
    if (active) {
        x[p] = xv;
        var[p] = v;
        qvar[p] = q;
    }
}

template <int SCHEME>
__global__ __launch_bounds__(RNG_BLOCK) SVMC_HESTON_ATTR void heston_chain_rng_kernel(double *__restrict__ x, double *__restrict__ var,
                                                                 double *__restrict__ qvar, size_t n,
                                                                 HestonChainSlices cs, uint64_t seed, uint32_t c3,
                                                                 uint64_t path_offset, uint32_t step_offset,
                                                                 double *__restrict__ x_snap, double *__restrict__ q_snap,
                                                                 double *__restrict__ partials, StateInit init)
{
    heston_chain_rng_body<SCHEME, GEN_LOOP_FULL>(x, var, qvar, n, cs, seed, c3, path_offset, step_offset, x_snap, q_snap, partials, init);
}

template <int SCHEME, int LOOP, int WAVES, int TB>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void heston_chain_rng_lat_kernel(
    double *__restrict__ x, double *__restrict__ var, double *__restrict__ qvar, size_t n, HestonChainSlices cs, uint64_t seed,
    uint32_t c3, uint64_t path_offset, uint32_t step_offset, double *__restrict__ x_snap, double *__restrict__ q_snap,
    double *__restrict__ partials, StateInit init)
{
    heston_chain_rng_body<SCHEME, LOOP>(x, var, qvar, n, cs, seed, c3, path_offset, step_offset, x_snap, q_snap, partials, init);
}

__global__ __launch_bounds__(BLOCK) void heston_w_kernel(double *__restrict__ x, double *__restrict__ var,
                                                         double *__restrict__ qvar, size_t n, int nb_steps,
                                                         HestonConsts c, const double *__restrict__ W0,
                                                         const double *__restrict__ W1, size_t ldw)
{
    const size_t p = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
    if (p >= n) return;
    double xv = x[p], v = var[p], q = qvar[p];
    const double *const w[2] = {W0 + p, W1 + p};
    streamed_time_loop<2>(w, ldw, nb_steps, [&](const double(&z)[2]) {
        heston_euler_step(c, xv, v, q, c.sdt * z[0], c.sdt * z[1]);
    });
    x[p] = xv;
    var[p] = v;
    qvar[p] = q;
}

__global__ __launch_bounds__(BLOCK) void heston_qe_w_kernel(double *__restrict__ x, double *__restrict__ var,
                                                            double *__restrict__ qvar, size_t n, int nb_steps,
                                                            QeConsts qc, const double *__restrict__ Z0,
                                                            const double *__restrict__ Z1,
                                                            const double *__restrict__ U, size_t ldw)
{
    __shared__ LogTabEntry s_tab[512];
    const LogTabEntry *tab = stage_log_table(s_tab);
    const size_t p = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
    if (p >= n) return;
    double xv = x[p], v = var[p], q = qvar[p], vsum = 0.0, ksum = 0.0;
    const double v_first = v;
    const double *const w[3] = {Z0 + p, Z1 + p, U + p};
    const QeVec qv = make_qe_vec(qc);
    streamed_time_loop<3>(w, ldw, nb_steps, [&](const double(&z)[3]) {
        heston_qe_step(qc, qv, tab, xv, v, vsum, ksum, z[0], z[1], [&]() { return z[2]; });
    });
    heston_qe_fold(qc, xv, q, vsum, ksum, v_first, v);
    x[p] = xv;
    var[p] = v;
    qvar[p] = q;
}

// ---------------------------------------------------------------------------------------------------
// Payoff reduction (utils/mc_payoffs.py:61-88).  Deterministic two-stage sums: per-block partials in a
// fixed tree order, then one block per output column adds the partials -- no fp64 atomics, so a given
// (n_path, grid) always reproduces the same bits.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void spot_sums_kernel(const double *__restrict__ x, size_t n, double forward,
                                                          double *__restrict__ partials)
{
    __shared__ double lds[8];
    const size_t stride = static_cast<size_t>(gridDim.x) * BLOCK;
    double v[2] = {0.0, 0.0};
    for (size_t i = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x; i < n; i += stride) {
        const double sp = forward * exp_full(x[i]);        // full-range exp (user data may hold +-inf)     :61
        if (sp == sp) {                                                                         // nanmean :62
            v[0] += sp;
            v[1] += 1.0;
        }
    }
    block_sum_store<2>(v, lds, partials + 2 * static_cast<size_t>(blockIdx.x), 2);
}

// ---- per-strike payoff sums: ONE pass over the terminal values of an expiry for all of its strikes ---------------
// One block column (blockIdx.y) = one GROUP: up to KT strikes of ONE expiry (an expiry with more strikes is split
// into several groups); the group descriptor carries that expiry's snapshot pointers, forward and recentring sums.  A
// thread reads a path's x ONCE and exponentiates it ONCE for all the strikes of the group (round 1 took a pass and an
// exp per chunk of 8 strikes: three passes for the 21 strikes of the BASELINE chains).
//   plain payoffs (C / P):  pay = max(sg (u - K), 0), sg = +1 call / -1 put, computed as max(fma(sg, u, -sg K), 0) with sg
//     a wave-uniform (SGPR) operand of the FMA: fmax returns 0 for a NaN underlying, which is np.where(u > K, u - K, 0)
//     (:75-82) -- a plain payoff is never NaN, so nanmean / nanstd count EVERY path and the count column is the block's
//     path count, not an accumulator: 5 VALU instructions per strike per path (fma, max, subtract the shift, two
//     accumulates) and 4 doubles of vector state per strike (two accumulators, two constants: 192 VGPRs at 24
//     strikes).  The time loop has NO per-strike condition: the unused strikes of a group carry c = -inf.
//   inverse payoffs (IC / IP, HAS_INV): pay / spot can be NaN (0/0, inf/inf): per-strike NaN test and count kept.
constexpr int PAYOFF_GROUPS = 8;       // groups per launch: 8 x 440 B of descriptors stay inside the 4 KB kernarg segment
#ifndef SVMC_PAYOFF_PREFETCH
#define SVMC_PAYOFF_PREFETCH 4
#endif
constexpr int PAYOFF_PREFETCH = SVMC_PAYOFF_PREFETCH;     // trips between a path's load and its use (A/B: tools/ubench)
#ifndef SVMC_PAYOFF_BLOCKS
#define SVMC_PAYOFF_BLOCKS 1024
#endif
constexpr unsigned PAYOFF_BLOCKS = SVMC_PAYOFF_BLOCKS;    // path blocks per payoff launch, all groups together
constexpr int PAYOFF_KT = 22;          // strikes per group (the BASELINE chains have 21 per expiry); 23 and 24 spilled 20-44 B per lane at the 256 registers two waves per SIMD leave, so wider expiries split

struct PayoffGroup {
    const double *x, *qvar, *spot_sums;
    double forward, ttm;
    double c[PAYOFF_KT];       // -sg K; -inf for the unused strikes of a group (their payoff is 0 and their sums are dropped)
    double shift[PAYOFF_KT];   // sums are taken of (payoff - shift): removes the E[p^2] - E[p]^2 cancellation
    uint32_t inv_mask;         // bit k set: divide by the recentred spot (IC / IP)
    uint32_t put_mask;         // bit k set: put, sg = -1; clear: call, sg = +1: pay = max(fma(sg, u, c), 0)
    int k;                     // live strikes in this group
    int col;                   // first output column of the group within this launch (in strikes)
    // round 6: the expiry's spot sums formed IN this kernel from the generators' per-wave partial columns (column 0 at
    // spot_partials, column 1 spot_rows further on) instead of by a reduce launch ahead of it; null: read spot_sums
    const double *spot_partials;
    unsigned spot_rows;
};
struct PayoffGroupPack {
    PayoffGroup g[PAYOFF_GROUPS];
};
// parameter sets of one chain (the calibration's bumped evaluations) share every descriptor; their snapshots, spot sums and
// output rows lie at fixed distances, and blockIdx.z picks the set: doubles between consecutive sets (zeros for one set)
struct PayoffSetStrides {
    size_t x, q, spot;
};
static_assert(sizeof(PayoffGroupPack) + sizeof(PayoffSetStrides) + 64 <= 4096, "the payoff descriptors travel in the kernel arguments");

// number of indices i < n visited by block b of a grid-stride loop (stride = grid * BLOCK, BLOCK consecutive per block)
__device__ __forceinline__ double block_path_count(size_t n, unsigned b, unsigned grid)
{
    const size_t stride = static_cast<size_t>(grid) * BLOCK, lo = static_cast<size_t>(b) * BLOCK;
    const size_t full = n / stride, rem = n % stride;
    const size_t part = (rem > lo) ? ((rem - lo < static_cast<size_t>(BLOCK)) ? rem - lo : BLOCK) : 0;
    return static_cast<double>(full * BLOCK + part);
}

// waves per SIMD the register allocator must leave room for: the accumulators (4 or 6 registers per strike) set it
// waves per SIMD of a payoff instantiation, stated EXACTLY (min = max): left a range, the compiler aims for more waves than
// the accumulators leave registers for and spills (round 3: eleven instantiations carried 20-100 bytes of scratch per lane).
// The thresholds are the widest groups whose build metadata shows no scratch (libsvmc.isa.json "metadata";
// tests/test_host_logic.py holds every payoff instantiation to scratch_bytes == 0).
constexpr int payoff_min_waves(int kt, bool has_inv) { return kt <= (has_inv ? 5 : 12) ? 3 : 2; }

template <int KT, bool HAS_INV, bool NEED_Q>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(payoff_min_waves(KT, HAS_INV), payoff_min_waves(KT, HAS_INV)))) void payoff_group_kernel(PayoffGroupPack pack, size_t n, double *__restrict__ partials, int ld, PayoffSetStrides sets)
{
    constexpr int NACC = HAS_INV ? 3 : 2;
    __shared__ double lds[4 * block_sum_padded(NACC * KT)];
    const PayoffGroup &d = pack.g[blockIdx.y];
    const size_t set = blockIdx.z;                          // parameter set (one launch prices gridDim.z of them)
    const double *__restrict__ x = d.x + set * sets.x;
    const double *__restrict__ qvar = NEED_Q ? d.qvar + set * sets.q : nullptr;
    const double forward = d.forward, inv_ttm_arg = d.ttm;
    double spot0, spot1;
    if (d.spot_partials != nullptr) {                       // (block-uniform)
        // [sum F exp(x), count] of this expiry from the generators' per-wave rows, in block_column_sum's order: the bits of
        // reduce_columns_kernel, formed by every block for itself (a few rows per thread out of L2) instead of by a launch
        __shared__ double s_spot[2];
        const double *__restrict__ sp = d.spot_partials;   // (one-set launches only: payoff_sums_impl)
        double t0, t1;
        block_column_sum2(sp, sp + d.spot_rows, d.spot_rows, lds, t0, t1);
        if (threadIdx.x == 0) {
            s_spot[0] = t0;
            s_spot[1] = t1;
            if (blockIdx.x == 0 && d.spot_sums != nullptr) {        // ... and left where a separate reduce would have put them
                double *out = const_cast<double *>(d.spot_sums);
                out[0] = t0;
                out[1] = t1;
            }
        }
        __syncthreads();
        spot0 = s_spot[0];
        spot1 = s_spot[1];
    } else {
        const double *__restrict__ spot_sums = d.spot_sums + set * sets.spot;
        spot0 = spot_sums[0];
        spot1 = spot_sums[1];
    }
    const double corr = spot0 / spot1 - forward;                                                // :62
    const size_t stride = static_cast<size_t>(gridDim.x) * BLOCK;
    const int nk = d.k;
    const uint32_t inv_mask = d.inv_mask;
    constexpr int NSUM = block_sum_padded(NACC * KT);      // the block sum's halving tree takes multiples of 8: zeros fill up
    double acc[NSUM], sg[KT], c[KT], shift[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        sg[k] = ((d.put_mask >> k) & 1u) ? -1.0 : 1.0;     // stays wave-uniform: the scalar operand of the FMA
        // plain chains: payoff - shift = max(sg u + c, 0) - shift = max(sg u + (c - shift), -shift): the recentring rides
        // in the FMA's addend and the max's floor, four instructions per strike per path instead of five
        if constexpr (HAS_INV) {
            c[k] = d.c[k];
            shift[k] = d.shift[k];
            // these two live in VECTOR registers: left wave-uniform the compiler keeps all three constants in SGPRs, runs
            // out (3 x 24 doubles) and re-reads the spills with v_readlane_b32 -- VALU instructions on top of the ones
            // that do the work
            asm volatile("" : "+v"(c[k]), "+v"(shift[k]));
        } else {
            // ... and both are RESULTS of vector arithmetic: they stay in vector registers without a pin, and fmax()
            // knows them quiet (an opaque operand costs a second v_max per strike per path to quiet it)
            c[k] = d.c[k] - d.shift[k];
            shift[k] = 0.0 - d.shift[k];
        }
    }
#pragma unroll
    for (int j = 0; j < NSUM; ++j) acc[j] = 0.0;
    constexpr bool need_q = NEED_Q;                        // options on realised variance (SVMC_Q_VAR)

    // one path's contribution to every strike of the group
    const auto add_path = [&](double xi, double qi) {
        const double spot = forward * exp_full(xi) - corr;                                      // :61-63
        const double u = need_q ? qi / inv_ttm_arg : spot;                                      // :65-68
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            if constexpr (HAS_INV) {
                double pay = fmax(fma(sg[k], u, c[k]), 0.0);                                    // :75-82
                if (inv_mask & (1u << k)) pay = pay / spot;
                if (pay == pay) {                                                               // nanmean/nanstd
                    const double dd = pay - shift[k];
                    acc[k] += dd;
                    acc[KT + k] = fma(dd, dd, acc[KT + k]);
                    acc[2 * KT + k] += 1.0;
                }
            } else {
                const double dd = fmax(fma(sg[k], u, c[k]), shift[k]);                          // :75-82, recentred
                acc[k] += dd;
                acc[KT + k] = fma(dd, dd, acc[KT + k]);
            }
        }
    };
    // A trip is ~130 VALU instructions (~0.25 us) and the kernel runs two or three waves per SIMD (its accumulators fill the
    // register file): a load issued one trip ahead is not back in time.  The trips every lane of the block makes go
    // through the batched double buffer of the streamed generators (loads PAYOFF_PREFETCH trips ahead, no per-lane
    // branches, so the waits are counted ones); the last, ragged trip is taken on its own.
    const size_t i0 = static_cast<size_t>(blockIdx.x) * BLOCK + threadIdx.x;
    const size_t block_last = static_cast<size_t>(blockIdx.x) * BLOCK + (BLOCK - 1);
    const int full_trips = (n > block_last) ? static_cast<int>((n - 1 - block_last) / stride + 1) : 0;     // block-uniform
    constexpr int PF = (KT >= (HAS_INV ? 15 : 22)) ? 2 : PAYOFF_PREFETCH;      // the widest groups have no registers to spare
    if constexpr (need_q) {
        const double *const w[2] = {x + i0, qvar + i0};
        streamed_time_loop<2, PF>(w, stride, full_trips, [&](const double(&v)[2]) { add_path(v[0], v[1]); });
    } else {
        const double *const w[1] = {x + i0};
        streamed_time_loop<1, PF>(w, stride, full_trips, [&](const double(&v)[1]) { add_path(v[0], 0.0); });
    }
    for (size_t i = i0 + static_cast<size_t>(full_trips) * stride; i < n; i += stride) add_path(x[i], need_q ? qvar[i] : 0.0);
    // output row layout: [sum d, sum d^2, count] per strike, interleaved, at column 3 (col + k)
    // (sets interleave within a block row: partials[path block][set][column], so that one column reduce serves them all)
    double *row = partials + (static_cast<size_t>(blockIdx.x) * gridDim.z + set) * ld + 3 * d.col;
    const double cnt = block_path_count(n, blockIdx.x, gridDim.x);
    block_sum_apply<NSUM>(acc, lds, NACC * KT, [&](int j, double t) {
        const int which = j / KT, k = j - which * KT;
        if (k < nk) {
            row[3 * k + which] = t;
            if (!HAS_INV && which == 0) row[3 * k + 2] = cnt;
        }
    });
}

// out[j] = sum_r partials[r * ld + j]; one block per column j
// element (row r, column j) sits at partials[r * row_stride + j * col_stride]: row-major block partials pass (ld, 1), the
// generators' per-wave spot partials are column-major and pass (1, rows)
__global__ __launch_bounds__(BLOCK) void reduce_columns_kernel(const double *__restrict__ partials_base, unsigned n_rows,
                                                               size_t row_stride, size_t col_stride, double *__restrict__ out)
{
    __shared__ double lds[4];
    const int j = blockIdx.x;
    const double t = block_column_sum(partials_base + static_cast<size_t>(j) * col_stride, n_rows, row_stride, lds);
    if (threadIdx.x == 0) out[j] = t;
}

static inline unsigned reduce_grid(size_t n)
{
    const size_t g = (n + BLOCK - 1) / BLOCK;
    return static_cast<unsigned>(g < 1 ? 1 : (g > MAX_REDUCE_GRID ? MAX_REDUCE_GRID : g));
}

static int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SVMC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return SVMC_OK;
}

static int check_state(const char *fn, const double *x, const double *v, const double *q, int nb_steps, double dt)
{
    if (x == nullptr || v == nullptr || q == nullptr) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": null state pointer");
    if (nb_steps < 0) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": nb_steps < 0");
    if (!(dt > 0.0)) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": dt must be positive");
    return SVMC_OK;
}

static inline uint32_t make_c3(uint32_t call_id) { return (call_id << 8); }

}  // namespace svmc

using namespace svmc;

extern "C" {

int svmc_fill_state(double *x, double *vol, double *qvar, size_t n_path, double x0, double vol0, double qvar0,
                    svmc_stream_t stream)
{
    SVMC_REQUIRE(x && vol && qvar, "svmc_fill_state: null state pointer");
    if (n_path == 0) return SVMC_OK;
    hipLaunchKernelGGL(fill_state_kernel, dim3(grid_for(n_path)), dim3(BLOCK), 0, as_stream(stream), x, vol, qvar,
                       n_path, x0, vol0, qvar0);
    return check_launch("svmc_fill_state");
}

int svmc_fill_normals(double *W0, double *W1, size_t ldw, size_t n_path, int nb_steps, uint64_t seed,
                      uint32_t call_id, uint64_t path_offset, uint32_t step_offset, svmc_stream_t stream)
{
    SVMC_REQUIRE(W0 && W1, "svmc_fill_normals: null output");
    SVMC_REQUIRE(ldw >= n_path && nb_steps >= 0, "svmc_fill_normals: ldw < n_path or nb_steps < 0");
    SVMC_REQUIRE(call_id < (1u << 24), "svmc_fill_normals: call_id must fit 24 bits");
    if (n_path == 0 || nb_steps == 0) return SVMC_OK;
    hipLaunchKernelGGL(fill_normals_kernel, dim3(rng_grid(n_path)), dim3(rng_block()), 0, as_stream(stream), W0, W1, ldw,
                       n_path, nb_steps, seed, make_c3(call_id), path_offset, step_offset);
    return check_launch("svmc_fill_normals");
}

int svmc_fill_uniforms(double *U, size_t ldw, size_t n_path, int nb_steps, uint64_t seed, uint32_t call_id,
                       uint64_t path_offset, uint32_t step_offset, svmc_stream_t stream)
{
    SVMC_REQUIRE(U, "svmc_fill_uniforms: null output");
    SVMC_REQUIRE(ldw >= n_path && nb_steps >= 0, "svmc_fill_uniforms: ldw < n_path or nb_steps < 0");
    SVMC_REQUIRE(call_id < (1u << 24), "svmc_fill_uniforms: call_id must fit 24 bits");
    if (n_path == 0 || nb_steps == 0) return SVMC_OK;
    hipLaunchKernelGGL(fill_uniforms_kernel, dim3(grid_for(n_path)), dim3(BLOCK), 0, as_stream(stream), U, ldw,
                       n_path, nb_steps, seed, make_c3(call_id), path_offset, step_offset);
    return check_launch("svmc_fill_uniforms");
}

static int logsv_rng_launch(const char *fn, double *x, double *sigma, double *qvar, size_t n_path, int nb_steps,
                            double dt, double theta, double kappa1, double kappa2, double beta, double volvol,
                            double vol_backbone_eta, int is_spot_measure, uint64_t seed, uint32_t call_id,
                            uint64_t path_offset, uint32_t step_offset, const SliceOut &so, svmc_stream_t stream,
                            const StateInit &init = StateInit())
{
    if (int rc = check_state(fn, x, sigma, qvar, nb_steps, dt)) return rc;
    if (call_id >= (1u << 24)) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": call_id must fit 24 bits");
    if (n_path == 0) return SVMC_OK;
    LogsvFast c = logsv_fast_in_log_units(make_logsv_fast(
        make_logsv_consts(dt, theta, kappa1, kappa2, beta, volvol, vol_backbone_eta, is_spot_measure)));
    if (const LogsvLatVariant *v = logsv_lat_variant(n_path))
        hipLaunchKernelGGL(v->slice, dim3(lat_grid(n_path, v->block)), dim3(v->block), 0, as_stream(stream), x, sigma, qvar, n_path,
                           nb_steps, c, seed, make_c3(call_id), path_offset, step_offset, so, init, armed_probe());
    else
        hipLaunchKernelGGL(logsv_rng_kernel, dim3(rng_grid(n_path)), dim3(rng_block()), 0, as_stream(stream), x, sigma, qvar,
                           n_path, nb_steps, c, seed, make_c3(call_id), path_offset, step_offset, so, init, armed_probe());
    return check_launch(fn);
}

// the [wave][2] spot partials of a fused slice kernel -> spot_sums[2]
static int finish_slice_sums(const char *fn, unsigned block_rows, double *spot_sums, void *workspace, svmc_stream_t stream)
{
    hipLaunchKernelGGL(reduce_columns_kernel, dim3(2), dim3(BLOCK), 0, as_stream(stream),
                       static_cast<const double *>(workspace), block_rows, size_t(1), static_cast<size_t>(block_rows), spot_sums);
    return check_launch(fn);
}

// allow_null_spot (the two on-device-RNG slice launchers, for their internal callers only -- the public entry points reject a null
// spot_sums first): with spot_sums == nullptr the launch leaves its per-wave partial columns in the workspace unreduced, and the
// payoff kernel of svmc_chain.hip's one-device tail sums them itself
static int check_slice_args(const char *fn, size_t n_path, const double *x_snapshot, const double *spot_sums,
                            const void *workspace, size_t workspace_bytes, bool allow_null_spot = false)
{
    if (x_snapshot == nullptr || (spot_sums == nullptr && !allow_null_spot) || workspace == nullptr)
        return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": null snapshot / spot_sums / workspace");
    if (workspace_bytes < static_cast<size_t>(wave_rows(n_path)) * 2 * sizeof(double))
        return fail(SVMC_ERR_WORKSPACE, std::string(fn) + ": workspace too small (svmc_slice_workspace_bytes)");
    return SVMC_OK;
}

int svmc_logsv_terminal_rng(double *x, double *sigma, double *qvar, size_t n_path, int nb_steps, double dt,
                            double theta, double kappa1, double kappa2, double beta, double volvol,
                            double vol_backbone_eta, int is_spot_measure, uint64_t seed, uint32_t call_id,
                            uint64_t path_offset, uint32_t step_offset, svmc_stream_t stream)
{
    const SliceOut none = {nullptr, nullptr, nullptr, 0.0};
    return logsv_rng_launch("svmc_logsv_terminal_rng", x, sigma, qvar, n_path, nb_steps, dt, theta, kappa1, kappa2, beta,
                            volvol, vol_backbone_eta, is_spot_measure, seed, call_id, path_offset, step_offset, none,
                            stream);
}

}  // extern "C"

static int logsv_slice_rng_impl(const char *fn, const StateInit &init, double *x, double *sigma, double *qvar, size_t n_path, int nb_steps, double dt, double theta,
                         double kappa1, double kappa2, double beta, double volvol, double vol_backbone_eta,
                         int is_spot_measure, uint64_t seed, uint32_t call_id, uint64_t path_offset,
                         uint32_t step_offset, double forward, double *x_snapshot, double *qvar_snapshot,
                         double *spot_sums, void *workspace, size_t workspace_bytes, svmc_stream_t stream)
{
    if (int rc = check_slice_args(fn, n_path, x_snapshot, spot_sums, workspace, workspace_bytes, true)) return rc;
    SVMC_REQUIRE(n_path > 0 && nb_steps > 0, std::string(fn) + ": n_path and nb_steps must be positive");
    const SliceOut so = {x_snapshot, qvar_snapshot, static_cast<double *>(workspace), forward, wave_rows(n_path)};
    if (int rc = logsv_rng_launch(fn, x, sigma, qvar, n_path, nb_steps, dt, theta, kappa1, kappa2, beta, volvol,
                                  vol_backbone_eta, is_spot_measure, seed, call_id, path_offset, step_offset, so, stream, init))
        return rc;
    if (spot_sums == nullptr) return SVMC_OK;
    return finish_slice_sums(fn, wave_rows(n_path), spot_sums, workspace, stream);
}

extern "C" {

int svmc_logsv_slice_rng(double *x, double *sigma, double *qvar, size_t n_path, int nb_steps, double dt, double theta,
                         double kappa1, double kappa2, double beta, double volvol, double vol_backbone_eta,
                         int is_spot_measure, uint64_t seed, uint32_t call_id, uint64_t path_offset,
                         uint32_t step_offset, double forward, double *x_snapshot, double *qvar_snapshot,
                         double *spot_sums, void *workspace, size_t workspace_bytes, svmc_stream_t stream)
{
    SVMC_REQUIRE(spot_sums != nullptr, "svmc_logsv_slice_rng: null spot_sums");
    return logsv_slice_rng_impl("svmc_logsv_slice_rng", StateInit(), x, sigma, qvar, n_path, nb_steps, dt, theta, kappa1, kappa2,
                                beta, volvol, vol_backbone_eta, is_spot_measure, seed, call_id, path_offset, step_offset,
                                forward, x_snapshot, qvar_snapshot, spot_sums, workspace, workspace_bytes, stream);
}

int svmc_logsv_slice_rng_from(double x0, double sigma0, double qvar0, double *x, double *sigma, double *qvar, size_t n_path,
                              int nb_steps, double dt, double theta, double kappa1, double kappa2, double beta, double volvol,
                              double vol_backbone_eta, int is_spot_measure, uint64_t seed, uint32_t call_id,
                              uint64_t path_offset, uint32_t step_offset, double forward, double *x_snapshot,
                              double *qvar_snapshot, double *spot_sums, void *workspace, size_t workspace_bytes,
                              svmc_stream_t stream)
{
    SVMC_REQUIRE(spot_sums != nullptr, "svmc_logsv_slice_rng_from: null spot_sums");
    const StateInit init = {1, x0, sigma0, qvar0};
    return logsv_slice_rng_impl("svmc_logsv_slice_rng_from", init, x, sigma, qvar, n_path, nb_steps, dt, theta, kappa1, kappa2,
                                beta, volvol, vol_backbone_eta, is_spot_measure, seed, call_id, path_offset, step_offset,
                                forward, x_snapshot, qvar_snapshot, spot_sums, workspace, workspace_bytes, stream);
}

}  // extern "C"

static int logsv_chain_rng_impl(const char *fn, const StateInit &init, double *x, double *sigma, double *qvar, size_t n_path, int n_slices, const int *nb_steps_host,
                         const double *dts_host, const double *etas_host, const double *forwards_host, double theta,
                         double kappa1, double kappa2, double beta, double volvol, int is_spot_measure, uint64_t seed,
                         uint32_t call_id, uint64_t path_offset, uint32_t step_offset, double *x_snapshots,
                         double *qvar_snapshots, double *spot_sums, void *workspace, size_t workspace_bytes,
                         svmc_stream_t stream)
{
    SVMC_REQUIRE(x && sigma && qvar && x_snapshots && workspace, std::string(fn) + ": null pointer");
    SVMC_REQUIRE(nb_steps_host && dts_host && forwards_host && n_slices >= 1, std::string(fn) + ": null grids / no slices");
    SVMC_REQUIRE(spot_sums != nullptr || n_slices <= MAX_CHAIN_SLICES, std::string(fn) + ": unreduced partials need one launch");
    SVMC_REQUIRE(call_id < (1u << 24), std::string(fn) + ": call_id must fit 24 bits");
    SVMC_REQUIRE(n_path > 0, std::string(fn) + ": n_path must be positive");
    for (int i = 0; i < n_slices; ++i)
        SVMC_REQUIRE(nb_steps_host[i] > 0 && dts_host[i] > 0.0, std::string(fn) + ": nb_steps and dt must be positive");
    const unsigned g = chain_grid(n_path);
    for (int i0 = 0; i0 < n_slices; i0 += MAX_CHAIN_SLICES) {
        ChainSlices cs;
        cs.m = (n_slices - i0 < MAX_CHAIN_SLICES) ? (n_slices - i0) : MAX_CHAIN_SLICES;
        if (workspace_bytes < static_cast<size_t>(wave_rows(n_path)) * 2 * cs.m * sizeof(double))
            return fail(SVMC_ERR_WORKSPACE, std::string(fn) + ": workspace too small (svmc_slice_workspace_bytes)");
        cs.total_steps = 0;
        for (int i = 0; i < MAX_CHAIN_SLICES; ++i) {
            const int j = (i < cs.m) ? i0 + i : i0;        // unused entries repeat a valid one
            cs.c[i] = logsv_fast_in_log_units(make_logsv_fast(make_logsv_consts(
                dts_host[j], theta, kappa1, kappa2, beta, volvol, etas_host ? etas_host[j] : 1.0, is_spot_measure)));
            cs.forward[i] = forwards_host[j];
            cs.nb_steps[i] = (i < cs.m) ? nb_steps_host[j] : 0;
            cs.total_steps += cs.nb_steps[i];
        }
        double *xs = x_snapshots + static_cast<size_t>(i0) * n_path;
        double *qs = qvar_snapshots ? qvar_snapshots + static_cast<size_t>(i0) * n_path : nullptr;
        if (const LogsvLatVariant *v = logsv_lat_variant(n_path))
            hipLaunchKernelGGL(v->chain, dim3(lat_grid(n_path, v->block)), dim3(v->block), 0, as_stream(stream), x, sigma, qvar, n_path,
                               cs, seed, make_c3(call_id), path_offset, step_offset, xs, qs, static_cast<double *>(workspace),
                               (i0 == 0) ? init : StateInit(), armed_probe());
        else
            hipLaunchKernelGGL(logsv_chain_rng_kernel, dim3(g), dim3(CHAIN_BLOCK), 0, as_stream(stream), x, sigma, qvar, n_path,
                               cs, seed, make_c3(call_id), path_offset, step_offset, xs, qs, static_cast<double *>(workspace),
                               (i0 == 0) ? init : StateInit(), armed_probe());
        if (spot_sums != nullptr)
            hipLaunchKernelGGL(reduce_columns_kernel, dim3(2 * cs.m), dim3(BLOCK), 0, as_stream(stream),
                               static_cast<const double *>(workspace), wave_rows(n_path), size_t(1), static_cast<size_t>(wave_rows(n_path)), spot_sums + 2 * i0);
        if (int rc = check_launch(fn)) return rc;
        step_offset += static_cast<uint32_t>(cs.total_steps);
    }
    return SVMC_OK;
}

extern "C" {

int svmc_logsv_chain_rng(double *x, double *sigma, double *qvar, size_t n_path, int n_slices, const int *nb_steps_host,
                         const double *dts_host, const double *etas_host, const double *forwards_host, double theta,
                         double kappa1, double kappa2, double beta, double volvol, int is_spot_measure, uint64_t seed,
                         uint32_t call_id, uint64_t path_offset, uint32_t step_offset, double *x_snapshots,
                         double *qvar_snapshots, double *spot_sums, void *workspace, size_t workspace_bytes,
                         svmc_stream_t stream)
{
    SVMC_REQUIRE(spot_sums != nullptr, "svmc_logsv_chain_rng: null spot_sums");
    return logsv_chain_rng_impl("svmc_logsv_chain_rng", StateInit(), x, sigma, qvar, n_path, n_slices, nb_steps_host, dts_host,
                                etas_host, forwards_host, theta, kappa1, kappa2, beta, volvol, is_spot_measure, seed, call_id,
                                path_offset, step_offset, x_snapshots, qvar_snapshots, spot_sums, workspace, workspace_bytes,
                                stream);
}

int svmc_logsv_chain_rng_from(double x0, double sigma0, double qvar0, double *x, double *sigma, double *qvar, size_t n_path,
                              int n_slices, const int *nb_steps_host, const double *dts_host, const double *etas_host,
                              const double *forwards_host, double theta, double kappa1, double kappa2, double beta,
                              double volvol, int is_spot_measure, uint64_t seed, uint32_t call_id, uint64_t path_offset,
                              uint32_t step_offset, double *x_snapshots, double *qvar_snapshots, double *spot_sums,
                              void *workspace, size_t workspace_bytes, svmc_stream_t stream)
{
    SVMC_REQUIRE(spot_sums != nullptr, "svmc_logsv_chain_rng_from: null spot_sums");
    const StateInit init = {1, x0, sigma0, qvar0};
    return logsv_chain_rng_impl("svmc_logsv_chain_rng_from", init, x, sigma, qvar, n_path, n_slices, nb_steps_host, dts_host,
                                etas_host, forwards_host, theta, kappa1, kappa2, beta, volvol, is_spot_measure, seed, call_id,
                                path_offset, step_offset, x_snapshots, qvar_snapshots, spot_sums, workspace, workspace_bytes,
                                stream);
}

static int logsv_w_launch(const char *fn, double *x, double *sigma, double *qvar, size_t n_path, int nb_steps, double dt,
                          double theta, double kappa1, double kappa2, double beta, double volvol,
                          double vol_backbone_eta, int is_spot_measure, const double *W0, const double *W1, size_t ldw,
                          const SliceOut &so, svmc_stream_t stream)
{
    if (int rc = check_state(fn, x, sigma, qvar, nb_steps, dt)) return rc;
    if (W0 == nullptr || W1 == nullptr) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": null W0/W1");
    if (ldw < n_path) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": ldw < n_path");
    if (n_path == 0 || (nb_steps == 0 && so.partials == nullptr)) return SVMC_OK;
    const LogsvConsts c = make_logsv_consts(dt, theta, kappa1, kappa2, beta, volvol, vol_backbone_eta, is_spot_measure);
    hipLaunchKernelGGL(logsv_w_kernel, dim3(grid_for(n_path)), dim3(BLOCK), 0, as_stream(stream), x, sigma, qvar,
                       n_path, nb_steps, c, W0, W1, ldw, so);
    return check_launch(fn);
}

int svmc_logsv_terminal_w(double *x, double *sigma, double *qvar, size_t n_path, int nb_steps, double dt,
                          double theta, double kappa1, double kappa2, double beta, double volvol,
                          double vol_backbone_eta, int is_spot_measure, const double *W0, const double *W1,
                          size_t ldw, svmc_stream_t stream)
{
    const SliceOut none = {nullptr, nullptr, nullptr, 0.0};
    return logsv_w_launch("svmc_logsv_terminal_w", x, sigma, qvar, n_path, nb_steps, dt, theta, kappa1, kappa2, beta,
                          volvol, vol_backbone_eta, is_spot_measure, W0, W1, ldw, none, stream);
}

int svmc_logsv_slice_w(double *x, double *sigma, double *qvar, size_t n_path, int nb_steps, double dt, double theta,
                       double kappa1, double kappa2, double beta, double volvol, double vol_backbone_eta,
                       int is_spot_measure, const double *W0, const double *W1, size_t ldw, double forward,
                       double *x_snapshot, double *qvar_snapshot, double *spot_sums, void *workspace,
                       size_t workspace_bytes, svmc_stream_t stream)
{
    const char *fn = "svmc_logsv_slice_w";
    if (int rc = check_slice_args(fn, n_path, x_snapshot, spot_sums, workspace, workspace_bytes)) return rc;
    SVMC_REQUIRE(n_path > 0 && nb_steps > 0, "svmc_logsv_slice_w: n_path and nb_steps must be positive");
    const SliceOut so = {x_snapshot, qvar_snapshot, static_cast<double *>(workspace), forward, wave_rows(n_path)};
    if (int rc = logsv_w_launch(fn, x, sigma, qvar, n_path, nb_steps, dt, theta, kappa1, kappa2, beta, volvol,
                                vol_backbone_eta, is_spot_measure, W0, W1, ldw, so, stream))
        return rc;
    return finish_slice_sums(fn, wave_rows(n_path), spot_sums, workspace, stream);
}

}  // extern "C"

namespace svmc {

// internal entry points of the graph-replayed chain driver (svmc_chain.hip); same checks as their C-ABI twins
int fill_state_indirect(double *x, double *vol, double *qvar, size_t n_path, const double *vol0_dev, hipStream_t stream)
{
    if (n_path == 0) return SVMC_OK;
    hipLaunchKernelGGL(fill_state_indirect_kernel, dim3(grid_for(n_path)), dim3(BLOCK), 0, stream, x, vol, qvar, n_path,
                       vol0_dev);
    return check_launch("fill_state_indirect");
}

int logsv_slice_w_indirect(double *x, double *sigma, double *qvar, size_t n_path, int nb_steps,
                           const double *consts_dev, const double *W0, const double *W1, size_t ldw, double forward,
                           double *x_snapshot, double *qvar_snapshot, double *spot_sums, void *workspace,
                           size_t workspace_bytes, hipStream_t stream)
{
    const char *fn = "logsv_slice_w_indirect";
    if (int rc = check_slice_args(fn, n_path, x_snapshot, spot_sums, workspace, workspace_bytes)) return rc;
    if (W0 == nullptr || W1 == nullptr || ldw < n_path || nb_steps <= 0 || n_path == 0)
        return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": bad randoms / sizes");
    const SliceOut so = {x_snapshot, qvar_snapshot, static_cast<double *>(workspace), forward, wave_rows(n_path)};
    hipLaunchKernelGGL(logsv_w_indirect_kernel, dim3(grid_for(n_path)), dim3(BLOCK), 0, stream, x, sigma, qvar, n_path,
                       nb_steps, reinterpret_cast<const LogsvConsts *>(consts_dev), W0, W1, ldw, so);
    if (int rc = check_launch(fn)) return rc;
    return finish_slice_sums(fn, wave_rows(n_path), spot_sums, workspace, reinterpret_cast<svmc_stream_t>(stream));
}

// every expiry of a chain on resident randoms in one launch + one column reduce (graph-replayed driver, svmc_chain.hip):
// consts_dev = [m] LogsvConsts, vol0_dev = the initial volatility (state is initialised in the kernel), x_snapshots [m][n],
// qvar_snapshots [m][n] or null, spot_sums [2m]
int logsv_chain_w_indirect(double *x, double *sigma, double *qvar, size_t n_path, int n_slices, const int *nb_steps_host,
                           const double *consts_dev, const double *vol0_dev, const double *const *W0s, const double *const *W1s,
                           size_t ldw, const double *forwards_host, double *x_snapshots, double *qvar_snapshots,
                           double *spot_sums, void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    const char *fn = "logsv_chain_w_indirect";
    if (n_slices < 1 || n_slices > MAX_CHAIN_SLICES || n_path == 0 || ldw < n_path)
        return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": bad sizes");
    const unsigned g = grid_for(n_path);
    if (workspace_bytes < static_cast<size_t>(wave_rows(n_path)) * 2 * n_slices * sizeof(double))
        return fail(SVMC_ERR_WORKSPACE, std::string(fn) + ": workspace too small (svmc_slice_workspace_bytes)");
    ChainWSlices cs;
    cs.m = n_slices;
    for (int i = 0; i < MAX_CHAIN_SLICES; ++i) {
        const int j = (i < n_slices) ? i : 0;
        if (W0s[j] == nullptr || W1s[j] == nullptr || nb_steps_host[j] <= 0)
            return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": bad randoms / step counts");
        cs.W0[i] = W0s[j];
        cs.W1[i] = W1s[j];
        cs.forward[i] = forwards_host[j];
        cs.nb_steps[i] = (i < n_slices) ? nb_steps_host[j] : 0;
    }
    hipLaunchKernelGGL(logsv_chain_w_indirect_kernel, dim3(g), dim3(BLOCK), 0, stream, x, sigma, qvar, n_path, cs,
                       reinterpret_cast<const LogsvConsts *>(consts_dev), vol0_dev, ldw, x_snapshots, qvar_snapshots,
                       static_cast<double *>(workspace));
    hipLaunchKernelGGL(reduce_columns_kernel, dim3(2 * n_slices), dim3(BLOCK), 0, stream,
                       static_cast<const double *>(workspace), wave_rows(n_path), size_t(1), static_cast<size_t>(wave_rows(n_path)), spot_sums);
    return check_launch(fn);
}

// P parameter sets of a chain on resident randoms in one launch + one column reduce (svmc_logsv_chain_price_fixed_sets):
// consts_dev = [m][P] LogsvConsts, vol0_dev = [P]; snapshots [P m][n] set-major, spot_sums [P m][2]
template <int P>
static void launch_chain_w_sets(unsigned g, hipStream_t stream, size_t n_path, const ChainWSlices &cs, const double *consts_dev,
                                const double *vol0_dev, size_t ldw, double *x_snapshots, double *qvar_snapshots, void *workspace)
{
    hipLaunchKernelGGL(logsv_chain_w_sets_kernel<P>, dim3(g), dim3(BLOCK), 0, stream, n_path, cs,
                       reinterpret_cast<const LogsvConsts *>(consts_dev), vol0_dev, ldw, x_snapshots, qvar_snapshots,
                       static_cast<double *>(workspace));
}

int logsv_chain_w_sets(size_t n_path, int n_sets, int n_slices, const int *nb_steps_host, const double *consts_dev,
                       const double *vol0_dev, const double *const *W0s, const double *const *W1s, size_t ldw,
                       const double *forwards_host, double *x_snapshots, double *qvar_snapshots, double *spot_sums,
                       void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    const char *fn = "logsv_chain_w_sets";
    if (n_slices < 1 || n_slices > MAX_CHAIN_SLICES || n_sets < 2 || n_sets > MAX_CHAIN_SETS || n_path == 0 || ldw < n_path)
        return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": bad sizes");
    const unsigned g = grid_for(n_path);
    const int cols = 2 * n_slices * n_sets;
    if (workspace_bytes < static_cast<size_t>(wave_rows(n_path)) * cols * sizeof(double))
        return fail(SVMC_ERR_WORKSPACE, std::string(fn) + ": workspace too small (svmc_slice_workspace_bytes)");
    ChainWSlices cs;
    cs.m = n_slices;
    for (int i = 0; i < MAX_CHAIN_SLICES; ++i) {
        const int j = (i < n_slices) ? i : 0;
        if (W0s[j] == nullptr || W1s[j] == nullptr || nb_steps_host[j] <= 0)
            return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": bad randoms / step counts");
        cs.W0[i] = W0s[j];
        cs.W1[i] = W1s[j];
        cs.forward[i] = forwards_host[j];
        cs.nb_steps[i] = (i < n_slices) ? nb_steps_host[j] : 0;
    }
    switch (n_sets) {
    case 2: launch_chain_w_sets<2>(g, stream, n_path, cs, consts_dev, vol0_dev, ldw, x_snapshots, qvar_snapshots, workspace); break;
    case 3: launch_chain_w_sets<3>(g, stream, n_path, cs, consts_dev, vol0_dev, ldw, x_snapshots, qvar_snapshots, workspace); break;
    case 4: launch_chain_w_sets<4>(g, stream, n_path, cs, consts_dev, vol0_dev, ldw, x_snapshots, qvar_snapshots, workspace); break;
    case 5: launch_chain_w_sets<5>(g, stream, n_path, cs, consts_dev, vol0_dev, ldw, x_snapshots, qvar_snapshots, workspace); break;
    case 6: launch_chain_w_sets<6>(g, stream, n_path, cs, consts_dev, vol0_dev, ldw, x_snapshots, qvar_snapshots, workspace); break;
    case 7: launch_chain_w_sets<7>(g, stream, n_path, cs, consts_dev, vol0_dev, ldw, x_snapshots, qvar_snapshots, workspace); break;
    default: launch_chain_w_sets<8>(g, stream, n_path, cs, consts_dev, vol0_dev, ldw, x_snapshots, qvar_snapshots, workspace); break;
    }
    hipLaunchKernelGGL(reduce_columns_kernel, dim3(cols), dim3(BLOCK), 0, stream, static_cast<const double *>(workspace),
                       wave_rows(n_path), size_t(1), static_cast<size_t>(wave_rows(n_path)), spot_sums);
    return check_launch(fn);
}

// P parameter sets of a chain on randoms regenerated from (seed, call_id) in one launch + one column reduce
// (svmc_logsv_chain_price_frozen_sets): consts_dev = [m][P] LogsvFast in log units (logsv_fast_to_doubles), vol0_dev = [P];
// snapshots [P m][n] set-major, spot_sums [P m][2]
template <int P>
static void launch_chain_rng_sets(unsigned g, hipStream_t stream, size_t n_path, const ChainRngSetsSlices &cs,
                                  const double *consts_dev, const double *vol0_dev, int p_total, int s0, uint64_t seed,
                                  uint32_t c3, uint64_t path_offset, double *x_snapshots, double *qvar_snapshots, void *workspace,
                                  uint64_t *probe)
{
    hipLaunchKernelGGL(logsv_chain_rng_sets_kernel<P>, dim3(g), dim3(BLOCK), 0, stream, n_path, cs,
                       reinterpret_cast<const LogsvFast *>(consts_dev), vol0_dev, p_total, s0, seed, c3, path_offset, 0u,
                       x_snapshots, qvar_snapshots, static_cast<double *>(workspace), probe);
}

int logsv_chain_rng_sets(size_t n_path, int n_sets, int n_slices, const int *nb_steps_host, const double *consts_dev,
                         const double *vol0_dev, const double *forwards_host, uint64_t seed, uint32_t call_id,
                         uint64_t path_offset, double *x_snapshots, double *qvar_snapshots, double *spot_sums, void *workspace,
                         size_t workspace_bytes, hipStream_t stream, bool allow_probe)
{
    const char *fn = "logsv_chain_rng_sets";
    if (n_slices < 1 || n_slices > MAX_CHAIN_SLICES || n_sets < 1 || n_sets > MAX_CHAIN_SETS || n_path == 0)
        return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": bad sizes");
    if (call_id >= (1u << 24)) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": call_id must fit 24 bits");
    const int cols = 2 * n_slices * n_sets;
    if (workspace_bytes < static_cast<size_t>(wave_rows(n_path)) * cols * sizeof(double))
        return fail(SVMC_ERR_WORKSPACE, std::string(fn) + ": workspace too small (svmc_slice_workspace_bytes)");
    ChainRngSetsSlices cs;
    cs.m = n_slices;
    for (int i = 0; i < MAX_CHAIN_SLICES; ++i) {
        const int j = (i < n_slices) ? i : 0;
        if (nb_steps_host[j] <= 0) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": bad step counts");
        cs.forward[i] = forwards_host[j];
        cs.nb_steps[i] = (i < n_slices) ? nb_steps_host[j] : 0;
    }
    const uint32_t c3 = make_c3(call_id);
    const unsigned g = grid_for(n_path);
    uint64_t *probe = allow_probe ? armed_probe() : nullptr;
#define SVMC_RNG_SETS_CASE(P, S0)                                                                                               \
    launch_chain_rng_sets<P>(g, stream, n_path, cs, consts_dev, vol0_dev, n_sets, S0, seed, c3, path_offset, x_snapshots,       \
                             qvar_snapshots, workspace, probe)
    switch (n_sets) {
    case 1: SVMC_RNG_SETS_CASE(1, 0); break;
    case 2: SVMC_RNG_SETS_CASE(2, 0); break;
    case 3: SVMC_RNG_SETS_CASE(3, 0); break;
    case 4: SVMC_RNG_SETS_CASE(4, 0); break;
    case 5: SVMC_RNG_SETS_CASE(5, 0); break;
    case 6: SVMC_RNG_SETS_CASE(6, 0); break;
    case 7: SVMC_RNG_SETS_CASE(7, 0); break;
    default:                                   // eight: two launches of four (register budget, see the kernel)
        SVMC_RNG_SETS_CASE(4, 0);
        SVMC_RNG_SETS_CASE(4, 4);
        break;
    }
#undef SVMC_RNG_SETS_CASE
    if (spot_sums != nullptr)                  // (null: the caller's payoff kernel sums the partial columns itself)
        hipLaunchKernelGGL(reduce_columns_kernel, dim3(cols), dim3(BLOCK), 0, stream, static_cast<const double *>(workspace),
                           wave_rows(n_path), size_t(1), static_cast<size_t>(wave_rows(n_path)), spot_sums);
    return check_launch(fn);
}

// the constants of one (slice, set) pair as the generators take them: LogsvFast in log units (what logsv_rng_launch passes)
void logsv_fast_to_doubles(double dt, double theta, double kappa1, double kappa2, double beta, double volvol, double eta,
                           int is_spot_measure, double *out)
{
    const LogsvFast c = logsv_fast_in_log_units(make_logsv_fast(make_logsv_consts(dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure)));
    static_assert(sizeof(LogsvFast) == LOGSV_FAST_CONSTS_DOUBLES * sizeof(double), "LogsvFast layout");
    memcpy(out, &c, sizeof(c));
}

// The price -> implied-vol step of a calibration objective on the device, where the payoff sums already are: one lane per
// quote finalises its price (utils/mc_payoffs.py:85-88) and inverts Black-76 (svmc_black.h, the host routine's solver).
// quotes[k] = {strike, payoff code, shift, forward, ttm, discfactor}: fixed for a chain, uploaded once by the caller.
constexpr int IV_QUOTE_DOUBLES = 6;
__global__ __launch_bounds__(64) void chain_implied_vols_kernel(const double *__restrict__ sums, const double *__restrict__ quotes,
                                                               size_t n_quotes, double n_path_total, double vol_lo,
                                                               double vol_hi, double *__restrict__ ivols,
                                                               double *__restrict__ sums_copy)
{
    const size_t k = static_cast<size_t>(blockIdx.x) * 64 + threadIdx.x;
    if (k >= n_quotes) return;
    const double *qd = quotes + IV_QUOTE_DOUBLES * k;
    double price, se;
    const double s0 = sums[3 * k], s1 = sums[3 * k + 1], s2 = sums[3 * k + 2];
    if (sums_copy != nullptr) {         // the caller's pinned host buffer: the sums go home from here, not by a copy node
        sums_copy[3 * k] = s0;
        sums_copy[3 * k + 1] = s1;
        sums_copy[3 * k + 2] = s2;
    }
    payoff_finalize_one(s0, s1, s2, qd[2], qd[5], n_path_total, &price, &se);
    const int code = static_cast<int>(qd[1]);
    // inverse quotes (IC / IP): the Black-76 value of (S - K)^+ / S is the vanilla value over the forward
    const bool call = code == SVMC_CALL || code == SVMC_INV_CALL;
    ivols[k] = black_implied_vol(code >= SVMC_INV_CALL ? price * qd[3] : price, qd[0], call, qd[3], qd[4], qd[5], vol_lo, vol_hi);
}

int chain_implied_vols(const double *sums_dev, const double *quotes_dev, size_t n_quotes, double n_path_total, double vol_lo,
                       double vol_hi, double *ivols_out, double *sums_copy, hipStream_t stream)
{
    if (n_quotes == 0) return SVMC_OK;
    hipLaunchKernelGGL(chain_implied_vols_kernel, dim3(static_cast<unsigned>((n_quotes + 63) / 64)), dim3(64), 0, stream, sums_dev,
                       quotes_dev, n_quotes, n_path_total, vol_lo, vol_hi, ivols_out, sums_copy);
    return check_launch("chain_implied_vols");
}

void logsv_consts_to_doubles(double dt, double theta, double kappa1, double kappa2, double beta, double volvol, double eta,
                             int is_spot_measure, double *out)
{
    const LogsvConsts c = make_logsv_consts(dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure);
    static_assert(sizeof(LogsvConsts) == LOGSV_CONSTS_DOUBLES * sizeof(double), "LogsvConsts layout");
    memcpy(out, &c, sizeof(c));
}

}  // namespace svmc

extern "C" {

int svmc_logsv_vol_paths(double *sigma_t, size_t ld, size_t n_path, int nb_steps, double dt, double v0, double theta,
                         double kappa1, double kappa2, double beta, double volvol, int is_spot_measure,
                         const double *brownians, size_t ldb, uint64_t seed, uint32_t call_id, uint64_t path_offset,
                         svmc_stream_t stream)
{
    SVMC_REQUIRE(sigma_t != nullptr, "svmc_logsv_vol_paths: null output");
    SVMC_REQUIRE(nb_steps >= 0 && dt > 0.0 && ld >= n_path, "svmc_logsv_vol_paths: bad nb_steps/dt/ld");
    SVMC_REQUIRE(brownians == nullptr || ldb >= n_path, "svmc_logsv_vol_paths: ldb < n_path");
    SVMC_REQUIRE(call_id < (1u << 24), "svmc_logsv_vol_paths: call_id must fit 24 bits");
    if (n_path == 0) return SVMC_OK;
    const double adj = is_spot_measure ? 0.0 : beta;                                            // :930-933
    const double vartheta2 = beta * beta + volvol * volvol;
    const double K = LOG_UNITS_PER_NAT;
    VolPathConsts c;
    c.c1 = kappa1 * theta * dt * K;
    c.c2 = (adj - kappa2) * dt * K;
    c.c3 = (kappa2 * theta - kappa1 - 0.5 * vartheta2) * dt * K;
    c.cz = sqrt(vartheta2) * K * (brownians != nullptr ? 1.0 : sqrt(dt));
    c.L0 = log(v0) * K;
    if (brownians != nullptr)
        hipLaunchKernelGGL(logsv_vol_paths_kernel<false>, dim3(grid_for(n_path)), dim3(BLOCK), 0, as_stream(stream),
                           sigma_t, ld, n_path, nb_steps, v0, c, brownians, ldb, seed, make_c3(call_id), path_offset, armed_probe());
    else
        hipLaunchKernelGGL(logsv_vol_paths_kernel<true>, dim3(static_cast<unsigned>((n_path + VOLPATHS_RNG_BLOCK - 1) / VOLPATHS_RNG_BLOCK)),
                           dim3(VOLPATHS_RNG_BLOCK), 0, as_stream(stream), sigma_t, ld, n_path, nb_steps, v0, c, brownians, ldb,
                           seed, make_c3(call_id), path_offset, armed_probe());
    return check_launch("svmc_logsv_vol_paths");
}

int svmc_row_power_sums(const double *a, size_t ld, size_t n_rows, size_t n_cols, double center, int n_moments,
                        double *sums, void *workspace, size_t workspace_bytes, svmc_stream_t stream)
{
    SVMC_REQUIRE(a != nullptr && sums != nullptr && workspace != nullptr, "svmc_row_power_sums: null pointer");
    SVMC_REQUIRE(n_moments >= 1 && n_moments <= ROW_MOMENTS_MAX, "svmc_row_power_sums: 1 <= n_moments <= 4");
    SVMC_REQUIRE(ld >= n_cols && n_cols > 0, "svmc_row_power_sums: ld < n_cols or no columns");
    SVMC_REQUIRE(n_rows < (1u << 30), "svmc_row_power_sums: too many rows");
    if (n_rows == 0) return SVMC_OK;
    const size_t need = n_rows * ROW_SEGMENTS * 2 * static_cast<size_t>(n_moments) * sizeof(double);
    if (workspace_bytes < need) return fail(SVMC_ERR_WORKSPACE, "svmc_row_power_sums: workspace too small (rows x 4 x 2 n_moments doubles)");
    double *partials = static_cast<double *>(workspace);
    const dim3 grid(static_cast<unsigned>(n_rows), ROW_SEGMENTS);
    switch (n_moments) {
    case 1: hipLaunchKernelGGL(row_power_sums_kernel<1>, grid, dim3(BLOCK), 0, as_stream(stream), a, ld, n_cols, center, partials); break;
    case 2: hipLaunchKernelGGL(row_power_sums_kernel<2>, grid, dim3(BLOCK), 0, as_stream(stream), a, ld, n_cols, center, partials); break;
    case 3: hipLaunchKernelGGL(row_power_sums_kernel<3>, grid, dim3(BLOCK), 0, as_stream(stream), a, ld, n_cols, center, partials); break;
    default: hipLaunchKernelGGL(row_power_sums_kernel<4>, grid, dim3(BLOCK), 0, as_stream(stream), a, ld, n_cols, center, partials); break;
    }
    // sums[row][j] = the four segments' partials added in segment order: partials is [rows x 2K outputs][4 addends] read as a
    // matrix whose "rows" are the addends -- element (r, c) at partials[r * 2K + c * 4 * 2K] would interleave; the layout
    // [row][seg][j] makes output (row, j) the column sum over seg with row stride 2K and column base row * 4 * 2K + j
    const int K2 = 2 * n_moments;
    for (int j = 0; j < K2; ++j)
        hipLaunchKernelGGL(reduce_columns_kernel, dim3(static_cast<unsigned>(n_rows)), dim3(BLOCK), 0, as_stream(stream), partials + j,
                           static_cast<unsigned>(ROW_SEGMENTS), static_cast<size_t>(K2), static_cast<size_t>(ROW_SEGMENTS) * K2,
                           sums + static_cast<size_t>(j) * n_rows);
    return check_launch("svmc_row_power_sums");
}

int svmc_expanding_mean_squares(const double *a, size_t ld, size_t n_rows, size_t n_cols, double *out, size_t ldo,
                                svmc_stream_t stream)
{
    SVMC_REQUIRE(a != nullptr && out != nullptr, "svmc_expanding_mean_squares: null pointer");
    SVMC_REQUIRE(ld >= n_cols && ldo >= n_cols, "svmc_expanding_mean_squares: leading dimension < n_cols");
    if (n_rows == 0 || n_cols == 0) return SVMC_OK;
    // two columns per lane (16-byte accesses) where the layout allows it: even column count and leading dimensions, 16-byte bases
    const bool pair = (n_cols % 2 == 0) && (ld % 2 == 0) && (ldo % 2 == 0) && (reinterpret_cast<uintptr_t>(a) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(out) % 16 == 0);
    if (pair)
        hipLaunchKernelGGL(expanding_mean_sq_kernel<true>, dim3(grid_for(n_cols / 2)), dim3(BLOCK), 0, as_stream(stream), a, ld, n_rows,
                           n_cols, out, ldo);
    else
        hipLaunchKernelGGL(expanding_mean_sq_kernel<false>, dim3(grid_for(n_cols)), dim3(BLOCK), 0, as_stream(stream), a, ld, n_rows,
                           n_cols, out, ldo);
    return check_launch("svmc_expanding_mean_squares");
}

// the compiled few-waves forms of the Heston generators, per scheme (see LOGSV_LAT_VARIANTS: entry 0 the product form, the rest
// measurement alternatives behind SVMC_GEN_VARIANT).  The Euler step has no LDS read of its own, so one pair ahead hides the
// draw's round trip completely; QE (100 registers by itself) runs the same loop at the 128-register budget.
using HestonSliceKernel = void (*)(double *, double *, double *, size_t, int, HestonConsts, QeConsts, uint64_t, uint32_t, uint64_t, uint32_t,
                                   SliceOut, StateInit);
using HestonChainKernel = void (*)(double *, double *, double *, size_t, HestonChainSlices, uint64_t, uint32_t, uint64_t, uint32_t, double *,
                                   double *, double *, StateInit);
struct HestonLatVariant {
    HestonSliceKernel slice;
    HestonChainKernel chain;
    int block;
};
#define SVMC_HESTON_LAT(S, LOOP, WAVES, TB) {heston_rng_lat_kernel<S, LOOP, WAVES, TB>, heston_chain_rng_lat_kernel<S, LOOP, WAVES, TB>, TB}
constexpr int N_HESTON_LAT_VARIANTS = 3;
static const HestonLatVariant HESTON_LAT_VARIANTS[3][N_HESTON_LAT_VARIANTS] = {
    {   // Euler with the reference's floor
        SVMC_HESTON_LAT(SVMC_HESTON_EULER_FLOOR, GEN_LOOP_PAIR, 4, 256),
        SVMC_HESTON_LAT(SVMC_HESTON_EULER_FLOOR, GEN_LOOP_FEW, 2, 256),
        SVMC_HESTON_LAT(SVMC_HESTON_EULER_FLOOR, GEN_LOOP_AHEAD, 4, 256),
    },
    {   // QE
        SVMC_HESTON_LAT(SVMC_HESTON_QE, GEN_LOOP_PAIR, 3, 256),
        SVMC_HESTON_LAT(SVMC_HESTON_QE, GEN_LOOP_FEW, 2, 256),
        SVMC_HESTON_LAT(SVMC_HESTON_QE, GEN_LOOP_AHEAD, 3, 256),
    },
    {   // QE, quadratic branch only
        SVMC_HESTON_LAT(HESTON_QE_QUAD, GEN_LOOP_PAIR, 4, 256),
        SVMC_HESTON_LAT(HESTON_QE_QUAD, GEN_LOOP_FEW, 2, 256),
        SVMC_HESTON_LAT(HESTON_QE_QUAD, GEN_LOOP_AHEAD, 3, 256),
    }};
#undef SVMC_HESTON_LAT

static const HestonLatVariant *heston_lat_variant(int scheme, size_t n_path)
{
    const HestonLatVariant *row = HESTON_LAT_VARIANTS[scheme];       // kernel scheme id: 0, 1 or HESTON_QE_QUAD
    const int forced = forced_gen_variant();
    if (forced == -1) return nullptr;
    if (forced >= 0) return &row[forced < N_HESTON_LAT_VARIANTS ? forced : 0];
    return few_waves_launch(n_path) ? &row[0] : nullptr;
}

static int heston_rng_launch(const char *fn, double *x, double *var, double *qvar, size_t n_path, int nb_steps, double dt,
                             double theta, double kappa, double rho, double volvol, int scheme, uint64_t seed,
                             uint32_t call_id, uint64_t path_offset, uint32_t step_offset, const SliceOut &so,
                             svmc_stream_t stream, const StateInit &init = StateInit())
{
    if (int rc = check_state(fn, x, var, qvar, nb_steps, dt)) return rc;
    if (scheme != SVMC_HESTON_EULER_FLOOR && scheme != SVMC_HESTON_QE)
        return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": unknown scheme");
    if (call_id >= (1u << 24)) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": call_id must fit 24 bits");
    if (n_path == 0) return SVMC_OK;
    const HestonConsts c = make_heston_consts(dt, theta, kappa, rho, volvol);
    const QeConsts qc = make_qe_consts(dt, theta, kappa, rho, volvol);
    const int ks = heston_kernel_scheme(scheme, qc);
    if (const HestonLatVariant *v = heston_lat_variant(ks, n_path))
        hipLaunchKernelGGL(v->slice, dim3(lat_grid(n_path, v->block)), dim3(v->block), 0, as_stream(stream), x, var, qvar, n_path, nb_steps,
                           c, qc, seed, make_c3(call_id), path_offset, step_offset, so, init);
    else if (ks == HESTON_QE_QUAD)
        hipLaunchKernelGGL(heston_rng_kernel<HESTON_QE_QUAD>, dim3(rng_grid(n_path)), dim3(rng_block()), 0, as_stream(stream), x, var, qvar,
                           n_path, nb_steps, c, qc, seed, make_c3(call_id), path_offset, step_offset, so, init);
    else if (scheme == SVMC_HESTON_QE)
        hipLaunchKernelGGL(heston_rng_kernel<SVMC_HESTON_QE>, dim3(rng_grid(n_path)), dim3(rng_block()), 0, as_stream(stream), x, var, qvar,
                           n_path, nb_steps, c, qc, seed, make_c3(call_id), path_offset, step_offset, so, init);
    else
        hipLaunchKernelGGL(heston_rng_kernel<SVMC_HESTON_EULER_FLOOR>, dim3(rng_grid(n_path)), dim3(rng_block()), 0, as_stream(stream), x,
                           var, qvar, n_path, nb_steps, c, qc, seed, make_c3(call_id), path_offset, step_offset, so, init);
    return check_launch(fn);
}

int svmc_heston_terminal_rng(double *x, double *var, double *qvar, size_t n_path, int nb_steps, double dt,
                             double theta, double kappa, double rho, double volvol, int scheme, uint64_t seed,
                             uint32_t call_id, uint64_t path_offset, uint32_t step_offset, svmc_stream_t stream)
{
    const SliceOut none = {nullptr, nullptr, nullptr, 0.0};
    if (nb_steps == 0) return check_state("svmc_heston_terminal_rng", x, var, qvar, nb_steps, dt);
    return heston_rng_launch("svmc_heston_terminal_rng", x, var, qvar, n_path, nb_steps, dt, theta, kappa, rho, volvol,
                             scheme, seed, call_id, path_offset, step_offset, none, stream);
}

}  // extern "C"

static int heston_slice_rng_impl(const char *fn, const StateInit &init, double *x, double *var, double *qvar, size_t n_path, int nb_steps, double dt, double theta,
                          double kappa, double rho, double volvol, int scheme, uint64_t seed, uint32_t call_id,
                          uint64_t path_offset, uint32_t step_offset, double forward, double *x_snapshot,
                          double *qvar_snapshot, double *spot_sums, void *workspace, size_t workspace_bytes,
                          svmc_stream_t stream)
{
    if (int rc = check_slice_args(fn, n_path, x_snapshot, spot_sums, workspace, workspace_bytes, true)) return rc;
    SVMC_REQUIRE(n_path > 0 && nb_steps > 0, std::string(fn) + ": n_path and nb_steps must be positive");
    const SliceOut so = {x_snapshot, qvar_snapshot, static_cast<double *>(workspace), forward, wave_rows(n_path)};
    if (int rc = heston_rng_launch(fn, x, var, qvar, n_path, nb_steps, dt, theta, kappa, rho, volvol, scheme, seed,
                                   call_id, path_offset, step_offset, so, stream, init))
        return rc;
    if (spot_sums == nullptr) return SVMC_OK;
    return finish_slice_sums(fn, wave_rows(n_path), spot_sums, workspace, stream);
}

extern "C" {

int svmc_heston_slice_rng(double *x, double *var, double *qvar, size_t n_path, int nb_steps, double dt, double theta,
                          double kappa, double rho, double volvol, int scheme, uint64_t seed, uint32_t call_id,
                          uint64_t path_offset, uint32_t step_offset, double forward, double *x_snapshot,
                          double *qvar_snapshot, double *spot_sums, void *workspace, size_t workspace_bytes,
                          svmc_stream_t stream)
{
    SVMC_REQUIRE(spot_sums != nullptr, "svmc_heston_slice_rng: null spot_sums");
    return heston_slice_rng_impl("svmc_heston_slice_rng", StateInit(), x, var, qvar, n_path, nb_steps, dt, theta, kappa, rho,
                                 volvol, scheme, seed, call_id, path_offset, step_offset, forward, x_snapshot, qvar_snapshot,
                                 spot_sums, workspace, workspace_bytes, stream);
}

int svmc_heston_slice_rng_from(double x0, double var0, double qvar0, double *x, double *var, double *qvar, size_t n_path,
                               int nb_steps, double dt, double theta, double kappa, double rho, double volvol, int scheme,
                               uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset, double forward,
                               double *x_snapshot, double *qvar_snapshot, double *spot_sums, void *workspace,
                               size_t workspace_bytes, svmc_stream_t stream)
{
    SVMC_REQUIRE(spot_sums != nullptr, "svmc_heston_slice_rng_from: null spot_sums");
    const StateInit init = {1, x0, var0, qvar0};
    return heston_slice_rng_impl("svmc_heston_slice_rng_from", init, x, var, qvar, n_path, nb_steps, dt, theta, kappa, rho,
                                 volvol, scheme, seed, call_id, path_offset, step_offset, forward, x_snapshot, qvar_snapshot,
                                 spot_sums, workspace, workspace_bytes, stream);
}

}  // extern "C"

static int heston_chain_rng_impl(const char *fn, const StateInit &init, double *x, double *var, double *qvar, size_t n_path, int n_slices, const int *nb_steps_host,
                          const double *dts_host, const double *forwards_host, double theta, double kappa, double rho,
                          double volvol, int scheme, uint64_t seed, uint32_t call_id, uint64_t path_offset,
                          uint32_t step_offset, double *x_snapshots, double *qvar_snapshots, double *spot_sums,
                          void *workspace, size_t workspace_bytes, svmc_stream_t stream)
{
    SVMC_REQUIRE(x && var && qvar && x_snapshots && workspace, std::string(fn) + ": null pointer");
    SVMC_REQUIRE(nb_steps_host && dts_host && forwards_host && n_slices >= 1, std::string(fn) + ": null grids / no slices");
    SVMC_REQUIRE(spot_sums != nullptr || n_slices <= MAX_CHAIN_SLICES, std::string(fn) + ": unreduced partials need one launch");
    SVMC_REQUIRE(scheme == SVMC_HESTON_EULER_FLOOR || scheme == SVMC_HESTON_QE, std::string(fn) + ": unknown scheme");
    SVMC_REQUIRE(call_id < (1u << 24), std::string(fn) + ": call_id must fit 24 bits");
    SVMC_REQUIRE(n_path > 0, std::string(fn) + ": n_path must be positive");
    for (int i = 0; i < n_slices; ++i)
        SVMC_REQUIRE(nb_steps_host[i] > 0 && dts_host[i] > 0.0, std::string(fn) + ": nb_steps and dt must be positive");
    const unsigned g = rng_grid(n_path);
    for (int i0 = 0; i0 < n_slices; i0 += MAX_CHAIN_SLICES) {
        HestonChainSlices cs;
        cs.m = (n_slices - i0 < MAX_CHAIN_SLICES) ? (n_slices - i0) : MAX_CHAIN_SLICES;
        if (workspace_bytes < static_cast<size_t>(wave_rows(n_path)) * 2 * cs.m * sizeof(double))
            return fail(SVMC_ERR_WORKSPACE, std::string(fn) + ": workspace too small (svmc_slice_workspace_bytes)");
        int steps = 0;
        for (int i = 0; i < MAX_CHAIN_SLICES; ++i) {
            const int j = (i < cs.m) ? i0 + i : i0;
            cs.c[i] = make_heston_consts(dts_host[j], theta, kappa, rho, volvol);
            cs.qc[i] = make_qe_consts(dts_host[j], theta, kappa, rho, volvol);
            cs.forward[i] = forwards_host[j];
            cs.nb_steps[i] = (i < cs.m) ? nb_steps_host[j] : 0;
            steps += cs.nb_steps[i];
        }
        double *xs = x_snapshots + static_cast<size_t>(i0) * n_path;
        double *qs = qvar_snapshots ? qvar_snapshots + static_cast<size_t>(i0) * n_path : nullptr;
        const StateInit init_i = (i0 == 0) ? init : StateInit();
        double *const ws = static_cast<double *>(workspace);
        int ks = scheme;
        if (scheme == SVMC_HESTON_QE) {                    // the specialised kernel only if EVERY slice of the launch allows it
            ks = HESTON_QE_QUAD;
            for (int i = 0; i < cs.m; ++i)
                if (heston_kernel_scheme(scheme, cs.qc[i]) != HESTON_QE_QUAD) ks = scheme;
        }
        if (const HestonLatVariant *v = heston_lat_variant(ks, n_path))
            hipLaunchKernelGGL(v->chain, dim3(lat_grid(n_path, v->block)), dim3(v->block), 0, as_stream(stream), x, var, qvar, n_path, cs,
                               seed, make_c3(call_id), path_offset, step_offset, xs, qs, ws, init_i);
        else if (ks == HESTON_QE_QUAD)
            hipLaunchKernelGGL(heston_chain_rng_kernel<HESTON_QE_QUAD>, dim3(g), dim3(rng_block()), 0, as_stream(stream), x, var, qvar,
                               n_path, cs, seed, make_c3(call_id), path_offset, step_offset, xs, qs, ws, init_i);
        else if (scheme == SVMC_HESTON_QE)
            hipLaunchKernelGGL(heston_chain_rng_kernel<SVMC_HESTON_QE>, dim3(g), dim3(rng_block()), 0, as_stream(stream), x, var, qvar,
                               n_path, cs, seed, make_c3(call_id), path_offset, step_offset, xs, qs, ws, init_i);
        else
            hipLaunchKernelGGL(heston_chain_rng_kernel<SVMC_HESTON_EULER_FLOOR>, dim3(g), dim3(rng_block()), 0, as_stream(stream), x, var,
                               qvar, n_path, cs, seed, make_c3(call_id), path_offset, step_offset, xs, qs, ws, init_i);
        if (spot_sums != nullptr)
            hipLaunchKernelGGL(reduce_columns_kernel, dim3(2 * cs.m), dim3(BLOCK), 0, as_stream(stream),
                               static_cast<const double *>(workspace), wave_rows(n_path), size_t(1), static_cast<size_t>(wave_rows(n_path)), spot_sums + 2 * i0);
        if (int rc = check_launch(fn)) return rc;
        step_offset += static_cast<uint32_t>(steps);
    }
    return SVMC_OK;
}

extern "C" {

int svmc_heston_chain_rng(double *x, double *var, double *qvar, size_t n_path, int n_slices, const int *nb_steps_host,
                          const double *dts_host, const double *forwards_host, double theta, double kappa, double rho,
                          double volvol, int scheme, uint64_t seed, uint32_t call_id, uint64_t path_offset,
                          uint32_t step_offset, double *x_snapshots, double *qvar_snapshots, double *spot_sums,
                          void *workspace, size_t workspace_bytes, svmc_stream_t stream)
{
    SVMC_REQUIRE(spot_sums != nullptr, "svmc_heston_chain_rng: null spot_sums");
    return heston_chain_rng_impl("svmc_heston_chain_rng", StateInit(), x, var, qvar, n_path, n_slices, nb_steps_host, dts_host,
                                 forwards_host, theta, kappa, rho, volvol, scheme, seed, call_id, path_offset, step_offset,
                                 x_snapshots, qvar_snapshots, spot_sums, workspace, workspace_bytes, stream);
}

int svmc_heston_chain_rng_from(double x0, double var0, double qvar0, double *x, double *var, double *qvar, size_t n_path,
                               int n_slices, const int *nb_steps_host, const double *dts_host, const double *forwards_host,
                               double theta, double kappa, double rho, double volvol, int scheme, uint64_t seed,
                               uint32_t call_id, uint64_t path_offset, uint32_t step_offset, double *x_snapshots,
                               double *qvar_snapshots, double *spot_sums, void *workspace, size_t workspace_bytes,
                               svmc_stream_t stream)
{
    SVMC_REQUIRE(spot_sums != nullptr, "svmc_heston_chain_rng_from: null spot_sums");
    const StateInit init = {1, x0, var0, qvar0};
    return heston_chain_rng_impl("svmc_heston_chain_rng_from", init, x, var, qvar, n_path, n_slices, nb_steps_host, dts_host,
                                 forwards_host, theta, kappa, rho, volvol, scheme, seed, call_id, path_offset, step_offset,
                                 x_snapshots, qvar_snapshots, spot_sums, workspace, workspace_bytes, stream);
}

// launch of rough_logsv_expiries_kernel: n_expiries simulations from time 0 side by side (grid.y).  EVERY from-origin launch of
// the rough model goes through this kernel -- the chain (svmc_rough_logsv_chain) and a single expiry (svmc_rough_logsv_slice /
// _terminal with from_origin != 0: grid.y = 1) -- so that one launch per expiry and all expiries in one launch are the same
// code on the same constants: identical bits by construction, not by how two kernels happen to be contracted.
static void launch_rough_expiries(const RoughConsts &c_in, int n_factors, int n_expiries, const int *nb_steps_host,
                                  const double *hs_host, const double *forwards_host, double *log_s, double *vol, double *qvar,
                                  size_t n_path, const double *Z0, const double *Z1, size_t ldw, uint64_t seed, uint32_t call_id,
                                  uint64_t path_offset, uint32_t step_offset, int from_origin, double *x_snapshots,
                                  double *qvar_snapshots, double *partials, hipStream_t st)
{
    RoughConsts c = c_in;
    RoughExpiries e{};
    e.m = n_expiries;
    for (int i = 0; i < MAX_CHAIN_SLICES; ++i) {
        const int j = i < n_expiries ? i : 0;
        const double h = hs_host[j];
        e.h[i] = h;
        e.inv_h[i] = 1.0 / h;
        e.sqrt_h[i] = sqrt(h);
        e.ito[i] = -0.5 * c.volvol_w * c.volvol_w * h;
        e.volvol_w_sqrt_h[i] = c.volvol_w * e.sqrt_h[i];
        e.forward[i] = forwards_host ? forwards_host[j] : 0.0;
        e.nb_steps[i] = nb_steps_host[j];
    }
    const bool draw = Z0 == nullptr;                        // the drawing kernels run the generators' block size
    const dim3 g(draw ? rng_grid(n_path) : grid_for(n_path), static_cast<unsigned>(n_expiries)), b(draw ? rng_block() : BLOCK);
    const uint32_t c3 = make_c3(call_id) | 3u;              // stream tag 3: the rough model's normals
#define SVMC_ROUGH_CHAIN_LAUNCH(NF)                                                                                          \
    do {                                                                                                                     \
        if (!draw)                                                                                                           \
            hipLaunchKernelGGL((rough_logsv_expiries_kernel<NF, false>), g, b, 0, st, log_s, vol, qvar, n_path, c, e, Z0, Z1, \
                               ldw, seed, c3, path_offset, step_offset, from_origin, x_snapshots, qvar_snapshots, partials); \
        else                                                                                                                 \
            hipLaunchKernelGGL((rough_logsv_expiries_kernel<NF, true>), g, b, 0, st, log_s, vol, qvar, n_path, c, e, Z0, Z1,  \
                               ldw, seed, c3, path_offset, step_offset, from_origin, x_snapshots, qvar_snapshots, partials); \
    } while (0)
    if (n_factors == 1) SVMC_ROUGH_CHAIN_LAUNCH(1);
    else if (n_factors == 2) SVMC_ROUGH_CHAIN_LAUNCH(2);
    else SVMC_ROUGH_CHAIN_LAUNCH(3);
#undef SVMC_ROUGH_CHAIN_LAUNCH
}

// the step-independent constants of the rough model
static RoughConsts make_rough_consts(int n_factors, const double *nodes_host, const double *weights_host, const double *v0_host,
                                     double theta, double kappa1, double kappa2, double rho, double volvol)
{
    RoughConsts c{};
    c.wsum = 0.0;
    c.w_lam_v0 = 0.0;
    for (int i = 0; i < n_factors; ++i) {
        c.nodes[i] = nodes_host[i];
        c.w[i] = weights_host[i];
        c.v0[i] = v0_host[i];
        c.wlam[i] = weights_host[i] * nodes_host[i];
        c.wsum += weights_host[i];
        c.w_lam_v0 += c.wlam[i] * v0_host[i];
    }
    c.theta = theta; c.kappa1 = kappa1; c.kappa2 = kappa2; c.rho = rho; c.rho_comp = sqrt(1.0 - rho * rho);
    c.volvol = volvol; c.inv_volvol = 1.0 / volvol; c.w_inv = 1.0 / c.wsum;
    c.volvol_w = volvol * c.wsum;
    return c;
}

static int rough_logsv_launch(const char *name, double *log_s, double *vol, double *qvar, size_t n_path, int nb_steps,
                              double h, int n_factors, const double *nodes_host, const double *weights_host,
                              const double *v0_host, double theta, double kappa1, double kappa2, double rho,
                              double volvol, const double *Z0, const double *Z1, size_t ldw, uint64_t seed,
                              uint32_t call_id, uint64_t path_offset, uint32_t step_offset, int from_origin,
                              const SliceOut &so, svmc_stream_t stream)
{
    const std::string fn(name);
    if (int rc = check_state(name, log_s, vol, qvar, nb_steps, h)) return rc;
    SVMC_REQUIRE(n_factors >= 1 && n_factors <= 3, fn + ": 1 <= n_factors <= 3");
    SVMC_REQUIRE(nodes_host && weights_host && v0_host, fn + ": null nodes/weights/v0");
    SVMC_REQUIRE((Z0 == nullptr) == (Z1 == nullptr), fn + ": Z0 and Z1 go together");
    SVMC_REQUIRE(Z0 == nullptr || ldw >= n_path, fn + ": ldw < n_path");
    SVMC_REQUIRE(call_id < (1u << 24), fn + ": call_id must fit 24 bits");
    SVMC_REQUIRE(volvol > 0.0 && rho * rho <= 1.0, fn + ": volvol > 0 and |rho| <= 1");
    if (n_path == 0 || (nb_steps == 0 && !from_origin && so.partials == nullptr)) return SVMC_OK;
    const RoughConsts c = make_rough_consts(n_factors, nodes_host, weights_host, v0_host, theta, kappa1, kappa2, rho, volvol);
    launch_rough_expiries(c, n_factors, 1, &nb_steps, &h, &so.forward, log_s, vol, qvar, n_path, Z0, Z1, ldw, seed, call_id,
                          path_offset, step_offset, from_origin, so.x_snap, so.q_snap, so.partials, as_stream(stream));
    return check_launch(name);
}

int svmc_rough_logsv_terminal(double *log_s, double *vol, double *qvar, size_t n_path, int nb_steps, double h,
                              int n_factors, const double *nodes_host, const double *weights_host,
                              const double *v0_host, double theta, double kappa1, double kappa2, double rho,
                              double volvol, const double *Z0, const double *Z1, size_t ldw, uint64_t seed,
                              uint32_t call_id, uint64_t path_offset, uint32_t step_offset, int from_origin,
                              svmc_stream_t stream)
{
    const SliceOut none = {nullptr, nullptr, nullptr, 0.0};
    return rough_logsv_launch("svmc_rough_logsv_terminal", log_s, vol, qvar, n_path, nb_steps, h, n_factors, nodes_host,
                              weights_host, v0_host, theta, kappa1, kappa2, rho, volvol, Z0, Z1, ldw, seed, call_id,
                              path_offset, step_offset, from_origin, none, stream);
}

int svmc_rough_logsv_slice(double *log_s, double *vol, double *qvar, size_t n_path, int nb_steps, double h,
                           int n_factors, const double *nodes_host, const double *weights_host, const double *v0_host,
                           double theta, double kappa1, double kappa2, double rho, double volvol, const double *Z0,
                           const double *Z1, size_t ldw, uint64_t seed, uint32_t call_id, uint64_t path_offset,
                           uint32_t step_offset, int from_origin, double forward, double *x_snapshot,
                           double *qvar_snapshot, double *spot_sums, void *workspace, size_t workspace_bytes,
                           svmc_stream_t stream)
{
    const char *fn = "svmc_rough_logsv_slice";
    if (int rc = check_slice_args(fn, n_path, x_snapshot, spot_sums, workspace, workspace_bytes)) return rc;
    SVMC_REQUIRE(n_path > 0 && nb_steps > 0, "svmc_rough_logsv_slice: n_path and nb_steps must be positive");
    const SliceOut so = {x_snapshot, qvar_snapshot, static_cast<double *>(workspace), forward, wave_rows(n_path)};
    if (int rc = rough_logsv_launch(fn, log_s, vol, qvar, n_path, nb_steps, h, n_factors, nodes_host, weights_host,
                                    v0_host, theta, kappa1, kappa2, rho, volvol, Z0, Z1, ldw, seed, call_id,
                                    path_offset, step_offset, from_origin, so, stream))
        return rc;
    return finish_slice_sums(fn, wave_rows(n_path), spot_sums, workspace, stream);
}

int svmc_rough_logsv_chain(double *log_s, double *vol, double *qvar, size_t n_path, int n_expiries, const int *nb_steps_host,
                           const double *hs_host, const double *forwards_host, int n_factors, const double *nodes_host,
                           const double *weights_host, const double *v0_host, double theta, double kappa1, double kappa2,
                           double rho, double volvol, const double *Z0, const double *Z1, size_t ldw, uint64_t seed,
                           uint32_t call_id, uint64_t path_offset, double *x_snapshots, double *qvar_snapshots,
                           double *spot_sums, void *workspace, size_t workspace_bytes, svmc_stream_t stream)
{
    const std::string fn("svmc_rough_logsv_chain");
    SVMC_REQUIRE(log_s && vol && qvar && x_snapshots && spot_sums && workspace, fn + ": null state / snapshot / spot_sums / workspace");
    SVMC_REQUIRE(nb_steps_host && hs_host && forwards_host, fn + ": null step counts / steps / forwards");
    SVMC_REQUIRE(n_expiries >= 1 && n_expiries <= MAX_CHAIN_SLICES, fn + ": 1 <= n_expiries <= 16");
    SVMC_REQUIRE(n_factors >= 1 && n_factors <= 3, fn + ": 1 <= n_factors <= 3");
    SVMC_REQUIRE(nodes_host && weights_host && v0_host, fn + ": null nodes/weights/v0");
    SVMC_REQUIRE((Z0 == nullptr) == (Z1 == nullptr), fn + ": Z0 and Z1 go together");
    SVMC_REQUIRE(Z0 == nullptr || ldw >= n_path, fn + ": ldw < n_path");
    SVMC_REQUIRE(call_id < (1u << 24), fn + ": call_id must fit 24 bits");
    SVMC_REQUIRE(volvol > 0.0 && rho * rho <= 1.0, fn + ": volvol > 0 and |rho| <= 1");
    SVMC_REQUIRE(n_path > 0, fn + ": n_path must be positive");
    if (workspace_bytes < static_cast<size_t>(wave_rows(n_path)) * 2 * n_expiries * sizeof(double))
        return fail(SVMC_ERR_WORKSPACE, fn + ": workspace too small (svmc_slice_workspace_bytes)");
    for (int i = 0; i < n_expiries; ++i)
        SVMC_REQUIRE(hs_host[i] > 0.0 && nb_steps_host[i] > 0, fn + ": steps and step counts must be positive");
    const RoughConsts c = make_rough_consts(n_factors, nodes_host, weights_host, v0_host, theta, kappa1, kappa2, rho, volvol);
    const hipStream_t st = as_stream(stream);
    launch_rough_expiries(c, n_factors, n_expiries, nb_steps_host, hs_host, forwards_host, log_s, vol, qvar, n_path, Z0, Z1, ldw,
                          seed, call_id, path_offset, 0u, 1, x_snapshots, qvar_snapshots, static_cast<double *>(workspace), st);
    hipLaunchKernelGGL(reduce_columns_kernel, dim3(2 * n_expiries), dim3(BLOCK), 0, st, static_cast<const double *>(workspace),
                       wave_rows(n_path), size_t(1), static_cast<size_t>(wave_rows(n_path)), spot_sums);
    return check_launch("svmc_rough_logsv_chain");
}

int svmc_heston_terminal_w(double *x, double *var, double *qvar, size_t n_path, int nb_steps, double dt,
                           double theta, double kappa, double rho, double volvol, const double *W0,
                           const double *W1, size_t ldw, svmc_stream_t stream)
{
    if (int rc = check_state("svmc_heston_terminal_w", x, var, qvar, nb_steps, dt)) return rc;
    SVMC_REQUIRE(W0 && W1, "svmc_heston_terminal_w: null W0/W1");
    SVMC_REQUIRE(ldw >= n_path, "svmc_heston_terminal_w: ldw < n_path");
    if (n_path == 0 || nb_steps == 0) return SVMC_OK;
    const HestonConsts c = make_heston_consts(dt, theta, kappa, rho, volvol);
    hipLaunchKernelGGL(heston_w_kernel, dim3(grid_for(n_path)), dim3(BLOCK), 0, as_stream(stream), x, var, qvar,
                       n_path, nb_steps, c, W0, W1, ldw);
    return check_launch("svmc_heston_terminal_w");
}

int svmc_heston_qe_terminal_w(double *x, double *var, double *qvar, size_t n_path, int nb_steps, double dt,
                              double theta, double kappa, double rho, double volvol, const double *Z0,
                              const double *Z1, const double *U, size_t ldw, svmc_stream_t stream)
{
    if (int rc = check_state("svmc_heston_qe_terminal_w", x, var, qvar, nb_steps, dt)) return rc;
    SVMC_REQUIRE(Z0 && Z1 && U, "svmc_heston_qe_terminal_w: null Z0/Z1/U");
    SVMC_REQUIRE(ldw >= n_path, "svmc_heston_qe_terminal_w: ldw < n_path");
    if (n_path == 0 || nb_steps == 0) return SVMC_OK;
    const QeConsts qc = make_qe_consts(dt, theta, kappa, rho, volvol);
    hipLaunchKernelGGL(heston_qe_w_kernel, dim3(grid_for(n_path)), dim3(BLOCK), 0, as_stream(stream), x, var, qvar,
                       n_path, nb_steps, qc, Z0, Z1, U, ldw);
    return check_launch("svmc_heston_qe_terminal_w");
}

int svmc_clock_probe_arm(int enable)
{
    uint64_t *&p = armed_probe();
    if (enable && p == nullptr) {
        SVMC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), 8 * sizeof(uint64_t)));
        SVMC_HIP_TRY(hipMemset(p, 0, 8 * sizeof(uint64_t)));
    } else if (!enable && p != nullptr) {
        SVMC_HIP_TRY(hipDeviceSynchronize());               // no launch that still stamps it
        (void)hipFree(p);
        p = nullptr;
    }
    return SVMC_OK;
}

int svmc_clock_probe_read(uint64_t *stamps, svmc_stream_t stream)
{
    SVMC_REQUIRE(stamps != nullptr, "svmc_clock_probe_read: null output");
    SVMC_REQUIRE(armed_probe() != nullptr, "svmc_clock_probe_read: this thread has not armed the probe (svmc_clock_probe_arm)");
    SVMC_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    SVMC_HIP_TRY(hipMemcpy(stamps, armed_probe(), 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return SVMC_OK;
}

int svmc_payoff_workspace_bytes(size_t *bytes)
{
    SVMC_REQUIRE(bytes != nullptr, "svmc_payoff_workspace_bytes: null output");
    *bytes = static_cast<size_t>(MAX_REDUCE_GRID) * 3 * PAYOFF_KT * PAYOFF_GROUPS * sizeof(double);
    return SVMC_OK;
}

int svmc_slice_workspace_bytes(size_t n_path, size_t *bytes)
{
    SVMC_REQUIRE(bytes != nullptr, "svmc_slice_workspace_bytes: null output");
    const size_t fused = static_cast<size_t>(wave_rows(n_path)) * 2 * MAX_CHAIN_SLICES * sizeof(double);
    const size_t payoff = static_cast<size_t>(MAX_REDUCE_GRID) * 3 * PAYOFF_KT * PAYOFF_GROUPS * sizeof(double);
    *bytes = fused > payoff ? fused : payoff;
    return SVMC_OK;
}

int svmc_spot_sums(const double *x, size_t n_path, double forward, double *spot_sums, void *workspace,
                   size_t workspace_bytes, svmc_stream_t stream)
{
    SVMC_REQUIRE(x && spot_sums && workspace, "svmc_spot_sums: null pointer");
    const unsigned g = reduce_grid(n_path);
    if (workspace_bytes < static_cast<size_t>(g) * 2 * sizeof(double))
        return fail(SVMC_ERR_WORKSPACE, "svmc_spot_sums: workspace too small (svmc_payoff_workspace_bytes)");
    double *partials = static_cast<double *>(workspace);
    hipLaunchKernelGGL(spot_sums_kernel, dim3(g), dim3(BLOCK), 0, as_stream(stream), x, n_path, forward, partials);
    hipLaunchKernelGGL(reduce_columns_kernel, dim3(2), dim3(BLOCK), 0, as_stream(stream), partials, g,
                       size_t(2), size_t(1), spot_sums);
    return check_launch("svmc_spot_sums");
}

}  // extern "C"

// the launches of svmc_payoff_sums / svmc_payoff_sums_chain: groups of <= PAYOFF_KT strikes of one expiry, <= PAYOFF_GROUPS
// groups per launch, every launch followed by its column reduce
// one kernel per group width: a chain's 21 (or 13) strikes per expiry are worked on as 21, not as the next multiple of 8
using PayoffKernel = void (*)(PayoffGroupPack, size_t, double *, int, PayoffSetStrides);
template <bool HAS_INV, bool NEED_Q, int... KS>
static PayoffKernel payoff_kernel_for(int kt, std::integer_sequence<int, KS...>)
{
    static const PayoffKernel table[] = {payoff_group_kernel<KS + 1, HAS_INV, NEED_Q>...};
    return table[kt - 1];
}
constexpr int PAYOFF_KT_INV = 16;      // inverse chains: three accumulators per strike, 24 would leave one wave per SIMD

// path blocks per strike group of a payoff launch: about a thousand blocks per launch in all (512 are resident at the kernel's
// two waves per SIMD) -- more groups, fewer and longer-running blocks each, so that a block's set-up (its constants) and
// wind-down (the 48-value block reduction) are amortised over more paths (C3's 4 groups: 118 -> 108 us) -- and no block with
// fewer than PAYOFF_MIN_TRIPS paths per lane: at a calibration's 10^5 paths 256 blocks per group made 1.5 trips each and the
// launch was set-up and wind-down only (4 groups 9.1 us; x 7 parameter sets 39 us).  A function of the path count and the
// group count alone -- never of the number of parameter sets -- so that every route adds the same partial sums.
#ifndef SVMC_PAYOFF_MIN_TRIPS
#define SVMC_PAYOFF_MIN_TRIPS 4
#endif
static inline unsigned payoff_path_blocks(size_t n_path, int n_groups)
{
    const unsigned g = reduce_grid(n_path);
    unsigned gx = (PAYOFF_BLOCKS + static_cast<unsigned>(n_groups) - 1u) / static_cast<unsigned>(n_groups);
    gx = (gx < 128u) ? 128u : gx;
    gx = (gx > g) ? g : gx;
    const size_t cap = (n_path + static_cast<size_t>(BLOCK) * SVMC_PAYOFF_MIN_TRIPS - 1) / (static_cast<size_t>(BLOCK) * SVMC_PAYOFF_MIN_TRIPS);
    if (cap < gx) gx = static_cast<unsigned>(cap < 1 ? 1 : cap);
    return gx;
}

template <bool HAS_INV>
static void launch_payoff_groups(int kt, dim3 grid, hipStream_t st, const PayoffGroupPack &pack, size_t n, int variable_type,
                                 double *partials, int ld, const PayoffSetStrides &sets)
{
    constexpr auto widths = std::make_integer_sequence<int, HAS_INV ? PAYOFF_KT_INV : PAYOFF_KT>();
    const PayoffKernel kern = (variable_type == SVMC_Q_VAR) ? payoff_kernel_for<HAS_INV, true>(kt, widths)
                                                            : payoff_kernel_for<HAS_INV, false>(kt, widths);
    hipLaunchKernelGGL(kern, grid, dim3(BLOCK), 0, st, pack, n, partials, ld, sets);
}

// spot_partials != null (one-set launches): the expiries' spot sums are formed inside the payoff kernel from the generators' per-wave
// partial columns (expiry i's pair 2 spot_rows i doubles in) -- no reduce launch ahead of this one, and
// spot_sums (nullable then) is only written.  partials_out != null: NO column reduce either -- the chain must fit one launch
// (payoff_sets_fit) and the caller ends it with chain_finish (one wave per quote: the column sums, the prices' implied vols,
// everything stored where the host reads it).
struct PayoffPartialsOut {
    unsigned rows = 0;      // path blocks of the launch = rows of the partials
    int cols = 0;           // strike columns per set
};
static int payoff_sums_impl(const char *fn, const double *const *xs, const double *const *qs, size_t n_path,
                            const double *forwards, const double *ttms, const double *spot_sums, int n_expiries,
                            const double *strikes, const int8_t *types, const double *shifts, const size_t *offsets,
                            int variable_type, double *sums, void *workspace, size_t workspace_bytes, svmc_stream_t stream,
                            int n_sets = 1, const PayoffSetStrides &sets = PayoffSetStrides{0, 0, 0},
                            const double *spot_partials = nullptr, unsigned spot_rows = 0, PayoffPartialsOut *partials_out = nullptr)
{
    // n_sets > 1: xs / qs / spot_sums / sums are those of set 0 and the others lie `sets` (and 3 x total sums) further on; the
    // chain must then fit ONE launch (payoff_sets_fit), whose path blocks are those of a one-set launch -- the same partial
    // sums added in the same order, hence the bits of n_sets calls
    const size_t total = offsets[n_expiries];
    if (spot_partials != nullptr && n_sets != 1)
        return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": spot sums inside the payoff kernel are for one-set launches");
    bool any_inv = false;
    for (size_t k = 0; k < total; ++k) {
        if (types[k] < SVMC_CALL || types[k] > SVMC_INV_PUT)
            return fail(SVMC_ERR_UNKNOWN_PAYOFF, "unknown option payoff code");
        any_inv = any_inv || types[k] >= SVMC_INV_CALL;
    }
    // strikes per group: 24 for plain chains (two accumulators per strike); 16 when the chain holds inverse options
    // (three accumulators per strike: 24 of them would leave one wave per SIMD)
    const size_t KG = any_inv ? PAYOFF_KT_INV : PAYOFF_KT;
    const unsigned g = reduce_grid(n_path);
    if (workspace_bytes < static_cast<size_t>(g) * 3 * PAYOFF_KT * PAYOFF_GROUPS * sizeof(double))
        return fail(SVMC_ERR_WORKSPACE, std::string(fn) + ": workspace too small (svmc_payoff_workspace_bytes)");
    if (n_path == 0 || total == 0) return SVMC_OK;
    double *partials = static_cast<double *>(workspace);
    PayoffGroupPack pack;
    memset(&pack, 0, sizeof(pack));
    int n_groups = 0, cols = 0, kt = 0;         // groups, strike columns and widest group of the pending launch
    size_t first_strike = 0;                    // global index of the pending launch's first strike
    bool has_inv = false;
    auto flush = [&]() -> int {
        if (n_groups == 0) return SVMC_OK;
        const unsigned gx = payoff_path_blocks(n_path, n_groups);
        const dim3 grid(gx, static_cast<unsigned>(n_groups), static_cast<unsigned>(n_sets));
        if (n_sets > 1 && (first_strike != 0 || static_cast<size_t>(cols) != total ||
                           static_cast<size_t>(gx) * n_sets * 3 * cols * sizeof(double) > workspace_bytes))
            return fail(SVMC_ERR_WORKSPACE, std::string(fn) + ": the parameter sets do not fit one payoff launch");
        if (partials_out != nullptr && (first_strike != 0 || static_cast<size_t>(cols) != total))
            return fail(SVMC_ERR_WORKSPACE, std::string(fn) + ": the chain does not fit one payoff launch");
        if (has_inv)
            launch_payoff_groups<true>(kt, grid, as_stream(stream), pack, n_path, variable_type, partials, 3 * cols, sets);
        else
            launch_payoff_groups<false>(kt, grid, as_stream(stream), pack, n_path, variable_type, partials, 3 * cols, sets);
        if (partials_out != nullptr) {
            partials_out->rows = gx;
            partials_out->cols = cols;
        } else {
            hipLaunchKernelGGL(reduce_columns_kernel, dim3(3 * cols * n_sets), dim3(BLOCK), 0, as_stream(stream), partials,
                               static_cast<unsigned>(gx), static_cast<size_t>(3 * cols) * n_sets, size_t(1), sums + 3 * first_strike);
        }
        first_strike += static_cast<size_t>(cols);
        n_groups = cols = kt = 0;
        has_inv = false;
        return check_launch(fn);
    };
    for (int i = 0; i < n_expiries; ++i) {
        if (xs[i] == nullptr) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": null snapshot");
        for (size_t k0 = offsets[i]; k0 < offsets[i + 1]; k0 += KG) {
            PayoffGroup &d = pack.g[n_groups];
            d.x = xs[i];
            d.qvar = (variable_type == SVMC_Q_VAR) ? qs[i] : nullptr;
            d.spot_sums = spot_sums ? spot_sums + 2 * i : nullptr;
            d.spot_partials = spot_partials ? spot_partials + 2 * static_cast<size_t>(spot_rows) * i : nullptr;
            d.spot_rows = spot_rows;
            d.forward = forwards[i];
            d.ttm = ttms[i];
            const size_t left = offsets[i + 1] - k0;
            d.k = static_cast<int>(left < KG ? left : KG);
            d.col = cols;
            d.inv_mask = 0u;
            d.put_mask = 0u;
            for (int k = 0; k < PAYOFF_KT; ++k) {
                const bool on = k < d.k;
                const int ty = on ? types[k0 + k] : SVMC_CALL;
                const double sgn = (ty == SVMC_CALL || ty == SVMC_INV_CALL) ? 1.0 : -1.0;
                if (sgn < 0.0) d.put_mask |= 1u << k;
                d.c[k] = on ? -sgn * strikes[k0 + k] : -__builtin_huge_val();      // unused: pay = max(u - inf, 0) = 0
                d.shift[k] = (on && shifts != nullptr) ? shifts[k0 + k] : 0.0;
                if (on && (ty == SVMC_INV_CALL || ty == SVMC_INV_PUT)) d.inv_mask |= 1u << k;
            }
            has_inv = has_inv || d.inv_mask != 0u;
            cols += d.k;
            kt = (d.k > kt) ? d.k : kt;
            if (++n_groups == PAYOFF_GROUPS)
                if (int rc = flush()) return rc;
        }
    }
    return flush();
}

namespace svmc {

// whether payoff_sums_chain_sets can take this chain: all its strike groups in one launch, the sets' partials in the workspace
bool payoff_sets_fit(size_t n_path, int n_expiries, const size_t *offsets, const int8_t *types, int n_sets, size_t workspace_bytes)
{
    const size_t total = offsets[n_expiries];
    if (n_path == 0 || total == 0 || n_sets < 1 || n_sets > 65535) return false;
    bool any_inv = false;
    for (size_t k = 0; k < total; ++k) any_inv = any_inv || types[k] >= SVMC_INV_CALL;
    const size_t KG = any_inv ? PAYOFF_KT_INV : PAYOFF_KT;
    size_t n_groups = 0;
    for (int i = 0; i < n_expiries; ++i) n_groups += (offsets[i + 1] - offsets[i] + KG - 1) / KG;
    if (n_groups == 0 || n_groups > static_cast<size_t>(PAYOFF_GROUPS)) return false;
    const unsigned g = reduce_grid(n_path), gx = payoff_path_blocks(n_path, static_cast<int>(n_groups));
    return static_cast<size_t>(gx) * n_sets * 3 * total * sizeof(double) <= workspace_bytes &&
           static_cast<size_t>(g) * 3 * PAYOFF_KT * PAYOFF_GROUPS * sizeof(double) <= workspace_bytes;
}

// svmc_payoff_sums_chain for n_sets parameter sets at once: ONE payoff launch (blockIdx.z = set) and ONE column reduce
// instead of n_sets of each; sums[set][3 x total].  Bit-equal to n_sets calls of svmc_payoff_sums_chain.
int payoff_sums_chain_sets(const double *const *x_snapshots_host, const double *const *qvar_snapshots_host, size_t n_path,
                           const double *forwards_host, const double *ttms_host, const double *spot_sums, int n_expiries,
                           const double *strikes_host, const int8_t *types_host, const double *shifts_host,
                           const size_t *strike_offsets_host, int variable_type, double *sums, void *workspace,
                           size_t workspace_bytes, hipStream_t stream, int n_sets, size_t x_set_stride, size_t q_set_stride,
                           size_t spot_set_stride)
{
    return payoff_sums_impl("payoff_sums_chain_sets", x_snapshots_host, qvar_snapshots_host, n_path, forwards_host, ttms_host,
                            spot_sums, n_expiries, strikes_host, types_host, shifts_host, strike_offsets_host, variable_type, sums,
                            workspace, workspace_bytes, reinterpret_cast<svmc_stream_t>(stream), n_sets,
                            PayoffSetStrides{x_set_stride, q_set_stride, spot_set_stride});
}

// The tail of an on-device-RNG chain on ONE device (svmc_chain.hip, round 6): (1) the payoff launch -- with spot_partials, every
// block forms its expiry's spot sums itself from the generators' per-wave partial columns ([expiry][2][spot_rows]: one batched
// round trip of loads for up to 2048 rows), without, it reads spot_sums as ever; (2) chain_finish_kernel, a wave per quote: the
// three column sums of the quote in reduce_columns_kernel's order of additions (the same bits), stored at sums_out -- the
// caller's pinned host buffer: no copy node.  Measured (tools/r06/chain_call_breakdown.py, 4 x 13 chain): reduce 4.8 + payoff 5.0 +
// reduce 4.9 + copy 4.3 us at 2^16 paths became payoff 7.5 + finish 4.4.  What was measured and NOT kept: the spot sums inside
// the payoff kernel above 2048 rows (every block repeats two or three dependent round trips: payoff 10.2 -> 17.9 us at 2 x 10^5
// paths, more than the reduce launch it saves), the same tail for the calibration objective's parameter sets (payoff of seven
// sets 19 -> 31 us), and the implied vols inside the finish kernel (one lane of 52 waves on 13 CUs: 24.6 us against 5.0 + 12.2
// for reduce + chain_implied_vols_kernel, whose one wave inverts 64 quotes side by side) -- profiles/r06_frozen_trace.txt.
// The chain must fit one payoff launch (payoff_sets_fit).
__global__ __launch_bounds__(BLOCK) void chain_finish_kernel(const double *__restrict__ partials, unsigned n_rows, int cols,
                                                             double *__restrict__ sums_out)
{
    const size_t q = static_cast<size_t>(blockIdx.x) * (BLOCK / 64) + (threadIdx.x >> 6);        // this wave's quote
    if (q >= static_cast<size_t>(cols)) return;                                                 // (wave-uniform)
    const size_t ld = 3 * static_cast<size_t>(cols);         // partials[path block][3 x cols]: column 3 q + which
    // lane l plays the threads l, 64 + l, 128 + l, 192 + l of block_column_sum one after the other -- the same additions in the
    // same order, the same bits -- with the loads of all three columns (a payoff launch has at most 1024 path blocks: four rows
    // per virtual thread) in flight before the first addition: one round trip, not twelve
    const unsigned lane = threadIdx.x & 63u;
    double t[3][4][4], sm[3];
#pragma unroll
    for (int which = 0; which < 3; ++which)
#pragma unroll
        for (int vw = 0; vw < 4; ++vw) column_rows_load<4>(partials + 3 * q + which, vw * 64u + lane, n_rows, ld, t[which][vw]);
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        double total = 0.0;
#pragma unroll
        for (int vw = 0; vw < 4; ++vw) {
            const double w = wave_sum(column_rows_add<4>(t[which][vw], vw * 64u + lane, n_rows));
            total = (vw == 0) ? w : total + w;
        }
        sm[which] = total;
    }
    if (lane != 0u) return;
    sums_out[3 * q] = sm[0];
    sums_out[3 * q + 1] = sm[1];
    sums_out[3 * q + 2] = sm[2];
}

int chain_payoff_and_finish(const double *const *x_snapshots_host, const double *const *qvar_snapshots_host, size_t n_path,
                            const double *forwards_host, const double *ttms_host, double *spot_sums, const double *spot_partials,
                            int n_expiries, const double *strikes_host, const int8_t *types_host, const double *shifts_host,
                            const size_t *strike_offsets_host, int variable_type, void *workspace, size_t workspace_bytes,
                            hipStream_t stream, double *sums_out)
{
    const char *fn = "chain_payoff_and_finish";
    const unsigned rows = wave_rows(n_path);
    PayoffPartialsOut po;
    if (int rc = payoff_sums_impl(fn, x_snapshots_host, qvar_snapshots_host, n_path, forwards_host, ttms_host, spot_sums, n_expiries,
                                  strikes_host, types_host, shifts_host, strike_offsets_host, variable_type, nullptr, workspace,
                                  workspace_bytes, reinterpret_cast<svmc_stream_t>(stream), 1, PayoffSetStrides{0, 0, 0}, spot_partials,
                                  rows, &po))
        return rc;
    if (po.cols == 0) return SVMC_OK;
    if (po.rows > 4u * BLOCK) return fail(SVMC_ERR_WORKSPACE, std::string(fn) + ": more path blocks than chain_finish_kernel sums");
    hipLaunchKernelGGL(chain_finish_kernel, dim3(static_cast<unsigned>((po.cols + BLOCK / 64 - 1) / (BLOCK / 64))), dim3(BLOCK), 0, stream,
                       static_cast<const double *>(workspace), po.rows, po.cols, sums_out);
    return check_launch(fn);
}

// the stepping launch of a chain WITHOUT the reduce of its per-wave partial columns (they stay in `workspace`):
// svmc_*_slice_rng_from / svmc_*_chain_rng_from with spot_sums = null, for svmc_chain.hip's one-device tail
int logsv_step_partials(double sigma0, double *x, double *sigma, double *qvar, size_t n_path, int n_slices, const int *nb_steps_host,
                        const double *dts_host, const double *etas_host, const double *forwards_host, double theta, double kappa1,
                        double kappa2, double beta, double volvol, int is_spot_measure, uint64_t seed, uint32_t call_id,
                        uint64_t path_offset, double *x_snapshots, double *qvar_snapshots, void *workspace, size_t workspace_bytes,
                        hipStream_t stream)
{
    const StateInit init = {1, 0.0, sigma0, 0.0};
    const svmc_stream_t st = reinterpret_cast<svmc_stream_t>(stream);
    if (n_slices == 1)
        return logsv_slice_rng_impl("logsv_step_partials", init, x, sigma, qvar, n_path, nb_steps_host[0], dts_host[0], theta, kappa1,
                                    kappa2, beta, volvol, etas_host ? etas_host[0] : 1.0, is_spot_measure, seed, call_id, path_offset, 0,
                                    forwards_host[0], x_snapshots, qvar_snapshots, nullptr, workspace, workspace_bytes, st);
    return logsv_chain_rng_impl("logsv_step_partials", init, x, sigma, qvar, n_path, n_slices, nb_steps_host, dts_host, etas_host,
                                forwards_host, theta, kappa1, kappa2, beta, volvol, is_spot_measure, seed, call_id, path_offset, 0,
                                x_snapshots, qvar_snapshots, nullptr, workspace, workspace_bytes, st);
}

int heston_step_partials(double var0, double *x, double *var, double *qvar, size_t n_path, int n_slices, const int *nb_steps_host,
                         const double *dts_host, const double *forwards_host, double theta, double kappa, double rho, double volvol,
                         int scheme, uint64_t seed, uint32_t call_id, uint64_t path_offset, double *x_snapshots,
                         double *qvar_snapshots, void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    const StateInit init = {1, 0.0, var0, 0.0};
    const svmc_stream_t st = reinterpret_cast<svmc_stream_t>(stream);
    if (n_slices == 1)
        return heston_slice_rng_impl("heston_step_partials", init, x, var, qvar, n_path, nb_steps_host[0], dts_host[0], theta, kappa, rho,
                                     volvol, scheme, seed, call_id, path_offset, 0, forwards_host[0], x_snapshots, qvar_snapshots,
                                     nullptr, workspace, workspace_bytes, st);
    return heston_chain_rng_impl("heston_step_partials", init, x, var, qvar, n_path, n_slices, nb_steps_host, dts_host, forwards_host,
                                 theta, kappa, rho, volvol, scheme, seed, call_id, path_offset, 0, x_snapshots, qvar_snapshots, nullptr,
                                 workspace, workspace_bytes, st);
}

// the largest launch whose payoff blocks form their spot sums themselves: 2048 rows = 131072 paths, ONE batched round trip of
// loads per block (block_column_sum2); above, the reduce launch is cheaper than what every block would repeat (see above)
bool spot_sums_in_payoff_kernel(size_t n_path) { return wave_rows(n_path) <= 8u * BLOCK; }

// reduce the generators' per-wave partial columns [n_cols][rows] into spot_sums[n_cols] (what the stepping entry points do
// themselves unless told not to)
int reduce_spot_partials(const void *workspace, size_t n_path, int n_cols, double *spot_sums, hipStream_t stream)
{
    if (n_cols <= 0) return SVMC_OK;
    hipLaunchKernelGGL(reduce_columns_kernel, dim3(static_cast<unsigned>(n_cols)), dim3(BLOCK), 0, stream,
                       static_cast<const double *>(workspace), wave_rows(n_path), size_t(1), static_cast<size_t>(wave_rows(n_path)),
                       spot_sums);
    return check_launch("reduce_spot_partials");
}

}  // namespace svmc

extern "C" {

int svmc_payoff_sums(const double *x, const double *qvar, size_t n_path, double forward, double ttm,
                     const double *spot_sums, const double *strikes_host, const int8_t *types_host,
                     const double *shifts_host, size_t n_strikes, int variable_type, double *sums, void *workspace,
                     size_t workspace_bytes, svmc_stream_t stream)
{
    if (variable_type != SVMC_LOG_RETURN && variable_type != SVMC_Q_VAR)
        return fail(SVMC_ERR_UNSUPPORTED_VARIABLE, "svmc_payoff_sums: variable_type must be LOG_RETURN or Q_VAR");
    SVMC_REQUIRE(x && spot_sums && sums && workspace, "svmc_payoff_sums: null pointer");
    SVMC_REQUIRE(variable_type != SVMC_Q_VAR || qvar != nullptr, "svmc_payoff_sums: Q_VAR needs qvar");
    SVMC_REQUIRE(n_strikes == 0 || (strikes_host && types_host), "svmc_payoff_sums: null strikes/types");
    const size_t offsets[2] = {0, n_strikes};
    return payoff_sums_impl("svmc_payoff_sums", &x, &qvar, n_path, &forward, &ttm, spot_sums, 1, strikes_host, types_host,
                            shifts_host, offsets, variable_type, sums, workspace, workspace_bytes, stream);
}

int svmc_payoff_sums_chain(const double *const *x_snapshots_host, const double *const *qvar_snapshots_host, size_t n_path,
                           const double *forwards_host, const double *ttms_host, const double *spot_sums,
                           int n_expiries, const double *strikes_host, const int8_t *types_host,
                           const double *shifts_host, const size_t *strike_offsets_host, int variable_type,
                           double *sums, void *workspace, size_t workspace_bytes, svmc_stream_t stream)
{
    if (variable_type != SVMC_LOG_RETURN && variable_type != SVMC_Q_VAR)
        return fail(SVMC_ERR_UNSUPPORTED_VARIABLE, "svmc_payoff_sums_chain: variable_type must be LOG_RETURN or Q_VAR");
    SVMC_REQUIRE(x_snapshots_host && forwards_host && ttms_host && spot_sums && strike_offsets_host && sums && workspace,
                 "svmc_payoff_sums_chain: null pointer");
    SVMC_REQUIRE(n_expiries >= 1, "svmc_payoff_sums_chain: n_expiries < 1");
    SVMC_REQUIRE(variable_type != SVMC_Q_VAR || qvar_snapshots_host != nullptr, "svmc_payoff_sums_chain: Q_VAR needs qvar");
    const size_t total = strike_offsets_host[n_expiries];
    SVMC_REQUIRE(total == 0 || (strikes_host && types_host), "svmc_payoff_sums_chain: null strikes/types");
    return payoff_sums_impl("svmc_payoff_sums_chain", x_snapshots_host, qvar_snapshots_host, n_path, forwards_host, ttms_host,
                            spot_sums, n_expiries, strikes_host, types_host, shifts_host, strike_offsets_host, variable_type,
                            sums, workspace, workspace_bytes, stream);
}

int svmc_payoff_finalize(const double *sums_host, const double *shifts_host, size_t n_strikes, double discfactor,
                         double n_path_total, double *prices_host, double *stderrs_host)
{
    SVMC_REQUIRE(sums_host && prices_host && stderrs_host, "svmc_payoff_finalize: null pointer");
    for (size_t k = 0; k < n_strikes; ++k)
        payoff_finalize_one(sums_host[3 * k], sums_host[3 * k + 1], sums_host[3 * k + 2], shifts_host ? shifts_host[k] : 0.0,
                            discfactor, n_path_total, prices_host + k, stderrs_host + k);
    return SVMC_OK;
}

int svmc_payoff_finalize_chain(const double *sums_host, const double *shifts_host, const double *discfactors_host,
                               size_t n_strikes, double n_path_total, double *prices_host, double *stderrs_host)
{
    SVMC_REQUIRE(sums_host && discfactors_host && prices_host && stderrs_host, "svmc_payoff_finalize_chain: null pointer");
    for (size_t k = 0; k < n_strikes; ++k)
        payoff_finalize_one(sums_host[3 * k], sums_host[3 * k + 1], sums_host[3 * k + 2], shifts_host ? shifts_host[k] : 0.0,
                            discfactors_host[k], n_path_total, prices_host + k, stderrs_host + k);
    return SVMC_OK;
}

}  // extern "C"
