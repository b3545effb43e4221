// svmc_rng.h -- counter-based randoms for the gfx950 kernels.
//
// Philox4x32-7 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11: seven rounds is
// the smallest Crush-resistant count of the 4x32 variant, ten the paper's safety default) keyed by the seed and
// indexed by (global path id, call index, stream, call id): a lane owns a path, so it derives every increment it
// needs in registers, no generator state lives in memory and the result is independent of how paths are sharded over
// GPUs.  Replaces the reference's serial MT19937+polar draw of two [nb_steps, nb_path] arrays
// (pricers/logsv_pricer.py:1025-1026, pricers/heston_pricer.py:369-370).
//
// Stream definition, version 2 (DESIGN.md "RNG"; CPU twin: oracle/svmc_oracle.c svo_draw_normals).  One call yields
// four 32-bit words = the Box-Muller pairs of TWO consecutive time steps:
//   (r0, r1, r2, r3) = philox4x32_7(ctr = (path_lo, path_hi, step >> 1, stream | call_id << 8), key = seed)
//   (ra, rb) = (r0, r1) for an even chain-global step index, (r2, r3) for an odd one
//   u1 = (ra + 1/2) 2^-32                                   in (0,1), exact; R = sqrt(-ln u1) <= 4.78 (|z| <= 6.76)
//   t  = 2 pi (rb + 1/2) 2^-32                              the angle, uniform on the full circle
//   (w0, w1) = sqrt(-2 ln u1) (cos t, sin t) = R (sqrt2 cos t, sqrt2 sin t):  the Box-Muller pair
//   stream 1:  uniform = 52 bits of r1:r0 (one call per draw);  Heston QE: pairs from stream 4 (as stream 0), the
//   exponential branch's uniform (r[step & 3] + 1/2) 2^-32 of stream 5's call step >> 2, drawn lazily.
// Resolution: a pair carries 64 random bits (32 radius, 32 angle) where version 1 spent 128 -- the price of
// halving the generator's share of the VALU-issue-bound stepping loop.  The radius is capped at sqrt(33 ln 2) = 4.78,
// i.e. |z| <= 6.76: the truncated mass is 1.4e-11 per normal (about 30 draws in 2^41, none expected in one C2 call
// of 2^31 normals); lattice spacings are 2^-32 in u1 and 2 pi 2^-32 in the angle -- far below the 1e-4 relative
// Monte Carlo error of any chain priced here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svmc_log_table.h"
#include "svmc_math.h"

namespace svmc {

// The tables of the draw, constant memory -> LDS once per block: 8 KB for neg_log_tab() and 8 KB for
// cossin_circle_tab32() (blocks are 256 threads: two + one entries per thread); the stepping kernels that call exp_tab() stage its 2 KB
// with them behind the same barrier.
__constant__ LogTabEntry g_log_table[512] = {SVMC_LOG_TABLE_INIT};
__constant__ CircleTabEntry g_circle_table[256] = {SVMC_CIRCLE_TABLE_INIT};   // sqrt2 (cos, sin) and the midpoint of 256 intervals
__constant__ double g_exp_table[256] = {SVMC_EXP_TABLE_INIT};

struct RngTables {
    const LogTabEntry *log;
    const CircleTabEntry *circle;
};

struct RngTablesLds {
    LogTabEntry log[512];
    CircleTabEntry circle[256];
};

// the log table alone (the streamed Heston QE kernel: its martingale correction takes logs, it draws nothing)
__device__ __forceinline__ const LogTabEntry *stage_log_table(LogTabEntry (&lds)[512])
{
    for (unsigned i = threadIdx.x; i < 512u; i += blockDim.x) lds[i] = g_log_table[i];
    __syncthreads();
    return lds;
}

__device__ __forceinline__ RngTables stage_rng_tables(RngTablesLds &lds)
{
    for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) {
        lds.log[i] = g_log_table[i];
        lds.log[i + 256u] = g_log_table[i + 256u];
        lds.circle[i] = g_circle_table[i];
    }
    __syncthreads();
    return RngTables{lds.log, lds.circle};
}

__device__ __forceinline__ RngTables stage_tables(RngTablesLds &lds, double (&lds_exp)[256])
{
    for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) {
        lds.log[i] = g_log_table[i];
        lds.log[i + 256u] = g_log_table[i + 256u];
        lds.circle[i] = g_circle_table[i];
    }
    for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) lds_exp[i] = g_exp_table[i];
    __syncthreads();
    return RngTables{lds.log, lds.circle};
}

#ifndef SVMC_PHILOX_ROUNDS
#define SVMC_PHILOX_ROUNDS 7
#endif
constexpr int PHILOX_ROUNDS = SVMC_PHILOX_ROUNDS;

__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t (&r)[4])
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < PHILOX_ROUNDS; ++i) {
        // one v_mad_u64_u32 per 32x32->64 product, one v_bitop3_b32 (xor3) per mixed word
        const uint64_t p0 = static_cast<uint64_t>(M0) * c0;
        const uint64_t p1 = static_cast<uint64_t>(M1) * c2;
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(p0 >> 32), c3, k1, 0x96);
        c1 = static_cast<uint32_t>(p1);
        c3 = static_cast<uint32_t>(p0);
        c0 = n0;
        c2 = n2;
        k0 += W0;
        k1 += W1;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// The same function with the work that does not depend on the step taken out of the time loop.  The counter is
// (path_lo, path_hi, call index, c3): in round 1 the product M0 * path_lo, the word path_hi ^ k0 and the whole new third word
// hi(M0 path_lo) ^ c3 ^ k1 are per-lane constants, M1 * step is wave-uniform (scalar unit), and round 2's product
// M1 * (third word) is a per-lane constant again.  Left to the compiler the xor3's of rounds 1-2 each needed a
// v_mov to get a second scalar operand in; here round 1 is one v_xor and round 2 one v_xor + one xor3.
struct PhiloxLane {
    uint32_t a;        // path_hi ^ k0
    uint32_t b;        // hi(M0 path_lo) ^ c3 ^ k1            (round-1 third word)
    uint32_t lo0;      // lo(M0 path_lo)                      (round-1 fourth word)
    uint32_t hi_b;     // hi(M1 b)
    uint32_t lo_b;     // lo(M1 b)                            (round-2 second word)
    uint32_t k0, k1;   // round-2 keys
};

__device__ __forceinline__ PhiloxLane philox_prepare(uint64_t seed, uint32_t c3, uint64_t path)
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    const uint32_t k0 = static_cast<uint32_t>(seed), k1 = static_cast<uint32_t>(seed >> 32);
    const uint64_t p0 = static_cast<uint64_t>(M0) * static_cast<uint32_t>(path);
    PhiloxLane l;
    l.a = static_cast<uint32_t>(path >> 32) ^ k0;
    l.b = static_cast<uint32_t>(p0 >> 32) ^ c3 ^ k1;
    l.lo0 = static_cast<uint32_t>(p0);
    const uint64_t pb = static_cast<uint64_t>(M1) * l.b;
    l.hi_b = static_cast<uint32_t>(pb >> 32);
    l.lo_b = static_cast<uint32_t>(pb);
    l.k0 = k0 + W0;
    l.k1 = k1 + W1;
    // opaque to the optimiser: otherwise it re-splits a into (path_hi, k0) and pays the second xor inside the loop
    asm volatile("" : "+v"(l.a), "+v"(l.b), "+v"(l.lo0), "+v"(l.hi_b), "+v"(l.lo_b));
    return l;
}

__device__ __forceinline__ void philox_draw(const PhiloxLane &l, uint32_t step, uint32_t (&r)[4])
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    // round 1: c = (hi(M1 step) ^ a, lo(M1 step), b, lo0)
    const uint64_t p1 = static_cast<uint64_t>(M1) * step;                  // wave-uniform
    uint32_t c0 = static_cast<uint32_t>(p1 >> 32) ^ l.a;
    const uint32_t c1_r1 = static_cast<uint32_t>(p1);                      // wave-uniform
    // round 2
    const uint64_t p0 = static_cast<uint64_t>(M0) * c0;
    uint32_t k0 = l.k0, k1 = l.k1;
    // the round keys are re-derived from (k0, k1) with scalar adds on every call: hoisted out of the time loop (what the
    // compiler does by itself) they would occupy a dozen SGPRs of a budget the loop already fills, and the spills come
    // back as v_readlane_b32 -- VALU instructions on the one port this loop is bound by
    asm volatile("" : "+s"(k0), "+s"(k1));
    c0 = l.hi_b ^ __builtin_amdgcn_readfirstlane(c1_r1 ^ k0);              // (uniform ^ uniform) stays on the scalar unit
    uint32_t c2 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(p0 >> 32), l.lo0, k1, 0x96);
    uint32_t c1 = l.lo_b;
    uint32_t c3 = static_cast<uint32_t>(p0);
    k0 += W0;
    k1 += W1;
#pragma unroll
    for (int i = 2; i < PHILOX_ROUNDS; ++i) {
        const uint64_t q0 = static_cast<uint64_t>(M0) * c0;
        const uint64_t q1 = static_cast<uint64_t>(M1) * c2;
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(q1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(q0 >> 32), c3, k1, 0x96);
        c1 = static_cast<uint32_t>(q1);
        c3 = static_cast<uint32_t>(q0);
        c0 = n0;
        c2 = n2;
        k0 += W0;
        k1 += W1;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// top 52 bits of hi:lo as the mantissa of a double in [1,2): two v_alignbit_b32
__device__ __forceinline__ double mantissa_1_2(uint32_t lo, uint32_t hi)
{
    const uint32_t mhi = __builtin_amdgcn_alignbit(0x3FFu, hi, 12);   // 0x3FF00000 | hi >> 12
    const uint32_t mlo = __builtin_amdgcn_alignbit(hi, lo, 12);       // (hi:lo) >> 12
    return __hiloint2double(static_cast<int>(mhi), static_cast<int>(mlo));
}

__device__ __forceinline__ void philox_draw(uint64_t seed, uint32_t c3, uint64_t path, uint32_t index, uint32_t (&r)[4])
{
    philox4x32(static_cast<uint32_t>(path), static_cast<uint32_t>(path >> 32), index, c3,
               static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), r);
}

// (k + 1/2) 2^-32 for a 32-bit word k: exact in fp64, in (0, 1)
__device__ __forceinline__ double uniform_32(uint32_t k)
{
    return fma(static_cast<double>(k), 0x1.0p-32, 0x1.0p-33);
}

// One Box-Muller pair from two words: radius from ra, direction from rb.  `shift1` is added to
// the second normal inside its final FMA (a model whose update has a constant term beside a multiple of z1 folds the
// constant in here for free: LogSV's per-step drift constant); 0.0 gives the plain pair.
__device__ __forceinline__ void normals_from_words(uint32_t ra, uint32_t rb, const RngTables &t, double shift1,
                                                   double &w0, double &w1)
{
    // u1 = (ra + 1/2) 2^-32: the half is an inline constant of the add and the 2^-32 an exponent offset of the logarithm
    const double R = sqrt_pos_1g(neg_log_tab<-32>(static_cast<double>(ra) + 0.5, t.log));  // sqrt(-ln u1): the sqrt2 lives in (a, b)
    double a, b;
    cossin_circle_tab32(rb, t.circle, a, b);
    w0 = R * a;
    w1 = fma(R, b, shift1);
}

// The time loop of every on-device-RNG generator: time steps [0, nb) of a lane whose first step has the chain-global
// index step0.  One Philox call serves the two steps 2c, 2c + 1, so the loop runs over CALLS and each half is guarded by
// a wave-uniform (scalar) test -- a slice that starts or ends on an odd step simply uses one half of its edge call,
// and slicing a chain differently never changes which randoms a (path, step) sees.
// step(z0, z1) advances the model by one time step on UNSCALED N(0,1); tick(t) runs once per call, before it, with
// the local index of the call's first step in this range (the progress priorities).
template <class Step, class Tick>
__device__ __forceinline__ void rng_time_loop(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab,
                                              double shift1, Step &&step, Tick &&tick)
{
    if (nb <= 0) return;
    const uint32_t first = step0, last = step0 + static_cast<uint32_t>(nb) - 1u;
    for (uint32_t c = first >> 1; c <= (last >> 1); ++c) {
        tick(static_cast<int>(2u * c - first));
        uint32_t r[4];
        philox_draw(lane, c, r);
        double z0, z1;
        if (2u * c >= first) {
            normals_from_words(r[0], r[1], tab, shift1, z0, z1);
            step(z0, z1);
        }
        if (2u * c + 1u <= last) {
            normals_from_words(r[2], r[3], tab, shift1, z0, z1);
            step(z0, z1);
        }
    }
}

template <class Step>
__device__ __forceinline__ void rng_time_loop(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab,
                                              Step &&step)
{
    rng_time_loop(lane, step0, nb, tab, 0.0, step, [](int) {});
}

// stream 0 for a single (path, step), from scratch: Box-Muller pair of UNSCALED N(0,1)  (svmc_fill_normals)
__device__ __forceinline__ void draw_normals(uint64_t seed, uint32_t c3, uint64_t path, uint32_t step,
                                             const RngTables &t, double &w0, double &w1)
{
    uint32_t r[4];
    philox_draw(seed, c3, path, step >> 1, r);
    const bool odd = (step & 1u) != 0u;
    normals_from_words(odd ? r[2] : r[0], odd ? r[3] : r[1], t, 0.0, w0, w1);
}

// Heston QE (streams 4 and 5).  The scheme needs a pair every step (z0 for the log-price, z1 for the quadratic branch)
// and a uniform only in the exponential branch: the pairs come from stream 4 exactly like stream 0's (one call per two
// steps, rng_time_loop), the uniforms from stream 5 -- word step & 3 of call step >> 2, i.e. one call per FOUR steps --
// drawn LAZILY: only when some lane of the wave is in the exponential branch at that step (a wave-uniform decision, so
// every lane of the wave takes part in the call and keeps its four words for the rest of the group).  A parameter set
// that never leaves the quadratic branch (Feller-satisfying sets on any sane grid) pays 12.5 Philox instructions per
// step instead of 25; one that always does pays 18.75.
struct QeUniforms {
    uint32_t r[4];
    uint32_t group = 0xFFFFFFFFu;      // the call (step >> 2) the words belong to: wave-uniform
};

__device__ __forceinline__ double qe_uniform(const PhiloxLane &lane_u, uint32_t step, QeUniforms &cache)
{
    const uint32_t g = step >> 2;
    if (cache.group != g) {
        philox_draw(lane_u, g, cache.r);
        cache.group = g;
    }
    const uint32_t k = step & 3u;                                     // wave-uniform selects
    const uint32_t w = (k == 0u) ? cache.r[0] : (k == 1u) ? cache.r[1] : (k == 2u) ? cache.r[2] : cache.r[3];
    return uniform_32(w);
}

// stream 1: one uniform in (0,1) with 52 random bits
__device__ __forceinline__ double draw_uniform(uint64_t seed, uint32_t c3, uint64_t path, uint32_t step)
{
    uint32_t r[4];
    philox_draw(seed, c3 | 1u, path, step, r);
    return mantissa_1_2(r[0], r[1]) - (1.0 - 0x1.0p-53);
}

}  // namespace svmc
