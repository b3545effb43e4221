// svmc_rng.h -- counter-based randoms for the gfx950 kernels.
//
// Philox4x32-7 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11: seven rounds is
// the smallest Crush-resistant count of the 4x32 variant, ten the paper's safety default) keyed by the seed and
// indexed by (global path id, call index, stream, call id): a lane owns a path, so it derives every increment it
// needs in registers, no generator state lives in memory and the result is independent of how paths are sharded over
// GPUs.  Replaces the reference's serial MT19937+polar draw of two [nb_steps, nb_path] arrays
// (pricers/logsv_pricer.py:1025-1026, pricers/heston_pricer.py:369-370).
//
// Stream definition, version 4 (DESIGN.md "RNG"; CPU twin: oracle/svmc_oracle.c svo_draw_normals).  One call yields
// four 32-bit words = the two normals of TWO consecutive time steps, each word turned into ONE N(0,1) variate by
// inversion:
//   (r0, r1, r2, r3) = philox4x32_7(ctr = (path_lo, path_hi, step >> 1, stream | call_id << 8), key = seed)
//   (ra, rb) = (r0, r1) for an even chain-global step index, (r2, r3) for an odd one
//   z(r) = sign(t) P_j(|t| - c_j),  t = (int32) r  (version 3: + 1/2):  the piecewise cubic of -Phi^-1(|t| 2^-32) of
//          svmc_icdf_table.h (tools/gen_icdf_table.py; svmc_math.h normal_icdf32), |z| <= 6.23, z = 0 at t = 0 and |t| = 2^31
//   (w0, w1) = (z(ra), z(rb))
//   stream 1:  uniform = 52 bits of r1:r0 (one call per draw);  Heston QE: normals from stream 4 (as stream 0), the
//   exponential branch's uniform (r[step & 3] + 1/2) 2^-32 of stream 5's call step >> 2, drawn lazily.
// Resolution: a normal carries 32 random bits (as in version 2, whose Box-Muller pair spent 32 on the radius and 32 on
// the angle); the lattice is 2^-32 in probability, symmetric about 0 (every magnitude 1 .. 2^31 - 1 with both signs, two words
// at 0), largest |z| = -Phi^-1(2^-32) = 6.23 (truncated mass 4.7e-10 per normal).  The cubic deviates from the exact inverse CDF by at most SVMC_ICDF_MAX_ABS_ERROR (7.431e-10,
// svmc_icdf_table.h) -- a smooth deterministic distortion five orders below the Monte Carlo error of any chain priced here, pinned against
// scipy's Phi^-1 in tests/test_oracle_golden.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef SVMC_ICDF_TABLE_HEADER          // A/B hook: another (M, degree) of tools/gen_icdf_table.py
#include SVMC_ICDF_TABLE_HEADER
#else
#include "svmc_icdf_table.h"
#endif
#ifndef SVMC_ICDF_RAW
#define SVMC_ICDF_RAW 0
#endif
#include "svmc_log_table.h"
#include "svmc_math.h"

namespace svmc {

// The tables, constant memory -> LDS once per block: the inverse-CDF pieces of the draw (SVMC_ICDF_SEGMENTS x 48 bytes,
// three arrays of 16-byte pieces), the 2 KB exp table of the kernels that call exp2u_tab() / exp_tab(), and -- Heston QE
// only -- the 8 KB log table of its martingale correction.
#if SVMC_ICDF_EDGE && SVMC_ICDF_DEG == 3
#define SVMC_ICDF_PIECES 2
__constant__ IcdfPiece g_icdf_table[2 * SVMC_ICDF_SEGMENTS] = {SVMC_ICDF_PIECE0_INIT, SVMC_ICDF_PIECE1_INIT};
#else
#define SVMC_ICDF_PIECES 3
__constant__ IcdfPiece g_icdf_table[3 * SVMC_ICDF_SEGMENTS] = {SVMC_ICDF_PIECE0_INIT, SVMC_ICDF_PIECE1_INIT, SVMC_ICDF_PIECE2_INIT};
#endif
__constant__ LogTabEntry g_log_table[512] = {SVMC_LOG_TABLE_INIT};
__constant__ double g_exp_table[256] = {SVMC_EXP_TABLE_INIT};

struct RngTables {
    const IcdfPiece *icdf;
    const LogTabEntry *log;        // null unless the kernel staged it (stage_rng_log_tables)
};

constexpr unsigned ICDF_PIECES = SVMC_ICDF_PIECES;
// SVMC_ICDF_MIXED (measurement builds only, round 6, VERDICT r05 item 2: "one LDS read per normal"): the EDGE form of the table
// with its two high-order coefficients kept as fp32 in LDS -- 24 bytes per normal (one ds_read_b128 + one ds_read_b64) instead of
// 32 (two ds_read_b128); the cubic then runs a3 d + a2 in fp32 and the two low-order steps in fp64 (12 VALU instructions per
// normal against the product's 8).  Another stream (other bits): never the product.  tools/r06/ab_icdf.sh builds and times it.
#ifndef SVMC_ICDF_MIXED
#define SVMC_ICDF_MIXED 0
#endif
struct RngTablesLds {
#if SVMC_ICDF_MIXED
    IcdfPiece icdf[SVMC_ICDF_SEGMENTS];
    float2 hi[SVMC_ICDF_SEGMENTS];
#else
    IcdfPiece icdf[ICDF_PIECES * SVMC_ICDF_SEGMENTS];
#endif
};

// the log table alone (the streamed Heston QE kernel: its martingale correction takes logs, it draws nothing)
__device__ __forceinline__ const LogTabEntry *stage_log_table(LogTabEntry (&lds)[512])
{
    for (unsigned i = threadIdx.x; i < 512u; i += blockDim.x) lds[i] = g_log_table[i];
    __syncthreads();
    return lds;
}

__device__ __forceinline__ void copy_icdf_table(RngTablesLds &lds)
{
#if SVMC_ICDF_MIXED
    static_assert(SVMC_ICDF_EDGE && !SVMC_ICDF_RAW && SVMC_ICDF_DEG == 3, "the mixed-precision table is the edge form of a cubic");
    for (unsigned i = threadIdx.x; i < SVMC_ICDF_SEGMENTS; i += blockDim.x) {
        lds.icdf[i] = g_icdf_table[i];
        const IcdfPiece h = g_icdf_table[SVMC_ICDF_SEGMENTS + i];
        lds.hi[i] = make_float2(static_cast<float>(h.a), static_cast<float>(h.b));
    }
#else
    for (unsigned i = threadIdx.x; i < ICDF_PIECES * SVMC_ICDF_SEGMENTS; i += blockDim.x) lds.icdf[i] = g_icdf_table[i];
#endif
}

__device__ __forceinline__ RngTables stage_rng_tables(RngTablesLds &lds)
{
    copy_icdf_table(lds);
    __syncthreads();
    return RngTables{lds.icdf, nullptr};
}

// Heston QE: the draw's table and the log table behind one barrier
__device__ __forceinline__ RngTables stage_rng_log_tables(RngTablesLds &lds, LogTabEntry (&lds_log)[512])
{
    copy_icdf_table(lds);
    for (unsigned i = threadIdx.x; i < 512u; i += blockDim.x) lds_log[i] = g_log_table[i];
    __syncthreads();
    return RngTables{lds.icdf, lds_log};
}

// a kernel template that draws in one instantiation and reads supplied normals in the other: the table's LDS only in the first
template <bool DRAWS>
struct RngTablesLdsIf {
    RngTablesLds t;
};
template <>
struct RngTablesLdsIf<false> {
    char unused;
};

template <bool DRAWS>
__device__ __forceinline__ RngTables stage_tables_if(RngTablesLdsIf<DRAWS> &lds, double (&lds_exp)[256])
{
    if constexpr (DRAWS) copy_icdf_table(lds.t);
    for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) lds_exp[i] = g_exp_table[i];
    __syncthreads();
    if constexpr (DRAWS) return RngTables{lds.t.icdf, nullptr};
    else return RngTables{nullptr, nullptr};
}

__device__ __forceinline__ RngTables stage_tables(RngTablesLds &lds, double (&lds_exp)[256])
{
    copy_icdf_table(lds);
    for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) lds_exp[i] = g_exp_table[i];
    __syncthreads();
    return RngTables{lds.icdf, nullptr};
}

#ifndef SVMC_TRIP_BARRIER
#define SVMC_TRIP_BARRIER 0
#endif
#ifndef SVMC_PHILOX_FENCE
#define SVMC_PHILOX_FENCE 0
#endif
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

template <bool HOIST_KEYS = false>
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
    // (HOIST_KEYS: the few-waves kernels, whose loop leaves two dozen SGPRs free -- there a wave issues one instruction of ANY
    // kind per ~4.5 cycles, so the twelve scalar adds of a call are twelve issue slots of the one wave a SIMD has)
    if constexpr (!HOIST_KEYS) asm volatile("" : "+s"(k0), "+s"(k1));
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

// The two normals of a time step from two words, each by inversion (svmc_math.h normal_icdf32)
#if SVMC_ICDF_MIXED
__device__ __forceinline__ double normal_icdf32_mixed(uint32_t w, const IcdfPiece *tab)
{
    const double t = icdf_lattice_point(w);
    const uint32_t hi = double_hi(t);
    const uint32_t off = (hi >> (16 - SVMC_ICDF_M)) & ((static_cast<uint32_t>(SVMC_ICDF_SEGMENTS) - 1u) << 4);
    const char *base = reinterpret_cast<const char *>(tab);
    const IcdfPiece e0 = *reinterpret_cast<const IcdfPiece *>(base + off);
    const float2 e1 = *reinterpret_cast<const float2 *>(base + 16 * SVMC_ICDF_SEGMENTS + (off >> 1));
    const double edge = bits_to_double(0u, hi & (0x7FFFFFFFu & ~((1u << (20 - SVMC_ICDF_M)) - 1u)));
    const double d = fabs(t) - edge;
    const float q = fmaf(e1.y, static_cast<float>(d), e1.x);
    double p = fma(static_cast<double>(q), d, e0.b);
    p = fma(p, d, e0.a);
    return copysign(p, t);
}
#endif

__device__ __forceinline__ void normals_from_words(uint32_t ra, uint32_t rb, const RngTables &t, double &w0, double &w1)
{
#if SVMC_ICDF_MIXED
    w0 = normal_icdf32_mixed(ra, t.icdf);
    w1 = normal_icdf32_mixed(rb, t.icdf);
    return;
#endif
    w0 = normal_icdf32<SVMC_ICDF_M, SVMC_ICDF_SEGMENTS, SVMC_ICDF_DEG, SVMC_ICDF_EDGE != 0, SVMC_ICDF_RAW != 0>(ra, t.icdf);
    w1 = normal_icdf32<SVMC_ICDF_M, SVMC_ICDF_SEGMENTS, SVMC_ICDF_DEG, SVMC_ICDF_EDGE != 0, SVMC_ICDF_RAW != 0>(rb, t.icdf);
}

// The four normals of one Philox call in two halves, for launches of one or two waves per SIMD where an LDS round trip is not
// hidden by other waves: draw_issue() converts the words and puts all eight table reads in flight (nothing is scheduled
// across its end), draw_finish() runs the four cubics.  normal_icdf32's operations on the same operands: the same bits.
struct DrawInFlight {
    double t[4];
    IcdfPiece e0[4], e1[4];
};
#if SVMC_ICDF_MIXED          // (the measurement build times the full-launch kernels only: the split forms just have to compile)
__device__ __forceinline__ void draw_issue(const uint32_t (&r)[4], const RngTables &tab, DrawInFlight &d)
{
    for (int k = 0; k < 4; ++k) d.t[k] = normal_icdf32_mixed(r[k], tab.icdf);
}
__device__ __forceinline__ void draw_finish(const DrawInFlight &d, double (&z)[4])
{
    for (int k = 0; k < 4; ++k) z[k] = d.t[k];
}
#else
__device__ __forceinline__ void draw_issue(const uint32_t (&r)[4], const RngTables &tab, DrawInFlight &d)
{
    static_assert(SVMC_ICDF_RAW != 0, "the split draw is written for the raw form of the table");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        d.t[k] = icdf_lattice_point(r[k]);
        const uint32_t off = (double_hi(d.t[k]) >> (16 - SVMC_ICDF_M)) & ((static_cast<uint32_t>(SVMC_ICDF_SEGMENTS) - 1u) << 4);
        const char *base = reinterpret_cast<const char *>(tab.icdf) + off;
        d.e0[k] = *reinterpret_cast<const IcdfPiece *>(base);
        d.e1[k] = *reinterpret_cast<const IcdfPiece *>(base + 16 * SVMC_ICDF_SEGMENTS);
    }
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void draw_finish(const DrawInFlight &d, double (&z)[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double a = fabs(d.t[k]);
        double p = fma(d.e1[k].b, a, d.e1[k].a);
        p = fma(p, a, d.e0[k].b);
        p = fma(p, a, d.e0[k].a);
        z[k] = copysign(p, d.t[k]);
    }
}
#endif

// The time loop of every on-device-RNG generator: time steps [0, nb) of a lane whose first step has the chain-global
// index step0.  One Philox call serves the two steps 2c, 2c + 1, so the loop runs over CALLS and each half is guarded by
// a wave-uniform (scalar) test -- a slice that starts or ends on an odd step simply uses one half of its edge call,
// and slicing a chain differently never changes which randoms a (path, step) sees.
// step(z0, z1) advances the model by one time step on UNSCALED N(0,1); tick(t) runs once per call, before it, with
// the local index of the call's first step in this range (the progress priorities).
template <class Step, class Tick>
__device__ __forceinline__ void rng_time_loop(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab,
                                              Step &&step, Tick &&tick)
{
    if (nb <= 0) return;
    const uint32_t first = step0, last = step0 + static_cast<uint32_t>(nb) - 1u;
    for (uint32_t c = first >> 1; c <= (last >> 1); ++c) {
        tick(static_cast<int>(2u * c - first));
#if SVMC_TRIP_BARRIER          // A/B hook (tools/r05/ab_pairing.sh): the waves of a block re-align every SVMC_TRIP_BARRIER-th trip
        if ((c % SVMC_TRIP_BARRIER) == 0u) __builtin_amdgcn_s_barrier();
#endif
        uint32_t r[4];
        philox_draw(lane, c, r);
#if SVMC_PHILOX_FENCE          // A/B hook: nothing is scheduled across the end of the Philox rounds
        __builtin_amdgcn_sched_barrier(0);
#endif
        double z0, z1;
        if (2u * c >= first) {
            normals_from_words(r[0], r[1], tab, z0, z1);
            step(z0, z1);
        }
        if (2u * c + 1u <= last) {
            normals_from_words(r[2], r[3], tab, z0, z1);
            step(z0, z1);
        }
    }
}

template <class Step>
__device__ __forceinline__ void rng_time_loop(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab,
                                              Step &&step)
{
    rng_time_loop(lane, step0, nb, tab, step, [](int) {});
}

// rng_time_loop for launches of a few waves per SIMD (a calibration- or default-sized path set: 10^5 paths are 1.5 waves per
// SIMD on this chip), where no other wave hides an LDS round trip: all eight table reads of a call are in flight before the
// first cubic (draw_issue / draw_finish) instead of four read-wait-evaluate rounds.  The same words for the same (path,
// step), the same operations on them: the same bits.  Costs 24 more live registers, which a full launch does not have.
template <class Step>
__device__ __forceinline__ void rng_time_loop_few_waves(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab,
                                                        Step &&step)
{
    if (nb <= 0) return;
    const uint32_t first = step0, last = step0 + static_cast<uint32_t>(nb) - 1u;
    for (uint32_t c = first >> 1; c <= (last >> 1); ++c) {
        uint32_t r[4];
        philox_draw(lane, c, r);
        DrawInFlight d;
        double z[4];
        draw_issue(r, tab, d);
        draw_finish(d, z);
        if (2u * c >= first) step(z[0], z[1]);
        if (2u * c + 1u <= last) step(z[2], z[3]);
    }
}

// rng_time_loop for launches of three to seven waves per SIMD (2^17 < paths < 2^20: the path counts the reference's own runs
// use -- 10^5 by default, 4 x 10^5 in its paper).  There the batched draw of rng_time_loop_few_waves turns against itself: the
// waves of a CU run the same code from the same start, so they all sit in the draw's LDS phase together (8 ds_read_b128 per
// wave, ~100 LDS cycles each with the bank conflicts of random indices) and then all in the VALU phase together -- LDS time and
// VALU time ADD (measured: 1250 cycles per step at three waves per SIMD, against 620 of issue and 670 of LDS).  Here every wave
// overlaps the two BY ITSELF: the words, indices and table reads of call c + 1 are issued before the two steps of call c, its
// cubics run after them (the draw does not depend on the state), so the LDS pipe works through a wave's reads while that
// wave's own steps issue.  The last trip draws a call nobody uses rather than branching inside the trip (a branch splits the
// scheduling region: logsv_chain_rng_sets_kernel measured it).  The edge halves of a slice that starts or ends on an odd step
// are peeled.  Which (path, step) sees which word is rng_time_loop's rule, the operations on them are normal_icdf32's: the
// same bits.  Costs ~40 live registers more than rng_time_loop (the eight table pieces in flight across two steps).
template <class Step>
__device__ __forceinline__ void rng_time_loop_ahead(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab,
                                                    Step &&step)
{
    if (nb <= 0) return;
    const uint32_t first = step0, last = step0 + static_cast<uint32_t>(nb) - 1u;
    uint32_t c = first >> 1, r[4];
    double a0, a1;
    if (first & 1u) {
        philox_draw(lane, c, r);
        normals_from_words(r[2], r[3], tab, a0, a1);
        step(a0, a1);
        ++c;
    }
    const uint32_t c_end = (last + 1u) >> 1;               // the full calls are [c, c_end)
    if (c < c_end) {
        DrawInFlight d;
        philox_draw(lane, c, r);
        draw_issue(r, tab, d);
        for (; c < c_end; ++c) {
            double z[4];
            draw_finish(d, z);
            philox_draw(lane, c + 1u, r);
            draw_issue(r, tab, d);
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

// The draw of ONE time step's pair of normals in three pieces (prepare: word -> t and table offset; read: the four
// ds_read_b128; finish: the two cubics), for the loops that keep one pair in flight under the step before it.  normal_icdf32's
// operations on the same operands: the same bits.
struct PairInFlight {
    double t[2];
    IcdfPiece e0[2], e1[2];
};
#if SVMC_ICDF_MIXED
__device__ __forceinline__ void pair_issue(uint32_t ra, uint32_t rb, const RngTables &tab, PairInFlight &d)
{
    d.t[0] = normal_icdf32_mixed(ra, tab.icdf);
    d.t[1] = normal_icdf32_mixed(rb, tab.icdf);
}
__device__ __forceinline__ void pair_finish(const PairInFlight &d, double &z0, double &z1)
{
    z0 = d.t[0];
    z1 = d.t[1];
}
#else
__device__ __forceinline__ void pair_issue(uint32_t ra, uint32_t rb, const RngTables &tab, PairInFlight &d)
{
    static_assert(SVMC_ICDF_RAW != 0, "the split draw is written for the raw form of the table");
    const uint32_t w[2] = {ra, rb};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        d.t[k] = icdf_lattice_point(w[k]);
        const uint32_t off = (double_hi(d.t[k]) >> (16 - SVMC_ICDF_M)) & ((static_cast<uint32_t>(SVMC_ICDF_SEGMENTS) - 1u) << 4);
        const char *base = reinterpret_cast<const char *>(tab.icdf) + off;
#if defined(SVMC_PROBE) && (SVMC_PROBE & 2)          // measurement build: no table reads in the draw
        asm volatile("" : : "v"(off));                 // the offset is still computed: the VALU stream is the product's
        d.e0[k] = IcdfPiece{1e-10, 1e-10};
        d.e1[k] = IcdfPiece{1e-20, 1e-30};
        (void)base;
#else
        d.e0[k] = *reinterpret_cast<const IcdfPiece *>(base);
        d.e1[k] = *reinterpret_cast<const IcdfPiece *>(base + 16 * SVMC_ICDF_SEGMENTS);
#endif
    }
}
__device__ __forceinline__ void pair_finish(const PairInFlight &d, double &z0, double &z1)
{
    double z[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double a = fabs(d.t[k]);
        double p = fma(d.e1[k].b, a, d.e1[k].a);
        p = fma(p, a, d.e0[k].b);
        p = fma(p, a, d.e0[k].a);
        z[k] = copysign(p, d.t[k]);
    }
    z0 = z[0];
    z1 = z[1];
}
#endif

// SVMC_PIPE_HOIST_KEYS (A/B hook, default 1): the few-waves loops below draw with the round keys hoisted out of the time loop
#ifndef SVMC_PIPE_HOIST_KEYS
#define SVMC_PIPE_HOIST_KEYS 1
#endif
// rng_time_loop with ONE PAIR ahead: the table reads of the next step's two normals are issued before this step runs and
// their cubics evaluated after it -- half the registers of rng_time_loop_ahead (one pair in flight, not four normals), for
// launches that still want five to seven waves per SIMD.  Same words, same operations: the same bits.
template <class Step>
__device__ __forceinline__ void rng_time_loop_pair_ahead(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab,
                                                         Step &&step)
{
    if (nb <= 0) return;
    const uint32_t first = step0, last = step0 + static_cast<uint32_t>(nb) - 1u;
    uint32_t c = first >> 1, r[4];
    double a0, a1;
    if (first & 1u) {
        philox_draw(lane, c, r);
        normals_from_words(r[2], r[3], tab, a0, a1);
        step(a0, a1);
        ++c;
    }
    const uint32_t c_end = (last + 1u) >> 1;               // the full calls are [c, c_end)
    if (c < c_end) {
        PairInFlight d;
        philox_draw(lane, c, r);
        pair_issue(r[0], r[1], tab, d);
        __builtin_amdgcn_sched_barrier(0);
        for (; c < c_end; ++c) {
            double z0, z1;
            pair_finish(d, z0, z1);
            pair_issue(r[2], r[3], tab, d);
            __builtin_amdgcn_sched_barrier(0);
            step(z0, z1);
            pair_finish(d, z0, z1);
            philox_draw<SVMC_PIPE_HOIST_KEYS != 0>(lane, c + 1u, r);   // the last trip draws a call nobody uses (no branch in the trip)
            pair_issue(r[0], r[1], tab, d);
            __builtin_amdgcn_sched_barrier(0);
            step(z0, z1);
        }
    }
    if (!(last & 1u)) {
        philox_draw(lane, last >> 1, r);
        normals_from_words(r[0], r[1], tab, a0, a1);
        step(a0, a1);
    }
}

// The same idea one level finer, for a step that itself waits on an LDS read (LogSV: the exp table): the step comes in two
// halves, front(z0, z1) -- everything up to and including the ISSUE of its table read -- and back() -- what consumes the
// value.  Between them runs work that does not depend on the state: the cubics of the NEXT step's pair (its reads are older
// than the step's own read, so they have landed when it has), the Philox call after this one, and the issue of the pair after
// that -- whose reads are younger than the step's own, so back() waits for the exp value only (s_waitcnt lgkmcnt(4)) and the
// pair's four reads stay in flight under back() and the next front().  A lone wave thus overlaps its own LDS round trips with
// its own arithmetic, and the waves of a CU cannot fall into a common LDS phase.  Same words, same operations: the same bits.
template <class Front, class Mid, class Back>
__device__ __forceinline__ void rng_time_loop_pipelined(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab,
                                                        Front &&front, Mid &&mid, Back &&back)
{
    if (nb <= 0) return;
    const uint32_t first = step0, last = step0 + static_cast<uint32_t>(nb) - 1u;
    uint32_t c = first >> 1, r[4];
    double a0, a1;
    if (first & 1u) {
        philox_draw(lane, c, r);
        normals_from_words(r[2], r[3], tab, a0, a1);
        front(a0, a1);
        mid();
        back();
        ++c;
    }
    const uint32_t c_end = (last + 1u) >> 1;               // the full calls are [c, c_end)
    if (c < c_end) {
        PairInFlight d;
        double z0, z1;
        philox_draw(lane, c, r);
        pair_issue(r[0], r[1], tab, d);
        pair_finish(d, z0, z1);
        pair_issue(r[2], r[3], tab, d);
        __builtin_amdgcn_sched_barrier(0);
        for (; c < c_end; ++c) {
            front(z0, z1);                                 // step 2c; its table read is now the youngest LDS operation
            __builtin_amdgcn_sched_barrier(0);
            pair_finish(d, z0, z1);                        // the normals of step 2c + 1
            philox_draw<SVMC_PIPE_HOIST_KEYS != 0>(lane, c + 1u, r);   // the last trip draws a call nobody uses (no branch in the trip)
            pair_issue(r[0], r[1], tab, d);                // ... of step 2c + 2: in flight under back() and the next front()
            mid();                                         // what of the step needs neither the table value nor another wave's time
            __builtin_amdgcn_sched_barrier(0);
            back();
            front(z0, z1);                                 // step 2c + 1
            __builtin_amdgcn_sched_barrier(0);
            pair_finish(d, z0, z1);                        // the normals of step 2c + 2
            pair_issue(r[2], r[3], tab, d);                // ... of step 2c + 3
            mid();
            __builtin_amdgcn_sched_barrier(0);
            back();
        }
    }
    if (!(last & 1u)) {
        philox_draw(lane, last >> 1, r);
        normals_from_words(r[0], r[1], tab, a0, a1);
        front(a0, a1);
        mid();
        back();
    }
}

template <class Front, class Back>
__device__ __forceinline__ void rng_time_loop_pipelined(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab,
                                                        Front &&front, Back &&back)
{
    rng_time_loop_pipelined(lane, step0, nb, tab, front, []() {}, back);
}

// the three forms of the generators' time loop, by launch size (generator_loop_for() in svmc_kernels.hip picks)
enum GenLoop { GEN_LOOP_FULL = 0, GEN_LOOP_FEW = 1, GEN_LOOP_PAIR = 2, GEN_LOOP_AHEAD = 3, GEN_LOOP_PIPE = 4 };
template <int LOOP, class Step>
__device__ __forceinline__ void gen_time_loop(const PhiloxLane &lane, uint32_t step0, int nb, const RngTables &tab, Step &&step)
{
    if constexpr (LOOP == GEN_LOOP_FEW) rng_time_loop_few_waves(lane, step0, nb, tab, step);
    else if constexpr (LOOP == GEN_LOOP_AHEAD) rng_time_loop_ahead(lane, step0, nb, tab, step);
    else if constexpr (LOOP == GEN_LOOP_PAIR || LOOP == GEN_LOOP_PIPE) rng_time_loop_pair_ahead(lane, step0, nb, tab, step);
    else rng_time_loop(lane, step0, nb, tab, step);
}

// stream 0 for a single (path, step), from scratch: the step's two UNSCALED N(0,1)  (svmc_fill_normals)
__device__ __forceinline__ void draw_normals(uint64_t seed, uint32_t c3, uint64_t path, uint32_t step,
                                             const RngTables &t, double &w0, double &w1)
{
    uint32_t r[4];
    philox_draw(seed, c3, path, step >> 1, r);
    const bool odd = (step & 1u) != 0u;
    normals_from_words(odd ? r[2] : r[0], odd ? r[3] : r[1], t, w0, w1);
}

// Heston QE (streams 4 and 5).  The scheme needs two normals every step (z0 for the log-price, z1 for the quadratic branch)
// and a uniform only in the exponential branch: the normals come from stream 4 exactly like stream 0's (one call per two
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
