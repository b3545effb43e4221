// svmc_rng.h -- counter-based randoms for the gfx950 kernels.
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11) keyed by the seed and indexed by
// (global path id, chain-global step id, stream, call id): a lane owns a path, so it derives every
// increment it needs in registers, no generator state lives in memory and the result is independent of
// how paths are sharded over GPUs.  Replaces the reference's serial MT19937+polar draw of two
// [nb_steps, nb_path] arrays (pricers/logsv_pricer.py:1025-1026, pricers/heston_pricer.py:369-370).
//
// Stream definition (DESIGN.md "RNG"; CPU twin: oracle/svmc_oracle.c svo_draw_normals):
//   (r0, r1, r2, r3) = philox4x32_10(ctr = (path_lo, path_hi, step, stream | call_id << 8), key = seed)
//   u1 = double(1.m) - (1 - 2^-53),  m = top 52 bits of r1:r0          in (0,1), exact
//   rr = double(1.m) - 1.5,          m = top 52 bits of r3:r2          in [-1/2, 1/2), exact
//   s0 = -1 if r2 & 1 else +1,  s1 = -1 if r2 & 2 else +1               two sign bits (not used by rr)
//   stream 0:  R = sqrt(-ln u1), x = (pi/2) rr,  (w0, w1) = R (s0 (cos x - sin x), s1 (cos x + sin x))
//              = sqrt(-2 ln u1) (s0 cos(x + pi/4), s1 sin(x + pi/4)):  a Box-Muller pair whose angle is uniform on the
//              circle by construction (x + pi/4 uniform on the first quadrant, independent signs)
//   stream 1:  uniform = u1
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svmc_log_table.h"
#include "svmc_math.h"

namespace svmc {

// The tables of the draw, constant memory -> LDS once per block: 8 KB for neg_log_tab() and 4 KB for cossin_diag_tab()
// (blocks are 256 threads: two + one entries per thread); the stepping kernels that call exp_tab() stage its 2 KB
// with them behind the same barrier.
__constant__ LogTabEntry g_log_table[512] = {SVMC_LOG_TABLE_INIT};
__constant__ DiagTabEntry g_diag_table[256] = {SVMC_DIAG_TABLE_INIT};
__constant__ double g_exp_table[256] = {SVMC_EXP_TABLE_INIT};

struct RngTables {
    const LogTabEntry *log;
    const DiagTabEntry *diag;
};

struct RngTablesLds {
    LogTabEntry log[512];
    DiagTabEntry diag[256];
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
        lds.diag[i] = g_diag_table[i];
    }
    __syncthreads();
    return RngTables{lds.log, lds.diag};
}

__device__ __forceinline__ RngTables stage_tables(RngTablesLds &lds, double (&lds_exp)[256])
{
    for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) {
        lds.log[i] = g_log_table[i];
        lds.log[i + 256u] = g_log_table[i + 256u];
        lds.diag[i] = g_diag_table[i];
    }
    for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) lds_exp[i] = g_exp_table[i];
    __syncthreads();
    return RngTables{lds.log, lds.diag};
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&r)[4])
{
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
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
// (path_lo, path_hi, step, c3): in round 1 the product M0 * path_lo, the word path_hi ^ k0 and the whole new third word
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
    c0 = l.hi_b ^ __builtin_amdgcn_readfirstlane(c1_r1 ^ k0);              // (uniform ^ uniform) stays on the scalar unit
    uint32_t c2 = __builtin_amdgcn_bitop3_b32(static_cast<uint32_t>(p0 >> 32), l.lo0, k1, 0x96);
    uint32_t c1 = l.lo_b;
    uint32_t c3 = static_cast<uint32_t>(p0);
    k0 += W0;
    k1 += W1;
#pragma unroll
    for (int i = 2; i < 10; ++i) {
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

__device__ __forceinline__ void philox_draw(uint64_t seed, uint32_t c3, uint64_t path, uint32_t step, uint32_t (&r)[4])
{
    philox4x32_10(static_cast<uint32_t>(path), static_cast<uint32_t>(path >> 32), step, c3,
                  static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32), r);
}

// the pair from the four words of one Philox call: radius from r1:r0, direction from r3:r2 (+ the sign bits r2 & 3)
__device__ __forceinline__ void normals_from_words(const uint32_t (&r)[4], const RngTables &t, double &w0, double &w1)
{
    const double u1 = mantissa_1_2(r[0], r[1]) - (1.0 - 0x1.0p-53);
    const double R = sqrt_pos_1g(neg_log_tab(u1, t.log));  // sqrt(-ln u1): the sqrt2 lives in (a, b)
    double a, b;
    cossin_diag_tab(r[2], r[2], r[3], t.diag, a, b);
    w0 = R * a;
    w1 = R * b;
}

// stream 4 (Heston QE): the Box-Muller pair AND the uniform of the exponential branch from ONE Philox call.
// u1 and the angle keep 42 mantissa bits each (r1 / r3 + the top 10 bits of r0 / r2), the two direction signs are r2 & 1 and r2 & 2, and
// the 32 bits left over (r0[21:0] : r2[11:2]) make u = (k + 0.5) 2^-32 -- exact in fp64, so the CPU twin gets the
// same bits.  42-bit radii reach 7.6 sigma; a 32-bit uniform truncates the exponential branch at e^-22.
__device__ __forceinline__ void qe_from_words(const uint32_t (&r)[4], const RngTables &t, double &w0, double &w1, double &u)
{
    const double u1 = mantissa_1_2(r[0] & 0xFFC00000u, r[1]) - (1.0 - 0x1.0p-53);
    const uint32_t k = ((r[0] & 0x3FFFFFu) << 10) | ((r[2] >> 2) & 0x3FFu);
    u = fma(static_cast<double>(k), 0x1.0p-32, 0x1.0p-33);
    const double R = sqrt_pos_1g(neg_log_tab(u1, t.log));
    double a, b;
    cossin_diag_tab(r[2], r[2] & 0xFFC00000u, r[3], t.diag, a, b);
    w0 = R * a;
    w1 = R * b;
}

// stream 0: Box-Muller pair of UNSCALED N(0,1)
__device__ __forceinline__ void draw_normals(uint64_t seed, uint32_t c3, uint64_t path, uint32_t step,
                                             const RngTables &t, double &w0, double &w1)
{
    uint32_t r[4];
    philox_draw(seed, c3, path, step, r);
    normals_from_words(r, t, w0, w1);
}

__device__ __forceinline__ void draw_qe(uint64_t seed, uint32_t c3, uint64_t path, uint32_t step,
                                        const RngTables &t, double &w0, double &w1, double &u)
{
    uint32_t r[4];
    philox_draw(seed, c3 | 4u, path, step, r);
    qe_from_words(r, t, w0, w1, u);
}

// the same pair from a prepared lane state (philox_prepare outside the time loop)
__device__ __forceinline__ void draw_normals(const PhiloxLane &lane, uint32_t step, const RngTables &t, double &w0,
                                             double &w1)
{
    uint32_t r[4];
    philox_draw(lane, step, r);
    normals_from_words(r, t, w0, w1);
}

// draw_qe from a prepared lane state (prepare it with the stream-4 tag: philox_prepare(seed, c3 | 4u, path))
__device__ __forceinline__ void draw_qe(const PhiloxLane &lane, uint32_t step, const RngTables &t, double &w0,
                                        double &w1, double &u)
{
    uint32_t r[4];
    philox_draw(lane, step, r);
    qe_from_words(r, t, w0, w1, u);
}

// stream 1: one uniform in (0,1)
__device__ __forceinline__ double draw_uniform(uint64_t seed, uint32_t c3, uint64_t path, uint32_t step)
{
    uint32_t r[4];
    philox_draw(seed, c3 | 1u, path, step, r);
    return mantissa_1_2(r[0], r[1]) - (1.0 - 0x1.0p-53);
}

}  // namespace svmc
