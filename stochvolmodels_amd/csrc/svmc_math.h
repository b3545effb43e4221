// svmc_math.h -- the fp64 elementary functions of the stepping kernels, written for the gfx950 VALU.
//
// The on-device-RNG LogSV step is VALU-issue bound (DESIGN.md "Rooflines"): with the device libm (OCML)
// it spends 34% of its issue slots in log+sqrt, 15% in sincospi, 11% in exp and 5% in the IEEE divide
// (tools/ubench/ablate.hip).  The versions below do the same fp64 arithmetic with ~2.5x fewer
// instructions by using what this path knows about its arguments:
//   neg_log(u)      u > 0 normal                 -> no denormal/negative/NaN handling, integer frexp
//   neg_log_tab     same, 512-entry table        -> no reciprocal, cubic tail (Heston QE's martingale correction)
//   log_state(s)    any s >= 0, inf, NaN         -> ln(sigma) at slice starts, constants in scalar registers
//   sqrt_pos(t)     t in (0, 2^10) normal        -> no scaling, v_rsq_f64 seed + one Goldschmidt/Newton pass
//   sqrt_pos_1g     same, Goldschmidt step only  -> 2^-47 (Heston's sqrt(v))
//   normal_icdf32   a raw 32-bit word            -> N(0,1) by a piecewise cubic of the inverse CDF, 1024-segment table
//   exp_fast(x)     |x| < ~1.4e6                 -> 2-constant Cody-Waite reduction, v_ldexp_f64 saturates
//   exp_tab(x)      log-volatilities             -> 256-entry table, quadratic tail, one reduction constant
//   rcp_fast(a)     a normal, away from 0/inf    -> v_rcp_f64 seed + Newton, no div_scale/div_fixup
// Accuracy (tests/test_math_accuracy.py, vs 80-bit libm on the host build; tests/test_gpu_parity.py on
// the device build): exp_fast, sin, cos, table log <= 2 ULP; -log <= 3 ULP; sqrt_pos, 1/x correctly rounded on the
// sampled ranges; exp_tab <= 1.5 ULP on |x| <= 1 (see there).  Coefficients: tools/gen_minimax.py.
//
// The same source compiles for the host (g++, used only by the accuracy test) -- there the hardware
// seeds are emulated with single-precision reciprocals, the worst seed the refinement must cope with.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SVMC_HD __host__ __device__ __forceinline__
#else
#define SVMC_HD static inline
#endif

namespace svmc {

SVMC_HD double bits_to_double(uint32_t lo, uint32_t hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __hiloint2double(static_cast<int>(hi), static_cast<int>(lo));
#else
    const uint64_t b = (static_cast<uint64_t>(hi) << 32) | lo;
    double d;
    memcpy(&d, &b, 8);
    return d;
#endif
}

SVMC_HD uint32_t double_hi(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<uint32_t>(__double2hiint(d));
#else
    uint64_t b;
    memcpy(&b, &d, 8);
    return static_cast<uint32_t>(b >> 32);
#endif
}

SVMC_HD uint32_t double_lo(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<uint32_t>(__double2loint(d));
#else
    uint64_t b;
    memcpy(&b, &d, 8);
    return static_cast<uint32_t>(b);
#endif
}

// Horner step a*b + K with a wave-uniform constant K.  hipcc turns fma(p, z, K) into v_mov_b64 tmp, K ;
// v_fmac_f64 tmp, p, z (the VOP2 form needs the addend in the destination), i.e. one extra VALU issue per
// coefficient -- 30 of the 170 instructions of the LogSV step.  Pinning K to an SGPR pair gives the VOP3
// form v_fma_f64 d, a, b, s[K] with no move (one scalar operand per VOP3 is within the gfx950 constant-bus
// limit); the s_mov_b32 that load K issue on the scalar unit, beside the VALU stream.
SVMC_HD double fma_k(double a, double b, double k)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
    return d;
#else
    return fma(a, b, k);
#endif
}

// hardware reciprocal / reciprocal-square-root seeds
SVMC_HD double rcp_seed(double a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(a);
#else
    return static_cast<double>(1.0f / static_cast<float>(a));
#endif
}

SVMC_HD double rsq_seed(double a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(a);
#else
    return static_cast<double>(1.0f / sqrtf(static_cast<float>(a)));
#endif
}

// 1/a for a normal and far from the overflow/underflow edges: seed + two Newton steps (quadratic each).
SVMC_HD double rcp_fast(double a)
{
    double y = rcp_seed(a);
    double e = fma(-a, y, 1.0);
    y = fma(y, e, y);
    e = fma(-a, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// 1/a to ~2^-48 (one Newton step on the 2^-24 seed): enough wherever the quotient enters a small term.
SVMC_HD double rcp_1n(double a)
{
    const double y = rcp_seed(a);
    return fma(y, fma(-a, y, 1.0), y);
}

// sqrt(t) for normal t > 0 (no scaling): Goldschmidt step on the rsq seed, then one Newton correction.
SVMC_HD double sqrt_pos(double t)
{
    const double y = rsq_seed(t);
    double g = t * y;
    double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);                     // relative error ~2^-47 after the Goldschmidt step
    const double d = fma(-g, g, t);
    return fma(d, h, g);                  // h is only 2^-24 accurate: its error enters at 2^-71
}

// sqrt(t) to 2^-47: the Goldschmidt step alone (4 instructions instead of 6).  The radius of the Box-Muller pair:
// the normals carry 7e-15 relative, the accuracy class of the step's reciprocal (rcp_1n).
SVMC_HD double sqrt_pos_1g(double t)
{
    const double y = rsq_seed(t);
    const double g = t * y;
    const double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    return fma(g, r, g);
}

// the same, handing out h = (rsq seed) / 2 ~ 1 / (2 sqrt(t)) as well: a low-accuracy reciprocal for residual corrections
SVMC_HD double sqrt_pos_1g_h(double t, double &half_rsq)
{
    const double y = rsq_seed(t);
    const double g = t * y;
    const double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    half_rsq = h;
    return fma(g, r, g);
}

// sqrt(t) for t >= 0 including exact zero (the rsq seed of 0 is +inf): Heston's variance before its first floor.
SVMC_HD double sqrt_pos0(double t)
{
    const double r = sqrt_pos(t);
    return (t == 0.0) ? 0.0 : r;
}

// the same to 2^-47 (Heston QE's diffusion coefficient, which can be exactly zero)
SVMC_HD double sqrt_pos0_1g(double t)
{
    const double r = sqrt_pos_1g(t);
    return (t == 0.0) ? 0.0 : r;
}

// exp(x) = 2^n * (1 + r + r^2 E(r)),  n = rint(x log2 e),  r = x - n ln2 (hi/lo),  |r| <= ln2/2
SVMC_HD double exp_fast(double x)
{
    const double n = rint(x * 0x1.71547652b82fep+0);
    double r = fma(-n, 0x1.62e42fee00000p-1, x);
    r = fma(-n, 0x1.a39ef35793c76p-33, r);
    double p = 0x1.af38a9b0ec855p-26;
    p = fma_k(p, r, 0x1.289185613a3d6p-22);
    p = fma_k(p, r, 0x1.71de0dae63bb3p-19);
    p = fma_k(p, r, 0x1.a019b90d2ae7ap-16);
    p = fma_k(p, r, 0x1.a01a01a7c41d5p-13);
    p = fma_k(p, r, 0x1.6c16c1788bd90p-10);
    p = fma_k(p, r, 0x1.11111111109b3p-7);
    p = fma_k(p, r, 0x1.5555555553d63p-5);
    p = fma_k(p, r, 0x1.5555555555556p-3);
    p = fma_k(p, r, 0x1.0000000000001p-1);
    const double y = fma(p, r * r, r) + 1.0;
    return ldexp(y, static_cast<int>(n));
}

// exp(x) for ANY x (the payoff reductions: terminal log-returns that may be +-inf or NaN where the reference's own
// paths overflowed): exp_fast on the finite range that matters, exact saturation outside it -- inf above 746, +0 below
// -746 (e^746 overflows, e^-746 underflows past the last denormal), NaN stays NaN.  <= 2 ULP; 25 instructions where
// the device libm's exp takes about 40 and twice the registers.
SVMC_HD double exp_full(double x)
{
    const double e = exp_fast((x > 746.0) ? 746.0 : ((x < -746.0) ? -746.0 : x));    // NaN fails both tests and passes through
    return (x > 746.0) ? __builtin_huge_val() : ((x < -746.0) ? 0.0 : e);
}

// exp(x) with a 256-entry table: x = n ln2/256 + r, n = 256 k + j, |r| <= ln2/512; exp(x) = 2^k T[j] (1 + r + r^2 Q(r)),
// T[j] = fl(2^(j/256)), Q quadratic, fitted on the reduced interval (1e-17).  n comes out of the 1.5 2^52 rounding trick
// (its integer form is the low word of the biased sum, so no rint / convert) and the reduction uses ONE constant for
// ln2/256: r is off by n |fl(c) - c|, i.e. 3.4e-17 |x| relative in the result.  The arguments here are
// log-volatilities: <= 1.3 ULP on |x| <= 1, 2.5 ULP at |x| = 5 (sigma between 0.007 and 150), 2.4e-14 at the ends of
// the double range (tests/test_math_accuracy.py).  11 instructions, 8 of them fp64 arithmetic, against exp_fast's 19;
// the table read goes through the LDS pipe beside the VALU stream.
SVMC_HD double exp_tab(double x, const double *tab)
{
    const double kf = fma(x, 0x1.71547652b82fep+8, 0x1.8p+52);
    const int ni = static_cast<int>(double_lo(kf));
    const double n = kf - 0x1.8p+52;
    const double r = fma(-n, 0x1.62e42fefa39efp-9, x);
    const double t = tab[ni & 255];
    double q = 0x1.55555660ffb24p-5;
    q = fma_k(q, r, 0x1.555556e6d4e0ep-3);
    q = fma_k(q, r, 0x1.0000000000000p-1);
    const double p = fma(q, r * r, r);
    return ldexp(fma(t, p, t), ni >> 8);
}

// The same exponential for an argument carried in units of ln2/256: exp2u_tab(y) = 2^(y/256) = exp(y ln2/256).  The LogSV
// stepping kernels keep the log-volatility in these units (their per-step constants are pre-multiplied by 256/ln2 on
// the host), so the reduction is EXACT -- n = rint(y) by the 1.5 2^52 trick, r = y - n with |r| <= 1/2 and no rounding
// -- and the accuracy no longer depends on the size of the argument: 2^(r/256) - 1 = r E(r) with a cubic E (5e-18),
// the table and the scaling as in exp_tab.  11 instructions, three of them plain adds where exp_tab has FMAs, and
// both constants of the reduction are gone (exp_tab needs one of its two in a VGPR pair).  <= 1.1 ULP.
// ... in three pieces, so that a kernel that advances SEVERAL independent states per lane can run each piece for all of
// them (all the table reads in flight together) and still perform, per state, exactly these operations in this order:
//   exp2u_reduce: n = rint(y) (returned as an integer), r = y - n;  exp2u_tail: r E(r);  exp2u_scale: 2^(n >> 8) T (1 + tail)
SVMC_HD void exp2u_reduce(double y, int &ni, double &r)
{
    const double kf = y + 0x1.8p+52;
    ni = static_cast<int>(double_lo(kf));
    r = y - (kf - 0x1.8p+52);
}

SVMC_HD double exp2u_tail(double r)
{
    double q = 0x1.3b2ab8452c312p-39;
    q = fma_k(q, r, 0x1.c6b0903967234p-29);
    q = fma_k(q, r, 0x1.ebfbdff82c584p-19);
    q = fma_k(q, r, 0x1.62e42fefa39d8p-9);
    return q * r;
}

// exp2u_tail with its four coefficients in VECTOR registers the caller keeps across its time loop (the few-waves generators:
// registers to spare, and every instruction of a lone wave is an issue slot): plain v_fma_f64 on the same operands -- the same
// bits as exp2u_tail -- but no inline assembly in the loop, so the scheduler may interleave the chain with the draw's integer
// work and the hazard recogniser does not pad each step of it with an s_nop.
struct Exp2uTailV {
    double k0, k1, k2, k3;
};
SVMC_HD Exp2uTailV exp2u_tail_consts()
{
    Exp2uTailV c{0x1.3b2ab8452c312p-39, 0x1.c6b0903967234p-29, 0x1.ebfbdff82c584p-19, 0x1.62e42fefa39d8p-9};
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(c.k0), "+v"(c.k1), "+v"(c.k2), "+v"(c.k3));
#endif
    return c;
}
SVMC_HD double exp2u_tail_v(double r, const Exp2uTailV &c)
{
    double q = fma(c.k0, r, c.k1);
    q = fma(q, r, c.k2);
    q = fma(q, r, c.k3);
    return q * r;
}

SVMC_HD double exp2u_scale(double t, double p, int ni)
{
    return ldexp(fma(t, p, t), ni >> 8);
}

SVMC_HD double exp2u_tab(double y, const double *tab)
{
    int ni;
    double r;
    exp2u_reduce(y, ni, r);
#if defined(SVMC_PROBE) && (SVMC_PROBE & 1)          // measurement build: no exp-table read
    const double t = 1.0;
    (void)tab;
#else
    const double t = tab[ni & 255];
#endif
    return exp2u_scale(t, exp2u_tail(r), ni);
}

// -ln(u) for any positive normal u (the RNG calls it on (0,1); Heston QE on arguments around 1).  u = m 2^k with m in [sqrt(1/2), sqrt(2)) taken from the exponent field,
// f = m - 1, s = f/(2+f), ln(1+f) = f - (f^2/2 - s (f^2/2 + R)), R = s^2 G(s^2)   (Cody-Waite / fdlibm form)
SVMC_HD double neg_log(double u)
{
#if defined(__clang__)
    // no implicit contraction: which of the products below the compiler would fuse with a following add depends on the code
    // around the (inlined) call, and the generators that take ln(sigma) at slice starts -- one launch per slice, or the whole
    // chain in one launch -- must produce the same bits
#pragma clang fp contract(off)
#endif
    uint32_t hx = double_hi(u);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    const int k = static_cast<int>(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    const double m = bits_to_double(double_lo(u), hx);
    const double dk = static_cast<double>(k);
    const double f = m - 1.0;
    const double s = f * rcp_1n(2.0 + f);   // s only multiplies the O(f^2) tail: 2^-48 is ample
    const double z = s * s;
    double p = 0x1.2b5900de53b32p-3;
    p = fma_k(p, z, 0x1.39fe51a7c18f9p-3);
    p = fma_k(p, z, 0x1.7462b51cb66b1p-3);
    p = fma_k(p, z, 0x1.c71c62e3f11e6p-3);
    p = fma_k(p, z, 0x1.2492492df281ap-2);
    p = fma_k(p, z, 0x1.99999999952d7p-2);
    p = fma_k(p, z, 0x1.5555555555558p-1);
    const double R = p * z;
    const double hfsq = (0.5 * f) * f;
    const double t2 = fma(-s, hfsq + R, hfsq);       // f^2/2 - s (f^2/2 + R)
    const double lg = f - t2;                        // ln(m)
    const double a = fma(dk, 0x1.a39ef35793c76p-33, lg);
    return fma(-dk, 0x1.62e42fee00000p-1, -a);      // -(k ln2 + ln m)
}

// ln(s) of a state variable: any s >= 0 including denormals, 0 (-> -inf), +inf (-> +inf) and NaN / negative (-> NaN).
// The generators take ln(sigma) at every slice start; the device libm's log keeps its constants in vector registers, which
// the compiler hoists out of the whole-chain kernel's slice loop and then spills across the time loop -- neg_log() takes
// its constants as scalar operands.  <= 4 ULP (tests/test_math_accuracy.py).
SVMC_HD double log_state(double s)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const bool tiny = s < 0x1.0p-1000;
    double r = -neg_log(tiny ? s * 0x1.0p+512 : s);
    r = tiny ? fma(-512.0, 0x1.62e42fefa39efp-1, r) : r;
    r = (s == __builtin_huge_val()) ? __builtin_huge_val() : r;
    r = (s == 0.0) ? -__builtin_huge_val() : r;
    return (s >= 0.0) ? r : __builtin_nan("");
}

// Table-assisted -ln(u), any positive normal u: the top 9 bits of the [sqrt(1/2), sqrt(2)) mantissa pick
// { fl(1/c_j), ln fl(1/c_j) } (tools/gen_log_table.py); f = fma(m, 1/c_j, -1) has |f| <= 2^-10, so ln(1+f) is
// f + f^2 P(f) with a cubic P -- no reciprocal and a 3-step Horner chain instead of the divide + 7-term polynomial of
// neg_log().  The interval containing m = 1 has c = 1 exactly, keeping full relative accuracy as u -> 1.
// On the device `tab` is an 8 KB LDS copy: a 16-byte ds_read per call, in the LDS pipe beside the VALU stream.
struct alignas(16) LogTabEntry {
    double inv_c, neg_log_c;
};

// EXP2_SCALE: returns -ln(u 2^EXP2_SCALE) -- the scale only shifts the integer exponent, so it is free
template <int EXP2_SCALE = 0>
SVMC_HD double neg_log_tab(double u, const LogTabEntry *tab)
{
    uint32_t hx = double_hi(u);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    const int k = static_cast<int>(hx >> 20) - (0x3ff - EXP2_SCALE);
    const uint32_t frac = hx & 0x000fffffu;
    const LogTabEntry e = tab[frac >> 11];
    const double m = bits_to_double(double_lo(u), frac + 0x3fe6a09eu);
    const double dk = static_cast<double>(k);
    const double f = fma(m, e.inv_c, -1.0);        // |f| <= 2^-10
    double p = 0x1.9999ac9fdd43ep-3;               // log1p(f) = f + f^2 P(f), P fitted on |f| <= 2^-10 (2e-17 relative)
    p = fma_k(p, f, -0x1.00000b18fcc98p-2);
    p = fma_k(p, f, 0x1.5555555555419p-2);
    p = fma_k(p, f, -0x1.ffffffffffe8fp-2);
    const double r = fma(f * f, p, f);             // log1p(f)
    const double nl = e.neg_log_c - r;             // -ln(m): with k = 0 and c = 1 this is -log1p(u - 1), relative accuracy kept
    // one constant for ln2: k ln2 is rounded once (in the fma), and fl(ln2) is off by 2^-55 relative -- at most
    // 0.1 ULP of any result with k != 0, all of which exceed 0.34
    return fma(-dk, 0x1.62e42fefa39efp-1, nl);
}

// One N(0,1) variate from ONE 32-bit word by inversion (random stream version 4, tools/gen_icdf_table.py): the word is
// read as a signed integer t = k (version 3: k + 1/2 -- an add per normal that bought nothing: the magnitudes 1 .. 2^31 - 1
// occur with both signs either way, and the two unpaired words k = 0 and k = -2^31 sit where the function is 0), and
//     z = sign(t) P_j(|t| - c_j)   ~   sign(t) * -Phi^-1(|t| 2^-32),      |z| <= 6.23,  z = 0 for t = 0 and |t| = 2^31
// with P_j the segment's cubic.  The segment index is read off the high word of t -- five exponent bits and the top
// SVMC_ICDF_M mantissa bits: 32 octaves x 2^M equal parts, i.e. geometric spacing towards the tail where Phi^-1 is singular
// -- by one shift and one mask that leave the BYTE offset of the segment's 16-byte pieces; the pieces sit in arrays one
// after the other so that one address serves all the ds_read_b128 (LDS pipe, beside the VALU stream).
//   RAW form (the committed table): c_j = 0, the cubic runs in |t| itself, pieces {a0, a1}, {a2, a3}: 8 VALU instructions
//   -- cvt, add, shift, and, three FMAs on |t| (the modulus is an operand modifier), v_bfi_b32 for the sign -- and 32 table
//   bytes per normal, where the Box-Muller pair of stream version 2 took 19 instructions per normal.
//   Edge form: c_j = the segment's lower edge = |t| with the mantissa below the segment bits cleared: 10 instructions.
//   Midpoint form: c_j from the table, pieces {c, a0}, {a1, a2}, {a3, a4}: 9 instructions but 40 table bytes -- measured
//   LDS-bound (the LDS pipe 95 % busy, 61 % of it bank conflicts of the randomly indexed reads) and no faster.
//   (profiles/r03_ab_stream_v3*.jsonl: 2.35 ms for stream v2's loop on C2, 1.86 midpoint, 1.80 edge, 1.70 raw)
// This evaluation order IS the stream's definition: the CPU twin (oracle/svmc_oracle.c svo_normal_from_word) evaluates the
// same expression.
struct alignas(16) IcdfPiece {
    double a, b;
};

// the lattice point of a word: the signed integer itself (stream version 4: magnitudes 0 .. 2^31, both ends map to z = 0, every
// other magnitude occurs with both signs -- exactly symmetric with no add), or k + 1/2 (version 3's table: SVMC_ICDF_HALF_LATTICE)
#ifndef SVMC_ICDF_HALF_LATTICE
#define SVMC_ICDF_HALF_LATTICE 0
#endif
SVMC_HD double icdf_lattice_point(uint32_t w)
{
#if SVMC_ICDF_HALF_LATTICE
    return static_cast<double>(static_cast<int32_t>(w)) + 0.5;
#else
    return static_cast<double>(static_cast<int32_t>(w));
#endif
}

template <int M, int SEGMENTS, int DEG, bool EDGE = false, bool RAW = false>
SVMC_HD double normal_icdf32(uint32_t w, const IcdfPiece *tab)
{
#if !SVMC_ICDF_HALF_LATTICE
    static_assert(RAW, "the integer lattice needs the raw form: t = 0 reads segment 0, whose line passes through the origin");
#endif
    const double t = icdf_lattice_point(w);
    const uint32_t hi = double_hi(t);
    const uint32_t off = (hi >> (16 - M)) & ((static_cast<uint32_t>(SEGMENTS) - 1u) << 4);
    const char *base = reinterpret_cast<const char *>(tab) + off;
#if defined(SVMC_PROBE) && (SVMC_PROBE & 2)          // measurement build: no table reads in the draw
    asm volatile("" : : "v"(off));                     // the offset is still computed: the VALU stream is the product's
    const IcdfPiece e0 = IcdfPiece{1e-10, 1e-10};
    const IcdfPiece e1 = IcdfPiece{1e-20, 1e-30};
    (void)base;
#else
    const IcdfPiece e0 = *reinterpret_cast<const IcdfPiece *>(base);
    const IcdfPiece e1 = *reinterpret_cast<const IcdfPiece *>(base + 16 * SEGMENTS);
#endif
    double p;
    if (RAW) {
        // pieces {a0, a1}, {a2, a3}: the cubic in |t| ITSELF.  Re-expanding a segment's polynomial about 0 makes its terms
        // O(1) quantities that cancel to the result -- (c / width)^k = 2^(k M) times the centred terms, i.e. rounding errors of
        // a few 1e-16 absolute, seven orders below the table's own 7e-10 -- and saves the centre: no table bytes, no v_and, no
        // subtraction
        const double a = fabs(t);
        p = fma(e1.b, a, e1.a);
        p = fma(p, a, e0.b);
        p = fma(p, a, e0.a);
    } else if (EDGE) {
        // pieces {a0, a1}, {a2, a3}: the polynomial runs in |t| - (the segment's lower edge), and the edge is |t| with the
        // mantissa below the segment bits cleared -- one v_and_b32 on the high word instead of 8 table bytes
        const double edge = bits_to_double(0u, hi & (0x7FFFFFFFu & ~((1u << (20 - M)) - 1u)));
        const double d = fabs(t) - edge;
        if (DEG == 4) {
            const IcdfPiece e2 = *reinterpret_cast<const IcdfPiece *>(base + 32 * SEGMENTS);
            p = fma(e2.a, d, e1.b);
        } else {
            p = e1.b;
        }
        p = fma(p, d, e1.a);
        p = fma(p, d, e0.b);
        p = fma(p, d, e0.a);
    } else {
        const IcdfPiece e2 = *reinterpret_cast<const IcdfPiece *>(base + 32 * SEGMENTS);
        const double d = fabs(t) - e0.a;
        p = (DEG == 4) ? fma(e2.b, d, e2.a) : e2.a;
        p = fma(p, d, e1.b);
        p = fma(p, d, e1.a);
        p = fma(p, d, e0.b);
    }
    return copysign(p, t);
}

}  // namespace svmc
