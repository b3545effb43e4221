// svmc_math.h -- the fp64 elementary functions of the stepping kernels, written for the gfx950 VALU.
//
// The on-device-RNG LogSV step is VALU-issue bound (DESIGN.md "Rooflines"): with the device libm (OCML)
// it spends 34% of its issue slots in log+sqrt, 15% in sincospi, 11% in exp and 5% in the IEEE divide
// (tools/ubench/ablate.hip).  The versions below do the same fp64 arithmetic with ~2.5x fewer
// instructions by using what this path knows about its arguments:
//   neg_log(u)      u > 0 normal                 -> no denormal/negative/NaN handling, integer frexp
//   neg_log_tab     same, 512-entry table        -> no reciprocal, cubic tail (the RNG's radius)
//   sqrt_pos(t)     t in (0, 2^10) normal        -> no scaling, v_rsq_f64 seed + one Goldschmidt/Newton pass
//   sqrt_pos_1g     same, Goldschmidt step only  -> 2^-47 (the RNG's radius)
//   cossin_circle_tab32  a raw 32-bit angle   -> no range reduction, no quadrant logic, 256-entry table
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
SVMC_HD double exp2u_tab(double y, const double *tab)
{
    const double kf = y + 0x1.8p+52;
    const int ni = static_cast<int>(double_lo(kf));
    const double r = y - (kf - 0x1.8p+52);
    const double t = tab[ni & 255];
    double q = 0x1.3b2ab8452c312p-39;
    q = fma_k(q, r, 0x1.c6b0903967234p-29);
    q = fma_k(q, r, 0x1.ebfbdff82c584p-19);
    q = fma_k(q, r, 0x1.62e42fefa39d8p-9);
    const double p = q * r;
    return ldexp(fma(t, p, t), ni >> 8);
}

// -ln(u) for any positive normal u (the RNG calls it on (0,1); Heston QE on arguments around 1).  u = m 2^k with m in [sqrt(1/2), sqrt(2)) taken from the exponent field,
// f = m - 1, s = f/(2+f), ln(1+f) = f - (f^2/2 - s (f^2/2 + R)), R = s^2 G(s^2)   (Cody-Waite / fdlibm form)
SVMC_HD double neg_log(double u)
{
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

struct alignas(32) CircleTabEntry {
    double a, b;      // sqrt2 (cos, sin) at the interval midpoint
    double c;         // the midpoint in units of 2^-32 turns: j 2^24 + 2^23 - 1/2
    double pad;
};

// The direction of a Box-Muller pair from ONE 32-bit word (stream version 2): the word IS the angle,
// t = 2 pi (w + 1/2) 2^-32 on the full circle -- no sign bits, no quadrant logic.  w[31:24] picks one of 256 intervals
// with midpoint t_j = 2 pi (j + 1/2)/256; D = w - (j 2^24 + 2^23 - 1/2) = cvt(w) - center[j] is the offset from it in
// units of 2^-32 turns, an exact double with |D| < 2^23, and with y = (2 pi 2^-32) D, |y| <= 0.01227:
//     a = sqrt2 cos t = A_j cos y - B_j sin y,    b = sqrt2 sin t = B_j cos y + A_j sin y,
// {A_j, B_j, c_j} = {sqrt2 cos t_j, sqrt2 sin t_j, the midpoint} from one 256-entry table of 32-byte entries (8 KB, LDS on
// the device: a ds_read_b128 and a ds_read_b64 off ONE address, beside the VALU stream).  sin y: three Taylor terms (next: 8e-18 absolute); cos y:
// four (next: 1e-20); the 2 pi 2^-32 scale lives in the coefficients.  12 VALU instructions + the index.
// Absolute accuracy 5e-16 on values up to sqrt2.
SVMC_HD void cossin_circle_tab32(uint32_t w, const CircleTabEntry *tab, double &a, double &b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(w));                    // a plain register value (see cossin_diag_tab)
#endif
    const CircleTabEntry &e = tab[w >> 24];
    const double D = static_cast<double>(w) - e.c;
    const double z = D * D;
    double ps = 0x1.466bc6775aae2p-154;            //  (2 pi 2^-32)^5 / 120
    ps = fma_k(ps, z, -0x1.4abbce625be53p-91);     // -(2 pi 2^-32)^3 / 6
    ps = fma_k(ps, z, 0x1.921fb54442d18p-30);      //   2 pi 2^-32
    const double sn = D * ps;                      // sin y
    double pc = -0x1.55d3c7e3cbffap-186;           // -(2 pi 2^-32)^6 / 720
    pc = fma_k(pc, z, 0x1.03c1f081b5ac4p-122);     //  (2 pi 2^-32)^4 / 24
    pc = fma_k(pc, z, -0x1.3bd3cc9be45dep-60);     // -(2 pi 2^-32)^2 / 2
    const double cs = fma_k(pc, z, 1.0);           // cos y
    a = fma(-e.b, sn, e.a * cs);
    b = fma(e.a, sn, e.b * cs);
}

}  // namespace svmc
