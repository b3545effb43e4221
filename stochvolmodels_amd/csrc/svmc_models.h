// svmc_models.h -- per-path time-step functions (one lane = one path, state in registers).
//
// The expressions keep the reference's evaluation order; hipcc contracts a*b+c into FMA (one rounding
// instead of two), so against the reference on identical W the state agrees to rounding level, not
// bitwise: tests state 1e-12 relative.
#pragma once
#include <hip/hip_runtime.h>

#include "svmc_math.h"

namespace svmc {

// ---- LogSV, Eq. (3.59): pricers/logsv_pricer.py:1032-1045 ------------------------------------------
struct LogsvConsts {
    double dt, sdt, theta, kappa1, kappa2, beta, volvol, eta, eta2;
    double alpha_half;      // alpha*0.5: -0.5 spot measure, +0.5 inverse measure  (:1032-1035)
    double adj;             // 0 | beta*eta                                        (:1035)
    double half_vartheta2;  // 0.5*(beta^2 + volvol^2)                             (:1037)
    double k1theta;         // kappa1*theta                                        (:1043)
};

inline LogsvConsts make_logsv_consts(double dt, double theta, double kappa1, double kappa2, double beta,
                                     double volvol, double eta, int is_spot_measure)
{
    LogsvConsts c;
    c.dt = dt;
    c.sdt = sqrt(dt);
    c.theta = theta;
    c.kappa1 = kappa1;
    c.kappa2 = kappa2;
    c.beta = beta;
    c.volvol = volvol;
    c.eta = eta;
    c.eta2 = eta * eta;
    c.alpha_half = is_spot_measure ? -0.5 : 0.5;
    c.adj = is_spot_measure ? 0.0 : beta * eta;
    c.half_vartheta2 = 0.5 * (beta * beta + volvol * volvol);
    c.k1theta = kappa1 * theta;
    return c;
}

// w0, w1 are the scaled increments sqrt(dt)*N(0,1).  The reference's evaluation order with every multiply-add written as
// an explicit fma(): which products the compiler would contract on its own depends on the code around the call, and the
// kernels that inline this step (one parameter set per lane, several per lane) must give the same bits.
__device__ __forceinline__ void logsv_step(const LogsvConsts &c, double &x, double &L, double &sigma,
                                           double &qvar, double w0, double w1)
{
    const double s = sigma;
    const double s2dt = ((c.eta2 * s) * s) * c.dt;                                              // :1041
    double drift = fma(c.k1theta, rcp_fast(s), -c.kappa1);      // k1theta/s: v_rcp_f64 + 2 Newton steps (<= 1 ULP)
    drift = fma(c.kappa2, c.theta - s, drift);
    drift = fma(c.adj, s, drift) - c.half_vartheta2;
    x = fma(c.eta * s, w0, fma(c.alpha_half, s2dt, x));                                         // :1042
    L = fma(c.volvol, w1, fma(c.beta, w0, fma(drift, c.dt, L)));                                // :1043
    const double sn = exp_fast(L);                                                              // :1044
    sigma = sn;
    qvar = fma(0.5, fma((c.eta2 * sn) * sn, c.dt, s2dt), qvar);                                 // :1045
}

// ---- the same LogSV step, regrouped for the issue-bound on-device-RNG kernel -------------------------
// Identical in exact arithmetic to logsv_step; constants are pre-multiplied on the host, the divide is a
// v_rcp_f64 seed + one Newton step (the quotient only enters the O(dt) drift), sigma^2 is carried between
// steps, exp is svmc_math.h's.  34 fp64 instructions instead of ~75.  Rounding differs from the reference
// order at the 1e-16 level per step; tests state 1e-10 on terminal states against the reference.
struct LogsvFast {
    double A;     // eta^2 dt
    double hA;    // eta^2 dt / 2
    double ahA;   // alpha/2 * eta^2 dt
    double B;     // eta sqrt(dt)
    double c1;    // kappa1 theta dt
    double c2;    // (adj - kappa2) dt
    double c3;    // (kappa2 theta - kappa1 - vartheta^2/2) dt
    double bs;    // beta sqrt(dt)
    double es;    // volvol sqrt(dt)
};

inline LogsvFast make_logsv_fast(const LogsvConsts &c)
{
    LogsvFast f;
    f.A = c.eta2 * c.dt;
    f.hA = 0.5 * f.A;
    f.ahA = c.alpha_half * f.A;
    f.B = c.eta * c.sdt;
    f.c1 = c.k1theta * c.dt;
    f.c2 = (c.adj - c.kappa2) * c.dt;
    f.c3 = (c.kappa2 * c.theta - c.kappa1 - c.half_vartheta2) * c.dt;
    f.bs = c.beta * c.sdt;
    f.es = c.volvol * c.sdt;
    return f;
}

// The accumulator-form kernels carry L = ln(sigma) in units of ln2/256 (svmc_math.h exp2u_tab: exact reduction): the
// five constants that advance L are multiplied by 256/ln2 once, on the host.
constexpr double LOG_UNITS_PER_NAT = 0x1.71547652b82fep+8;     // 256 / ln2

inline LogsvFast logsv_fast_in_log_units(LogsvFast f)
{
    f.c1 *= LOG_UNITS_PER_NAT;
    f.c2 *= LOG_UNITS_PER_NAT;
    f.c3 *= LOG_UNITS_PER_NAT;
    f.bs *= LOG_UNITS_PER_NAT;
    f.es *= LOG_UNITS_PER_NAT;
    return f;
}

// z0, z1 are UNSCALED N(0,1); s2 = sigma^2 is carried; exp_of(L) is the exponential to use (exp_fast, or exp_tab with
// the block's LDS table in the issue-bound kernels)
template <class Exp>
__device__ __forceinline__ void logsv_step_fast(const LogsvFast &f, double &x, double &L, double &sigma, double &s2,
                                                double &qvar, double z0, double z1, Exp &&exp_of)
{
    const double s = sigma;
    const double y = rcp_1n(s);
    x = fma(f.ahA, s2, x);
    x = fma(f.B * s, z0, x);
    const double d = fma(f.c1, y, fma(f.c2, s, f.c3));
    L = fma(f.es, z1, fma(f.bs, z0, L + d));
    const double sn = exp_of(L);
    const double s2n = sn * sn;
    qvar = fma(f.hA, s2 + s2n, qvar);
    sigma = sn;
    s2 = s2n;
}

__device__ __forceinline__ void logsv_step_fast(const LogsvFast &f, double &x, double &L, double &sigma, double &s2,
                                                double &qvar, double z0, double z1)
{
    logsv_step_fast(f, x, L, sigma, s2, qvar, z0, z1, [](double v) { return exp_fast(v); });
}

// The same step with the two running sums of sigma^2 (the drift of x and the quadratic variance) replaced by ONE
// accumulator acc = sum_{t=1..T} sigma_t^2; the caller folds it in where x and qvar are needed (logsv_fold_acc):
//     x_T    = x_0 + B xacc + ahA (acc + sigma_0^2 - sigma_T^2),  xacc = sum_t sigma_t z0_t
//     qvar_T = qvar_0 + hA (2 acc + sigma_0^2 - sigma_T^2)                          [= hA sum (sigma_t^2 + sigma_{t+1}^2)]
// and L is advanced by single FMAs (no constant has to be moved into a vector register): 9 arithmetic instructions
// around the exp and the reciprocal instead of 12.  Identical in exact arithmetic; rounding differs at 1e-16.
template <class Exp>
__device__ __forceinline__ void logsv_step_acc(const LogsvFast &f, double &xacc, double &L, double &sigma, double &s2,
                                               double &acc, double z0, double z1, Exp &&exp_of)
{
    const double s = sigma;
    const double y = rcp_1n(s);
    xacc = fma(s, z0, xacc);                      // sum sigma_t z0_t; the factor B = eta sqrt(dt) is applied in the fold
    L = fma(f.c2, s, L);
    L = fma(f.c1, y, L);
    L = L + f.c3;
    L = fma(f.bs, z0, L);
    L = fma(f.es, z1, L);
    const double sn = exp_of(L);
    acc = fma(sn, sn, acc);
    sigma = sn;
    (void)s2;                                     // sigma^2 is not carried: the fold squares the terminal sigma itself
}

// logsv_step_acc in two halves around its exp-table read (rng_time_loop_pipelined puts state-independent work between them):
// front = everything up to the ISSUE of the table read, back = what consumes the value.  logsv_step_acc's operations with
// exp2u_tab as the exponential, in its order: the same bits.
struct LogsvStepInFlight {
    double r, t, p;
    int ni;
};
__device__ __forceinline__ void logsv_step_acc_front(const LogsvFast &f, double &xacc, double &L, double sigma, double z0, double z1,
                                                     const double *exp_table, LogsvStepInFlight &h)
{
    const double y = rcp_1n(sigma);
    xacc = fma(sigma, z0, xacc);
    L = fma(f.c2, sigma, L);
    L = fma(f.c1, y, L);
    L = L + f.c3;
    L = fma(f.bs, z0, L);
    L = fma(f.es, z1, L);
    exp2u_reduce(L, h.ni, h.r);
#if defined(SVMC_PROBE) && (SVMC_PROBE & 1)          // measurement build: no exp-table read on the step's chain
    h.t = 1.0;
#else
    h.t = exp_table[h.ni & 255];
#endif
}
// the exponential's polynomial tail: needs the reduced argument only, so it runs in the loop's middle region, among the draw's
// integer work, instead of as a dependent chain (with its hazard s_nop's) right before the table value is consumed
__device__ __forceinline__ void logsv_step_acc_mid(LogsvStepInFlight &h, const Exp2uTailV &k)
{
    h.p = exp2u_tail_v(h.r, k);
}
__device__ __forceinline__ void logsv_step_acc_back(double &sigma, double &acc, const LogsvStepInFlight &h)
{
    const double sn = exp2u_scale(h.t, h.p, h.ni);
    acc = fma(sn, sn, acc);
    sigma = sn;
}

// logsv_step_acc for P independent states of one lane (P parameter sets on the same two normals), piece by piece ACROSS the
// states: the reciprocals, the five updates of L, the exp's reduction, all P table reads, the tails, the scalings.  Per
// state these are logsv_step_acc's operations in logsv_step_acc's order -- the same bits -- but the P dependent chains
// (about twenty fp64 operations around an LDS round trip each) now overlap instead of running one after the other.
template <int P>
__device__ __forceinline__ void logsv_step_acc_sets(const double (&c1)[P], const double (&c2)[P], const double (&c3)[P],
                                                    const double (&bs)[P], const double (&es)[P], double (&xacc)[P],
                                                    double (&L)[P], double (&sigma)[P], double (&acc)[P], double z0, double z1,
                                                    const double *exp_table)
{
    double y[P], r[P], t[P];
    int ni[P];
#pragma unroll
    for (int s = 0; s < P; ++s) y[s] = rcp_1n(sigma[s]);
#pragma unroll
    for (int s = 0; s < P; ++s) xacc[s] = fma(sigma[s], z0, xacc[s]);
#pragma unroll
    for (int s = 0; s < P; ++s) {
        double l = fma(c2[s], sigma[s], L[s]);
        l = fma(c1[s], y[s], l);
        l = l + c3[s];
        l = fma(bs[s], z0, l);
        L[s] = fma(es[s], z1, l);
    }
#pragma unroll
    for (int s = 0; s < P; ++s) exp2u_reduce(L[s], ni[s], r[s]);
#pragma unroll
    for (int s = 0; s < P; ++s) t[s] = exp_table[ni[s] & 255];
#pragma unroll
    for (int s = 0; s < P; ++s) r[s] = exp2u_tail(r[s]);
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const double sn = exp2u_scale(t[s], r[s], ni[s]);
        acc[s] = fma(sn, sn, acc[s]);
        sigma[s] = sn;
    }
}

// logsv_step_acc_sets in two halves around the P exp-table reads (rng_time_loop_pipelined): front = everything up to the ISSUE
// of the reads, back = what consumes them.  Per state logsv_step_acc_sets' operations in its order: the same bits.
template <int P>
struct LogsvSetsInFlight {
    double r[P], t[P];
    int ni[P];
};
template <int P>
__device__ __forceinline__ void logsv_step_acc_sets_front(const double (&c1)[P], const double (&c2)[P], const double (&c3)[P],
                                                          const double (&bs)[P], const double (&es)[P], double (&xacc)[P],
                                                          double (&L)[P], const double (&sigma)[P], double z0, double z1,
                                                          const double *exp_table, LogsvSetsInFlight<P> &h)
{
    double y[P];
#pragma unroll
    for (int s = 0; s < P; ++s) y[s] = rcp_1n(sigma[s]);
#pragma unroll
    for (int s = 0; s < P; ++s) xacc[s] = fma(sigma[s], z0, xacc[s]);
#pragma unroll
    for (int s = 0; s < P; ++s) {
        double l = fma(c2[s], sigma[s], L[s]);
        l = fma(c1[s], y[s], l);
        l = l + c3[s];
        l = fma(bs[s], z0, l);
        L[s] = fma(es[s], z1, l);
    }
#pragma unroll
    for (int s = 0; s < P; ++s) exp2u_reduce(L[s], h.ni[s], h.r[s]);
#pragma unroll
    for (int s = 0; s < P; ++s) h.t[s] = exp_table[h.ni[s] & 255];
}
// (the tails in the loop's middle region, on coefficients in vector registers: see logsv_step_acc_mid)
template <int P>
__device__ __forceinline__ void logsv_step_acc_sets_mid(LogsvSetsInFlight<P> &h, const Exp2uTailV &k)
{
#pragma unroll
    for (int s = 0; s < P; ++s) h.r[s] = exp2u_tail_v(h.r[s], k);
}
template <int P, bool TAILS_DONE = false>
__device__ __forceinline__ void logsv_step_acc_sets_back(double (&sigma)[P], double (&acc)[P], LogsvSetsInFlight<P> &h)
{
    if constexpr (!TAILS_DONE) {
#pragma unroll
        for (int s = 0; s < P; ++s) h.r[s] = exp2u_tail(h.r[s]);
    }
#pragma unroll
    for (int s = 0; s < P; ++s) {
        const double sn = exp2u_scale(h.t[s], h.r[s], h.ni[s]);
        acc[s] = fma(sn, sn, acc[s]);
        sigma[s] = sn;
    }
}

// sigma^2 as a rounded product of its own: without this the compiler may fuse the multiplication into the subtraction of
// logsv_fold_acc (s2_start - sigma_T^2 as one FMA) in one kernel and not in another -- whichever way the inlined code around
// it falls -- and the one-slice, whole-chain and streamed generators must agree to the bit (a persistent-launch variant of
// the generator differed from the one-round kernel in 180 of 2^20 paths by exactly this, profiles/r03_launch_tail.txt)
__device__ __forceinline__ double square_rn(double v)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    return v * v;
}

// fold the accumulator of logsv_step_acc into x and qvar (s2_start = sigma^2 when acc was last zero); a path whose
// sigma overflowed keeps the reference's outcome (x -> -+inf, qvar -> inf) instead of inf - inf
__device__ __forceinline__ void logsv_fold_acc(const LogsvFast &f, double &x, double &qvar, double xacc, double acc,
                                               double s2_start, double s2_now)
{
    x = fma(f.B, xacc, x);
    const double tail = s2_start - s2_now;
    const bool finite = acc < __builtin_huge_val();
    const double d = finite ? acc + tail : acc;
    const double e = finite ? (acc + acc) + tail : acc;
    x = fma(f.ahA, d, x);
    qvar = fma(f.hA, e, qvar);
}

// ---- Heston Euler with the reference's floor: pricers/heston_pricer.py:372-379 ---------------------
struct HestonConsts {
    double dt, sdt, theta, kappa, rho, rho_1, volvol;
};

inline HestonConsts make_heston_consts(double dt, double theta, double kappa, double rho, double volvol)
{
    HestonConsts c;
    c.dt = dt;
    c.sdt = sqrt(dt);
    c.theta = theta;
    c.kappa = kappa;
    c.rho = rho;
    c.rho_1 = sqrt(1.0 - rho * rho);
    c.volvol = volvol;
    return c;
}

__device__ __forceinline__ void heston_euler_step(const HestonConsts &c, double &x, double &var, double &qvar,
                                                  double w0, double w1)
{
    const double v = var;
    const double s = sqrt_pos0(v);                           // v >= 1e-4 after the first floor    :374
    const double s2dt = v * c.dt;                                                               // :375
    x = (x - 0.5 * s2dt) + s * w0;                                                              // :376
    qvar = qvar + s2dt;                                                                         // :377
    const double vn = (v + (c.kappa * (c.theta - v)) * c.dt) + (s * c.volvol) * (c.rho * w0 + c.rho_1 * w1);
    var = (vn > 1e-4) ? vn : ((vn != vn) ? vn : 1e-4);                  // np.maximum(v, 1e-4)     :379
}

// The same Euler step for the issue-bound on-device-RNG kernels: the drift of x and the quadratic variance are both
// dt * sum v_t, so ONE accumulator serves both (folded in by heston_fold_acc), x accumulates sum sqrt(v_t) z0_t with
// the factor sqrt(dt) applied at the fold, and the variance update uses host-premultiplied constants.  z0, z1 are
// UNSCALED N(0,1).  Identical in exact arithmetic to heston_euler_step, floor included.
struct HestonEulerFast {
    double dt, sdt, one_m_kdt, ktdt, a0, a1;   // 1 - kappa dt, kappa theta dt, volvol sqrt(dt) rho, volvol sqrt(dt) rho_1
};

__host__ __device__ inline HestonEulerFast make_heston_euler_fast(const HestonConsts &c)
{
    HestonEulerFast f;
    f.dt = c.dt;
    f.sdt = c.sdt;
    f.one_m_kdt = 1.0 - c.kappa * c.dt;
    f.ktdt = c.kappa * c.theta * c.dt;
    f.a0 = c.volvol * c.sdt * c.rho;
    f.a1 = c.volvol * c.sdt * c.rho_1;
    return f;
}

__device__ __forceinline__ void heston_euler_step_acc(const HestonEulerFast &f, double &xacc, double &var, double &vacc,
                                                      double z0, double z1)
{
    const double v = var;
    const double s = sqrt_pos_1g(v);                                    // v > 0: heston_euler_guard_zero + the floor
    vacc = vacc + v;
    xacc = fma(s, z0, xacc);
    const double m = fma(f.a1, z1, f.a0 * z0);
    const double vn = fma(s, m, fma(v, f.one_m_kdt, f.ktdt));
    // np.maximum(v, 1e-4) :379 as ONE v_max_f64 (the compare / NaN test / two selects of the literal form are four of the
    // step's 66 instructions).  v_max drops a NaN where np.maximum keeps it: heston_fold_acc puts it back.
    var = fmax(vn, 1e-4);
}

// Only the incoming variance of a slice can be exactly zero (every later one is floored at 1e-4), and sqrt_pos_1g does
// not take zero: 2^-1000 in its place gives sqrt = 2^-500, which the accumulators and the next variance absorb without
// a trace (x, v, qvar come out as with an exact zero; a one-step slice keeps dt 2^-1000 in qvar).
__device__ __forceinline__ double heston_euler_guard_zero(double var)
{
    return (var == 0.0) ? 0x1.0p-1000 : var;
}

// `var` is the variance the steps ended with.  A NaN variance can only come from a NaN / infinite incoming variance (then
// vacc, which summed it, is not finite either) or from NaN / infinite constants: either way the reference's variance is
// NaN from there on (np.maximum propagates it), and so is ours after this.
__device__ __forceinline__ void heston_fold_acc(const HestonEulerFast &f, double &x, double &qvar, double &var, double xacc,
                                                double vacc)
{
    const double q = f.dt * vacc;
    x = fma(f.sdt, xacc, fma(-0.5, q, x));
    qvar = qvar + q;
    const double finite_or_nan = ((f.one_m_kdt * 0.0 + f.ktdt * 0.0) + (f.a0 * 0.0 + f.a1 * 0.0)) + vacc * 0.0;   // 0 or NaN
    var = var + finite_or_nan;
}

// ---- Andersen QE-M (J. Comp. Fin. 11(3), 2008); CPU twin: oracle/svmc_oracle.c heston_qe_step -------
struct QeConsts {
    double dt, theta, E, c1, c2, K1m, K2, K3, K4, A, twoA, K0_plain2, K13_2, m0;     // ..2: twice the constant
    double c1h, c2h;                                                                 // c1 / 2, c2 / 2: s2 / 2 = v0 c1h + c2h
    // what the parameters alone decide (wave-uniform, scalar branches in the step):
    int quad_only;        // psi = s2/m^2 is largest at v0 = 0, where it is volvol^2 / (2 kappa theta): at most 3/2 (less a margin) ->
                          // no path ever takes the exponential branch and the step does not test for it
    int e_below_one;      // A <= 0 (rho <= 0 on any sane grid): 1 - 2 A a >= 1, the martingale correction always exists
};

// The two constants that multiply the variance, held in VECTOR registers: an instruction takes one scalar-register operand
// on this chip, so fma(v0, E, m0) and fma(v0, c1, c2) with all four in scalar registers each cost a v_mov on top
struct QeVec {
    double E, c1h;
};
__device__ __forceinline__ QeVec make_qe_vec(const QeConsts &c)
{
    QeVec v = {c.E, c.c1h};
    asm volatile("" : "+v"(v.E), "+v"(v.c1h));
    return v;
}

inline QeConsts make_qe_consts(double dt, double theta, double kappa, double rho, double volvol)
{
    QeConsts c;
    const double g1 = 0.5, g2 = 0.5;
    const double E = exp(-kappa * dt);
    const double kre = kappa * rho / volvol - 0.5;
    c.dt = dt;
    c.theta = theta;
    c.E = E;
    c.c1 = volvol * volvol * E * (1.0 - E) / kappa;
    c.c2 = theta * volvol * volvol * (1.0 - E) * (1.0 - E) / (2.0 * kappa);
    const double K1 = g1 * dt * kre - rho / volvol;
    c.K2 = g2 * dt * kre + rho / volvol;
    c.K3 = g1 * dt * (1.0 - rho * rho);
    c.K4 = g2 * dt * (1.0 - rho * rho);
    c.A = c.K2 + 0.5 * c.K4;
    c.twoA = 2.0 * c.A;
    const double K13 = K1 + 0.5 * c.K3;
    c.K0_plain2 = 2.0 * (-rho * kappa * theta / volvol * dt);
    c.K13_2 = 2.0 * K13;
    c.m0 = theta * (1.0 - E);                             // E[v1 | v0] = v0 E + theta (1 - E)
    c.K1m = K1 - K13;                                     // the martingale correction's -K13 v0 rides on the K1 v0 term
    c.c1h = 0.5 * c.c1;
    c.c2h = 0.5 * c.c2;
    // d psi / d v0 = [c1 m0 - 2 E c2 - E c1 v0] / m^3 and c1 m0 = 2 E c2: psi falls from psi(0) = c2 / m0^2 = volvol^2 / (2 kappa theta)
    c.quad_only = (kappa > 0.0 && theta > 0.0 && volvol * volvol <= 3.0 * kappa * theta * (1.0 - 1e-6)) ? 1 : 0;
    c.e_below_one = (c.A <= 0.0) ? 1 : 0;
    return c;
}

// ln(1 - e) for the martingale correction ln(1 - 2 A a): |e| is O(volvol^2 dt) on any sane grid.  Below 2^-10 (the C3 base
// set sits at 2^-11.3) five series terms do (next: e^6 / 6 <= 1.4e-19); below 2^-6 eight (next: e^9 / 9 <= 6e-18 at the
// switch); larger |e| take the table logarithm.  The choice is made PER LANE (a wave whose lanes all take one branch skips
// the others): a path's rounding must not depend on which other paths share its wave -- results are bit-independent of how
// a job is sharded and where a path sits in a launch.
// inv_out = 1 / (1 - e), which the correction needs beside the logarithm: below 2^-10 the geometric series to e^5 (next:
// e^6 <= 8.7e-19) -- five FMAs where the hardware reciprocal and its Newton step cost the issue slots of seven -- above,
// rcp_1n
__device__ __forceinline__ double log_one_minus(double e, const LogTabEntry *tab, double &inv_out)
{
    const double ae = fabs(e);
    if (ae < 0x1.0p-10) {
        double p = 0x1.999999999999ap-3;                  // 1/5
        p = fma_k(p, e, 0x1.0000000000000p-2);            // 1/4
        p = fma_k(p, e, 0x1.5555555555555p-2);            // 1/3
        p = fma_k(p, e, 0x1.0000000000000p-1);            // 1/2
        p = fma_k(p, e, 1.0);
        double q = e + 1.0;
        q = fma_k(q, e, 1.0);
        q = fma_k(q, e, 1.0);
        q = fma_k(q, e, 1.0);
        inv_out = fma_k(q, e, 1.0);
        return -e * p;
    }
    const double one_minus_e = 1.0 - e;
    inv_out = rcp_1n(one_minus_e);
    if (ae < 0x1.0p-6) {
        double p = 0x1.0000000000000p-3;                  // 1/8
        p = fma_k(p, e, 0x1.2492492492492p-3);            // 1/7
        p = fma_k(p, e, 0x1.5555555555555p-3);            // 1/6
        p = fma_k(p, e, 0x1.999999999999ap-3);            // 1/5
        p = fma_k(p, e, 0x1.0000000000000p-2);            // 1/4
        p = fma_k(p, e, 0x1.5555555555555p-2);            // 1/3
        p = fma_k(p, e, 0x1.0000000000000p-1);            // 1/2
        p = fma_k(p, e, 1.0);
        return -e * p;
    }
    return -neg_log_tab(one_minus_e, tab);
}

// The martingale correction 2 K0* = ln(1 - e) - 2 A alpha / (1 - e), e = 2 A a = 2 A (m - alpha), as ONE polynomial where |e| is
// small (round 6).  With M = 2 A m:  2 A alpha = M - e, so
//     2 K0* = -M + ln(1 - e) + (1 - M) e / (1 - e) = -M - M e + sum_{k >= 2} (k - 1)/k - M) e^k ... i.e. coefficients
//     c_0 = c_1 = -M,  c_k = (k - 1)/k - M,
// to e^5 below 2^-10 (the next term, (5/6 - M) e^6, is under 9e-19): one product for M, four subtractions and five FMAs where
// the two separate series (logarithm and reciprocal), the product 2 A alpha and the final FMA took sixteen instructions.  Larger
// |e| keep the round-5 forms: log_one_minus (eight-term series to 2^-6, the table logarithm above) and the hardware reciprocal.
// The regime is chosen PER LANE: a path's rounding does not depend on which other paths share its wave.
__device__ __forceinline__ double qe_martingale_kd(double e, double al, double m, double twoA, const LogTabEntry *tab)
{
    if (fabs(e) < 0x1.0p-10) {
        const double M = twoA * m;
        double p = 0x1.999999999999ap-1 - M;              // 4/5 - M
        p = fma(p, e, 0.75 - M);
        p = fma(p, e, 0x1.5555555555555p-1 - M);          // 2/3 - M
        p = fma(p, e, 0.5 - M);
        p = fma(p, e, -M);
        return fma(p, e, -M);
    }
    double inv;
    const double ln_den = log_one_minus(e, tab, inv);
    return fma(-(twoA * al), inv, ln_den);                // ln(1 - 2 A a) - 2 A b^2 a / (1 - 2 A a)
}

// One QE-M step.  z0 drives the log-price, z1 the quadratic branch; draw_u() hands over the uniform of the exponential
// branch -- called by the whole wave as soon as one lane is there (the streamed kernel loads it; the on-device draw is a
// lazily evaluated Philox call shared by four steps, svmc_rng.h qe_uniform).
//   quadratic (psi = s2/m^2 <= 3/2).  With alpha = sqrt(m^2 - s2/2) (in [m/2, m]) the textbook quantities are
//     b^2 = 2 alpha (alpha + m) / s2,   1 + b^2 = 2 m (alpha + m) / s2,   a = m / (1 + b^2) = m - alpha,   b^2 a = alpha,
//     v1 = a (b + z1)^2 = (sqrt(alpha) + sqrt(m - alpha) z1)^2 = alpha + 2 sqrt(alpha (m - alpha)) z1 + (m - alpha) z1^2:
//     two square roots and NO quotient for the new variance; the martingale correction
//     2 K0* = ln(1 - e) - 2 A alpha / (1 - e),  e = 2 A a = 2 A (m - alpha),  takes 1 / (1 - e) -- a five-term series
//     wherever the logarithm's is (|e| < 2^-10: every sane grid), the hardware reciprocal elsewhere.  (Round 4 formed
//     N = s2 b^2, T = s2 (1 + b^2), sqrt(2 m^2 w), sqrt(N s2) and 1/(T dn): ten instructions and a reciprocal more per
//     step.)  m - alpha cancels for small psi; a residual step on (m - alpha)(m + alpha) = s2/2 repairs it (below).
//   exponential, with D = s2 + m^2:  p = (s2 - m^2)/D,  1 - p = 2 m^2/D,  beta = 2 m/D,  u <= p  <=>  u D <= s2 - m^2,
//     v1 = ln(2 m^2 / (D (1 - u))) D / (2 m);   1/(D (1 - u)) and 1/m from rcp(D (1 - u) m);
//     p + beta (1 - p)/(beta - A) = ((s2 - m^2) e + 4 m m^2) / (D e),  e = 2 m - A D;   1/(D e) is the second reciprocal.
// Identical in exact arithmetic to the CPU twin (oracle/svmc_oracle.c, the textbook form); reciprocals are seed + one Newton
// step (2^-48), square roots stop after the Goldschmidt step (2^-47: the scheme matches two moments of the variance, not its
// bits), logs go through the LDS table.  x advances by sqrt(K3 v0 + K4 v1) z0 alone; the drift terms are SUMS over the
// steps -- K1m sum v0 + K2 sum v1 + sum K0* -- and are folded in once (heston_qe_fold: `vsum` = sum v1, `ksum` = sum 2 K0*,
// sum v0 = v_first + vsum - v_last), as is the quadratic variance dt (vsum + (v_first - v_last)/2).
// QUAD (compile time): the parameters rule the exponential branch out (c.quad_only) AND keep the martingale correction's
// argument below one (c.e_below_one) -- both C3 sets, every Feller-satisfying set with rho <= 0: a kernel instantiated so carries
// neither branch, their registers, nor the uniform's Philox state.  The statements that remain are the general form's, so are
// the bits.
template <bool QUAD = false, class DrawU>
__device__ __forceinline__ void heston_qe_step(const QeConsts &c, const QeVec &cv, const LogTabEntry *tab, double &x,
                                               double &var, double &vsum, double &ksum, double z0, double z1, DrawU &&draw_u)
{
    const double v0 = var;
    const double m = fma(v0, cv.E, c.m0);
    const double s2h = fma(v0, cv.c1h, c.c2h);            // s2 / 2
    const double m2 = m * m;
    const double w = m2 - s2h;                            // alpha^2
    double v1, Kd;                                        // Kd = 2 K0, K0 without its -K13 v0 term (that rides in K1m)
    bool quad = true;
    double u = m;                                         // any defined value: only lanes past the test below read it
    if constexpr (!QUAD) {
        if (!c.quad_only) {                               // (wave-uniform)
            quad = w >= 0.25 * m2;                        // psi <= psi_c = 3/2 (s2 <= 3/2 m^2), decided without the divide
            // the uniform of the exponential branch is fetched by the WHOLE wave as soon as one of its lanes needs it (the
            // on-device draw is a Philox call that serves four steps: every lane must take part in it)
            if (!__all(quad)) u = draw_u();
        }
    }
    if (QUAD || quad) {
        double h;                                         // ~ 1 / (2 alpha)
        const double al = sqrt_pos_1g_h(w, h);            // alpha to 2^-47
        // a = m - alpha cancels: alpha's 2^-47 would reach it multiplied by alpha / a = 4 / psi (3e-12 at C3's psi ~ 0.01).
        // One residual step on a (m + alpha) = s2 / 2 with the crude 1 / (m + alpha) ~ h (off by psi / 8) takes the error
        // back to eps wherever it mattered.  (volvol = 0: a is 0 up to rounding, of either sign -- |a| + 1e-300 below)
        const double a0 = m - al;
        const double a = fma(fma(-a0, m + al, s2h), h, a0);
        const double ga = sqrt_pos_1g(fma(al, fabs(a), 1e-300));
        const double t = fma(z1, fma(a, z1, ga + ga), al);        // (sqrt(alpha) + sqrt(a) z1)^2 up to rounding: clamped at 0
        v1 = fmax(t, 0.0);
        if (c.A == 0.0) {                                 // wave-uniform: rho = 0 makes the martingale factor 1
            Kd = 0.0;
        } else {
            const double e = c.twoA * a;                  // 2 A a
            if (QUAD || c.e_below_one || e < 1.0) {
                Kd = qe_martingale_kd(e, al, m, c.twoA, tab);
            } else {
                Kd = fma(c.K13_2, v0, c.K0_plain2);       // the plain drift: cancels the folded -K13 v0
            }
        }
    } else {
        const double s2 = s2h + s2h;
        const double D = s2 + m2, dm = s2 - m2;
        const bool zero = (u * D <= dm);                  // u <= p
        const double q1 = D * (1.0 - u);
        const double r = rcp_1n(q1 * m);
        const double iq1 = m * r, im = q1 * r;            // 1/(D (1 - u)),  1/m
        const double lg = neg_log_tab((m2 + m2) * iq1, tab);   // -ln((1 - p)/(1 - u)): <= 0 wherever u > p
        v1 = zero ? 0.0 : (-lg) * ((0.5 * D) * im);
        if (c.A == 0.0) {
            Kd = 0.0;
        } else {
            const double e = fma(-c.A, D, m + m);         // D (beta - A)
            if (e > 0.0) {
                const double r2 = rcp_1n(D * e);
                Kd = 2.0 * neg_log_tab(fma(dm, e, 4.0 * m * m2) * r2, tab);
            } else {
                Kd = fma(c.K13_2, v0, c.K0_plain2);
            }
        }
    }
    // + 1e-300: v0 = v1 = 0 (two exponential-branch zeros in a row) must not reach the rsq seed; sqrt(1e-300) z0 is nothing
    const double sq = sqrt_pos_1g(fma(c.K4, v1, fma(c.K3, v0, 1e-300)));
    x = fma(sq, z0, x);
    ksum = ksum + Kd;
    vsum = vsum + v1;
    var = v1;
}

// qvar += dt * sum_t (v_{t-1} + v_t)/2 over the steps since vsum was zero (v_first = the variance they started from)
// and x += K1m sum v0 + K2 sum v1 + sum K0* (see heston_qe_step)
__device__ __forceinline__ void heston_qe_fold(const QeConsts &c, double &x, double &qvar, double vsum, double ksum, double v_first,
                                               double v_last)
{
    const double dv = v_first - v_last;
    x = fma(c.K1m, vsum + dv, fma(c.K2, vsum, fma(0.5, ksum, x)));
    qvar = fma(c.dt, fma(0.5, dv, vsum), qvar);
}

}  // namespace svmc
