// svmc_analytic.hip -- the analytic (Fourier) side of the chain pricers on gfx950: SURVEY.md row a11 / config C5.
//
//   logsv_mgf_grid_kernel     one 16-lane DPP row per transform-grid point Phi_j, one COMPONENT per lane: the 5-dim
//                             complex quadratic ODE A' = A^T M A + L A + H of the affine expansion (pricers/logsv/
//                             affine_expansion.py:67-205, 229-303, 570-685; svmc_ode.h), integrated from the previous
//                             expiry's A with the Dormand-Prince 8(5,3) pair (DOP853; svmc_dop853.h) and per-point step control;
//                             log E = sum_k A_k (sigma0-theta)^k
//   heston_mgf_grid_kernel    closed-form Heston MGF (pricers/heston_pricer.py:183-214)
//   mgf_vanilla_slice_kernel  one block per strike: Simpson-weighted sum over the grid of
//                             Re[ w_j / (pi (p_j^2 + 1/4)) exp(-x_K Phi_j + log E_j) ]   (utils/mgf_pricer.py:174-221)
//
// The reference runs a Python loop of 1000 scipy.solve_ivp calls per expiry (~4 s); here every grid point is a
// row of lanes and an expiry is one launch.  The work is tiny (1000 points) and latency-bound; it is on the GPU so that the
// analytic-vs-MC sweep of config C5 needs no host ODE solver.  CPU twin: oracle/svmc_oracle_analytic.c.
#include "svmc_internal.h"

#include <cstdlib>

#include "svmc_math.h"
#include "svmc_ode.h"
#include "svmc_dop853.h"

#ifndef SVMC_ODE_DOP853
#define SVMC_ODE_DOP853 1              // A/B hook: 0 = the Dormand-Prince 5(4) pair of rounds 1-3
#endif

namespace svmc {

__device__ __forceinline__ double cabs_(cd a) { return hypot(a.re, a.im); }
__device__ __forceinline__ cd operator/(cd a, cd b)
{
    const double d = b.re * b.re + b.im * b.im;
    return cd{(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
__device__ __forceinline__ cd cexp_(cd z)
{
    double s, c;
    sincos(z.im, &s, &c);
    const double e = exp(z.re);
    return cd{e * c, e * s};
}
__device__ __forceinline__ cd csqrt_(cd z)   // principal branch
{
    const double r = cabs_(z);
    if (r == 0.0) return cd{0.0, 0.0};
    const double t = sqrt(0.5 * (r + fabs(z.re)));
    if (z.re >= 0.0) return cd{t, z.im / (2.0 * t)};
    return cd{fabs(z.im) / (2.0 * t), copysign(t, z.im)};
}
__device__ __forceinline__ cd clog_(cd z) { return cd{log(cabs_(z)), atan2(z.im, z.re)}; }

// ---- one grid point per LANE: the form for LONG grids ----------------------------------------------------------------
// The row form below wins on latency (a lone wave per SIMD, a third of the instructions per step) and loses on volume: it
// spends 16 lanes per point.  The 40 000-point psi grid of the quadratic-variance transform fills the chip either way --
// 625 waves one lane per point, 10 000 waves in rows -- and is then bound by total instruction issue, where a point costs
// 1850 instructions per step here against 16 x 800 / 5 useful in rows.  svmc_logsv_mgf_grid_batch picks by grid length.
// A' = A^T M^(k) A + L^(k) A + H^(k): the non-zero entries of :146-182 written out
__device__ __forceinline__ void ode_rhs(const OdeConsts &c, cd phi, cd psi, const cd (&A)[5], cd (&out)[5])
{
    const double qv = c.qv, qv2 = c.qv2, v2 = c.vartheta2, th = c.theta, th2 = c.theta2;
    const cd bphi = c.b * phi;
    const cd A1 = A[1], A2 = A[2];
    const cd rhs = (c.spot ? phi * (phi + 1.0) : phi * (phi - 1.0)) - 2.0 * psi;
    const cd L01 = c.lamda - th2 * bphi;
    const cd L11 = -c.kappa_p - 2.0 * th * bphi, L12 = 2.0 * ((c.lamda + qv) - th2 * bphi);
    const cd L21 = -c.kappa2_p - bphi, L22 = (v2 - 2.0 * c.kappa_p) - 4.0 * th * bphi;
    const cd A11 = A1 * A1, A12 = A1 * A2, A22 = A2 * A2;
    out[0] = 0.5 * qv2 * A11 + L01 * A1 + qv2 * A2 + 0.5 * th2 * c.eta2 * rhs;
    out[1] = qv * A11 + 2.0 * qv2 * A12 + L11 * A1 + L12 * A2 + th * c.eta2 * rhs;
    out[2] = 0.5 * v2 * A11 + 2.0 * qv2 * A22 + 4.0 * qv * A12 + L21 * A1 + L22 * A2 + 0.5 * c.eta2 * rhs;
    if (c.second) {
        const cd A3 = A[3], A4 = A[4];
        const cd kb = c.kappa2_p + bphi;
        const cd L23 = 3.0 * (2.0 * qv - th2 * bphi);
        const cd L33 = 3.0 * ((v2 - c.kappa_p) - 2.0 * th * bphi), L34 = 4.0 * (3.0 * qv - th2 * bphi);
        const cd A13 = A1 * A3, A14 = A1 * A4, A23 = A2 * A3, A24 = A2 * A4;
        out[1] = out[1] + 3.0 * qv2 * A3;
        out[2] = out[2] + 3.0 * qv2 * A13 + L23 * A3 + 6.0 * qv2 * A4;
        out[3] = 4.0 * qv * A22 + 2.0 * v2 * A12 + 6.0 * qv * A13 + 4.0 * qv2 * A14 + 6.0 * qv2 * A23 - 2.0 * (kb * A2) +
                 L33 * A3 + L34 * A4;
        out[4] = 2.0 * v2 * A22 + 4.5 * qv2 * (A3 * A3) + 3.0 * v2 * A13 + 8.0 * qv * A14 + 12.0 * qv * A23 + 8.0 * qv2 * A24 -
                 3.0 * (kb * A3) + 2.0 * (L22 * A4);
    } else {
        out[3] = C(0.0);
        out[4] = C(0.0);
    }
}

// A step controller that keeps rejecting -- a coefficient system that blows up before the expiry for a parameter vector a
// calibrator wandered into -- shrinks h by 0.2 a try; below this fraction of the interval (SciPy gives up at 10 ulp of t with
// "required step size is less than spacing between numbers" and hands back the state it reached) the grid point is given up
// and its state set to NaN -- the inversion drops a NaN term, as the reference's np.nansum does (utils/mgf_pricer.py:205) --
// instead of the kernel grinding through a 10^6-try cap (seconds) and inverting whatever half-integrated state it reached.
// The clip of the LAST step to the expiry comes after this test: a legitimately tiny final remainder never trips it.
constexpr double ODE_STEP_FLOOR = 0x1.0p-46;               // 1.4e-14 of the interval: 20 rejections in a row from ttm / 8
// ... and a system that is merely STIFF beyond reason (vol-of-vol of 2000 %: the explicit pair's stability bound, not its
// accuracy, sets the step) would take its ~10^5-10^6 steps at ~4.5 us each -- 4.5 s a launch at the old 10^6 cap, after which
// the state reached so far was inverted into finite garbage (tools/r04/analytic_blowup_probe.py).  The coefficient systems of
// every parameter set of the test suite take under 128 tries (the suite passes with the cap there: -DSVMC_ODE_MAX_TRIES=128);
// at 2^15 the point is given up the same way.
#ifndef SVMC_ODE_MAX_TRIES
#define SVMC_ODE_MAX_TRIES (1 << 15)
#endif
constexpr int ODE_MAX_TRIES = SVMC_ODE_MAX_TRIES;

// Dormand-Prince 5(4), FSAL, mixed error scale, RMS norm -- the CPU twin's dopri5() (same tableau, same controller; the
// error norm and the step factor are evaluated as noted below)
__device__ void dopri5(const OdeConsts &c, cd phi, cd psi, double ttm, cd (&y)[5], double rtol, double atol)
{
    constexpr double a21 = 1.0 / 5, a31 = 3.0 / 40, a32 = 9.0 / 40, a41 = 44.0 / 45, a42 = -56.0 / 15, a43 = 32.0 / 9,
                     a51 = 19372.0 / 6561, a52 = -25360.0 / 2187, a53 = 64448.0 / 6561, a54 = -212.0 / 729,
                     a61 = 9017.0 / 3168, a62 = -355.0 / 33, a63 = 46732.0 / 5247, a64 = 49.0 / 176,
                     a65 = -5103.0 / 18656, b1 = 35.0 / 384, b3 = 500.0 / 1113, b4 = 125.0 / 192, b5 = -2187.0 / 6784,
                     b6 = 11.0 / 84, e1 = 71.0 / 57600, e3 = -71.0 / 16695, e4 = 71.0 / 1920, e5 = -17253.0 / 339200,
                     e6 = 22.0 / 525, e7 = -1.0 / 40;
    cd k1[5], k2[5], k3[5], k4[5], k5[5], k6[5], k7[5], yt[5], yn[5];
    double t = 0.0, h = ttm / 32.0;
    int tries = 0;
    ode_rhs(c, phi, psi, y, k1);
    while (t < ttm && tries < ODE_MAX_TRIES) {
        ++tries;
        if (!(h >= ODE_STEP_FLOOR * ttm)) break;          // the controller's step collapsed (or went NaN): given up below
        if (t + h > ttm) h = ttm - t;
#pragma unroll
        for (int i = 0; i < 5; ++i) yt[i] = y[i] + h * (a21 * k1[i]);
        ode_rhs(c, phi, psi, yt, k2);
#pragma unroll
        for (int i = 0; i < 5; ++i) yt[i] = y[i] + h * (a31 * k1[i] + a32 * k2[i]);
        ode_rhs(c, phi, psi, yt, k3);
#pragma unroll
        for (int i = 0; i < 5; ++i) yt[i] = y[i] + h * (a41 * k1[i] + a42 * k2[i] + a43 * k3[i]);
        ode_rhs(c, phi, psi, yt, k4);
#pragma unroll
        for (int i = 0; i < 5; ++i) yt[i] = y[i] + h * (a51 * k1[i] + a52 * k2[i] + a53 * k3[i] + a54 * k4[i]);
        ode_rhs(c, phi, psi, yt, k5);
#pragma unroll
        for (int i = 0; i < 5; ++i)
            yt[i] = y[i] + h * (a61 * k1[i] + a62 * k2[i] + a63 * k3[i] + a64 * k4[i] + a65 * k5[i]);
        ode_rhs(c, phi, psi, yt, k6);
#pragma unroll
        for (int i = 0; i < 5; ++i) yn[i] = y[i] + h * (b1 * k1[i] + b3 * k3[i] + b4 * k4[i] + b5 * k5[i] + b6 * k6[i]);
        ode_rhs(c, phi, psi, yn, k7);
        // the error norm of the twin, err = sqrt(mean_i (|e_i| / sc_i)^2) with sc_i = atol + rtol max(|y_i|, |yn_i|), kept
        // SQUARED: one square root per component (of the larger squared modulus) instead of three hypot() calls, and the
        // step factor 0.9 err^(-1/5) = 0.9 exp(-0.1 ln err^2) from the package's own exp / log -- a fifth of the 2100
        // instructions of a step were the libm hypot / pow of this block
        double err2 = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const cd e = h * (e1 * k1[i] + e3 * k3[i] + e4 * k4[i] + e5 * k5[i] + e6 * k6[i] + e7 * k7[i]);
            const double m2 = fmax(fma(y[i].re, y[i].re, y[i].im * y[i].im), fma(yn[i].re, yn[i].re, yn[i].im * yn[i].im));
            const double sc = fma(rtol, sqrt_pos0_1g(m2), atol);
            err2 += fma(e.re, e.re, e.im * e.im) * rcp_1n(sc * sc);
        }
        const double err_sq = err2 * 0.2;                                   // err^2
        if (err_sq <= 1.0) {
            t += h;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                y[i] = yn[i];
                k1[i] = k7[i];
            }
        }
        // a non-finite estimate (an overflowed trial state) is a rejection that SHRINKS the step, not the 5 x of err = 0
        const double fac = !(err_sq < 0x1.0p+1000) ? 0.2 : ((err_sq > 0.0) ? 0.9 * exp_fast(0.1 * neg_log(err_sq)) : 5.0);
        h *= fmin(5.0, fmax(0.2, fac));
    }
    if (!(t >= ttm)) {                                      // given up (ODE_STEP_FLOOR / ODE_MAX_TRIES): no state is better than a wrong one
#pragma unroll
        for (int i = 0; i < 5; ++i) y[i] = cd{__builtin_nan(""), __builtin_nan("")};
    }
}


// Dormand-Prince 8(5,3) (DOP853; coefficients: svmc_dop853.h, generated from scipy's table), the same controller as SciPy's
// DOP853: err5 = sum_j E5_j k_j and err3 = sum_j E3_j k_j scaled by sc_i = atol + rtol max(|y_i|, |yn_i|),
// error = h |err5|^2 / sqrt((|err5|^2 + 0.01 |err3|^2) n), accepted below 1, step factor 0.9 error^(-1/8) within [0.2, 10] (at
// most 1 straight after a rejection).  Round 4: at the committed tolerance (1e-10) the hardest grid points take 58 steps of
// 12 evaluations where the 5(4) pair took 302 steps of 6 -- 2.4 x fewer right-hand sides (counted with SciPy's own DOP853 and
// RK45 on this system) -- and the launch is latency-bound on exactly that count.  The CPU twin keeps its 5(4) pair: the two
// meet at the tolerance, as either meets the reference's tightened solver.
__device__ void dop853(const OdeConsts &c, cd phi, cd psi, double ttm, cd (&y)[5], double rtol, double atol)
{
    cd K1[5] = {}, K2[5] = {}, K3[5] = {}, K4[5] = {}, K5[5] = {}, K6[5] = {}, K7[5] = {}, K8[5] = {}, K9[5] = {}, K10[5] = {},
       K11[5] = {}, K12[5] = {}, yt[5], yn[5], kn[5];
    double t = 0.0, h = ttm / 8.0;
    int tries = 0;
    bool rejected = false;
    ode_rhs(c, phi, psi, y, K1);
#define SVMC_D853_LANE_STAGE(S, KS)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                                            \
    {                                                                                                                        \
        const cd k1 = K1[i], k2 = K2[i], k3 = K3[i], k4 = K4[i], k5 = K5[i], k6 = K6[i], k7 = K7[i], k8 = K8[i], k9 = K9[i],  \
                 k10 = K10[i], k11 = K11[i];                                                                                 \
        (void)k2; (void)k3; (void)k4; (void)k5; (void)k6; (void)k7; (void)k8; (void)k9; (void)k10; (void)k11;               \
        yt[i] = y[i] + h * (SVMC_D853_STAGE_##S);                                                                            \
    }                                                                                                                        \
    ode_rhs(c, phi, psi, yt, KS)
    while (t < ttm && tries < ODE_MAX_TRIES) {
        ++tries;
        if (!(h >= ODE_STEP_FLOOR * ttm)) break;          // the controller's step collapsed (or went NaN): given up below
        if (t + h > ttm) h = ttm - t;
        SVMC_D853_LANE_STAGE(2, K2);
        SVMC_D853_LANE_STAGE(3, K3);
        SVMC_D853_LANE_STAGE(4, K4);
        SVMC_D853_LANE_STAGE(5, K5);
        SVMC_D853_LANE_STAGE(6, K6);
        SVMC_D853_LANE_STAGE(7, K7);
        SVMC_D853_LANE_STAGE(8, K8);
        SVMC_D853_LANE_STAGE(9, K9);
        SVMC_D853_LANE_STAGE(10, K10);
        SVMC_D853_LANE_STAGE(11, K11);
        SVMC_D853_LANE_STAGE(12, K12);
        double e5 = 0.0, e3 = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const cd k1 = K1[i], k6 = K6[i], k7 = K7[i], k8 = K8[i], k9 = K9[i], k10 = K10[i], k11 = K11[i], k12 = K12[i];
            yn[i] = y[i] + h * (SVMC_D853_B);
            const cd err5 = SVMC_D853_E5, err3 = SVMC_D853_E3;
            const double m2 = fmax(fma(y[i].re, y[i].re, y[i].im * y[i].im), fma(yn[i].re, yn[i].re, yn[i].im * yn[i].im));
            const double sc = fma(rtol, sqrt_pos0_1g(m2), atol);
            const double inv = rcp_1n(sc * sc);
            e5 += fma(err5.re, err5.re, err5.im * err5.im) * inv;
            e3 += fma(err3.re, err3.re, err3.im * err3.im) * inv;
        }
        // error^2 = h^2 e5^2 / ((e5 + 0.01 e3) n), n = 5 ALWAYS: a first-order chain (three live components, the other two
        // identically zero) is normed over five where SciPy's DOP853 would take len(scale) = 3 -- its effective tolerance is
        // sqrt(5/3) looser than "SciPy's controller at rtol".  Deliberate: one norm for both orders keeps the lane-per-component
        // kernels' row reduction uniform, and the difference is a factor 1.3 in a tolerance that sits six orders below the
        // reference's own (rtol 1e-3); the first-order goldens hold at the same 1e-8.
        const double denom = fma(0.01, e3, e5);
        // a trial step that left the finite range (a quadratic system: a step too long for a far grid point overflows inside its
        // stages and the estimators come back inf or NaN) is a rejection with the smallest factor, not a number to take a power of
        const bool finite = denom < 0x1.0p+1000;           // false for inf and NaN
        const double err_sq = !finite ? __builtin_huge_val() : ((denom > 0.0) ? (h * h) * (e5 * e5) * rcp_1n(denom * 5.0) : 0.0);
        const bool accept = err_sq < 1.0;
        double fac = !finite ? 0.2 : ((err_sq > 0.0) ? 0.9 * exp_fast(0.0625 * neg_log(err_sq)) : 10.0);
        if (accept) {
            ode_rhs(c, phi, psi, yn, kn);
            t += h;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                y[i] = yn[i];
                K1[i] = kn[i];
            }
            fac = fmin(rejected ? 1.0 : 10.0, fac);
            rejected = false;
        } else {
            fac = fmax(0.2, fmin(fac, 1.0));
            rejected = true;
        }
        h *= fac;
    }
#undef SVMC_D853_LANE_STAGE
    if (!(t >= ttm)) {                                      // given up (ODE_STEP_FLOOR / ODE_MAX_TRIES): no state is better than a wrong one
#pragma unroll
        for (int i = 0; i < 5; ++i) y[i] = cd{__builtin_nan(""), __builtin_nan("")};
    }
}


// ---- one grid point per 16-lane DPP row ---------------------------------------------------------------------------
constexpr int ODE_ROW = 16;                 // lanes per grid point: components 0..4 work, 5..15 ride along on zero rows
constexpr int ODE_POINTS_PER_BLOCK = 4;     // a block is one wave

// lane N of the caller's row, to every lane of the row: v_mov_b32_dpp row_newbcast (two per double).  All 16 lanes of a row
// take every branch together, so the source lane is always enabled.
template <int N>
__device__ __forceinline__ double row_bcast(double v)
{
    // mov_dpp with bound_ctrl: no "old" value to preserve, so no register initialisation ahead of each of the ~100 moves a step
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ cd row_bcast(cd v)
{
    return cd{row_bcast<N>(v.re), row_bcast<N>(v.im)};
}

// the lane's derivative at the state whose component it holds in `v`
__device__ __forceinline__ cd ode_rhs_row(const OdeLane &k, cd v, bool second)
{
    const cd A1 = row_bcast<1>(v), A2 = row_bcast<2>(v);
    cd A3 = C(0.0), A4 = C(0.0);
    if (second) {                                          // wave-uniform
        A3 = row_bcast<3>(v);
        A4 = row_bcast<4>(v);
    }
    return ode_rhs_lane(k, A1, A2, A3, A4, second);
}

// Dormand-Prince 5(4), FSAL, mixed error scale, RMS norm over the five components -- the CPU twin's dopri5() (same tableau,
// same controller), each lane carrying ONE component: y, the seven stage derivatives and the trial states are single
// complex numbers here.  The error norm is the sum of the five lanes' terms taken in component order by every lane of the
// row, so the whole row sees the same number and takes the same decision.
__device__ void dopri5_row(const OdeLane &k, bool second, double ttm, cd &y, double rtol, double atol)
{
    constexpr double a21 = 1.0 / 5, a31 = 3.0 / 40, a32 = 9.0 / 40, a41 = 44.0 / 45, a42 = -56.0 / 15, a43 = 32.0 / 9,
                     a51 = 19372.0 / 6561, a52 = -25360.0 / 2187, a53 = 64448.0 / 6561, a54 = -212.0 / 729,
                     a61 = 9017.0 / 3168, a62 = -355.0 / 33, a63 = 46732.0 / 5247, a64 = 49.0 / 176,
                     a65 = -5103.0 / 18656, b1 = 35.0 / 384, b3 = 500.0 / 1113, b4 = 125.0 / 192, b5 = -2187.0 / 6784,
                     b6 = 11.0 / 84, e1 = 71.0 / 57600, e3 = -71.0 / 16695, e4 = 71.0 / 1920, e5 = -17253.0 / 339200,
                     e6 = 22.0 / 525, e7 = -1.0 / 40;
    double t = 0.0, h = ttm / 32.0;
    int tries = 0;
    cd k1 = ode_rhs_row(k, y, second);
    while (t < ttm && tries < ODE_MAX_TRIES) {
        ++tries;
        if (!(h >= ODE_STEP_FLOOR * ttm)) break;          // row-uniform (h is): given up below
        if (t + h > ttm) h = ttm - t;
        const cd k2 = ode_rhs_row(k, y + h * (a21 * k1), second);
        const cd k3 = ode_rhs_row(k, y + h * (a31 * k1 + a32 * k2), second);
        const cd k4 = ode_rhs_row(k, y + h * (a41 * k1 + a42 * k2 + a43 * k3), second);
        const cd k5 = ode_rhs_row(k, y + h * (a51 * k1 + a52 * k2 + a53 * k3 + a54 * k4), second);
        const cd k6 = ode_rhs_row(k, y + h * (a61 * k1 + a62 * k2 + a63 * k3 + a64 * k4 + a65 * k5), second);
        const cd yn = y + h * (b1 * k1 + b3 * k3 + b4 * k4 + b5 * k5 + b6 * k6);
        const cd k7 = ode_rhs_row(k, yn, second);
        // the twin's err = sqrt(mean_i (|e_i| / sc_i)^2) with sc_i = atol + rtol max(|y_i|, |yn_i|), kept SQUARED: one square
        // root per component (of the larger squared modulus), and the step factor 0.9 err^(-1/5) = 0.9 exp(-0.1 ln err^2)
        // from the package's own exp / log
        const cd e = h * (e1 * k1 + e3 * k3 + e4 * k4 + e5 * k5 + e6 * k6 + e7 * k7);
        const double m2 = fmax(fma(y.re, y.re, y.im * y.im), fma(yn.re, yn.re, yn.im * yn.im));
        const double sc = fma(rtol, sqrt_pos0_1g(m2), atol);
        const double term = fma(e.re, e.re, e.im * e.im) * rcp_1n(sc * sc);
        const double err2 = (((row_bcast<0>(term) + row_bcast<1>(term)) + row_bcast<2>(term)) + row_bcast<3>(term)) + row_bcast<4>(term);
        const double err_sq = err2 * 0.2;                                   // err^2
        if (err_sq <= 1.0) {                               // row-uniform
            t += h;
            y = yn;
            k1 = k7;
        }
        // a non-finite estimate (an overflowed trial state) is a rejection that SHRINKS the step, not the 5 x of err = 0
        const double fac = !(err_sq < 0x1.0p+1000) ? 0.2 : ((err_sq > 0.0) ? 0.9 * exp_fast(0.1 * neg_log(err_sq)) : 5.0);
        h *= fmin(5.0, fmax(0.2, fac));
    }
    if (!(t >= ttm)) y = cd{__builtin_nan(""), __builtin_nan("")};      // given up (ODE_STEP_FLOOR / ODE_MAX_TRIES); row-uniform
}

// DOP853 with one component per lane (dop853 above, the row form of dopri5_row): twelve stage derivatives, the trial states and
// both error estimators are single complex numbers per lane; the two squared norms are summed over the row in component order
// by every lane, so the whole row takes the same decision.
__device__ void dop853_row(const OdeLane &k, bool second, double ttm, cd &y, double rtol, double atol)
{
    double t = 0.0, h = ttm / 8.0;
    int tries = 0;
    bool rejected = false;
    cd k1 = ode_rhs_row(k, y, second);
    while (t < ttm && tries < ODE_MAX_TRIES) {
        ++tries;
        if (!(h >= ODE_STEP_FLOOR * ttm)) break;          // row-uniform (h is): given up below
        if (t + h > ttm) h = ttm - t;
        const cd k2 = ode_rhs_row(k, y + h * (SVMC_D853_STAGE_2), second);
        const cd k3 = ode_rhs_row(k, y + h * (SVMC_D853_STAGE_3), second);
        const cd k4 = ode_rhs_row(k, y + h * (SVMC_D853_STAGE_4), second);
        const cd k5 = ode_rhs_row(k, y + h * (SVMC_D853_STAGE_5), second);
        const cd k6 = ode_rhs_row(k, y + h * (SVMC_D853_STAGE_6), second);
        const cd k7 = ode_rhs_row(k, y + h * (SVMC_D853_STAGE_7), second);
        const cd k8 = ode_rhs_row(k, y + h * (SVMC_D853_STAGE_8), second);
        const cd k9 = ode_rhs_row(k, y + h * (SVMC_D853_STAGE_9), second);
        const cd k10 = ode_rhs_row(k, y + h * (SVMC_D853_STAGE_10), second);
        const cd k11 = ode_rhs_row(k, y + h * (SVMC_D853_STAGE_11), second);
        const cd k12 = ode_rhs_row(k, y + h * (SVMC_D853_STAGE_12), second);
        (void)k2; (void)k3;
        const cd yn = y + h * (SVMC_D853_B);
        const cd err5 = SVMC_D853_E5, err3 = SVMC_D853_E3;
        const double m2 = fmax(fma(y.re, y.re, y.im * y.im), fma(yn.re, yn.re, yn.im * yn.im));
        const double sc = fma(rtol, sqrt_pos0_1g(m2), atol);
        const double inv = rcp_1n(sc * sc);
        const double t5 = fma(err5.re, err5.re, err5.im * err5.im) * inv, t3 = fma(err3.re, err3.re, err3.im * err3.im) * inv;
        const double e5 = (((row_bcast<0>(t5) + row_bcast<1>(t5)) + row_bcast<2>(t5)) + row_bcast<3>(t5)) + row_bcast<4>(t5);
        const double e3 = (((row_bcast<0>(t3) + row_bcast<1>(t3)) + row_bcast<2>(t3)) + row_bcast<3>(t3)) + row_bcast<4>(t3);
        const double denom = fma(0.01, e3, e5);
        const bool finite = denom < 0x1.0p+1000;           // false for inf and NaN: an overflowed trial step (dop853 above)
        const double err_sq = !finite ? __builtin_huge_val() : ((denom > 0.0) ? (h * h) * (e5 * e5) * rcp_1n(denom * 5.0) : 0.0);
        const bool accept = err_sq < 1.0;                  // row-uniform
        double fac = !finite ? 0.2 : ((err_sq > 0.0) ? 0.9 * exp_fast(0.0625 * neg_log(err_sq)) : 10.0);
        if (accept) {
            k1 = ode_rhs_row(k, yn, second);
            t += h;
            y = yn;
            fac = fmin(rejected ? 1.0 : 10.0, fac);
            rejected = false;
        } else {
            fac = fmax(0.2, fmin(fac, 1.0));
            rejected = true;
        }
        h *= fac;
    }
    if (!(t >= ttm)) y = cd{__builtin_nan(""), __builtin_nan("")};      // given up (ODE_STEP_FLOOR / ODE_MAX_TRIES); row-uniform
}


constexpr int AB = 64;  // one wave per block
constexpr int MAX_ODE_SETS = 16;       // parameter sets per launch (kernel-argument block: 16 x 112 B)

// The parameter sets of one launch: blockIdx.y picks the set, every set has its own transform grid (the grid scale
// follows sigma0), coefficients and output.  One set's 1000 points are 250 one-wave blocks, busy for the latency of
// their slowest point; independent sets -- the five of config C5, the bumps of a finite-difference gradient -- cost
// nothing extra side by side.
struct OdeBatch {
    OdeConsts c[MAX_ODE_SETS];
    double y0[MAX_ODE_SETS];           // sigma0 - theta
};

__global__ __launch_bounds__(AB) void logsv_mgf_grid_kernel(const cd *__restrict__ phi, const cd *__restrict__ psi,
                                                            size_t n_grid, double ttm, OdeBatch sets,
                                                            cd *__restrict__ a, cd *__restrict__ log_mgf, double rtol,
                                                            double atol)
{
    const int comp = threadIdx.x & (ODE_ROW - 1);
    const size_t j = static_cast<size_t>(blockIdx.x) * ODE_POINTS_PER_BLOCK + (threadIdx.x / ODE_ROW);
    if (j >= n_grid) return;                               // whole rows leave together
    const OdeConsts c = sets.c[blockIdx.y];
    const double y0 = sets.y0[blockIdx.y];
    const size_t g = static_cast<size_t>(blockIdx.y) * n_grid + j;     // this set's grid point
    const bool second = c.second != 0;
    const int n = second ? 5 : 3;
    const bool mine = comp < n;
    const OdeLane k = make_ode_lane(c, phi[g], psi[g], mine ? comp : -1);
    cd y = mine ? a[g * n + comp] : C(0.0);
#if SVMC_ODE_DOP853
    dop853_row(k, second, ttm, y, rtol, atol);
#else
    dopri5_row(k, second, ttm, y, rtol, atol);
#endif
    if (mine) a[g * n + comp] = y;
    // log E = sum_k A_k (sigma0 - theta)^k, affine_expansion.py:674-685, in component order
    const cd A0 = row_bcast<0>(y), A1 = row_bcast<1>(y), A2 = row_bcast<2>(y), A3 = row_bcast<3>(y), A4 = row_bcast<4>(y);
    if (comp == 0) {
        cd lm = C(0.0);
        double yk = 1.0;
        lm = lm + yk * A0;
        yk *= y0;
        lm = lm + yk * A1;
        yk *= y0;
        lm = lm + yk * A2;
        if (second) {
            yk *= y0;
            lm = lm + yk * A3;
            yk *= y0;
            lm = lm + yk * A4;
        }
        log_mgf[g] = lm;
    }
}

__global__ __launch_bounds__(AB) void logsv_mgf_grid_lane_kernel(const cd *__restrict__ phi, const cd *__restrict__ psi,
                                                            size_t n_grid, double ttm, OdeBatch sets,
                                                            cd *__restrict__ a, cd *__restrict__ log_mgf, double rtol,
                                                            double atol)
{
    const size_t j = static_cast<size_t>(blockIdx.x) * AB + threadIdx.x;
    if (j >= n_grid) return;
    const OdeConsts c = sets.c[blockIdx.y];
    const double y0 = sets.y0[blockIdx.y];
    const size_t g = static_cast<size_t>(blockIdx.y) * n_grid + j;     // this set's grid point
    const int n = c.second ? 5 : 3;
    cd A[5] = {C(0.0), C(0.0), C(0.0), C(0.0), C(0.0)};
    for (int k = 0; k < n; ++k) A[k] = a[g * n + k];
#if SVMC_ODE_DOP853
    dop853(c, phi[g], psi[g], ttm, A, rtol, atol);
#else
    dopri5(c, phi[g], psi[g], ttm, A, rtol, atol);
#endif
    cd lm = C(0.0);
    double yk = 1.0;
    for (int k = 0; k < n; ++k) {
        a[g * n + k] = A[k];
        lm = lm + yk * A[k];                                          // affine_expansion.py:674-685
        yk *= y0;
    }
    log_mgf[g] = lm;
}

__global__ __launch_bounds__(AB) void heston_mgf_grid_kernel(const cd *__restrict__ phi, const cd *__restrict__ psi,
                                                             size_t n_grid, double ttm, double v0, double theta,
                                                             double kappa, double volvol, double rho, cd *__restrict__ a,
                                                             cd *__restrict__ b, int have_t0, cd *__restrict__ log_mgf)
{
    const size_t j = static_cast<size_t>(blockIdx.x) * AB + threadIdx.x;
    if (j >= n_grid) return;
    const double volvol2 = volvol * volvol;
    const cd ph = phi[j], ps = psi[j];
    const cd b1 = (rho * volvol) * ph + kappa;                                                   // :197
    const cd b0 = 0.5 * (ph * (ph + 1.0)) - ps;                                                 // :198
    const cd zeta = csqrt_(b1 * b1 - (2.0 * volvol2) * b0);                                     // :199
    const cd exp_zeta = cexp_(-(ttm * zeta));
    const cd psi_p = zeta - b1, psi_m = zeta + b1;
    cd c_p, c_m;
    if (!have_t0) {
        c_p = psi_p / (2.0 * zeta);
        c_m = psi_m / (2.0 * zeta);
    } else {
        c_p = (psi_p + volvol2 * b[j]) / (2.0 * zeta);
        c_m = (psi_m - volvol2 * b[j]) / (2.0 * zeta);
    }
    const cd den = c_p * exp_zeta + c_m;
    const cd b_t1 = -((psi_p * c_m - psi_m * c_p * exp_zeta) / (volvol2 * den));                // :207
    cd a_t1 = -(theta * kappa / volvol2) * (ttm * psi_p + 2.0 * clog_(den));                    // :208
    if (have_t0) a_t1 = a_t1 + a[j];
    a[j] = a_t1;
    b[j] = b_t1;
    log_mgf[j] = a_t1 + v0 * b_t1;
}

struct StrikeArgs {
    double x[32];   // log(forward / strike)
    int k;
};

// one block per strike; legacy Simpson weights (utils/mgf_pricer.py:158-171): 1,4,2,...,  every odd index 4
__global__ __launch_bounds__(256) void mgf_vanilla_slice_kernel(const cd *__restrict__ phi, const cd *__restrict__ log_mgf,
                                                                int n_grid, StrikeArgs sa, double *__restrict__ capped,
                                                                int capped_ld)
{
    __shared__ double lds[4];
    const double PI = 3.14159265358979323846;
    phi += static_cast<size_t>(blockIdx.y) * n_grid;                 // blockIdx.y: the parameter set of a batched call
    log_mgf += static_cast<size_t>(blockIdx.y) * n_grid;
    capped += static_cast<size_t>(blockIdx.y) * capped_ld;
    const double x = sa.x[blockIdx.x];
    const double h = phi[1].im - phi[0].im;
    double s = 0.0;
    for (int j = threadIdx.x; j < n_grid; j += 256) {
        double w = 2.0;
        if (j == 0 || j == n_grid - 1) w = 1.0;
        if (j & 1) w = 4.0;
        const double p = phi[j].im;
        const double pw = ((h / 3.0) * w / PI) / (p * p + 0.25);
        const cd e = cexp_(log_mgf[j] - x * phi[j]);
        const double term = pw * e.re;
        if (term == term) s += term;                                                            // nansum
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) capped[blockIdx.x] = ((lds[0] + lds[1]) + lds[2]) + lds[3];
}

// options on quadratic variance, utils/mgf_pricer.py:322-356: one block per strike,
// sum_j Re[ w_j / (pi psi_j^2) exp(K ttm psi_j + log E_j) ] over the 40 000-point psi grid
__global__ __launch_bounds__(256) void mgf_qvar_slice_kernel(const cd *__restrict__ psi, const cd *__restrict__ log_mgf,
                                                             int n_grid, StrikeArgs sa, double *__restrict__ capped)
{
    __shared__ double lds[4];
    const double PI = 3.14159265358979323846;
    const double kt = sa.x[blockIdx.x];                  // strike * ttm
    const double h = psi[1].im - psi[0].im;
    double s = 0.0;
    for (int j = threadIdx.x; j < n_grid; j += 256) {
        double w = 2.0;
        if (j == 0 || j == n_grid - 1) w = 1.0;
        if (j & 1) w = 4.0;
        const cd ps = psi[j];
        const cd term = (C((h / 3.0) * w / PI) / (ps * ps)) * cexp_(kt * ps + log_mgf[j]);
        if (term.re == term.re) s += term.re;                                                   // nansum
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) capped[blockIdx.x] = ((lds[0] + lds[1]) + lds[2]) + lds[3];
}

static int check_launch_a(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SVMC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return SVMC_OK;
}

}  // namespace svmc

using namespace svmc;

extern "C" {

int svmc_logsv_mgf_grid_batch(const double *phi, const double *psi, size_t n_grid, int n_sets, double ttm,
                              const double *params_host, int is_spot_measure, int expansion_order, double *a,
                              double *log_mgf, double rtol, double atol, svmc_stream_t stream)
{
    const char *fn = "svmc_logsv_mgf_grid_batch";
    SVMC_REQUIRE(phi && psi && a && log_mgf && params_host, "svmc_logsv_mgf_grid_batch: null pointer");
    SVMC_REQUIRE(expansion_order == 1 || expansion_order == 2, "svmc_logsv_mgf_grid_batch: expansion_order must be 1 or 2");
    SVMC_REQUIRE(ttm > 0.0 && rtol > 0.0 && atol > 0.0, "svmc_logsv_mgf_grid_batch: ttm, rtol, atol must be positive");
    SVMC_REQUIRE(n_sets >= 1, "svmc_logsv_mgf_grid_batch: n_sets < 1");
    if (n_grid == 0) return SVMC_OK;
    const int n_coef = (expansion_order == 2) ? 5 : 3;
    for (int s0 = 0; s0 < n_sets; s0 += MAX_ODE_SETS) {
        const int m = (n_sets - s0 < MAX_ODE_SETS) ? (n_sets - s0) : MAX_ODE_SETS;
        OdeBatch sets;
        for (int i = 0; i < MAX_ODE_SETS; ++i) {
            const double *p = params_host + static_cast<size_t>(SVMC_LOGSV_SET_DOUBLES) * (s0 + (i < m ? i : 0));
            // p = {sigma0, theta, kappa1, kappa2, beta, volvol, vol_backbone_eta}
            sets.c[i] = make_ode_consts(p[1], p[2], p[3], p[4], p[5], is_spot_measure, expansion_order, p[6]);
            sets.y0[i] = p[0] - p[1];
        }
        const size_t off = static_cast<size_t>(s0) * n_grid;
        // rows while the launch stays within two waves per SIMD (8192 points at one set), lanes beyond: see above
        static const size_t row_max = getenv("SVMC_MGF_ROW_MAX_POINTS") ? strtoull(getenv("SVMC_MGF_ROW_MAX_POINTS"), nullptr, 10) : 8192;
        if (n_grid * static_cast<size_t>(m) <= row_max)
            hipLaunchKernelGGL(logsv_mgf_grid_kernel,
                               dim3(static_cast<unsigned>((n_grid + ODE_POINTS_PER_BLOCK - 1) / ODE_POINTS_PER_BLOCK), static_cast<unsigned>(m)),
                               dim3(AB), 0, as_stream(stream), reinterpret_cast<const cd *>(phi) + off,
                               reinterpret_cast<const cd *>(psi) + off, n_grid, ttm, sets, reinterpret_cast<cd *>(a) + off * n_coef,
                               reinterpret_cast<cd *>(log_mgf) + off, rtol, atol);
        else
            hipLaunchKernelGGL(logsv_mgf_grid_lane_kernel, dim3(static_cast<unsigned>((n_grid + AB - 1) / AB), static_cast<unsigned>(m)),
                               dim3(AB), 0, as_stream(stream), reinterpret_cast<const cd *>(phi) + off,
                               reinterpret_cast<const cd *>(psi) + off, n_grid, ttm, sets, reinterpret_cast<cd *>(a) + off * n_coef,
                               reinterpret_cast<cd *>(log_mgf) + off, rtol, atol);
    }
    return check_launch_a(fn);
}

int svmc_logsv_mgf_grid(const double *phi, const double *psi, size_t n_grid, double ttm, double sigma0, double theta,
                        double kappa1, double kappa2, double beta, double volvol, int is_spot_measure,
                        int expansion_order, double vol_backbone_eta, double *a, double *log_mgf, double rtol,
                        double atol, svmc_stream_t stream)
{
    const double set[SVMC_LOGSV_SET_DOUBLES] = {sigma0, theta, kappa1, kappa2, beta, volvol, vol_backbone_eta, 0.0};
    return svmc_logsv_mgf_grid_batch(phi, psi, n_grid, 1, ttm, set, is_spot_measure, expansion_order, a, log_mgf, rtol, atol,
                                     stream);
}

int svmc_heston_mgf_grid(const double *phi, const double *psi, size_t n_grid, double ttm, double v0, double theta,
                         double kappa, double volvol, double rho, double *a, double *b, int have_t0, double *log_mgf,
                         svmc_stream_t stream)
{
    SVMC_REQUIRE(phi && psi && a && b && log_mgf, "svmc_heston_mgf_grid: null pointer");
    if (n_grid == 0) return SVMC_OK;
    hipLaunchKernelGGL(heston_mgf_grid_kernel, dim3(static_cast<unsigned>((n_grid + AB - 1) / AB)), dim3(AB), 0,
                       as_stream(stream), reinterpret_cast<const cd *>(phi), reinterpret_cast<const cd *>(psi), n_grid, ttm,
                       v0, theta, kappa, volvol, rho, reinterpret_cast<cd *>(a), reinterpret_cast<cd *>(b), have_t0,
                       reinterpret_cast<cd *>(log_mgf));
    return check_launch_a("svmc_heston_mgf_grid");
}

int svmc_mgf_vanilla_slice_batch(const double *phi, const double *log_mgf, size_t n_grid, int n_sets, double forward,
                                 const double *strikes_host, size_t n_strikes, double *capped, svmc_stream_t stream)
{
    SVMC_REQUIRE(phi && log_mgf && capped, "svmc_mgf_vanilla_slice: null pointer");
    SVMC_REQUIRE(n_grid >= 3 && n_grid < (1u << 30), "svmc_mgf_vanilla_slice: grid too short or too long");
    SVMC_REQUIRE(n_strikes == 0 || strikes_host, "svmc_mgf_vanilla_slice: null strikes");
    SVMC_REQUIRE(n_sets >= 1 && n_sets <= 65535, "svmc_mgf_vanilla_slice: n_sets out of range");
    for (size_t k0 = 0; k0 < n_strikes; k0 += 32) {
        StrikeArgs sa;
        sa.k = static_cast<int>((n_strikes - k0 < 32) ? (n_strikes - k0) : 32);
        for (int k = 0; k < 32; ++k) sa.x[k] = (k < sa.k) ? log(forward / strikes_host[k0 + k]) : 0.0;    // :199
        hipLaunchKernelGGL(mgf_vanilla_slice_kernel, dim3(sa.k, static_cast<unsigned>(n_sets)), dim3(256), 0,
                           as_stream(stream), reinterpret_cast<const cd *>(phi), reinterpret_cast<const cd *>(log_mgf),
                           static_cast<int>(n_grid), sa, capped + k0, static_cast<int>(n_strikes));
    }
    return check_launch_a("svmc_mgf_vanilla_slice");
}

int svmc_mgf_vanilla_slice(const double *phi, const double *log_mgf, size_t n_grid, double forward,
                           const double *strikes_host, size_t n_strikes, double *capped, svmc_stream_t stream)
{
    return svmc_mgf_vanilla_slice_batch(phi, log_mgf, n_grid, 1, forward, strikes_host, n_strikes, capped, stream);
}

int svmc_mgf_qvar_slice(const double *psi, const double *log_mgf, size_t n_grid, double ttm, const double *strikes_host,
                        size_t n_strikes, double *capped, svmc_stream_t stream)
{
    SVMC_REQUIRE(psi && log_mgf && capped, "svmc_mgf_qvar_slice: null pointer");
    SVMC_REQUIRE(n_grid >= 3 && n_grid < (1u << 30), "svmc_mgf_qvar_slice: grid too short or too long");
    SVMC_REQUIRE(n_strikes == 0 || strikes_host, "svmc_mgf_qvar_slice: null strikes");
    for (size_t k0 = 0; k0 < n_strikes; k0 += 32) {
        StrikeArgs sa;
        sa.k = static_cast<int>((n_strikes - k0 < 32) ? (n_strikes - k0) : 32);
        for (int k = 0; k < 32; ++k) sa.x[k] = (k < sa.k) ? strikes_host[k0 + k] * ttm : 0.0;             // :343
        hipLaunchKernelGGL(mgf_qvar_slice_kernel, dim3(sa.k), dim3(256), 0, as_stream(stream),
                           reinterpret_cast<const cd *>(psi), reinterpret_cast<const cd *>(log_mgf),
                           static_cast<int>(n_grid), sa, capped + k0);
    }
    return check_launch_a("svmc_mgf_qvar_slice");
}

}  // extern "C"
