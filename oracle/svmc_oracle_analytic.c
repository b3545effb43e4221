/*
 * svmc_oracle_analytic.c -- CPU restatement of the analytic (Fourier / affine-expansion) side of the LogSV and
 * Heston chain pricers: SURVEY.md row a11 / config C5.  TEST INFRASTRUCTURE ONLY (see svmc_oracle.h).
 *
 *   pricers/logsv/affine_expansion.py:67-184   func_a_ode_quadratic_terms  (M, L, H of Eq. 4.17 / 4.25)
 *   pricers/logsv/affine_expansion.py:187-205  func_rhs                    A' = A^T M A + L A + H
 *   pricers/logsv/affine_expansion.py:229-303  solve_ode_for_a             (SciPy RK45 there; see below)
 *   pricers/logsv/affine_expansion.py:570-685  compute_logsv_a_mgf_grid    log E = sum_k A_k y^k
 *   utils/mgf_pricer.py:158-171, 174-221       legacy Simpson weights, vanilla_slice_pricer_with_mgf_grid
 *   pricers/heston_pricer.py:183-214           compute_heston_mgf_grid     (closed form)
 *
 * The one deliberate difference: the reference integrates the coefficient ODEs with scipy.integrate.solve_ivp at
 * its default tolerances (RK45, rtol 1e-3, atol 1e-6); here it is an embedded Dormand-Prince 5(4) pair with
 * rtol/atol supplied by the caller (tests use 1e-10 / 1e-12).  Goldens therefore come in two flavours:
 * the reference as shipped (agreement to its own solver tolerance, ~1e-4 in price) and the reference with
 * solve_ivp tightened to rtol 1e-11 (agreement ~1e-8) -- tests/golden/make_golden.py g_analytic_tight.
 */
#include "svmc_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>

typedef double complex cd;

typedef struct {
    double theta, theta2, vartheta2, qv, qv2, b, eta2, lamda, kappa2_p, kappa_p;
    int spot, second;
} ode_consts;

static ode_consts make_ode_consts(double theta, double kappa1, double kappa2, double beta, double volvol,
                                  int is_spot_measure, int expansion_order, double eta)
{
    ode_consts c;
    c.theta = theta;
    c.theta2 = theta * theta;
    c.vartheta2 = beta * beta + volvol * volvol;
    c.qv = theta * c.vartheta2;
    c.qv2 = c.theta2 * c.vartheta2;
    c.b = beta * eta;                                   /* beta * vol_backbone_eta */
    c.eta2 = eta * eta;
    c.spot = is_spot_measure;
    c.second = (expansion_order == 2);
    if (is_spot_measure) {                              /* :132-140 */
        c.lamda = 0.0;
        c.kappa2_p = kappa2;
        c.kappa_p = kappa1 + kappa2 * theta;
    } else {
        c.lamda = beta * c.theta2 * eta;
        c.kappa2_p = kappa2 - beta * eta;
        c.kappa_p = kappa1 + kappa2 * theta - 2.0 * beta * theta * eta;
    }
    return c;
}

/* A' = A^T M^(k) A + L^(k) A + H^(k), entries of :146-182 written out (M is sparse and symmetric) */
static void ode_rhs(const ode_consts *c, cd phi, cd psi, const cd A[5], cd out[5])
{
    const double qv = c->qv, qv2 = c->qv2, v2 = c->vartheta2, th = c->theta, th2 = c->theta2;
    const cd bphi = c->b * phi;
    const cd A1 = A[1], A2 = A[2], A3 = c->second ? A[3] : 0.0, A4 = c->second ? A[4] : 0.0;
    const cd rhs = c->spot ? (phi * (phi + 1.0) - 2.0 * psi) : (phi * (phi - 1.0) - 2.0 * psi);
    const cd H0 = 0.5 * th2 * c->eta2 * rhs, H1 = th * c->eta2 * rhs, H2 = 0.5 * c->eta2 * rhs;
    const cd L01 = c->lamda - th2 * bphi, L02 = qv2;
    const cd L11 = -c->kappa_p - 2.0 * th * bphi, L12 = 2.0 * (c->lamda + qv - th2 * bphi);
    const cd L21 = -c->kappa2_p - bphi, L22 = v2 - 2.0 * c->kappa_p - 4.0 * th * bphi;
    cd q0 = 0.5 * qv2 * A1 * A1;
    cd q1 = qv * A1 * A1 + 2.0 * qv2 * A1 * A2;
    cd q2 = 0.5 * v2 * A1 * A1 + 2.0 * qv2 * A2 * A2 + 4.0 * qv * A1 * A2;
    out[0] = q0 + L01 * A1 + L02 * A2 + H0;
    out[1] = q1 + L11 * A1 + L12 * A2 + H1;
    out[2] = q2 + L21 * A1 + L22 * A2 + H2;
    if (c->second) {
        const cd L13 = 3.0 * qv2;
        const cd L23 = 3.0 * (2.0 * qv - th2 * bphi), L24 = 6.0 * qv2;
        const cd L32 = -2.0 * (c->kappa2_p + bphi), L33 = 3.0 * (v2 - c->kappa_p - 2.0 * th * bphi),
                 L34 = 4.0 * (3.0 * qv - th2 * bphi);
        const cd L43 = -3.0 * (c->kappa2_p + bphi), L44 = 2.0 * (v2 - 2.0 * c->kappa_p - 4.0 * th * bphi);
        out[1] += L13 * A3;
        out[2] += 3.0 * qv2 * A1 * A3 + L23 * A3 + L24 * A4;
        out[3] = 4.0 * qv * A2 * A2 + 2.0 * v2 * A1 * A2 + 6.0 * qv * A1 * A3 + 4.0 * qv2 * A1 * A4 + 6.0 * qv2 * A2 * A3
                 + L32 * A2 + L33 * A3 + L34 * A4;
        out[4] = 2.0 * v2 * A2 * A2 + 4.5 * qv2 * A3 * A3 + 3.0 * v2 * A1 * A3 + 8.0 * qv * A1 * A4 + 12.0 * qv * A2 * A3
                 + 8.0 * qv2 * A2 * A4 + L43 * A3 + L44 * A4;
    } else {
        out[3] = 0.0;
        out[4] = 0.0;
    }
}

/* the right-hand side on its own: tests/test_math_accuracy.py checks the device's one-component-per-lane rows
 * (stochvolmodels_amd/csrc/svmc_ode.h) against it.  A and out: 5 complex numbers as (re, im) pairs. */
void svo_logsv_ode_rhs(double theta, double kappa1, double kappa2, double beta, double volvol, int is_spot_measure,
                       int expansion_order, double eta, const double *phi, const double *psi, const double *A, double *out)
{
    const ode_consts c = make_ode_consts(theta, kappa1, kappa2, beta, volvol, is_spot_measure, expansion_order, eta);
    cd a[5], o[5];
    for (int i = 0; i < 5; ++i) a[i] = A[2 * i] + I * A[2 * i + 1];
    ode_rhs(&c, phi[0] + I * phi[1], psi[0] + I * psi[1], a, o);
    for (int i = 0; i < 5; ++i) {
        out[2 * i] = creal(o[i]);
        out[2 * i + 1] = cimag(o[i]);
    }
}

/* Dormand-Prince 5(4), FSAL, per-component mixed error scale, RMS norm; returns the number of accepted steps */
static int dopri5(const ode_consts *c, cd phi, cd psi, double ttm, cd y[5], double rtol, double atol)
{
    static const double a21 = 1.0 / 5, a31 = 3.0 / 40, a32 = 9.0 / 40, a41 = 44.0 / 45, a42 = -56.0 / 15, a43 = 32.0 / 9,
                        a51 = 19372.0 / 6561, a52 = -25360.0 / 2187, a53 = 64448.0 / 6561, a54 = -212.0 / 729,
                        a61 = 9017.0 / 3168, a62 = -355.0 / 33, a63 = 46732.0 / 5247, a64 = 49.0 / 176, a65 = -5103.0 / 18656,
                        b1 = 35.0 / 384, b3 = 500.0 / 1113, b4 = 125.0 / 192, b5 = -2187.0 / 6784, b6 = 11.0 / 84,
                        e1 = 71.0 / 57600, e3 = -71.0 / 16695, e4 = 71.0 / 1920, e5 = -17253.0 / 339200, e6 = 22.0 / 525,
                        e7 = -1.0 / 40;
    cd k1[5], k2[5], k3[5], k4[5], k5[5], k6[5], k7[5], yt[5], yn[5];
    double t = 0.0, h = ttm / 32.0;
    int steps = 0, tries = 0;
    ode_rhs(c, phi, psi, y, k1);
    while (t < ttm && tries < 1000000) {
        ++tries;
        if (t + h > ttm) h = ttm - t;
        for (int i = 0; i < 5; ++i) yt[i] = y[i] + h * (a21 * k1[i]);
        ode_rhs(c, phi, psi, yt, k2);
        for (int i = 0; i < 5; ++i) yt[i] = y[i] + h * (a31 * k1[i] + a32 * k2[i]);
        ode_rhs(c, phi, psi, yt, k3);
        for (int i = 0; i < 5; ++i) yt[i] = y[i] + h * (a41 * k1[i] + a42 * k2[i] + a43 * k3[i]);
        ode_rhs(c, phi, psi, yt, k4);
        for (int i = 0; i < 5; ++i) yt[i] = y[i] + h * (a51 * k1[i] + a52 * k2[i] + a53 * k3[i] + a54 * k4[i]);
        ode_rhs(c, phi, psi, yt, k5);
        for (int i = 0; i < 5; ++i) yt[i] = y[i] + h * (a61 * k1[i] + a62 * k2[i] + a63 * k3[i] + a64 * k4[i] + a65 * k5[i]);
        ode_rhs(c, phi, psi, yt, k6);
        for (int i = 0; i < 5; ++i) yn[i] = y[i] + h * (b1 * k1[i] + b3 * k3[i] + b4 * k4[i] + b5 * k5[i] + b6 * k6[i]);
        ode_rhs(c, phi, psi, yn, k7);
        double err2 = 0.0;
        for (int i = 0; i < 5; ++i) {
            cd e = h * (e1 * k1[i] + e3 * k3[i] + e4 * k4[i] + e5 * k5[i] + e6 * k6[i] + e7 * k7[i]);
            double sc = atol + rtol * fmax(cabs(y[i]), cabs(yn[i]));
            double r = cabs(e) / sc;
            err2 += r * r;
        }
        double err = sqrt(err2 / 5.0);
        if (err <= 1.0) {
            t += h;
            for (int i = 0; i < 5; ++i) { y[i] = yn[i]; k1[i] = k7[i]; }
            ++steps;
        }
        double fac = (err > 0.0) ? 0.9 * pow(err, -0.2) : 5.0;
        h *= fmin(5.0, fmax(0.2, fac));
    }
    return steps;
}

/* compute_logsv_a_mgf_grid, pricers/logsv/affine_expansion.py:570-685: a (in: A(0) per grid point, out: A(ttm)),
 * log_mgf = sum_k A_k (sigma0 - theta)^k.  Arrays are interleaved complex128 like NumPy's. */
void svo_logsv_mgf_grid(size_t n_grid, const double *phi, const double *psi, double ttm, double sigma0, double theta,
                        double kappa1, double kappa2, double beta, double volvol, int is_spot_measure,
                        int expansion_order, double vol_backbone_eta, double *a, double *log_mgf, double rtol, double atol)
{
    const ode_consts c = make_ode_consts(theta, kappa1, kappa2, beta, volvol, is_spot_measure, expansion_order, vol_backbone_eta);
    const int n = (expansion_order == 2) ? 5 : 3;
    const double y = sigma0 - theta;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8)
#endif
    for (size_t j = 0; j < n_grid; ++j) {
        cd A[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < n; ++k) A[k] = a[2 * (j * n + k)] + I * a[2 * (j * n + k) + 1];
        dopri5(&c, phi[2 * j] + I * phi[2 * j + 1], psi[2 * j] + I * psi[2 * j + 1], ttm, A, rtol, atol);
        cd lm = 0.0;
        double yk = 1.0;
        for (int k = 0; k < n; ++k) {
            a[2 * (j * n + k)] = creal(A[k]);
            a[2 * (j * n + k) + 1] = cimag(A[k]);
            lm += A[k] * yk;
            yk *= y;
        }
        log_mgf[2 * j] = creal(lm);
        log_mgf[2 * j + 1] = cimag(lm);
    }
}

/* compute_heston_mgf_grid, pricers/heston_pricer.py:183-214 (formula (14) of Sepp 2007) */
void svo_heston_mgf_grid(size_t n_grid, const double *phi, const double *psi, double ttm, double v0, double theta,
                         double kappa, double volvol, double rho, double *a, double *b, int have_t0, double *log_mgf)
{
    const double volvol2 = volvol * volvol;
    for (size_t j = 0; j < n_grid; ++j) {
        cd ph = phi[2 * j] + I * phi[2 * j + 1], ps = psi[2 * j] + I * psi[2 * j + 1];
        cd b1 = kappa + rho * volvol * ph;
        cd b0 = 0.5 * ph * (ph + 1.0) - ps;
        cd zeta = csqrt(b1 * b1 - 2.0 * b0 * volvol2);
        cd exp_zeta = cexp(-zeta * ttm);
        cd psi_p = -b1 + zeta, psi_m = b1 + zeta, c_p, c_m;
        cd a0 = a[2 * j] + I * a[2 * j + 1], bt0 = b[2 * j] + I * b[2 * j + 1];
        if (!have_t0) {
            c_p = psi_p / (2.0 * zeta);
            c_m = psi_m / (2.0 * zeta);
        } else {
            c_p = (psi_p + volvol2 * bt0) / (2.0 * zeta);
            c_m = (psi_m - volvol2 * bt0) / (2.0 * zeta);
        }
        cd b_t1 = -(-psi_m * c_p * exp_zeta + psi_p * c_m) / (volvol2 * (c_p * exp_zeta + c_m));
        cd a_t1 = -(theta * kappa / volvol2) * (psi_p * ttm + 2.0 * clog(c_p * exp_zeta + c_m));
        if (have_t0) a_t1 += a0;
        cd lm = a_t1 + b_t1 * v0;
        a[2 * j] = creal(a_t1); a[2 * j + 1] = cimag(a_t1);
        b[2 * j] = creal(b_t1); b[2 * j + 1] = cimag(b_t1);
        log_mgf[2 * j] = creal(lm); log_mgf[2 * j + 1] = cimag(lm);
    }
}

/* vanilla_slice_pricer_with_mgf_grid, utils/mgf_pricer.py:174-221, for grids with |Re phi| = 1/2 (the only ones the
 * chain pricers build).  Legacy Simpson weights (:158-171): 1,4,2,...; every odd index gets 4 -- including the
 * last one when the grid has an even number of points.  Returns 0, or -1 for an unsupported payoff code. */
int svo_mgf_vanilla_slice(size_t n_grid, const double *phi, const double *log_mgf, double forward, size_t n_strikes,
                          const double *strikes, const int8_t *types, double discfactor, int is_spot_measure,
                          double *prices)
{
    const double PI = 3.14159265358979323846;
    const double h = phi[2 * 1 + 1] - phi[1];
    for (size_t k = 0; k < n_strikes; ++k) {
        const double x = log(forward / strikes[k]);
        double capped = 0.0;
        for (size_t j = 0; j < n_grid; ++j) {
            double w = 2.0;
            if (j == 0 || j == n_grid - 1) w = 1.0;
            if (j % 2 == 1) w = 4.0;
            const double p = phi[2 * j + 1];
            const double pw = ((h / 3.0) * w / PI) / (p * p + 0.25);
            cd ph = phi[2 * j] + I * phi[2 * j + 1], lm = log_mgf[2 * j] + I * log_mgf[2 * j + 1];
            double term = creal(pw * cexp(-x * ph + lm));
            if (term == term) capped += term;                                          /* nansum */
        }
        const int ty = types[k];
        if (is_spot_measure) {
            if (ty == SVO_CALL) prices[k] = discfactor * (forward - strikes[k] * capped);
            else if (ty == SVO_PUT) prices[k] = discfactor * (strikes[k] - strikes[k] * capped);
            else return -1;
        } else {
            if (ty == SVO_INV_CALL || ty == SVO_CALL) prices[k] = forward * discfactor * (1.0 - capped);
            else prices[k] = forward * discfactor * (exp(-x) - capped);
        }
    }
    return 0;
}

/* slice_qvar_pricer_with_a_grid, utils/mgf_pricer.py:322-356: calls on the annualised quadratic variance.
 * Returns 0, or -1 for a payoff code other than 'C' (the reference raises ValueError("not implemented")). */
int svo_mgf_qvar_slice(size_t n_grid, const double *psi, const double *log_mgf, double ttm, size_t n_strikes,
                       const double *strikes, const int8_t *types, double discfactor, double *prices)
{
    const double PI = 3.14159265358979323846;
    const double h = psi[2 * 1 + 1] - psi[1];
    for (size_t k = 0; k < n_strikes; ++k) {
        if (types[k] != SVO_CALL) return -1;
        double sum = 0.0;
        for (size_t j = 0; j < n_grid; ++j) {
            double w = 2.0;
            if (j == 0 || j == n_grid - 1) w = 1.0;
            if (j % 2 == 1) w = 4.0;
            cd ps = psi[2 * j] + I * psi[2 * j + 1], lm = log_mgf[2 * j] + I * log_mgf[2 * j + 1];
            double term = creal((((h / 3.0) * w / PI) / (ps * ps)) * cexp((strikes[k] * ttm) * ps + lm));
            if (term == term) sum += term;
        }
        prices[k] = fmax(discfactor * sum / ttm, 1e-10);                              /* :344 */
    }
    return 0;
}
