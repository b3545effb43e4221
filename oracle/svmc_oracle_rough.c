/*
 * svmc_oracle_rough.c -- CPU restatement of the rough-LogSV (Markovian lift, N <= 3 factors) terminal-state
 * simulator: SURVEY.md row f.4.  TEST INFRASTRUCTURE ONLY (see svmc_oracle.h).
 *
 *   pricers/rough_logsv/split_simulation.py:86-128   drift_ode_solve2          (RK4 on the factor drift)
 *   pricers/rough_logsv/split_simulation.py:228-246  diffus_sde_solve_f64      (exact lognormal step of the weighted sum)
 *   pricers/rough_logsv/split_simulation.py:249-278  drift_diffus_strand_f64   (Strang: D(h/2) S(h) D(h/2))
 *   pricers/rough_logsv/split_simulation.py:281-332  log_spot_full_solve2_f64  (log-spot / quadratic-variance update)
 *   pricers/rough_logsv/split_simulation.py:335-356  log_spot_full_combined_f64 (time loop)
 *
 * The reference is @njit(fastmath=True) NumPy array code; here one path at a time, same expression order.
 * Pinned by the reference's own committed regression vector
 * (tests/test_rough_logsv_pricer_regression/test_rough_logsv_pricer_pricing_regression.npz, rtol 1e-7).
 */
#include "svmc_oracle.h"

#include <math.h>

#define NMAX 3

static void rk4_drift(int n, const double *nodes, const double *w, const double *v0, double theta, double kappa1,
                      double kappa2, const double *z0, double h, double *zh)
{
    double s1[NMAX], s2[NMAX], s3[NMAX], s4[NMAX], zt[NMAX], zw, c;
    zw = 0.0;
    for (int i = 0; i < n; ++i) zw += w[i] * z0[i];
    c = (kappa1 + kappa2 * zw) * (theta - zw);
    for (int i = 0; i < n; ++i) s1[i] = -nodes[i] * (z0[i] - v0[i]) + c;
    zw = 0.0;
    for (int i = 0; i < n; ++i) { zt[i] = z0[i] + 0.5 * h * s1[i]; zw += w[i] * zt[i]; }
    c = (kappa1 + kappa2 * zw) * (theta - zw);
    for (int i = 0; i < n; ++i) s2[i] = -nodes[i] * (zt[i] - v0[i]) + c;
    zw = 0.0;
    for (int i = 0; i < n; ++i) { zt[i] = z0[i] + 0.5 * h * s2[i]; zw += w[i] * zt[i]; }
    c = (kappa1 + kappa2 * zw) * (theta - zw);
    for (int i = 0; i < n; ++i) s3[i] = -nodes[i] * (zt[i] - v0[i]) + c;
    zw = 0.0;
    for (int i = 0; i < n; ++i) { zt[i] = z0[i] + h * s3[i]; zw += w[i] * zt[i]; }
    c = (kappa1 + kappa2 * zw) * (theta - zw);
    for (int i = 0; i < n; ++i) s4[i] = -nodes[i] * (zt[i] - v0[i]) + c;
    for (int i = 0; i < n; ++i) zh[i] = z0[i] + (h / 6.0) * (s1[i] + 2.0 * s2[i] + 2.0 * s3[i] + s4[i]);
}

/* in-place advance of (log_s, vol[N][n_path], y) over nb_steps of size h; Z0 drives the factors, Z1 the spot */
void svo_rough_logsv_terminal_w(size_t n_path, int nb_steps, double h, int n_factors, const double *nodes,
                                const double *weights, const double *v0, double theta, double kappa1, double kappa2,
                                double rho, double volvol, double *log_s, double *vol, double *y, const double *Z0,
                                const double *Z1, size_t ldw)
{
    const int n = n_factors;
    double wsum = 0.0, w_lam_v0 = 0.0, wlam[NMAX];
    for (int i = 0; i < n; ++i) { wsum += weights[i]; wlam[i] = weights[i] * nodes[i]; w_lam_v0 += wlam[i] * v0[i]; }
    const double w_inv = 1.0 / wsum, volvol_ = volvol * wsum, rho_comp = sqrt(1.0 - rho * rho), sqrt_h = sqrt(h);
    for (size_t p = 0; p < n_path; ++p) {
        double v[NMAX], ls = log_s[p], yy = y[p];
        for (int i = 0; i < n; ++i) v[i] = vol[(size_t)i * n_path + p];
        for (int t = 0; t < nb_steps; ++t) {
            const double z0 = Z0[(size_t)t * ldw + p], z1 = Z1[(size_t)t * ldw + p];
            double d[NMAX], sn[NMAX], vh[NMAX];
            rk4_drift(n, nodes, weights, v0, theta, kappa1, kappa2, v, 0.5 * h, d);                 /* :273 */
            double yw = 0.0;
            for (int i = 0; i < n; ++i) yw += weights[i] * d[i];
            const double dW = z0 * sqrt_h;
            const double Yh = yw * exp(-0.5 * volvol_ * volvol_ * h + volvol_ * dW);               /* :238-239 */
            const double Q = 1.0 / wsum * (Yh - yw);
            for (int i = 0; i < n; ++i) sn[i] = d[i] + Q;
            rk4_drift(n, nodes, weights, v0, theta, kappa1, kappa2, sn, 0.5 * h, vh);               /* :275 */
            double volw_h = 0.0;
            for (int i = 0; i < n; ++i) volw_h += weights[i] * vh[i];
            if (!(volw_h > 0.0)) {                                                                  /* NaN or <= 0, :299-300 */
                for (int i = 0; i < n; ++i) vh[i] = 1e-6;
                volw_h = 0.0;
                for (int i = 0; i < n; ++i) volw_h += weights[i] * vh[i];
            }
            double vw = 0.0, w_lam_vol = 0.0, w_lam_vol_h = 0.0;
            for (int i = 0; i < n; ++i) { vw += weights[i] * v[i]; w_lam_vol += wlam[i] * v[i]; w_lam_vol_h += wlam[i] * vh[i]; }
            const double sq_vw = vw * vw, sq_vhw = volw_h * volw_h;
            const double term1 = 1.0 / volvol * (((volw_h - vw) / h + 0.5 * w_lam_vol + 0.5 * w_lam_vol_h - w_lam_v0) * w_inv
                                                 - kappa1 * theta + (kappa1 - kappa2 * theta) * (0.5 * vw + 0.5 * volw_h)
                                                 + kappa2 * (0.5 * sq_vw + 0.5 * sq_vhw)) * h;       /* :319-321 */
            const double term2 = 0.5 * h * sq_vw + 0.5 * h * sq_vhw;
            ls = ls - 0.5 * term2 + rho * term1 + rho_comp * sqrt(term2) * z1;                      /* :324 */
            yy = yy + 0.5 * h * (vw * vw + volw_h * volw_h);                                        /* :326 */
            for (int i = 0; i < n; ++i) v[i] = vh[i];
        }
        for (int i = 0; i < n; ++i) vol[(size_t)i * n_path + p] = v[i];
        log_s[p] = ls;
        y[p] = yy;
    }
}
