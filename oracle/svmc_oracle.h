/*
 * svmc_oracle.h -- CPU restatement (plain C, IEEE fp64, no fast-math, no FMA contraction) of the
 * Monte Carlo hot path of ArturSepp/StochVolModels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under stochvolmodels_amd/ may include, link, import or execute
 * this code; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the timed CPU baseline, never as the product.
 *
 * Parity pinning: every function below is checked against golden vectors produced by importing the
 * unmodified Python reference (NumPy mode) in the build container -- tests/golden/make_golden.py,
 * tests/test_oracle_golden.py.
 *
 * Citations are relative to the reference checkout (src/stochvolmodels/...).
 */
#ifndef SVMC_ORACLE_H
#define SVMC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* option payoff codes, utils/config.py:8-15 (C, P, IC, IP) */
enum { SVO_CALL = 0, SVO_PUT = 1, SVO_INV_CALL = 2, SVO_INV_PUT = 3 };
/* VariableType, utils/config.py:18-24 */
enum { SVO_LOG_RETURN = 1, SVO_Q_VAR = 2, SVO_SIGMA = 3 };
/* Heston discretisation */
enum { SVO_HESTON_EULER_FLOOR = 0, SVO_HESTON_QE = 1 };

int svo_set_threads(int n);

/* utils/funcs.py:24-48  set_time_grid */
void svo_set_time_grid(double ttm, int nb_steps_per_year, int *nb_steps, double *dt);

/* pricers/logsv_pricer.py:950-1047  simulate_logsv_x_vol_terminal, W0/W1 supplied (unscaled N(0,1),
 * step-major [nb_steps][ldw], path p at column p).  State updated in place. */
void svo_logsv_terminal_w(size_t n_path, int nb_steps, double dt,
                          double *x, double *sigma, double *qvar,
                          double theta, double kappa1, double kappa2, double beta, double volvol,
                          double eta, int is_spot_measure,
                          const double *W0, const double *W1, size_t ldw);

/* pricers/heston_pricer.py:334-381  simulate_heston_x_vol_terminal with the normals supplied
 * (unscaled, same layout).  `var` is the variance; returns variance like the reference. */
void svo_heston_terminal_w(size_t n_path, int nb_steps, double dt,
                           double *x, double *var, double *qvar,
                           double theta, double kappa, double rho, double volvol,
                           const double *W0, const double *W1, size_t ldw);

/* Andersen (2008) QE-M scheme for Heston (NOT in the reference).  Z0 drives the
 * log-price, Z1 the quadratic branch of the variance, U the exponential branch. psi_c = 1.5,
 * gamma1 = gamma2 = 0.5.  qvar accumulates the trapezoid of v. */
void svo_heston_qe_terminal_w(size_t n_path, int nb_steps, double dt,
                              double *x, double *var, double *qvar,
                              double theta, double kappa, double rho, double volvol,
                              const double *Z0, const double *Z1, const double *U, size_t ldw);

/* utils/mc_payoffs.py:10-88  compute_mc_vars_payoff.  Returns 0, or -1 unknown payoff code,
 * -2 unsupported variable type (SIGMA). */
int svo_payoff(size_t n_path, const double *x, const double *qvar,
               double ttm, double forward, double discfactor,
               size_t n_strikes, const double *strikes, const int8_t *types, int variable_type,
               double *prices, double *stderrs);

/* ---- counter-based RNG shared (by specification, not by code) with the HIP kernels ------------ */

/* Philox4x32 (Salmon et al., SC'11): the round function at any round count; _10 = the Random123 known-answer setting,
 * the svmc streams use 7 rounds. */
void svo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]);
void svo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* The svmc draw (stream definition version 2, svmc_oracle.c): counter = (path_lo, path_hi, call index, stream |
 * call_id << 8), key = seed.  stream 0 -> the Box-Muller pair (w0, w1) of a time step (call step >> 1, half step & 1);
 * stream 1 -> one uniform in (0,1) (call = step). */
/* one N(0,1) variate from one 32-bit word: the piecewise cubic of the inverse normal CDF (stream version 4) */
double svo_normal_from_word(uint32_t w);
void svo_draw_normals(uint64_t seed, uint32_t call_id, uint64_t path, uint32_t step,
                      double *w0, double *w1);
double svo_draw_uniform(uint64_t seed, uint32_t call_id, uint64_t path, uint32_t step);

/* Materialise the streams the kernels consume: W0/W1[t*ldw + p] for global path ids
 * path_offset + p, global step ids step_offset + t. */
void svo_draw_qe(uint64_t seed, uint32_t call_id, uint64_t path, uint32_t step, double *w0, double *w1, double *u);
void svo_fill_normals_stream(uint64_t seed, uint32_t call_id, uint32_t stream, uint64_t path_offset,
                             uint32_t step_offset, size_t n_path, int nb_steps, double *W0, double *W1, size_t ldw);
void svo_fill_normals(uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset,
                      size_t n_path, int nb_steps, double *W0, double *W1, size_t ldw);
void svo_fill_uniforms(uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset,
                       size_t n_path, int nb_steps, double *U, size_t ldw);

/* Same generators driven by the on-the-fly draw (no materialised arrays). */
void svo_logsv_terminal_rng(size_t n_path, int nb_steps, double dt,
                            double *x, double *sigma, double *qvar,
                            double theta, double kappa1, double kappa2, double beta, double volvol,
                            double eta, int is_spot_measure,
                            uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset);
void svo_heston_terminal_rng(size_t n_path, int nb_steps, double dt,
                             double *x, double *var, double *qvar,
                             double theta, double kappa, double rho, double volvol, int scheme,
                             uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset);

/* pricers/logsv_pricer.py:870-947 simulate_vol_paths.  brownians: SCALED increments [nb_steps][ldb], or NULL for
 * the counter-based draw (stream 2: step t uses component t&1 of the pair of counter step t>>1). */
void svo_logsv_vol_paths(double *sigma_t, size_t ld, size_t n_path, int nb_steps, double dt, double v0,
                         double theta, double kappa1, double kappa2, double beta, double volvol,
                         int is_spot_measure, const double *brownians, size_t ldb,
                         uint64_t seed, uint32_t call_id, uint64_t path_offset);

/* ---- rough LogSV (svmc_oracle_rough.c): pricers/rough_logsv/split_simulation.py:335-356; vol is [n_factors][n_path] */
void svo_rough_logsv_terminal_w(size_t n_path, int nb_steps, double h, int n_factors, const double *nodes,
                                const double *weights, const double *v0, double theta, double kappa1, double kappa2,
                                double rho, double volvol, double *log_s, double *vol, double *y, const double *Z0,
                                const double *Z1, size_t ldw);

/* ---- analytic side (svmc_oracle_analytic.c); complex arrays are interleaved (re, im) like numpy.complex128 ---- */
/* the right-hand side of that ODE on its own (A, out: 5 complex numbers as (re, im) pairs): the check of the device's
 * one-component-per-lane form (stochvolmodels_amd/csrc/svmc_ode.h) */
void svo_logsv_ode_rhs(double theta, double kappa1, double kappa2, double beta, double volvol, int is_spot_measure,
                       int expansion_order, double eta, const double *phi, const double *psi, const double *A, double *out);
void svo_logsv_mgf_grid(size_t n_grid, const double *phi, const double *psi, double ttm, double sigma0, double theta,
                        double kappa1, double kappa2, double beta, double volvol, int is_spot_measure,
                        int expansion_order, double vol_backbone_eta, double *a, double *log_mgf, double rtol, double atol);
void svo_heston_mgf_grid(size_t n_grid, const double *phi, const double *psi, double ttm, double v0, double theta,
                         double kappa, double volvol, double rho, double *a, double *b, int have_t0, double *log_mgf);
int svo_mgf_qvar_slice(size_t n_grid, const double *psi, const double *log_mgf, double ttm, size_t n_strikes,
                       const double *strikes, const int8_t *types, double discfactor, double *prices);
int svo_mgf_vanilla_slice(size_t n_grid, const double *phi, const double *log_mgf, double forward, size_t n_strikes,
                          const double *strikes, const int8_t *types, double discfactor, int is_spot_measure,
                          double *prices);

#ifdef __cplusplus
}
#endif
#endif
