/*
 * svmc_oracle.c -- CPU restatement of the StochVolModels Monte Carlo hot path (see svmc_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP kernels and the timed CPU baseline of bench.py.
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off, strict IEEE fp64 -- the reference's generator is
 * @njit(fastmath=False), pricers/logsv_pricer.py:950).
 *
 * Each arithmetic expression keeps the reference's left-to-right NumPy evaluation order so that, fed
 * the same W0/W1, results agree with the reference to the last bit except for libm-vs-NumPy exp/log
 * rounding (<= 1 ULP per call; tests state 1e-13).
 */
#include "svmc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* thread count of the OpenMP loops (bench.py's all-host-cores CPU baseline); returns the count in effect */
int svo_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * utils/funcs.py:24-48 set_time_grid
 *   nb_steps = int(ttm * nb_steps_per_year) + 1            (:44, truncation of the fp64 product)
 *   grid_t   = np.linspace(0.0, ttm, nb_steps + 1)         (:45)  -> step = ttm / nb_steps,
 *   dt       = grid_t[1] - grid_t[0]                       (:47)     grid_t[1] = 1*step (+0.0),
 *                                                                    or == ttm when nb_steps == 1
 * ---------------------------------------------------------------------------------------------- */
void svo_set_time_grid(double ttm, int nb_steps_per_year, int *nb_steps, double *dt)
{
    int n = (int)(ttm * (double)nb_steps_per_year) + 1;
    *nb_steps = n;
    *dt = (n == 1) ? ttm : ttm / (double)n;
}

/* ------------------------------------------------------------------------------------------------
 * pricers/logsv_pricer.py:1028-1045 (W supplied) -- per step, per path:
 *   W0_ = sdt * W0 ; W1_ = sdt * W1                                             (:1028-1030)
 *   alpha, adj = (-1, 0) | (+1, beta*eta)                                       (:1032-1035)
 *   vartheta2 = beta*beta + volvol*volvol ; eta2 = eta*eta ; L = log(sigma0)    (:1037-1039)
 *   s2dt  = eta2 * sigma * sigma * dt                                           (:1041)
 *   x     = x + alpha*0.5*s2dt + eta*sigma*w0                                   (:1042)
 *   L     = L + ((k1*theta/sigma - k1) + k2*(theta-sigma) + adj*sigma - 0.5*vartheta2)*dt
 *             + beta*w0 + volvol*w1                                             (:1043)
 *   sigma = exp(L)                                                              (:1044)
 *   qvar  = qvar + 0.5*(s2dt + eta2*sigma*sigma*dt)                             (:1045)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    double dt, sdt, theta, kappa1, kappa2, beta, volvol, eta, eta2;
    double alpha_half, adj, half_vartheta2, k1theta;
} logsv_consts;

static logsv_consts logsv_make_consts(double dt, double theta, double kappa1, double kappa2,
                                      double beta, double volvol, double eta, int is_spot_measure)
{
    logsv_consts c;
    double alpha = is_spot_measure ? -1.0 : 1.0;
    c.dt = dt;
    c.sdt = sqrt(dt);
    c.theta = theta;
    c.kappa1 = kappa1;
    c.kappa2 = kappa2;
    c.beta = beta;
    c.volvol = volvol;
    c.eta = eta;
    c.eta2 = eta * eta;
    c.alpha_half = alpha * 0.5;                       /* Python scalar product, evaluated first */
    c.adj = is_spot_measure ? 0.0 : beta * eta;
    c.half_vartheta2 = 0.5 * (beta * beta + volvol * volvol);
    c.k1theta = kappa1 * theta;                       /* scalar product precedes the array divide */
    return c;
}

static inline void logsv_step(const logsv_consts *c, double *x, double *L, double *sigma, double *qvar,
                              double w0, double w1)
{
    double s = *sigma;
    double s2dt = ((c->eta2 * s) * s) * c->dt;
    double drift = ((((c->k1theta / s) - c->kappa1) + c->kappa2 * (c->theta - s)) + c->adj * s)
                   - c->half_vartheta2;
    *x = (*x + c->alpha_half * s2dt) + (c->eta * s) * w0;
    *L = ((*L + drift * c->dt) + c->beta * w0) + c->volvol * w1;
    s = exp(*L);
    *sigma = s;
    *qvar = *qvar + 0.5 * (s2dt + ((c->eta2 * s) * s) * c->dt);
}

void svo_logsv_terminal_w(size_t n_path, int nb_steps, double dt,
                          double *x, double *sigma, double *qvar,
                          double theta, double kappa1, double kappa2, double beta, double volvol,
                          double eta, int is_spot_measure,
                          const double *W0, const double *W1, size_t ldw)
{
    logsv_consts c = logsv_make_consts(dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure);
    double *L = (double *)malloc(n_path * sizeof(double));
    for (size_t p = 0; p < n_path; ++p) L[p] = log(sigma[p]);
    /* step-major like the reference: one pass over all paths per time step */
    for (int t = 0; t < nb_steps; ++t) {
        const double *w0 = W0 + (size_t)t * ldw;
        const double *w1 = W1 + (size_t)t * ldw;
        for (size_t p = 0; p < n_path; ++p)
            logsv_step(&c, &x[p], &L[p], &sigma[p], &qvar[p], c.sdt * w0[p], c.sdt * w1[p]);
    }
    free(L);
}

/* ------------------------------------------------------------------------------------------------
 * pricers/heston_pricer.py:368-379
 *   w0 = sqrt(dt)*N ; w1 = sqrt(dt)*N ; rho_1 = sqrt(1-rho*rho)
 *   sigma = sqrt(v) ; s2dt = v*dt
 *   x    = x - 0.5*s2dt + sigma*w0
 *   qvar = qvar + s2dt
 *   v    = v + kappa*(theta-v)*dt + sigma*volvol*(rho*w0 + rho_1*w1)
 *   v    = max(v, 1e-4)
 * ---------------------------------------------------------------------------------------------- */
static inline void heston_euler_step(double dt, double theta, double kappa, double rho, double rho_1,
                                     double volvol, double *x, double *v, double *qvar,
                                     double w0, double w1)
{
    double var = *v;
    double s = sqrt(var);
    double s2dt = var * dt;
    *x = (*x - 0.5 * s2dt) + s * w0;
    *qvar = *qvar + s2dt;
    var = (var + (kappa * (theta - var)) * dt) + (s * volvol) * (rho * w0 + rho_1 * w1);
    *v = (var > 1e-4) ? var : ((var != var) ? var : 1e-4);   /* np.maximum propagates NaN */
}

void svo_heston_terminal_w(size_t n_path, int nb_steps, double dt,
                           double *x, double *var, double *qvar,
                           double theta, double kappa, double rho, double volvol,
                           const double *W0, const double *W1, size_t ldw)
{
    double sdt = sqrt(dt), rho_1 = sqrt(1.0 - rho * rho);
    for (int t = 0; t < nb_steps; ++t) {
        const double *w0 = W0 + (size_t)t * ldw;
        const double *w1 = W1 + (size_t)t * ldw;
        for (size_t p = 0; p < n_path; ++p)
            heston_euler_step(dt, theta, kappa, rho, rho_1, volvol, &x[p], &var[p], &qvar[p],
                              sdt * w0[p], sdt * w1[p]);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Andersen QE-M (L. Andersen, "Simple and efficient simulation of the Heston model", J. Comp. Fin.
 * 11(3), 2008, sections 3.2.4, 4.2, 4.3.2).  New capability relative to the reference, whose only
 * Heston scheme is the floored Euler above; its oracle is the reference's analytic Heston price.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    double dt, theta, E, c1, c2, K1, K2, K3, K4, A, K0_plain, K13;
} qe_consts;

static qe_consts qe_make_consts(double dt, double theta, double kappa, double rho, double volvol)
{
    qe_consts c;
    const double g1 = 0.5, g2 = 0.5;
    double E = exp(-kappa * dt);
    double kre = kappa * rho / volvol - 0.5;
    c.dt = dt;
    c.theta = theta;
    c.E = E;
    c.c1 = volvol * volvol * E * (1.0 - E) / kappa;                 /* s2 = v*c1 + c2 */
    c.c2 = theta * volvol * volvol * (1.0 - E) * (1.0 - E) / (2.0 * kappa);
    c.K1 = g1 * dt * kre - rho / volvol;
    c.K2 = g2 * dt * kre + rho / volvol;
    c.K3 = g1 * dt * (1.0 - rho * rho);
    c.K4 = g2 * dt * (1.0 - rho * rho);
    c.A = c.K2 + 0.5 * c.K4;
    c.K0_plain = -rho * kappa * theta / volvol * dt;
    c.K13 = c.K1 + 0.5 * c.K3;
    return c;
}

static inline void heston_qe_step(const qe_consts *c, double *x, double *v, double *qvar,
                                  double z0, double z1, double u)
{
    double v0 = *v, v1, K0;
    double m = c->theta + (v0 - c->theta) * c->E;
    double s2 = v0 * c->c1 + c->c2;
    double psi = s2 / (m * m);
    if (psi <= 1.5) {
        double ip = 2.0 / psi;
        double b2 = ip - 1.0 + sqrt(ip * (ip - 1.0));
        double a = m / (1.0 + b2);
        double b = sqrt(b2);
        double den = 1.0 - 2.0 * c->A * a;
        v1 = a * (b + z1) * (b + z1);
        K0 = (den > 0.0) ? (-c->A * b2 * a / den + 0.5 * log(den) - c->K13 * v0) : c->K0_plain;
    } else {
        double p = (psi - 1.0) / (psi + 1.0);
        double bt = (1.0 - p) / m;
        v1 = (u <= p) ? 0.0 : log((1.0 - p) / (1.0 - u)) / bt;
        K0 = (c->A < bt) ? (-log(p + bt * (1.0 - p) / (bt - c->A)) - c->K13 * v0) : c->K0_plain;
    }
    *x = *x + K0 + c->K1 * v0 + c->K2 * v1 + sqrt(c->K3 * v0 + c->K4 * v1) * z0;
    *qvar = *qvar + 0.5 * c->dt * (v0 + v1);
    *v = v1;
}

void svo_heston_qe_terminal_w(size_t n_path, int nb_steps, double dt,
                              double *x, double *var, double *qvar,
                              double theta, double kappa, double rho, double volvol,
                              const double *Z0, const double *Z1, const double *U, size_t ldw)
{
    qe_consts c = qe_make_consts(dt, theta, kappa, rho, volvol);
    for (int t = 0; t < nb_steps; ++t) {
        const double *z0 = Z0 + (size_t)t * ldw, *z1 = Z1 + (size_t)t * ldw, *u = U + (size_t)t * ldw;
        for (size_t p = 0; p < n_path; ++p)
            heston_qe_step(&c, &x[p], &var[p], &qvar[p], z0[p], z1[p], u[p]);
    }
}

/* ------------------------------------------------------------------------------------------------
 * utils/mc_payoffs.py:61-88 compute_mc_vars_payoff
 *   spots = F*exp(x); c = nanmean(spots) - F; spots -= c                      (:61-63)
 *   u = spots | qvar/ttm | NotImplementedError                                (:65-70)
 *   C : where(u>K, u-K, 0)   P : where(u<K, K-u, 0)   IC/IP: same / spots     (:75-82)
 *   price = DF*nanmean(payoff); std = DF*nanstd(payoff)  (ddof 0)             (:85-86)
 *   return price, std/sqrt(len(x))                                            (:88)
 * Sums are pairwise (as NumPy's are), so rounding stays O(log n) ULP.
 * ---------------------------------------------------------------------------------------------- */
static double pairwise_sum(const double *a, size_t n)
{
    if (n <= 8) {
        double s = 0.0;
        for (size_t i = 0; i < n; ++i) s += a[i];
        return s;
    }
    size_t h = n / 2;
    return pairwise_sum(a, h) + pairwise_sum(a + h, n - h);
}

int svo_payoff(size_t n_path, const double *x, const double *qvar,
               double ttm, double forward, double discfactor,
               size_t n_strikes, const double *strikes, const int8_t *types, int variable_type,
               double *prices, double *stderrs)
{
    if (variable_type != SVO_LOG_RETURN && variable_type != SVO_Q_VAR) return -2;
    for (size_t k = 0; k < n_strikes; ++k)
        if (types[k] < SVO_CALL || types[k] > SVO_INV_PUT) return -1;

    double *spots = (double *)malloc(n_path * sizeof(double));
    double *buf = (double *)calloc(n_path ? n_path : 1, sizeof(double));
    size_t cnt = 0;
    for (size_t p = 0; p < n_path; ++p) {
        spots[p] = forward * exp(x[p]);
        if (spots[p] == spots[p]) buf[cnt++] = spots[p];
    }
    double corr = pairwise_sum(buf, cnt) / (double)cnt - forward;   /* nanmean; 0/0 -> NaN like NumPy */
    for (size_t p = 0; p < n_path; ++p) spots[p] = spots[p] - corr;

    for (size_t k = 0; k < n_strikes; ++k) {
        double K = strikes[k];
        int ty = types[k];
        cnt = 0;
        for (size_t p = 0; p < n_path; ++p) {
            double u = (variable_type == SVO_LOG_RETURN) ? spots[p] : qvar[p] / ttm;
            double pay;
            if (ty == SVO_CALL || ty == SVO_INV_CALL) pay = (u > K) ? (u - K) : 0.0;
            else pay = (u < K) ? (K - u) : 0.0;
            if (ty == SVO_INV_CALL || ty == SVO_INV_PUT) pay = pay / spots[p];
            if (pay == pay) buf[cnt++] = pay;
        }
        double mean = pairwise_sum(buf, cnt) / (double)cnt;
        for (size_t i = 0; i < cnt; ++i) buf[i] = (buf[i] - mean) * (buf[i] - mean);
        double var = pairwise_sum(buf, cnt) / (double)cnt;
        prices[k] = discfactor * mean;
        stderrs[k] = discfactor * sqrt(var) / sqrt((double)n_path);
    }
    free(spots);
    free(buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Counter-based randoms.  Philox4x32: J. Salmon, M. Moraes, R. Dror, D. Shaw, "Parallel random
 * numbers: as easy as 1, 2, 3", SC'11 (constants and round function from the paper; the round function is
 * checked at its 10-round setting against the Random123 known-answer vectors in tests/test_oracle_golden.py).
 * The svmc streams use SEVEN rounds, the smallest Crush-resistant count the paper reports for 4x32.
 *
 * svmc stream definition, version 4 (DESIGN.md section "RNG"; device twin stochvolmodels_amd/csrc/svmc_rng.h).
 * One call yields the two normals of TWO consecutive time steps, each 32-bit word turned into ONE N(0,1) variate by
 * inversion:
 *   key = (seed_lo, seed_hi);  ctr = (path_lo, path_hi, step >> 1, stream | call_id << 8)
 *   r0..r3 = philox4x32_7(ctr, key);  (ra, rb) = (r0, r1) for an even step, (r2, r3) for an odd one
 *   z(r):  t = (int32) r (version 3: + 1/2; here z = 0 at t = 0 and |t| = 2^31, every other magnitude with both signs);
 *          j = the segment of |t| -- (low 5 bits of the biased fp64
 *          exponent) << M | (top M mantissa bits), 32 octaves x 2^M equal parts;
 *          z = sign(t) * fma(fma(fma(a3, |t|, a2), |t|, a1), |t|, a0)  with {a0..a3}[j] from svo_icdf_table.h, the
 *          piecewise cubic of -Phi^-1(|t| 2^-32) generated by tools/gen_icdf_table.py (the same bytes as the product's
 *          csrc/svmc_icdf_table.h; tests/test_oracle_golden.py pins the table against scipy's Phi^-1 at
 *          SVMC_ICDF_MAX_ABS_ERROR).  This evaluation order, FMAs included, IS the definition of the stream.
 *   streams 0, 3:  (w0, w1) = (z(ra), z(rb))
 *   stream 1:  one call per draw, uniform = ((r0 | r1<<32) >> 12) 2^-52 + 2^-53
 *   stream 2 (vol paths, one Brownian per step): normal t = z(word t & 3 of call t >> 2)
 *   Heston QE: normals from stream 4 (as stream 0), uniform (r[step & 3] + 1/2) 2^-32 of stream 5's call step >> 2
 * ---------------------------------------------------------------------------------------------- */
#define SVO_PHILOX_ROUNDS 7

void svo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4])
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void svo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    svo_philox4x32(ctr, key, 10, out);
}

static inline double m52(uint32_t lo, uint32_t hi)
{
    return (double)((((uint64_t)hi << 32) | lo) >> 12) * 0x1.0p-52;   /* in [0,1), exact */
}

static inline void philox_draw(uint64_t seed, uint32_t call_id, uint64_t path, uint32_t index, uint32_t stream,
                               uint32_t r[4])
{
    uint32_t ctr[4] = { (uint32_t)path, (uint32_t)(path >> 32), index, stream | (call_id << 8) };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    svo_philox4x32(ctr, key, SVO_PHILOX_ROUNDS, r);
}

#include "svo_icdf_table.h"
#if !SVMC_ICDF_RAW || SVMC_ICDF_DEG != 3
#error "the oracle restates the raw-form cubic table (coefficients in powers of |t|)"
#endif
static const double svo_icdf_p0[SVMC_ICDF_SEGMENTS][2] = { SVMC_ICDF_PIECE0_INIT };    /* {a0, a1} */
static const double svo_icdf_p1[SVMC_ICDF_SEGMENTS][2] = { SVMC_ICDF_PIECE1_INIT };    /* {a2, a3} */

/* one N(0,1) variate from one word (svmc_math.h normal_icdf32) */
double svo_normal_from_word(uint32_t w)
{
#if SVMC_ICDF_HALF_LATTICE
    const double t = (double)(int32_t)w + 0.5;        /* stream version 3's lattice */
#else
    const double t = (double)(int32_t)w;              /* stream version 4: the signed word itself (z = 0 at t = 0 and |t| = 2^31) */
#endif
    uint64_t bits;
    memcpy(&bits, &t, 8);
    const uint32_t hi = (uint32_t)(bits >> 32);
    const uint32_t j = (hi >> (20 - SVMC_ICDF_M)) & (SVMC_ICDF_SEGMENTS - 1u);
    const double a = fabs(t);
    double p = fma(svo_icdf_p1[j][1], a, svo_icdf_p1[j][0]);
    p = fma(p, a, svo_icdf_p0[j][1]);
    p = fma(p, a, svo_icdf_p0[j][0]);
    return copysign(p, t);
}

/* the two normals of one (ra, rb) word pair */
static inline void pair_from_words(uint32_t ra, uint32_t rb, double *w0, double *w1)
{
    *w0 = svo_normal_from_word(ra);
    *w1 = svo_normal_from_word(rb);
}

static void draw_normals_stream(uint64_t seed, uint32_t call_id, uint64_t path, uint32_t step, uint32_t stream,
                                double *w0, double *w1)
{
    uint32_t r[4];
    philox_draw(seed, call_id, path, step >> 1, stream, r);
    if (step & 1u) pair_from_words(r[2], r[3], w0, w1);
    else pair_from_words(r[0], r[1], w0, w1);
}

/* the same pairs for a run of consecutive steps of ONE path: the call of steps 2c, 2c + 1 is evaluated once and kept
 * (the full-size parity tests and bench.py's all-cores leg walk 10^9 path-steps through here) */
typedef struct {
    uint32_t r[4];
    uint32_t call;
    int valid;
} pair_cache;

static inline void draw_normals_cached(pair_cache *pc, uint64_t seed, uint32_t call_id, uint64_t path, uint32_t step,
                                       uint32_t stream, double *w0, double *w1)
{
    if (!pc->valid || pc->call != (step >> 1)) {
        philox_draw(seed, call_id, path, step >> 1, stream, pc->r);
        pc->call = step >> 1;
        pc->valid = 1;
    }
    if (step & 1u) pair_from_words(pc->r[2], pc->r[3], w0, w1);
    else pair_from_words(pc->r[0], pc->r[1], w0, w1);
}

void svo_draw_normals(uint64_t seed, uint32_t call_id, uint64_t path, uint32_t step,
                      double *w0, double *w1)
{
    draw_normals_stream(seed, call_id, path, step, 0u, w0, w1);
}

/* Heston QE: the pair from stream 4 (like stream 0: one call per two steps), the exponential branch's uniform from
 * stream 5 (word step & 3 of call step >> 2) -- device twins rng_time_loop / qe_uniform, csrc/svmc_rng.h */
void svo_draw_qe(uint64_t seed, uint32_t call_id, uint64_t path, uint32_t step, double *w0, double *w1, double *u)
{
    uint32_t r[4];
    draw_normals_stream(seed, call_id, path, step, 4u, w0, w1);
    philox_draw(seed, call_id, path, step >> 2, 5u, r);
    *u = ((double)r[step & 3u] + 0.5) * 0x1.0p-32;
}

double svo_draw_uniform(uint64_t seed, uint32_t call_id, uint64_t path, uint32_t step)
{
    uint32_t r[4];
    philox_draw(seed, call_id, path, step, 1u, r);
    return m52(r[0], r[1]) + 0x1.0p-53;
}

void svo_fill_normals(uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset,
                      size_t n_path, int nb_steps, double *W0, double *W1, size_t ldw)
{
    for (int t = 0; t < nb_steps; ++t)
        for (size_t p = 0; p < n_path; ++p)
            svo_draw_normals(seed, call_id, path_offset + p, step_offset + (uint32_t)t,
                             &W0[(size_t)t * ldw + p], &W1[(size_t)t * ldw + p]);
}

/* same generator on a named stream tag (3 = the rough-LogSV kernel's device-side normals) */
void svo_fill_normals_stream(uint64_t seed, uint32_t call_id, uint32_t stream, uint64_t path_offset,
                             uint32_t step_offset, size_t n_path, int nb_steps, double *W0, double *W1, size_t ldw)
{
    for (int t = 0; t < nb_steps; ++t)
        for (size_t p = 0; p < n_path; ++p)
            draw_normals_stream(seed, call_id, path_offset + p, step_offset + (uint32_t)t, stream,
                                &W0[(size_t)t * ldw + p], &W1[(size_t)t * ldw + p]);
}

void svo_fill_uniforms(uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset,
                       size_t n_path, int nb_steps, double *U, size_t ldw)
{
    for (int t = 0; t < nb_steps; ++t)
        for (size_t p = 0; p < n_path; ++p)
            U[(size_t)t * ldw + p] = svo_draw_uniform(seed, call_id, path_offset + p, step_offset + (uint32_t)t);
}

void svo_logsv_terminal_rng(size_t n_path, int nb_steps, double dt,
                            double *x, double *sigma, double *qvar,
                            double theta, double kappa1, double kappa2, double beta, double volvol,
                            double eta, int is_spot_measure,
                            uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset)
{
    logsv_consts c = logsv_make_consts(dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure);
    /* paths are independent: with -fopenmp this loop is the all-host-cores CPU baseline of bench.py */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t p = 0; p < n_path; ++p) {
        double xp = x[p], sp = sigma[p], qp = qvar[p], L = log(sp), w0, w1;
        pair_cache pc = { {0u, 0u, 0u, 0u}, 0u, 0 };
        for (int t = 0; t < nb_steps; ++t) {
            draw_normals_cached(&pc, seed, call_id, path_offset + p, step_offset + (uint32_t)t, 0u, &w0, &w1);
            logsv_step(&c, &xp, &L, &sp, &qp, c.sdt * w0, c.sdt * w1);
        }
        x[p] = xp; sigma[p] = sp; qvar[p] = qp;
    }
}

void svo_heston_terminal_rng(size_t n_path, int nb_steps, double dt,
                             double *x, double *var, double *qvar,
                             double theta, double kappa, double rho, double volvol, int scheme,
                             uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset)
{
    const double sdt = sqrt(dt), rho_1 = sqrt(1.0 - rho * rho);
    const qe_consts c = qe_make_consts(dt, theta, kappa, rho, volvol);
    /* paths are independent: OpenMP over paths for the full-size parity tests (tests/test_gpu_fullsize.py) */
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t p = 0; p < n_path; ++p) {
        double xp = x[p], vp = var[p], qp = qvar[p], w0, w1;
        pair_cache pc = { {0u, 0u, 0u, 0u}, 0u, 0 };
        for (int t = 0; t < nb_steps; ++t) {
            uint32_t step = step_offset + (uint32_t)t;
            if (scheme == SVO_HESTON_QE) {
                uint32_t r[4];
                draw_normals_cached(&pc, seed, call_id, path_offset + p, step, 4u, &w0, &w1);
                philox_draw(seed, call_id, path_offset + p, step >> 2, 5u, r);
                heston_qe_step(&c, &xp, &vp, &qp, w0, w1, ((double)r[step & 3u] + 0.5) * 0x1.0p-32);
            } else {
                draw_normals_cached(&pc, seed, call_id, path_offset + p, step, 0u, &w0, &w1);
                heston_euler_step(dt, theta, kappa, rho, rho_1, volvol, &xp, &vp, &qp, sdt * w0, sdt * w1);
            }
        }
        x[p] = xp; var[p] = vp; qvar[p] = qp;
    }
}

/* ------------------------------------------------------------------------------------------------
 * pricers/logsv_pricer.py:920-947 simulate_vol_paths
 *   sigma0 = v0 ; L = log(sigma0) ; sigma_t[0] = sigma0
 *   brownians = sqrt(dt) * N(0,1)   (when not supplied)                          (:925)
 *   adj = 0 | beta ; vartheta = sqrt(beta^2 + volvol^2)                          (:930-936)
 *   L = L + ((k1*theta/sigma - k1) + k2*(theta-sigma) + adj*sigma - 0.5*vartheta2)*dt + vartheta*w   (:942)
 *   sigma = exp(L) ; sigma_t[t+1] = sigma                                        (:943-944)
 * ---------------------------------------------------------------------------------------------- */
void svo_logsv_vol_paths(double *sigma_t, size_t ld, size_t n_path, int nb_steps, double dt, double v0,
                         double theta, double kappa1, double kappa2, double beta, double volvol,
                         int is_spot_measure, const double *brownians, size_t ldb,
                         uint64_t seed, uint32_t call_id, uint64_t path_offset)
{
    double adj = is_spot_measure ? 0.0 : beta;
    double vartheta2 = beta * beta + volvol * volvol, vartheta = sqrt(vartheta2), sdt = sqrt(dt);
    double k1theta = kappa1 * theta;
    for (size_t p = 0; p < n_path; ++p) {
        double s = v0, L = log(v0), z0 = 0.0, z1 = 0.0;
        sigma_t[p] = s;
        for (int t = 0; t < nb_steps; ++t) {
            double w;
            if (brownians) {
                w = brownians[(size_t)t * ldb + p];
            } else {
                if ((t & 1) == 0) draw_normals_stream(seed, call_id, path_offset + p, (uint32_t)(t >> 1), 2u, &z0, &z1);   /* = words (t & 2), (t & 2) + 1 of call t >> 2 */
                w = sdt * ((t & 1) ? z1 : z0);
            }
            L = (L + ((((k1theta / s) - kappa1) + kappa2 * (theta - s)) + adj * s - 0.5 * vartheta2) * dt) + vartheta * w;
            s = exp(L);
            sigma_t[(size_t)(t + 1) * ld + p] = s;
        }
    }
}
