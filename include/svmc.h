/*
 * svmc.h -- C ABI of libsvmc.so, the MI355X (gfx950) Monte Carlo engine behind the StochVolModels
 * hot path (LogSV / Heston terminal-state generators + forward-recentred payoff reduction).
 *
 * The reference (ArturSepp/StochVolModels, pure Python + Numba) has no FFI of its own; each entry point
 * below names the reference function it replaces (paths relative to src/stochvolmodels/).  The binding
 * a maintainer of the reference would add is a ctypes stub: see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns an svmc status code (0 = SVMC_OK); svmc_last_error() gives the message of
 *     the last failure on the calling thread.  No exceptions cross the boundary.
 *   - all `double *` state / random / output pointers are DEVICE pointers on the current HIP device
 *     (svmc_set_device) unless the parameter name ends in `_host`.  Buffers are caller-owned.
 *   - every compute call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default
 *     stream).  Outputs are valid after svmc_stream_synchronize(stream).
 *   - all arithmetic is IEEE fp64.  Option payoff codes are int8: C=0, P=1, IC=2, IP=3
 *     (utils/config.py:8-15); variable types are 1=LOG_RETURN, 2=Q_VAR, 3=SIGMA (utils/config.py:18-24).
 *   - random layout is the reference's: W[t * ldw + p], step-major, UNSCALED N(0,1)
 *     (pricers/logsv_pricer.py:1028-1030).
 *   - the library is thread-compatible: concurrent calls must use distinct streams and buffers.
 */
#ifndef SVMC_H
#define SVMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVMC_VERSION 100 /* 0.1.0 */

#if defined(__GNUC__) || defined(__clang__)
#define SVMC_API __attribute__((visibility("default")))
#else
#define SVMC_API
#endif

/* status codes */
#define SVMC_OK 0
#define SVMC_ERR_INVALID_ARGUMENT 1
#define SVMC_ERR_HIP 2
#define SVMC_ERR_UNKNOWN_PAYOFF 3       /* reference: ValueError("unknown option payoff code"), utils/mc_payoffs.py:84 */
#define SVMC_ERR_UNSUPPORTED_VARIABLE 4 /* reference: NotImplementedError for VariableType.SIGMA, utils/mc_payoffs.py:69-70 */
#define SVMC_ERR_WORKSPACE 5
#define SVMC_ERR_RCCL 6                 /* RCCL missing at run time, or a collective / communicator call failed */

#define SVMC_CALL 0
#define SVMC_PUT 1
#define SVMC_INV_CALL 2
#define SVMC_INV_PUT 3

#define SVMC_LOG_RETURN 1
#define SVMC_Q_VAR 2
#define SVMC_SIGMA 3

#define SVMC_HESTON_EULER_FLOOR 0 /* the reference's scheme, pricers/heston_pricer.py:373-379 */
#define SVMC_HESTON_QE 1          /* Andersen QE-M; new capability */

typedef void *svmc_stream_t;  /* hipStream_t */
typedef void *svmc_session_t; /* opaque: one GPU's resident chain-pricing buffers */
typedef void *svmc_event_t;  /* hipEvent_t */
typedef void *svmc_comm_t;   /* ncclComm_t (RCCL) */

/* ---- library / device plumbing (no reference counterpart: the reference is single-process NumPy) -- */
SVMC_API int svmc_version(void);
/* The version of the counter-based random stream the generators draw from (SVMC_RNG_STREAM_VERSION): results for a
 * given seed are reproducible only within one stream version.  1 = Philox4x32-10 + 52-bit Box-Muller (round 1);
 * 2 = Philox4x32-7, one call per two steps, 32-bit radius / angle Box-Muller (round 2); 3 = Philox4x32-7, one call per two
 * steps, each 32-bit word turned into one normal by a piecewise-cubic inverse CDF on the lattice (int32) word + 1/2 (round 3);
 * 4 = the same on the lattice (int32) word itself -- one instruction less per normal, still exactly symmetric (round 6).
 * CHANGELOG.md. */
#define SVMC_RNG_STREAM_VERSION 4
SVMC_API int svmc_rng_stream_version(void);
SVMC_API const char *svmc_last_error(void);
SVMC_API int svmc_device_count(int *count);
SVMC_API int svmc_set_device(int device);
SVMC_API int svmc_get_device(int *device);
SVMC_API int svmc_device_info(int device, char *name, size_t name_len, int *compute_units, int *clock_khz,
                     size_t *hbm_bytes);
SVMC_API int svmc_malloc(void **dptr, size_t bytes);
SVMC_API int svmc_free(void *dptr);
SVMC_API int svmc_host_alloc(void **hptr, size_t bytes); /* pinned host memory for async H2D/D2H */
SVMC_API int svmc_host_free(void *hptr);
SVMC_API int svmc_memset(void *dptr, int value, size_t bytes, svmc_stream_t stream);
SVMC_API int svmc_memcpy_h2d(void *dst, const void *src_host, size_t bytes, svmc_stream_t stream);
SVMC_API int svmc_memcpy_d2h(void *dst_host, const void *src, size_t bytes, svmc_stream_t stream);
SVMC_API int svmc_memcpy_d2d(void *dst, const void *src, size_t bytes, svmc_stream_t stream);
/* strided upload of a [height][width] double sub-matrix (a rank's path range of a host W array) */
SVMC_API int svmc_memcpy2d_h2d(void *dst, size_t dst_pitch_bytes, const void *src_host, size_t src_pitch_bytes,
                      size_t width_bytes, size_t height, svmc_stream_t stream);
SVMC_API int svmc_stream_create(svmc_stream_t *stream);
SVMC_API int svmc_stream_destroy(svmc_stream_t stream);
SVMC_API int svmc_stream_synchronize(svmc_stream_t stream);
SVMC_API int svmc_event_create(svmc_event_t *event);
SVMC_API int svmc_event_destroy(svmc_event_t event);
SVMC_API int svmc_event_record(svmc_event_t event, svmc_stream_t stream);
SVMC_API int svmc_event_synchronize(svmc_event_t event);
SVMC_API int svmc_event_elapsed_ms(svmc_event_t start, svmc_event_t stop, float *ms); /* synchronises on `stop` */
/* The shader clock an on-device-RNG LogSV stepping launch (svmc_logsv_*_rng*, svmc_logsv_vol_paths) ran at, measured inside
 * that kernel (no reference counterpart: measurement plumbing of bench.py's roofline).  OFF by default -- the
 * kernels take a null probe pointer and do nothing.  svmc_clock_probe_arm(1) arms the CALLING THREAD: it gets its own 8-word
 * device buffer (on the current device) and its later launches hand it to the kernel, where thread 0 of the launch's first
 * and last block stamp s_memtime (tick = shader cycle) and s_memrealtime (100 MHz) at kernel entry and after the time loop.
 * svmc_clock_probe_read synchronises `stream` and returns stamps[8] = [first block | last block][t_entry, r_entry, t_exit,
 * r_exit] of the thread's latest armed launch; clock = (t_exit - t_entry) / (r_exit - r_entry) x 100 MHz.
 * svmc_clock_probe_arm(0) disarms and frees.  Per thread, not global: engines on other threads neither see nor disturb it. */
SVMC_API int svmc_clock_probe_arm(int enable);
SVMC_API int svmc_clock_probe_read(uint64_t *stamps, svmc_stream_t stream);

/* ---- state ----------------------------------------------------------------------------------------
 * x0 = zeros, sigma0 = v0 * ones, qvar0 = zeros of pricers/logsv_pricer.py:832-834 and
 * pricers/heston_pricer.py:303-305, written directly in HBM. */
SVMC_API int svmc_fill_state(double *x, double *vol, double *qvar, size_t n_path, double x0, double vol0,
                    double qvar0, svmc_stream_t stream);

/* ---- counter-based randoms (replaces np.random.normal inside the generators,
 * pricers/logsv_pricer.py:1025-1026, pricers/heston_pricer.py:369-370) ------------------------------
 * Philox4x32-7, key = seed, counter = (path_lo, path_hi, call index, stream | call_id << 8).  Stream 0: one call
 * serves the two normals (w0, w1) of each of the two time steps 2c, 2c + 1: every 32-bit word becomes one N(0,1)
 * variate by a piecewise-cubic inverse normal CDF (stream version 4, |z| <= 6.23); stream 1: one 52-bit uniform in
 * (0,1) per call.  `path` is the GLOBAL path id path_offset + p, `step` the chain-global step id step_offset + t, so
 * results do not depend on how paths are sharded over GPUs or how a chain is sliced.  `call_id` distinguishes the
 * calls of one seed: it occupies 24 bits of the counter (every entry point rejects call_id >= 2^24), so a host that
 * numbers its calls must re-seed before its 16 777 216th call -- the Python host's set_seed() counter wraps there and
 * says so (utils/funcs.py).  Exact definition: DESIGN.md "RNG"; CPU twin: oracle/svmc_oracle.c. */
SVMC_API int svmc_fill_normals(double *W0, double *W1, size_t ldw, size_t n_path, int nb_steps, uint64_t seed,
                      uint32_t call_id, uint64_t path_offset, uint32_t step_offset, svmc_stream_t stream);
SVMC_API int svmc_fill_uniforms(double *U, size_t ldw, size_t n_path, int nb_steps, uint64_t seed,
                       uint32_t call_id, uint64_t path_offset, uint32_t step_offset, svmc_stream_t stream);

/* ---- LogSV generator: simulate_logsv_x_vol_terminal, pricers/logsv_pricer.py:950-1047 ------------
 * In-place update of (x, sigma, qvar) over nb_steps log-Euler steps of size dt (Eq. 3.59).
 * _rng draws its increments on device (reference branch W0 is None, :1022-1026);
 * _w consumes supplied unscaled normals (reference branch :1028-1030). */
SVMC_API int svmc_logsv_terminal_rng(double *x, double *sigma, double *qvar, size_t n_path, int nb_steps, double dt,
                            double theta, double kappa1, double kappa2, double beta, double volvol,
                            double vol_backbone_eta, int is_spot_measure, uint64_t seed, uint32_t call_id,
                            uint64_t path_offset, uint32_t step_offset, svmc_stream_t stream);
/* Fused slice op for chain pricing: advance the state like svmc_logsv_terminal_rng AND, in the same kernel, copy
 * the terminal x (and qvar when qvar_snapshot != NULL) to the per-expiry snapshots and reduce
 * spot_sums[0..1] = { sum over non-NaN of forward*exp(x), count } (utils/mc_payoffs.py:61-62; the first step of
 * compute_mc_vars_payoff, see the payoff section below).  workspace >= svmc_slice_workspace_bytes(n_path). */
SVMC_API int svmc_logsv_slice_rng(double *x, double *sigma, double *qvar, size_t n_path, int nb_steps, double dt,
                                  double theta, double kappa1, double kappa2, double beta, double volvol,
                                  double vol_backbone_eta, int is_spot_measure, uint64_t seed, uint32_t call_id,
                                  uint64_t path_offset, uint32_t step_offset, double forward, double *x_snapshot,
                                  double *qvar_snapshot, double *spot_sums, void *workspace,
                                  size_t workspace_bytes, svmc_stream_t stream);
/* ALL expiries of a chain in one stepping launch (per 16 slices): slice i advances nb_steps_host[i] steps of
 * dts_host[i] with vol backbone etas_host[i] (NULL = 1), then writes x_snapshots[i][n_path] (and qvar_snapshots[i]
 * when non-NULL) and spot_sums[2i..2i+1] for forwards_host[i] -- the expiry loop of logsv_mc_chain_pricer
 * (pricers/logsv_pricer.py:840-865) without a launch boundary, and its tail, per expiry.  Same bits as calling
 * svmc_logsv_slice_rng slice by slice with step_offset advanced by the steps done. */
SVMC_API int svmc_logsv_chain_rng(double *x, double *sigma, double *qvar, size_t n_path, int n_slices,
                                  const int *nb_steps_host, const double *dts_host, const double *etas_host,
                                  const double *forwards_host, double theta, double kappa1, double kappa2, double beta,
                                  double volvol, int is_spot_measure, uint64_t seed, uint32_t call_id,
                                  uint64_t path_offset, uint32_t step_offset, double *x_snapshots, double *qvar_snapshots,
                                  double *spot_sums, void *workspace, size_t workspace_bytes, svmc_stream_t stream);
SVMC_API int svmc_logsv_terminal_w(double *x, double *sigma, double *qvar, size_t n_path, int nb_steps, double dt,
                          double theta, double kappa1, double kappa2, double beta, double volvol,
                          double vol_backbone_eta, int is_spot_measure, const double *W0, const double *W1,
                          size_t ldw, svmc_stream_t stream);
/* svmc_logsv_terminal_w + the slice epilogue of svmc_logsv_slice_rng (snapshots, spot sums) in the same launch:
 * one expiry of logsv_mc_chain_pricer_fixed_randoms (pricers/logsv_pricer.py:1140-1160) */
SVMC_API int svmc_logsv_slice_w(double *x, double *sigma, double *qvar, size_t n_path, int nb_steps, double dt,
                                double theta, double kappa1, double kappa2, double beta, double volvol,
                                double vol_backbone_eta, int is_spot_measure, const double *W0, const double *W1,
                                size_t ldw, double forward, double *x_snapshot, double *qvar_snapshot,
                                double *spot_sums, void *workspace, size_t workspace_bytes, svmc_stream_t stream);

/* ---- LogSV volatility paths on the full time grid: simulate_vol_paths, pricers/logsv_pricer.py:870-947 ---
 * sigma_t[(t+1)*ld + p] for t = 0..nb_steps-1, row 0 = v0 (written too): an explicit Euler scheme on L = ln sigma
 * driven by ONE Brownian motion with volatility vartheta = sqrt(beta^2 + volvol^2) (:937-945).  `brownians` are
 * the reference's SCALED increments sqrt(dt)*N(0,1), [nb_steps][ldb]; NULL draws them on device (stream 2 of the
 * counter-based generator: step t uses component t&1 of the Box-Muller pair of counter step t>>1).
 * HBM-write-bound: 8 B per path-step (+8 B read when brownians are supplied). */
SVMC_API int svmc_logsv_vol_paths(double *sigma_t, size_t ld, size_t n_path, int nb_steps, double dt, double v0,
                                  double theta, double kappa1, double kappa2, double beta, double volvol,
                                  int is_spot_measure, const double *brownians, size_t ldb, uint64_t seed,
                                  uint32_t call_id, uint64_t path_offset, svmc_stream_t stream);

/* ---- reductions of a resident [n_rows][ld] path array over the PATH axis: what the reference's callers of simulate_vol_paths
 * do with the array on the host (papers/logsv_model_with_quadratic_drift/moments_vol_qvar.py:48 -- np.mean / np.std along axis 1
 * of (sigma_t - theta)^k, k = 1..4 -- and :98 -- of sigma_t and of the expanding time average of sigma_t^2), done in HBM so that
 * 8 numbers per time step cross PCIe instead of nb_path.
 *   svmc_row_power_sums          sums[j * n_rows + t] = sum over the n_cols paths of (a[t * ld + p] - center)^(j + 1),
 *                                j = 0 .. 2 n_moments - 1 (n_moments <= 4): mean_k = S_k / n, var_k = S_2k / n - mean_k^2 of the
 *                                k-th power; deterministic (fixed tree); workspace >= n_rows x 4 x 2 n_moments doubles
 *   svmc_expanding_mean_squares  out[t * ldo + p] = mean of a[u * ld + p]^2 over u = 0 .. t (pandas expanding().mean() of the
 *                                squares, moments_vol_qvar.py:102); out may not alias a */
SVMC_API int svmc_row_power_sums(const double *a, size_t ld, size_t n_rows, size_t n_cols, double center, int n_moments,
                                 double *sums, void *workspace, size_t workspace_bytes, svmc_stream_t stream);
SVMC_API int svmc_expanding_mean_squares(const double *a, size_t ld, size_t n_rows, size_t n_cols, double *out, size_t ldo,
                                         svmc_stream_t stream);

/* ---- Black-76 implied vols of one slice, HOST arrays in and out (no device work): the price -> vol step between a
 * pricer and the calibration objective, OptionChain.compute_model_ivols_from_chain_data data/option_chain.py:327-346
 * (which delegates to the third-party vanilla_option_pricers.infer_bsm_ivols_from_model_chain_prices; parity with it
 * is unpinned).  optiontypes: SVMC_CALL / SVMC_PUT, and SVMC_INV_CALL / SVMC_INV_PUT as the vanilla inversion of price x
 * forward (the Black-76 value of (S - K)^+ / S is the vanilla value over the forward); another code gives
 * SVMC_ERR_UNKNOWN_PAYOFF.  A quote whose price is not strictly inside (price(vol_lo), price(vol_hi)) -- or is NaN -- gets NaN. */
SVMC_API int svmc_black_implied_vols(const double *prices, const double *strikes, const int8_t *optiontypes,
                                     size_t n_strikes, double forward, double ttm, double discfactor, double vol_lo,
                                     double vol_hi, double *ivols);

/* ---- rough LogSV (Markovian lift, n_factors <= 3): log_spot_full_combined_f64, pricers/rough_logsv/
 * split_simulation.py:335-356 (Strang splitting :249-278 over drift_ode_solve2 :86-128 and diffus_sde_solve_f64
 * :228-246; log-spot update :281-332).  In-place advance of (log_s[n], vol[n_factors][n], qvar[n]) over nb_steps of
 * size h.  nodes / weights / v0 are HOST arrays of n_factors entries (v0 = the factors' fixed reference level,
 * sigma0 / sum(weights) in the chain pricer :1187).  Z0 (factors) and Z1 (spot) are UNSCALED N(0,1), [nb_steps][ldw],
 * the reference's only interface for this model; pass both NULL to draw them on device (stream 3 of the
 * counter-based generator).  from_origin != 0 starts every path at (0, v0, 0) instead of reading the state arrays
 * (the chain pricer re-simulates each expiry from time 0, :1206-1216).  rho = beta / volvol, volvol = sqrt(beta^2 + orthog_vol^2) (:1190-1191). */
SVMC_API int svmc_rough_logsv_terminal(double *log_s, double *vol, double *qvar, size_t n_path, int nb_steps, double h,
                                       int n_factors, const double *nodes_host, const double *weights_host,
                                       const double *v0_host, double theta, double kappa1, double kappa2, double rho,
                                       double volvol, const double *Z0, const double *Z1, size_t ldw, uint64_t seed,
                                       uint32_t call_id, uint64_t path_offset, uint32_t step_offset, int from_origin,
                                       svmc_stream_t stream);
/* the same with the slice epilogue (x_snapshot <- log_s, qvar_snapshot <- qvar, spot sums): one expiry of
 * rough_logsv_mc_chain_pricer_fixed_randoms (pricers/logsv_pricer.py:1206-1230) in one stepping launch */
SVMC_API int svmc_rough_logsv_slice(double *log_s, double *vol, double *qvar, size_t n_path, int nb_steps, double h,
                                    int n_factors, const double *nodes_host, const double *weights_host,
                                    const double *v0_host, double theta, double kappa1, double kappa2, double rho,
                                    double volvol, const double *Z0, const double *Z1, size_t ldw, uint64_t seed,
                                    uint32_t call_id, uint64_t path_offset, uint32_t step_offset, int from_origin,
                                    double forward, double *x_snapshot, double *qvar_snapshot, double *spot_sums,
                                    void *workspace, size_t workspace_bytes, svmc_stream_t stream);
/* ALL expiries of rough_logsv_mc_chain_pricer_fixed_randoms (pricers/logsv_pricer.py:1206-1230) in ONE stepping launch: the
 * reference re-simulates every expiry from time 0 with its own step, so the expiries are independent simulations and run
 * side by side (grid.y = expiry).  Expiry i takes nb_steps_host[i] steps of hs_host[i] on rows 0 .. nb_steps_host[i] - 1 of
 * Z0 / Z1 (both NULL: drawn on device, stream 3, steps 0 ..); x_snapshots [n_expiries][n_path], qvar_snapshots the same or
 * NULL, spot_sums [2 n_expiries]; the state arrays receive the last expiry's terminal state.  Same bits per expiry as
 * svmc_rough_logsv_slice(from_origin = 1).  n_expiries <= 16. */
SVMC_API int svmc_rough_logsv_chain(double *log_s, double *vol, double *qvar, size_t n_path, int n_expiries,
                                    const int *nb_steps_host, const double *hs_host, const double *forwards_host,
                                    int n_factors, const double *nodes_host, const double *weights_host,
                                    const double *v0_host, double theta, double kappa1, double kappa2, double rho,
                                    double volvol, const double *Z0, const double *Z1, size_t ldw, uint64_t seed,
                                    uint32_t call_id, uint64_t path_offset, double *x_snapshots, double *qvar_snapshots,
                                    double *spot_sums, void *workspace, size_t workspace_bytes, svmc_stream_t stream);

/* ---- Heston generator: simulate_heston_x_vol_terminal, pricers/heston_pricer.py:334-381 ----------
 * `var` is the variance (the reference returns variance, not vol).  scheme = SVMC_HESTON_EULER_FLOOR
 * reproduces the reference (Euler, floor max(v, 1e-4)); SVMC_HESTON_QE is Andersen's QE-M. */
SVMC_API int svmc_heston_terminal_rng(double *x, double *var, double *qvar, size_t n_path, int nb_steps, double dt,
                             double theta, double kappa, double rho, double volvol, int scheme,
                             uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset,
                             svmc_stream_t stream);
SVMC_API int svmc_heston_slice_rng(double *x, double *var, double *qvar, size_t n_path, int nb_steps, double dt,
                                   double theta, double kappa, double rho, double volvol, int scheme,
                                   uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset,
                                   double forward, double *x_snapshot, double *qvar_snapshot, double *spot_sums,
                                   void *workspace, size_t workspace_bytes, svmc_stream_t stream);
/* all expiries of a Heston chain in one stepping launch (per 16 slices), as svmc_logsv_chain_rng: the expiry loop of
 * heston_mc_chain_pricer (pricers/heston_pricer.py:308-329); same bits as svmc_heston_slice_rng slice by slice */
SVMC_API int svmc_heston_chain_rng(double *x, double *var, double *qvar, size_t n_path, int n_slices,
                                   const int *nb_steps_host, const double *dts_host, const double *forwards_host,
                                   double theta, double kappa, double rho, double volvol, int scheme, uint64_t seed,
                                   uint32_t call_id, uint64_t path_offset, uint32_t step_offset, double *x_snapshots,
                                   double *qvar_snapshots, double *spot_sums, void *workspace, size_t workspace_bytes,
                                   svmc_stream_t stream);
/* The same four generators started from a UNIFORM state -- every path at (x0, vol0, qvar0), what a chain pricing begins with
 * (x0 = 0, sigma0 | v0, qvar0 = 0: pricers/logsv_pricer.py:823-826, pricers/heston_pricer.py:303-305) -- passed as three
 * constants: x / vol / qvar are then outputs only, and the svmc_fill_state launch (and its 24 bytes per path written, written
 * back and read again) disappears from the call. */
SVMC_API int svmc_logsv_slice_rng_from(double x0, double sigma0, double qvar0, double *x, double *sigma, double *qvar,
                                       size_t n_path, int nb_steps, double dt, double theta, double kappa1, double kappa2,
                                       double beta, double volvol, double vol_backbone_eta, int is_spot_measure,
                                       uint64_t seed, uint32_t call_id, uint64_t path_offset, uint32_t step_offset,
                                       double forward, double *x_snapshot, double *qvar_snapshot, double *spot_sums,
                                       void *workspace, size_t workspace_bytes, svmc_stream_t stream);
SVMC_API int svmc_logsv_chain_rng_from(double x0, double sigma0, double qvar0, double *x, double *sigma, double *qvar,
                                       size_t n_path, int n_slices, const int *nb_steps_host, const double *dts_host,
                                       const double *etas_host, const double *forwards_host, double theta, double kappa1,
                                       double kappa2, double beta, double volvol, int is_spot_measure, uint64_t seed,
                                       uint32_t call_id, uint64_t path_offset, uint32_t step_offset, double *x_snapshots,
                                       double *qvar_snapshots, double *spot_sums, void *workspace, size_t workspace_bytes,
                                       svmc_stream_t stream);
SVMC_API int svmc_heston_slice_rng_from(double x0, double var0, double qvar0, double *x, double *var, double *qvar,
                                        size_t n_path, int nb_steps, double dt, double theta, double kappa, double rho,
                                        double volvol, int scheme, uint64_t seed, uint32_t call_id, uint64_t path_offset,
                                        uint32_t step_offset, double forward, double *x_snapshot, double *qvar_snapshot,
                                        double *spot_sums, void *workspace, size_t workspace_bytes, svmc_stream_t stream);
SVMC_API int svmc_heston_chain_rng_from(double x0, double var0, double qvar0, double *x, double *var, double *qvar,
                                        size_t n_path, int n_slices, const int *nb_steps_host, const double *dts_host,
                                        const double *forwards_host, double theta, double kappa, double rho, double volvol,
                                        int scheme, uint64_t seed, uint32_t call_id, uint64_t path_offset,
                                        uint32_t step_offset, double *x_snapshots, double *qvar_snapshots, double *spot_sums,
                                        void *workspace, size_t workspace_bytes, svmc_stream_t stream);
SVMC_API int svmc_heston_terminal_w(double *x, double *var, double *qvar, size_t n_path, int nb_steps, double dt,
                           double theta, double kappa, double rho, double volvol, const double *W0,
                           const double *W1, size_t ldw, svmc_stream_t stream);
SVMC_API int svmc_heston_qe_terminal_w(double *x, double *var, double *qvar, size_t n_path, int nb_steps, double dt,
                              double theta, double kappa, double rho, double volvol, const double *Z0,
                              const double *Z1, const double *U, size_t ldw, svmc_stream_t stream);

/* ---- payoff reduction: compute_mc_vars_payoff, utils/mc_payoffs.py:10-88 -------------------------
 * Split at the two points where the reference needs a global quantity, so that a multi-GPU caller can
 * all-reduce the partial sums in between (sums are plain fp64 and add across ranks):
 *   1. svmc_spot_sums     spot_sums[0..1] = { sum over non-NaN of F*exp(x), count of non-NaN }  (:61-62)
 *   2. svmc_payoff_sums   per strike k, with d = payoff - shift_k over the non-NaN payoffs:
 *                         sums[3k..3k+2] = { sum d, sum d^2, count }; the recentring
 *                         c = spot_sums[0]/spot_sums[1] - F is read from device memory             (:63-86)
 *   3. svmc_payoff_finalize (host) price = DF*(shift + sum/cnt),
 *                         stderr = DF*sqrt(sum2/cnt - (sum/cnt)^2)/sqrt(N_total)                   (:85-88)
 * shifts_host (nullable = zeros) are any per-strike constants near the mean payoff (the host mirror uses
 * the intrinsic value at the forward); they only remove the cancellation in E[p^2] - E[p]^2.
 * `workspace` is a device scratch buffer of at least svmc_payoff_workspace_bytes() bytes. */
SVMC_API int svmc_payoff_workspace_bytes(size_t *bytes);
SVMC_API int svmc_slice_workspace_bytes(size_t n_path, size_t *bytes); /* covers the fused slice ops too */
SVMC_API int svmc_spot_sums(const double *x, size_t n_path, double forward, double *spot_sums, void *workspace,
                   size_t workspace_bytes, svmc_stream_t stream);
SVMC_API int svmc_payoff_sums(const double *x, const double *qvar, size_t n_path, double forward, double ttm,
                     const double *spot_sums, const double *strikes_host, const int8_t *types_host,
                     const double *shifts_host, size_t n_strikes, int variable_type, double *sums,
                     void *workspace, size_t workspace_bytes, svmc_stream_t stream);
/* svmc_payoff_sums for ALL expiries of a chain in one pair of launches (per 20 strike-chunks of 8): expiry i reads
 * the snapshots x_snapshots_host[i] (and qvar_snapshots_host[i] for Q_VAR; HOST arrays of DEVICE pointers), its
 * recentring sums spot_sums[2i..2i+1], forwards_host[i], ttms_host[i] and the strikes
 * [strike_offsets_host[i], strike_offsets_host[i+1]) of the concatenated strikes / types / shifts; sums gets the
 * 3 * strike_offsets_host[n_expiries] doubles in chain order.  Bit-identical to per-expiry svmc_payoff_sums. */
SVMC_API int svmc_payoff_sums_chain(const double *const *x_snapshots_host, const double *const *qvar_snapshots_host,
                                    size_t n_path, const double *forwards_host, const double *ttms_host,
                                    const double *spot_sums, int n_expiries, const double *strikes_host,
                                    const int8_t *types_host, const double *shifts_host,
                                    const size_t *strike_offsets_host, int variable_type, double *sums, void *workspace,
                                    size_t workspace_bytes, svmc_stream_t stream);
SVMC_API int svmc_payoff_finalize(const double *sums_host, const double *shifts_host, size_t n_strikes,
                         double discfactor, double n_path_total, double *prices_host, double *stderrs_host);
/* the same for all the strikes of a chain in one call: discfactors_host[k] is the discount factor of strike k's expiry
 * (the host mirror's chain driver finalises 8 x 21 strikes in one call instead of eight) */
SVMC_API int svmc_payoff_finalize_chain(const double *sums_host, const double *shifts_host, const double *discfactors_host,
                               size_t n_strikes, double n_path_total, double *prices_host, double *stderrs_host);

/* ---- fused single-GPU chain drivers: one call per option chain -----------------------------------------------
 * logsv_mc_chain_pricer (pricers/logsv_pricer.py:806-867) and heston_mc_chain_pricer (pricers/heston_pricer.py:
 * 285-331) as host C++ over the kernels above.  A session owns the resident state / snapshot / scratch buffers of
 * n_path paths on the current device and its own stream; chains of up to max_expiries maturities and
 * max_strikes_total strikes can be priced with it, any number of times.  Chain arrays are HOST arrays in the
 * reference's layout: ttms / forwards / discfactors [n_expiries], strikes and int8 payoff codes concatenated over the
 * expiries with strike_offsets[n_expiries + 1] delimiting each slice; prices / stderrs come back in the same
 * concatenated layout.  nb_steps_per_year is the reference's rule nb_steps_i = int((T_i - T_{i-1}) * spy) + 1
 * (utils/funcs.py:44).  vol_backbone_etas_host may be NULL (ones).  svmc_session_state copies the terminal state out
 * (x, sigma | variance, qvar; any pointer may be NULL).  Multi-GPU: give the session a communicator
 * (svmc_session_set_comm, below) and the same calls shard the job; the Python host mirror
 * (stochvolmodels_amd/mc_chain.py) places the same two all-reduces between the same kernels. */
SVMC_API int svmc_session_create(svmc_session_t *session, size_t n_path, int max_expiries, size_t max_strikes_total);
/* A session over state arrays and a stream the CALLER owns (device pointers x, vol, qvar of n_path doubles each; stream may be
 * NULL, the null stream): the session allocates its snapshots / sums / workspace only, every chain it prices leaves the terminal
 * state in the caller's arrays, and svmc_session_destroy frees neither them nor the stream.  path_offset = the global id of the
 * arrays' first path (a lone shard of a bigger job; 0 otherwise).  This is how the Python host prices a chain on ONE GPU: its
 * engine's resident state, one C-ABI call per chain instead of a dozen (logsv_mc_chain_pricer, pricers/logsv_pricer.py:806-867). */
SVMC_API int svmc_session_create_on(svmc_session_t *session, size_t n_path, int max_expiries, size_t max_strikes_total,
                                    double *x, double *vol, double *qvar, uint64_t path_offset, svmc_stream_t stream);
SVMC_API int svmc_session_destroy(svmc_session_t session);
/* Measurement: with timing enabled, svmc_logsv_chain_price / svmc_heston_chain_price bracket their stepping launch (and its
 * spot-sum reduce) with HIP events on the session's stream; svmc_session_last_stepping_ms returns the elapsed time of the last
 * such call (-1 if none was timed).  This is how bench.py times the dominant kernel inside its timed region without a profiler. */
SVMC_API int svmc_session_time_stepping(svmc_session_t session, int enable);
SVMC_API int svmc_session_last_stepping_ms(svmc_session_t session, float *ms);
SVMC_API int svmc_session_state(svmc_session_t session, double *x_host, double *vol_host, double *qvar_host);
SVMC_API int svmc_logsv_chain_price(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                                    const double *discfactors_host, const double *vol_backbone_etas_host,
                                    int n_expiries, const double *strikes_host, const int8_t *types_host,
                                    const size_t *strike_offsets_host, double v0, double theta, double kappa1,
                                    double kappa2, double beta, double volvol, int is_spot_measure,
                                    int nb_steps_per_year, int variable_type, uint64_t seed, uint32_t call_id,
                                    double *prices_host, double *stderrs_host);
/* the same chain on supplied randoms resident in HBM: logsv_mc_chain_pricer_fixed_randoms, pricers/logsv_pricer.py:
 * 1100-1162.  W0s[i] / W1s[i] are DEVICE pointers to the UNSCALED N(0,1) of expiry i, [nb_steps_host[i]][ldw]; the
 * arrays of pointers, step counts and dts are host arrays of n_expiries entries.  This is the inner loop of an MC
 * calibration (:244-265): one call per optimizer iterate, nothing but the chain and the prices crosses PCIe. */
SVMC_API int svmc_logsv_chain_price_fixed(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                                          const double *discfactors_host, const double *vol_backbone_etas_host,
                                          int n_expiries, const double *strikes_host, const int8_t *types_host,
                                          const size_t *strike_offsets_host, double v0, double theta, double kappa1,
                                          double kappa2, double beta, double volvol, int is_spot_measure,
                                          int variable_type, const double *const *W0s, const double *const *W1s,
                                          const int *nb_steps_host, const double *dts_host, size_t ldw,
                                          double *prices_host, double *stderrs_host);
/* the same call with the price -> implied-vol step of a calibration objective (data/option_chain.py:327-346) done where
 * the prices are: ivols_host [sum K_i] receives the Black-76 implied vols of the quotes of a LOG_RETURN chain ('IC' / 'IP':
 * of price x forward) on the bracket [1e-6, 10] (NaN outside it and for Q_VAR chains) -- on the graph route one more
 * kernel node and one more copy-back in the captured graph (svmc_black.h: the solver of svmc_black_implied_vols), on
 * the others the host routine after the prices.  ivols_host may be NULL (= svmc_logsv_chain_price_fixed). */
SVMC_API int svmc_logsv_chain_price_fixed_iv(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                                             const double *discfactors_host, const double *vol_backbone_etas_host,
                                             int n_expiries, const double *strikes_host, const int8_t *types_host,
                                             const size_t *strike_offsets_host, double v0, double theta, double kappa1,
                                             double kappa2, double beta, double volvol, int is_spot_measure,
                                             int variable_type, const double *const *W0s, const double *const *W1s,
                                             const int *nb_steps_host, const double *dts_host, size_t ldw,
                                             double *prices_host, double *stderrs_host, double *ivols_host);
/* the same chain for SEVERAL PARAMETER SETS on the same resident randoms -- the base point of an optimizer iterate and its
 * finite-difference neighbours -- with the randoms read once: params_host [n_sets][6 + n_expiries] = {v0, theta, kappa1,
 * kappa2, beta, volvol, vol_backbone_eta of each expiry}, outputs [n_sets][sum K_i] (ivols_host may be NULL).  With 2..8
 * sets and a session created for n_sets chains (svmc_session_create(.., max_expiries >= n_sets * n_expiries,
 * max_strikes_total >= n_sets * sum K_i)) the replayed graph steps all sets in ONE launch, every lane carrying the n_sets
 * states of its path (the single-set launch is bound by reading the randoms); otherwise the sets are priced one after the
 * other.  Either way each set gets the bits svmc_logsv_chain_price_fixed_iv gives it. */
SVMC_API int svmc_logsv_chain_price_fixed_sets(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                                               const double *discfactors_host, int n_expiries, const double *strikes_host,
                                               const int8_t *types_host, const size_t *strike_offsets_host, int n_sets,
                                               const double *params_host, int is_spot_measure, int variable_type,
                                               const double *const *W0s, const double *const *W1s,
                                               const int *nb_steps_host, const double *dts_host, size_t ldw,
                                               double *prices_host, double *stderrs_host, double *ivols_host);
/* The calibration chain with NO resident randoms: the fixed randoms of an MC calibration (pricers/logsv_pricer.py:244-265,
 * :520-527 draw them once and re-price on them at every optimizer iterate) are the counter-based stream of (seed, call_id),
 * regenerated in registers by every evaluation -- the same draws every time, nothing in HBM (10^5 paths x 364 steps of
 * resident W0 / W1 are 582 MB, and streaming them back is the floor of svmc_logsv_chain_price_fixed*).  n_sets parameter
 * sets (params_host [n_sets][6 + n_expiries] as for svmc_logsv_chain_price_fixed_sets) are stepped by ONE launch per 8 sets,
 * every lane drawing its path's normals once per step and advancing all the sets on them; outputs [n_sets][sum K_i].
 * Set q's prices are those of svmc_logsv_chain_price(seed, call_id) with set q's parameters on the grid (nb_steps_host,
 * dts_host), BIT FOR BIT (the same step arithmetic; tests/test_gpu_parity.py).  The session must be created for n_sets
 * chains (svmc_session_create(.., max_expiries >= min(n_sets, 8) x n_expiries, max_strikes_total >= min(n_sets, 8) x
 * sum K_i)).  The launches are captured into one hipGraph per set count and replayed; with a communicator attached
 * they are issued directly, the two all-reduces between them (global path ids: the job's result does not depend on the
 * sharding).  ivols_host may be NULL. */
SVMC_API int svmc_logsv_chain_price_frozen_sets(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                                                const double *discfactors_host, int n_expiries, const double *strikes_host,
                                                const int8_t *types_host, const size_t *strike_offsets_host, int n_sets,
                                                const double *params_host, int is_spot_measure, int variable_type,
                                                const int *nb_steps_host, const double *dts_host, uint64_t seed,
                                                uint32_t call_id, double *prices_host, double *stderrs_host,
                                                double *ivols_host);
/* svmc_logsv_chain_price_fixed captures its launches (ONE stepping launch for all expiries that also initialises the
 * state and writes the spot sums' partials, their reduce, the payoff sums, D2H; a launch per expiry beyond 16 expiries)
 * into a hipGraph the first time it sees a (chain, randoms) combination and replays it afterwards -- the model
 * constants travel in a small device block the graph's first node refreshes.  On by default; results are identical
 * either way.  svmc_session_graph_launches counts the replays (diagnostics). */
SVMC_API int svmc_session_use_graphs(svmc_session_t session, int enable);
SVMC_API int svmc_session_graph_launches(svmc_session_t session, size_t *count);
SVMC_API int svmc_heston_chain_price(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                                     const double *discfactors_host, int n_expiries, const double *strikes_host,
                                     const int8_t *types_host, const size_t *strike_offsets_host, double v0,
                                     double theta, double kappa, double rho, double volvol, int scheme,
                                     int nb_steps_per_year, int variable_type, uint64_t seed, uint32_t call_id,
                                     double *prices_host, double *stderrs_host);

/* ---- multi-GPU below the host language: RCCL over xGMI ----------------------------------------------------------
 * One process (or thread) per GPU.  Paths are independent, so rank r holds the global path ids
 * [path_offset, path_offset + n_path) of a job of n_path_total paths in its own session; the counter-based randoms
 * are indexed by the GLOBAL path id, so the job's result does not depend on how it is sharded.  The only cross-rank
 * couplings are the two reductions of compute_mc_vars_payoff -- the forward recentring's [sum F exp(x), count] per
 * expiry (utils/mc_payoffs.py:61-63) and [sum d, sum d^2, count] per strike (:85-86): with a communicator attached,
 * svmc_logsv_chain_price / svmc_heston_chain_price / svmc_logsv_chain_price_fixed issue them as two in-place fp64
 * ncclAllReduce(SUM) calls on the session's stream (2 M and 3 sum K_i doubles: a few KB, latency-bound on xGMI),
 * stream-ordered between the kernels, and every rank returns the job's prices and standard errors.
 * This is SURVEY.md 8b's design target `svmc_chain_price(... optional RCCL comm handle)`.
 *
 * RCCL is resolved at run time: the copy already mapped into the process if there is one (PyTorch-ROCm carries its
 * own), else librccl.so.1.  A communicator must be driven by the library that created it: create it with
 * svmc_rccl_comm_create, or make sure a foreign ncclComm_t comes from that same copy (svmc_rccl_origin says which).
 *   rank 0:     svmc_rccl_unique_id(id, sizeof id)  -> ship the SVMC_RCCL_UNIQUE_ID_BYTES to the other ranks
 *   every rank: svmc_set_device(r); svmc_rccl_comm_create(&comm, id, sizeof id, world, r);
 *               svmc_session_create(&s, n_local, ...); svmc_session_set_comm(s, comm, r, world, n_total, offset);
 *               svmc_logsv_chain_price(s, ...)      -> identical prices on every rank
 * svmc_session_set_comm(session, NULL, ...) detaches.  Graph replay of the fixed-randoms driver is bypassed while a
 * communicator is attached. */
#define SVMC_RCCL_UNIQUE_ID_BYTES 128
SVMC_API int svmc_rccl_available(void);              /* 1 when an RCCL could be resolved, else 0 (svmc_rccl_origin: why) */
SVMC_API const char *svmc_rccl_origin(void);         /* where RCCL was resolved from, or the reason it was not */
SVMC_API int svmc_rccl_unique_id(void *id_out, size_t bytes);
SVMC_API int svmc_rccl_comm_create(svmc_comm_t *comm, const void *id_bytes, size_t bytes, int world, int rank);
SVMC_API int svmc_rccl_comm_destroy(svmc_comm_t comm);
/* ncclCommCount / ncclCommUserRank of a communicator: the number of ranks RCCL itself sees in it (bench.py reports it as
 * rccl_ranks_seen) and this rank's index; rank_out may be NULL */
SVMC_API int svmc_rccl_comm_count(svmc_comm_t comm, int *world_out, int *rank_out);
SVMC_API int svmc_rccl_all_reduce_sum(svmc_comm_t comm, double *buf, size_t n, svmc_stream_t stream);
SVMC_API int svmc_session_set_comm(svmc_session_t session, svmc_comm_t comm, int rank, int world,
                                   uint64_t n_path_total, uint64_t path_offset);
/* The same sharding with the two sum all-reduces handed to the CALLER: fn(user, device_buf, n, stream) must leave in
 * device_buf (n doubles, written by kernels queued on `stream`) the element-wise sum over all ranks, visible to work queued
 * on `stream` afterwards, and return SVMC_OK -- MPI, a host-staged exchange, or several sessions of ONE process on one GPU
 * summing through host memory (tests/test_gpu_parity.py runs 2- and 3-shard jobs that way and requires the unsharded
 * session's prices to reduction-order rounding, 1e-12: the per-wave partial rows depend on where a shard starts).
 * fn == NULL detaches, as svmc_session_set_comm(session, NULL, ...). */
typedef int (*svmc_all_reduce_fn)(void *user, double *device_buf, size_t n, svmc_stream_t stream);
SVMC_API int svmc_session_set_reducer(svmc_session_t session, svmc_all_reduce_fn fn, void *user, int rank, int world,
                                      uint64_t n_path_total, uint64_t path_offset);

/* ---- single-process multi-device sessions ----------------------------------------------------------------------
 * The same sharding WITHOUT a process per GPU: logsv_mc_chain_pricer / heston_mc_chain_pricer (pricers/logsv_pricer.py:
 * 806-867, pricers/heston_pricer.py:285-331 -- one single-process NumPy loop in the reference) for a caller that owns
 * several devices from one process.  A multi-session is n_shards sessions of one job of n_path_total paths -- shard r
 * holds the global path ids [N r / R, N (r + 1) / R) on device devices_host[r] (NULL: r mod the visible devices), each
 * driven by its own host thread of the library -- and one svmc_multi_*_chain_price call prices the chain on all of them
 * concurrently and returns the job's prices (every shard finalises the same all-reduced sums; svmc_multi_info's
 * shards_agree says whether they came out bit-identical, as they must).  Arguments are those of svmc_logsv_chain_price /
 * svmc_heston_chain_price with the multi-session in place of the session.
 * reduce_mode picks the transport of the two sum all-reduces (utils/mc_payoffs.py:61-63, :85-86):
 *   SVMC_MULTI_REDUCE_RCCL  ncclCommInitAll, one communicator per device, the collectives issued on each shard's stream
 *                           (error if RCCL does not resolve or two shards share a device);
 *   SVMC_MULTI_REDUCE_HOST  a sum through page-locked host memory in rank order, one meeting point of the shard threads
 *                           per all-reduce -- no dependency beyond the HIP runtime, works with several shards on ONE
 *                           device (tests/test_gpu_multi.py prices C4's 8 x 2^21 shards that way on a one-GPU box);
 *   SVMC_MULTI_REDUCE_AUTO  RCCL where it can be had, else HOST; svmc_multi_info reports which one is in use.
 * Results equal one session of n_path_total paths up to the order of the final additions (the randoms are indexed by the
 * global path id) and do not depend on the transport.  A failing shard releases the others from the host meeting point
 * and the call returns its status and message; inside an RCCL collective a lost peer cannot be recovered from (RCCL's own
 * semantics).  Calls on one multi-session must not overlap; distinct multi-sessions are independent. */
typedef void *svmc_multi_t;
#define SVMC_MULTI_REDUCE_AUTO 0
#define SVMC_MULTI_REDUCE_HOST 1
#define SVMC_MULTI_REDUCE_RCCL 2
SVMC_API int svmc_multi_create(svmc_multi_t *multi, int n_shards, const int *devices_host, uint64_t n_path_total,
                               int max_expiries, size_t max_strikes_total, int reduce_mode);
SVMC_API int svmc_multi_destroy(svmc_multi_t multi);
/* any output may be NULL: shard count, transport in use (SVMC_MULTI_REDUCE_HOST / _RCCL), the ranks RCCL's warm-up
 * all-reduce counted (0 in host mode), and whether all shards returned identical bits in the last pricing call */
SVMC_API int svmc_multi_info(svmc_multi_t multi, int *n_shards, int *reduce_mode, int *rccl_ranks_seen, int *shards_agree);
/* shard r: its device, path range and the host wall time of its part of the last call (a slow device shows here) */
SVMC_API int svmc_multi_shard_info(svmc_multi_t multi, int shard, int *device, uint64_t *path_offset, uint64_t *n_path,
                                   double *last_call_ms);
SVMC_API int svmc_multi_logsv_chain_price(svmc_multi_t multi, const double *ttms_host, const double *forwards_host,
                                          const double *discfactors_host, const double *vol_backbone_etas_host,
                                          int n_expiries, const double *strikes_host, const int8_t *types_host,
                                          const size_t *strike_offsets_host, double v0, double theta, double kappa1,
                                          double kappa2, double beta, double volvol, int is_spot_measure,
                                          int nb_steps_per_year, int variable_type, uint64_t seed, uint32_t call_id,
                                          double *prices_host, double *stderrs_host);
SVMC_API int svmc_multi_heston_chain_price(svmc_multi_t multi, const double *ttms_host, const double *forwards_host,
                                           const double *discfactors_host, int n_expiries, const double *strikes_host,
                                           const int8_t *types_host, const size_t *strike_offsets_host, double v0,
                                           double theta, double kappa, double rho, double volvol, int scheme,
                                           int nb_steps_per_year, int variable_type, uint64_t seed, uint32_t call_id,
                                           double *prices_host, double *stderrs_host);
/* the terminal state of the whole job, shard by shard into [n_path_total] host arrays (any pointer may be NULL) */
SVMC_API int svmc_multi_state(svmc_multi_t multi, double *x_host, double *vol_host, double *qvar_host);

/* ---- analytic side: affine-expansion MGF + Fourier inversion ------------------
 * Complex arrays are interleaved (re, im) doubles, i.e. numpy.complex128 / C99 double complex, on the device.
 *   svmc_logsv_mgf_grid    compute_logsv_a_mgf_grid, pricers/logsv/affine_expansion.py:570-685 (numerical path):
 *                          per grid point integrate A' = A^T M A + L A + H (:67-205) over ttm from a[] (in: A at the
 *                          previous expiry, out: A at this one; [n_grid][5] for expansion_order 2, [n_grid][3] for 1)
 *                          and return log_mgf = sum_k A_k (sigma0 - theta)^k.  The reference calls SciPy RK45 at its
 *                          default rtol 1e-3 / atol 1e-6; here the Dormand-Prince 8(5,3) pair (DOP853) takes rtol/atol.
 *   svmc_heston_mgf_grid   compute_heston_mgf_grid, pricers/heston_pricer.py:183-214 (closed form; a, b carried).
 *   svmc_mgf_vanilla_slice the strike sums of vanilla_slice_pricer_with_mgf_grid, utils/mgf_pricer.py:174-221, for
 *                          grids with |Re phi| = 1/2: capped[k] = nansum_j Re[ w_j/(pi (p_j^2+1/4))
 *                          exp(-ln(F/K_k) phi_j + log_mgf_j) ] with the legacy Simpson weights (:158-171); the
 *                          payoff algebra on top (:208-219) is host code. */
SVMC_API int svmc_logsv_mgf_grid(const double *phi, const double *psi, size_t n_grid, double ttm, double sigma0,
                                 double theta, double kappa1, double kappa2, double beta, double volvol,
                                 int is_spot_measure, int expansion_order, double vol_backbone_eta, double *a,
                                 double *log_mgf, double rtol, double atol, svmc_stream_t stream);
/* Batched over parameter sets: n_sets independent LogSV models advance their transform grids over the same interval in
 * ONE launch (the five sets of config C5; the bumped parameter vectors of a finite-difference gradient).  Set s has its
 * own grid phi / psi [s][n_grid] (the grid scale follows sigma0), coefficients a [s][n_grid][n_coef] and output log_mgf
 * [s][n_grid]; params_host [s][SVMC_LOGSV_SET_DOUBLES] = {sigma0, theta, kappa1, kappa2, beta, volvol, vol_backbone_eta,
 * reserved}.  svmc_mgf_vanilla_slice_batch: the strike sums of every set for one (forward, strikes) slice, capped
 * [s][n_strikes].  Results are bit-identical to n_sets single calls. */
#define SVMC_LOGSV_SET_DOUBLES 8
SVMC_API int svmc_logsv_mgf_grid_batch(const double *phi, const double *psi, size_t n_grid, int n_sets, double ttm,
                                       const double *params_host, int is_spot_measure, int expansion_order, double *a,
                                       double *log_mgf, double rtol, double atol, svmc_stream_t stream);
SVMC_API int svmc_mgf_vanilla_slice_batch(const double *phi, const double *log_mgf, size_t n_grid, int n_sets,
                                          double forward, const double *strikes_host, size_t n_strikes, double *capped,
                                          svmc_stream_t stream);
SVMC_API int svmc_heston_mgf_grid(const double *phi, const double *psi, size_t n_grid, double ttm, double v0,
                                  double theta, double kappa, double volvol, double rho, double *a, double *b,
                                  int have_t0, double *log_mgf, svmc_stream_t stream);
/* the strike sums of slice_qvar_pricer_with_a_grid, utils/mgf_pricer.py:322-356 (options on the annualised quadratic
 * variance): capped[k] = nansum_j Re[ w_j/(pi psi_j^2) exp(K_k ttm psi_j + log_mgf_j) ], legacy Simpson weights */
SVMC_API int svmc_mgf_qvar_slice(const double *psi, const double *log_mgf, size_t n_grid, double ttm,
                                 const double *strikes_host, size_t n_strikes, double *capped,
                                 svmc_stream_t stream);
SVMC_API int svmc_mgf_vanilla_slice(const double *phi, const double *log_mgf, size_t n_grid, double forward,
                                    const double *strikes_host, size_t n_strikes, double *capped,
                                    svmc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SVMC_H */
