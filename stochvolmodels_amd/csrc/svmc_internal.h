// svmc_internal.h -- error plumbing shared by the translation units of libsvmc.so
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#include "svmc.h"

namespace svmc {

std::string &last_error_ref();
int fail(int code, const std::string &msg);

#define SVMC_HIP_TRY(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return ::svmc::fail(SVMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));   \
    } while (0)

#define SVMC_REQUIRE(cond, msg)                                                                    \
    do {                                                                                           \
        if (!(cond)) return ::svmc::fail(SVMC_ERR_INVALID_ARGUMENT, std::string(msg));             \
    } while (0)

inline hipStream_t as_stream(svmc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// rows of the generators' per-WAVE spot partials ([column][wave], svmc_kernels.hip SliceOut): one per 64 paths.  A fused launch
// writes 2 columns per (expiry, parameter set), so its workspace holds wave_rows(n) x 2 x columns doubles
inline unsigned wave_rows(size_t n) { return static_cast<unsigned>((n + 63) / 64); }

// svmc_kernels.hip: launches whose model constants live in device memory (graph replay, svmc_chain.hip)
constexpr int LOGSV_CONSTS_DOUBLES = 13;
int fill_state_indirect(double *x, double *vol, double *qvar, size_t n_path, const double *vol0_dev, hipStream_t stream);
int logsv_slice_w_indirect(double *x, double *sigma, double *qvar, size_t n_path, int nb_steps, const double *consts_dev,
                           const double *W0, const double *W1, size_t ldw, double forward, double *x_snapshot,
                           double *qvar_snapshot, double *spot_sums, void *workspace, size_t workspace_bytes,
                           hipStream_t stream);
constexpr int MAX_FUSED_SLICES = 16;    // = MAX_CHAIN_SLICES of svmc_kernels.hip
int logsv_chain_w_indirect(double *x, double *sigma, double *qvar, size_t n_path, int n_slices, const int *nb_steps_host,
                           const double *consts_dev, const double *vol0_dev, const double *const *W0s, const double *const *W1s,
                           size_t ldw, const double *forwards_host, double *x_snapshots, double *qvar_snapshots,
                           double *spot_sums, void *workspace, size_t workspace_bytes, hipStream_t stream);
constexpr int MAX_FUSED_SETS = 8;       // = MAX_CHAIN_SETS of svmc_kernels.hip
int logsv_chain_w_sets(size_t n_path, int n_sets, int n_slices, const int *nb_steps_host, const double *consts_dev,
                       const double *vol0_dev, const double *const *W0s, const double *const *W1s, size_t ldw,
                       const double *forwards_host, double *x_snapshots, double *qvar_snapshots, double *spot_sums,
                       void *workspace, size_t workspace_bytes, hipStream_t stream);
// the same chain for n_sets parameter sets on randoms REGENERATED in registers from (seed, call_id): no resident randoms
constexpr int LOGSV_FAST_CONSTS_DOUBLES = 9;
int logsv_chain_rng_sets(size_t n_path, int n_sets, int n_slices, const int *nb_steps_host, const double *consts_dev,
                         const double *vol0_dev, const double *forwards_host, uint64_t seed, uint32_t call_id,
                         uint64_t path_offset, double *x_snapshots, double *qvar_snapshots, double *spot_sums, void *workspace,
                         size_t workspace_bytes, hipStream_t stream, bool allow_probe);
void logsv_fast_to_doubles(double dt, double theta, double kappa1, double kappa2, double beta, double volvol, double eta,
                           int is_spot_measure, double *out);
// payoff sums of n_sets parameter sets' snapshots in ONE launch and ONE column reduce (bit-equal to n_sets calls of
// svmc_payoff_sums_chain); payoff_sets_fit says whether the chain's strike groups fit one launch and the workspace
bool payoff_sets_fit(size_t n_path, int n_expiries, const size_t *offsets, const int8_t *types, int n_sets, size_t workspace_bytes);
int payoff_sums_chain_sets(const double *const *x_snapshots_host, const double *const *qvar_snapshots_host, size_t n_path,
                           const double *forwards_host, const double *ttms_host, const double *spot_sums, int n_expiries,
                           const double *strikes_host, const int8_t *types_host, const double *shifts_host,
                           const size_t *strike_offsets_host, int variable_type, double *sums, void *workspace,
                           size_t workspace_bytes, hipStream_t stream, int n_sets, size_t x_set_stride, size_t q_set_stride,
                           size_t spot_set_stride);
// ---- the one-device tail of an on-device-RNG chain (round 6): stepping WITHOUT the reduce of its per-wave spot partials, then the
// payoff kernel (up to 2048 partial rows: every block sums its expiry's two columns itself) and chain_finish_kernel (a wave per
// quote: column sums in reduce_columns_kernel's order, stored where the host reads them)
int logsv_step_partials(double sigma0, double *x, double *sigma, double *qvar, size_t n_path, int n_slices, const int *nb_steps_host,
                        const double *dts_host, const double *etas_host, const double *forwards_host, double theta, double kappa1,
                        double kappa2, double beta, double volvol, int is_spot_measure, uint64_t seed, uint32_t call_id,
                        uint64_t path_offset, double *x_snapshots, double *qvar_snapshots, void *workspace, size_t workspace_bytes,
                        hipStream_t stream);
int heston_step_partials(double var0, double *x, double *var, double *qvar, size_t n_path, int n_slices, const int *nb_steps_host,
                         const double *dts_host, const double *forwards_host, double theta, double kappa, double rho, double volvol,
                         int scheme, uint64_t seed, uint32_t call_id, uint64_t path_offset, double *x_snapshots,
                         double *qvar_snapshots, void *workspace, size_t workspace_bytes, hipStream_t stream);
bool spot_sums_in_payoff_kernel(size_t n_path);
int reduce_spot_partials(const void *workspace, size_t n_path, int n_cols, double *spot_sums, hipStream_t stream);
int chain_payoff_and_finish(const double *const *x_snapshots_host, const double *const *qvar_snapshots_host, size_t n_path,
                            const double *forwards_host, const double *ttms_host, double *spot_sums, const double *spot_partials,
                            int n_expiries, const double *strikes_host, const int8_t *types_host, const double *shifts_host,
                            const size_t *strike_offsets_host, int variable_type, void *workspace, size_t workspace_bytes,
                            hipStream_t stream, double *sums_out);
constexpr int IV_QUOTE_DOUBLES_HOST = 6;   // = IV_QUOTE_DOUBLES of svmc_kernels.hip: {strike, code, shift, forward, ttm, df}
// ivols_out and sums_copy (nullable: the kernel also stores the 3 x n_quotes sums it read) may be device-visible pinned host
// memory: the results of a graph then reach the host without copy nodes
int chain_implied_vols(const double *sums_dev, const double *quotes_dev, size_t n_quotes, double n_path_total, double vol_lo,
                       double vol_hi, double *ivols_out, double *sums_copy, hipStream_t stream);
// svmc_comm.hip: ncclCommInitAll -- comms_out[n] communicators for the devices[n] of this process (svmc_multi.hip)
int rccl_comm_init_all(int n, const int *devices, void **comms_out);
void logsv_consts_to_doubles(double dt, double theta, double kappa1, double kappa2, double beta, double volvol, double eta,
                             int is_spot_measure, double *out);

}  // namespace svmc
