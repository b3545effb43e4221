// Host build of stochvolmodels_amd/csrc/svmc_math.h for tests/test_math_accuracy.py (g++ only).
#include <stddef.h>
#include "svmc_log_table.h"
#include "svmc_math.h"
static const svmc::LogTabEntry LOG_TAB[512] = {SVMC_LOG_TABLE_INIT};
static const double EXP_TAB[256] = {SVMC_EXP_TABLE_INIT};
#include "svmc_icdf_table.h"
static const svmc::IcdfPiece ICDF_TAB[2 * SVMC_ICDF_SEGMENTS] = {SVMC_ICDF_PIECE0_INIT, SVMC_ICDF_PIECE1_INIT};
extern "C" {
void probe_exp(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = svmc::exp_fast(x[i]); }
void probe_exp_full(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = svmc::exp_full(x[i]); }
void probe_exp_tab(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = svmc::exp_tab(x[i], EXP_TAB); }
void probe_exp2u_tab(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = svmc::exp2u_tab(x[i], EXP_TAB); }
void probe_neg_log(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = svmc::neg_log(x[i]); }
void probe_neg_log_tab(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = svmc::neg_log_tab(x[i], LOG_TAB); }
void probe_sqrt(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = svmc::sqrt_pos(x[i]); }
void probe_sqrt_1g(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = svmc::sqrt_pos_1g(x[i]); }
void probe_rcp(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = svmc::rcp_fast(x[i]); }
// one normal from one raw 32-bit word (random stream version 3)
void probe_normal_icdf32(const uint32_t *w, double *z, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        z[i] = svmc::normal_icdf32<SVMC_ICDF_M, SVMC_ICDF_SEGMENTS, SVMC_ICDF_DEG, SVMC_ICDF_EDGE != 0, SVMC_ICDF_RAW != 0>(w[i], ICDF_TAB);
}
void probe_log_state(const double *x, double *y, size_t n) { for (size_t i = 0; i < n; ++i) y[i] = svmc::log_state(x[i]); }
}
