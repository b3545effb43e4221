// svmc_chain.hip -- fused single-GPU chain drivers of the C ABI: one call prices a whole option chain.
//
// Restates, in host C++ over the kernels of svmc_kernels.hip, the expiry loop of logsv_mc_chain_pricer
// (pricers/logsv_pricer.py:806-867) and heston_mc_chain_pricer (pricers/heston_pricer.py:285-331): state
// x0 = 0, vol0 = v0, qvar0 = 0; per slice nb_steps_i = int((T_i - T_{i-1}) * spy) + 1 (utils/funcs.py:44) and
// dt_i = (T_i - T_{i-1}) / nb_steps_i; the state is carried slice to slice in HBM; payoffs per slice by
// compute_mc_vars_payoff (utils/mc_payoffs.py:10-88).  All launches of a chain are queued back to back on the
// session's stream and the host synchronises once, for the D2H of the 3*sum(K_i) sums.
//
// This is the path a C/C++ host takes (examples/price_chain.c).  The Python host drives the same kernels through
// mc_chain.py because it also has to place the two all-reduces of the multi-GPU case between the phases.
#include <cmath>
#include <vector>

#include "svmc_internal.h"

namespace svmc {

struct Session {
    size_t n_path = 0;
    int max_expiries = 0;
    size_t max_strikes = 0;
    double *x = nullptr, *vol = nullptr, *qvar = nullptr, *snap = nullptr, *spot = nullptr, *sums = nullptr;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    hipStream_t stream = nullptr;
};

static void session_release(Session *s)
{
    if (s == nullptr) return;
    for (void *p : {static_cast<void *>(s->x), static_cast<void *>(s->vol), static_cast<void *>(s->qvar),
                    static_cast<void *>(s->snap), static_cast<void *>(s->spot), static_cast<void *>(s->sums), s->ws})
        if (p != nullptr) (void)hipFree(p);
    if (s->stream != nullptr) (void)hipStreamDestroy(s->stream);
    delete s;
}

// utils/funcs.py:44-47
static void time_grid(double ttm, int spy, int &nb_steps, double &dt)
{
    nb_steps = static_cast<int>(ttm * static_cast<double>(spy)) + 1;
    dt = (nb_steps == 1) ? ttm : ttm / static_cast<double>(nb_steps);
}

// intrinsic value at the forward for LOG_RETURN, zero otherwise (the shift of svmc_payoff_sums)
static double payoff_shift(double strike, int type, double forward, int variable_type)
{
    if (variable_type != SVMC_LOG_RETURN) return 0.0;
    const bool call = (type == SVMC_CALL || type == SVMC_INV_CALL);
    const double intrinsic = call ? std::fmax(forward - strike, 0.0) : std::fmax(strike - forward, 0.0);
    return (type >= SVMC_INV_CALL) ? intrinsic / forward : intrinsic;
}

struct ChainView {
    int m;
    const double *ttms, *forwards, *discfactors, *strikes;
    const int8_t *types;
    const size_t *offsets;   // [m + 1] into strikes/types
};

static int check_chain(const char *fn, const Session *s, const ChainView &c, int variable_type, const double *prices,
                       const double *stderrs)
{
    if (s == nullptr) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": null session");
    if (!c.ttms || !c.forwards || !c.discfactors || !c.strikes || !c.types || !c.offsets || !prices || !stderrs)
        return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": null pointer");
    if (c.m < 1 || c.m > s->max_expiries) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": expiries exceed the session");
    if (c.offsets[c.m] > s->max_strikes) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": strikes exceed the session");
    if (variable_type == SVMC_SIGMA) return fail(SVMC_ERR_UNSUPPORTED_VARIABLE, std::string(fn) + ": VariableType.SIGMA");
    if (variable_type != SVMC_LOG_RETURN && variable_type != SVMC_Q_VAR)
        return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": unknown variable type");
    for (size_t k = 0; k < c.offsets[c.m]; ++k)
        if (c.types[k] < SVMC_CALL || c.types[k] > SVMC_INV_PUT) return fail(SVMC_ERR_UNKNOWN_PAYOFF, "unknown option payoff code");
    double prev = 0.0;
    for (int i = 0; i < c.m; ++i) {
        if (!(c.ttms[i] > prev)) return fail(SVMC_ERR_INVALID_ARGUMENT, std::string(fn) + ": ttms must be positive and increasing");
        prev = c.ttms[i];
    }
    return SVMC_OK;
}

// phases 3-4 of mc_chain.py: per-strike sums of every slice, one D2H, host finalisation
static int reduce_and_finalize(Session *s, const ChainView &c, int variable_type, double *prices, double *stderrs)
{
    const size_t n = s->n_path;
    std::vector<double> shifts(c.offsets[c.m]);
    for (int i = 0; i < c.m; ++i) {
        const size_t k0 = c.offsets[i], k = c.offsets[i + 1] - k0;
        for (size_t j = 0; j < k; ++j) shifts[k0 + j] = payoff_shift(c.strikes[k0 + j], c.types[k0 + j], c.forwards[i], variable_type);
        const double *qsnap = (variable_type == SVMC_Q_VAR) ? s->snap + static_cast<size_t>(c.m + i) * n : nullptr;
        if (int rc = svmc_payoff_sums(s->snap + static_cast<size_t>(i) * n, qsnap, n, c.forwards[i], c.ttms[i],
                                      s->spot + 2 * i, c.strikes + k0, c.types + k0, shifts.data() + k0, k, variable_type,
                                      s->sums + 3 * k0, s->ws, s->ws_bytes, s->stream))
            return rc;
    }
    std::vector<double> sums(3 * c.offsets[c.m]);
    SVMC_HIP_TRY(hipMemcpyAsync(sums.data(), s->sums, sums.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    SVMC_HIP_TRY(hipStreamSynchronize(s->stream));
    for (int i = 0; i < c.m; ++i) {
        const size_t k0 = c.offsets[i], k = c.offsets[i + 1] - k0;
        if (int rc = svmc_payoff_finalize(sums.data() + 3 * k0, shifts.data() + k0, k, c.discfactors[i],
                                          static_cast<double>(n), prices + k0, stderrs + k0))
            return rc;
    }
    return SVMC_OK;
}

}  // namespace svmc

using namespace svmc;

extern "C" {

int svmc_session_create(svmc_session_t *session, size_t n_path, int max_expiries, size_t max_strikes_total)
{
    SVMC_REQUIRE(session != nullptr, "svmc_session_create: null output");
    SVMC_REQUIRE(n_path > 0 && max_expiries > 0 && max_strikes_total > 0, "svmc_session_create: sizes must be positive");
    Session *s = new Session;
    s->n_path = n_path;
    s->max_expiries = max_expiries;
    s->max_strikes = max_strikes_total;
    size_t ws = 0;
    int rc = svmc_slice_workspace_bytes(n_path, &ws);
    s->ws_bytes = ws;
    const size_t nb = n_path * sizeof(double);
    hipError_t e = hipSuccess;
    if (rc == SVMC_OK) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->x), nb);
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->vol), nb);
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->qvar), nb);
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->snap), 2 * static_cast<size_t>(max_expiries) * nb);
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->spot), 2 * static_cast<size_t>(max_expiries) * sizeof(double));
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->sums), 3 * max_strikes_total * sizeof(double));
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(&s->ws, ws);
    if (rc != SVMC_OK || e != hipSuccess) {
        session_release(s);
        return rc != SVMC_OK ? rc : fail(SVMC_ERR_HIP, std::string("svmc_session_create: ") + hipGetErrorString(e));
    }
    *session = reinterpret_cast<svmc_session_t>(s);
    return SVMC_OK;
}

int svmc_session_destroy(svmc_session_t session)
{
    session_release(reinterpret_cast<Session *>(session));
    return SVMC_OK;
}

int svmc_logsv_chain_price(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                           const double *discfactors_host, const double *vol_backbone_etas_host, int n_expiries,
                           const double *strikes_host, const int8_t *types_host, const size_t *strike_offsets_host,
                           double v0, double theta, double kappa1, double kappa2, double beta, double volvol,
                           int is_spot_measure, int nb_steps_per_year, int variable_type, uint64_t seed,
                           uint32_t call_id, double *prices_host, double *stderrs_host)
{
    const char *fn = "svmc_logsv_chain_price";
    Session *s = reinterpret_cast<Session *>(session);
    const ChainView c = {n_expiries, ttms_host, forwards_host, discfactors_host, strikes_host, types_host, strike_offsets_host};
    if (int rc = check_chain(fn, s, c, variable_type, prices_host, stderrs_host)) return rc;
    SVMC_REQUIRE(nb_steps_per_year > 0, "svmc_logsv_chain_price: nb_steps_per_year must be positive");
    const size_t n = s->n_path;
    if (int rc = svmc_fill_state(s->x, s->vol, s->qvar, n, 0.0, v0, 0.0, s->stream)) return rc;           // :832-834
    double t0 = 0.0;
    uint32_t step0 = 0;
    for (int i = 0; i < c.m; ++i) {                                                                       // :840-865
        int nb;
        double dt;
        time_grid(c.ttms[i] - t0, nb_steps_per_year, nb, dt);
        const double eta = vol_backbone_etas_host ? vol_backbone_etas_host[i] : 1.0;
        double *qsnap = (variable_type == SVMC_Q_VAR) ? s->snap + static_cast<size_t>(c.m + i) * n : nullptr;
        if (int rc = svmc_logsv_slice_rng(s->x, s->vol, s->qvar, n, nb, dt, theta, kappa1, kappa2, beta, volvol, eta,
                                          is_spot_measure, seed, call_id, 0, step0, c.forwards[i],
                                          s->snap + static_cast<size_t>(i) * n, qsnap, s->spot + 2 * i, s->ws, s->ws_bytes,
                                          s->stream))
            return rc;
        step0 += static_cast<uint32_t>(nb);
        t0 = c.ttms[i];
    }
    return reduce_and_finalize(s, c, variable_type, prices_host, stderrs_host);
}

int svmc_logsv_chain_price_fixed(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                                 const double *discfactors_host, const double *vol_backbone_etas_host, int n_expiries,
                                 const double *strikes_host, const int8_t *types_host, const size_t *strike_offsets_host,
                                 double v0, double theta, double kappa1, double kappa2, double beta, double volvol,
                                 int is_spot_measure, int variable_type, const double *const *W0s, const double *const *W1s,
                                 const int *nb_steps_host, const double *dts_host, size_t ldw, double *prices_host,
                                 double *stderrs_host)
{
    const char *fn = "svmc_logsv_chain_price_fixed";
    Session *s = reinterpret_cast<Session *>(session);
    const ChainView c = {n_expiries, ttms_host, forwards_host, discfactors_host, strikes_host, types_host, strike_offsets_host};
    if (int rc = check_chain(fn, s, c, variable_type, prices_host, stderrs_host)) return rc;
    SVMC_REQUIRE(W0s && W1s && nb_steps_host && dts_host, "svmc_logsv_chain_price_fixed: null randoms / grids");
    const size_t n = s->n_path;
    if (int rc = svmc_fill_state(s->x, s->vol, s->qvar, n, 0.0, v0, 0.0, s->stream)) return rc;           // :1128-1130
    for (int i = 0; i < c.m; ++i) {                                                                       // :1136-1160
        const double eta = vol_backbone_etas_host ? vol_backbone_etas_host[i] : 1.0;
        double *qsnap = (variable_type == SVMC_Q_VAR) ? s->snap + static_cast<size_t>(c.m + i) * n : nullptr;
        if (int rc = svmc_logsv_slice_w(s->x, s->vol, s->qvar, n, nb_steps_host[i], dts_host[i], theta, kappa1, kappa2,
                                        beta, volvol, eta, is_spot_measure, W0s[i], W1s[i], ldw, c.forwards[i],
                                        s->snap + static_cast<size_t>(i) * n, qsnap, s->spot + 2 * i, s->ws, s->ws_bytes,
                                        s->stream))
            return rc;
    }
    return reduce_and_finalize(s, c, variable_type, prices_host, stderrs_host);
}

int svmc_heston_chain_price(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                            const double *discfactors_host, int n_expiries, const double *strikes_host,
                            const int8_t *types_host, const size_t *strike_offsets_host, double v0, double theta,
                            double kappa, double rho, double volvol, int scheme, int nb_steps_per_year,
                            int variable_type, uint64_t seed, uint32_t call_id, double *prices_host,
                            double *stderrs_host)
{
    const char *fn = "svmc_heston_chain_price";
    Session *s = reinterpret_cast<Session *>(session);
    const ChainView c = {n_expiries, ttms_host, forwards_host, discfactors_host, strikes_host, types_host, strike_offsets_host};
    if (int rc = check_chain(fn, s, c, variable_type, prices_host, stderrs_host)) return rc;
    SVMC_REQUIRE(nb_steps_per_year > 0, "svmc_heston_chain_price: nb_steps_per_year must be positive");
    const size_t n = s->n_path;
    if (int rc = svmc_fill_state(s->x, s->vol, s->qvar, n, 0.0, v0, 0.0, s->stream)) return rc;           // :303-305
    double t0 = 0.0;
    uint32_t step0 = 0;
    for (int i = 0; i < c.m; ++i) {                                                                       // :308-329
        int nb;
        double dt;
        time_grid(c.ttms[i] - t0, nb_steps_per_year, nb, dt);
        double *qsnap = (variable_type == SVMC_Q_VAR) ? s->snap + static_cast<size_t>(c.m + i) * n : nullptr;
        if (int rc = svmc_heston_slice_rng(s->x, s->vol, s->qvar, n, nb, dt, theta, kappa, rho, volvol, scheme, seed,
                                           call_id, 0, step0, c.forwards[i], s->snap + static_cast<size_t>(i) * n, qsnap,
                                           s->spot + 2 * i, s->ws, s->ws_bytes, s->stream))
            return rc;
        step0 += static_cast<uint32_t>(nb);
        t0 = c.ttms[i];
    }
    return reduce_and_finalize(s, c, variable_type, prices_host, stderrs_host);
}

int svmc_session_state(svmc_session_t session, double *x_host, double *vol_host, double *qvar_host)
{
    Session *s = reinterpret_cast<Session *>(session);
    SVMC_REQUIRE(s != nullptr, "svmc_session_state: null session");
    const size_t nb = s->n_path * sizeof(double);
    if (x_host) SVMC_HIP_TRY(hipMemcpyAsync(x_host, s->x, nb, hipMemcpyDeviceToHost, s->stream));
    if (vol_host) SVMC_HIP_TRY(hipMemcpyAsync(vol_host, s->vol, nb, hipMemcpyDeviceToHost, s->stream));
    if (qvar_host) SVMC_HIP_TRY(hipMemcpyAsync(qvar_host, s->qvar, nb, hipMemcpyDeviceToHost, s->stream));
    SVMC_HIP_TRY(hipStreamSynchronize(s->stream));
    return SVMC_OK;
}

}  // extern "C"
