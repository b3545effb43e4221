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
#include <cstdlib>
#include <vector>

#include "svmc_internal.h"
#include "svmc_black.h"
#include <limits>
#include <cstring>

namespace svmc {

constexpr double IV_VOL_LO = 1e-6, IV_VOL_HI = 10.0;       // the bracket of the implied vols (data/option_chain.py's host mirror)

// A captured chain on fixed randoms (svmc_logsv_chain_price_fixed): everything that shapes the launches -- chain,
// randoms, step counts -- is frozen in `key`; the model constants live in `params_dev`, refreshed from the pinned
// `params_host` by the graph's first node, so one hipGraphLaunch re-prices the chain for a new parameter set.
struct FixedGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<unsigned char> key;
    double *params_host = nullptr, *params_dev = nullptr, *sums_host = nullptr;   // pinned / device / pinned
    size_t params_doubles = 0, sums_doubles = 0;
    // implied vols on the device (svmc_logsv_chain_price_fixed_iv): per-quote constants, results device / pinned
    double *quotes_dev = nullptr, *ivols_dev = nullptr, *ivols_host = nullptr;
};

struct Session {
    // multi-GPU (svmc_session_set_comm): this session holds the paths [path_offset, path_offset + n_path) of a job of
    // n_total paths spread over `world` ranks; `comm` is the ncclComm_t the two reductions of a chain go through
    void *comm = nullptr;
    // ... or a caller-supplied all-reduce (svmc_session_set_reducer): another transport than RCCL, or a test harness
    svmc_all_reduce_fn reduce_fn = nullptr;
    void *reduce_user = nullptr;
    bool sharded() const { return comm != nullptr || reduce_fn != nullptr; }
    int rank = 0, world = 1;
    uint64_t n_total = 0, path_offset = 0;
    bool use_graphs = true;
    size_t graph_launches = 0;
    FixedGraph fixed;
    FixedGraph fixed_sets;                  // svmc_logsv_chain_price_fixed_sets: several parameter sets per replay
    FixedGraph frozen[MAX_FUSED_SETS + 1];  // svmc_logsv_chain_price_frozen_sets: one captured chain per set count (an SLSQP
                                            // iterate alternates between its base point, 1 set, and its bumped neighbours)
    size_t n_path = 0;
    int max_expiries = 0;
    size_t max_strikes = 0;
    double *x = nullptr, *vol = nullptr, *qvar = nullptr, *snap = nullptr, *spot = nullptr, *sums = nullptr;
    double *sums_pinned = nullptr;          // page-locked landing buffer of the payoff sums: 3 max_strikes doubles (a copy into
                                            // pageable memory costs 16 us more per chain: tools/ubench/sync_latency.py)
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // the generators' per-wave spot partials [2 max_expiries][wave_rows(n_path)], apart from `ws` (the payoff launch's block
    // partials): on one device the payoff kernel reads them while it writes those (one_device_tail)
    double *spot_ws = nullptr;
    size_t spot_ws_bytes = 0;
    hipStream_t stream = nullptr;
    bool owns_state = true, owns_stream = true;   // false: svmc_session_create_on -- the caller's state arrays / stream
    // svmc_session_time_stepping: HIP events around the stepping launch (+ its spot-sum reduce) of the on-device-RNG chain
    // drivers, on the session's stream -- what a host that times the dominant kernel (bench.py) reads back after the call
    bool time_stepping = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_stepping_ms = -1.0f;
};

// events around a stepping call of a timed session (no-ops otherwise)
static void stepping_begin(Session *s)
{
    if (s->time_stepping && s->ev0 != nullptr) (void)hipEventRecord(s->ev0, s->stream);
}
static void stepping_end(Session *s)
{
    if (s->time_stepping && s->ev1 != nullptr) (void)hipEventRecord(s->ev1, s->stream);
}
// after the chain's synchronisation
static void stepping_read(Session *s)
{
    s->last_stepping_ms = -1.0f;
    if (s->time_stepping && s->ev0 != nullptr && s->ev1 != nullptr) {
        float ms = -1.0f;
        if (hipEventElapsedTime(&ms, s->ev0, s->ev1) == hipSuccess) s->last_stepping_ms = ms;
    }
}

static void fixed_graph_release(FixedGraph &g)
{
    if (g.exec != nullptr) (void)hipGraphExecDestroy(g.exec);
    if (g.graph != nullptr) (void)hipGraphDestroy(g.graph);
    if (g.params_host != nullptr) (void)hipHostFree(g.params_host);
    if (g.sums_host != nullptr) (void)hipHostFree(g.sums_host);
    if (g.params_dev != nullptr) (void)hipFree(g.params_dev);
    if (g.quotes_dev != nullptr) (void)hipFree(g.quotes_dev);
    if (g.ivols_dev != nullptr) (void)hipFree(g.ivols_dev);
    if (g.ivols_host != nullptr) (void)hipHostFree(g.ivols_host);
    g = FixedGraph();
}

static void session_release(Session *s)
{
    if (s == nullptr) return;
    fixed_graph_release(s->fixed);
    fixed_graph_release(s->fixed_sets);
    for (FixedGraph &g : s->frozen) fixed_graph_release(g);
    if (s->owns_state)
        for (void *p : {static_cast<void *>(s->x), static_cast<void *>(s->vol), static_cast<void *>(s->qvar)})
            if (p != nullptr) (void)hipFree(p);
    for (void *p : {static_cast<void *>(s->snap), static_cast<void *>(s->spot), static_cast<void *>(s->sums), s->ws,
                    static_cast<void *>(s->spot_ws)})
        if (p != nullptr) (void)hipFree(p);
    if (s->sums_pinned != nullptr) (void)hipHostFree(s->sums_pinned);
    if (s->ev0 != nullptr) (void)hipEventDestroy(s->ev0);
    if (s->ev1 != nullptr) (void)hipEventDestroy(s->ev1);
    if (s->owns_stream && s->stream != nullptr) (void)hipStreamDestroy(s->stream);
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

// phase 3 of mc_chain.py: per-strike sums of every slice, queued on the session's stream
static int enqueue_payoff_sums(Session *s, const ChainView &c, int variable_type, std::vector<double> &shifts)
{
    const size_t n = s->n_path;
    shifts.resize(c.offsets[c.m]);
    std::vector<const double *> xs(c.m), qs(c.m);
    for (int i = 0; i < c.m; ++i) {
        for (size_t k = c.offsets[i]; k < c.offsets[i + 1]; ++k)
            shifts[k] = payoff_shift(c.strikes[k], c.types[k], c.forwards[i], variable_type);
        xs[i] = s->snap + static_cast<size_t>(i) * n;
        qs[i] = s->snap + static_cast<size_t>(c.m + i) * n;
    }
    return svmc_payoff_sums_chain(xs.data(), variable_type == SVMC_Q_VAR ? qs.data() : nullptr, n, c.forwards, c.ttms,
                                  s->spot, c.m, c.strikes, c.types, shifts.data(), c.offsets, variable_type, s->sums,
                                  s->ws, s->ws_bytes, s->stream);
}

// the same for parameter set `set` of P (svmc_logsv_chain_price_fixed_sets): its snapshot rows, spot sums and output
// block -- one launch pair per set with the single-set launch shape, hence the single-set bits
static int enqueue_payoff_sums_of_set(Session *s, const ChainView &c, int variable_type, const std::vector<double> &shifts,
                                      int set, int n_sets)
{
    const size_t n = s->n_path, K = c.offsets[c.m];
    std::vector<const double *> xs(c.m), qs(c.m);
    for (int i = 0; i < c.m; ++i) {
        xs[i] = s->snap + static_cast<size_t>(set * c.m + i) * n;
        qs[i] = s->snap + static_cast<size_t>((n_sets + set) * c.m + i) * n;
    }
    return svmc_payoff_sums_chain(xs.data(), variable_type == SVMC_Q_VAR ? qs.data() : nullptr, n, c.forwards, c.ttms,
                                  s->spot + 2 * static_cast<size_t>(set) * c.m, c.m, c.strikes, c.types, shifts.data(), c.offsets,
                                  variable_type, s->sums + 3 * K * static_cast<size_t>(set), s->ws, s->ws_bytes, s->stream);
}

// the payoff sums of all n_sets sets: one launch and one column reduce where the chain allows it (the bits of the per-set loop)
static int enqueue_payoff_sums_of_sets(Session *s, const ChainView &c, int variable_type, const std::vector<double> &shifts,
                                       int n_sets)
{
    if (n_sets > 1 && payoff_sets_fit(s->n_path, c.m, c.offsets, c.types, n_sets, s->ws_bytes)) {
        const size_t n = s->n_path;
        std::vector<const double *> xs(c.m), qs(c.m);
        for (int i = 0; i < c.m; ++i) {
            xs[i] = s->snap + static_cast<size_t>(i) * n;
            qs[i] = s->snap + static_cast<size_t>(n_sets * c.m + i) * n;
        }
        return payoff_sums_chain_sets(xs.data(), variable_type == SVMC_Q_VAR ? qs.data() : nullptr, n, c.forwards, c.ttms, s->spot,
                                      c.m, c.strikes, c.types, shifts.data(), c.offsets, variable_type, s->sums, s->ws, s->ws_bytes,
                                      s->stream, n_sets, static_cast<size_t>(c.m) * n, static_cast<size_t>(c.m) * n,
                                      2 * static_cast<size_t>(c.m));
    }
    for (int q = 0; q < n_sets; ++q)
        if (int rc = enqueue_payoff_sums_of_set(s, c, variable_type, shifts, q, n_sets)) return rc;
    return SVMC_OK;
}

// phase 4: host finalisation of the downloaded sums (utils/mc_payoffs.py:85-88); the standard error divides by the
// path count of the WHOLE job
static int finalize_prices(const Session *s, const ChainView &c, const double *sums, const std::vector<double> &shifts,
                           double *prices, double *stderrs)
{
    const double n_all = static_cast<double>(s->sharded() ? s->n_total : s->n_path);
    for (int i = 0; i < c.m; ++i) {
        const size_t k0 = c.offsets[i], k = c.offsets[i + 1] - k0;
        if (int rc = svmc_payoff_finalize(sums + 3 * k0, shifts.data() + k0, k, c.discfactors[i], n_all, prices + k0,
                                          stderrs + k0))
            return rc;
    }
    return SVMC_OK;
}

// The two cross-rank couplings of compute_mc_vars_payoff, as in-place fp64 sum all-reduces on the session's stream,
// stream-ordered against the kernels on either side (no host synchronisation): phase 2 = [sum F exp(x), count] per
// expiry (utils/mc_payoffs.py:61-63), phase 3' = [sum d, sum d^2, count] per strike (:85-86).  No-ops without a comm.
static int all_reduce(Session *s, double *buf, size_t n)
{
    if (!s->sharded() || n == 0) return SVMC_OK;
    if (s->reduce_fn != nullptr) {
        last_error_ref().clear();
        const int rc = s->reduce_fn(s->reduce_user, buf, n, reinterpret_cast<svmc_stream_t>(s->stream));
        if (rc == SVMC_OK) return SVMC_OK;
        const std::string why = last_error_ref();          // what the callback itself said, if it went through this library
        return fail(rc, "the session's all-reduce callback failed" + (why.empty() ? std::string() : ": " + why));
    }
    return svmc_rccl_all_reduce_sum(s->comm, buf, n, reinterpret_cast<svmc_stream_t>(s->stream));
}

// The tail of an on-device-RNG chain on ONE device (no communicator): the stepping launch left its per-wave spot partials in
// s->spot_ws unreduced; the payoff kernel sums them itself (up to 2048 rows; a reduce launch ahead of it otherwise) and
// chain_finish_kernel, a wave per quote, forms the column sums in reduce_columns_kernel's order (the bits of the five-node tail
// below) and stores them straight into the pinned host buffer.  Whether a chain takes it: one_device_tail().
static bool one_device_tail(const Session *s, const ChainView &c)
{
    static const bool off = getenv("SVMC_CHAIN_TAIL_NODES") != nullptr && atoi(getenv("SVMC_CHAIN_TAIL_NODES")) == 5;   // A/B, tests
    return !off && !s->sharded() && c.m <= MAX_FUSED_SLICES &&
           static_cast<size_t>(wave_rows(s->n_path)) * 2 * static_cast<size_t>(c.m) * sizeof(double) <= s->spot_ws_bytes &&
           payoff_sets_fit(s->n_path, c.m, c.offsets, c.types, 1, s->ws_bytes);
}

static void payoff_shifts_of(const ChainView &c, int variable_type, std::vector<double> &shifts)
{
    shifts.resize(c.offsets[c.m]);
    for (int i = 0; i < c.m; ++i)
        for (size_t k = c.offsets[i]; k < c.offsets[i + 1]; ++k)
            shifts[k] = payoff_shift(c.strikes[k], c.types[k], c.forwards[i], variable_type);
}

static int enqueue_one_device_tail(Session *s, const ChainView &c, int variable_type, const std::vector<double> &shifts, double *sums_out)
{
    const size_t n = s->n_path;
    std::vector<const double *> xs(c.m), qs(c.m);
    for (int i = 0; i < c.m; ++i) {
        xs[i] = s->snap + static_cast<size_t>(i) * n;
        qs[i] = s->snap + static_cast<size_t>(c.m + i) * n;
    }
    const bool in_kernel = spot_sums_in_payoff_kernel(n);
    if (!in_kernel)
        if (int rc = reduce_spot_partials(s->spot_ws, n, 2 * c.m, s->spot, s->stream)) return rc;
    return chain_payoff_and_finish(xs.data(), variable_type == SVMC_Q_VAR ? qs.data() : nullptr, n, c.forwards, c.ttms, s->spot,
                                   in_kernel ? s->spot_ws : nullptr, c.m, c.strikes, c.types, shifts.data(), c.offsets, variable_type,
                                   s->ws, s->ws_bytes, s->stream, sums_out);
}

// partials_pending: the stepping launch(es) left the per-wave spot partials in s->spot_ws unreduced (the on-device-RNG drivers);
// false: s->spot already holds this rank's spot sums (the slice-by-slice fixed-randoms route)
static int reduce_and_finalize(Session *s, const ChainView &c, int variable_type, double *prices, double *stderrs,
                               bool partials_pending)
{
    if (partials_pending && one_device_tail(s, c)) {
        std::vector<double> shifts;
        payoff_shifts_of(c, variable_type, shifts);
        if (int rc = enqueue_one_device_tail(s, c, variable_type, shifts, s->sums_pinned)) return rc;
        SVMC_HIP_TRY(hipStreamSynchronize(s->stream));
        stepping_read(s);
        return finalize_prices(s, c, s->sums_pinned, shifts, prices, stderrs);
    }
    if (partials_pending)
        if (int rc = reduce_spot_partials(s->spot_ws, s->n_path, 2 * c.m, s->spot, s->stream)) return rc;
    if (int rc = all_reduce(s, s->spot, 2 * static_cast<size_t>(c.m))) return rc;
    std::vector<double> shifts;
    if (int rc = enqueue_payoff_sums(s, c, variable_type, shifts)) return rc;
    if (int rc = all_reduce(s, s->sums, 3 * c.offsets[c.m])) return rc;
    SVMC_HIP_TRY(hipMemcpyAsync(s->sums_pinned, s->sums, 3 * c.offsets[c.m] * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    SVMC_HIP_TRY(hipStreamSynchronize(s->stream));
    stepping_read(s);
    return finalize_prices(s, c, s->sums_pinned, shifts, prices, stderrs);
}

template <class T>
static void key_append(std::vector<unsigned char> &key, const T *p, size_t count)
{
    const unsigned char *b = reinterpret_cast<const unsigned char *>(p);
    key.insert(key.end(), b, b + count * sizeof(T));
}

}  // namespace svmc

using namespace svmc;

extern "C" {

static int session_create(const char *fn, svmc_session_t *session, size_t n_path, int max_expiries, size_t max_strikes_total,
                          double *x, double *vol, double *qvar, bool borrow_stream, hipStream_t stream_in);

int svmc_session_create(svmc_session_t *session, size_t n_path, int max_expiries, size_t max_strikes_total)
{
    return session_create("svmc_session_create", session, n_path, max_expiries, max_strikes_total, nullptr, nullptr, nullptr, false,
                          nullptr);
}

int svmc_session_create_on(svmc_session_t *session, size_t n_path, int max_expiries, size_t max_strikes_total, double *x,
                           double *vol, double *qvar, uint64_t path_offset, svmc_stream_t stream)
{
    SVMC_REQUIRE(x != nullptr && vol != nullptr && qvar != nullptr, "svmc_session_create_on: null state array");
    if (int rc = session_create("svmc_session_create_on", session, n_path, max_expiries, max_strikes_total, x, vol, qvar, true,
                                as_stream(stream)))
        return rc;
    reinterpret_cast<Session *>(*session)->path_offset = path_offset;
    return SVMC_OK;
}

static int session_create(const char *fn, svmc_session_t *session, size_t n_path, int max_expiries, size_t max_strikes_total,
                          double *x, double *vol, double *qvar, bool borrow_stream, hipStream_t stream_in)
{
    SVMC_REQUIRE(session != nullptr, std::string(fn) + ": null output");
    SVMC_REQUIRE(n_path > 0 && max_expiries > 0 && max_strikes_total > 0, std::string(fn) + ": sizes must be positive");
    Session *s = new Session;
    s->n_path = n_path;
    s->max_expiries = max_expiries;
    s->max_strikes = max_strikes_total;
    s->owns_state = x == nullptr;
    s->owns_stream = !borrow_stream;
    s->x = x;
    s->vol = vol;
    s->qvar = qvar;
    s->stream = stream_in;
    size_t ws = 0;
    int rc = svmc_slice_workspace_bytes(n_path, &ws);
    // the multi-set replay (svmc_logsv_chain_price_fixed_sets) writes two partial columns per (expiry, set): a session created
    // for m x P chains (max_expiries = m P) holds them all, whatever n_path -- the slice workspace alone covers 16 columns pairs
    const size_t fused_sets = static_cast<size_t>(wave_rows(n_path)) * 2 * static_cast<size_t>(max_expiries) * sizeof(double);
    if (fused_sets > ws) ws = fused_sets;
    s->ws_bytes = ws;
    const size_t nb = n_path * sizeof(double);
    hipError_t e = hipSuccess;
    if (rc == SVMC_OK && s->owns_stream) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (rc == SVMC_OK && e == hipSuccess && s->owns_state) e = hipMalloc(reinterpret_cast<void **>(&s->x), nb);
    if (rc == SVMC_OK && e == hipSuccess && s->owns_state) e = hipMalloc(reinterpret_cast<void **>(&s->vol), nb);
    if (rc == SVMC_OK && e == hipSuccess && s->owns_state) e = hipMalloc(reinterpret_cast<void **>(&s->qvar), nb);
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->snap), 2 * static_cast<size_t>(max_expiries) * nb);
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->spot), 2 * static_cast<size_t>(max_expiries) * sizeof(double));
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->sums), 3 * max_strikes_total * sizeof(double));
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(&s->ws, ws);
    s->spot_ws_bytes = (fused_sets > 2 * sizeof(double)) ? fused_sets : 2 * sizeof(double);
    if (rc == SVMC_OK && e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&s->spot_ws), s->spot_ws_bytes);
    if (rc == SVMC_OK && e == hipSuccess)
        e = hipHostMalloc(reinterpret_cast<void **>(&s->sums_pinned), 3 * max_strikes_total * sizeof(double), hipHostMallocDefault);
    if (rc != SVMC_OK || e != hipSuccess) {
        session_release(s);
        return rc != SVMC_OK ? rc : fail(SVMC_ERR_HIP, std::string(fn) + ": " + hipGetErrorString(e));
    }
    *session = reinterpret_cast<svmc_session_t>(s);
    return SVMC_OK;
}

int svmc_session_set_comm(svmc_session_t session, svmc_comm_t comm, int rank, int world, uint64_t n_path_total,
                          uint64_t path_offset)
{
    Session *s = reinterpret_cast<Session *>(session);
    SVMC_REQUIRE(s != nullptr, "svmc_session_set_comm: null session");
    s->reduce_fn = nullptr;
    s->reduce_user = nullptr;
    if (comm == nullptr) {                     // back to a single-GPU session
        s->comm = nullptr;
        s->rank = 0;
        s->world = 1;
        s->n_total = 0;
        s->path_offset = 0;
        return SVMC_OK;
    }
    SVMC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "svmc_session_set_comm: need 0 <= rank < world");
    SVMC_REQUIRE(path_offset + s->n_path <= n_path_total, "svmc_session_set_comm: the session's paths exceed the job");
    s->comm = comm;
    s->rank = rank;
    s->world = world;
    s->n_total = n_path_total;
    s->path_offset = path_offset;
    return SVMC_OK;
}

int svmc_session_set_reducer(svmc_session_t session, svmc_all_reduce_fn fn, void *user, int rank, int world,
                             uint64_t n_path_total, uint64_t path_offset)
{
    Session *s = reinterpret_cast<Session *>(session);
    SVMC_REQUIRE(s != nullptr, "svmc_session_set_reducer: null session");
    if (fn == nullptr) return svmc_session_set_comm(session, nullptr, 0, 1, 0, 0);
    SVMC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "svmc_session_set_reducer: need 0 <= rank < world");
    SVMC_REQUIRE(path_offset + s->n_path <= n_path_total, "svmc_session_set_reducer: the session's paths exceed the job");
    s->comm = nullptr;
    s->reduce_fn = fn;
    s->reduce_user = user;
    s->rank = rank;
    s->world = world;
    s->n_total = n_path_total;
    s->path_offset = path_offset;
    return SVMC_OK;
}

int svmc_session_time_stepping(svmc_session_t session, int enable)
{
    Session *s = reinterpret_cast<Session *>(session);
    SVMC_REQUIRE(s != nullptr, "svmc_session_time_stepping: null session");
    if (enable && s->ev0 == nullptr) {
        SVMC_HIP_TRY(hipEventCreate(&s->ev0));
        SVMC_HIP_TRY(hipEventCreate(&s->ev1));
    }
    s->time_stepping = enable != 0;
    s->last_stepping_ms = -1.0f;
    return SVMC_OK;
}

int svmc_session_last_stepping_ms(svmc_session_t session, float *ms)
{
    Session *s = reinterpret_cast<Session *>(session);
    SVMC_REQUIRE(s != nullptr && ms != nullptr, "svmc_session_last_stepping_ms: null argument");
    *ms = s->last_stepping_ms;
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
    // the start state (0, v0, 0) of every path (:832-834) travels as three constants: no fill launch
    double t0 = 0.0;
    std::vector<int> nbs(c.m);
    std::vector<double> dts(c.m);
    for (int i = 0; i < c.m; ++i) {                                                                       // :840-865
        time_grid(c.ttms[i] - t0, nb_steps_per_year, nbs[i], dts[i]);
        t0 = c.ttms[i];
    }
    // the stepping launch (a single expiry: the plain slice kernel -- the same bits, and the one bench.py profiles); its per-wave
    // spot partials stay in s->spot_ws, reduce_and_finalize decides who sums them
    double *qsnap = (variable_type == SVMC_Q_VAR) ? s->snap + static_cast<size_t>(c.m) * n : nullptr;
    stepping_begin(s);
    if (c.m > MAX_FUSED_SLICES) {          // more expiries than one stepping launch takes: launch by launch, each reducing its own
        if (int rc = svmc_logsv_chain_rng_from(0.0, v0, 0.0, s->x, s->vol, s->qvar, n, c.m, nbs.data(), dts.data(), vol_backbone_etas_host,
                                               c.forwards, theta, kappa1, kappa2, beta, volvol, is_spot_measure, seed, call_id,
                                               s->path_offset, 0, s->snap, qsnap, s->spot, s->ws, s->ws_bytes, s->stream))
            return rc;
        stepping_end(s);
        return reduce_and_finalize(s, c, variable_type, prices_host, stderrs_host, false);
    }
    if (int rc = logsv_step_partials(v0, s->x, s->vol, s->qvar, n, c.m, nbs.data(), dts.data(), vol_backbone_etas_host, c.forwards, theta,
                                     kappa1, kappa2, beta, volvol, is_spot_measure, seed, call_id, s->path_offset, s->snap, qsnap,
                                     s->spot_ws, s->spot_ws_bytes, s->stream))
        return rc;
    stepping_end(s);
    return reduce_and_finalize(s, c, variable_type, prices_host, stderrs_host, true);
}

int svmc_logsv_chain_price_fixed(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                                 const double *discfactors_host, const double *vol_backbone_etas_host, int n_expiries,
                                 const double *strikes_host, const int8_t *types_host, const size_t *strike_offsets_host,
                                 double v0, double theta, double kappa1, double kappa2, double beta, double volvol,
                                 int is_spot_measure, int variable_type, const double *const *W0s, const double *const *W1s,
                                 const int *nb_steps_host, const double *dts_host, size_t ldw, double *prices_host,
                                 double *stderrs_host)
{
    return svmc_logsv_chain_price_fixed_iv(session, ttms_host, forwards_host, discfactors_host, vol_backbone_etas_host,
                                           n_expiries, strikes_host, types_host, strike_offsets_host, v0, theta, kappa1, kappa2,
                                           beta, volvol, is_spot_measure, variable_type, W0s, W1s, nb_steps_host, dts_host, ldw,
                                           prices_host, stderrs_host, nullptr);
}

// host side of the implied vols for the routes that do not replay a graph (multi-GPU, graphs off): the same solver
static void implied_vols_on_host(const ChainView &c, int variable_type, const double *prices, double *ivols)
{
    for (int i = 0; i < c.m; ++i)
        for (size_t k = c.offsets[i]; k < c.offsets[i + 1]; ++k) {
            // quotes on the log-return only: options on the realised variance have no Black vol on the forward; inverse
            // options (IC / IP) as in chain_implied_vols_kernel: the vanilla inversion of price x forward
            const bool call = c.types[k] == SVMC_CALL || c.types[k] == SVMC_INV_CALL;
            const double px = c.types[k] >= SVMC_INV_CALL ? prices[k] * c.forwards[i] : prices[k];
            ivols[k] = variable_type == SVMC_LOG_RETURN
                           ? black_implied_vol(px, c.strikes[k], call, c.forwards[i], c.ttms[i], c.discfactors[i], IV_VOL_LO, IV_VOL_HI)
                           : std::numeric_limits<double>::quiet_NaN();
        }
}

int svmc_logsv_chain_price_fixed_iv(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                                    const double *discfactors_host, const double *vol_backbone_etas_host, int n_expiries,
                                    const double *strikes_host, const int8_t *types_host, const size_t *strike_offsets_host,
                                    double v0, double theta, double kappa1, double kappa2, double beta, double volvol,
                                    int is_spot_measure, int variable_type, const double *const *W0s,
                                    const double *const *W1s, const int *nb_steps_host, const double *dts_host, size_t ldw,
                                    double *prices_host, double *stderrs_host, double *ivols_host)
{
    const char *fn = "svmc_logsv_chain_price_fixed";
    Session *s = reinterpret_cast<Session *>(session);
    const ChainView c = {n_expiries, ttms_host, forwards_host, discfactors_host, strikes_host, types_host, strike_offsets_host};
    if (int rc = check_chain(fn, s, c, variable_type, prices_host, stderrs_host)) return rc;
    SVMC_REQUIRE(W0s && W1s && nb_steps_host && dts_host, "svmc_logsv_chain_price_fixed: null randoms / grids");
    const size_t n = s->n_path;
    if (s->use_graphs && !s->sharded()) {
        // ---- replay path: the launch structure is captured once per (chain, randoms) and replayed per parameter set
        std::vector<unsigned char> key;
        const int want_iv = (ivols_host != nullptr && variable_type == SVMC_LOG_RETURN) ? 1 : 0;
        key_append(key, &c.m, 1);
        key_append(key, &variable_type, 1);
        key_append(key, &want_iv, 1);
        key_append(key, &ldw, 1);
        key_append(key, c.ttms, c.m);
        key_append(key, c.discfactors, want_iv ? c.m : 0);
        key_append(key, c.forwards, c.m);
        key_append(key, c.offsets, c.m + 1);
        key_append(key, c.strikes, c.offsets[c.m]);
        key_append(key, c.types, c.offsets[c.m]);
        key_append(key, W0s, c.m);
        key_append(key, W1s, c.m);
        key_append(key, nb_steps_host, c.m);
        FixedGraph &g = s->fixed;
        const size_t n_params = 1 + static_cast<size_t>(c.m) * LOGSV_CONSTS_DOUBLES, n_sums = 3 * c.offsets[c.m];
        std::vector<double> shifts(c.offsets[c.m]);
        for (int i = 0; i < c.m; ++i)
            for (size_t k = c.offsets[i]; k < c.offsets[i + 1]; ++k)
                shifts[k] = payoff_shift(c.strikes[k], c.types[k], c.forwards[i], variable_type);
        if (g.exec == nullptr || g.key != key) {
            fixed_graph_release(g);
            g.params_doubles = n_params;
            g.sums_doubles = n_sums;
            SVMC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g.params_host), n_params * sizeof(double), hipHostMallocDefault));
            SVMC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g.sums_host), (n_sums ? n_sums : 1) * sizeof(double), hipHostMallocDefault));
            SVMC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.params_dev), n_params * sizeof(double)));
            const size_t n_quotes = c.offsets[c.m];
            if (want_iv && n_quotes) {
                // the quotes' constants are part of the key: uploaded once, outside the graph
                std::vector<double> quotes(IV_QUOTE_DOUBLES_HOST * n_quotes);
                for (int i = 0; i < c.m; ++i)
                    for (size_t k = c.offsets[i]; k < c.offsets[i + 1]; ++k) {
                        double *qd = quotes.data() + IV_QUOTE_DOUBLES_HOST * k;
                        qd[0] = c.strikes[k];
                        qd[1] = static_cast<double>(c.types[k]);
                        qd[2] = shifts[k];
                        qd[3] = c.forwards[i];
                        qd[4] = c.ttms[i];
                        qd[5] = c.discfactors[i];
                    }
                SVMC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.quotes_dev), quotes.size() * sizeof(double)));
                SVMC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.ivols_dev), n_quotes * sizeof(double)));
                SVMC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g.ivols_host), n_quotes * sizeof(double), hipHostMallocDefault));
                SVMC_HIP_TRY(hipMemcpy(g.quotes_dev, quotes.data(), quotes.size() * sizeof(double), hipMemcpyHostToDevice));
            }
            SVMC_HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed));
            int rc = SVMC_OK;
            hipError_t e = hipMemcpyAsync(g.params_dev, g.params_host, n_params * sizeof(double), hipMemcpyHostToDevice, s->stream);
            if (e == hipSuccess && c.m <= MAX_FUSED_SLICES) {
                // the whole chain in one launch: state initialised in the kernel (:1128-1130), slice loop inside (:1136-1160)
                double *qsnaps = (variable_type == SVMC_Q_VAR) ? s->snap + static_cast<size_t>(c.m) * n : nullptr;
                rc = logsv_chain_w_indirect(s->x, s->vol, s->qvar, n, c.m, nb_steps_host, g.params_dev + 1, g.params_dev, W0s, W1s,
                                            ldw, c.forwards, s->snap, qsnaps, s->spot, s->ws, s->ws_bytes, s->stream);
            } else if (e == hipSuccess) {
                rc = fill_state_indirect(s->x, s->vol, s->qvar, n, g.params_dev, s->stream);                       // :1128-1130
            }
            for (int i = 0; c.m > MAX_FUSED_SLICES && i < c.m && e == hipSuccess && rc == SVMC_OK; ++i) {           // :1136-1160
                double *qsnap = (variable_type == SVMC_Q_VAR) ? s->snap + static_cast<size_t>(c.m + i) * n : nullptr;
                rc = logsv_slice_w_indirect(s->x, s->vol, s->qvar, n, nb_steps_host[i],
                                            g.params_dev + 1 + static_cast<size_t>(i) * LOGSV_CONSTS_DOUBLES, W0s[i], W1s[i],
                                            ldw, c.forwards[i], s->snap + static_cast<size_t>(i) * n, qsnap, s->spot + 2 * i,
                                            s->ws, s->ws_bytes, s->stream);
            }
            std::vector<double> unused;
            if (e == hipSuccess && rc == SVMC_OK) rc = enqueue_payoff_sums(s, c, variable_type, unused);
            // results home: with implied vols the last kernel writes both the sums and the vols into the pinned host buffers
            // (two copy nodes fewer per replay); without, the sums' copy
            if (e == hipSuccess && rc == SVMC_OK && g.ivols_dev != nullptr)
                rc = chain_implied_vols(s->sums, g.quotes_dev, n_quotes, static_cast<double>(s->n_path), IV_VOL_LO, IV_VOL_HI,
                                        g.ivols_host, g.sums_host, s->stream);
            else if (e == hipSuccess && rc == SVMC_OK && n_sums)
                e = hipMemcpyAsync(g.sums_host, s->sums, n_sums * sizeof(double), hipMemcpyDeviceToHost, s->stream);
            const hipError_t e_end = hipStreamEndCapture(s->stream, &g.graph);      // always leave capture mode
            if (rc != SVMC_OK) { fixed_graph_release(g); return rc; }
            if (e != hipSuccess || e_end != hipSuccess) {
                fixed_graph_release(g);
                return fail(SVMC_ERR_HIP, std::string("svmc_logsv_chain_price_fixed: graph capture: ") +
                                              hipGetErrorString(e != hipSuccess ? e : e_end));
            }
            SVMC_HIP_TRY(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
            g.key = key;
        }
        g.params_host[0] = v0;
        for (int i = 0; i < c.m; ++i)
            logsv_consts_to_doubles(dts_host[i], theta, kappa1, kappa2, beta, volvol,
                                    vol_backbone_etas_host ? vol_backbone_etas_host[i] : 1.0, is_spot_measure,
                                    g.params_host + 1 + static_cast<size_t>(i) * LOGSV_CONSTS_DOUBLES);
        for (int i = 0; i < c.m; ++i)
            SVMC_REQUIRE(dts_host[i] > 0.0 && nb_steps_host[i] > 0, "svmc_logsv_chain_price_fixed: dt and nb_steps must be positive");
        SVMC_HIP_TRY(hipGraphLaunch(g.exec, s->stream));
        SVMC_HIP_TRY(hipStreamSynchronize(s->stream));
        ++s->graph_launches;
        if (int rc = finalize_prices(s, c, g.sums_host, shifts, prices_host, stderrs_host)) return rc;
        if (ivols_host != nullptr) {
            if (g.ivols_host != nullptr) memcpy(ivols_host, g.ivols_host, c.offsets[c.m] * sizeof(double));
            else implied_vols_on_host(c, variable_type, prices_host, ivols_host);      // Q_VAR chains: NaN
        }
        return SVMC_OK;
    }
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
    if (int rc = reduce_and_finalize(s, c, variable_type, prices_host, stderrs_host, false)) return rc;
    if (ivols_host != nullptr) implied_vols_on_host(c, variable_type, prices_host, ivols_host);
    return SVMC_OK;
}

int svmc_logsv_chain_price_fixed_sets(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                                      const double *discfactors_host, int n_expiries, const double *strikes_host,
                                      const int8_t *types_host, const size_t *strike_offsets_host, int n_sets,
                                      const double *params_host, int is_spot_measure, int variable_type,
                                      const double *const *W0s, const double *const *W1s, const int *nb_steps_host,
                                      const double *dts_host, size_t ldw, double *prices_host, double *stderrs_host,
                                      double *ivols_host)
{
    const char *fn = "svmc_logsv_chain_price_fixed_sets";
    Session *s = reinterpret_cast<Session *>(session);
    const ChainView c = {n_expiries, ttms_host, forwards_host, discfactors_host, strikes_host, types_host, strike_offsets_host};
    if (int rc = check_chain(fn, s, c, variable_type, prices_host, stderrs_host)) return rc;
    SVMC_REQUIRE(W0s && W1s && nb_steps_host && dts_host && params_host, "svmc_logsv_chain_price_fixed_sets: null randoms / grids / parameters");
    SVMC_REQUIRE(n_sets >= 1, "svmc_logsv_chain_price_fixed_sets: n_sets must be positive");
    const size_t K = c.offsets[c.m], row = 6 + static_cast<size_t>(c.m);          // doubles per parameter set
    // the routes without a multi-set graph (one set, more sets than a launch takes, graphs off, a communicator attached, a
    // session not sized for n_sets chains): the sets one after the other through the single-set entry -- the same numbers
    const bool batched = s->use_graphs && !s->sharded() && n_sets >= 2 && n_sets <= MAX_FUSED_SETS && c.m <= MAX_FUSED_SLICES &&
                         c.m * n_sets <= s->max_expiries && K * static_cast<size_t>(n_sets) <= s->max_strikes &&
                         static_cast<size_t>(wave_rows(s->n_path)) * 2 * static_cast<size_t>(c.m) * n_sets * sizeof(double) <= s->ws_bytes;
    if (!batched) {
        for (int q = 0; q < n_sets; ++q) {
            const double *pr = params_host + row * q;
            if (int rc = svmc_logsv_chain_price_fixed_iv(session, ttms_host, forwards_host, discfactors_host, pr + 6, n_expiries,
                                                         strikes_host, types_host, strike_offsets_host, pr[0], pr[1], pr[2], pr[3],
                                                         pr[4], pr[5], is_spot_measure, variable_type, W0s, W1s, nb_steps_host,
                                                         dts_host, ldw, prices_host + K * q, stderrs_host + K * q,
                                                         ivols_host ? ivols_host + K * q : nullptr))
                return rc;
        }
        return SVMC_OK;
    }
    const size_t n = s->n_path;
    const int P = n_sets;
    for (int i = 0; i < c.m; ++i)
        SVMC_REQUIRE(dts_host[i] > 0.0 && nb_steps_host[i] > 0, "svmc_logsv_chain_price_fixed_sets: dt and nb_steps must be positive");
    std::vector<unsigned char> key;
    const int want_iv = (ivols_host != nullptr && variable_type == SVMC_LOG_RETURN) ? 1 : 0;
    key_append(key, &c.m, 1);
    key_append(key, &P, 1);
    key_append(key, &variable_type, 1);
    key_append(key, &want_iv, 1);
    key_append(key, &ldw, 1);
    key_append(key, c.ttms, c.m);
    key_append(key, c.discfactors, want_iv ? c.m : 0);
    key_append(key, c.forwards, c.m);
    key_append(key, c.offsets, c.m + 1);
    key_append(key, c.strikes, K);
    key_append(key, c.types, K);
    key_append(key, W0s, c.m);
    key_append(key, W1s, c.m);
    key_append(key, nb_steps_host, c.m);
    FixedGraph &g = s->fixed_sets;
    // parameter block: [P] initial volatilities, then [m][P] LogsvConsts
    const size_t n_params = static_cast<size_t>(P) + static_cast<size_t>(c.m) * P * LOGSV_CONSTS_DOUBLES, n_sums = 3 * K * P;
    std::vector<double> shifts(K);
    for (int i = 0; i < c.m; ++i)
        for (size_t k = c.offsets[i]; k < c.offsets[i + 1]; ++k)
            shifts[k] = payoff_shift(c.strikes[k], c.types[k], c.forwards[i], variable_type);
    if (g.exec == nullptr || g.key != key) {
        fixed_graph_release(g);
        g.params_doubles = n_params;
        g.sums_doubles = n_sums;
        SVMC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g.params_host), n_params * sizeof(double), hipHostMallocDefault));
        SVMC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g.sums_host), (n_sums ? n_sums : 1) * sizeof(double), hipHostMallocDefault));
        SVMC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.params_dev), n_params * sizeof(double)));
        const size_t n_quotes = K * P;
        if (want_iv && n_quotes) {
            std::vector<double> quotes(IV_QUOTE_DOUBLES_HOST * n_quotes);
            for (int q = 0; q < P; ++q)
                for (int i = 0; i < c.m; ++i)
                    for (size_t k = c.offsets[i]; k < c.offsets[i + 1]; ++k) {
                        double *qd = quotes.data() + IV_QUOTE_DOUBLES_HOST * (K * q + k);
                        qd[0] = c.strikes[k];
                        qd[1] = static_cast<double>(c.types[k]);
                        qd[2] = shifts[k];
                        qd[3] = c.forwards[i];
                        qd[4] = c.ttms[i];
                        qd[5] = c.discfactors[i];
                    }
            SVMC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.quotes_dev), quotes.size() * sizeof(double)));
            SVMC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.ivols_dev), n_quotes * sizeof(double)));
            SVMC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g.ivols_host), n_quotes * sizeof(double), hipHostMallocDefault));
            SVMC_HIP_TRY(hipMemcpy(g.quotes_dev, quotes.data(), quotes.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        SVMC_HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed));
        int rc = SVMC_OK;
        hipError_t e = hipMemcpyAsync(g.params_dev, g.params_host, n_params * sizeof(double), hipMemcpyHostToDevice, s->stream);
        if (e == hipSuccess) {
            double *qsnaps = (variable_type == SVMC_Q_VAR) ? s->snap + static_cast<size_t>(c.m) * P * n : nullptr;
            rc = logsv_chain_w_sets(n, P, c.m, nb_steps_host, g.params_dev + P, g.params_dev, W0s, W1s, ldw, c.forwards, s->snap,
                                    qsnaps, s->spot, s->ws, s->ws_bytes, s->stream);
        }
        if (e == hipSuccess && rc == SVMC_OK) rc = enqueue_payoff_sums_of_sets(s, c, variable_type, shifts, P);
        if (e == hipSuccess && rc == SVMC_OK && g.ivols_dev != nullptr)       // (the last kernel writes the pinned host buffers)
            rc = chain_implied_vols(s->sums, g.quotes_dev, n_quotes, static_cast<double>(s->n_path), IV_VOL_LO, IV_VOL_HI,
                                    g.ivols_host, g.sums_host, s->stream);
        else if (e == hipSuccess && rc == SVMC_OK && n_sums)
            e = hipMemcpyAsync(g.sums_host, s->sums, n_sums * sizeof(double), hipMemcpyDeviceToHost, s->stream);
        const hipError_t e_end = hipStreamEndCapture(s->stream, &g.graph);      // always leave capture mode
        if (rc != SVMC_OK) { fixed_graph_release(g); return rc; }
        if (e != hipSuccess || e_end != hipSuccess) {
            fixed_graph_release(g);
            return fail(SVMC_ERR_HIP, std::string(fn) + ": graph capture: " + hipGetErrorString(e != hipSuccess ? e : e_end));
        }
        SVMC_HIP_TRY(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
        g.key = key;
    }
    for (int q = 0; q < P; ++q) {
        const double *pr = params_host + row * q;
        g.params_host[q] = pr[0];
        for (int i = 0; i < c.m; ++i)
            logsv_consts_to_doubles(dts_host[i], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6 + i], is_spot_measure,
                                    g.params_host + P + (static_cast<size_t>(i) * P + q) * LOGSV_CONSTS_DOUBLES);
    }
    SVMC_HIP_TRY(hipGraphLaunch(g.exec, s->stream));
    SVMC_HIP_TRY(hipStreamSynchronize(s->stream));
    ++s->graph_launches;
    const double n_all = static_cast<double>(s->n_path);
    for (int q = 0; q < P; ++q)
        for (int i = 0; i < c.m; ++i) {
            const size_t k0 = c.offsets[i], k = c.offsets[i + 1] - k0;
            if (int rc = svmc_payoff_finalize(g.sums_host + 3 * (K * q + k0), shifts.data() + k0, k, c.discfactors[i], n_all,
                                              prices_host + K * q + k0, stderrs_host + K * q + k0))
                return rc;
        }
    if (ivols_host != nullptr) {
        if (g.ivols_host != nullptr) {
            memcpy(ivols_host, g.ivols_host, K * P * sizeof(double));
        } else {
            for (int q = 0; q < P; ++q) implied_vols_on_host(c, variable_type, prices_host + K * q, ivols_host + K * q);
        }
    }
    return SVMC_OK;
}

int svmc_logsv_chain_price_frozen_sets(svmc_session_t session, const double *ttms_host, const double *forwards_host,
                                       const double *discfactors_host, int n_expiries, const double *strikes_host,
                                       const int8_t *types_host, const size_t *strike_offsets_host, int n_sets,
                                       const double *params_host, int is_spot_measure, int variable_type,
                                       const int *nb_steps_host, const double *dts_host, uint64_t seed, uint32_t call_id,
                                       double *prices_host, double *stderrs_host, double *ivols_host)
{
    const char *fn = "svmc_logsv_chain_price_frozen_sets";
    Session *s = reinterpret_cast<Session *>(session);
    const ChainView c = {n_expiries, ttms_host, forwards_host, discfactors_host, strikes_host, types_host, strike_offsets_host};
    if (int rc = check_chain(fn, s, c, variable_type, prices_host, stderrs_host)) return rc;
    SVMC_REQUIRE(nb_steps_host && dts_host && params_host, std::string(fn) + ": null grids / parameters");
    SVMC_REQUIRE(n_sets >= 1, std::string(fn) + ": n_sets must be positive");
    SVMC_REQUIRE(call_id < (1u << 24), std::string(fn) + ": call_id must fit 24 bits");
    const size_t K = c.offsets[c.m], row = 6 + static_cast<size_t>(c.m);          // doubles per parameter set
    for (int i = 0; i < c.m; ++i)
        SVMC_REQUIRE(dts_host[i] > 0.0 && nb_steps_host[i] > 0, std::string(fn) + ": dt and nb_steps must be positive");
    if (n_sets > MAX_FUSED_SETS) {              // more sets than a launch takes: in launches of MAX_FUSED_SETS
        for (int q = 0; q < n_sets; q += MAX_FUSED_SETS) {
            const int P = (n_sets - q < MAX_FUSED_SETS) ? n_sets - q : MAX_FUSED_SETS;
            if (int rc = svmc_logsv_chain_price_frozen_sets(session, ttms_host, forwards_host, discfactors_host, n_expiries,
                                                            strikes_host, types_host, strike_offsets_host, P, params_host + row * q,
                                                            is_spot_measure, variable_type, nb_steps_host, dts_host, seed, call_id,
                                                            prices_host + K * q, stderrs_host + K * q,
                                                            ivols_host ? ivols_host + K * q : nullptr))
                return rc;
        }
        return SVMC_OK;
    }
    const int P = n_sets;
    SVMC_REQUIRE(c.m <= MAX_FUSED_SLICES, std::string(fn) + ": at most 16 expiries");
    SVMC_REQUIRE(c.m * P <= s->max_expiries && K * static_cast<size_t>(P) <= s->max_strikes &&
                     static_cast<size_t>(wave_rows(s->n_path)) * 2 * static_cast<size_t>(c.m) * P * sizeof(double) <= s->ws_bytes,
                 std::string(fn) + ": the session must be created for n_sets chains (max_expiries >= n_sets x n_expiries, "
                                   "max_strikes_total >= n_sets x sum K_i)");
    const size_t n = s->n_path;
    const bool graph = s->use_graphs && !s->sharded();
    const int want_iv = (ivols_host != nullptr && variable_type == SVMC_LOG_RETURN) ? 1 : 0;
    // everything that shapes the launches; the model constants travel in the parameter block
    std::vector<unsigned char> key;
    const int sharded = s->sharded() ? 1 : 0;
    key_append(key, &c.m, 1);
    key_append(key, &P, 1);
    key_append(key, &variable_type, 1);
    key_append(key, &want_iv, 1);
    key_append(key, &sharded, 1);
    key_append(key, &seed, 1);
    key_append(key, &call_id, 1);
    key_append(key, &s->path_offset, 1);
    key_append(key, c.ttms, c.m);
    key_append(key, c.discfactors, want_iv ? c.m : 0);
    key_append(key, c.forwards, c.m);
    key_append(key, c.offsets, c.m + 1);
    key_append(key, c.strikes, K);
    key_append(key, c.types, K);
    key_append(key, nb_steps_host, c.m);
    FixedGraph &g = s->frozen[P];
    // parameter block: [P] initial volatilities, then [m][P] LogsvFast (log units)
    const size_t n_params = static_cast<size_t>(P) + static_cast<size_t>(c.m) * P * LOGSV_FAST_CONSTS_DOUBLES, n_sums = 3 * K * P;
    const size_t n_quotes = K * P;
    const double n_all = static_cast<double>(s->sharded() ? s->n_total : s->n_path);
    std::vector<double> shifts(K);
    for (int i = 0; i < c.m; ++i)
        for (size_t k = c.offsets[i]; k < c.offsets[i + 1]; ++k)
            shifts[k] = payoff_shift(c.strikes[k], c.types[k], c.forwards[i], variable_type);
    // the chain's launches, queued on the session's stream: captured into the graph, or issued as they are (a communicator
    // attached, graphs off) with the two all-reduces between them
    auto enqueue = [&]() -> int {
        SVMC_HIP_TRY(hipMemcpyAsync(g.params_dev, g.params_host, n_params * sizeof(double), hipMemcpyHostToDevice, s->stream));
        double *qsnaps = (variable_type == SVMC_Q_VAR) ? s->snap + static_cast<size_t>(c.m) * P * n : nullptr;
        // (a captured launch never carries the thread's clock probe: the graph outlives the probe's buffer)
        if (int rc = logsv_chain_rng_sets(n, P, c.m, nb_steps_host, g.params_dev + P, g.params_dev, c.forwards, seed, call_id,
                                          s->path_offset, s->snap, qsnaps, s->spot, s->ws, s->ws_bytes, s->stream, !graph))
            return rc;
        if (int rc = all_reduce(s, s->spot, 2 * static_cast<size_t>(c.m) * P)) return rc;
        if (int rc = enqueue_payoff_sums_of_sets(s, c, variable_type, shifts, P)) return rc;
        if (int rc = all_reduce(s, s->sums, n_sums)) return rc;
        if (g.ivols_dev != nullptr) {               // (the last kernel writes the pinned host buffers: no copy nodes)
            if (int rc = chain_implied_vols(s->sums, g.quotes_dev, n_quotes, n_all, IV_VOL_LO, IV_VOL_HI, g.ivols_host, g.sums_host,
                                            s->stream))
                return rc;
        } else if (n_sums) {
            SVMC_HIP_TRY(hipMemcpyAsync(g.sums_host, s->sums, n_sums * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        }
        return SVMC_OK;
    };
    if (g.params_dev == nullptr || g.key != key) {
        fixed_graph_release(g);
        g.params_doubles = n_params;
        g.sums_doubles = n_sums;
        SVMC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g.params_host), n_params * sizeof(double), hipHostMallocDefault));
        SVMC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g.sums_host), (n_sums ? n_sums : 1) * sizeof(double), hipHostMallocDefault));
        SVMC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.params_dev), n_params * sizeof(double)));
        if (want_iv && n_quotes) {
            std::vector<double> quotes(IV_QUOTE_DOUBLES_HOST * n_quotes);
            for (int q = 0; q < P; ++q)
                for (int i = 0; i < c.m; ++i)
                    for (size_t k = c.offsets[i]; k < c.offsets[i + 1]; ++k) {
                        double *qd = quotes.data() + IV_QUOTE_DOUBLES_HOST * (K * q + k);
                        qd[0] = c.strikes[k];
                        qd[1] = static_cast<double>(c.types[k]);
                        qd[2] = shifts[k];
                        qd[3] = c.forwards[i];
                        qd[4] = c.ttms[i];
                        qd[5] = c.discfactors[i];
                    }
            SVMC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.quotes_dev), quotes.size() * sizeof(double)));
            SVMC_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g.ivols_dev), n_quotes * sizeof(double)));
            SVMC_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g.ivols_host), n_quotes * sizeof(double), hipHostMallocDefault));
            SVMC_HIP_TRY(hipMemcpy(g.quotes_dev, quotes.data(), quotes.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        g.key = key;
    }
    if (graph && g.exec == nullptr) {      // captured at the first replayed call of this shape (an un-replayed call may come first)
        SVMC_HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeRelaxed));
        const int rc = enqueue();
        const hipError_t e_end = hipStreamEndCapture(s->stream, &g.graph);      // always leave capture mode
        if (rc != SVMC_OK) { fixed_graph_release(g); return rc; }
        if (e_end != hipSuccess) {
            fixed_graph_release(g);
            return fail(SVMC_ERR_HIP, std::string(fn) + ": graph capture: " + hipGetErrorString(e_end));
        }
        SVMC_HIP_TRY(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    }
    for (int q = 0; q < P; ++q) {
        const double *pr = params_host + row * q;
        g.params_host[q] = pr[0];
        for (int i = 0; i < c.m; ++i)
            logsv_fast_to_doubles(dts_host[i], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6 + i], is_spot_measure,
                                  g.params_host + P + (static_cast<size_t>(i) * P + q) * LOGSV_FAST_CONSTS_DOUBLES);
    }
    if (graph) {
        SVMC_HIP_TRY(hipGraphLaunch(g.exec, s->stream));
        ++s->graph_launches;
    } else if (int rc = enqueue()) {
        return rc;
    }
    SVMC_HIP_TRY(hipStreamSynchronize(s->stream));
    for (int q = 0; q < P; ++q)
        for (int i = 0; i < c.m; ++i) {
            const size_t k0 = c.offsets[i], k = c.offsets[i + 1] - k0;
            if (int rc = svmc_payoff_finalize(g.sums_host + 3 * (K * q + k0), shifts.data() + k0, k, c.discfactors[i], n_all,
                                              prices_host + K * q + k0, stderrs_host + K * q + k0))
                return rc;
        }
    if (ivols_host != nullptr) {
        if (g.ivols_host != nullptr) {
            memcpy(ivols_host, g.ivols_host, K * P * sizeof(double));
        } else {
            for (int q = 0; q < P; ++q) implied_vols_on_host(c, variable_type, prices_host + K * q, ivols_host + K * q);
        }
    }
    return SVMC_OK;
}

int svmc_session_use_graphs(svmc_session_t session, int enable)
{
    Session *s = reinterpret_cast<Session *>(session);
    SVMC_REQUIRE(s != nullptr, "svmc_session_use_graphs: null session");
    s->use_graphs = enable != 0;
    return SVMC_OK;
}

int svmc_session_graph_launches(svmc_session_t session, size_t *count)
{
    Session *s = reinterpret_cast<Session *>(session);
    SVMC_REQUIRE(s != nullptr && count != nullptr, "svmc_session_graph_launches: null pointer");
    *count = s->graph_launches;
    return SVMC_OK;
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
    // the start state (0, v0, 0) of every path (:303-305) travels as three constants: no fill launch
    double t0 = 0.0;
    std::vector<int> nbs(c.m);
    std::vector<double> dts(c.m);
    for (int i = 0; i < c.m; ++i) {                                                                       // :308-329
        time_grid(c.ttms[i] - t0, nb_steps_per_year, nbs[i], dts[i]);
        t0 = c.ttms[i];
    }
    double *qsnap = (variable_type == SVMC_Q_VAR) ? s->snap + static_cast<size_t>(c.m) * n : nullptr;
    stepping_begin(s);
    if (c.m > MAX_FUSED_SLICES) {
        if (int rc = svmc_heston_chain_rng_from(0.0, v0, 0.0, s->x, s->vol, s->qvar, n, c.m, nbs.data(), dts.data(), c.forwards, theta, kappa,
                                                rho, volvol, scheme, seed, call_id, s->path_offset, 0, s->snap, qsnap, s->spot, s->ws,
                                                s->ws_bytes, s->stream))
            return rc;
        stepping_end(s);
        return reduce_and_finalize(s, c, variable_type, prices_host, stderrs_host, false);
    }
    if (int rc = heston_step_partials(v0, s->x, s->vol, s->qvar, n, c.m, nbs.data(), dts.data(), c.forwards, theta, kappa, rho, volvol, scheme,
                                      seed, call_id, s->path_offset, s->snap, qsnap, s->spot_ws, s->spot_ws_bytes, s->stream))
        return rc;
    stepping_end(s);
    return reduce_and_finalize(s, c, variable_type, prices_host, stderrs_host, true);
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
