// svmc_multi.hip -- ONE process, SEVERAL devices: the single-process multi-GPU driver of the C ABI (include/svmc.h,
// "single-process multi-device sessions").
//
// The reference is a single-process Python library (logsv_mc_chain_pricer, pricers/logsv_pricer.py:806-867, runs one
// NumPy loop); the drop-in user who owns a node of eight MI355X should not need a launcher, a rendezvous and eight
// interpreters to use them.  A multi-session is R shards of one job -- shard r holds the global path ids
// [N r / R, N (r + 1) / R) in its own svmc_session_t on its own device, driven by its own host thread -- and one call
// prices the chain on all of them: the shards step concurrently, meet twice for the two sum all-reduces of
// compute_mc_vars_payoff (utils/mc_payoffs.py:61-63, :85-86; a few KB each) and every shard finalises the same prices.
// The all-reduce is RCCL (ncclCommInitAll: one communicator per device, rings over xGMI) when RCCL resolves and the
// devices are distinct, else a sum through page-locked host memory in rank order -- the same bits on every shard, no
// dependency beyond the HIP runtime.  The counter-based randoms are indexed by the GLOBAL path id (svmc_rng.h), so the
// job's result does not depend on R; against one session of N paths it differs by the order of the final additions only.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "svmc_internal.h"

namespace svmc {

// A barrier for the R shard threads that can be ABORTED: a shard that fails between two meeting points must not leave the
// others waiting for it.  Arrival takes the mutex; the wait spins on the generation counter first (the meeting points of a
// chain are microseconds apart) and sleeps on the condition variable after that.
class AbortableBarrier {
public:
    explicit AbortableBarrier(int n) : n_(n) {}
    bool arrive_and_wait()
    {
        uint64_t g;
        {
            std::lock_guard<std::mutex> l(m_);
            if (abort_.load(std::memory_order_relaxed)) return false;
            g = gen_.load(std::memory_order_relaxed);
            if (++count_ == n_) {
                count_ = 0;
                gen_.store(g + 1, std::memory_order_release);
                cv_.notify_all();
                return true;
            }
        }
        for (int i = 0; i < 20000; ++i) {
            if (gen_.load(std::memory_order_acquire) != g) return true;
            if (abort_.load(std::memory_order_acquire)) return false;
            std::this_thread::yield();
        }
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return gen_.load(std::memory_order_acquire) != g || abort_.load(std::memory_order_acquire); });
        return gen_.load(std::memory_order_acquire) != g;
    }
    void abort()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            abort_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
    }
    void reset()      // only while no thread is inside arrive_and_wait (between jobs)
    {
        std::lock_guard<std::mutex> l(m_);
        count_ = 0;
        abort_.store(false, std::memory_order_release);
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    const int n_;
    int count_ = 0;
    std::atomic<uint64_t> gen_{0};
    std::atomic<bool> abort_{false};
};

struct Multi;

struct Shard {
    Multi *owner = nullptr;
    int rank = 0, device = 0;
    uint64_t offset = 0;
    size_t n_path = 0;
    svmc_session_t session = nullptr;
    void *comm = nullptr;                       // RCCL mode: this device's ncclComm_t
    std::thread thread;
    std::vector<double> prices, stderrs;        // every shard finalises the job's prices; shard 0's go to the caller
    int rc = SVMC_OK;
    std::string error;
    double last_ms = 0.0;
    unsigned reduce_calls = 0;                  // host mode: parity of the double-buffered exchange slots
};

struct Multi {
    int R = 0;
    uint64_t n_total = 0;
    int max_expiries = 0;
    size_t max_strikes = 0;
    int mode = SVMC_MULTI_REDUCE_HOST;          // the mode in use (AUTO is resolved at creation)
    int rccl_ranks_seen = 0;
    bool shards_agree = true;
    std::vector<Shard> shards;
    AbortableBarrier *barrier = nullptr;
    // host all-reduce: slots[parity][rank][cap] written by their rank, summed in rank order by everybody; result[rank][cap]
    double *slots = nullptr, *result = nullptr;
    size_t cap = 0;
    // test hook (SVMC_MULTI_FAULT="shard,k" at creation): that shard's k-th host all-reduce fails once -- a failure BETWEEN the
    // two meeting points of a chain, which no argument check can produce (tests/test_gpu_multi.py)
    int fault_shard = -1;
    long fault_countdown = -1;
    // job dispatch
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::function<int(Shard &)> job;
    uint64_t job_id = 0;
    int pending = 0;
    bool quit = false;
};

// svmc_all_reduce_fn of the host mode: device buffer -> this rank's pinned slot, meet, sum the R slots in rank order (every
// shard performs the same additions in the same order: identical bits everywhere), result -> device buffer on the stream.
// The slots are double-buffered by call parity: a shard may write its slot of the NEXT exchange while a slower one still
// reads this exchange's, and it can only come back to this parity after the next meeting point, which the slow one reaches
// after it finished reading.  One meeting point per all-reduce.
static int host_all_reduce(void *user, double *device_buf, size_t n, svmc_stream_t stream)
{
    Shard &s = *static_cast<Shard *>(user);
    Multi &mu = *s.owner;
    if (n > mu.cap) return fail(SVMC_ERR_WORKSPACE, "multi-session host all-reduce: buffer exceeds the exchange slots");
    if (s.rank == mu.fault_shard && mu.fault_countdown >= 0 && mu.fault_countdown-- == 0)
        return fail(SVMC_ERR_HIP, "fault injection (SVMC_MULTI_FAULT): this shard's all-reduce fails");
    const size_t parity = s.reduce_calls++ & 1u;
    double *mine = mu.slots + (parity * mu.R + s.rank) * mu.cap;
    SVMC_HIP_TRY(hipMemcpyAsync(mine, device_buf, n * sizeof(double), hipMemcpyDeviceToHost, as_stream(stream)));
    SVMC_HIP_TRY(hipStreamSynchronize(as_stream(stream)));      // also completes the previous exchange's upload of `result`
    if (!mu.barrier->arrive_and_wait()) return fail(SVMC_ERR_HIP, "multi-session all-reduce aborted: another shard failed");
    double *res = mu.result + static_cast<size_t>(s.rank) * mu.cap;
    const double *base = mu.slots + parity * mu.R * mu.cap;
    for (size_t i = 0; i < n; ++i) {
        double t = base[i];
        for (int r = 1; r < mu.R; ++r) t += base[static_cast<size_t>(r) * mu.cap + i];
        res[i] = t;
    }
    SVMC_HIP_TRY(hipMemcpyAsync(device_buf, res, n * sizeof(double), hipMemcpyHostToDevice, as_stream(stream)));
    return SVMC_OK;
}

static void worker_main(Shard *sp)
{
    Shard &s = *sp;
    Multi &mu = *s.owner;
    uint64_t seen = 0;
    for (;;) {
        std::function<int(Shard &)> job;
        {
            std::unique_lock<std::mutex> l(mu.m);
            mu.cv_job.wait(l, [&] { return mu.quit || mu.job_id != seen; });
            if (mu.quit) return;
            seen = mu.job_id;
            job = mu.job;
        }
        const auto t0 = std::chrono::steady_clock::now();
        int rc = (hipSetDevice(s.device) == hipSuccess) ? SVMC_OK : fail(SVMC_ERR_HIP, "hipSetDevice failed");
        if (rc == SVMC_OK) rc = job(s);
        s.last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        s.rc = rc;
        if (rc != SVMC_OK) {
            s.error = last_error_ref();
            mu.barrier->abort();               // nobody waits for a shard that is not coming
        }
        {
            std::lock_guard<std::mutex> l(mu.m);
            if (--mu.pending == 0) mu.cv_done.notify_all();
        }
    }
}

// run `job` on every shard's thread and wait; the first failing shard's status and message become the caller's
static int run_on_shards(Multi &mu, std::function<int(Shard &)> job)
{
    mu.barrier->reset();
    {
        std::lock_guard<std::mutex> l(mu.m);
        for (Shard &s : mu.shards) {
            s.rc = SVMC_OK;
            s.error.clear();
            // every job starts its exchanges at slot parity 0 on EVERY shard: a job that failed half-way may have left the
            // shards' counters apart, and shards on different parities would read each other's stale slots -- silently
            s.reduce_calls = 0;
        }
        mu.job = std::move(job);
        mu.pending = mu.R;
        ++mu.job_id;
    }
    mu.cv_job.notify_all();
    {
        std::unique_lock<std::mutex> l(mu.m);
        mu.cv_done.wait(l, [&] { return mu.pending == 0; });
    }
    // report the shard that FAILED, not one that was merely released from a meeting point by the failure
    const Shard *bad = nullptr;
    for (const Shard &s : mu.shards) {
        if (s.rc == SVMC_OK) continue;
        const bool released = s.error.find("all-reduce aborted") != std::string::npos;
        if (bad == nullptr || (!released && bad->error.find("all-reduce aborted") != std::string::npos)) bad = &s;
    }
    if (bad != nullptr)
        return fail(bad->rc, "shard " + std::to_string(bad->rank) + " (device " + std::to_string(bad->device) + "): " + bad->error);
    return SVMC_OK;
}

static void multi_release(Multi *mu)
{
    if (mu == nullptr) return;
    if (!mu->shards.empty() && mu->shards[0].thread.joinable()) {
        (void)run_on_shards(*mu, [](Shard &s) {
            if (s.session != nullptr) (void)svmc_session_destroy(s.session);
            s.session = nullptr;
            if (s.comm != nullptr) (void)svmc_rccl_comm_destroy(s.comm);
            s.comm = nullptr;
            return SVMC_OK;
        });
        {
            std::lock_guard<std::mutex> l(mu->m);
            mu->quit = true;
        }
        mu->cv_job.notify_all();
        for (Shard &s : mu->shards)
            if (s.thread.joinable()) s.thread.join();
    }
    if (mu->slots != nullptr) (void)hipHostFree(mu->slots);
    if (mu->result != nullptr) (void)hipHostFree(mu->result);
    delete mu->barrier;
    delete mu;
}

// after a pricing call: every shard holds the job's prices -- they must be the same bits (they all finalise the same sums)
static void compare_shards(Multi &mu, size_t n)
{
    mu.shards_agree = true;
    for (int r = 1; r < mu.R; ++r)
        if (memcmp(mu.shards[r].prices.data(), mu.shards[0].prices.data(), n * sizeof(double)) != 0 ||
            memcmp(mu.shards[r].stderrs.data(), mu.shards[0].stderrs.data(), n * sizeof(double)) != 0)
            mu.shards_agree = false;
}

}  // namespace svmc

using namespace svmc;

extern "C" {

int svmc_multi_create(svmc_multi_t *multi, int n_shards, const int *devices_host, uint64_t n_path_total, int max_expiries,
                      size_t max_strikes_total, int reduce_mode)
{
    SVMC_REQUIRE(multi != nullptr, "svmc_multi_create: null output");
    SVMC_REQUIRE(n_shards >= 1 && n_shards <= 64, "svmc_multi_create: 1 <= n_shards <= 64");
    SVMC_REQUIRE(n_path_total >= static_cast<uint64_t>(n_shards) && n_path_total < (1ull << 56),
                 "svmc_multi_create: need n_shards <= n_path_total < 2^56");
    SVMC_REQUIRE(max_expiries > 0 && max_strikes_total > 0, "svmc_multi_create: sizes must be positive");
    SVMC_REQUIRE(reduce_mode == SVMC_MULTI_REDUCE_AUTO || reduce_mode == SVMC_MULTI_REDUCE_HOST || reduce_mode == SVMC_MULTI_REDUCE_RCCL,
                 "svmc_multi_create: unknown reduce mode");
    int n_dev = 0;
    SVMC_HIP_TRY(hipGetDeviceCount(&n_dev));
    SVMC_REQUIRE(n_dev >= 1, "svmc_multi_create: no HIP device visible");
    int caller_device = 0;
    SVMC_HIP_TRY(hipGetDevice(&caller_device));
    Multi *mu = new Multi;
    mu->R = n_shards;
    mu->n_total = n_path_total;
    mu->max_expiries = max_expiries;
    mu->max_strikes = max_strikes_total;
    mu->barrier = new AbortableBarrier(n_shards);
    mu->shards.resize(n_shards);
    if (const char *f = std::getenv("SVMC_MULTI_FAULT")) {
        int sh = -1;
        long k = -1;
        if (std::sscanf(f, "%d,%ld", &sh, &k) == 2) {
            mu->fault_shard = sh;
            mu->fault_countdown = k;
        }
    }
    bool distinct = true;
    for (int r = 0; r < n_shards; ++r) {
        Shard &s = mu->shards[r];
        s.owner = mu;
        s.rank = r;
        s.device = devices_host ? devices_host[r] : r % n_dev;
        if (s.device < 0 || s.device >= n_dev) {
            multi_release(mu);
            return fail(SVMC_ERR_INVALID_ARGUMENT, "svmc_multi_create: device index out of range");
        }
        for (int q = 0; q < r; ++q) distinct = distinct && mu->shards[q].device != s.device;
        // the same split as the Python host's dist.shard_range: [N r / R, N (r + 1) / R)
        const uint64_t R = static_cast<uint64_t>(n_shards);
        const uint64_t lo = n_path_total * static_cast<uint64_t>(r) / R, hi = n_path_total * static_cast<uint64_t>(r + 1) / R;
        s.offset = lo;
        s.n_path = static_cast<size_t>(hi - lo);
        s.prices.assign(max_strikes_total, 0.0);
        s.stderrs.assign(max_strikes_total, 0.0);
    }
    // the transport of the two all-reduces
    std::vector<void *> comms(n_shards, nullptr);
    std::string rccl_note;
    mu->mode = SVMC_MULTI_REDUCE_HOST;
    if (reduce_mode != SVMC_MULTI_REDUCE_HOST) {
        if (!distinct) {
            rccl_note = "two shards share a device (RCCL takes one rank per device)";
        } else if (!svmc_rccl_available()) {
            rccl_note = svmc_rccl_origin();
        } else {
            std::vector<int> devs(n_shards);
            for (int r = 0; r < n_shards; ++r) devs[r] = mu->shards[r].device;
            if (rccl_comm_init_all(n_shards, devs.data(), comms.data()) == SVMC_OK) mu->mode = SVMC_MULTI_REDUCE_RCCL;
            else rccl_note = last_error_ref();
        }
        if (mu->mode != SVMC_MULTI_REDUCE_RCCL && reduce_mode == SVMC_MULTI_REDUCE_RCCL) {
            multi_release(mu);
            (void)hipSetDevice(caller_device);
            return fail(SVMC_ERR_RCCL, "svmc_multi_create: RCCL requested but unavailable: " + rccl_note);
        }
        (void)hipSetDevice(caller_device);      // ncclCommInitAll walks the devices
    }
    if (mu->mode == SVMC_MULTI_REDUCE_HOST) {
        const size_t a = 2 * static_cast<size_t>(max_expiries), b = 3 * max_strikes_total;
        mu->cap = (a > b ? a : b) + 8;
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&mu->slots), 2 * mu->cap * n_shards * sizeof(double), hipHostMallocPortable);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&mu->result), mu->cap * n_shards * sizeof(double), hipHostMallocPortable);
        if (e != hipSuccess) {
            multi_release(mu);
            return fail(SVMC_ERR_HIP, std::string("svmc_multi_create: pinned exchange buffers: ") + hipGetErrorString(e));
        }
    }
    for (int r = 0; r < n_shards; ++r) {
        mu->shards[r].comm = comms[r];
        mu->shards[r].thread = std::thread(worker_main, &mu->shards[r]);
    }
    // every shard builds its session on its own device from its own thread (a session's stream and buffers belong to the
    // device that was current when it was created), attaches the transport and, with RCCL, runs one all-reduce so that the
    // rings are built here and not inside the first chain priced
    const int rc = run_on_shards(*mu, [mu](Shard &s) {
        if (int rc = svmc_session_create(&s.session, s.n_path, mu->max_expiries, mu->max_strikes)) return rc;
        if (mu->mode == SVMC_MULTI_REDUCE_RCCL) {
            // everything that can fail on ONE shard alone (the session above: an out-of-memory device; the communicator's
            // attachment; the warm-up buffer) happens BEFORE the shards meet: a shard that fails here returns, worker_main aborts
            // the barrier, and the others leave with an error instead of entering a collective whose peer never comes (inside
            // ncclAllReduce a missing peer cannot be recovered from; a failed CREATION can and must -- round-5 advisor finding)
            double *warm = nullptr;
            int rc = svmc_session_set_comm(s.session, s.comm, s.rank, mu->R, mu->n_total, s.offset);
            if (rc == SVMC_OK && hipMalloc(reinterpret_cast<void **>(&warm), sizeof(double)) != hipSuccess)
                rc = fail(SVMC_ERR_HIP, "svmc_multi_create: warm-up buffer");
            const double one = 1.0;
            if (rc == SVMC_OK && hipMemcpy(warm, &one, sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(SVMC_ERR_HIP, "svmc_multi_create: warm-up upload");
            if (rc != SVMC_OK) {
                if (warm != nullptr) (void)hipFree(warm);
                return rc;
            }
            if (!mu->barrier->arrive_and_wait()) {
                (void)hipFree(warm);
                return fail(SVMC_ERR_HIP, "multi-session all-reduce aborted: another shard failed while the multi-session was created");
            }
            rc = svmc_rccl_all_reduce_sum(s.comm, warm, 1, nullptr);
            if (rc == SVMC_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(SVMC_ERR_HIP, "warm-up all-reduce failed");
            double seen = 0.0;
            if (rc == SVMC_OK && hipMemcpy(&seen, warm, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
                rc = fail(SVMC_ERR_HIP, "svmc_multi_create: warm-up download");
            (void)hipFree(warm);
            if (rc == SVMC_OK && s.rank == 0) mu->rccl_ranks_seen = static_cast<int>(seen + 0.5);
            return rc;
        }
        return svmc_session_set_reducer(s.session, host_all_reduce, &s, s.rank, mu->R, mu->n_total, s.offset);
    });
    if (rc != SVMC_OK) {
        const std::string msg = last_error_ref();
        multi_release(mu);
        (void)hipSetDevice(caller_device);
        return fail(rc, "svmc_multi_create: " + msg);
    }
    *multi = reinterpret_cast<svmc_multi_t>(mu);
    return SVMC_OK;
}

int svmc_multi_destroy(svmc_multi_t multi)
{
    multi_release(reinterpret_cast<Multi *>(multi));
    return SVMC_OK;
}

int svmc_multi_info(svmc_multi_t multi, int *n_shards, int *reduce_mode, int *rccl_ranks_seen, int *shards_agree)
{
    Multi *mu = reinterpret_cast<Multi *>(multi);
    SVMC_REQUIRE(mu != nullptr, "svmc_multi_info: null multi-session");
    if (n_shards != nullptr) *n_shards = mu->R;
    if (reduce_mode != nullptr) *reduce_mode = mu->mode;
    if (rccl_ranks_seen != nullptr) *rccl_ranks_seen = mu->rccl_ranks_seen;
    if (shards_agree != nullptr) *shards_agree = mu->shards_agree ? 1 : 0;
    return SVMC_OK;
}

int svmc_multi_shard_info(svmc_multi_t multi, int shard, int *device, uint64_t *path_offset, uint64_t *n_path, double *last_call_ms)
{
    Multi *mu = reinterpret_cast<Multi *>(multi);
    SVMC_REQUIRE(mu != nullptr && shard >= 0 && shard < mu->R, "svmc_multi_shard_info: null multi-session / no such shard");
    const Shard &s = mu->shards[shard];
    if (device != nullptr) *device = s.device;
    if (path_offset != nullptr) *path_offset = s.offset;
    if (n_path != nullptr) *n_path = s.n_path;
    if (last_call_ms != nullptr) *last_call_ms = s.last_ms;
    return SVMC_OK;
}

int svmc_multi_logsv_chain_price(svmc_multi_t multi, const double *ttms_host, const double *forwards_host,
                                 const double *discfactors_host, const double *vol_backbone_etas_host, int n_expiries,
                                 const double *strikes_host, const int8_t *types_host, const size_t *strike_offsets_host,
                                 double v0, double theta, double kappa1, double kappa2, double beta, double volvol,
                                 int is_spot_measure, int nb_steps_per_year, int variable_type, uint64_t seed, uint32_t call_id,
                                 double *prices_host, double *stderrs_host)
{
    Multi *mu = reinterpret_cast<Multi *>(multi);
    SVMC_REQUIRE(mu != nullptr, "svmc_multi_logsv_chain_price: null multi-session");
    SVMC_REQUIRE(strike_offsets_host != nullptr && prices_host != nullptr && stderrs_host != nullptr && n_expiries >= 1,
                 "svmc_multi_logsv_chain_price: null pointer");
    SVMC_REQUIRE(n_expiries <= mu->max_expiries && strike_offsets_host[n_expiries] <= mu->max_strikes,
                 "svmc_multi_logsv_chain_price: the chain exceeds the multi-session");
    const size_t K = strike_offsets_host[n_expiries];
    const int rc = run_on_shards(*mu, [=](Shard &s) {
        return svmc_logsv_chain_price(s.session, ttms_host, forwards_host, discfactors_host, vol_backbone_etas_host, n_expiries,
                                      strikes_host, types_host, strike_offsets_host, v0, theta, kappa1, kappa2, beta, volvol,
                                      is_spot_measure, nb_steps_per_year, variable_type, seed, call_id, s.prices.data(),
                                      s.stderrs.data());
    });
    if (rc != SVMC_OK) return rc;
    compare_shards(*mu, K);
    memcpy(prices_host, mu->shards[0].prices.data(), K * sizeof(double));
    memcpy(stderrs_host, mu->shards[0].stderrs.data(), K * sizeof(double));
    return SVMC_OK;
}

int svmc_multi_heston_chain_price(svmc_multi_t multi, const double *ttms_host, const double *forwards_host,
                                  const double *discfactors_host, int n_expiries, const double *strikes_host,
                                  const int8_t *types_host, const size_t *strike_offsets_host, double v0, double theta,
                                  double kappa, double rho, double volvol, int scheme, int nb_steps_per_year, int variable_type,
                                  uint64_t seed, uint32_t call_id, double *prices_host, double *stderrs_host)
{
    Multi *mu = reinterpret_cast<Multi *>(multi);
    SVMC_REQUIRE(mu != nullptr, "svmc_multi_heston_chain_price: null multi-session");
    SVMC_REQUIRE(strike_offsets_host != nullptr && prices_host != nullptr && stderrs_host != nullptr && n_expiries >= 1,
                 "svmc_multi_heston_chain_price: null pointer");
    SVMC_REQUIRE(n_expiries <= mu->max_expiries && strike_offsets_host[n_expiries] <= mu->max_strikes,
                 "svmc_multi_heston_chain_price: the chain exceeds the multi-session");
    const size_t K = strike_offsets_host[n_expiries];
    const int rc = run_on_shards(*mu, [=](Shard &s) {
        return svmc_heston_chain_price(s.session, ttms_host, forwards_host, discfactors_host, n_expiries, strikes_host, types_host,
                                       strike_offsets_host, v0, theta, kappa, rho, volvol, scheme, nb_steps_per_year,
                                       variable_type, seed, call_id, s.prices.data(), s.stderrs.data());
    });
    if (rc != SVMC_OK) return rc;
    compare_shards(*mu, K);
    memcpy(prices_host, mu->shards[0].prices.data(), K * sizeof(double));
    memcpy(stderrs_host, mu->shards[0].stderrs.data(), K * sizeof(double));
    return SVMC_OK;
}

int svmc_multi_state(svmc_multi_t multi, double *x_host, double *vol_host, double *qvar_host)
{
    Multi *mu = reinterpret_cast<Multi *>(multi);
    SVMC_REQUIRE(mu != nullptr, "svmc_multi_state: null multi-session");
    return run_on_shards(*mu, [=](Shard &s) {
        return svmc_session_state(s.session, x_host ? x_host + s.offset : nullptr, vol_host ? vol_host + s.offset : nullptr,
                                  qvar_host ? qvar_host + s.offset : nullptr);
    });
}

}  // extern "C"
