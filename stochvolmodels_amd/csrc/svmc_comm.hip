// svmc_comm.hip -- RCCL below Python: the communicator entry points of the C ABI (include/svmc.h).
//
// The only cross-path couplings of the Monte Carlo path are the two reductions of compute_mc_vars_payoff
// (utils/mc_payoffs.py:61-63 and :85-86).  With paths sharded over the GPUs of a node they become two small fp64 sum
// all-reduces per chain over xGMI; the fused chain drivers of svmc_chain.hip issue them on the session's
// stream when the session carries a communicator (svmc_session_set_comm), so a C / C++ host scales over GPUs without a
// Python layer.  The Python host can use the same entry points (stochvolmodels_amd.dist.RcclComm) instead of
// torch.distributed.
//
// RCCL is resolved at RUN time (dlsym / dlopen), not at link time: libsvmc.so stays loadable on a box without RCCL, and
// inside a process that already carries an RCCL (PyTorch-ROCm bundles one) the SAME copy is used -- a communicator must
// be driven by the library that created it.  Only the public types of <rccl/rccl.h> are used here.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// RCCL is resolved at run time: on a build box without its headers, the handful of public types this file uses
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
}
#endif

#include "svmc_internal.h"

namespace svmc {

struct RcclApi {
    ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*error_string)(ncclResult_t) = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*comm_user_rank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*comm_init_all)(ncclComm_t *, int, const int *) = nullptr;      // optional: single-process multi-device
    std::string origin, error;
    bool ok = false;
};

static RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        // 1. an RCCL already mapped into the process (the one PyTorch-ROCm carries, or one the host linked)
        if (dlsym(RTLD_DEFAULT, "ncclAllReduce") != nullptr) {
            h = RTLD_DEFAULT;
            api.origin = "process (RTLD_DEFAULT)";
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (h != nullptr) {
                    api.origin = name;
                    break;
                }
            }
            if (h == nullptr) {
                api.error = std::string("RCCL not found (dlopen librccl.so.1): ") + dlerror();
                return;
            }
        }
        auto sym = [&](const char *n) { return dlsym(h, n); };
        api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(sym("ncclGetUniqueId"));
        api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(sym("ncclCommInitRank"));
        api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(sym("ncclCommDestroy"));
        api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(sym("ncclAllReduce"));
        api.error_string = reinterpret_cast<decltype(api.error_string)>(sym("ncclGetErrorString"));
        api.comm_count = reinterpret_cast<decltype(api.comm_count)>(sym("ncclCommCount"));
        api.comm_user_rank = reinterpret_cast<decltype(api.comm_user_rank)>(sym("ncclCommUserRank"));
        api.comm_init_all = reinterpret_cast<decltype(api.comm_init_all)>(sym("ncclCommInitAll"));
        api.ok = api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_reduce && api.error_string;
        if (!api.ok) api.error = "RCCL (" + api.origin + ") lacks a required symbol";
    });
    return api;
}

static int rccl_fail(const char *what, ncclResult_t r)
{
    return fail(SVMC_ERR_RCCL, std::string(what) + ": " + (rccl().error_string ? rccl().error_string(r) : "RCCL error"));
}

// one communicator per device of ONE process (ncclCommInitAll): the transport of a multi-session (svmc_multi.hip) whose
// shards sit on distinct devices; each communicator is then driven by its shard's own host thread
int rccl_comm_init_all(int n, const int *devices, void **comms_out)
{
    if (!rccl().ok) return fail(SVMC_ERR_RCCL, rccl().error);
    if (rccl().comm_init_all == nullptr) return fail(SVMC_ERR_RCCL, "RCCL (" + rccl().origin + ") lacks ncclCommInitAll");
    static_assert(sizeof(ncclComm_t) == sizeof(void *), "ncclComm_t is a pointer");
    const ncclResult_t r = rccl().comm_init_all(reinterpret_cast<ncclComm_t *>(comms_out), n, devices);
    if (r != ncclSuccess) return rccl_fail("ncclCommInitAll", r);
    return SVMC_OK;
}

}  // namespace svmc

using namespace svmc;

extern "C" {

int svmc_rccl_available(void)
{
    return rccl().ok ? 1 : 0;
}

int svmc_rccl_unique_id(void *id_out, size_t bytes)
{
    SVMC_REQUIRE(id_out != nullptr && bytes >= sizeof(ncclUniqueId), "svmc_rccl_unique_id: need SVMC_RCCL_UNIQUE_ID_BYTES of output");
    if (!rccl().ok) return fail(SVMC_ERR_RCCL, rccl().error);
    ncclUniqueId id;
    const ncclResult_t r = rccl().get_unique_id(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    static_assert(sizeof(ncclUniqueId) == SVMC_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return SVMC_OK;
}

int svmc_rccl_comm_create(svmc_comm_t *comm, const void *id_bytes, size_t bytes, int world, int rank)
{
    SVMC_REQUIRE(comm != nullptr && id_bytes != nullptr && bytes >= sizeof(ncclUniqueId), "svmc_rccl_comm_create: null / short id");
    SVMC_REQUIRE(world >= 1 && rank >= 0 && rank < world, "svmc_rccl_comm_create: need 0 <= rank < world");
    if (!rccl().ok) return fail(SVMC_ERR_RCCL, rccl().error);
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t r = rccl().comm_init_rank(&c, world, id, rank);       // binds the CURRENT HIP device
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    *comm = reinterpret_cast<svmc_comm_t>(c);
    return SVMC_OK;
}

int svmc_rccl_comm_destroy(svmc_comm_t comm)
{
    if (comm == nullptr) return SVMC_OK;
    if (!rccl().ok) return fail(SVMC_ERR_RCCL, rccl().error);
    const ncclResult_t r = rccl().comm_destroy(reinterpret_cast<ncclComm_t>(comm));
    if (r != ncclSuccess) return rccl_fail("ncclCommDestroy", r);
    return SVMC_OK;
}

int svmc_rccl_all_reduce_sum(svmc_comm_t comm, double *buf, size_t n, svmc_stream_t stream)
{
    SVMC_REQUIRE(comm != nullptr && (buf != nullptr || n == 0), "svmc_rccl_all_reduce_sum: null communicator / buffer");
    if (n == 0) return SVMC_OK;
    if (!rccl().ok) return fail(SVMC_ERR_RCCL, rccl().error);
    const ncclResult_t r = rccl().all_reduce(buf, buf, n, ncclDouble, ncclSum, reinterpret_cast<ncclComm_t>(comm),
                                             as_stream(stream));
    if (r != ncclSuccess) return rccl_fail("ncclAllReduce", r);
    return SVMC_OK;
}

int svmc_rccl_comm_count(svmc_comm_t comm, int *world_out, int *rank_out)
{
    SVMC_REQUIRE(comm != nullptr && world_out != nullptr, "svmc_rccl_comm_count: null communicator / output");
    if (!rccl().ok) return fail(SVMC_ERR_RCCL, rccl().error);
    if (rccl().comm_count == nullptr) return fail(SVMC_ERR_RCCL, "RCCL (" + rccl().origin + ") lacks ncclCommCount");
    ncclResult_t r = rccl().comm_count(reinterpret_cast<ncclComm_t>(comm), world_out);
    if (r != ncclSuccess) return rccl_fail("ncclCommCount", r);
    if (rank_out != nullptr && rccl().comm_user_rank != nullptr) {
        r = rccl().comm_user_rank(reinterpret_cast<ncclComm_t>(comm), rank_out);
        if (r != ncclSuccess) return rccl_fail("ncclCommUserRank", r);
    }
    return SVMC_OK;
}

const char *svmc_rccl_origin(void)
{
    return rccl().ok ? rccl().origin.c_str() : rccl().error.c_str();
}

}  // extern "C"
