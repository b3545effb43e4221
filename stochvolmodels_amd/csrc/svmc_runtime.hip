// svmc_runtime.hip -- device / memory / stream / event plumbing of the C ABI (include/svmc.h).
#include "svmc_internal.h"
#include "svmc_black.h"

#include <cmath>
#include <limits>

#include <cstring>

namespace svmc {

std::string &last_error_ref()
{
    thread_local std::string err;
    return err;
}

int fail(int code, const std::string &msg)
{
    last_error_ref() = msg;
    return code;
}

}  // namespace svmc

using namespace svmc;

extern "C" {

int svmc_version(void) { return SVMC_VERSION; }
int svmc_rng_stream_version(void) { return SVMC_RNG_STREAM_VERSION; }

const char *svmc_last_error(void) { return last_error_ref().c_str(); }

int svmc_device_count(int *count)
{
    SVMC_REQUIRE(count != nullptr, "svmc_device_count: null output");
    SVMC_HIP_TRY(hipGetDeviceCount(count));
    return SVMC_OK;
}

int svmc_set_device(int device)
{
    SVMC_HIP_TRY(hipSetDevice(device));
    return SVMC_OK;
}

int svmc_get_device(int *device)
{
    SVMC_REQUIRE(device != nullptr, "svmc_get_device: null output");
    SVMC_HIP_TRY(hipGetDevice(device));
    return SVMC_OK;
}

int svmc_device_info(int device, char *name, size_t name_len, int *compute_units, int *clock_khz,
                     size_t *hbm_bytes)
{
    hipDeviceProp_t prop;
    SVMC_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name != nullptr && name_len > 0) {
        std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units != nullptr) *compute_units = prop.multiProcessorCount;
    if (clock_khz != nullptr) *clock_khz = prop.clockRate;
    if (hbm_bytes != nullptr) *hbm_bytes = prop.totalGlobalMem;
    return SVMC_OK;
}

int svmc_malloc(void **dptr, size_t bytes)
{
    SVMC_REQUIRE(dptr != nullptr, "svmc_malloc: null output");
    SVMC_HIP_TRY(hipMalloc(dptr, bytes ? bytes : 8));
    return SVMC_OK;
}

int svmc_free(void *dptr)
{
    if (dptr != nullptr) SVMC_HIP_TRY(hipFree(dptr));
    return SVMC_OK;
}

int svmc_host_alloc(void **hptr, size_t bytes)
{
    SVMC_REQUIRE(hptr != nullptr, "svmc_host_alloc: null output");
    SVMC_HIP_TRY(hipHostMalloc(hptr, bytes ? bytes : 8, hipHostMallocDefault));
    return SVMC_OK;
}

int svmc_host_free(void *hptr)
{
    if (hptr != nullptr) SVMC_HIP_TRY(hipHostFree(hptr));
    return SVMC_OK;
}

int svmc_memset(void *dptr, int value, size_t bytes, svmc_stream_t stream)
{
    SVMC_HIP_TRY(hipMemsetAsync(dptr, value, bytes, as_stream(stream)));
    return SVMC_OK;
}

int svmc_memcpy_h2d(void *dst, const void *src_host, size_t bytes, svmc_stream_t stream)
{
    SVMC_HIP_TRY(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return SVMC_OK;
}

int svmc_memcpy_d2h(void *dst_host, const void *src, size_t bytes, svmc_stream_t stream)
{
    SVMC_HIP_TRY(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return SVMC_OK;
}

int svmc_memcpy_d2d(void *dst, const void *src, size_t bytes, svmc_stream_t stream)
{
    SVMC_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return SVMC_OK;
}

int svmc_memcpy2d_h2d(void *dst, size_t dst_pitch_bytes, const void *src_host, size_t src_pitch_bytes,
                      size_t width_bytes, size_t height, svmc_stream_t stream)
{
    if (width_bytes == 0 || height == 0) return SVMC_OK;
    if (dst_pitch_bytes == width_bytes && src_pitch_bytes == width_bytes) {      // whole rows: one linear copy
        SVMC_HIP_TRY(hipMemcpyAsync(dst, src_host, width_bytes * height, hipMemcpyHostToDevice, as_stream(stream)));
        return SVMC_OK;
    }
    SVMC_HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch_bytes, src_host, src_pitch_bytes, width_bytes, height,
                                  hipMemcpyHostToDevice, as_stream(stream)));
    return SVMC_OK;
}

int svmc_stream_create(svmc_stream_t *stream)
{
    SVMC_REQUIRE(stream != nullptr, "svmc_stream_create: null output");
    hipStream_t s;
    SVMC_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = reinterpret_cast<svmc_stream_t>(s);
    return SVMC_OK;
}

int svmc_stream_destroy(svmc_stream_t stream)
{
    if (stream != nullptr) SVMC_HIP_TRY(hipStreamDestroy(as_stream(stream)));
    return SVMC_OK;
}

int svmc_stream_synchronize(svmc_stream_t stream)
{
    SVMC_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return SVMC_OK;
}

int svmc_event_create(svmc_event_t *event)
{
    SVMC_REQUIRE(event != nullptr, "svmc_event_create: null output");
    hipEvent_t e;
    SVMC_HIP_TRY(hipEventCreate(&e));
    *event = reinterpret_cast<svmc_event_t>(e);
    return SVMC_OK;
}

int svmc_event_destroy(svmc_event_t event)
{
    if (event != nullptr) SVMC_HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return SVMC_OK;
}

int svmc_event_record(svmc_event_t event, svmc_stream_t stream)
{
    SVMC_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return SVMC_OK;
}

int svmc_event_synchronize(svmc_event_t event)
{
    SVMC_HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return SVMC_OK;
}

int svmc_event_elapsed_ms(svmc_event_t start, svmc_event_t stop, float *ms)
{
    SVMC_REQUIRE(ms != nullptr, "svmc_event_elapsed_ms: null output");
    SVMC_HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
    SVMC_HIP_TRY(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return SVMC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// Black-76 implied volatilities of one slice on the HOST (svmc_black.h: the same solver ends the fixed-randoms graph on
// the device, where the prices already are -- svmc_logsv_chain_price_fixed_iv; a stand-alone launch plus copy for a few
// dozen quotes would only be slower than this).
// ---------------------------------------------------------------------------------------------------

int svmc_black_implied_vols(const double *prices, const double *strikes, const int8_t *optiontypes, size_t n_strikes,
                            double forward, double ttm, double discfactor, double vol_lo, double vol_hi, double *ivols)
{
    SVMC_REQUIRE(n_strikes == 0 || (prices && strikes && optiontypes && ivols), "svmc_black_implied_vols: null array");
    SVMC_REQUIRE(forward > 0.0 && ttm > 0.0 && discfactor > 0.0, "svmc_black_implied_vols: forward, ttm, discfactor > 0");
    SVMC_REQUIRE(0.0 < vol_lo && vol_lo < vol_hi, "svmc_black_implied_vols: need 0 < vol_lo < vol_hi");
    for (size_t k = 0; k < n_strikes; ++k)
        if (optiontypes[k] < SVMC_CALL || optiontypes[k] > SVMC_INV_PUT)
            return svmc::fail(SVMC_ERR_UNKNOWN_PAYOFF, "unknown option payoff code");
    for (size_t k = 0; k < n_strikes; ++k) {
        // an inverse option pays (S - K)^+ / S: in units of the underlying its Black-76 value is the vanilla value over the
        // forward -- DF (N(d1) - K/F N(d2)) -- so its implied vol is that of price x forward
        const bool inverse = optiontypes[k] >= SVMC_INV_CALL;
        const bool call = optiontypes[k] == SVMC_CALL || optiontypes[k] == SVMC_INV_CALL;
        ivols[k] = svmc::black_implied_vol(inverse ? prices[k] * forward : prices[k], strikes[k], call, forward, ttm, discfactor,
                                           vol_lo, vol_hi);
    }
    return SVMC_OK;
}
