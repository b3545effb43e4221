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

}  // namespace svmc
