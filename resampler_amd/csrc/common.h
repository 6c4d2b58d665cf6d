// common.h -- shared host-side helpers of libresampler_amd: error reporting and HIP checks.
#pragma once

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/resampler_amd.h"

namespace rsmp {

// Thread-local message returned by rsmp_last_error().
std::string& last_error_slot();
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace rsmp

#define RSMP_HIP_CHECK(expr)                                                                 \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::rsmp::fail(RSMP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                \
                                hipGetErrorString(_e), __FILE__, __LINE__);                  \
    } while (0)
