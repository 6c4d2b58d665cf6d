// common.h -- shared host-side helpers of libresampler_amd: error reporting and HIP checks.
#pragma once

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/resampler_amd.h"

namespace rsmp {

// The library's diagnostic and A/B switches (RSMP_FIR_DEBUG, RSMP_FIR_WTRACE, RSMP_LS_TRACE, RSMP_FIR_SPLIT_PLANES,
// RSMP_LS_EXACT ...) exist only under ONE environment switch: without RSMP_DEBUG=1 none of them is read, so nothing in
// the environment changes which kernel runs or what it computes.  tests/test_knobs_gpu.py runs every switch that can
// change results in a process of its own.
inline const char* knob(const char* name) {
    static const bool on = [] { const char* e = getenv("RSMP_DEBUG"); return e && *e && *e != '0'; }();
    return on ? getenv(name) : nullptr;
}


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
