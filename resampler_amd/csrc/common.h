// common.h -- shared host-side helpers of libresampler_amd: error reporting and HIP checks.
#pragma once

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

#include <hip/hip_runtime.h>

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

// hipEventRecord / hipStreamWaitEvent for a stream that may be the caller's RSMP_STREAM_LEGACY.  This image's runtime
// takes the legacy handle (hipStreamLegacy == (hipStream_t)1) in kernel launches and copies, but an event RECORDED with
// it keeps the handle as its stream, and a later hipStreamWaitEvent on that event (or with the handle as the waiting
// stream) dereferences it -- a segfault inside hip::hipStreamWaitEvent_common, found by tools/run_bulk_probe.py under
// torch's default stream (runs planned ahead are the first path that waits for an event recorded on the caller's
// stream).  The null stream IS the legacy default stream in this library's build (no per-thread default stream), so
// events are given that.
inline hipStream_t event_stream(hipStream_t s) { return s == reinterpret_cast<hipStream_t>(RSMP_STREAM_LEGACY) ? nullptr : s; }
inline hipError_t event_record(hipEvent_t ev, hipStream_t s) { return hipEventRecord(ev, event_stream(s)); }
inline hipError_t stream_wait_event(hipStream_t s, hipEvent_t ev) { return hipStreamWaitEvent(event_stream(s), ev, 0); }

}  // namespace rsmp

#define RSMP_HIP_CHECK(expr)                                                                 \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::rsmp::fail(RSMP_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                \
                                hipGetErrorString(_e), __FILE__, __LINE__);                  \
    } while (0)
