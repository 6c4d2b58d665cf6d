#include "common.h"

namespace rsmp {

std::string& last_error_slot() {
    thread_local std::string slot;
    return slot;
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error_slot() = buf;
    return code;
}

}  // namespace rsmp

extern "C" const char* rsmp_last_error(void) { return rsmp::last_error_slot().c_str(); }
extern "C" const char* rsmp_version(void) { return "resampler_amd 0.1.0 (gfx950)"; }

extern "C" uint32_t rsmp_sample_rate_hz(int sample_rate) {
    // impl From<SampleRate> for u32 (reference src/lib.rs:219-236)
    static const uint32_t hz[10] = {22050, 16000, 32000, 44100, 48000,
                                    88200, 96000, 176400, 192000, 384000};
    return (sample_rate >= 0 && sample_rate < 10) ? hz[sample_rate] : 0u;
}
