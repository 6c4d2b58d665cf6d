// fir_mirror_core.h -- the ResamplerFir streaming state machine (src/resampler_fir.rs:509-621) as ONE
// function that compiles for the host and for the device.
//
// The host mirror (FirMirror, fir_plan.cpp) plans bulk / per-call launches with it; the lock-step
// batch kernel (fir_lockstep.hip) runs it on the GPU, one lane per stream, so that a step of a
// thousand streams in a thousand different states costs the host nothing.  Both must reproduce the
// reference's f64 recurrence `position += ratio` (:589) bit for bit -- it decides the (consumed,
// produced) counts and, next to integer positions, the window / phase row of an output -- so the
// arithmetic below is restricted to operations that are exact or correctly rounded on both sides
// (f64 add / sub / fma / floor / frexp / ldexp; the one division only seeds a search that is then
// corrected with exact fma tests).  Build with -ffp-contract=off.
//
// Closed form: inside one binade [2^e, 2^(e+1)) every rounded add moves the position by the same
// multiple of the binade's ulp, so a run of outputs is p_k = p0 + k*inc with p0, inc and every p_k
// exactly representable (the round-half-even parity of a tie is settled by comparing two consecutive
// increments).  A call is ~12 runs instead of hundreds of dependent adds.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdint>

namespace rsmp {

constexpr uint32_t kMirrorInputCapacity = 4096;  // INPUT_CAPACITY, resampler_fir.rs:18
constexpr uint32_t kMirrorBufferSize = 8192;     // BUFFER_SIZE,    resampler_fir.rs:19

// Plain data: lives in host objects and in HBM alike.
struct FirMirrorState {
    double ratio;             // in_hz / out_hz in f64 (resampler_fir.rs:313)
    uint64_t num, den;        // the same ratio as a reduced fraction
    uint64_t taps;
    uint64_t read_position;   // resampler_fir.rs:190
    uint64_t available;       // available_frames, :191
    double position;          // :192
    uint64_t abs_out;         // output frames produced since reset
    uint64_t abs_consumed;    // input frames retired since reset
    double drift;             // f64 position minus exact rational position at the last integer-position output
    uint32_t periodic_ok;     // 0 once |drift| exceeded what the class tables tolerate
    uint32_t pad;
};

struct FirCallCounts {
    uint64_t accepted;   // input frames copied into the resampler (frames_to_copy, :526-528)
    uint64_t produced;   // output frames produced (:588)
    uint64_t consumed;   // frames retired from the front of the buffer (:596)
};

// Largest k >= 0 with p0 + k*inc < bound, given p0 < bound, inc > 0 and every p0 + k*inc up to
// the bound exactly representable.
__host__ __device__ inline uint64_t mirror_last_below(double p0, double inc, double bound) {
    double est = floor((bound - p0) / inc);
    if (est < 0.0) est = 0.0;
    uint64_t k = static_cast<uint64_t>(est);
    while (k > 0 && fma(static_cast<double>(k), inc, p0) >= bound) --k;
    while (fma(static_cast<double>(k + 1), inc, p0) < bound) ++k;
    return k;
}

// One reference resample() call in frames.  `Sink` receives the exact position runs
//   sink.run(out_index_of_first_frame, count, p0, inc)          (p_k = p0 + k*inc, inc == 0: one frame)
// and, when sink.want_wraps() and the stream is still periodic, the call-relative indices of outputs
// whose exact position n*num/den is an integer but whose f64 position landed just below it
//   sink.wrap(out_index)
// (floor() then picks the previous input frame and the phase clamps to row 1023, :562-564, instead
// of row 0 of the next frame -- the one discrete choice that depends on the sign of the f64 drift).
template <class Sink>
__host__ __device__ inline FirCallCounts mirror_call(FirMirrorState& st, uint64_t input_frames,
                                                     uint64_t output_capacity, Sink& sink) {
    // resampler_fir.rs:524-528
    const uint64_t write_position = st.read_position + st.available;
    const uint64_t remaining_capacity = kMirrorBufferSize > write_position ? kMirrorBufferSize - write_position : 0;
    uint64_t accepted = input_frames < remaining_capacity ? input_frames : remaining_capacity;
    if (accepted > kMirrorInputCapacity - st.available) accepted = kMirrorInputCapacity - st.available;
    st.available += accepted;

    // Output loop (:542-590): frames are produced while floor(pos) + taps <= available, i.e.
    // while pos < available - taps + 1, and while the output has room.
    uint64_t count = 0;
    double pos = st.position;
    const bool any = st.available >= st.taps;
    const double limit = any ? static_cast<double>(st.available - st.taps) + 1.0 : 0.0;
    const bool rational = sink.want_wraps() && st.periodic_ok != 0;

    while (any && count < output_capacity && pos < limit) {
        uint64_t run = 0;
        double inc = 0.0;
        if (pos > 0.0) {
            int e;
            (void)frexp(pos, &e);
            const double top = ldexp(1.0, e);  // pos in [top/2, top)
            const double p1 = pos + st.ratio;
            if (p1 < top) {
                inc = p1 - pos;  // exact: same binade
                const double p2 = p1 + st.ratio;
                // Equal consecutive increments: RN(ratio) on this binade's grid, with the
                // round-half-even parity (if ratio is a tie on this grid) already settled.
                if (p2 < top && (p2 - p1) == inc) {
                    uint64_t n = mirror_last_below(pos, inc, top);  // p_n < top: steps 0..n regular
                    const uint64_t n_valid = mirror_last_below(pos, inc, limit) + 1;  // p_k < limit
                    if (n_valid < n) n = n_valid;
                    const uint64_t room = output_capacity - count;
                    if (room < n) n = room;
                    run = n;
                }
            }
        }
        if (run == 0) {
            run = 1;
            inc = 0.0;
        }
        sink.run(count, run, pos, inc);
        if (rational) {
            // Outputs whose exact position n_abs*num/den is an integer: the rounded f64 position is
            // that integer +- drift.
            const uint64_t first_abs = st.abs_out + count;
            uint64_t k = (st.den - first_abs % st.den) % st.den;
            for (; k < run; k += st.den) {
                const double p = fma(static_cast<double>(k), inc, pos);
                const double fr = p - floor(p);
                const double dist = fr > 0.5 ? 1.0 - fr : fr;
                st.drift = fr > 0.5 ? fr - 1.0 : fr;
                if (dist > 1e-5) st.periodic_ok = 0;
                if (fr > 0.5) sink.wrap(count + k);
            }
        }
        pos = (inc == 0.0) ? pos + st.ratio : fma(static_cast<double>(run), inc, pos);
        count += run;
    }

    // :596-602
    uint64_t consumed = static_cast<uint64_t>(floor(pos));
    if (consumed > st.available) consumed = st.available;
    st.read_position += consumed;
    st.available -= consumed;
    st.position = pos - static_cast<double>(consumed);
    // :605-615 (the device keeps no ring; only the index bookkeeping matters for `accepted`)
    if (st.read_position > kMirrorInputCapacity) st.read_position = 0;

    st.abs_out += count;
    st.abs_consumed += consumed;
    return FirCallCounts{accepted, count, consumed};
}

}  // namespace rsmp
