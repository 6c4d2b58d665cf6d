// fir_plan.h -- host mirror of the ResamplerFir streaming state machine and the exact position
// planner that lets one GPU launch reproduce any number of reference resample() calls.
//
// The reference advances an f64 `position += ratio` once per output frame and subtracts the
// consumed frame count once per call (resampler_fir.rs:589, :602).  That recurrence is serial and
// its rounding decides both the (consumed, produced) counts and, where the position lands next to
// an integer, which input window / phase row an output uses.  The mirror replays it exactly but
// in closed form: inside one binade [2^e, 2^(e+1)) every rounded add moves the position by the
// same multiple of the binade's ulp, so a whole run of outputs is p_k = p0 + k*inc with p0, inc
// and every p_k exactly representable.  A call is a dozen such runs instead of hundreds of adds,
// and the GPU evaluates p_k independently per output frame with one f64 FMA.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/resampler_amd.h"
#include "fir_mirror_core.h"

namespace rsmp {

struct FirCallResult {
    size_t accepted;   // input frames copied into the resampler (frames_to_copy, :526-528)
    size_t produced;   // output frames produced (:588)
    size_t consumed;   // frames retired from the front of the buffer (:596)
};

class FirMirror {
public:
    FirMirror(uint32_t in_hz, uint32_t out_hz, size_t taps);

    void reset();  // resampler_fir.rs:638-642

    // One reference resample() call (frames, not values).  Appends the exact position runs to
    // `segs` (may be null) using `in_base` as the input-frame index of local position 0 and
    // `out_start` as the index of the call's first output frame.  For rational rate pairs also
    // appends to `wraps` (may be null) the launch-relative indices of outputs whose exact
    // position is an integer but whose f64 position landed just below it (see fir_periodic).
    FirCallResult call(size_t input_frames, size_t output_capacity, int64_t in_base,
                       uint32_t out_start, std::vector<rsmp_fir_segment>* segs,
                       std::vector<uint32_t>* wraps);

    size_t buffer_size_output_frames() const;  // resampler_fir.rs:456-465, per channel

    double ratio() const { return st_.ratio; }
    size_t taps() const { return st_.taps; }
    size_t read_position() const { return st_.read_position; }
    size_t available() const { return st_.available; }
    double position() const { return st_.position; }

    // Rational view in_hz/out_hz = num/den (reduced) and absolute counters since reset().
    uint64_t num() const { return st_.num; }
    uint64_t den() const { return st_.den; }
    uint64_t abs_out() const { return st_.abs_out; }
    uint64_t abs_consumed() const { return st_.abs_consumed; }
    // False once the f64 position has drifted further from n*num/den than the periodic kernel
    // tolerates (never observed; the generic kernel is used from then on).
    bool periodic_ok() const { return st_.periodic_ok != 0; }
    // Signed distance (f64 position - exact rational position) seen at the most recent output
    // whose exact position is an integer.
    double drift() const { return st_.drift; }

    // The plain-data state (shared with the device-side planner, fir_mirror_core.h).
    const FirMirrorState& state() const { return st_; }
    void set_state(const FirMirrorState& s) { st_ = s; }

private:
    FirMirrorState st_;
};

}  // namespace rsmp

// The host-only plan handle of the C ABI (rsmp_fir_plan_*): a mirror without a device.
struct rsmp_fir_plan {
    rsmp::FirMirror mirror;
    explicit rsmp_fir_plan(uint32_t i, uint32_t o, size_t t) : mirror(i, o, t) {}
};
