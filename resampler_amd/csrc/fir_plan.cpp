// fir_plan.cpp -- see fir_plan.h.  Build with -ffp-contract=off: every `a + b` below must be the
// single IEEE-754 double add the reference performs; std::fma is used only where the product and
// sum are provably exact.
#include "fir_plan.h"

#include <cmath>
#include <numeric>

#include "common.h"
#include "filter_design.h"

#pragma STDC FP_CONTRACT OFF

namespace rsmp {

FirMirror::FirMirror(uint32_t in_hz, uint32_t out_hz, size_t taps)
    : ratio_(static_cast<double>(in_hz) / static_cast<double>(out_hz)), taps_(taps) {
    const uint64_t g = std::gcd<uint64_t>(in_hz, out_hz);
    num_ = in_hz / g;
    den_ = out_hz / g;
}

void FirMirror::reset() {
    read_position_ = 0;
    available_ = 0;
    position_ = 0.0;
    abs_out_ = 0;
    abs_consumed_ = 0;
    periodic_ok_ = true;
    drift_ = 0.0;
}

size_t FirMirror::buffer_size_output_frames() const {
    const double max_usable = static_cast<double>(kInputCapacity - taps_);
    return static_cast<size_t>(std::ceil(max_usable / ratio_)) + 2;
}

namespace {

// Largest k >= 0 with p0 + k*inc < bound, given p0 < bound, inc > 0 and every p0 + k*inc up to
// the bound exactly representable.
inline uint64_t last_below(double p0, double inc, double bound) {
    double est = std::floor((bound - p0) / inc);
    if (est < 0.0) est = 0.0;
    uint64_t k = static_cast<uint64_t>(est);
    while (k > 0 && std::fma(static_cast<double>(k), inc, p0) >= bound) --k;
    while (std::fma(static_cast<double>(k + 1), inc, p0) < bound) ++k;
    return k;
}

}  // namespace

FirCallResult FirMirror::call(size_t input_frames, size_t output_capacity, int64_t in_base,
                              uint32_t out_start, std::vector<rsmp_fir_segment>* segs,
                              std::vector<uint32_t>* wraps) {
    // resampler_fir.rs:524-528
    const size_t write_position = read_position_ + available_;
    const size_t remaining_capacity = kBufferSize > write_position ? kBufferSize - write_position : 0;
    size_t accepted = input_frames < remaining_capacity ? input_frames : remaining_capacity;
    if (accepted > kInputCapacity - available_) accepted = kInputCapacity - available_;
    available_ += accepted;

    // Output loop (:542-590): frames are produced while floor(pos) + taps <= available, i.e.
    // while pos < available - taps + 1, and while the output has room.
    size_t count = 0;
    double pos = position_;
    const bool any = available_ >= taps_;
    const double limit = any ? static_cast<double>(available_ - taps_) + 1.0 : 0.0;
    const bool rational = wraps != nullptr && periodic_ok_;

    auto note_wraps = [&](double p0, double inc, size_t n, size_t first_count) {
        // Outputs whose exact position n_abs*num/den is an integer: the rounded f64 position is
        // that integer +- drift.  Below it, floor() picks the previous input frame and the phase
        // clamps to row 1023 (:562-564) instead of row 0 -- a 1/1024-sample step the periodic
        // kernel must reproduce per output.
        const uint64_t first_abs = abs_out_ + first_count;
        uint64_t k = (den_ - first_abs % den_) % den_;
        for (; k < n; k += den_) {
            const double p = std::fma(static_cast<double>(k), inc, p0);
            const double fr = p - std::floor(p);
            const double dist = fr > 0.5 ? 1.0 - fr : fr;
            drift_ = fr > 0.5 ? fr - 1.0 : fr;
            if (dist > 1e-5) periodic_ok_ = false;
            if (fr > 0.5) wraps->push_back(out_start + static_cast<uint32_t>(first_count + k));
        }
    };

    while (any && count < output_capacity && pos < limit) {
        size_t run = 0;
        double inc = 0.0;
        if (pos > 0.0) {
            int e;
            (void)std::frexp(pos, &e);
            const double top = std::ldexp(1.0, e);  // pos in [top/2, top)
            const double p1 = pos + ratio_;
            if (p1 < top) {
                inc = p1 - pos;  // exact: same binade
                const double p2 = p1 + ratio_;
                // Equal consecutive increments: RN(ratio) on this binade's grid, with the
                // round-half-even parity (if ratio is a tie on this grid) already settled.
                if (p2 < top && (p2 - p1) == inc) {
                    uint64_t n = last_below(pos, inc, top);  // p_n < top: steps 0..n regular
                    const uint64_t n_valid = last_below(pos, inc, limit) + 1;  // p_k < limit
                    if (n_valid < n) n = n_valid;
                    const uint64_t room = output_capacity - count;
                    if (room < n) n = room;
                    run = static_cast<size_t>(n);
                }
            }
        }
        if (run == 0) {
            run = 1;
            inc = 0.0;
        }
        if (segs)
            segs->push_back(rsmp_fir_segment{out_start + static_cast<uint32_t>(count),
                                             static_cast<uint32_t>(run), in_base, pos, inc});
        if (rational) note_wraps(pos, inc, run, count);
        pos = (inc == 0.0) ? pos + ratio_ : std::fma(static_cast<double>(run), inc, pos);
        count += run;
    }

    // :596-602
    size_t consumed = static_cast<size_t>(std::floor(pos));
    if (consumed > available_) consumed = available_;
    read_position_ += consumed;
    available_ -= consumed;
    position_ = pos - static_cast<double>(consumed);
    // :605-615 (the device keeps no ring; only the index bookkeeping matters for `accepted`)
    if (read_position_ > kInputCapacity) read_position_ = 0;

    abs_out_ += count;
    abs_consumed_ += consumed;
    return FirCallResult{accepted, count, consumed};
}

}  // namespace rsmp

// ---- C ABI: host-only plan handle -----------------------------------------------------------
struct rsmp_fir_plan {
    rsmp::FirMirror mirror;
    explicit rsmp_fir_plan(uint32_t i, uint32_t o, size_t t) : mirror(i, o, t) {}
};

extern "C" rsmp_fir_plan* rsmp_fir_plan_new(uint32_t input_rate_hz, uint32_t output_rate_hz,
                                            int latency) {
    const size_t taps = rsmp::latency_taps(latency);
    if (!taps || !input_rate_hz || !output_rate_hz) {
        rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_plan_new: invalid argument");
        return nullptr;
    }
    return new rsmp_fir_plan(input_rate_hz, output_rate_hz, taps);
}

extern "C" void rsmp_fir_plan_free(rsmp_fir_plan* p) { delete p; }
extern "C" void rsmp_fir_plan_reset(rsmp_fir_plan* p) { if (p) p->mirror.reset(); }

extern "C" void rsmp_fir_plan_state(const rsmp_fir_plan* p, size_t* read_position,
                                    size_t* available_frames, double* position) {
    if (read_position) *read_position = p->mirror.read_position();
    if (available_frames) *available_frames = p->mirror.available();
    if (position) *position = p->mirror.position();
}

extern "C" int rsmp_fir_plan_call(rsmp_fir_plan* p, size_t input_frames,
                                  size_t output_capacity_frames, size_t* frames_accepted,
                                  size_t* frames_produced, rsmp_fir_segment* segs, size_t max_segs,
                                  size_t* n_segs) {
    if (!p) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_plan_call: null plan");
    std::vector<rsmp_fir_segment> v;
    const rsmp::FirCallResult r =
        p->mirror.call(input_frames, output_capacity_frames, 0, 0, segs ? &v : nullptr, nullptr);
    if (frames_accepted) *frames_accepted = r.accepted;
    if (frames_produced) *frames_produced = r.produced;
    if (n_segs) *n_segs = v.size();
    if (segs) {
        if (v.size() > max_segs)
            return rsmp::fail(RSMP_ERR_CAPACITY, "rsmp_fir_plan_call: %zu segments, room for %zu",
                              v.size(), max_segs);
        for (size_t i = 0; i < v.size(); ++i) segs[i] = v[i];
    }
    return RSMP_OK;
}
