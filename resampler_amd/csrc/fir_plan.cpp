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

FirMirror::FirMirror(uint32_t in_hz, uint32_t out_hz, size_t taps) {
    st_ = FirMirrorState{};
    st_.ratio = static_cast<double>(in_hz) / static_cast<double>(out_hz);
    st_.taps = taps;
    const uint64_t g = std::gcd<uint64_t>(in_hz, out_hz);
    st_.num = in_hz / g;
    st_.den = out_hz / g;
    st_.periodic_ok = 1;
}

void FirMirror::reset() {
    st_.read_position = 0;
    st_.available = 0;
    st_.position = 0.0;
    st_.abs_out = 0;
    st_.abs_consumed = 0;
    st_.periodic_ok = 1;
    st_.drift = 0.0;
    st_.next_int = 0;
}

size_t FirMirror::buffer_size_output_frames() const {
    const double max_usable = static_cast<double>(kInputCapacity - st_.taps);
    return static_cast<size_t>(std::ceil(max_usable / st_.ratio)) + 2;
}

namespace {

// Host sink of mirror_call: appends runs / wraps to the caller's vectors.
struct VectorSink {
    std::vector<rsmp_fir_segment>* segs;
    std::vector<uint32_t>* wraps;
    int64_t in_base;
    uint32_t out_start;
    bool want_wraps() const { return wraps != nullptr; }
    void run(uint64_t first, uint64_t count, double p0, double inc) {
        if (segs)
            segs->push_back(rsmp_fir_segment{out_start + static_cast<uint32_t>(first),
                                             static_cast<uint32_t>(count), in_base, p0, inc});
    }
    void wrap(uint64_t index) { wraps->push_back(out_start + static_cast<uint32_t>(index)); }
};

}  // namespace

FirCallResult FirMirror::call(size_t input_frames, size_t output_capacity, int64_t in_base,
                              uint32_t out_start, std::vector<rsmp_fir_segment>* segs,
                              std::vector<uint32_t>* wraps) {
    VectorSink sink{segs, wraps, in_base, out_start};
    const FirCallCounts c = mirror_call(st_, input_frames, output_capacity, sink);
    return FirCallResult{static_cast<size_t>(c.accepted), static_cast<size_t>(c.produced),
                         static_cast<size_t>(c.consumed)};
}

}  // namespace rsmp

// ---- C ABI: host-only plan handle -----------------------------------------------------------
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
extern "C" rsmp_fir_plan* rsmp_fir_plan_clone(const rsmp_fir_plan* p) { return p ? new rsmp_fir_plan(*p) : nullptr; }
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

// The driver loop of resample/src/main.rs:226-254 on the mirror alone (what rsmp_fir_resample_bulk plans,
// fir_api.cpp plan_job_uncached): every call offers min(chunk, remaining) frames and the full output
// capacity; stops after max_calls calls (0 = no limit) or when the input is used up.
extern "C" int rsmp_fir_plan_bulk(rsmp_fir_plan* p, size_t in_frames, size_t chunk_frames, size_t max_calls,
                                  size_t* frames_accepted, size_t* frames_produced, size_t* n_calls) {
    if (!p || chunk_frames == 0) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_plan_bulk: invalid argument");
    const size_t cap_frames = p->mirror.buffer_size_output_frames();
    size_t offset = 0, produced = 0, calls = 0;
    while (offset < in_frames && (max_calls == 0 || calls < max_calls)) {
        const size_t remaining = in_frames - offset;
        const rsmp::FirCallResult c =
            p->mirror.call(remaining < chunk_frames ? remaining : chunk_frames, cap_frames, 0, 0, nullptr, nullptr);
        produced += c.produced;
        offset += c.accepted;
        ++calls;
        if (c.accepted == 0) break;
    }
    if (frames_accepted) *frames_accepted = offset;
    if (frames_produced) *frames_produced = produced;
    if (n_calls) *n_calls = calls;
    return RSMP_OK;
}

