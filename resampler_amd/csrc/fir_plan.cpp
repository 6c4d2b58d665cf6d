// fir_plan.cpp -- see fir_plan.h.  Build with -ffp-contract=off: every `a + b` below must be the
// single IEEE-754 double add the reference performs; std::fma is used only where the product and
// sum are provably exact.
#include "fir_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>

#include "common.h"
#include "filter_design.h"
#include "fir_mirror_fast.h"

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


// Self-test of the run planner's fast path (fir_mirror_fast.h) against mirror_call, on the host: `calls` calls of
// `in_frames` frames from the plan's current state, predicted in runs of `run_len` calls.  Every call is done both ways
// (mirror_call_fast falling back to mirror_call where a check fails, exactly as the device planner does) and compared:
// counts, every bit of the state, the outputs taking the row-1023 variant, the drift.  The plan ends in the state after
// the calls.  mismatches: calls that differed (must be 0); slow_calls: calls the fast path declined.
extern "C" int rsmp_fir_plan_selftest_fast(rsmp_fir_plan* p, size_t in_frames, size_t calls, size_t run_len,
                                           size_t* mismatches, size_t* slow_calls, size_t* lean_calls) {
    if (!p || run_len == 0 || in_frames == 0 || in_frames > rsmp::kMirrorInputCapacity)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_plan_selftest_fast: invalid argument");
    struct WrapSink {
        std::vector<uint64_t>* w;
        bool want_wraps() const { return true; }
        void run(uint64_t, uint64_t, double, double) {}
        void wrap(uint64_t index) { w->push_back(index); }
    };
    rsmp::FirMirrorState ref = p->mirror.state(), fast = ref;
    const uint64_t cap = p->mirror.buffer_size_output_frames();
    size_t bad = 0, slow = 0, n_lean = 0, off_track = 0;
    std::vector<uint64_t> w_ref, w_fast;
    for (size_t done = 0; done < calls; done += run_len) {
        const uint32_t len = static_cast<uint32_t>(std::min(run_len, calls - done));
        const rsmp::MirrorRunBase base = rsmp::mirror_run_base(fast, static_cast<uint32_t>(in_frames), len);
        const rsmp::MirrorBinades bn = rsmp::mirror_binades(fast.ratio, base.e0);
        const bool chain_ready = rsmp::mirror_chain_ready(bn);
        rsmp::MirrorEdges edges;
        for (uint32_t i = 0; i <= rsmp::kPredBinades; ++i) rsmp::mirror_edge(base, i, edges.q[i], edges.r[i]);
        for (uint32_t c = 0; c < len; ++c) {
            w_ref.clear();
            w_fast.clear();
            WrapSink s_ref{&w_ref}, s_fast{&w_fast};
            const rsmp::FirCallCounts c_ref = rsmp::mirror_call(ref, in_frames, cap, s_ref);
            rsmp::FirCallCounts c_fast{};
            const rsmp::FirMirrorState before = fast;
            bool took_fast = false;
            if (base.usable) {
                const rsmp::MirrorPred pr = rsmp::mirror_predict(base, c);
                {   // K1's form of it (the edge divisions shared): the same prediction, field for field
                    const rsmp::MirrorPred pe = rsmp::mirror_predict_edges(base, edges, c);
                    if (std::memcmp(&pe, &pr, sizeof pr) != 0) ++bad;
                }
                took_fast = rsmp::mirror_call_fast(fast, static_cast<uint32_t>(in_frames), cap, pr, bn, c_fast,
                                                   [](uint32_t, uint32_t, double, double) {});
                // the unchecked chain, where the device planner takes it: must land on the same state
                if (took_fast && pr.ties == 0 && chain_ready && before.abs_out == pr.m0) {
                    rsmp::FirMirrorState lean = before;
                    rsmp::FirCallCounts c_lean{};
                    rsmp::mirror_call_chain(lean, static_cast<uint32_t>(in_frames), pr, bn, c_lean);
                    if (std::memcmp(&lean, &fast, sizeof lean) != 0 || c_lean.produced != c_fast.produced ||
                        c_lean.consumed != c_fast.consumed)
                        ++bad;
                }
                // ... and the chain kernel's form of it (the counters apart, the call's shape unrolled; calls with an output
                // exactly at a binade's edge included), where the kernel takes it: on the prediction's track
                const rsmp::ChainPlan cp = rsmp::mirror_chain_plan(pr);
                if (took_fast && (cp.ctl & rsmp::kChainLean) && chain_ready && before.abs_out == pr.m0 && before.abs_consumed == pr.c0 &&
                    pr.n_total + 1 < cap) {
                    rsmp::ChainScalars sc{before.abs_out, before.abs_consumed, static_cast<uint32_t>(before.read_position),
                                          static_cast<uint32_t>(before.available)};
                    double pos = before.position;
                    const uint32_t shape = (cp.ctl >> 8) & 0xFu;
                    const bool ties = ((cp.ctl >> 12) & 0xFFFu) != 0;
                    const uint32_t cons = ties ? rsmp::mirror_chain_step_any<true>(shape, pos, sc, static_cast<uint32_t>(in_frames), before.ratio, bn, pr.n_total, cp)
                                               : rsmp::mirror_chain_step_any<false>(shape, pos, sc, static_cast<uint32_t>(in_frames), before.ratio, bn, pr.n_total, cp);
                    ++n_lean;
                    if (c + 1 < len) {   // (diagnostic: does the next call's prediction start where this call ends?)
                        const rsmp::MirrorPred nx = rsmp::mirror_predict(base, c + 1);
                        if (nx.c0 - pr.c0 != cons || nx.m0 != pr.m0 + pr.n_total) ++off_track;
                    }
                    if (cons != c_fast.consumed || c_fast.produced != pr.n_total || std::memcmp(&pos, &fast.position, sizeof pos) != 0 ||
                        sc.abs_out != fast.abs_out || sc.abs_consumed != fast.abs_consumed || sc.read_position != fast.read_position ||
                        sc.available != fast.available || pr.ni_after != fast.next_int)
                        ++bad;
                }
                if (took_fast) {   // the outputs at integer positions: by replay from the call's start state
                    rsmp::FirMirrorState start = before;
                    start.read_position = 0;
                    if (rsmp::mirror_replay_wraps(start, static_cast<uint32_t>(in_frames), pr, bn, s_fast)) {
                        fast.drift = start.drift;
                        if (start.periodic_ok == 0) fast.periodic_ok = 0;
                    }
                }
            }
            if (!took_fast) {
                ++slow;
                c_fast = rsmp::mirror_call(fast, in_frames, cap, s_fast);
            }
            const bool same = c_ref.accepted == c_fast.accepted && c_ref.produced == c_fast.produced &&
                              c_ref.consumed == c_fast.consumed && std::memcmp(&ref, &fast, sizeof ref) == 0 && w_ref == w_fast;
            if (!same) ++bad;
        }
    }
    p->mirror.set_state(ref);
    if (mismatches) *mismatches = bad;
    if (slow_calls) *slow_calls = slow;
    if (lean_calls) *lean_calls = n_lean;
    if (rsmp::knob("RSMP_SELFTEST_VERBOSE")) fprintf(stderr, "[rsmp] selftest: %zu calls, %zu lean, %zu of them leave the prediction's track, %zu slow\n", calls, n_lean, off_track, slow);
    return RSMP_OK;
}
