// fir_mirror_core.h -- the ResamplerFir streaming state machine (src/resampler_fir.rs:509-621) as ONE
// function that compiles for the host and for the device.
//
// The host mirror (FirMirror, fir_plan.cpp) plans bulk / per-call launches with it; the lock-step
// batch kernel (fir_lockstep.hip) runs it on the GPU, one lane per stream, so that a step of a
// thousand streams in a thousand different states costs the host nothing.  Both must reproduce the
// reference's f64 recurrence `position += ratio` (:589) bit for bit -- it decides the (consumed,
// produced) counts and, next to integer positions, the window / phase row of an output -- so the
// arithmetic below is restricted to operations that are exact or correctly rounded on both sides
// (f64 add / sub / fma / floor and exponent-field arithmetic; the one division and the multiply
// by its result only seed a search that is then corrected with exact fma tests).  Build with -ffp-contract=off.
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
    uint32_t next_int;        // outputs until the next one whose exact position is an integer:
                              // (den - abs_out % den) % den (den < 2^32: the rates are u32)
};

struct FirCallCounts {
    uint64_t accepted;   // input frames copied into the resampler (frames_to_copy, :526-528)
    uint64_t produced;   // output frames produced (:588)
    uint64_t consumed;   // frames retired from the front of the buffer (:596)
};

// Largest k >= 0 with p0 + k*inc < bound, given p0 < bound, inc > 0, every p0 + k*inc up to the bound
// exactly representable, and a seed `est` (any value; the two loops make the answer exact).
template <class Idx>
__host__ __device__ inline Idx mirror_refine_last_below(double p0, double inc, double bound, double est) {
    Idx k = est >= 1.0 ? static_cast<Idx>(est) : Idx(0);
    while (k > 0 && fma(static_cast<double>(k), inc, p0) >= bound) --k;
    while (fma(static_cast<double>(k + 1), inc, p0) < bound) ++k;
    return k;
}

__host__ __device__ inline uint64_t mirror_bits(double v) {
    union { double d; uint64_t u; } x;
    x.d = v;
    return x.u;
}
__host__ __device__ inline double mirror_from_bits(uint64_t u) {
    union { double d; uint64_t u; } x;
    x.u = u;
    return x.d;
}

// The outputs of one position run (p_k = pos + k*inc, k < run; `count` = call-relative index of its first output)
// whose exact position n_abs*num/den is an integer (k = next_int, next_int + den, ...): the rounded f64 position is
// that integer + d_k with |d_k| tiny, and d_k moves LINEARLY with k inside a run (every p_k is exact), so its sign
// changes at most once: evaluate the two ends, search the change if there is one.  Reports the wrapped ones (d < 0:
// floor() picks the previous frame) to the sink, leaves the last deviation in st.drift and returns the new next_int
// (outputs from the END of the run to the next integer position).  Requires next_int < run.
template <class Idx, class Sink>
__host__ __device__ inline Idx mirror_run_wraps(FirMirrorState& st, Idx next_int, Idx count, Idx run, double pos, double inc,
                                                Idx den, double den_d, Sink& sink) {
    const Idx n_int = (run - 1 - next_int) / den + 1;
    const double next_d = static_cast<double>(next_int);
    auto dev = [&](Idx i) -> double {   // signed distance of the i-th such position from its integer
        const double p = fma(fma(static_cast<double>(i), den_d, next_d), inc, pos);
        const double fr = p - floor(p);
        return fr > 0.5 ? fr - 1.0 : fr;
    };
    const double d_first = dev(0), d_last = n_int > 1 ? dev(n_int - 1) : d_first;
    st.drift = d_last;
    // How far the f64 position may sit from the exact rational one before the stream stops being treated as periodic.
    // What the periodic kernels and the run planner take from exact arithmetic -- which input frame an output's window
    // starts at, which binade or call an output falls into -- holds while no output other than those AT an integer
    // position can change sides of one: exact positions are multiples of 1 / den, so while |drift| < 1 / den; and the
    // wrap variant's single clamped row (row 1023, :562-564) while |drift| < 1 / 1024.  With margin: 0.6 / den, at most
    // 2^-12.  (Rounds 1-4: 1e-5 for every ratio, which a stream reaches after ~8 hours of audio -- from then on it took
    // the reference-form path, several times slower; 147 / 160 now has 2.4e-4: eight days.  VERDICT r04 item 9.)
    const double by_den = 0.6 / den_d;
    const double bound = by_den < 0x1p-12 ? by_den : 0x1p-12;
    if ((d_first < 0.0 ? -d_first : d_first) > bound || (d_last < 0.0 ? -d_last : d_last) > bound)
        st.periodic_ok = 0;
    // wrapped = below the integer (d < 0): floor() picks the previous frame
    Idx w_begin = 0, w_end = 0;   // [w_begin, w_end) of the n_int positions
    if (d_first < 0.0 && d_last < 0.0) {
        w_end = n_int;
    } else if (d_first < 0.0 || d_last < 0.0) {
        Idx lo = 0, hi = n_int - 1;   // the sign at lo differs from the sign at hi
        while (hi - lo > 1) {
            const Idx mid = lo + (hi - lo) / 2;
            if ((dev(mid) < 0.0) == (d_first < 0.0)) lo = mid; else hi = mid;
        }
        if (d_first < 0.0) { w_begin = 0; w_end = hi; } else { w_begin = hi; w_end = n_int; }
    }
    for (Idx i = w_begin; i < w_end; ++i)
        sink.wrap(static_cast<uint64_t>(count) + next_int + static_cast<uint64_t>(i) * den);
    return static_cast<Idx>(static_cast<uint64_t>(next_int) + static_cast<uint64_t>(n_int) * den - run);
}

// One reference resample() call in frames.  `Sink` receives the exact position runs
//   sink.run(out_index_of_first_frame, count, p0, inc)          (p_k = p0 + k*inc, inc == 0: one frame)
// and, when sink.want_wraps() and the stream is still periodic, the call-relative indices of outputs
// whose exact position n*num/den is an integer but whose f64 position landed just below it
//   sink.wrap(out_index)
// (floor() then picks the previous input frame and the phase clamps to row 1023, :562-564, instead
// of row 0 of the next frame -- the one discrete choice that depends on the sign of the f64 drift).
// The output loop of one call (:542-590) in closed form.  Idx = uint32_t when the output capacity
// is below 2^31 (32-bit integer and conversion instructions on the device), uint64_t otherwise.
template <class Idx, class Sink>
__host__ __device__ inline uint64_t mirror_output_loop(FirMirrorState& st, Idx output_capacity, double limit,
                                                       double& pos_io, Sink& sink) {
    Idx count = 0;
    double pos = pos_io;
    const bool rational = sink.want_wraps() && st.periodic_ok != 0;
    const double inv_ratio = 1.0 / st.ratio;   // seeds the run-length searches only (they are then made exact)
    Idx next_int = static_cast<Idx>(st.next_int);
    const Idx den = static_cast<Idx>(st.den);   // (only used when rational: den < 2^32)
    const double den_d = static_cast<double>(st.den);
    while (count < output_capacity && pos < limit) {
        Idx run = 0;
        double inc = 0.0;
        const uint64_t pbits = mirror_bits(pos);
        if (pos > 0.0 && (pbits >> 52) != 0) {   // positive and normal
            const double top = mirror_from_bits(((pbits >> 52) + 1) << 52);  // pos in [top/2, top)
            const double p1 = pos + st.ratio;
            if (p1 < top) {
                inc = p1 - pos;  // exact: same binade
                const double p2 = p1 + st.ratio;
                // Equal consecutive increments: RN(ratio) on this binade's grid, with the
                // round-half-even parity (if ratio is a tie on this grid) already settled.
                if (p2 < top && (p2 - p1) == inc) {
                    // n = the largest k with p_k < top: the outputs at p_0 .. p_n lie on this binade's
                    // grid (n + 1 of them); the add after p_n crosses into the next binade and rounds there.
                    const Idx room = output_capacity - count;
                    const double est = floor((top - pos) * inv_ratio);
                    Idx m;   // outputs of this run
                    if (est > static_cast<double>(room) + 4.0) {
                        m = room;
                    } else {
                        m = mirror_refine_last_below<Idx>(pos, inc, top, est) + 1;
                        if (room < m) m = room;
                    }
                    if (!(fma(static_cast<double>(m - 1), inc, pos) < limit))   // p_(m-1) must be below the limit
                        m = mirror_refine_last_below<Idx>(pos, inc, limit, floor((limit - pos) * inv_ratio)) + 1;
                    run = m;
                }
            }
        }
        if (run == 0) {
            run = 1;
            inc = 0.0;
        }
        sink.run(count, run, pos, inc);
        if (rational && next_int < run) {
            next_int = mirror_run_wraps<Idx>(st, next_int, count, run, pos, inc, den, den_d, sink);
        } else if (rational) {
            next_int -= run;
        }
        // next position: p_(run-1) + ratio, one rounded add -- exact (= p_run) while it stays inside the
        // binade, the reference's rounding when it crosses into the next one
        pos = fma(static_cast<double>(run - 1), inc, pos) + st.ratio;
        count += run;
    }
    if (rational) {
        st.next_int = static_cast<uint32_t>(next_int);
    } else if (count != 0) {   // not tracked run by run: one modulo per call
        const uint64_t ph = (st.abs_out + count) % st.den;
        st.next_int = static_cast<uint32_t>(ph ? st.den - ph : 0);
    }
    pos_io = pos;
    return count;
}

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
    if (st.available >= st.taps) {
        const double limit = static_cast<double>(st.available - st.taps) + 1.0;
        count = output_capacity < 0x7FFFFFF0ull
                    ? mirror_output_loop<uint32_t>(st, static_cast<uint32_t>(output_capacity), limit, pos, sink)
                    : mirror_output_loop<uint64_t>(st, output_capacity, limit, pos, sink);
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
