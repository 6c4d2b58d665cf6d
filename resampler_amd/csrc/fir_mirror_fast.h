// fir_mirror_fast.h -- the ResamplerFir state machine (src/resampler_fir.rs:509-621) for RUNS of equal calls, with the
// serial part cut down to what is serial.
//
// fir_mirror_core.h replays a call's f64 recurrence `position += ratio` (:589) in closed form, a dozen position runs
// per call; what it spends per run -- finding the run's length (a division, two searches), the outputs at integer
// positions, the bookkeeping -- is ~200 dependent instructions, ~14 k cycles per call on one GPU lane.  A run of k
// calls per stream (rsmp_fir_lockstep_run) is k times that, and nothing of it shrinks with the batch.  But only the
// f64 VALUES are serial.  The STRUCTURE of a call -- how many outputs lie below each power of two, how many the call
// produces, how many frames it retires -- follows from exact integer arithmetic on the stream's absolute counters
// (output m sits at input position m * num / den; a call that has accepted A frames in total produces the outputs
// below A - taps + 1), for every call of the run at once, independently:
//   mirror_predict      one (stream, call): the predicted structure (MirrorPred), any number of them in parallel;
//   mirror_call_fast    the serial chain of one call, given its prediction: per binade p1 = pos + ratio, inc = p1 - pos,
//                       last = pos + (n - 1) inc, pos = last + ratio -- four dependent operations -- and CHECKS that
//                       make the prediction safe rather than trusted: every run inside one binade with equal
//                       increments (the premise of the closed form), every produced position below the call's limit,
//                       the next one not.  Then the call is the plain recurrence, whatever predicted it.  A failed
//                       check (an output exactly AT a boundary, where the sign of the f64 drift decides: once in
//                       ~den calls) leaves the state untouched and the caller takes mirror_call for that call;
//   the outputs at integer positions (the row-1023 variant, the drift) are found afterwards by REPLAYING a call's
//   chain from its recorded start position -- bit-identical by construction -- in parallel over the calls.
// tests/test_fir_host.py checks mirror_call_fast against mirror_call call by call (counts, state bits, wrapped
// outputs) through rsmp_fir_plan_selftest_fast.
#pragma once

#include "fir_mirror_core.h"

namespace rsmp {

constexpr uint32_t kPredBinades = 12;

struct MirrorRunBase {        // what a run of equal calls starts from
    uint64_t abs_out0, abs_consumed0, avail0;
    uint64_t num, den, taps;
    uint32_t in_frames;       // frames offered (and accepted) per call
    uint32_t e0;              // binades [2^(e0+i), 2^(e0+i+1)), i < kPredBinades, in closed form; outputs below 2^e0 one by one
    uint32_t usable;          // 0: no predictions for this stream (counters beyond the exact range): every call by mirror_call
};

constexpr uint16_t kPredLimitTie = 0x8000, kPredIrregular = 0x4000;
struct MirrorPred {           // the predicted structure of one call
    uint64_t m0, c0;          // abs_out / abs_consumed at the call's start
    uint32_t n_total;         // outputs of the call
    uint16_t n_low;           // ... of them below 2^e0
    uint16_t ties;            // where the f64 drift has a say: bit i < 12: binade i's first output sits EXACTLY at 2^(e0+i);
                              // kPredLimitTie: an output sits exactly at the call's limit; kPredIrregular: counts clamped
    uint32_t ni_before, ni_after;   // next_int at the call's start / end
    uint16_t n[kPredBinades]; // outputs in [2^(e0+i), 2^(e0+i+1)) (cut by the call's limit)
};

// Per stream: what the closed form needs to know about the binades [2^(e0+i), 2^(e0+i+1)).  Inside one binade every
// f64 there is a multiple of the binade's ulp G, so a rounded add of the ratio moves ANY of them by the same amount --
// the ratio rounded to a multiple of G -- unless the ratio sits exactly half-way between two multiples (a tie, settled
// by the parity of the sum: at most one binade per ratio, the one whose G is twice the ratio's lowest set bit).
struct MirrorBinades {
    double half0;                 // 2^e0
    double inc[kPredBinades];     // the ratio rounded to binade i's grid; 0: a tie there, no closed form
};
__host__ __device__ inline MirrorBinades mirror_binades(double ratio, uint32_t e0) {
    MirrorBinades b;
    b.half0 = mirror_from_bits(static_cast<uint64_t>(1023 + e0) << 52);
    double half = b.half0;
#pragma unroll
    for (uint32_t i = 0; i < kPredBinades; ++i, half += half) {
        const double inc = (half + ratio) - half;    // (ratio < half / 2: the sum stays inside the binade)
        const double r = ratio - inc;                // exact
        const double half_ulp = half * 1.1102230246251565e-16;   // 2^-53
        b.inc[i] = (r == half_ulp || r == -half_ulp) ? 0.0 : inc;
    }
    return b;
}

// first binade with at least four outputs (the closed form needs three points of a run inside it)
__host__ __device__ inline uint32_t mirror_first_binade(double ratio) {
    uint32_t e = 0;
    double b = 1.0;
    while (b < 4.0 * ratio && e < 40) { b *= 2.0; ++e; }
    return e;
}

__host__ __device__ inline MirrorRunBase mirror_run_base(const FirMirrorState& st, uint32_t in_frames, uint32_t calls) {
    MirrorRunBase b;
    b.abs_out0 = st.abs_out;
    b.abs_consumed0 = st.abs_consumed;
    b.avail0 = st.available;
    b.num = st.num;
    b.den = st.den;
    b.taps = st.taps;
    b.in_frames = in_frames;
    b.e0 = mirror_first_binade(st.ratio);
    // every product below stays under 2^63: frames < 2^40, num / den < 2^21, 2^(e0 + 12) < 2^40
    const uint64_t frames_end = st.abs_consumed + st.available + static_cast<uint64_t>(in_frames) * calls;
    b.usable = st.num != 0 && st.den != 0 && st.num < (1ull << 21) && st.den < (1ull << 21) && frames_end < (1ull << 40) &&
               st.abs_out < (1ull << 40) && b.e0 + kPredBinades < 40 && st.periodic_ok != 0;
    return b;
}

// ceil(x * den / num): the number of outputs m >= 0 with m * num / den < x
__host__ __device__ inline uint64_t mirror_outputs_below(uint64_t x, uint64_t num, uint64_t den) {
    return (x * den + num - 1) / num;
}

// The structure of call c (0-based) of the run, in exact arithmetic.
__host__ __device__ inline MirrorPred mirror_predict(const MirrorRunBase& b, uint32_t c) {
    MirrorPred pr;
    const uint64_t a_prev = b.abs_consumed0 + b.avail0 + static_cast<uint64_t>(c) * b.in_frames;   // frames accepted before the call
    const uint64_t a_now = a_prev + b.in_frames;
    uint64_t m0 = b.abs_out0, c0 = b.abs_consumed0;
    if (c != 0) {
        if (a_prev + 1 > b.taps) {
            const uint64_t m = mirror_outputs_below(a_prev + 1 - b.taps, b.num, b.den);
            if (m > m0) m0 = m;
        }
        c0 = m0 * b.num / b.den;      // floor of the next output's position, capped by what has been accepted
        if (c0 > a_prev) c0 = a_prev;
        if (c0 < b.abs_consumed0) c0 = b.abs_consumed0;
    }
    uint64_t m1 = m0;
    uint16_t ties = 0;
    if (a_now + 1 > b.taps && a_now - c0 >= b.taps) {
        const uint64_t t = (a_now + 1 - b.taps) * b.den;
        const uint64_t m = (t + b.num - 1) / b.num;
        if (m > m1) m1 = m;
        if (t % b.num == 0 && m >= m0) ties |= kPredLimitTie;   // output m sits exactly at the limit
    }
    pr.m0 = m0;
    pr.c0 = c0;
    pr.n_total = static_cast<uint32_t>(m1 - m0);
    const uint64_t r0 = m0 % b.den, r1 = m1 % b.den;
    pr.ni_before = static_cast<uint32_t>(r0 ? b.den - r0 : 0);
    pr.ni_after = static_cast<uint32_t>(r1 ? b.den - r1 : 0);
    auto below = [&](uint32_t e, bool& tie) -> uint32_t {   // outputs of the call with position < 2^e
        const uint64_t t = (c0 + (1ull << e)) * b.den;
        uint64_t m = (t + b.num - 1) / b.num;
        tie = t % b.num == 0 && m >= m0 && m < m1;    // output m of this call sits exactly at 2^e
        if (m < m0) m = m0;
        if (m > m1) m = m1;
        return static_cast<uint32_t>(m - m0);
    };
    bool tie = false;
    uint32_t prev = below(b.e0, tie);
    if (tie) ties |= 1u;
    if (prev > 0xFFFFu) { prev = 0xFFFFu; ties |= kPredIrregular; }
    pr.n_low = static_cast<uint16_t>(prev);
    for (uint32_t i = 0; i < kPredBinades; ++i) {
        const uint32_t v = below(b.e0 + i + 1, tie);
        if (tie && i + 1 < kPredBinades) ties |= static_cast<uint16_t>(1u << (i + 1));
        if (v - prev > 0xFFFFu) ties |= kPredIrregular;
        pr.n[i] = static_cast<uint16_t>(v - prev > 0xFFFFu ? 0xFFFFu : v - prev);
        prev = v;
    }
    // (a ratio with a power-of-two denominator is exact in f64: its positions carry no drift, nothing ever sits a hair
    // below where exact arithmetic puts it)
    if ((b.den & (b.den - 1)) == 0) ties = 0;
    if (prev != pr.n_total) ties |= kPredIrregular;   // (outputs at 2^(e0+12) and beyond: no such call exists)
    pr.ties = ties;
    return pr;
}

// The same prediction with the thirteen binade-edge divisions shared by all calls of a stream's run: (c0 + 2^e) den / num
// = c0 den / num + 2^e den / num, whose second term is a constant of the stream and the run (MirrorEdges: floor and
// remainder per edge) -- one division of c0 den per call, an add with carry per edge.  Six 64-bit divisions a call instead
// of eighteen: K1 is a kernel of CODE (a 64-bit division is ~70 instructions), which is what it loses beside the split
// kernel of the run before (72 us for a 128-stream shard against 10 alone).  Bit for bit mirror_predict
// (rsmp_fir_plan_selftest_fast compares the two for every call).
struct MirrorEdges {
    uint64_t q[kPredBinades + 1];   // floor(2^(e0+i) den / num)
    uint32_t r[kPredBinades + 1];   // ... and the remainder (< num < 2^21)
};
__host__ __device__ inline void mirror_edge(const MirrorRunBase& b, uint32_t i, uint64_t& q, uint32_t& r) {
    const uint64_t t = (1ull << (b.e0 + i)) * b.den;   // < 2^40 * 2^21
    q = t / b.num;
    r = static_cast<uint32_t>(t - q * b.num);
}
__host__ __device__ inline MirrorPred mirror_predict_edges(const MirrorRunBase& b, const MirrorEdges& ed, uint32_t c) {
    MirrorPred pr;
    const uint64_t a_prev = b.abs_consumed0 + b.avail0 + static_cast<uint64_t>(c) * b.in_frames;
    const uint64_t a_now = a_prev + b.in_frames;
    uint64_t m0 = b.abs_out0, c0 = b.abs_consumed0;
    if (c != 0) {
        if (a_prev + 1 > b.taps) {
            const uint64_t m = mirror_outputs_below(a_prev + 1 - b.taps, b.num, b.den);
            if (m > m0) m0 = m;
        }
        c0 = m0 * b.num / b.den;
        if (c0 > a_prev) c0 = a_prev;
        if (c0 < b.abs_consumed0) c0 = b.abs_consumed0;
    }
    uint64_t m1 = m0;
    uint16_t ties = 0;
    if (a_now + 1 > b.taps && a_now - c0 >= b.taps) {
        const uint64_t t = (a_now + 1 - b.taps) * b.den;
        const uint64_t qf = t / b.num;
        const bool exact = t == qf * b.num;
        const uint64_t m = qf + (exact ? 0u : 1u);
        if (m > m1) m1 = m;
        if (exact && m >= m0) ties |= kPredLimitTie;
    }
    pr.m0 = m0;
    pr.c0 = c0;
    pr.n_total = static_cast<uint32_t>(m1 - m0);
    const uint64_t r0 = m0 % b.den, r1 = m1 % b.den;
    pr.ni_before = static_cast<uint32_t>(r0 ? b.den - r0 : 0);
    pr.ni_after = static_cast<uint32_t>(r1 ? b.den - r1 : 0);
    // c0 den = qA num + rA, once; edge i: (qA + q_i) num + (rA + r_i)
    const uint64_t A = c0 * b.den;
    const uint64_t qA = A / b.num;
    const uint32_t rA = static_cast<uint32_t>(A - qA * b.num), num32 = static_cast<uint32_t>(b.num);
    auto below = [&](uint32_t i, bool& tie) -> uint32_t {   // outputs of the call with position < 2^(e0+i)
        uint32_t rem = rA + ed.r[i];
        const bool carry = rem >= num32;
        rem -= carry ? num32 : 0u;
        uint64_t m = qA + ed.q[i] + (carry ? 1u : 0u) + (rem != 0 ? 1u : 0u);
        tie = rem == 0 && m >= m0 && m < m1;
        if (m < m0) m = m0;
        if (m > m1) m = m1;
        return static_cast<uint32_t>(m - m0);
    };
    bool tie = false;
    uint32_t prev = below(0, tie);
    if (tie) ties |= 1u;
    if (prev > 0xFFFFu) { prev = 0xFFFFu; ties |= kPredIrregular; }
    pr.n_low = static_cast<uint16_t>(prev);
    for (uint32_t i = 0; i < kPredBinades; ++i) {
        const uint32_t v = below(i + 1, tie);
        if (tie && i + 1 < kPredBinades) ties |= static_cast<uint16_t>(1u << (i + 1));
        if (v - prev > 0xFFFFu) ties |= kPredIrregular;
        pr.n[i] = static_cast<uint16_t>(v - prev > 0xFFFFu ? 0xFFFFu : v - prev);
        prev = v;
    }
    if ((b.den & (b.den - 1)) == 0) ties = 0;
    if (prev != pr.n_total) ties |= kPredIrregular;
    pr.ties = ties;
    return pr;
}

// One call, given its predicted structure.  Returns false -- `st` untouched -- when a check fails; otherwise the call is
// done exactly as mirror_call does it (same counts, same state bits).  on_run(first, count, p0, inc) receives the
// position runs (inc == 0: a single output), as Sink::run of mirror_call.
// The one place where the counts themselves hang on the f64 drift is an output whose exact position IS the call's limit
// (an integer): a hair below it in f64 and the call produces it, otherwise the next call does.  For ratios like 1/3
// that is every call, so it is part of the fast path: a call may produce ONE output beyond its prediction (the loop of
// :542-590 simply goes on while pos < limit), and a call that finds its first predicted output already produced
// (abs_out one ahead of the prediction, the frames retired as predicted -- so only while ratio < 1, where the output
// behind an integer position shares its input frame) drops it from the structure.
template <class OnRun>
__host__ __device__ inline bool mirror_call_fast(FirMirrorState& st, uint32_t in_frames, uint64_t output_capacity,
                                                 const MirrorPred& pr, const MirrorBinades& bn, FirCallCounts& out, OnRun&& on_run) {
    // resampler_fir.rs:524-528: the whole offer must be accepted
    const uint64_t write_position = st.read_position + st.available;
    if (write_position + in_frames > kMirrorBufferSize || st.available + in_frames > kMirrorInputCapacity) return false;
    const uint64_t ahead = st.abs_out - pr.m0;   // outputs of this call's prediction that the previous call produced
    if (ahead > 1 || ahead > pr.n_total || st.abs_consumed != pr.c0 || pr.n_total + 1 >= output_capacity) return false;
    const uint32_t target = pr.n_total - static_cast<uint32_t>(ahead);
    uint32_t skip = static_cast<uint32_t>(ahead);
    const uint64_t avail = st.available + in_frames;
    const double ratio = st.ratio;
    double pos = st.position;
    uint32_t cnt = 0;
    bool ok = true;
    if (avail >= st.taps) {
        const double limit = static_cast<double>(avail - st.taps) + 1.0;
        auto plain = [&](uint32_t n) {   // one by one
            for (uint32_t k = 0; k < n; ++k) {
                ok = ok && pos < limit;
                on_run(cnt, 1u, pos, 0.0);
                pos += ratio;
                ++cnt;
            }
        };
        {
            uint32_t n = pr.n_low;   // below 2^e0
            if (skip && n) { --n; skip = 0; }
            plain(n);
        }
        double half = bn.half0;
#pragma unroll
        for (uint32_t i = 0; i < kPredBinades; ++i, half += half) {   // binade [half, 2 half)
            uint32_t n = pr.n[i];
            if (skip && n) { --n; skip = 0; }
            if (n == 0) continue;
            // The binade's first output may sit EXACTLY at `half` in exact arithmetic and a hair below it in f64 (an output
            // at an integer position with the drift negative): it is then still the lower binade's -- one plain add.
            const uint32_t pre = pos < half ? 1u : 0u;
            const double inc = bn.inc[i];
            const bool closed = n - pre >= 3 && inc != 0.0;
            plain(closed ? pre : n);
            if (!closed) continue;
            n -= pre;
            // p_k = pos + k inc, k < n: exact while they stay inside the binade (inc = the ratio rounded to the binade's
            // grid: what every rounded add of the reference moves by there) -- and they must all lie below the call's limit
            const double last = fma(static_cast<double>(n - 1), inc, pos);
            ok = ok && pos >= half && last < half + half && last < limit;
            on_run(cnt, n, pos, inc);
            pos = last + ratio;   // the add that may leave the binade: the reference's own rounded add
            cnt += n;
        }
        if (ok && cnt == target && pos < limit) plain(1);   // the output AT the limit, a hair below it in f64
        ok = ok && !(pos < limit);   // the loop of :542-590 stops here
    }
    if (!ok || cnt < target || cnt > target + 1) return false;
    // :596-615
    uint64_t consumed = static_cast<uint64_t>(floor(pos));
    if (consumed > avail) consumed = avail;
    st.read_position += consumed;
    st.available = avail - consumed;
    st.position = pos - static_cast<double>(consumed);
    if (st.read_position > kMirrorInputCapacity) st.read_position = 0;
    st.abs_out += cnt;
    st.abs_consumed += consumed;
    st.next_int = cnt == target ? pr.ni_after : (pr.ni_after ? pr.ni_after - 1 : static_cast<uint32_t>(st.den) - 1);
    out = FirCallCounts{in_frames, cnt, consumed};
    return true;
}

// The chain alone, for a call whose prediction has nothing for the f64 drift to decide (pr.ties == 0) on a stream with
// no tie binade (mirror_chain_ready): two dependent operations per binade, no checks -- those are mirror_call_fast's,
// which the replay of the call (mirror_replay_wraps) runs in parallel afterwards.  The caller has made sure that the
// state is where the prediction starts (abs_out == m0, abs_consumed == c0) and that the whole offer is accepted.
__host__ __device__ inline bool mirror_chain_ready(const MirrorBinades& bn) {
    bool ok = true;
#pragma unroll
    for (uint32_t i = 0; i < kPredBinades; ++i) ok = ok && bn.inc[i] != 0.0;
    return ok;
}
__host__ __device__ inline void mirror_call_chain(FirMirrorState& st, uint32_t in_frames, const MirrorPred& pr,
                                                  const MirrorBinades& bn, FirCallCounts& out) {
    const uint64_t avail = st.available + in_frames;
    const double ratio = st.ratio;
    double pos = st.position;
    for (uint32_t k = 0; k < pr.n_low; ++k) pos += ratio;
#pragma unroll
    for (uint32_t i = 0; i < kPredBinades; ++i) {
        const uint32_t n = pr.n[i];
        if (n >= 3) {
            pos = fma(static_cast<double>(n - 1), bn.inc[i], pos) + ratio;
        } else {
            for (uint32_t k = 0; k < n; ++k) pos += ratio;
        }
    }
    uint64_t consumed = static_cast<uint64_t>(floor(pos));
    if (consumed > avail) consumed = avail;
    st.read_position += consumed;
    st.available = avail - consumed;
    st.position = pos - static_cast<double>(consumed);
    if (st.read_position > kMirrorInputCapacity) st.read_position = 0;
    st.abs_out += pr.n_total;
    st.abs_consumed += consumed;
    st.next_int = pr.ni_after;
    out = FirCallCounts{in_frames, pr.n_total, consumed};
}

// ---- the chain kernel's form of a call (round 5) ---------------------------------------------------------------------------
// A lone wave issues an instruction every ~4 ns whatever it is, so a call costs what it ISSUES: round 4's loop rebuilt a
// MirrorPred from fourteen v_readlane, kept the counters in 64-bit vector arithmetic and branched per binade -- 220
// instructions, 0.91 us, the Amdahl term of a shard.  Here the lanes prepare, for 64 calls at a time
// and in parallel, what the serial chain needs per call (ChainPlan: the closed form's multipliers as f64, the shape of
// the call as two control words), the integer half of the state lives apart (on the device: in SCALAR registers, updated
// by the scalar unit in the shadow of the f64 latency), and the chain itself is unrolled for the call's shape: the
// binades below the top one are full (>= 4 outputs each: two instructions), the top one is cut by the call's limit.
// Calls with an output EXACTLY at a binade's edge (5 % of config 4's: an integer position that is a power of two, where
// the f64 drift decides which binade's grid the add before it rounds on) take the same chain with that one comparison per
// marked binade (TIES); the rest of mirror_call_fast's checks are made by the replay of the call, as for every call of
// the unchecked chain.  Bit for bit mirror_call_chain / mirror_call_fast: the same rounded operations in the same order;
// min(floor(pos), avail) is taken in f64 (exact: both are integers below 2^53).
struct ChainScalars {
    uint64_t abs_out, abs_consumed;
    uint32_t read_position, available;
};
constexpr uint32_t kChainLean = 1u << 31;
struct ChainPlan {
    double m[kPredBinades];   // binade i: n_i - 1 as f64 where n_i >= 3 (the closed form's multiplier), else 0
    uint32_t ctl;             // bits 0-7 n_low, 8-11 L (the top non-empty binade), 12-23 the prediction's tie bits, 31 kChainLean
    uint32_t n_last;          // outputs in binade L
};
// What a call's prediction says about the chain's shape.  kChainLean: the chain below applies -- no output at the call's
// limit, no clamped count, the binades below the top one full, n_low small.
__host__ __device__ inline ChainPlan mirror_chain_plan(const MirrorPred& pr) {
    ChainPlan p;
    uint32_t top = 0, n_top = pr.n[0];
    bool any = false, full = true;
#pragma unroll
    for (uint32_t i = 0; i < kPredBinades; ++i) {
        const uint32_t n = pr.n[i];
        p.m[i] = n >= 3 ? static_cast<double>(n - 1) : 0.0;
        if (n != 0) { top = i; any = true; n_top = n; }   // (n_top = pr.n[top] without an index that is not a constant: on the device
    }                                                      // that index put the whole prediction record into LDS)
#pragma unroll
    for (uint32_t i = 0; i < kPredBinades; ++i)
        if (i < top && pr.n[i] < 4) full = false;   // (a binade below the top one: >= 4, so that a tie's single add leaves >= 3)
    const bool lean = any && full && pr.n_low < 256 && (pr.ties & (kPredLimitTie | kPredIrregular)) == 0;
    p.ctl = pr.n_low | (top << 8) | ((pr.ties & 0xFFFu) << 12) | (lean ? kChainLean : 0u);
    p.n_last = n_top;
    return p;
}
// One call of shape L (ctl bits 8-11): the f64 chain on `pos`, then what the call retires (:596-615).  `m[i]`, i <= L, and
// the two control words are the call's ChainPlan (on the device: v_readlane of the lane that holds it).  Returns the frames
// the call retires.
// (UNIFORM: every lane of the wave runs the SAME call -- the retired count is taken through a scalar register; false: a call per lane,
// the parallel chain of fir_lockstep_run.hip)
template <uint32_t L, bool TIES, bool UNIFORM = true>
__host__ __device__ inline uint32_t mirror_chain_step(double& pos_io, ChainScalars& sc, uint32_t in_frames, double ratio,
                                                      const MirrorBinades& bn, uint32_t n_total, uint32_t ctl, uint32_t n_last,
                                                      const double (&m)[kPredBinades]) {
    double pos = pos_io;
    for (uint32_t k = ctl & 0xFFu; k != 0; --k) pos += ratio;
    double half = bn.half0;
#pragma unroll
    for (uint32_t i = 0; i < L; ++i, half += half) {   // full binades
        if (TIES && ((ctl >> (12 + i)) & 1u) && pos < half) {
            // the binade's first output, exactly at `half` in exact arithmetic, is a hair below it in f64: still the lower
            // binade's -- one plain add (mirror_call_fast: `pre`)
            pos += ratio;
            pos = fma(m[i] - 1.0, bn.inc[i], pos) + ratio;
        } else {
            pos = fma(m[i], bn.inc[i], pos) + ratio;
        }
    }
    {   // the top binade, cut by the call's limit: any count >= 1
        uint32_t n = n_last;
        if (TIES && ((ctl >> (12 + L)) & 1u) && pos < half) {
            pos += ratio;
            n -= 1;
            if (n >= 3) pos = fma(static_cast<double>(n - 1), bn.inc[L], pos) + ratio;
            else for (uint32_t k = 0; k < n; ++k) pos += ratio;
        } else if (n >= 3) {
            pos = fma(m[L], bn.inc[L], pos) + ratio;
        } else {
            for (uint32_t k = 0; k < n; ++k) pos += ratio;
        }
    }
    const uint32_t avail = sc.available + in_frames;
    const double fl = floor(pos), avd = static_cast<double>(avail);
    const double cd = fl > avd ? avd : fl;        // (double)min(floor(pos), avail), :596-597
    pos_io = pos - cd;                            // :602
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t consumed = UNIFORM ? static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(cd))))
                                      : static_cast<uint32_t>(cd);
#else
    const uint32_t consumed = static_cast<uint32_t>(cd);
#endif
    uint32_t rp = sc.read_position + consumed;
    if (rp > kMirrorInputCapacity) rp = 0;        // :605-615
    sc.read_position = rp;
    sc.available = avail - consumed;
    sc.abs_out += n_total;
    sc.abs_consumed += consumed;
    return consumed;
}
// (host: the shape as a run-time value)
template <bool TIES, uint32_t L = 0>
__host__ inline uint32_t mirror_chain_step_any(uint32_t shape, double& pos, ChainScalars& sc, uint32_t in_frames, double ratio,
                                               const MirrorBinades& bn, uint32_t n_total, const ChainPlan& p) {
    if constexpr (L < kPredBinades) {
        if (shape == L) return mirror_chain_step<L, TIES>(pos, sc, in_frames, ratio, bn, n_total, p.ctl, p.n_last, p.m);
        return mirror_chain_step_any<TIES, L + 1>(shape, pos, sc, in_frames, ratio, bn, n_total, p);
    } else {
        return 0;
    }
}

// The outputs at integer positions of one call done by mirror_call_fast, found by replaying it from its start state
// (`st`: the state the call started from; only position, available and the counters matter).  Sink::wrap as for
// mirror_call; st.drift / st.periodic_ok are updated as mirror_call updates them.  Returns whether the call had an
// output at an integer position (so that st.drift is this call's).
template <class Sink>
__host__ __device__ inline bool mirror_replay_wraps(FirMirrorState& st, uint32_t in_frames, const MirrorPred& pr,
                                                    const MirrorBinades& bn, Sink& sink, bool* checks_ok = nullptr) {
    // (st.abs_out may be one ahead of the prediction: see mirror_call_fast)
    const uint32_t ahead = static_cast<uint32_t>(st.abs_out - pr.m0);
    uint32_t next_int = ahead ? (pr.ni_before ? pr.ni_before - 1 : static_cast<uint32_t>(st.den) - 1) : pr.ni_before;
    if (next_int > pr.n_total - ahead) {   // (== : only the output a call may produce beyond its prediction)
        if (checks_ok) {   // no output at an integer position: the replay has only the chain's premises to confirm
            FirMirrorState work = st;
            FirCallCounts cc;
            *checks_ok = mirror_call_fast(work, in_frames, ~0ull, pr, bn, cc, [](uint32_t, uint32_t, double, double) {});
        }
        return false;
    }
    const uint32_t den = static_cast<uint32_t>(st.den);
    const double den_d = static_cast<double>(st.den);
    FirMirrorState work = st;
    FirCallCounts cc;
    auto on_run = [&](uint32_t first, uint32_t count, double p0, double inc) {
        if (next_int < count) next_int = mirror_run_wraps<uint32_t>(st, next_int, first, count, p0, inc, den, den_d, sink);
        else next_int -= count;
    };
    bool any = false;
    auto on_run_any = [&](uint32_t first, uint32_t count, double p0, double inc) {
        any = any || next_int < count;
        on_run(first, count, p0, inc);
    };
    const bool ok = mirror_call_fast(work, in_frames, ~0ull, pr, bn, cc, on_run_any);
    if (checks_ok) *checks_ok = ok;
    return any;
}

}  // namespace rsmp
