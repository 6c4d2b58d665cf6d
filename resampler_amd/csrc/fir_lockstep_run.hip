// fir_lockstep_run.hip -- the planner of rsmp_fir_lockstep_run: k consecutive resample() calls per stream
// (src/resampler_fir.rs:509-621, driven like resample/src/main.rs:226-254) planned ON THE DEVICE, so that the
// calls of a whole run become ONE launch of the bulk kernels per rate pair instead of k launches of the
// one-call-per-stream kernel.
//
// Three kernels plan the k calls of every stream (fir_mirror_fast.h): K1 the structure of every call in exact integer
// arithmetic (one thread per stream and call), K2 the serial f64 position chain (one wave per stream), K3 the outputs at
// integer positions by a replay of every call in parallel.  They leave, in HBM,
//   * the per-call (consumed, produced) counts of every stream: [k][n] pairs;
//   * the run's stream descriptor (FirStreamDesc, fir_kernels.h) exactly as the host planner of the bulk entry
//     points (fir_api.cpp, launch_jobs) would have built it: outputs / frames accepted / frames retired by the
//     run, the absolute counters it starts from, where its buffered frames are and where the tail goes;
//   * the bitmap of outputs that take the row-1023 variant (position just below an integer, :562-564);
//   * the stream's state after the run.
// The bulk kernels (fir_split.hip, fir_periodic.hip) then read those descriptors: nothing of a run passes
// through the host, whatever states the streams are in.
#include <algorithm>

#include <hip/hip_ext.h>

#include "fir_lockstep.h"

#include "common.h"
#include "fir_mirror_fast.h"

namespace rsmp {

namespace {

struct RunSink {             // mirror_call sink: the wrapped outputs go straight into the run's bitmap
    uint32_t* bits;
    uint32_t rel;            // (absolute index of the call's first output) - wrap_k0 * den
    uint32_t den, n_bits;
    bool periodic, overflow, atomic;
    __host__ __device__ bool want_wraps() const { return periodic; }
    __host__ __device__ void run(uint64_t, uint64_t, double, double) {}
    __device__ void wrap(uint64_t index) {
        const uint32_t K = (rel + static_cast<uint32_t>(index)) / den;
        if (K >= n_bits) overflow = true;
        else if (atomic) (void)atomicOr(bits + (K >> 5), 1u << (K & 31));   // (the calls of a stream are replayed in parallel)
        else bits[K >> 5] |= 1u << (K & 31);
    }
};

constexpr uint32_t kCallSlow = 1, kCallAhead = 2, kCallHasInt = 4, kCallLean = 8;
struct CallRec {             // what the chain leaves per call for the replay of its outputs at integer positions
    double pos;              // the f64 position the call started from
    double drift;            // slow calls: the stream's drift after the call
    uint32_t flags, pad;
};
static_assert(sizeof(CallRec) == 24, "CallRec layout");

// K1 -- the structure of every call of the run, in exact integer arithmetic: one thread per (stream, call), a workgroup
// per stream and 256 calls; its first threads make the stream's binade-edge constants (MirrorEdges: thirteen divisions,
// once per workgroup instead of once per call).
__global__ __launch_bounds__(256) void fir_lockstep_predict_kernel(LsRunArgs a, uint32_t blocks_per_stream, LsCommitArgs cm) {
    __shared__ MirrorEdges edges;
    const uint32_t gs = blockIdx.x / blocks_per_stream;
    const uint32_t c = (blockIdx.x - gs * blocks_per_stream) * 256u + threadIdx.x;
    // (cm.n_streams != 0: the states before this run are still in the scratch copies of the run planned before it; they are read
    // from there, and the stream's first thread puts them -- and what else that plan left -- in place for the kernels behind)
    const FirMirrorState* const states_now = cm.n_streams ? cm.sp_states : a.states_in;
    const MirrorRunBase base = mirror_run_base(states_now[gs], a.in_frames, a.k);
    if (base.usable && threadIdx.x <= kPredBinades) mirror_edge(base, threadIdx.x, edges.q[threadIdx.x], edges.r[threadIdx.x]);
    __syncthreads();
    if (c >= a.k) return;
    if (base.usable) a.preds[static_cast<size_t>(gs) * a.k + c] = mirror_predict_edges(base, edges, c);
    if (c == 0) {
        if (cm.n_streams) {
            cm.states[gs] = cm.sp_states[gs];
            cm.cursor[gs] = cm.sp_cursor[gs];
            cm.last_counts[2 * gs] = cm.sp_last_counts[2 * gs];
            cm.last_counts[2 * gs + 1] = cm.sp_last_counts[2 * gs + 1];
            const uint32_t f = cm.sp_status[gs];
            if (f) cm.status[gs] |= f;
        }
        a.states_before[gs] = states_now[gs];   // (the chain overwrites the states; the replay starts from these)
        if (a.zero_status) a.zero_status[gs] = 0;
    }
    for (uint32_t w = c; w < a.wrap_words; w += a.k) a.wrap_bits[static_cast<size_t>(gs) * a.wrap_words + w] = 0;
}

// K2 -- the serial chain: one wave per stream walks the stream's k calls.  Every lane runs the chain on the same
// values, so the control flow stays uniform (scalar branches, no masking).  Predictions and per-call results are staged
// through the wave's own registers, 64 calls at a time: lane j loads the prediction of call j (the next 64 while these
// 64 run) and keeps call j's record and counts until the 64 are written back.  No LDS: the bulk kernels of the run
// before, which fill a CU's LDS with their ring of images, share their CUs with this kernel when the run is planned
// ahead (fir_lockstep_api.cpp).
//
// Round 5: what is SERIAL per call is the f64 position alone (~25 dependent operations); round 4's loop issued 220
// instructions around it -- fourteen v_readlane to rebuild a MirrorPred, the counters in 64-bit vector arithmetic,
// the checks of the lean rule call by call -- on a wave that is alone on its SIMD: 0.91 us per call whatever the batch,
// the Amdahl term of a shard.  Now
//   * the lanes decide IN PARALLEL, 64 calls at a time, which calls may take the unchecked chain (nothing for the f64
//     drift to decide, room in the output, the whole offer accepted: from the prediction and the exact closed form of
//     the buffered frames) and whether a call's prediction continues its predecessor's (m0 = m0' + n', the frames
//     retired in between): two ballots;
//   * a call on that track takes mirror_chain_lean: nine v_readlane (the binade counts as packed words), the counters in
//     scalar registers, `on track` kept by one scalar compare of the frames it retired with the prediction's;
//   * anything else (6 % of config 4's calls) takes round 4's path unchanged: mirror_call_fast with every premise
//     checked, mirror_call where that declines.
static_assert(sizeof(MirrorPred) == 56, "MirrorPred is staged as fourteen 32-bit words");
__device__ inline uint32_t rl(uint32_t v, uint32_t lane) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>(lane)));
}
__device__ inline uint32_t rfl(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }
__device__ inline uint64_t rfl64(uint64_t v) {
    return static_cast<uint64_t>(rfl(static_cast<uint32_t>(v))) | (static_cast<uint64_t>(rfl(static_cast<uint32_t>(v >> 32))) << 32);
}
__device__ inline double rl_f64(double v, uint32_t lane) {
    const uint64_t b = mirror_bits(v);
    return mirror_from_bits(static_cast<uint64_t>(rl(static_cast<uint32_t>(b), lane)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(b >> 32), lane)) << 32));
}
// Call s of the chunk, whose ChainPlan lane s holds, on the chain of its shape (mirror_chain_step<L, TIES>): the
// multipliers of the binades in use come over by v_readlane.
template <bool TIES, uint32_t L = 0>
__device__ inline uint32_t chain_step_of_shape(uint32_t shape, uint32_t s, const ChainPlan& cp, double& pos, ChainScalars& sc, uint32_t in_frames,
                                               double ratio, const MirrorBinades& bn, uint32_t n_total, uint32_t ctl, uint32_t n_last) {
    if constexpr (L < kPredBinades) {
        if (shape == L) {
            double m[kPredBinades];
#pragma unroll
            for (uint32_t i = 0; i < kPredBinades; ++i) m[i] = i <= L ? rl_f64(cp.m[i], s) : 0.0;
            return mirror_chain_step<L, TIES>(pos, sc, in_frames, ratio, bn, n_total, ctl, n_last, m);
        }
        return chain_step_of_shape<TIES, L + 1>(shape, s, cp, pos, sc, in_frames, ratio, bn, n_total, ctl, n_last);
    } else {
        return 0;
    }
}

// A RUN of calls of one shape without ties, all on the prediction's track: the loop the chain spends its time in, with
// the shape a template parameter of the loop instead of a compare chain per call.  Calls s, s + 1, ... while the masks say
// so; returns the first call it did not take.  (The counters are in `sc`, the position in `pos`.)
template <uint32_t L>
__device__ __forceinline__ uint32_t chain_fast_run(uint32_t s, uint32_t nc, uint64_t fast_mask, uint64_t succ_mask, uint32_t lane, uint32_t my_n_total,
                                                   const ChainPlan& my_cp, uint32_t my_cpred, double& pos, ChainScalars& sc, bool& on_track, double& my_pos,
                                                   uint32_t& lean_last_n, uint32_t in_frames, double ratio, const MirrorBinades& bn) {
#pragma unroll 1
    do {
        my_pos = lane == s ? pos : my_pos;   // lane s keeps call s's start position
        const uint32_t n_total = rl(my_n_total, s), ctl = rl(my_cp.ctl, s), n_last = rl(my_cp.n_last, s), cpred = rl(my_cpred, s);
        double m[kPredBinades];
#pragma unroll
        for (uint32_t i = 0; i < kPredBinades; ++i) m[i] = i <= L ? rl_f64(my_cp.m[i], s) : 0.0;
        const uint32_t cons = mirror_chain_step<L, false>(pos, sc, in_frames, ratio, bn, n_total, ctl, n_last, m);
        on_track = cons == cpred && ((succ_mask >> s) & 1ull);
        lean_last_n = n_total;
        ++s;
    } while (s < nc && on_track && ((fast_mask >> s) & 1ull));
    return s;
}
template <uint32_t L = 0>
__device__ __forceinline__ uint32_t chain_fast_run_of(uint32_t shape, uint32_t s, uint32_t nc, uint64_t fast_mask, uint64_t succ_mask, uint32_t lane,
                                                      uint32_t my_n_total, const ChainPlan& my_cp, uint32_t my_cpred, double& pos, ChainScalars& sc,
                                                      bool& on_track, double& my_pos, uint32_t& lean_last_n, uint32_t in_frames, double ratio,
                                                      const MirrorBinades& bn) {
    if constexpr (L < kPredBinades) {
        if (shape == L)
            return chain_fast_run<L>(s, nc, fast_mask, succ_mask, lane, my_n_total, my_cp, my_cpred, pos, sc, on_track, my_pos, lean_last_n, in_frames, ratio, bn);
        return chain_fast_run_of<L + 1>(shape, s, nc, fast_mask, succ_mask, lane, my_n_total, my_cp, my_cpred, pos, sc, on_track, my_pos, lean_last_n, in_frames,
                                        ratio, bn);
    } else {
        return s;
    }
}

// ---- the chain of a chunk of calls IN PARALLEL (round 6) ------------------------------------------------------------------------
// What is serial per call is the f64 position -- but a lean call's effect on it is a SHIFT that does not depend on where exactly the
// call starts: a rounded add moves every f64 of a binade by the same multiple of that binade's grid (MirrorBinades), the call's start
// p is itself a multiple of the grid of the TOP binade the call before it reached (its last add rounded there, the integer it then
// retires changes nothing), and every grid below divides that one.  So D = F(p) - p is the same for every p of the call's SHAPE (the
// prediction's per-binade counts) -- except where this call climbs one binade higher than the one before: the add that enters the
// higher binade rounds p's lowest bit away, up or down by the ratio's residue there, and D depends on that ONE bit of p.  A call is
// therefore a map on (residue bit r, position): r -> (D[r], r'), and maps compose associatively.  Per chunk of up to 64 calls, a call
// per lane: the lane runs its call's chain (mirror_chain_step, the very function of the serial loop) from two representative starts
// -- the position a plain f64 sum of the prediction puts the call at, rounded to the higher grid, + r grid steps --, a prefix scan
// composes the maps (six shuffle steps), and from the chunk's true start every call's start follows.  Then every lane runs its call
// ONCE MORE from that start and the chunk is accepted only if each call reproduces the shift the scan used and retires what the
// prediction says: p[k + 1] = F_k(p[k]) for every k from the true p[0] IS the serial recurrence, bit for bit.  Anything else -- a
// call that is not lean, calls whose top binades differ by more than one, a start off the grid (a stream's first chunk), a tie the
// representative saw from the other side -- leaves the chunk to the serial loop below, unchanged.  ~2 us per chunk against 64 x 0.41.
struct ChainMap {
    double d0, d1;        // the shift for start residue 0 / 1
    uint32_t bits;        // bit 0 / 1: the residue after the call for start residue 0 / 1; bit 2 / 3: that start is valid
};
__device__ __forceinline__ ChainMap chain_map_identity() { return ChainMap{0.0, 0.0, 0x2u | 0xCu}; }   // r -> r, both valid
// `first` then `second`
__device__ __forceinline__ ChainMap chain_map_compose(const ChainMap& first, const ChainMap& second) {
    ChainMap m;
    const uint32_t o0 = first.bits & 1u, o1 = (first.bits >> 1) & 1u;
    m.d0 = first.d0 + (o0 ? second.d1 : second.d0);
    m.d1 = first.d1 + (o1 ? second.d1 : second.d0);
    const uint32_t so0 = (second.bits >> o0) & 1u, so1 = (second.bits >> o1) & 1u;
    const uint32_t v0 = ((first.bits >> 2) & 1u) & ((second.bits >> (2 + o0)) & 1u), v1 = ((first.bits >> 3) & 1u) & ((second.bits >> (2 + o1)) & 1u);
    m.bits = so0 | (so1 << 1) | (v0 << 2) | (v1 << 3);
    return m;
}
__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    const uint64_t b = mirror_bits(v);
    const uint32_t lo = static_cast<uint32_t>(__shfl_up(static_cast<int>(static_cast<uint32_t>(b)), d, 64));
    const uint32_t hi = static_cast<uint32_t>(__shfl_up(static_cast<int>(static_cast<uint32_t>(b >> 32)), d, 64));
    return mirror_from_bits(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}
// The prefix scans' steps as DPP moves (a ds_bpermute per 32 bits was ~250 cycles a step): the value of the lane 1 / 2 / 4 / 8 to the
// left within a row of 16, then lane 15 of rows 0 / 2 to rows 1 / 3, then lane 31 to rows 2 and 3 -- lanes without a source get `old`.
constexpr int kDppShr1 = 0x111, kDppShr2 = 0x112, kDppShr4 = 0x114, kDppShr8 = 0x118, kDppBcast15 = 0x142, kDppBcast31 = 0x143;
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v, uint32_t old) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(old), static_cast<int>(v), CTRL, ROWS, 0xF, false));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_f64(double v) {   // (`old`: 0.0)
    const uint64_t b = mirror_bits(v);
    return mirror_from_bits(static_cast<uint64_t>(dpp_u32<CTRL, ROWS>(static_cast<uint32_t>(b), 0u)) |
                            (static_cast<uint64_t>(dpp_u32<CTRL, ROWS>(static_cast<uint32_t>(b >> 32), 0u)) << 32));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ ChainMap dpp_map(const ChainMap& v) {   // (`old`: the identity)
    return ChainMap{dpp_f64<CTRL, ROWS>(v.d0), dpp_f64<CTRL, ROWS>(v.d1), dpp_u32<CTRL, ROWS>(v.bits, 0x2u | 0xCu)};
}
// inclusive prefix sums over the wave's 64 lanes (exact where the terms are multiples of one power of two and small)
__device__ __forceinline__ double wave_prefix_sum(double v) {
    v += dpp_f64<kDppShr1, 0xF>(v);
    v += dpp_f64<kDppShr2, 0xF>(v);
    v += dpp_f64<kDppShr4, 0xF>(v);
    v += dpp_f64<kDppShr8, 0xF>(v);
    v += dpp_f64<kDppBcast15, 0xA>(v);
    v += dpp_f64<kDppBcast31, 0xC>(v);
    return v;
}
// ... and of maps: lane j gets map 0 then 1 then ... then j
__device__ __forceinline__ ChainMap wave_prefix_maps(ChainMap v) {
    v = chain_map_compose(dpp_map<kDppShr1, 0xF>(v), v);
    v = chain_map_compose(dpp_map<kDppShr2, 0xF>(v), v);
    v = chain_map_compose(dpp_map<kDppShr4, 0xF>(v), v);
    v = chain_map_compose(dpp_map<kDppShr8, 0xF>(v), v);
    v = chain_map_compose(dpp_map<kDppBcast15, 0xA>(v), v);
    v = chain_map_compose(dpp_map<kDppBcast31, 0xC>(v), v);
    return v;
}
// The lane's own call from `start`: the position after it (already less what it retires) and the frames it retires.
// (TIES: some lane's call has an output exactly on a binade's edge in exact arithmetic -- mirror_chain_step looks which side f64 puts it)
template <bool TIES, uint32_t L = 0>
__device__ inline uint32_t chain_eval_lane(uint32_t shape, double& pos, uint32_t avail, uint32_t in_frames, double ratio, const MirrorBinades& bn,
                                           uint32_t n_total, const ChainPlan& cp) {
    if constexpr (L < kPredBinades) {
        if (shape == L) {
            ChainScalars tmp{0, 0, 0, avail};
            return mirror_chain_step<L, TIES, false>(pos, tmp, in_frames, ratio, bn, n_total, cp.ctl, cp.n_last, cp.m);
        }
        return chain_eval_lane<TIES, L + 1>(shape, pos, avail, in_frames, ratio, bn, n_total, cp);
    } else {
        return 0xFFFFFFFFu;
    }
}

// (Waves per workgroup: one, or -- small batches -- kLsPlanPack, a stream each.  A wave of this kernel is a chain of dependent
// f64 operations: four of them on one CU run as fast as one per CU (eight do not: the kernel is code, and a CU fetches it once for all its waves), and a batch of 128 streams then occupies 32 CUs
// instead of a wave on each of 128 -- where no workgroup of the split kernel, which needs a CU's whole register file, could
// start until that wave was through: whether the bulk launch or the planner reached the chip first made a run of the
// 128-stream shard take 0.72 or 0.91 us per step, profiles/r06/ab_c4_shard.txt.)
// (PCHAIN: the build with the parallel chain and without round 5's run of equal-shape calls -- chain_fast_run, twelve instantiations --
// and the other way round for RSMP_LS_PCHAIN=0: a wave of this kernel is bound by instruction fetch, and neither build needs both)
template <bool PCHAIN>
__global__ __launch_bounds__(64 * kLsPlanPack) void fir_lockstep_chain_kernel(LsRunArgs a) {
    const uint32_t gs = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (gs >= a.n_streams) return;
    const LockstepStream ls = a.streams[gs];
    const LsRunStream rs = a.rs[gs];
    FirMirrorState st = a.states_in[gs];
    uint32_t* bits = a.wrap_bits + static_cast<size_t>(gs) * a.wrap_words;
    const MirrorRunBase base = mirror_run_base(st, a.in_frames, a.k);
    const MirrorBinades bn = mirror_binades(st.ratio, base.e0);
    const bool chain_ready = base.usable && mirror_chain_ready(bn);

    const uint32_t C = rs.channels;
    const uint64_t abs_out0 = st.abs_out, abs_consumed0 = st.abs_consumed;
    const uint64_t k0 = abs_out0 / rs.wrap_unit;
    const uint32_t hist_frames = static_cast<uint32_t>(st.available);
    const bool wraps_exist = rs.wrap_unit == rs.den;   // (a super period of an exact ratio: no output ever wraps)
    RunSink sink{bits, static_cast<uint32_t>(abs_out0 - k0 * rs.wrap_unit), rs.den, a.wrap_words * 32u, wraps_exist, false, false};
    uint32_t flags = 0;
    uint32_t last_c0 = 0, last_c1 = 0, lean_last_n = 0;   // the latest call's counts (a lean call's: in_frames, lean_last_n)
    bool last_lean = false;
    const uint64_t* preds = reinterpret_cast<const uint64_t*>(a.preds + static_cast<size_t>(gs) * a.k);
    CallRec* recs = reinterpret_cast<CallRec*>(a.call_recs) + static_cast<size_t>(gs) * a.k;
    // the state while the stream runs on the unchecked chain: the f64 position in a vector register, the counters in
    // scalar ones; `st` holds the state only while st_valid (at the start, after a call off that track)
    const double ratio = st.ratio;
    double pos = st.position;
    ChainScalars sc{rfl64(st.abs_out), rfl64(st.abs_consumed), rfl(static_cast<uint32_t>(st.read_position)),
                    rfl(static_cast<uint32_t>(st.available))};
    bool st_valid = true, on_track = true;
    uint32_t lean_ni_after = 0;   // next_int behind the latest lean call (scalar; read from the prediction when `st` is filled)
    const uint64_t frames0 = rfl64(base.abs_consumed0 + base.avail0);   // frames accepted before the run
    const uint32_t out_cap = rfl(static_cast<uint32_t>(ls.out_cap_frames < 0xFFFFFFFFull ? ls.out_cap_frames : 0xFFFFFFFFull));
    uint64_t mine[7], ahead[7], succ[2], succ_ahead[2];
    auto fetch = [&](uint32_t c0, uint64_t (&v)[7], uint64_t (&sv)[2]) {
        const uint32_t c = c0 + lane < a.k ? c0 + lane : a.k - 1;
        const uint32_t cn = c + 1 < a.k ? c + 1 : a.k - 1;
#pragma unroll
        for (int i = 0; i < 7; ++i) v[i] = __builtin_nontemporal_load(preds + static_cast<size_t>(c) * 7 + i);
        sv[0] = __builtin_nontemporal_load(preds + static_cast<size_t>(cn) * 7);       // the next call's m0, c0
        sv[1] = __builtin_nontemporal_load(preds + static_cast<size_t>(cn) * 7 + 1);
    };
    fetch(0, ahead, succ_ahead);
    for (uint32_t c0 = 0; c0 < a.k; c0 += 64) {
        const uint32_t nc = a.k - c0 < 64u ? a.k - c0 : 64u;
#pragma unroll
        for (int i = 0; i < 7; ++i) mine[i] = ahead[i];
        succ[0] = succ_ahead[0];
        succ[1] = succ_ahead[1];
        if (c0 + 64 < a.k) fetch(c0 + 64, ahead, succ_ahead);
        // ---- the lanes' part: which of these 64 calls may take the unchecked chain, and which continue their predecessor
        const uint32_t my_call = c0 + lane;
        const uint32_t my_n_total = static_cast<uint32_t>(mine[2]);
        MirrorPred my_pr;
        __builtin_memcpy(&my_pr, mine, sizeof my_pr);
        const ChainPlan my_cp = mirror_chain_plan(my_pr);   // (what the chain needs of this lane's call: multipliers, shape)
        // frames buffered when call j starts, IF the stream is on the prediction's track there: accepted - retired
        const uint64_t my_avail = frames0 + static_cast<uint64_t>(my_call) * a.in_frames - mine[1];
        const bool my_struct_ok = lane < nc && (my_cp.ctl & kChainLean) && my_n_total + 1 < out_cap && my_avail + a.in_frames <= kMirrorInputCapacity;
        const bool my_last = my_call + 1 >= a.k;
        const bool my_succ_ok = my_last || succ[0] == mine[0] + my_n_total;
        const uint32_t my_cpred = static_cast<uint32_t>(succ[1] - mine[1]);   // frames the prediction has the call retire
        const uint64_t lean_mask = __ballot(my_struct_ok && chain_ready);
        const uint64_t succ_mask = __ballot(my_succ_ok);
        // the chunk's common shape (its first such call's): runs of calls of that shape without ties take chain_fast_run
        const bool my_fast = my_struct_ok && chain_ready && (my_cp.ctl & 0xFFF000u) == 0;
        const uint64_t any_fast = __ballot(my_fast);
        const uint32_t shape_d = any_fast ? rl((my_cp.ctl >> 8) & 0xFu, static_cast<uint32_t>(__builtin_ctzll(any_fast))) : 0xFFu;
        const uint64_t fast_mask = __ballot(my_fast && ((my_cp.ctl >> 8) & 0xFu) == shape_d);
        uint64_t nonlean_mask = 0;
        double my_pos = 0.0, my_drift = 0.0;
        uint32_t my_flags = 0, my_c0 = 0, my_c1 = 0;
        // ---- a RUN of lean calls [s0, e0) of the chunk in parallel (every call of it lean, each one's prediction continuing its
        // predecessor's); true: done, the state is behind call e0 - 1
        auto parallel_run = [&](uint32_t s0, uint32_t e0) -> bool {
            const uint64_t full_mask = ((e0 >= 64u ? ~0ull : (1ull << e0) - 1ull)) & ~((1ull << s0) - 1ull);
            const bool in_seg = lane >= s0 && lane < e0;
            bool chunk_done = false;
            {
            const double p_start = st_valid ? st.position : pos;
            const uint32_t my_shape = (my_cp.ctl >> 8) & 0xFu;
            // where the prediction puts every call's start in plain f64 (good to ~1e-12: the same SHAPE as the true start unless the f64
            // drift is smaller than that -- then the check below fails and the serial loop takes the chunk): the run's start + the outputs
            // since x the ratio - the frames retired since, from the predictions' absolute counters
            const uint64_t m0_s0 = static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[0]), s0)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[0] >> 32), s0)) << 32);
            const uint64_t c0_s0 = static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[1]), s0)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[1] >> 32), s0)) << 32);
            const uint32_t outs_before = in_seg ? static_cast<uint32_t>(mine[0] - m0_s0) : 0u, retired_before = in_seg ? static_cast<uint32_t>(mine[1] - c0_s0) : 0u;
            const double x_start = p_start + (static_cast<double>(outs_before) * ratio - static_cast<double>(retired_before));
            const bool any_ties = __any(in_seg && (my_cp.ctl & 0xFFF000u) != 0);
            auto eval = [&](double& pe) -> uint32_t {
                return any_ties ? chain_eval_lane<true>(my_shape, pe, static_cast<uint32_t>(my_avail), a.in_frames, ratio, bn, my_n_total, my_cp)
                                : chain_eval_lane<false>(my_shape, pe, static_cast<uint32_t>(my_avail), a.in_frames, ratio, bn, my_n_total, my_cp);
            };
            // the representative start: on the grid of [4096, 8192) -- a multiple of every grid a call of at most 4096 buffered frames rounds on
            const double kCoarse = 1099511627776.0;   // 2^40
            const double rep = floor(x_start * kCoarse + 0.5) * (1.0 / kCoarse);
            // ... from which the call's LAST add (the one behind its last output) lands in this lane's top binade
            double pe0 = rep;
            const uint32_t cons0 = in_seg ? eval(pe0) : 0u;
            const double raw0 = pe0 + static_cast<double>(cons0 == 0xFFFFFFFFu ? 0u : cons0);
            const uint32_t my_exp = in_seg ? static_cast<uint32_t>((mirror_bits(raw0) >> 52) & 0x7FFu) : 0u;   // biased exponent of the end position
            const uint32_t exp0 = rl(my_exp, s0);
            const uint64_t eq = __ballot(in_seg && my_exp == exp0), up = __ballot(in_seg && my_exp == exp0 + 1u),
                           dn = __ballot(in_seg && my_exp + 1u == exp0);
            uint32_t e_lo = exp0, e_hi = exp0;
            bool shapes_ok = exp0 >= 1023u - 2u && exp0 <= 1023u + 12u;
            if ((eq | up) == full_mask) e_hi = up ? exp0 + 1u : exp0;
            else if ((eq | dn) == full_mask) e_lo = exp0 - 1u;
            else shapes_ok = false;
            const bool two = e_hi != e_lo;
            // the grids of the lower / higher top binade
            const double u_lo = mirror_from_bits(static_cast<uint64_t>(e_lo - 52u) << 52);
            const double inv_lo = mirror_from_bits(static_cast<uint64_t>(2046u + 52u - e_lo) << 52);   // 1 / u_lo
            const double q_start = p_start * inv_lo;            // exact (a power of two)
            shapes_ok = shapes_ok && q_start == floor(q_start) && q_start < 9007199254740992.0;   // the start lies on the lower grid
            if (shapes_ok) {
                const uint32_t r_start = two ? static_cast<uint32_t>(static_cast<uint64_t>(q_start) & 1ull) : 0u;
                ChainMap me = chain_map_identity();
                double d_own0 = 0.0, d_own1 = 0.0;   // (scalars: an array indexed by the residue would live in LDS)
                if (in_seg) {
                    uint32_t bits = 0;
#pragma unroll
                    for (uint32_t r = 0; r < 2; ++r) {
                        const double s0 = rep + (r ? u_lo : 0.0);
                        double pe = r == 0 ? pe0 : s0;
                        uint32_t cons = cons0;
                        if (r == 1) cons = two ? eval(pe) : 0u;
                        const double d = pe - s0;                 // exact: both on the lower grid
                        const double q_end = pe * inv_lo;
                        // (the run's LAST call has no successor to predict what it retires: whatever the chain says)
                        const bool ok = (r == 0 || two) && (cons == my_cpred || my_last) && cons != 0xFFFFFFFFu && q_end == floor(q_end) && pe >= 0.0;
                        const uint32_t r_end = two ? static_cast<uint32_t>(static_cast<uint64_t>(q_end < 0.0 ? 0.0 : q_end) & 1ull) : 0u;
                        if (r == 0) d_own0 = d; else d_own1 = d;
                        bits |= (r_end << r) | ((ok ? 1u : 0u) << (2 + r));
                    }
                    if (!two) {   // (one grid: residue 1 does not exist; the map ignores it)
                        d_own1 = d_own0;
                        bits = (bits & 0x5u) | ((bits & 1u) << 1) | ((bits & 4u) << 1);
                    }
                    me = ChainMap{d_own0, d_own1, bits};
                }
                // (one grid: every map is r -> 0 with one shift -- a prefix sum of the shifts and a vote on the validity)
                ChainMap incl;
                if (two) {
                    incl = wave_prefix_maps(me);
                } else {
                    const double sum = wave_prefix_sum(me.d0);
                    const bool all_ok = __all(!in_seg || ((me.bits >> 2) & 1u) != 0);
                    incl = ChainMap{sum, sum, all_ok ? 0xCu : 0u};
                }
                // this lane's call starts behind calls 0 .. lane - 1
                ChainMap excl;
                excl.d0 = shfl_up_f64(incl.d0, 1);
                excl.d1 = two ? shfl_up_f64(incl.d1, 1) : excl.d0;
                excl.bits = two ? static_cast<uint32_t>(__shfl_up(static_cast<int>(incl.bits), 1, 64)) : incl.bits;
                if (lane <= s0) excl = chain_map_identity();
                const double p_mine = p_start + (r_start ? excl.d1 : excl.d0);
                const uint32_t r_mine = (excl.bits >> r_start) & 1u;
                const bool guess_ok = ((incl.bits >> (2 + r_start)) & 1u) != 0;
                // the check: the call once more, from the start the scan gives it
                bool same = true;
                uint32_t my_cons = 0;
                if (in_seg) {
                    double pe = p_mine;
                    my_cons = eval(pe);
                    same = guess_ok && (my_cons == my_cpred || my_last) && my_cons != 0xFFFFFFFFu && (pe - p_mine) == (r_mine ? d_own1 : d_own0);
                }
                if (__all(same)) {
                    if (st_valid) {   // (the counters leave `st`)
                        sc = ChainScalars{rfl64(st.abs_out), rfl64(st.abs_consumed), rfl(static_cast<uint32_t>(st.read_position)),
                                          rfl(static_cast<uint32_t>(st.available))};
                        st_valid = false;
                    }
                    const uint32_t last = e0 - 1;
                    pos = rl_f64(p_start + (r_start ? incl.d1 : incl.d0), last);
                    my_pos = in_seg ? p_mine : my_pos;
                    // the counters behind the chunk, from the last call's prediction; the ring's read position (:605-615: back to 0 when it
                    // passes the capacity) from the frames retired before every call -- call by call only if some call does send it back
                    // (the ring's read position goes back to 0 every capacity / call size calls: from one such call to the next by a vote
                    // on the frames retired since, not call by call)
                    const uint32_t retired_through = retired_before + my_cons;   // frames the run has retired through this lane's call
                    uint32_t rp_base = sc.read_position;   // the read position where the count `since` starts ...
                    uint32_t since = 0;                    // ... the frames retired up to there
                    uint64_t left = full_mask;
                    for (;;) {
                        const uint64_t over = __ballot(in_seg && rp_base + (retired_through - since) > kMirrorInputCapacity) & left;
                        if (!over) break;
                        const uint32_t j = static_cast<uint32_t>(__builtin_ctzll(over));   // the first call that sends it back
                        since = rl(retired_through, j);
                        rp_base = 0;
                        left = j >= 63u ? 0ull : left & ~((2ull << j) - 1ull);
                    }
                    const uint32_t rp = rp_base + (rl(retired_through, last) - since);
                    const uint32_t n_last_total = rl(my_n_total, last), c_last = rl(my_cons, last);
                    const uint64_t m0_last = static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[0]), last)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[0] >> 32), last)) << 32);
                    const uint64_t c0_last = static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[1]), last)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[1] >> 32), last)) << 32);
                    sc.abs_out = m0_last + n_last_total;
                    sc.abs_consumed = c0_last + c_last;
                    sc.read_position = rp;
                    sc.available = rl(static_cast<uint32_t>(my_avail), last) + a.in_frames - c_last;
                    on_track = ((succ_mask >> last) & 1ull) != 0;
                    lean_last_n = n_last_total;
                    last_lean = true;
                    chunk_done = true;
                }
            }
        }
            return chunk_done;
        };
#pragma unroll 1
        for (uint32_t s = 0; s < nc; ++s) {
            if constexpr (PCHAIN) {
                if (!on_track && st_valid) {   // (behind a call off the track: is the state where this call's prediction starts?  As below.)
                    const uint64_t m0 = static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[0]), s)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[0] >> 32), s)) << 32);
                    const uint64_t cc0 = static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[1]), s)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[1] >> 32), s)) << 32);
                    on_track = rfl64(st.abs_out) == m0 && rfl64(st.abs_consumed) == cc0 &&
                               rfl(static_cast<uint32_t>(st.available)) == rl(static_cast<uint32_t>(my_avail), s) &&
                               rfl(static_cast<uint32_t>(st.read_position)) + rfl(static_cast<uint32_t>(st.available)) + a.in_frames <= kMirrorBufferSize;
                }
                if (on_track && ((lean_mask >> s) & 1ull)) {
                    // the run of lean calls from s: up to the first call that is not lean, or just behind the first whose successor's
                    // prediction does not continue it
                    const uint64_t not_lean = ~lean_mask >> s, not_succ = ~succ_mask >> s;
                    uint32_t e = s + (not_lean ? static_cast<uint32_t>(__builtin_ctzll(not_lean)) : 64u - s);
                    const uint32_t e2 = s + (not_succ ? static_cast<uint32_t>(__builtin_ctzll(not_succ)) + 1u : 64u - s);
                    e = e < e2 ? e : e2;
                    e = e < nc ? e : nc;
                    if (e >= s + 8u && parallel_run(s, e)) {
                        s = e - 1;
                        continue;
                    }
                }
            }
            if constexpr (!PCHAIN) {
            if (on_track && ((fast_mask >> s) & 1ull)) {
                if (st_valid) {   // (the counters leave `st`)
                    pos = st.position;
                    sc = ChainScalars{rfl64(st.abs_out), rfl64(st.abs_consumed), rfl(static_cast<uint32_t>(st.read_position)),
                                      rfl(static_cast<uint32_t>(st.available))};
                    st_valid = false;
                }
                s = chain_fast_run_of(shape_d, s, nc, fast_mask, succ_mask, lane, my_n_total, my_cp, my_cpred, pos, sc, on_track, my_pos, lean_last_n,
                                      a.in_frames, ratio, bn);
                last_lean = true;
                if (s >= nc) break;
            }
            }
            const double pos0 = st_valid ? st.position : pos;
            my_pos = lane == s ? pos0 : my_pos;   // lane s keeps call s's start position
            if (!on_track) {   // (behind a call off the track: is the state where this call's prediction starts?)
                const uint64_t m0 = static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[0]), s)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[0] >> 32), s)) << 32);
                const uint64_t cc0 = static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[1]), s)) | (static_cast<uint64_t>(rl(static_cast<uint32_t>(mine[1] >> 32), s)) << 32);
                // (... with as many frames buffered as the lanes' closed form has it)
                on_track = rfl64(st.abs_out) == m0 && rfl64(st.abs_consumed) == cc0 &&
                           rfl(static_cast<uint32_t>(st.available)) == rl(static_cast<uint32_t>(my_avail), s) &&
                           rfl(static_cast<uint32_t>(st.read_position)) + rfl(static_cast<uint32_t>(st.available)) + a.in_frames <= kMirrorBufferSize;
            }
            if (on_track && ((lean_mask >> s) & 1ull)) {
                if (st_valid) {   // (the counters leave `st`)
                    pos = st.position;
                    sc = ChainScalars{rfl64(st.abs_out), rfl64(st.abs_consumed), rfl(static_cast<uint32_t>(st.read_position)),
                                      rfl(static_cast<uint32_t>(st.available))};
                    st_valid = false;
                }
                const uint32_t n_total = rl(my_n_total, s);
                const uint32_t ctl = rl(my_cp.ctl, s), n_last = rl(my_cp.n_last, s);
                const uint32_t cpred = rl(my_cpred, s);
                const uint32_t shape = (ctl >> 8) & 0xFu;
                const uint32_t cons = (ctl & 0xFFF000u) ? chain_step_of_shape<true>(shape, s, my_cp, pos, sc, a.in_frames, ratio, bn, n_total, ctl, n_last)
                                                        : chain_step_of_shape<false>(shape, s, my_cp, pos, sc, a.in_frames, ratio, bn, n_total, ctl, n_last);
                on_track = cons == cpred && ((succ_mask >> s) & 1ull);
                lean_last_n = n_total;     // (the run's totals follow from the counters at its end, see below)
                last_lean = true;
                continue;
            }
            // ---- off the track: round 4's path (every premise checked, or the plain state machine)
            uint32_t cur[14];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                cur[2 * i] = rl(static_cast<uint32_t>(mine[i]), s);
                cur[2 * i + 1] = rl(static_cast<uint32_t>(mine[i] >> 32), s);
            }
            MirrorPred pr;
            __builtin_memcpy(&pr, cur, sizeof pr);
            if (!st_valid) {   // the counters go back into `st` (next_int: a call on the track starts where its prediction says)
                st.position = pos;
                st.abs_out = sc.abs_out;
                st.abs_consumed = sc.abs_consumed;
                st.read_position = sc.read_position;
                st.available = sc.available;
                st.next_int = pr.ni_before;   // (on the track: = the previous call's ni_after)
                if (!on_track) {   // (the previous lean call left the track: its own end decides)
                    const uint64_t ph = sc.abs_out % st.den;
                    st.next_int = static_cast<uint32_t>(ph ? st.den - ph : 0);
                }
                st_valid = true;
            }
            double drift = 0.0;
            uint32_t cf = st.abs_out != pr.m0 ? kCallAhead : 0u;
            FirCallCounts c;
            if (!base.usable || !mirror_call_fast(st, a.in_frames, ls.out_cap_frames, pr, bn, c, [](uint32_t, uint32_t, double, double) {})) {
                const uint32_t ni = st.next_int;
                sink.rel = static_cast<uint32_t>(st.abs_out - k0 * rs.wrap_unit);   // (the call's first output, relative to the bitmap's start)
                sink.periodic = wraps_exist && st.periodic_ok != 0;
                c = mirror_call(st, a.in_frames, ls.out_cap_frames, sink);
                cf = kCallSlow | (ni < c.produced ? kCallHasInt : 0u);
                drift = st.drift;
            }
            on_track = false;   // (looked at again at the top of the next call)
            nonlean_mask |= 1ull << s;
            if (c.accepted != a.in_frames) flags |= kLsStatusPartialAccept;
            last_c0 = static_cast<uint32_t>(c.accepted) * C;
            last_c1 = static_cast<uint32_t>(c.produced) * C;
            const bool me = lane == s;   // lane s keeps call s's record and counts
            my_drift = me ? drift : my_drift;
            my_flags = me ? cf : my_flags;
            my_c0 = me ? last_c0 : my_c0;
            my_c1 = me ? last_c1 : my_c1;
            last_lean = false;
        }
        if (!st_valid) lean_ni_after = rl(static_cast<uint32_t>(mine[3] >> 32), nc - 1);   // (the latest call's ni_after, should the run end here)
        if (lane < nc) {
            const bool was_lean = ((nonlean_mask >> lane) & 1ull) == 0;
            recs[c0 + lane] = CallRec{my_pos, was_lean ? 0.0 : my_drift, was_lean ? kCallLean : my_flags, 0u};
            uint32_t* counts = a.counts + 2 * (static_cast<size_t>(c0 + lane) * a.n_streams + rs.caller);
            counts[0] = was_lean ? a.in_frames * C : my_c0;
            counts[1] = was_lean ? my_n_total * C : my_c1;
        }
    }
    if (!st_valid) {   // the run ended on the chain: the state goes back into `st`
        st.position = pos;
        st.abs_out = sc.abs_out;
        st.abs_consumed = sc.abs_consumed;
        st.read_position = sc.read_position;
        st.available = sc.available;
        st.next_int = lean_ni_after;
    }
    if (lane != 0) return;
    if (sink.overflow) flags |= kLsStatusRunOverflow;
    // the run's totals, from the counters it ends with: outputs, frames retired, frames accepted (= retired + what the
    // buffered frames grew by)
    const uint32_t n_out = static_cast<uint32_t>(st.abs_out - abs_out0);
    const uint32_t consumed = static_cast<uint32_t>(st.abs_consumed - abs_consumed0);
    const uint32_t accepted = static_cast<uint32_t>(st.abs_consumed + st.available - abs_consumed0 - hist_frames);
    if (last_lean) {
        last_c0 = a.in_frames * C;
        last_c1 = lean_last_n * C;
    }

    // the calls' outputs follow each other: behind what was appended before, or (no `append`) from the front of `out`
    const uint64_t cursor = a.append ? a.cursor_in[gs] : 0;
    a.cursor_out[gs] = cursor + static_cast<uint64_t>(n_out) * C;
    FirStreamDesc* d = a.descs + gs;
    d->in = ls.in + a.in_offset * C;
    d->hist = a.hist_parity ? ls.hist_alt : ls.hist;
    d->hist_next = a.hist_parity ? ls.hist : ls.hist_alt;
    d->out = ls.out + cursor;
    d->wrap_bits = bits;
    d->n_out = n_out;
    d->hist_frames = hist_frames;
    d->in_frames = accepted;
    d->tail_start = consumed;
    d->tail_frames = static_cast<uint32_t>(st.available);
    d->abs_out = abs_out0;
    d->abs_consumed = abs_consumed0;
    d->wrap_k0 = k0;
    a.states_out[gs] = st;
    // the last call's counts, where rsmp_fir_lockstep_counts looks for them
    a.last_counts[2 * gs] = last_c0;
    a.last_counts[2 * gs + 1] = last_c1;
    if (flags) a.status[gs] |= flags;
}

// K3 -- the outputs at integer positions (the row-1023 variant's bitmap, the stream's drift): one wave per stream, a
// lane per call replays its call's chain from the recorded start position (mirror_replay_wraps).  The drift the
// stream ends with is that of the last call that had such an output.
// (Waves per workgroup = per stream: one walks the run's chunks of 64 calls one after the other -- large batches: "a wave
// per chunk, four times the waves at 256 calls, was measured slower, 59 against 33 us per run of config 4" --; small batches,
// whose period is the planner's own latency, give every chunk of a round of kLsWrapWaves chunks a wave of its own and
// settle the stream's drift -- that of the LAST call with an output at an integer position -- through LDS: 32 -> ~10 us
// per run at 128 streams x 256 calls.)
constexpr uint32_t kLsWrapWaves = 16;   // (at most: a run of 256 calls has four chunks, a bulk launch of 4096 calls sixty-four)
__global__ __launch_bounds__(64 * kLsWrapWaves) void fir_lockstep_wraps_kernel(LsRunArgs a) {
    __shared__ double s_drift[kLsWrapWaves];
    __shared__ uint32_t s_chunk[kLsWrapWaves], s_flags[kLsWrapWaves];
    const uint32_t gs = blockIdx.x, lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = blockDim.x >> 6;
    const LsRunStream rs = a.rs[gs];
    const FirMirrorState st0 = a.states_before[gs];   // the state before the run
    const MirrorRunBase base = mirror_run_base(st0, a.in_frames, a.k);
    const MirrorBinades bn = mirror_binades(st0.ratio, base.e0);
    const bool wraps_exist = rs.wrap_unit == rs.den && st0.periodic_ok != 0;
    uint32_t* bits = a.wrap_bits + static_cast<size_t>(gs) * a.wrap_words;
    const uint64_t k0 = st0.abs_out / rs.wrap_unit;
    const MirrorPred* preds = a.preds + static_cast<size_t>(gs) * a.k;
    const CallRec* recs = reinterpret_cast<const CallRec*>(a.call_recs) + static_cast<size_t>(gs) * a.k;
    double drift = st0.drift;
    uint32_t have_chunk = 0;   // 1 + the index of this wave's last chunk with an output at an integer position
    bool aperiodic = false, overflow = false, unchecked = false;
    for (uint32_t c0 = 64u * wave, chunk = wave; c0 < a.k; c0 += 64u * n_waves, chunk += n_waves) {
        const uint32_t c = c0 + lane;
        bool has_int = false;
        double my_drift = 0.0;
        if (c < a.k) {
            const CallRec rec = recs[c];
            if (rec.flags & kCallSlow) {
                has_int = (rec.flags & kCallHasInt) != 0;
                my_drift = rec.drift;
            } else if (wraps_exist || (rec.flags & kCallLean)) {
                const MirrorPred pr = preds[c];
                FirMirrorState st = st0;   // the call's start state, as far as the replay looks at it
                st.read_position = 0;
                st.abs_out = pr.m0 + ((rec.flags & kCallAhead) ? 1u : 0u);
                st.abs_consumed = pr.c0;
                st.available = st0.abs_consumed + st0.available + static_cast<uint64_t>(c) * a.in_frames - pr.c0;
                st.position = rec.pos;
                RunSink sink{bits, static_cast<uint32_t>(st.abs_out - k0 * rs.wrap_unit), rs.den, a.wrap_words * 32u, wraps_exist, false, true};
                bool checked = true;
                has_int = mirror_replay_wraps(st, a.in_frames, pr, bn, sink, &checked) && wraps_exist;
                unchecked = unchecked || !checked;   // (the chain took this call without checks: they were made here)
                my_drift = st.drift;
                aperiodic = aperiodic || st.periodic_ok == 0;
                overflow = overflow || sink.overflow;
            }
        }
        // the stream ends with the drift of its LAST call that had an output at an integer position
        const unsigned long long m = __ballot(has_int);
        if (m) {
            const int last = 63 - __builtin_clzll(m);
            drift = __shfl(my_drift, last, 64);
            have_chunk = chunk + 1;
        }
    }
    aperiodic = __any(aperiodic);
    overflow = __any(overflow);
    unchecked = __any(unchecked);
    if (lane == 0) {
        s_drift[wave] = drift;
        s_chunk[wave] = have_chunk;
        s_flags[wave] = (aperiodic ? 1u : 0u) | (overflow ? 2u : 0u) | (unchecked ? 4u : 0u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool have = false;
        uint32_t best = 0, fl = 0;
        for (uint32_t w = 0; w < n_waves; ++w) {
            fl |= s_flags[w];
            if (s_chunk[w] > best) {
                best = s_chunk[w];
                drift = s_drift[w];
                have = true;
            }
        }
        aperiodic = (fl & 1u) != 0;
        overflow = (fl & 2u) != 0;
        unchecked = (fl & 4u) != 0;
        // the class tables of the run's descriptor follow the stream's drift (LsRunStream; written here and not by the
        // chain kernel: four more values alive across its loop were 28 more spilled registers, 10 % of its time)
        FirStreamDesc* d = a.descs + gs;
        d->class_coef = rs.class_coef;
        d->class_wrap_coef = rs.class_wrap_coef;
        d->class_meta = rs.class_meta;
        d->drift = rs.drift;
        if (have) a.states_out[gs].drift = drift;
        if (aperiodic) a.states_out[gs].periodic_ok = 0;
        uint32_t flags = (overflow ? kLsStatusRunOverflow : 0u) | (unchecked ? kLsStatusPlannerCheck : 0u);
        if (a.states_out[gs].periodic_ok == 0 || aperiodic) flags |= kLsStatusAperiodic;
        if (flags) a.status[gs] |= flags;
    }
}

// a loop of steps as a run: the step's counts (internal order, 64-bit) to row s of the run's [k][n] table
__global__ __launch_bounds__(256) void fir_lockstep_gather_counts_kernel(const uint64_t* last_counts, const LsRunStream* rs,
                                                                        uint32_t* counts, uint32_t n) {
    const uint32_t gs = blockIdx.x * 256u + threadIdx.x;
    if (gs >= n) return;
    const uint32_t i = rs[gs].caller;
    counts[2 * i] = static_cast<uint32_t>(last_counts[2 * gs]);
    counts[2 * i + 1] = static_cast<uint32_t>(last_counts[2 * gs + 1]);
}

__global__ __launch_bounds__(256) void fir_lockstep_commit_kernel(LsCommitArgs a) {
    const uint32_t gs = blockIdx.x * 256u + threadIdx.x;
    if (gs >= a.n_streams) return;
    a.states[gs] = a.sp_states[gs];
    a.cursor[gs] = a.sp_cursor[gs];
    a.last_counts[2 * gs] = a.sp_last_counts[2 * gs];
    a.last_counts[2 * gs + 1] = a.sp_last_counts[2 * gs + 1];
    const uint32_t f = a.sp_status[gs];
    if (f) a.status[gs] |= f;
}

__global__ __launch_bounds__(64) void fir_lockstep_gather_drift_kernel(const FirMirrorState* states, const uint32_t* reps, double* out, uint32_t n) {
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    if (c < n) out[c] = states[reps[c]].drift;
}

// (fir_lockstep.h: launch_fir_lockstep_probe_wait / _set)
__global__ __launch_bounds__(64) void fir_lockstep_probe_wait_kernel(uint32_t* flag, uint32_t token, uint32_t timeout_ticks, uint32_t* result) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t seen = 0;
    for (;;) {
        seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == token ? 1u : 0u;
        if (seen || __builtin_amdgcn_s_memrealtime() - t0 >= timeout_ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
    if (threadIdx.x == 0) __hip_atomic_store(result, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ __launch_bounds__(64) void fir_lockstep_probe_set_kernel(uint32_t* flag, uint32_t token) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void fir_lockstep_patch_tables_kernel(LsPatchArgs a) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t < a.n_groups) {
        const uint32_t cls = a.groups[t].pad0;
        for (uint32_t i = 0; i < a.n_patches; ++i)
            if ((a.p[i].flags & 1u) && a.p[i].cls == cls && a.groups[t].periodic) {
                a.groups[t].class_coef = a.p[i].step_coef;
                a.groups[t].class_meta = a.p[i].step_meta;
            }
    } else if (a.rs && t - a.n_groups < a.n_streams) {
        const uint32_t gs = t - a.n_groups;
        for (uint32_t i = 0; i < a.n_patches; ++i)
            if ((a.p[i].flags & 2u) && gs - a.p[i].first < a.p[i].count) {
                a.rs[gs].class_coef = a.p[i].run_coef;
                a.rs[gs].class_wrap_coef = a.p[i].run_wrap_coef;
                a.rs[gs].class_meta = a.p[i].run_meta;
                a.rs[gs].drift = a.p[i].drift;
            }
    }
}

}  // namespace

hipError_t launch_fir_lockstep_gather_drift(const FirMirrorState* states, const uint32_t* reps, double* out, uint32_t n,
                                            hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fir_lockstep_gather_drift_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, states, reps, out, n);
    return hipGetLastError();
}

hipError_t launch_fir_lockstep_probe_wait(uint32_t* flag, uint32_t token, uint32_t timeout_ticks, uint32_t* result, hipStream_t stream) {
    hipLaunchKernelGGL(fir_lockstep_probe_wait_kernel, dim3(1), dim3(64), 0, stream, flag, token, timeout_ticks, result);
    return hipGetLastError();
}

hipError_t launch_fir_lockstep_probe_set(uint32_t* flag, uint32_t token, hipStream_t stream) {
    hipLaunchKernelGGL(fir_lockstep_probe_set_kernel, dim3(1), dim3(64), 0, stream, flag, token);
    return hipGetLastError();
}

hipError_t launch_fir_lockstep_patch_tables(const LsPatchArgs& args, hipStream_t stream) {
    if (args.n_patches == 0) return hipSuccess;
    const uint32_t threads = args.n_groups + (args.rs ? args.n_streams : 0u);
    hipLaunchKernelGGL(fir_lockstep_patch_tables_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, args);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void fir_lockstep_rebase_kernel(FirStreamDesc* descs, const LockstepStream* streams, const LsRunStream* rs,
                                                                  uint64_t in_offset, uint32_t n) {
    const uint32_t gs = blockIdx.x * 256u + threadIdx.x;
    if (gs >= n) return;
    descs[gs].in = streams[gs].in + in_offset * rs[gs].channels;   // (as the chain kernel sets them: d->in = ls.in + in_offset * C)
    descs[gs].out = streams[gs].out;
}

hipError_t launch_fir_lockstep_rebase(FirStreamDesc* descs, const LockstepStream* streams, const LsRunStream* rs, uint64_t in_offset,
                                      uint32_t n_streams, hipStream_t stream) {
    hipLaunchKernelGGL(fir_lockstep_rebase_kernel, dim3((n_streams + 255) / 256), dim3(256), 0, stream, descs, streams, rs, in_offset, n_streams);
    return hipGetLastError();
}

hipError_t launch_fir_lockstep_commit(const LsCommitArgs& args, hipStream_t stream) {
    hipLaunchKernelGGL(fir_lockstep_commit_kernel, dim3((args.n_streams + 255) / 256), dim3(256), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_fir_lockstep_gather_counts(const uint64_t* last_counts, const LsRunStream* rs, uint32_t* counts, uint32_t n,
                                             hipStream_t stream) {
    hipLaunchKernelGGL(fir_lockstep_gather_counts_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, last_counts, rs, counts, n);
    return hipGetLastError();
}

uint32_t lockstep_plan_pack(size_t n_streams) {
    static const uint32_t knob = [] { const char* e = rsmp::knob("RSMP_LS_PACK"); const int v = e ? atoi(e) : 0; return v == 1 || v == 2 || v == 4 ? static_cast<uint32_t>(v) : 0u; }();
    if (n_streams >= kLsPlanPackBelow) return 1u;
    return knob ? knob : kLsPlanPack;
}

uint32_t lockstep_replay_cus(size_t n_streams, uint32_t k) {
    if (lockstep_plan_pack(n_streams) <= 1) return 0;
    const uint32_t chunks = (k + 63) / 64, waves = std::min<uint32_t>(kLsWrapWaves, chunks);
    return static_cast<uint32_t>((n_streams * waves + 15) / 16);
}

hipError_t launch_fir_lockstep_plan(const LsRunArgs& args_in, hipStream_t stream, int parts, const LsCommitArgs* commit, hipEvent_t k1_done) {
    if (args_in.n_streams == 0 || args_in.k == 0) return hipSuccess;
    static const bool pchain = [] { const char* e = rsmp::knob("RSMP_LS_PCHAIN"); return !e || atoi(e) != 0; }();
    LsRunArgs args = args_in;
    args.parallel_chain = pchain ? 1u : 0u;
    const uint32_t blocks_per_stream = (args.k + 255) / 256;
    if (parts & 1) {
        LsCommitArgs cm{};
        if (commit) cm = *commit;
        if (k1_done)
            hipExtLaunchKernelGGL(fir_lockstep_predict_kernel, dim3(blocks_per_stream * args.n_streams), dim3(256), 0, stream, nullptr, k1_done, 0, args,
                                  blocks_per_stream, cm);
        else
            hipLaunchKernelGGL(fir_lockstep_predict_kernel, dim3(blocks_per_stream * args.n_streams), dim3(256), 0, stream, args, blocks_per_stream, cm);
    }
    if (parts & 2) {
        const uint32_t pack = lockstep_plan_pack(args.n_streams);
        if (pchain) hipLaunchKernelGGL(fir_lockstep_chain_kernel<true>, dim3((args.n_streams + pack - 1) / pack), dim3(64 * pack), 0, stream, args);
        else hipLaunchKernelGGL(fir_lockstep_chain_kernel<false>, dim3((args.n_streams + pack - 1) / pack), dim3(64 * pack), 0, stream, args);
        // (the replay: a wave per chunk of 64 calls for small batches, one wave per stream otherwise)
        const uint32_t chunks = (args.k + 63) / 64;
        const uint32_t wwaves = pack > 1 ? std::min<uint32_t>(kLsWrapWaves, chunks) : 1u;
        hipLaunchKernelGGL(fir_lockstep_wraps_kernel, dim3(args.n_streams), dim3(64 * wwaves), 0, stream, args);
    }
    return hipGetLastError();
}

}  // namespace rsmp
