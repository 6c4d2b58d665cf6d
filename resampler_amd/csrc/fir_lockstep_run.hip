// fir_lockstep_run.hip -- the planner of rsmp_fir_lockstep_run: k consecutive resample() calls per stream
// (src/resampler_fir.rs:509-621, driven like resample/src/main.rs:226-254) planned ON THE DEVICE, so that the
// calls of a whole run become ONE launch of the bulk kernels per rate pair instead of k launches of the
// one-call-per-stream kernel.
//
// One lane per stream replays the reference's control flow for the k calls (fir_mirror_core.h: the f64 position
// recurrence in closed form) and leaves, in HBM,
//   * the per-call (consumed, produced) counts of every stream: [k][n] pairs;
//   * the run's stream descriptor (FirStreamDesc, fir_kernels.h) exactly as the host planner of the bulk entry
//     points (fir_api.cpp, launch_jobs) would have built it: outputs / frames accepted / frames retired by the
//     run, the absolute counters it starts from, where its buffered frames are and where the tail goes;
//   * the bitmap of outputs that take the row-1023 variant (position just below an integer, :562-564);
//   * the stream's state after the run.
// The bulk kernels (fir_split.hip, fir_periodic.hip) then read those descriptors: nothing of a run passes
// through the host, whatever states the streams are in.
#include "fir_lockstep.h"

#include "common.h"

namespace rsmp {

namespace {

struct RunSink {             // mirror_call sink: the wrapped outputs go straight into the run's bitmap
    uint32_t* bits;
    uint32_t rel;            // (absolute index of the call's first output) - wrap_k0 * den
    uint32_t den, n_bits;
    bool periodic, overflow;
    __host__ __device__ bool want_wraps() const { return periodic; }
    __host__ __device__ void run(uint64_t, uint64_t, double, double) {}
    __host__ __device__ void wrap(uint64_t index) {
        const uint32_t K = (rel + static_cast<uint32_t>(index)) / den;
        if (K < n_bits) bits[K >> 5] |= 1u << (K & 31);
        else overflow = true;
    }
};

__global__ __launch_bounds__(64) void fir_lockstep_plan_kernel(LsRunArgs a) {
    const uint32_t first = a.waves[2 * blockIdx.x], count = a.waves[2 * blockIdx.x + 1];
    if (threadIdx.x >= count) return;
    const uint32_t gs = first + threadIdx.x;
    const LockstepStream ls = a.streams[gs];
    const LsRunStream rs = a.rs[gs];
    FirMirrorState st = a.states_in[gs];
    FirStreamDesc* d = a.descs + gs;
    uint32_t* bits = a.wrap_bits + static_cast<size_t>(gs) * a.wrap_words;
    for (uint32_t w = 0; w < a.wrap_words; ++w) bits[w] = 0;

    const uint32_t C = rs.channels;
    const uint64_t abs_out0 = st.abs_out, abs_consumed0 = st.abs_consumed;
    const uint64_t k0 = abs_out0 / rs.wrap_unit;
    const uint32_t hist_frames = static_cast<uint32_t>(st.available);
    RunSink sink{bits, static_cast<uint32_t>(abs_out0 - k0 * rs.wrap_unit), rs.den, a.wrap_words * 32u,
                 rs.wrap_unit == rs.den, false};   // (a super period of an exact ratio: no output ever wraps)
    uint32_t n_out = 0, accepted = 0, consumed = 0, flags = 0;
    uint32_t* counts = a.counts + 2 * static_cast<size_t>(rs.caller);
    for (uint32_t s = 0; s < a.k; ++s) {
        sink.periodic = rs.wrap_unit == rs.den && st.periodic_ok != 0;
        const FirCallCounts c = mirror_call(st, a.in_frames, ls.out_cap_frames, sink);
        if (c.accepted != a.in_frames) flags |= kLsStatusPartialAccept;
        counts[0] = static_cast<uint32_t>(c.accepted) * C;
        counts[1] = static_cast<uint32_t>(c.produced) * C;
        counts += 2 * static_cast<size_t>(a.n_streams);
        n_out += static_cast<uint32_t>(c.produced);
        accepted += static_cast<uint32_t>(c.accepted);
        consumed += static_cast<uint32_t>(c.consumed);
        sink.rel += static_cast<uint32_t>(c.produced);
    }
    if (st.periodic_ok == 0) flags |= kLsStatusAperiodic;
    if (sink.overflow) flags |= kLsStatusRunOverflow;

    uint64_t cursor = 0;
    if (a.append) {
        cursor = a.cursor_in[gs];
        a.cursor_out[gs] = cursor + static_cast<uint64_t>(n_out) * C;
    }
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
    counts -= 2 * static_cast<size_t>(a.n_streams);
    a.last_counts[2 * gs] = counts[0];
    a.last_counts[2 * gs + 1] = counts[1];
    if (flags) a.status[gs] |= flags;
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

}  // namespace

hipError_t launch_fir_lockstep_gather_counts(const uint64_t* last_counts, const LsRunStream* rs, uint32_t* counts, uint32_t n,
                                             hipStream_t stream) {
    hipLaunchKernelGGL(fir_lockstep_gather_counts_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, last_counts, rs, counts, n);
    return hipGetLastError();
}

hipError_t launch_fir_lockstep_plan(const LsRunArgs& args, uint32_t n_waves, hipStream_t stream) {
    if (n_waves == 0) return hipSuccess;
    hipLaunchKernelGGL(fir_lockstep_plan_kernel, dim3(n_waves), dim3(64), 0, stream, args);
    return hipGetLastError();
}

}  // namespace rsmp
