// fir_generic.hip -- generic polyphase FIR kernel for gfx950 (any rate pair, any call pattern).
//
// Replaces the reference's per-output-frame loop body (src/resampler_fir.rs:542-590) and its
// convolution leaf (src/fir/avx.rs:5-61).  One launch covers every output frame of every call
// the host mirror planned: each frame recovers its exact f64 position from its run descriptor
// (p = p0 + k*inc, one FMA, exact -- see fir_plan.h), derives input offset / phase rows / frac
// exactly as :544-565, and computes the dual-row dot product + per-lane lerp.
//
// Mapping: 8 lanes share one output frame (lane g owns float4 chunks g, g+8, ... of both phase
// rows, so one row read is a fully coalesced 512-byte burst at 128 taps), partial sums are lerped
// per lane like the reference's SIMD lanes (avx.rs:41-45) and reduced with three DPP butterfly
// steps.  A wave therefore produces 8 frames at a time; a 256-thread workgroup walks a tile of
// 256 consecutive output frames.  Samples come straight from [hist|in] in HBM/L2 (each lane
// reads 16-byte-contiguous frames); this kernel is the correctness backbone and the latency
// path -- the throughput path for rational rate pairs is fir_periodic.hip.
#include <algorithm>
#include <hip/hip_ext.h>

#include "fir_kernels.h"

namespace rsmp {

namespace {

constexpr int kLanesPerFrame = 8;
constexpr int kBlock = 256;
constexpr int kFramesPerPass = kBlock / kLanesPerFrame;  // 32 == kFirTile
constexpr uint32_t kChannelsPerBlock = 2;

__device__ __forceinline__ float group_sum8(float v) {
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// Sample `c` of virtual frame v of [hist|in].
__device__ __forceinline__ float load_sample(const FirStreamDesc& d, int64_t v, uint32_t c) {
    return v < static_cast<int64_t>(d.hist_frames)
               ? d.hist[static_cast<size_t>(v) * d.channels + c]
               : fir_in_value(d, static_cast<size_t>(v - d.hist_frames) * d.channels + c);
}

// fuse_tail: the first workgroup of every stream also copies the stream's still-buffered frames into hist_next
// (the job of fir_tail_copy_kernel: a second launch is a fifth of a streaming call's latency).
__global__ __launch_bounds__(kBlock) void fir_generic_kernel(const FirStreamDesc* __restrict__ descs, uint32_t fuse_tail) {
    const FirStreamDesc d = descs[blockIdx.y];
    const uint32_t tile = blockIdx.x;
    const uint32_t tile_first = tile * kFirTile;
    if (fuse_tail && tile == 0 && blockIdx.z == 0) {
        const size_t total = static_cast<size_t>(d.tail_frames) * d.channels;
        const size_t first = static_cast<size_t>(d.tail_start) * d.channels;
        const size_t hist_values = static_cast<size_t>(d.hist_frames) * d.channels;
        for (size_t i = threadIdx.x; i < total; i += kBlock) {
            const size_t src = first + i;
            d.hist_next[i] = src < hist_values ? d.hist[src] : fir_in_value(d, src - hist_values);
        }
    }
    if (tile_first >= d.n_out) return;

    const int g = threadIdx.x & (kLanesPerFrame - 1);
    const int slot = threadIdx.x / kLanesPerFrame;
    const uint32_t taps = d.taps;
    const uint32_t channels = d.channels;

    uint32_t seg_idx = d.tile_seg[tile];
    // blockIdx.z splits the channels (2 per workgroup): short calls with many channels are
    // latency bound, so they get parallel workgroups instead of a serial channel loop.
    const uint32_t c_begin = blockIdx.z * kChannelsPerBlock;
    if (c_begin >= channels) return;
    const uint32_t c_end = c_begin + kChannelsPerBlock < channels ? c_begin + kChannelsPerBlock : channels;

    for (uint32_t base = tile_first; base < tile_first + kFirTile; base += kFramesPerPass) {
        const uint32_t n = base + slot;
        const bool live = n < d.n_out;
        // Exact position of frame n: walk forward to its run (runs are sorted by out_start).
        double p0 = 0.0, inc = 0.0;
        int64_t in_base = 0;
        uint32_t k = 0;
        if (live) {
            uint32_t s = seg_idx;
            rsmp_fir_segment sg = d.segs[s];
            while (n >= sg.out_start + sg.count) sg = d.segs[++s];
            seg_idx = s;
            p0 = sg.p0;
            inc = sg.inc;
            in_base = sg.in_base;
            k = n - sg.out_start;
        }
        const double p = fma(static_cast<double>(k), inc, p0);
        const double fl = floor(p);                                   // :544
        const double fract = p - fl;                                  // :558 (p >= 0)
        double phase_f = fract * 1024.0;                              // :562
        phase_f = phase_f < 1023.0 ? phase_f : 1023.0;
        const uint32_t phase1 = static_cast<uint32_t>(phase_f);       // :563
        const uint32_t phase2 = phase1 + 1 < 1023u ? phase1 + 1 : 1023u;  // :564
        const float frac = static_cast<float>(phase_f - static_cast<double>(phase1));  // :565
        const int64_t v0 = in_base + static_cast<int64_t>(fl);
        const float4* __restrict__ row1 =
            reinterpret_cast<const float4*>(d.coeffs + static_cast<size_t>(phase1) * taps);
        const float4* __restrict__ row2 =
            reinterpret_cast<const float4*>(d.coeffs + static_cast<size_t>(phase2) * taps);
        const float one_minus_frac = 1.0f - frac;                     // avx.rs:42

        for (uint32_t c = c_begin; c < c_end; ++c) {
            float a1 = 0.0f, a2 = 0.0f;
            if (live) {
                for (uint32_t q = g; q < taps / 4; q += kLanesPerFrame) {
                    const float4 k1 = row1[q];
                    const float4 k2 = row2[q];
                    const int64_t v = v0 + 4 * static_cast<int64_t>(q);
                    const float x0 = load_sample(d, v, c);
                    const float x1 = load_sample(d, v + 1, c);
                    const float x2 = load_sample(d, v + 2, c);
                    const float x3 = load_sample(d, v + 3, c);
                    a1 = fmaf(k1.x, x0, a1); a2 = fmaf(k2.x, x0, a2);
                    a1 = fmaf(k1.y, x1, a1); a2 = fmaf(k2.y, x1, a2);
                    a1 = fmaf(k1.z, x2, a1); a2 = fmaf(k2.z, x2, a2);
                    a1 = fmaf(k1.w, x3, a1); a2 = fmaf(k2.w, x3, a2);
                }
            }
            // per-lane lerp, then horizontal sum (avx.rs:41-58)
            const float part = a1 * one_minus_frac + a2 * frac;
            const float y = group_sum8(part);
            if (live && g == 0) d.out[static_cast<size_t>(n) * channels + c] = y;
        }
    }
}

__global__ __launch_bounds__(kBlock) void fir_tail_copy_kernel(const FirStreamDesc* __restrict__ descs) {
    const FirStreamDesc d = descs[blockIdx.y];
    const size_t total = static_cast<size_t>(d.tail_frames) * d.channels;
    const size_t first = static_cast<size_t>(d.tail_start) * d.channels;
    const size_t hist_values = static_cast<size_t>(d.hist_frames) * d.channels;
    for (size_t i = blockIdx.x * static_cast<size_t>(kBlock) + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kBlock) {
        const size_t src = first + i;
        d.hist_next[i] = src < hist_values ? d.hist[src] : fir_in_value(d, src - hist_values);
    }
}

// Chunks of 1024 output frames that a periodic launch marked (fir_nonfinite.h), re-evaluated in the
// reference's form.  The position of output m is the exact rational m * num / den plus the drift the
// stream's class table was built for -- the same phase rows and frac the periodic kernel pre-mixed -- and
// outputs at an integer position take the previous frame and row 1023 where the wrap bitmap says so
// (resampler_fir.rs:544, :562-565).
__device__ __forceinline__ void fir_repair_body(const FirStreamDesc* __restrict__ descs, uint32_t n_streams, const NfArgs& nf) {
    if (__hip_atomic_load(nf.words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nf.tag) return;
    const int g = threadIdx.x & (kLanesPerFrame - 1);
    const uint32_t slot = threadIdx.x / kLanesPerFrame;
    const uint32_t total = n_streams * nf.chunks;
    for (uint32_t idx = blockIdx.x; idx < total; idx += gridDim.x) {
        const uint32_t word = __hip_atomic_load(nf.words + 1 + (idx >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!((word >> (idx & 31)) & 1u)) continue;
        const uint32_t s = idx / nf.chunks, chunk = idx - s * nf.chunks;
        const FirStreamDesc d = descs[s];
        const uint32_t n_begin = chunk << kNfChunkShift;
        const uint32_t n_end = n_begin + (1u << kNfChunkShift) < d.n_out ? n_begin + (1u << kNfChunkShift) : d.n_out;
        const uint32_t taps = d.taps, channels = d.channels;
        for (uint32_t base = n_begin; base < n_end; base += kFramesPerPass) {
            const uint32_t n = base + slot;
            const bool live = n < n_end;
            const uint64_t m = d.abs_out + (live ? n : n_begin);
            const uint64_t t = m * d.num;
            const uint64_t off = t / d.den, rem = t - off * d.den;
            double fract = static_cast<double>(rem) / static_cast<double>(d.den) + d.drift;
            bool wrapped = false;
            if (rem == 0) {
                fract = d.drift > 0.0 ? d.drift : 0.0;
                if (d.wrap_bits) {
                    const uint64_t K = m / d.den - d.wrap_k0;
                    wrapped = (d.wrap_bits[K >> 5] >> (K & 31)) & 1u;
                }
            }
            if (fract < 0.0) fract = 0.0;
            double phase_f = fract * 1024.0;                              // :562
            phase_f = phase_f < 1023.0 ? phase_f : 1023.0;
            uint32_t phase1 = static_cast<uint32_t>(phase_f);             // :563
            uint32_t phase2 = phase1 + 1 < 1023u ? phase1 + 1 : 1023u;    // :564
            float frac = static_cast<float>(phase_f - static_cast<double>(phase1));  // :565
            int64_t v0 = static_cast<int64_t>(off) - static_cast<int64_t>(d.abs_consumed);
            if (wrapped) {   // the f64 position was just below the integer: previous frame, row 1023, frac 0
                v0 -= 1;
                phase1 = phase2 = 1023u;
                frac = 0.0f;
            }
            const float4* __restrict__ row1 = reinterpret_cast<const float4*>(d.coeffs + static_cast<size_t>(phase1) * taps);
            const float4* __restrict__ row2 = reinterpret_cast<const float4*>(d.coeffs + static_cast<size_t>(phase2) * taps);
            const float one_minus_frac = 1.0f - frac;
            for (uint32_t c = 0; c < channels; ++c) {
                float a1 = 0.0f, a2 = 0.0f;
                if (live) {
                    for (uint32_t q = g; q < taps / 4; q += kLanesPerFrame) {
                        const float4 k1 = row1[q];
                        const float4 k2 = row2[q];
                        const int64_t v = v0 + 4 * static_cast<int64_t>(q);
                        const float x0 = load_sample(d, v, c);
                        const float x1 = load_sample(d, v + 1, c);
                        const float x2 = load_sample(d, v + 2, c);
                        const float x3 = load_sample(d, v + 3, c);
                        a1 = fmaf(k1.x, x0, a1); a2 = fmaf(k2.x, x0, a2);
                        a1 = fmaf(k1.y, x1, a1); a2 = fmaf(k2.y, x1, a2);
                        a1 = fmaf(k1.z, x2, a1); a2 = fmaf(k2.z, x2, a2);
                        a1 = fmaf(k1.w, x3, a1); a2 = fmaf(k2.w, x3, a2);
                    }
                }
                const float part = a1 * one_minus_frac + a2 * frac;
                const float y = group_sum8(part);
                if (live && g == 0) d.out[static_cast<size_t>(n) * channels + c] = y;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) (void)atomicAnd(nf.words + 1 + (idx >> 5), ~(1u << (idx & 31)));   // the bitmap is zero again for the next launch
    }
}
__global__ __launch_bounds__(kBlock) void fir_repair_kernel(const FirStreamDesc* __restrict__ descs,
                                                            uint32_t n_streams, NfArgs nf) {
    fir_repair_body(descs, n_streams, nf);
}
struct RepairMulti {
    const FirStreamDesc* descs[kMaxRepairJobs];
    uint32_t n_streams[kMaxRepairJobs];
    NfArgs nf[kMaxRepairJobs];
    // (a lock-step run's last launch: the tails of ALL its streams are copied by one more row of the grid -- a launch and its
    // gap less per run, fir_lockstep_api.cpp)
    const FirStreamDesc* tail_descs;
    uint32_t n_tail, n_jobs;
};
__global__ __launch_bounds__(kBlock) void fir_repair_multi_kernel(const RepairMulti m) {   // grid = (blocks, jobs [+ 1])
    if (blockIdx.y >= m.n_jobs) {   // the tail row: a workgroup per stream, round the row
        for (uint32_t s = blockIdx.x; s < m.n_tail; s += gridDim.x) {
            const FirStreamDesc d = m.tail_descs[s];
            const size_t total = static_cast<size_t>(d.tail_frames) * d.channels;
            const size_t first = static_cast<size_t>(d.tail_start) * d.channels;
            const size_t hist_values = static_cast<size_t>(d.hist_frames) * d.channels;
            for (size_t i = threadIdx.x; i < total; i += kBlock) {
                const size_t src = first + i;
                d.hist_next[i] = src < hist_values ? d.hist[src] : fir_in_value(d, src - hist_values);
            }
        }
        return;
    }
    fir_repair_body(m.descs[blockIdx.y], m.n_streams[blockIdx.y], m.nf[blockIdx.y]);
}

}  // namespace

hipError_t launch_fir_repair(const FirStreamDesc* d_descs, uint32_t n_streams, const NfArgs& nf, hipStream_t stream) {
    if (n_streams == 0 || !nf.words || nf.chunks == 0) return hipSuccess;
    const uint32_t total = n_streams * nf.chunks;
    hipLaunchKernelGGL(fir_repair_kernel, dim3(total < 512 ? total : 512), dim3(kBlock), 0, stream, d_descs, n_streams, nf);
    return hipGetLastError();
}

hipError_t launch_fir_repair_multi(const RepairJob* jobs, size_t n_jobs, hipStream_t stream, const FirStreamDesc* tail_descs,
                                   uint32_t n_tail, uint32_t max_tail_values, hipEvent_t done, bool* done_attached) {
    if (done_attached) *done_attached = false;
    bool tail_left = tail_descs != nullptr && n_tail != 0 && max_tail_values != 0;
    for (size_t j = 0; j < n_jobs;) {
        RepairMulti m{};
        uint32_t n = 0, max_total = 0;
        for (; j < n_jobs && n < kMaxRepairJobs; ++j) {
            if (jobs[j].n_streams == 0 || !jobs[j].nf.words || jobs[j].nf.chunks == 0) continue;
            m.descs[n] = jobs[j].d_descs;
            m.n_streams[n] = jobs[j].n_streams;
            m.nf[n] = jobs[j].nf;
            max_total = std::max(max_total, jobs[j].n_streams * jobs[j].nf.chunks);
            ++n;
        }
        if (n == 0) continue;
        m.n_jobs = n;
        const bool with_tail = tail_left && j >= n_jobs;   // (the last launch takes the tails along)
        if (with_tail) {
            m.tail_descs = tail_descs;
            m.n_tail = n_tail;
            tail_left = false;
            max_total = std::max(max_total, n_tail);
        }
        const dim3 grid(max_total < 512 ? max_total : 512, n + (with_tail ? 1u : 0u));
        if (done && done_attached && j >= n_jobs && !tail_left) {   // (the last launch of the lot: it completes `done` itself)
            hipExtLaunchKernelGGL(fir_repair_multi_kernel, grid, dim3(kBlock), 0, stream, nullptr, done, 0, m);
            *done_attached = true;
        } else {
            hipLaunchKernelGGL(fir_repair_multi_kernel, grid, dim3(kBlock), 0, stream, m);
        }
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    if (tail_left) return launch_fir_tail_copy(tail_descs, n_tail, max_tail_values, stream);   // (no repair job at all)
    return hipSuccess;
}

hipError_t launch_fir_generic(const FirStreamDesc* d_descs, uint32_t n_streams, uint32_t max_out,
                              uint32_t max_channels, hipStream_t stream, bool fuse_tail) {
    if (n_streams == 0 || max_out == 0) return hipSuccess;
    const uint32_t cz = (max_channels + kChannelsPerBlock - 1) / kChannelsPerBlock;
    const dim3 grid((max_out + kFirTile - 1) / kFirTile, n_streams, cz ? cz : 1);
    hipLaunchKernelGGL(fir_generic_kernel, grid, dim3(kBlock), 0, stream, d_descs, fuse_tail ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_fir_tail_copy(const FirStreamDesc* d_descs, uint32_t n_streams,
                                uint32_t max_tail_values, hipStream_t stream) {
    if (n_streams == 0 || max_tail_values == 0) return hipSuccess;
    uint32_t blocks = (max_tail_values + kBlock - 1) / kBlock;
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(fir_tail_copy_kernel, dim3(blocks, n_streams), dim3(kBlock), 0, stream,
                       d_descs);
    return hipGetLastError();
}

}  // namespace rsmp
