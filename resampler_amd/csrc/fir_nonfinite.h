// fir_nonfinite.h -- keeping the periodic kernels' inf / NaN behaviour equal to the reference's.
//
// The periodic kernels pre-mix the two phase rows, multiply a few samples next to an output's true window
// by zero padding coefficients (0 * inf = NaN) and, in the split kernel, cut operands into 16-bit planes
// (an infinity, or with two fp16 planes any sample of magnitude >= 16, does not survive that).  Finite audio
// never notices; a drop-in must still give the reference's answer for everything else.  So every store
// path checks the sums it is about to write (one add tree + one compare per lane and tile); a non-finite
// sum marks the 1024-output-frame chunk of its stream in a launch-wide bitmap, and a repair launch that
// follows the periodic launch re-evaluates the marked chunks in the reference's own form -- two phase
// rows, eight partial sums, per-lane lerp (src/fir/avx.rs:25-58) -- straight from [hist | in].
// With clean input the repair launch reads one word and exits.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

namespace rsmp {

constexpr uint32_t kNfChunkShift = 10;   // output frames per chunk: 1024

struct NfArgs {
    uint32_t* words;    // [0]: tag of the last launch that marked something; [1 ..]: bitmap, bit = stream * chunks + chunk
    uint32_t tag;       // this launch's tag (never 0)
    uint32_t chunks;    // chunks per stream
};

// `bad`: the sum over EVERY value this lane is about to store for its outputs n_first .. n_first + n_count - 1
// (launch-relative output frames of stream `stream`, of which [0, n_limit) exist) is not finite.  Every frame
// is tested: a non-finite sample spoils only about taps * out_hz / in_hz consecutive outputs, which with few
// taps and heavy down-sampling is fewer than a lane's frames (Sample8 at 192 -> 44.1 kHz: ~4), so a sampled
// test could step over a spoilt run.  (inf - inf and sums that overflow are also `bad`: the repair launch
// then recomputes a chunk that did not need it, which changes nothing.)  The marked range is still widened
// by the lane's frame count on both sides: the pre-mixed rows' zero padding (0 * inf) reaches that far.
__device__ __forceinline__ void nf_mark(const NfArgs& nf, bool bad, uint32_t stream, int32_t n_first, int32_t n_count,
                                        int32_t n_limit) {
    if (__builtin_expect(__any(bad), 0)) {
        if (bad && nf.words) {
            const int32_t a = n_first - n_count < 0 ? 0 : n_first - n_count;
            int32_t b = n_first + 2 * n_count - 1;
            if (b >= n_limit) b = n_limit - 1;
            if (a <= b) {
                for (int32_t c = a >> kNfChunkShift; c <= (b >> kNfChunkShift); ++c) {
                    const uint32_t bit = stream * nf.chunks + static_cast<uint32_t>(c);
                    (void)atomicOr(nf.words + 1 + (bit >> 5), 1u << (bit & 31));
                }
                (void)atomicExch(nf.words, nf.tag);
            }
        }
    }
}

__device__ __forceinline__ bool nf_is_bad(float sum) { return !(fabsf(sum) <= FLT_MAX); }
// the eight sums (four frames x two channels) of a matrix-core lane, as one value to test
template <typename V4>
__device__ __forceinline__ float nf_sum8(const V4& a, const V4& b) {
    return ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
}

}  // namespace rsmp
