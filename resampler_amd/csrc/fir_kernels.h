// fir_kernels.h -- device-side work descriptors and launch wrappers of the FIR path.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/resampler_amd.h"
#include "fir_nonfinite.h"

namespace rsmp {

// One stream's share of a launch.  A launch covers `n_out` output frames of every stream in the
// batch; the input of a stream is the virtual concatenation [hist | in]: `hist_frames` frames
// that were already buffered inside the resampler when the launch started (the reference's
// input_buffers[read_position .. read_position+available_frames], resampler_fir.rs:187-191)
// followed by the frames accepted during the launch.  Everything is interleaved f32.
struct FirStreamDesc {
    const float* in;               // frames accepted in this launch (device)
    const float* hist;             // frames buffered before the launch (device)
    float* out;                    // output frames (device)
    const float* coeffs;           // [1024][taps] polyphase table (device)
    const rsmp_fir_segment* segs;  // exact position runs, sorted by out_start (device)
    const uint32_t* tile_seg;      // index of the run containing output frame tile*kFirTile
    float* hist_next;              // receives the frames still buffered after the launch
    const float* class_coef;       // periodic kernel: class table [tile][row_len][8] (device)
    const float* class_wrap_coef;  // periodic kernel: wrap-variant coefficients [tile][row_len]
    const void* class_meta;        // periodic kernel: TileMeta[tile]
    const uint32_t* wrap_bits;     // periodic kernel: bitmap of outputs taking the wrap variant
    const uint32_t* wraps;         // fix-up kernel: outputs needing the row-1023 variant
    uint32_t n_out;                // output frames in this launch
    uint32_t n_segs;
    uint32_t hist_frames;
    uint32_t in_frames;            // frames accepted in this launch
    uint32_t tail_start;           // first frame of [hist|in] that stays buffered afterwards
    uint32_t tail_frames;          // how many stay buffered
    uint32_t n_wraps;
    uint32_t channels;
    uint32_t taps;
    // periodic kernel: in_hz/out_hz = num/den reduced, absolute counters at launch start
    uint32_t num, den;
    uint64_t abs_out;              // output frames produced since reset, before this launch
    uint64_t abs_consumed;         // input frames retired since reset, before this launch
    uint64_t wrap_k0;              // wrap_bits bit K <-> absolute output (wrap_k0 + K) * den
    double drift;                  // periodic kernels: the f64 drift the stream's class table was built for
    // 0: `in` is interleaved f32.  16 / 24 / 32: `in` is a WAV file's little-endian PCM of that width, `channels` samples
    // a frame, converted where it is read as resample/src/main.rs:128-137 converts it (fir_pcm_value below); `hist`
    // -- the frames an earlier launch left buffered -- is f32 always.
    uint32_t in_bits;
    uint32_t pad_bits;
};

// One sample of a stream's `in`, index `i` in values (frame * channels + channel).
// main.rs:131: `sample as f32 / (1 << (bits - 1)) as f32`; the literal is an i32, so 32-bit files divide by -2^31.
__device__ __forceinline__ float fir_pcm_value(const void* in, uint32_t bits, size_t i) {
    if (bits == 16) return static_cast<float>(static_cast<const int16_t*>(in)[i]) * (1.0f / 32768.0f);
    if (bits == 32) return static_cast<float>(static_cast<const int32_t*>(in)[i]) * (-1.0f / 2147483648.0f);
    const uint8_t* p = static_cast<const uint8_t*>(in) + 3 * i;
    const uint32_t u = static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) | (static_cast<uint32_t>(p[2]) << 16);
    return static_cast<float>(static_cast<int32_t>(u << 8) >> 8) * (1.0f / 8388608.0f);
}
__device__ __forceinline__ float fir_in_value(const FirStreamDesc& d, size_t i) {
    return d.in_bits == 0 ? d.in[i] : fir_pcm_value(d.in, d.in_bits, i);
}

constexpr uint32_t kFirTile = 32;   // output frames per workgroup tile (generic kernel): one pass of 32 x 8 lanes

// Generic kernel: any ratio, reference-form two-row interpolation; grid = (max tiles, streams).
hipError_t launch_fir_generic(const FirStreamDesc* d_descs, uint32_t n_streams, uint32_t max_out,
                              uint32_t max_channels, hipStream_t stream, bool fuse_tail = false);
// The same arithmetic for LONG launches (fir_generic_bulk.hip): tiles of up to 4096 output frames per workgroup, their outputs sorted
// by phase row, the tile's input window in LDS.  hipErrorNotSupported where no tile fits the LDS (fir_generic_bulk_tile == 0);
// does not copy the tails (launch_fir_tail_copy).
uint32_t fir_generic_bulk_tile(uint32_t max_channels, uint32_t max_taps, double max_ratio);
// (uniform_channels: 1 / 2 = every stream has that many channels -- the builds for them --, anything else = any counts)
hipError_t launch_fir_generic_bulk(const FirStreamDesc* d_descs, uint32_t n_streams, uint32_t max_out, uint32_t max_channels,
                                   uint32_t max_taps, double max_ratio, hipStream_t stream, uint32_t uniform_channels = 0,
                                   uint32_t uniform_taps = 0);   // (uniform_taps: 128 = every stream has 128 taps)
constexpr uint32_t kFirBulkMinOut = 8192;   // launches whose longest generic stream produces fewer frames keep fir_generic_kernel
// Re-evaluates, in the reference's two-row form, the output chunks a periodic launch marked as non-finite
// (fir_nonfinite.h); exits at once when the launch marked nothing.
hipError_t launch_fir_repair(const FirStreamDesc* d_descs, uint32_t n_streams, const NfArgs& nf, hipStream_t stream);
// The same for up to eight groups of streams (each with its own marks) in one launch.
struct RepairJob {
    const FirStreamDesc* d_descs;
    uint32_t n_streams;
    NfArgs nf;
};
constexpr uint32_t kMaxRepairJobs = 8;
// (tail_descs: the launch also copies the tails of these n_tail streams -- what launch_fir_tail_copy does -- in one more row
// of its grid)
// (done / done_attached: an event the LAST of these launches completes itself -- hipExtLaunchKernel's stop event, no packet of its
// own behind the kernel as hipEventRecord puts there; *done_attached = false: there was no such launch, record it yourself)
hipError_t launch_fir_repair_multi(const RepairJob* jobs, size_t n_jobs, hipStream_t stream, const FirStreamDesc* tail_descs = nullptr,
                                   uint32_t n_tail = 0, uint32_t max_tail_values = 0, hipEvent_t done = nullptr, bool* done_attached = nullptr);
// Copies the still-buffered tail of [hist|in] into hist_next; grid = (blocks, streams).
hipError_t launch_fir_tail_copy(const FirStreamDesc* d_descs, uint32_t n_streams,
                                uint32_t max_tail_values, hipStream_t stream);

}  // namespace rsmp
