// fft_kernels.h -- device-side descriptors and launch wrappers of the FFT overlap-add path.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rsmp {

constexpr int kMaxFftStages = 8;

// Device image of one FftResamplerPlan (fft_plan.h).  All pointers are HBM pointers.
struct FftPlanDev {
    uint32_t fft_in, fft_out;            // frames per block = complex FFT lengths (N/2 trick)
    uint32_t n_stages_f, n_stages_i;
    uint32_t radix_f[kMaxFftStages], radix_i[kMaxFftStages];
    uint32_t tw_off_f[kMaxFftStages], tw_off_i[kMaxFftStages];
    const float2* tw_f;                  // forward stage twiddles (unique per column)
    const float2* tw_i;                  // inverse stage twiddles
    const float2* rc_f;                  // real-FFT expansion twiddles (x0.5)
    const float2* rc_i;                  // real-FFT reduction twiddles (conjugated)
    uint32_t n_rc_f, n_rc_i;
    const float2* filter;                // fft_in + 1 bins
    uint32_t new_length;                 // bins multiplied by the filter, the rest are zero
    uint32_t lds_complex;                // max(fft_in, fft_out) + 1
    const float2* chirp_f;               // exp(-2 pi i n / (2 fft_in)), n < fft_in   (fft_pair.hip: two channels as one complex signal)
    const float2* chirp_i;               // exp(-2 pi i n / (2 fft_out)), n < fft_out
};

// One stream's share of a launch: n_blocks consecutive blocks of all its channels.
struct FftStreamDesc {
    const float* in;      // interleaved, n_blocks * fft_in * channels values
    float* out;           // interleaved, n_blocks * fft_out * channels values
    const float* overlap; // stream state before the launch: [channels][fft_out] (resampler_fft.rs:51)
    float* overlap_next;  // receives the state after the launch (another buffer: the workgroup that reads
                          // the old state and the one that writes the new one are not ordered)
    uint32_t n_blocks;
    uint32_t channels;
    uint32_t in_bits;     // 0: `in` is f32; 16 / 24 / 32: `in` is little-endian PCM of that width (two-channel streams on the wave kernel only)
    uint32_t pad;
};

// Blocks a workgroup walks in sequence (carrying the overlap on chip); the first block of a run
// that does not start the launch recomputes its predecessor as halo.
constexpr uint32_t kFftRun = 16;

// pcm_bits != 0: every stream's `in` is PCM of that width (FftStreamDesc::in_bits); hipErrorNotSupported where no kernel
// that reads PCM serves the plan / channel count.
hipError_t launch_fft_ola(const FftPlanDev& plan, const FftStreamDesc* d_descs, uint32_t n_streams,
                          uint32_t max_blocks, uint32_t max_channels, uint32_t min_channels,
                          hipStream_t stream, uint32_t pcm_bits = 0);
// Wave-per-transform build (fft_wave.hip) for the plans it is instantiated for; hipErrorNotSupported
// otherwise (launch_fft_ola then falls back to the workgroup kernels by itself).
hipError_t launch_fft_ola_wave(const FftPlanDev& plan, const FftStreamDesc* d_descs, uint32_t n_streams,
                               uint32_t max_blocks, uint32_t max_channels, uint32_t min_channels,
                               hipStream_t stream);
// One wave per two-channel stream, the frame (L, R) as the complex sample L + i R (fft_pair.hip); hipErrorNotSupported for
// the plans it is not instantiated for.
hipError_t launch_fft_ola_pair(const FftPlanDev& plan, const FftStreamDesc* d_descs, uint32_t n_streams, uint32_t max_blocks,
                               hipStream_t stream, uint32_t pcm_bits = 0);
// Whether this library's wave kernels are the operation-for-operation build (libresampler_amd_fftexact.so).
bool fft_wave_is_exact();
// filter_spectrum[0 .. fft_in] = forward real FFT of d_filter_time[0 .. 2*fft_in)
// (resampler_fft.rs:375-376); plan.filter is ignored.
hipError_t launch_fft_filter_spectrum(const FftPlanDev& plan, const float* d_filter_time,
                                      float2* d_filter_spectrum, hipStream_t stream);
size_t fft_big_lds_bytes(const FftPlanDev& plan);   // one-buffer kernel of the largest plans
size_t fft_ola_lds_bytes(const FftPlanDev& plan, uint32_t channels);

}  // namespace rsmp
