// fft_plan.h -- host-side planning of the ResamplerFft path: block sizes and radix factors per
// rate pair (reference src/fft/planner.rs), factor merging/ordering (src/fft/optimizer.rs), the
// N/2-trick stage list (src/fft/radix_fft.rs:222-246) and every twiddle table the kernel
// consumes, computed in f64 and rounded to f32 exactly as the reference does
// (radix_fft.rs:251-258, :273-399).  Pure host code, no HIP.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace rsmp {

struct Complex32 { float re, im; };   // fft/mod.rs:11-16, repr(C)

// ConversionConfig::from_sample_rates + scale_for_throughput (planner.rs:35-245).  Rates must be
// members of the SampleRate enum.  Returns false otherwise.
bool fft_conversion_config(uint32_t in_hz, uint32_t out_hz, bool scale_for_throughput,
                           size_t* fft_size_in, std::vector<int>* factors_in,
                           size_t* fft_size_out, std::vector<int>* factors_out);

// optimize_factors (optimizer.rs:6-64).
std::vector<int> optimize_factors(std::vector<int> factors);

// One direction of the real FFT of (even) length n = product(factors): RadixFFT::new
// (radix_fft.rs:105-183).
struct RealFftPlan {
    size_t n = 0, n2 = 0;
    std::vector<int> stages;               // n2-point complex Stockham stage radices, in order
    // Stage twiddles, unique values only: stage s > 0 with stride p and radix r holds p*(r-1)
    // entries w[col*(r-1) + (k-1)] = exp(-2*pi*i*col*k/(p*r)); stage 0 has none.
    std::vector<Complex32> stage_twiddles;
    std::vector<uint32_t> stage_twiddle_offset;   // per stage, into stage_twiddles
    // Real<->complex twiddles for k = 1 .. n/4-1: forward x0.5 (:377-386), inverse conjugated
    // (:388-397).
    std::vector<Complex32> rc_twiddles;
    bool ok = false;
};
RealFftPlan make_real_fft_plan(const std::vector<int>& factors, bool inverse);

// Everything ResamplerFft::new derives for a rate pair (resampler_fft.rs:75-119, :338-383),
// except the filter spectrum, which is produced on the device by the same forward transform the
// resampler uses (fft_kernels.hip) from `filter_time`.
struct FftResamplerPlan {
    size_t fft_in = 0, fft_out = 0;     // frames per block in / out
    RealFftPlan forward, inverse;       // lengths 2*fft_in and 2*fft_out
    size_t new_length = 0;              // bins multiplied by the filter (resampler_fft.rs:396-399)
    std::vector<float> filter_time;     // fft_in windowed-sinc taps / (2*fft_in), zero padded to 2*fft_in
    bool ok = false;
};
FftResamplerPlan make_fft_resampler_plan(uint32_t in_hz, uint32_t out_hz);

}  // namespace rsmp
