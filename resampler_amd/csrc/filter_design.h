// filter_design.h -- host-side design-time tables: Kaiser-windowed sinc prototype, polyphase
// split, cutoff rule.  Runs once per configuration; results are cached process-wide and
// uploaded to HBM by the FIR / FFT front-ends.
//
// Follows the reference's src/window.rs (functions cited in filter_design.cpp) bit for bit: the
// GPU kernels consume exactly the numbers the reference CPU path consumes.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

namespace rsmp {

constexpr size_t kPhases = 1024;         // resampler_fir.rs:17
constexpr size_t kInputCapacity = 4096;  // resampler_fir.rs:18
constexpr size_t kBufferSize = 8192;     // resampler_fir.rs:19

enum class WindowType { Periodic, Symmetric };  // window.rs:5-15

double bessel_i0(double x);
std::vector<float> make_kaiser_window(size_t sample_count, double beta, WindowType type);
double calculate_cutoff_kaiser(size_t sample_count, double beta);
// Row-major [factor][sample_count].
std::vector<float> make_sincs_for_kaiser(size_t sample_count, size_t factor, float f_cutoff,
                                         double beta, WindowType type);

// Latency / Attenuation enums of the C ABI -> taps / Kaiser beta; 0 / negative when invalid.
size_t latency_taps(int latency);            // resampler_fir.rs:153-161
double attenuation_beta(int attenuation);    // resampler_fir.rs:117-123

struct FirDesign {
    double ratio;    // input_rate / output_rate in f64 (resampler_fir.rs:313)
    float cutoff;    // f32 cutoff actually used (resampler_fir.rs:318-326)
    size_t taps;
    double beta;
};
FirDesign fir_design(uint32_t in_hz, uint32_t out_hz, size_t taps, double beta);

// Process-wide cache keyed like the reference's FIR_CACHE (resampler_fir.rs:91-95, 425-443):
// (cutoff bits, taps, attenuation).  Table layout [1024][taps].
std::shared_ptr<const std::vector<float>> get_or_create_fir_coeffs(float cutoff, size_t taps,
                                                                   int attenuation);

}  // namespace rsmp
