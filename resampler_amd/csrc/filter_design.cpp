// filter_design.cpp -- see filter_design.h.  Build with -ffp-contract=off: the reference never
// fuses a*b+c here, and these tables define the numbers the kernels consume.
#include "filter_design.h"

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#include "common.h"

#pragma STDC FP_CONTRACT OFF

namespace rsmp {

// window.rs:96-112: power series of I0, <= 1499 terms, stop once the partial sum stalls.
double bessel_i0(double x) {
    const double base = x * x / 4.0;
    double term = 1.0, result = 1.0;
    for (int idx = 1; idx < 1500; ++idx) {
        term = term * base / static_cast<double>(idx * idx);
        const double previous = result;
        result += term;
        if (result == previous) break;
    }
    return result;
}

// window.rs:66-94
std::vector<float> make_kaiser_window(size_t sample_count, double beta, WindowType type) {
    std::vector<float> window(sample_count);
    const double i0_beta = bessel_i0(beta);
    for (size_t i = 0; i < sample_count; ++i) {
        const double x = static_cast<double>(i);
        const double nx = (type == WindowType::Periodic)
                              ? x / (static_cast<double>(sample_count) / 2.0) - 1.0
                              : 2.0 * x / static_cast<double>(sample_count - 1) - 1.0;
        const double sq = nx * nx;
        window[i] = static_cast<float>(bessel_i0(beta * std::sqrt(1.0 - sq)) / i0_beta);
    }
    return window;
}

// window.rs:114-131
double calculate_cutoff_kaiser(size_t sample_count, double beta) {
    const double n = static_cast<double>(sample_count);
    const double a_db = beta / 0.1102 + 8.7;
    const double delta_f_nyquist = (a_db - 7.95) / (14.36 * n);
    const double cutoff = 1.0 - (delta_f_nyquist * 1.005);
    return cutoff < 0.7 ? 0.7 : (cutoff > 1.0 ? 1.0 : cutoff);
}

// window.rs:17-55
std::vector<float> make_sincs_for_kaiser(size_t sample_count, size_t factor, float f_cutoff,
                                         double beta, WindowType type) {
    const size_t total = sample_count * factor;
    const std::vector<float> window = make_kaiser_window(total, beta, type);
    std::vector<float> proto(total);
    const float pi = 3.14159265358979323846f;
    float sum = 0.0f;  // f32 running sum, as the reference (window.rs:27,41)
    for (size_t x = 0; x < total; ++x) {
        const int32_t centred = static_cast<int32_t>(x) - static_cast<int32_t>(total / 2);
        const float arg = static_cast<float>(centred) * f_cutoff / static_cast<float>(factor);
        float s = 1.0f;
        if (arg != 0.0f) {
            const float a = arg * pi;
            s = sinf(a) / a;
        }
        const float v = window[x] * s;
        sum += v;
        proto[x] = v;
    }
    sum /= static_cast<float>(factor);
    std::vector<float> sincs(total);
    for (size_t p = 0; p < sample_count; ++p)
        for (size_t n = 0; n < factor; ++n)
            sincs[(factor - n - 1) * sample_count + p] = proto[factor * p + n] / sum;
    return sincs;
}

size_t latency_taps(int latency) {
    switch (latency) {
        case RSMP_LATENCY_SAMPLE8: return 16;
        case RSMP_LATENCY_SAMPLE16: return 32;
        case RSMP_LATENCY_SAMPLE32: return 64;
        case RSMP_LATENCY_SAMPLE64: return 128;
        default: return 0;
    }
}

double attenuation_beta(int attenuation) {
    switch (attenuation) {
        case RSMP_ATTENUATION_DB60: return 7.0;
        case RSMP_ATTENUATION_DB90: return 10.0;
        case RSMP_ATTENUATION_DB120: return 13.0;
        default: return -1.0;
    }
}

// resampler_fir.rs:311-326
FirDesign fir_design(uint32_t in_hz, uint32_t out_hz, size_t taps, double beta) {
    const double in_f = static_cast<double>(in_hz), out_f = static_cast<double>(out_hz);
    FirDesign d;
    d.ratio = in_f / out_f;
    const double base = calculate_cutoff_kaiser(taps, beta);
    const double cutoff = (in_f <= out_f) ? base : base * (out_f / in_f);
    d.cutoff = static_cast<float>(cutoff);
    d.taps = taps;
    d.beta = beta;
    return d;
}

std::shared_ptr<const std::vector<float>> get_or_create_fir_coeffs(float cutoff, size_t taps,
                                                                   int attenuation) {
    using Key = std::tuple<uint32_t, size_t, int>;
    static std::mutex mu;
    static std::map<Key, std::shared_ptr<const std::vector<float>>> cache;
    uint32_t bits;
    std::memcpy(&bits, &cutoff, sizeof bits);
    const Key key{bits, taps, attenuation};
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    auto table = std::make_shared<const std::vector<float>>(make_sincs_for_kaiser(
        taps, kPhases, cutoff, attenuation_beta(attenuation), WindowType::Symmetric));
    cache.emplace(key, table);
    return table;
}

}  // namespace rsmp

extern "C" double rsmp_design_cutoff_kaiser(size_t sample_count, double beta) {
    return rsmp::calculate_cutoff_kaiser(sample_count, beta);
}

extern "C" int rsmp_design_fir_coeffs(uint32_t input_rate_hz, uint32_t output_rate_hz, int latency,
                                      int attenuation, float* out, size_t out_len) {
    const size_t taps = rsmp::latency_taps(latency);
    const double beta = rsmp::attenuation_beta(attenuation);
    if (!taps || beta < 0 || !input_rate_hz || !output_rate_hz || !out)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_design_fir_coeffs: invalid argument");
    if (out_len < rsmp::kPhases * taps)
        return rsmp::fail(RSMP_ERR_CAPACITY, "rsmp_design_fir_coeffs: need %zu floats",
                          rsmp::kPhases * taps);
    const rsmp::FirDesign d = rsmp::fir_design(input_rate_hz, output_rate_hz, taps, beta);
    auto table = rsmp::get_or_create_fir_coeffs(d.cutoff, taps, attenuation);
    std::memcpy(out, table->data(), sizeof(float) * table->size());
    return RSMP_OK;
}
