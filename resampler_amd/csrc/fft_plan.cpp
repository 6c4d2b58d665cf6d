// fft_plan.cpp -- see fft_plan.h.  Build with -ffp-contract=off (f64 twiddle angles must be the
// reference's `constant * index` products, radix_fft.rs:252-254).
#include "fft_plan.h"

#include <algorithm>
#include <cmath>

#include "common.h"
#include "filter_design.h"

#pragma STDC FP_CONTRACT OFF

namespace rsmp {

namespace {

// SampleRate::family / family_multiplier (lib.rs:191-216)
bool family_of(uint32_t hz, uint32_t* family) {
    switch (hz) {
        case 22050: case 44100: case 88200: case 176400: *family = 22050; return true;
        case 16000: case 32000: *family = 16000; return true;
        case 48000: case 96000: case 192000: case 384000: *family = 48000; return true;
        default: return false;
    }
}

// decompose_multiplier (planner.rs:183-207): 8s first, then one 2 or 4.
void append_multiplier(size_t multiplier, std::vector<int>* factors) {
    if (multiplier == 1) return;
    size_t bits = 0;
    while ((size_t{1} << bits) < multiplier) ++bits;
    for (size_t i = 0; i < bits / 3; ++i) factors->push_back(8);
    if (bits % 3 == 1) factors->push_back(2);
    if (bits % 3 == 2) factors->push_back(4);
}

// compute_twiddle_f32 (radix_fft.rs:251-258)
Complex32 twiddle_f32(size_t index, size_t fft_len) {
    const double constant = -2.0 * 3.14159265358979323846264338327950288 / static_cast<double>(fft_len);
    const double angle = constant * static_cast<double>(index);
    return Complex32{static_cast<float>(std::cos(angle)), static_cast<float>(std::sin(angle))};
}

bool try_transform(std::vector<int>* factors, std::initializer_list<int> remove,
                   std::initializer_list<int> add) {
    std::vector<int> tmp = *factors;
    for (int r : remove) {
        auto it = std::find(tmp.begin(), tmp.end(), r);
        if (it == tmp.end()) return false;
        tmp.erase(it);
    }
    for (int a : add) tmp.push_back(a);
    *factors = tmp;
    return true;
}

}  // namespace

bool fft_conversion_config(uint32_t in_hz, uint32_t out_hz, bool scale_for_throughput,
                           size_t* fft_size_in, std::vector<int>* factors_in,
                           size_t* fft_size_out, std::vector<int>* factors_out) {
    uint32_t fam_in, fam_out;
    if (!family_of(in_hz, &fam_in) || !family_of(out_hz, &fam_out)) return false;
    const size_t mul_in = in_hz / fam_in, mul_out = out_hz / fam_out;
    size_t base_in, base_out;
    std::vector<int> fi, fo;
    const std::vector<int> f588{3, 4, 7, 7}, f1280{4, 4, 4, 4, 5}, f64{2, 2, 2, 2, 2, 2},
        f192{4, 4, 4, 3}, f640{2, 4, 4, 4, 5}, f882{2, 3, 3, 7, 7};
    if (fam_in == fam_out) { base_in = 2; fi = {2}; base_out = 2; fo = {2}; }                    // :46-53
    else if (fam_in == 22050 && fam_out == 48000) { base_in = 588; fi = f588; base_out = 1280; fo = f1280; }   // :56-73
    else if (fam_in == 48000 && fam_out == 22050) { base_in = 1280; fi = f1280; base_out = 588; fo = f588; }   // :75-92
    else if (fam_in == 16000 && fam_out == 48000) { base_in = 64; fi = f64; base_out = 192; fo = f192; }       // :95-105
    else if (fam_in == 48000 && fam_out == 16000) { base_in = 192; fi = f192; base_out = 64; fo = f64; }       // :107-117
    else if (fam_in == 16000 && fam_out == 22050) { base_in = 640; fi = f640; base_out = 882; fo = f882; }     // :120-137
    else { base_in = 882; fi = f882; base_out = 640; fo = f640; }                                               // :139-156
    append_multiplier(mul_in, &fi);     // :159-171
    append_multiplier(mul_out, &fo);
    size_t size_in = base_in * mul_in, size_out = base_out * mul_out;
    if (scale_for_throughput) {         // :212-245, TARGET_INPUT_SAMPLES = 512
        float m = std::ceil(512.0f / static_cast<float>(size_in));
        if (m < 1.0f) m = 1.0f;
        const size_t multiplier = static_cast<size_t>(m);
        size_t p2 = 1;
        while (p2 < multiplier) p2 <<= 1;
        size_in *= p2;
        size_out *= p2;
        append_multiplier(p2, &fi);
        append_multiplier(p2, &fo);
    }
    *fft_size_in = size_in;
    *fft_size_out = size_out;
    *factors_in = fi;
    *factors_out = fo;
    return true;
}

std::vector<int> optimize_factors(std::vector<int> factors) {
    auto desc = [](int a, int b) { return a > b; };
    std::stable_sort(factors.begin(), factors.end(), desc);
    for (;;) {
        const bool changed = try_transform(&factors, {4, 2}, {8}) ||
                             try_transform(&factors, {2, 2, 2}, {8}) ||
                             try_transform(&factors, {4, 4}, {8, 2}) ||
                             try_transform(&factors, {2, 2}, {4});
        if (!changed) break;
        std::stable_sort(factors.begin(), factors.end(), desc);
    }
    std::stable_sort(factors.begin(), factors.end());
    return factors;
}

RealFftPlan make_real_fft_plan(const std::vector<int>& factors, bool inverse) {
    RealFftPlan p;
    if (factors.empty()) return p;
    size_t n = 1;
    for (int r : factors) {
        if (!(r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 8)) return p;
        n *= static_cast<size_t>(r);
    }
    if (n % 2 != 0) return p;   // radix_fft.rs:109-113
    p.n = n;
    p.n2 = n / 2;
    // compute_factors (radix_fft.rs:222-246): take one factor 2 out of the list
    std::vector<int> f = factors;
    if (f.size() == 1) {
        if (f[0] == 2) f.clear();
        else if (f[0] == 4) f = {2};
        else if (f[0] == 8) f = {4};
        else return p;
    } else {
        auto it = std::find(f.begin(), f.end(), 2);
        if (it != f.end()) f.erase(it);
        else if ((it = std::find(f.begin(), f.end(), 8)) != f.end()) *it = 4;
        else if ((it = std::find(f.begin(), f.end(), 4)) != f.end()) *it = 2;
        else return p;
    }
    p.stages = optimize_factors(f);
    // Unique stage twiddles (the reference replicates them per iteration, :273-362)
    size_t stride = 1;
    for (size_t s = 0; s < p.stages.size(); ++s) {
        const size_t r = static_cast<size_t>(p.stages[s]);
        p.stage_twiddle_offset.push_back(static_cast<uint32_t>(p.stage_twiddles.size()));
        if (s > 0) {
            const size_t stage_size = stride * r;
            for (size_t col = 0; col < stride; ++col)
                for (size_t k = 1; k < r; ++k) p.stage_twiddles.push_back(twiddle_f32(col * k, stage_size));
        }
        stride *= r;
    }
    const size_t count = (n % 4 == 0) ? n / 4 : n / 4 + 1;   // twiddle_count, :366-372
    for (size_t k = 1; k < count; ++k) {
        const Complex32 t = twiddle_f32(k, n);
        p.rc_twiddles.push_back(inverse ? Complex32{t.re, -t.im} : Complex32{t.re * 0.5f, t.im * 0.5f});
    }
    p.ok = true;
    return p;
}

FftResamplerPlan make_fft_resampler_plan(uint32_t in_hz, uint32_t out_hz) {
    FftResamplerPlan plan;
    std::vector<int> fin, fout;
    if (!fft_conversion_config(in_hz, out_hz, true, &plan.fft_in, &fin, &plan.fft_out, &fout))
        return plan;
    fin.push_back(2);    // resampler_fft.rs:345-348: real FFT length = 2 x block
    fout.push_back(2);
    plan.forward = make_real_fft_plan(fin, false);
    plan.inverse = make_real_fft_plan(fout, true);
    if (!plan.forward.ok || !plan.inverse.ok) return plan;
    // resampler_fft.rs:353-359
    const double cutoff =
        plan.fft_in > plan.fft_out
            ? calculate_cutoff_kaiser(plan.fft_out, 10.0) *
                  (static_cast<double>(plan.fft_out) / static_cast<double>(plan.fft_in))
            : calculate_cutoff_kaiser(plan.fft_in, 10.0);
    const std::vector<float> sincs = make_sincs_for_kaiser(plan.fft_in, 1, static_cast<float>(cutoff),
                                                           10.0, WindowType::Periodic);
    plan.filter_time.assign(2 * plan.fft_in, 0.0f);
    for (size_t i = 0; i < plan.fft_in; ++i)
        plan.filter_time[i] = sincs[i] / static_cast<float>(2 * plan.fft_in);   // :371-373
    plan.new_length = plan.fft_in < plan.fft_out ? plan.fft_in + 1 : plan.fft_out;  // :396-399
    plan.ok = true;
    return plan;
}

}  // namespace rsmp

// ---- host-only C ABI ---------------------------------------------------------------------------
extern "C" int rsmp_fft_plan_sizes(uint32_t input_rate_hz, uint32_t output_rate_hz,
                                   size_t* fft_size_input, size_t* fft_size_output,
                                   int* forward_stages, size_t* n_forward_stages,
                                   int* inverse_stages, size_t* n_inverse_stages,
                                   size_t max_stages) {
    const rsmp::FftResamplerPlan p = rsmp::make_fft_resampler_plan(input_rate_hz, output_rate_hz);
    if (!p.ok)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "ResamplerFft: %u -> %u Hz is not a SampleRate pair",
                          input_rate_hz, output_rate_hz);
    if (p.forward.stages.size() > max_stages || p.inverse.stages.size() > max_stages)
        return rsmp::fail(RSMP_ERR_CAPACITY, "rsmp_fft_plan_sizes: need room for %zu stages",
                          std::max(p.forward.stages.size(), p.inverse.stages.size()));
    if (fft_size_input) *fft_size_input = p.fft_in;
    if (fft_size_output) *fft_size_output = p.fft_out;
    if (n_forward_stages) *n_forward_stages = p.forward.stages.size();
    if (n_inverse_stages) *n_inverse_stages = p.inverse.stages.size();
    for (size_t i = 0; forward_stages && i < p.forward.stages.size(); ++i) forward_stages[i] = p.forward.stages[i];
    for (size_t i = 0; inverse_stages && i < p.inverse.stages.size(); ++i) inverse_stages[i] = p.inverse.stages[i];
    return RSMP_OK;
}
