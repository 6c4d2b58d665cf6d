/*
 * cli.c -- oracle restatement of the reference CLI's helpers around the hot path (TEST INFRASTRUCTURE
 * ONLY, see oracle.h): the comparison interpolators (resample/src/interpolation_resampler.rs:41-126) and
 * the WAV sample conversion + mono duplication (resample/src/main.rs:128-156).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "oracle.h"

/* InterpolationResampler::resample_linear (interpolation_resampler.rs:41-77). Returns output frames. */
size_t orc_interp_linear(size_t channels, uint32_t in_hz, uint32_t out_hz, const float* input, size_t in_len,
                         float* output, size_t out_cap) {
    const size_t input_frames = in_len / channels;
    const double ratio = (double)out_hz / (double)in_hz;
    const size_t output_frames = (size_t)ceil((double)input_frames * ratio);
    if (output_frames * channels > out_cap) return 0;
    for (size_t o = 0; o < output_frames; ++o) {
        const double input_pos = (double)o / ratio;
        const size_t idx = (size_t)floor(input_pos);
        const float frac = (float)(input_pos - (double)idx);
        if (idx >= input_frames - 1) {
            for (size_t ch = 0; ch < channels; ++ch) output[o * channels + ch] = input[(input_frames - 1) * channels + ch];
            continue;
        }
        for (size_t ch = 0; ch < channels; ++ch) {
            const float s0 = input[idx * channels + ch], s1 = input[(idx + 1) * channels + ch];
            output[o * channels + ch] = s0 * (1.0f - frac) + s1 * frac;
        }
    }
    return output_frames;
}

/* InterpolationResampler::resample_hermite (interpolation_resampler.rs:84-126). */
size_t orc_interp_hermite(size_t channels, uint32_t in_hz, uint32_t out_hz, const float* input, size_t in_len,
                          float* output, size_t out_cap) {
    const size_t input_frames = in_len / channels;
    const double ratio = (double)out_hz / (double)in_hz;
    const size_t output_frames = (size_t)ceil((double)input_frames * ratio);
    if (output_frames * channels > out_cap) return 0;
    for (size_t o = 0; o < output_frames; ++o) {
        const double input_pos = (double)o / ratio;
        const size_t idx = (size_t)floor(input_pos);
        const float frac = (float)(input_pos - (double)idx);
        for (size_t ch = 0; ch < channels; ++ch) {
            const size_t i_prev = idx > 0 ? idx - 1 : 0;
            const size_t i_cur = idx < input_frames - 1 ? idx : input_frames - 1;
            const size_t i_n1 = idx + 1 < input_frames - 1 ? idx + 1 : input_frames - 1;
            const size_t i_n2 = idx + 2 < input_frames - 1 ? idx + 2 : input_frames - 1;
            const float previous = input[i_prev * channels + ch], current = input[i_cur * channels + ch];
            const float next_1 = input[i_n1 * channels + ch], next_2 = input[i_n2 * channels + ch];
            const float c0 = current;
            const float c1 = (next_1 - previous) * 0.5f;
            const float c2 = previous - current * 2.5f + next_1 * 2.0f - next_2 * 0.5f;
            const float c3 = (next_2 - previous) * 0.5f + (current - next_1) * 1.5f;
            output[o * channels + ch] = ((c3 * frac + c2) * frac + c1) * frac + c0;
        }
    }
    return output_frames;
}

/* main.rs:128-156: integer PCM -> f32 (`s as f32 / (1 << (bits - 1)) as f32`), mono duplicated to stereo.
 * pcm: little-endian samples of `bits` (16, 24 packed, 32); n_samples over all channels; out holds
 * n_samples values (stereo in) or 2 * n_samples (mono in).  Returns values written. */
size_t orc_pcm_to_stereo_f32(const uint8_t* pcm, int bits, int channels, size_t n_samples, float* out) {
    /* `(1 << (spec.bits_per_sample - 1)) as f32` (main.rs:131): an i32 literal, so bits == 32 gives
     * 1i32 << 31 = i32::MIN = -2^31 and 32-bit samples come out with inverted polarity. */
    const float max_value = bits == 32 ? -2147483648.0f : (float)((int32_t)1 << (bits - 1));
    size_t w = 0;
    for (size_t i = 0; i < n_samples; ++i) {
        int32_t s;
        if (bits == 16) s = (int16_t)((uint16_t)pcm[2 * i] | ((uint16_t)pcm[2 * i + 1] << 8));
        else if (bits == 24) {
            const uint32_t u = (uint32_t)pcm[3 * i] | ((uint32_t)pcm[3 * i + 1] << 8) | ((uint32_t)pcm[3 * i + 2] << 16);
            s = (int32_t)(u << 8) >> 8;
        } else s = (int32_t)((uint32_t)pcm[4 * i] | ((uint32_t)pcm[4 * i + 1] << 8) | ((uint32_t)pcm[4 * i + 2] << 16) | ((uint32_t)pcm[4 * i + 3] << 24));
        const float v = (float)s / max_value;
        out[w++] = v;
        if (channels == 1) out[w++] = v;
    }
    return w;
}
