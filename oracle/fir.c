/*
 * fir.c -- oracle restatement of ResamplerFir (src/resampler_fir.rs) and of the convolution
 * leaves (src/fir/mod.rs scalar spec, src/fir/avx.rs AVX+FMA).  TEST INFRASTRUCTURE ONLY.
 */
#include "oracle.h"

#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PHASES 1024u          /* resampler_fir.rs:17 */
#define INPUT_CAPACITY 4096u  /* resampler_fir.rs:18 */
#define BUFFER_SIZE 8192u     /* resampler_fir.rs:19 */

/* fir/mod.rs:47-62 */
float orc_convolve_interp_scalar(const float* input, const float* c1, const float* c2, float frac,
                                 size_t taps) {
    float sum1 = 0.0f, sum2 = 0.0f;
    for (size_t i = 0; i < taps; i++) {
        float v = input[i];
        sum1 += c1[i] * v;
        sum2 += c2[i] * v;
    }
    return sum1 * (1.0f - frac) + sum2 * frac;
}

int orc_have_avx_fma(void) {
    return __builtin_cpu_supports("avx") && __builtin_cpu_supports("fma");
}

/* fir/avx.rs:5-61 -- 8 lanes, two FMA accumulators, per-lane lerp, then
 * hi128+lo128, [2,3,0,1] swap-add, lane-1 add. */
__attribute__((target("avx,fma")))
float orc_convolve_interp_avx_fma(const float* input, const float* c1, const float* c2,
                                  float frac, size_t taps) {
    size_t iters = taps / 8;
    __m256 acc1 = _mm256_setzero_ps();
    __m256 acc2 = _mm256_setzero_ps();
    for (size_t i = 0; i < iters; i++) {
        size_t off = i * 8;
        __m256 x = _mm256_loadu_ps(input + off);
        __m256 k1 = _mm256_load_ps(c1 + off);
        __m256 k2 = _mm256_load_ps(c2 + off);
        acc1 = _mm256_fmadd_ps(k1, x, acc1);
        acc2 = _mm256_fmadd_ps(k2, x, acc2);
    }
    __m256 fv = _mm256_set1_ps(frac);
    __m256 omf = _mm256_set1_ps(1.0f - frac);
    __m256 w1 = _mm256_mul_ps(acc1, omf);
    __m256 w2 = _mm256_mul_ps(acc2, fv);
    __m256 interp = _mm256_add_ps(w1, w2);
    __m128 high = _mm256_extractf128_ps(interp, 1);
    __m128 low = _mm256_castps256_ps128(interp);
    __m128 sum128 = _mm_add_ps(high, low);
    __m128 shuf = _mm_shuffle_ps(sum128, sum128, 0x4E); /* 0b01_00_11_10 */
    __m128 s1 = _mm_add_ps(sum128, shuf);
    __m128 shuf2 = _mm_shuffle_ps(s1, s1, 0x01);        /* 0b00_00_00_01 */
    __m128 s2 = _mm_add_ps(s1, shuf2);
    return _mm_cvtss_f32(s2);
}

int orc_have_avx512f(void) { return __builtin_cpu_supports("avx512f"); }

/* fir/avx512.rs:5-50 -- 16 lanes, two FMA accumulators, per-lane lerp, one horizontal sum.  What the reference's runtime
 * dispatch takes on a CPU with avx512f and taps >= 16 (resampler_fir.rs:331-345: avx512f is asked for FIRST), i.e. what
 * its published Zen 5 figures were measured on (CHANGELOG.md:77).  The horizontal sum is `_mm512_reduce_add_ps`: halves,
 * quarters, [2,3,2,3], lane 1 -- the same tree in GCC's header and in stdarch.  Held to the scalar spec within the
 * reference's own 1e-5 (fir/mod.rs:137-192); not the leaf the parity oracle uses (north_star names AVX+FMA). */
__attribute__((target("avx512f")))
float orc_convolve_interp_avx512(const float* input, const float* c1, const float* c2,
                                 float frac, size_t taps) {
    size_t iters = taps / 16;
    __m512 acc1 = _mm512_setzero_ps();
    __m512 acc2 = _mm512_setzero_ps();
    for (size_t i = 0; i < iters; i++) {
        size_t off = i * 16;
        __m512 x = _mm512_loadu_ps(input + off);
        __m512 k1 = _mm512_load_ps(c1 + off);
        __m512 k2 = _mm512_load_ps(c2 + off);
        acc1 = _mm512_fmadd_ps(k1, x, acc1);
        acc2 = _mm512_fmadd_ps(k2, x, acc2);
    }
    __m512 fv = _mm512_set1_ps(frac);
    __m512 omf = _mm512_set1_ps(1.0f - frac);
    __m512 interp = _mm512_add_ps(_mm512_mul_ps(acc1, omf), _mm512_mul_ps(acc2, fv));
    return _mm512_reduce_add_ps(interp);
}

typedef float (*convolve_fn)(const float*, const float*, const float*, float, size_t);

struct orc_fir {
    size_t channels;
    float* coeffs;          /* [PHASES][taps], 64-byte aligned (resampler_fir.rs:25-51) */
    float* input_buffers;   /* planar, BUFFER_SIZE per channel (:329) */
    size_t read_position;
    size_t available_frames;
    double position;
    double ratio;
    size_t taps;
    convolve_fn convolve;
};

static double beta_for(int attenuation_db) {   /* resampler_fir.rs:117-123 */
    switch (attenuation_db) {
        case 60: return 7.0;
        case 90: return 10.0;
        case 120: return 13.0;
        default: return -1.0;
    }
}

/* resampler_fir.rs:295-404 */
orc_fir* orc_fir_new(size_t channels, uint32_t in_hz, uint32_t out_hz, size_t taps,
                     int attenuation_db, int convolve_kind) {
    double beta = beta_for(attenuation_db);
    if (in_hz == 0 || out_hz == 0 || channels == 0 || beta < 0.0) return NULL;
    if (!(taps == 16 || taps == 32 || taps == 64 || taps == 128)) return NULL;
    if (convolve_kind == ORC_CONVOLVE_AVX_FMA && !orc_have_avx_fma()) return NULL;
    if (convolve_kind == ORC_CONVOLVE_AVX512 && !(orc_have_avx512f() && taps >= 16)) return NULL;

    orc_fir* r = (orc_fir*)calloc(1, sizeof(orc_fir));
    double in_f = (double)in_hz, out_f = (double)out_hz;
    r->channels = channels;
    r->ratio = in_f / out_f;                                        /* :313 */
    r->taps = taps;
    double base_cutoff = orc_calculate_cutoff_kaiser(taps, beta);   /* :317 */
    double cutoff = (in_f <= out_f) ? base_cutoff : base_cutoff * (out_f / in_f); /* :318-324 */
    r->coeffs = (float*)aligned_alloc(64, sizeof(float) * PHASES * taps);
    /* :326 cutoff as f32, :407-408 symmetric window, :410-416 row-major flatten */
    orc_make_sincs_for_kaiser(taps, PHASES, (float)cutoff, beta, ORC_WINDOW_SYMMETRIC, r->coeffs);
    r->input_buffers = (float*)calloc(BUFFER_SIZE * channels, sizeof(float));
    r->convolve = convolve_kind == ORC_CONVOLVE_AVX512    ? orc_convolve_interp_avx512
                  : convolve_kind == ORC_CONVOLVE_AVX_FMA ? orc_convolve_interp_avx_fma
                                                          : orc_convolve_interp_scalar;
    return r;
}

void orc_fir_free(orc_fir* r) {
    if (!r) return;
    free(r->coeffs);
    free(r->input_buffers);
    free(r);
}

/* resampler_fir.rs:456-465 */
size_t orc_fir_buffer_size_output(const orc_fir* r) {
    double max_usable = (double)(INPUT_CAPACITY - r->taps);
    size_t max_output_frames = (size_t)ceil(max_usable / r->ratio) + 2;
    return max_output_frames * r->channels;
}

size_t orc_fir_delay(const orc_fir* r) { return r->taps / 2; }

void orc_fir_reset(orc_fir* r) {
    r->read_position = 0;
    r->available_frames = 0;
    r->position = 0.0;
}

const float* orc_fir_coeffs(const orc_fir* r) { return r->coeffs; }
double orc_fir_ratio(const orc_fir* r) { return r->ratio; }

void orc_fir_state(const orc_fir* r, size_t* read_position, size_t* available_frames,
                   double* position) {
    if (read_position) *read_position = r->read_position;
    if (available_frames) *available_frames = r->available_frames;
    if (position) *position = r->position;
}

/* Test support (no reference counterpart: the state is private, resampler_fir.rs:189-192): puts the three
 * scalars where a run over the stream's earlier input left them and refills the planar ring with the
 * `available_frames` interleaved frames that precede the point.  Lets the CPU tests run one piece of a
 * stream on its own (tests/test_sharding_gloo.py). */
int orc_fir_seek(orc_fir* r, size_t read_position, size_t available_frames, double position,
                 const float* history, size_t history_len) {
    size_t ch = r->channels;
    if (read_position + available_frames > BUFFER_SIZE || history_len < available_frames * ch) return 1;
    const float* h = history + (history_len - available_frames * ch);
    for (size_t f = 0; f < available_frames; f++)
        for (size_t c = 0; c < ch; c++) r->input_buffers[BUFFER_SIZE * c + read_position + f] = h[f * ch + c];
    r->read_position = read_position;
    r->available_frames = available_frames;
    r->position = position;
    return 0;
}

static size_t min_sz(size_t a, size_t b) { return a < b ? a : b; }

/* resampler_fir.rs:509-621 */
int orc_fir_resample(orc_fir* r, const float* in, size_t in_len, float* out, size_t out_len,
                     size_t* consumed, size_t* produced) {
    size_t ch = r->channels;
    if (in_len % ch != 0) return 1;
    if (out_len % ch != 0) return 2;

    size_t input_frames = in_len / ch;
    size_t output_capacity = out_len / ch;

    size_t write_position = r->read_position + r->available_frames;
    size_t remaining_capacity = BUFFER_SIZE > write_position ? BUFFER_SIZE - write_position : 0;
    size_t frames_to_copy =
        min_sz(min_sz(input_frames, remaining_capacity), INPUT_CAPACITY - r->available_frames);

    for (size_t f = 0; f < frames_to_copy; f++)                      /* :531-537 */
        for (size_t c = 0; c < ch; c++)
            r->input_buffers[BUFFER_SIZE * c + write_position + f] = in[f * ch + c];
    r->available_frames += frames_to_copy;

    size_t output_frame_count = 0;
    for (;;) {                                                        /* :542-590 */
        size_t input_offset = (size_t)floor(r->position);
        if (input_offset + r->taps > r->available_frames) break;
        if (output_frame_count >= output_capacity) break;

        double position_fract = r->position - trunc(r->position);    /* f64::fract */
        double phase_f = position_fract * (double)PHASES;
        if (phase_f > (double)(PHASES - 1)) phase_f = (double)(PHASES - 1);  /* .min() :562 */
        size_t phase1 = (size_t)phase_f;
        size_t phase2 = min_sz(phase1 + 1, PHASES - 1);
        float frac = (float)(phase_f - (double)phase1);

        for (size_t c = 0; c < ch; c++) {
            size_t actual_pos = r->read_position + input_offset;
            const float* slice = r->input_buffers + BUFFER_SIZE * c + actual_pos;
            out[output_frame_count * ch + c] = r->convolve(
                slice, r->coeffs + phase1 * r->taps, r->coeffs + phase2 * r->taps, frac, r->taps);
        }
        output_frame_count++;
        r->position += r->ratio;                                      /* :589 */
    }

    size_t consumed_frames = min_sz((size_t)floor(r->position), r->available_frames); /* :596 */
    r->read_position += consumed_frames;
    r->available_frames -= consumed_frames;
    r->position -= (double)consumed_frames;

    if (r->read_position > INPUT_CAPACITY) {                          /* :605-615 */
        for (size_t c = 0; c < ch; c++) {
            float* buf = r->input_buffers + BUFFER_SIZE * c;
            memmove(buf, buf + r->read_position, sizeof(float) * r->available_frames);
        }
        r->read_position = 0;
    }

    *consumed = frames_to_copy * ch;
    *produced = output_frame_count * ch;
    return 0;
}

/* The control flow of `calls` consecutive resample() calls of `in_frames` frames each WITHOUT their samples: what
 * resampler_fir.rs:509-621 does to read_position / available_frames / position (the f64 recurrence `position += ratio`,
 * :589, output by output) when every call's output buffer has room (output_capacity = buffer_size_output()).  The input
 * ring is left untouched -- orc_fir_seek hands the buffered frames back before the next real call.  For soak tests that
 * age a stream by hours of audio in seconds (tools/soak_lockstep.py); returns the output frames the calls produce. */
unsigned long long orc_fir_skip_calls(orc_fir* r, size_t calls, size_t in_frames, unsigned long long* consumed_frames) {
    unsigned long long produced = 0, consumed = 0;
    const size_t cap = orc_fir_buffer_size_output(r) / r->channels;
    for (size_t c = 0; c < calls; c++) {
        size_t write_position = r->read_position + r->available_frames;
        size_t remaining_capacity = BUFFER_SIZE > write_position ? BUFFER_SIZE - write_position : 0;
        size_t frames_to_copy = min_sz(min_sz(in_frames, remaining_capacity), INPUT_CAPACITY - r->available_frames);
        r->available_frames += frames_to_copy;
        size_t n = 0;
        for (;;) {                                                    /* :542-590 */
            size_t input_offset = (size_t)floor(r->position);
            if (input_offset + r->taps > r->available_frames) break;
            if (n >= cap) break;
            n++;
            r->position += r->ratio;                                  /* :589 */
        }
        size_t consumed_now = min_sz((size_t)floor(r->position), r->available_frames);   /* :596 */
        r->read_position += consumed_now;
        r->available_frames -= consumed_now;
        r->position -= (double)consumed_now;
        if (r->read_position > INPUT_CAPACITY) r->read_position = 0;  /* :605-615 */
        produced += n;
        consumed += frames_to_copy;
    }
    if (consumed_frames) *consumed_frames = consumed;
    return produced;
}

/* resample/src/main.rs:226-254 with the chunk length as a parameter (CLI: 512 values). */
size_t orc_fir_resample_all(orc_fir* r, const float* in, size_t in_len, size_t chunk_len,
                            float* out, size_t out_cap, size_t* calls, size_t max_calls,
                            size_t* n_calls) {
    size_t bso = orc_fir_buffer_size_output(r);
    float* tmp = (float*)malloc(sizeof(float) * bso);
    size_t in_off = 0, out_off = 0, nc = 0;
    while (in_off < in_len) {
        size_t remaining = in_len - in_off;
        size_t chunk = remaining < chunk_len ? remaining : chunk_len;
        size_t consumed = 0, produced = 0;
        int rc = orc_fir_resample(r, in + in_off, chunk, tmp, bso, &consumed, &produced);
        if (rc != 0) break;
        if (calls && nc < max_calls) {
            calls[2 * nc] = consumed;
            calls[2 * nc + 1] = produced;
        }
        nc++;
        size_t ncopy = produced;
        if (out_off + ncopy > out_cap) ncopy = out_cap - out_off;
        memcpy(out + out_off, tmp, sizeof(float) * ncopy);
        out_off += ncopy;
        in_off += consumed;
        if (consumed == 0) break;
    }
    if (n_calls) *n_calls = nc;
    free(tmp);
    return out_off;
}

/* The criterion bench's inner loop (benches/benchmark_resampler_fir.rs:50-89): `iterations` calls of resample() on the SAME
 * `in_len` values (1024 there), the output buffer of buffer_size_output() values; returns the values produced in total.
 * (One C call for the whole loop: a ctypes call per 8 us resample() would be a tenth of what is timed.) */
unsigned long long orc_fir_bench_calls(orc_fir* r, const float* in, size_t in_len, float* out, size_t out_len,
                                       size_t iterations) {
    unsigned long long total = 0;
    for (size_t i = 0; i < iterations; i++) {
        size_t consumed = 0, produced = 0;
        if (orc_fir_resample(r, in, in_len, out, out_len, &consumed, &produced) != 0) break;
        total += produced;
    }
    return total;
}
