/*
 * oracle.h -- CPU oracle for the resampler hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (hasenbanck/resampler v0.5.1),
 * written from reading the reference sources; every function cites the reference file:line
 * it follows.  It is the *checker* for the HIP path: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library
 * (resampler_amd/csrc -> libresampler_amd.so) never links or calls anything in here.
 *
 * Pinning status: the reference is a Rust crate and there is no rustc/cargo in the build
 * image, so the reference itself cannot be run (no oracle/_ref).  The oracle is pinned
 * against every known-answer value the reference's own unit tests hold for this path
 * (tests/golden/reference_known_answers.json, checked by tests/test_oracle_*.py):
 *   window.rs:152-385 (bessel_i0, Kaiser windows, cutoffs, sinc tables),
 *   planner.rs:248-443, optimizer.rs:72-165 (FFT plan tables),
 *   radix_fft.rs:723-1486 (FFT properties), resampler_fir.rs:741-815 (>= 90 dB stop band),
 *   resampler_fft.rs:439-566 (DC / sine amplitudes).
 * No reference test pins an end-to-end sample value or a (consumed, produced) sequence;
 * for those the oracle is the authority ("parity unpinned by the reference").
 */
#ifndef RESAMPLER_ORACLE_H
#define RESAMPLER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- window.rs ---------------------------------------------------------------------- */
enum { ORC_WINDOW_PERIODIC = 0, ORC_WINDOW_SYMMETRIC = 1 };

double orc_bessel_i0(double x);                                           /* window.rs:96-112  */
void orc_make_kaiser_window(size_t n, double beta, int window_type, float* out); /* :66-94   */
double orc_calculate_cutoff_kaiser(size_t sample_count, double beta);     /* window.rs:114-131 */
/* out is [factor][sample_count] row-major (window.rs:17-55). */
void orc_make_sincs_for_kaiser(size_t sample_count, size_t factor, float f_cutoff, double beta,
                               int window_type, float* out);

/* ---- fir/mod.rs, fir/avx.rs ----------------------------------------------------------- */
float orc_convolve_interp_scalar(const float* input, const float* c1, const float* c2, float frac,
                                 size_t taps);                            /* fir/mod.rs:47-62 */
float orc_convolve_interp_avx_fma(const float* input, const float* c1, const float* c2,
                                  float frac, size_t taps);               /* fir/avx.rs:5-61  */
int orc_have_avx_fma(void);
float orc_convolve_interp_avx512(const float* input, const float* c1, const float* c2,
                                 float frac, size_t taps);                /* fir/avx512.rs:5-50 (timing + differential test only) */
int orc_have_avx512f(void);

/* ---- resampler_fir.rs ------------------------------------------------------------------ */
typedef struct orc_fir orc_fir;

enum { ORC_CONVOLVE_SCALAR = 0, ORC_CONVOLVE_AVX_FMA = 1, ORC_CONVOLVE_AVX512 = 2 };

/* taps in {16,32,64,128}; attenuation_db in {60,90,120}; returns NULL on invalid arguments
 * (the reference panics on zero rates, resampler_fir.rs:302-309). */
orc_fir* orc_fir_new(size_t channels, uint32_t in_hz, uint32_t out_hz, size_t taps,
                     int attenuation_db, int convolve_kind);
void orc_fir_free(orc_fir* r);
size_t orc_fir_buffer_size_output(const orc_fir* r);                      /* :456-465 */
size_t orc_fir_delay(const orc_fir* r);                                   /* :630-632 */
void orc_fir_reset(orc_fir* r);                                           /* :638-642 */
/* 0 = Ok, 1 = InvalidInputBufferSize, 2 = InvalidOutputBufferSize (error.rs:3-8). */
int orc_fir_resample(orc_fir* r, const float* in, size_t in_len, float* out, size_t out_len,
                     size_t* consumed, size_t* produced);                 /* :509-621 */
const float* orc_fir_coeffs(const orc_fir* r);  /* [1024][taps] */
double orc_fir_ratio(const orc_fir* r);
int orc_fir_seek(orc_fir* r, size_t read_position, size_t available_frames, double position,
                 const float* history, size_t history_len);   /* test support: start mid-stream */
void orc_fir_state(const orc_fir* r, size_t* read_position, size_t* available_frames,
                   double* position);
/* The CLI driver loop (resample/src/main.rs:226-254) with a caller-chosen chunk length (in f32
 * values, the CLI uses 512).  Returns number of f32 values written to out (<= out_cap);
 * optional per-call counts are appended to calls[2*i], calls[2*i+1] up to max_calls. */
/* `calls` resample() calls of in_frames frames each, control flow only (no samples; see fir.c) */
unsigned long long orc_fir_skip_calls(orc_fir* r, size_t calls, size_t in_frames, unsigned long long* consumed_frames);
/* benches/benchmark_resampler_fir.rs:50-89: `iterations` resample() calls on the same input; values produced in total */
unsigned long long orc_fir_bench_calls(orc_fir* r, const float* in, size_t in_len, float* out, size_t out_len,
                                       size_t iterations);
size_t orc_fir_resample_all(orc_fir* r, const float* in, size_t in_len, size_t chunk_len,
                            float* out, size_t out_cap, size_t* calls, size_t max_calls,
                            size_t* n_calls);

/* ---- fft ------------------------------------------------------------------------------ */
typedef struct { float re, im; } orc_c32;                                 /* fft/mod.rs:11-16 */

/* planner.rs:35-245.  Sample rates must be members of the SampleRate enum (lib.rs:167-188);
 * returns 0 on success, -1 otherwise.  Factors are written as radix integers. */
int orc_fft_plan(uint32_t in_hz, uint32_t out_hz, size_t* fft_size_in, size_t* fft_size_out,
                 int* factors_in, size_t* n_factors_in, int* factors_out, size_t* n_factors_out,
                 int scale_for_throughput);
/* optimizer.rs:6-64 : in-place, returns new count. */
size_t orc_optimize_factors(int* factors, size_t n);

typedef struct orc_rfft orc_rfft;
/* RadixFFT::new (radix_fft.rs:105-183): factors multiply to the REAL length n (even). */
orc_rfft* orc_rfft_new(const int* factors, size_t n_factors, int inverse);
/* simd != 0: the reference's AVX + FMA butterflies and real <-> complex passes (fft_avx.c; scalar on a CPU without them) */
orc_rfft* orc_rfft_new_simd(const int* factors, size_t n_factors, int inverse, int simd);
void orc_rfft_free(orc_rfft* f);
size_t orc_rfft_len(const orc_rfft* f);
size_t orc_rfft_stage_factors(const orc_rfft* f, int* out);  /* the n/2-point stage list */
/* forward: n reals -> n/2+1 complex (radix_fft.rs:540-562). */
void orc_rfft_forward(orc_rfft* f, const float* in, orc_c32* out);
/* inverse: n/2+1 complex -> n reals, unnormalised (radix_fft.rs:565-589). */
void orc_rfft_inverse(orc_rfft* f, const orc_c32* in, float* out);
/* one out-of-place Stockham stage of the scalar butterfly specs (butterfly{2,3,4,5,7,8}/mod.rs) on n complex values */
void orc_butterfly_stage(const orc_c32* src, orc_c32* dst, size_t n, int radix, size_t stride, const orc_c32* tw);

typedef struct orc_fft_resampler orc_fft_resampler;
orc_fft_resampler* orc_fft_new(size_t channels, uint32_t in_hz, uint32_t out_hz); /* resampler_fft.rs:75-119 */
orc_fft_resampler* orc_fft_new_simd(size_t channels, uint32_t in_hz, uint32_t out_hz, int simd);
void orc_fft_free(orc_fft_resampler* r);
size_t orc_fft_chunk_size_input(const orc_fft_resampler* r);              /* :135-138 */
size_t orc_fft_chunk_size_output(const orc_fft_resampler* r);             /* :142-145 */
size_t orc_fft_delay(const orc_fft_resampler* r);                         /* :151-153 */
int orc_fft_resample(orc_fft_resampler* r, const float* in, size_t in_len, float* out,
                     size_t out_len);                                     /* :182-240 */
const orc_c32* orc_fft_filter_spectrum(const orc_fft_resampler* r, size_t* len);
/* The CLI driver loop resample_batch (resample/src/main.rs:256-313); returns values written, 0 on error. */
size_t orc_fft_resample_all(orc_fft_resampler* r, const float* in, size_t in_len, float* out, size_t out_cap);

/* ---- resample/src (CLI helpers around the path) --------------------------------------------- */
/* interpolation_resampler.rs:41-126; return output frames (0 when out_cap is too small). */
size_t orc_interp_linear(size_t channels, uint32_t in_hz, uint32_t out_hz, const float* input, size_t in_len,
                         float* output, size_t out_cap);
size_t orc_interp_hermite(size_t channels, uint32_t in_hz, uint32_t out_hz, const float* input, size_t in_len,
                          float* output, size_t out_cap);
/* main.rs:128-156: integer PCM -> f32, mono duplicated to stereo. */
size_t orc_pcm_to_stereo_f32(const uint8_t* pcm, int bits, int channels, size_t n_samples, float* out);

#ifdef __cplusplus
}
#endif
#endif
