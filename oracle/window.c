/*
 * window.c -- oracle restatement of the reference filter design (src/window.rs).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Compile with -ffp-contract=off: the reference is
 * Rust, which never contracts a*b+c into an FMA on its own.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>

/* window.rs:96-112 -- power series for I0, at most 1499 terms, stop when the sum stalls. */
double orc_bessel_i0(double x) {
    double base = x * x / 4.0;
    double term = 1.0;
    double result = 1.0;
    for (int idx = 1; idx < 1500; idx++) {
        term = term * base / (double)(idx * idx);
        double previous = result;
        result += term;
        if (result == previous) break;
    }
    return result;
}

/* window.rs:66-94 -- Kaiser window in f64, stored as f32. */
void orc_make_kaiser_window(size_t n, double beta, int window_type, float* out) {
    double bessel_beta = orc_bessel_i0(beta);
    for (size_t index = 0; index < n; index++) {
        double x = (double)index;
        double nx;
        if (window_type == ORC_WINDOW_PERIODIC) {
            nx = x / ((double)n / 2.0) - 1.0;                 /* window.rs:75-78 */
        } else {
            nx = 2.0 * x / (double)(n - 1) - 1.0;             /* window.rs:79-82 */
        }
        double value = orc_bessel_i0(beta * sqrt(1.0 - nx * nx)) / bessel_beta; /* :86 */
        out[index] = (float)value;
    }
}

/* window.rs:114-131 */
double orc_calculate_cutoff_kaiser(size_t sample_count, double beta) {
    double n = (double)sample_count;
    double a_db = beta / 0.1102 + 8.7;
    double delta_f_nyquist = (a_db - 7.95) / (14.36 * n);
    const double SAFETY_MARGIN = 1.005;
    double cutoff = 1.0 - (delta_f_nyquist * SAFETY_MARGIN);
    if (cutoff < 0.7) cutoff = 0.7;
    if (cutoff > 1.0) cutoff = 1.0;
    return cutoff;
}

/* window.rs:29-37 -- sinc in f32 with the f32 constant PI. */
static float sinc_f32(float value) {
    const float PI_F32 = 3.14159265358979323846f;
    if (value == 0.0f) return 1.0f;
    float a = value * PI_F32;
    return sinf(a) / a;
}

/* window.rs:17-55 -- windowed-sinc prototype, f32 running sum, polyphase split
 * sincs[factor-n-1][p] = y[factor*p+n] / (sum/factor). */
void orc_make_sincs_for_kaiser(size_t sample_count, size_t factor, float f_cutoff, double beta,
                               int window_type, float* out) {
    size_t totpoints = sample_count * factor;
    float* y = (float*)malloc(sizeof(float) * totpoints);
    float* window = (float*)malloc(sizeof(float) * totpoints);
    orc_make_kaiser_window(totpoints, beta, window_type, window);
    float sum = 0.0f;
    for (size_t x = 0; x < totpoints; x++) {
        int32_t d = (int32_t)x - (int32_t)(totpoints / 2);
        float arg = (float)d * f_cutoff / (float)factor;      /* window.rs:40 */
        float val = window[x] * sinc_f32(arg);
        sum += val;
        y[x] = val;
    }
    sum /= (float)factor;                                      /* window.rs:44 */
    for (size_t p = 0; p < sample_count; p++) {
        for (size_t n = 0; n < factor; n++) {
            out[(factor - n - 1) * sample_count + p] = y[factor * p + n] / sum;  /* :50 */
        }
    }
    free(window);
    free(y);
}
