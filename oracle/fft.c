/*
 * fft.c -- oracle restatement of the reference's FFT resampler: planner (src/fft/planner.rs),
 * factor optimiser (src/fft/optimizer.rs), the N/2-trick real FFT on a mixed-radix Stockham
 * autosort (src/fft/radix_fft.rs, src/fft/stockham_autosort.rs, scalar butterflies in
 * src/fft/butterflies/butterfly{2,3,4,5,7,8}/mod.rs, real<->complex passes in
 * src/fft/real_complex/mod.rs) and ResamplerFft / FftResampler (src/resampler_fft.rs).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference packs stage twiddles per SIMD width (radix_fft.rs:273-362); the values are the
 * same for every layout -- w_k(i) = exp(-2*pi*i * (i mod stride) * k / (stride*r)) computed in
 * f64 and rounded to f32 -- so the oracle indexes them directly.  Arithmetic follows the scalar
 * butterfly specs (the AVX kernels fuse some multiply-adds; both are within the 1e-6 test
 * tolerances of the reference's own SIMD-vs-scalar tests, butterflies/mod.rs:129-290).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef orc_c32 c32;

static inline c32 c_new(float re, float im) { c32 r = {re, im}; return r; }
static inline c32 c_add(c32 a, c32 b) { return c_new(a.re + b.re, a.im + b.im); }
static inline c32 c_sub(c32 a, c32 b) { return c_new(a.re - b.re, a.im - b.im); }
/* fft/mod.rs:52-57 */
static inline c32 c_mul(c32 a, c32 b) {
    return c_new(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}

/* ---- lib.rs:167-254: SampleRate families -------------------------------------------------- */
static int family_of(uint32_t hz, uint32_t* family_hz) {
    switch (hz) {
        case 22050: case 44100: case 88200: case 176400: *family_hz = 22050; return 0;
        case 16000: case 32000: *family_hz = 16000; return 0;
        case 48000: case 96000: case 192000: case 384000: *family_hz = 48000; return 0;
        default: return -1;
    }
}

/* planner.rs:183-207 */
static size_t decompose_multiplier(size_t multiplier, int* out) {
    if (multiplier == 1) return 0;
    size_t bits = 0;
    while (((size_t)1 << bits) < multiplier) bits++;
    size_t n8 = bits / 3, rem = bits % 3, n = 0;
    for (size_t i = 0; i < n8; i++) out[n++] = 8;
    if (rem == 1) out[n++] = 2;
    if (rem == 2) out[n++] = 4;
    return n;
}

/* planner.rs:35-245 */
int orc_fft_plan(uint32_t in_hz, uint32_t out_hz, size_t* fft_size_in, size_t* fft_size_out,
                 int* factors_in, size_t* n_factors_in, int* factors_out, size_t* n_factors_out,
                 int scale_for_throughput) {
    uint32_t fam_in, fam_out;
    if (family_of(in_hz, &fam_in) || family_of(out_hz, &fam_out)) return -1;
    size_t mul_in = in_hz / fam_in, mul_out = out_hz / fam_out;
    size_t base_in, base_out, ni = 0, no = 0;
    static const int F_588[] = {3, 4, 7, 7}, F_1280[] = {4, 4, 4, 4, 5};
    static const int F_64[] = {2, 2, 2, 2, 2, 2}, F_192[] = {4, 4, 4, 3};
    static const int F_640[] = {2, 4, 4, 4, 5}, F_882[] = {2, 3, 3, 7, 7};
    static const int F_2[] = {2};
    const int *fi, *fo;
    size_t nfi, nfo;
#define SET(bi, a, na, bo, b, nb) do { base_in = bi; fi = a; nfi = na; base_out = bo; fo = b; nfo = nb; } while (0)
    if (fam_in == fam_out) SET(2, F_2, 1, 2, F_2, 1);
    else if (fam_in == 22050 && fam_out == 48000) SET(588, F_588, 4, 1280, F_1280, 5);
    else if (fam_in == 48000 && fam_out == 22050) SET(1280, F_1280, 5, 588, F_588, 4);
    else if (fam_in == 16000 && fam_out == 48000) SET(64, F_64, 6, 192, F_192, 4);
    else if (fam_in == 48000 && fam_out == 16000) SET(192, F_192, 4, 64, F_64, 6);
    else if (fam_in == 16000 && fam_out == 22050) SET(640, F_640, 5, 882, F_882, 5);
    else SET(882, F_882, 5, 640, F_640, 5);   /* 22050 -> 16000 */
#undef SET
    for (size_t i = 0; i < nfi; i++) factors_in[ni++] = fi[i];
    for (size_t i = 0; i < nfo; i++) factors_out[no++] = fo[i];
    ni += decompose_multiplier(mul_in, factors_in + ni);      /* planner.rs:161-171 */
    no += decompose_multiplier(mul_out, factors_out + no);
    size_t size_in = base_in * mul_in, size_out = base_out * mul_out;
    if (scale_for_throughput) {                                /* planner.rs:212-245 */
        float m = ceilf(512.0f / (float)size_in);
        if (m < 1.0f) m = 1.0f;
        size_t multiplier = (size_t)m, p2 = 1;
        while (p2 < multiplier) p2 <<= 1;
        size_in *= p2;
        size_out *= p2;
        ni += decompose_multiplier(p2, factors_in + ni);
        no += decompose_multiplier(p2, factors_out + no);
    }
    *fft_size_in = size_in;
    *fft_size_out = size_out;
    *n_factors_in = ni;
    *n_factors_out = no;
    return 0;
}

/* ---- optimizer.rs:6-64 -------------------------------------------------------------------- */
static void sort_desc_stable(int* f, size_t n) {     /* sort_by_key(Reverse(radix)) is stable */
    for (size_t i = 1; i < n; i++) {
        int v = f[i];
        size_t j = i;
        while (j > 0 && f[j - 1] < v) { f[j] = f[j - 1]; j--; }
        f[j] = v;
    }
}
static void sort_asc_stable(int* f, size_t n) {
    for (size_t i = 1; i < n; i++) {
        int v = f[i];
        size_t j = i;
        while (j > 0 && f[j - 1] > v) { f[j] = f[j - 1]; j--; }
        f[j] = v;
    }
}
static int try_transform(int* f, size_t* n, const int* remove, size_t nr, const int* add, size_t na) {
    int tmp[64];
    size_t tn = *n;
    memcpy(tmp, f, sizeof(int) * tn);
    for (size_t r = 0; r < nr; r++) {
        size_t pos = tn;
        for (size_t i = 0; i < tn; i++) if (tmp[i] == remove[r]) { pos = i; break; }
        if (pos == tn) return 0;
        memmove(tmp + pos, tmp + pos + 1, sizeof(int) * (tn - pos - 1));
        tn--;
    }
    for (size_t a = 0; a < na; a++) tmp[tn++] = add[a];
    memcpy(f, tmp, sizeof(int) * tn);
    *n = tn;
    return 1;
}
size_t orc_optimize_factors(int* f, size_t n) {
    static const int R0[] = {4, 2}, A0[] = {8};
    static const int R1[] = {2, 2, 2}, A1[] = {8};
    static const int R2[] = {4, 4}, A2[] = {8, 2};
    static const int R3[] = {2, 2}, A3[] = {4};
    sort_desc_stable(f, n);
    for (;;) {
        int changed = try_transform(f, &n, R0, 2, A0, 1) || try_transform(f, &n, R1, 3, A1, 1) ||
                      try_transform(f, &n, R2, 2, A2, 2) || try_transform(f, &n, R3, 2, A3, 1);
        if (!changed) break;
        sort_desc_stable(f, n);
    }
    sort_asc_stable(f, n);
    return n;
}

/* ---- radix_fft.rs -------------------------------------------------------------------------- */
struct orc_rfft {
    size_t n, n2;
    int factors[32];
    size_t n_factors;
    int inverse;
    c32* rc_twiddles;     /* expansion (forward) or reduction (inverse) twiddles, k = 1.. */
    size_t n_rc;
    c32* scratch;         /* 3 * n */
    c32* stage_tw[24];    /* per stage (stride > 1): [k][q - 1] = twiddle_f32(k * q, stride * r), k < stride, q = 1 .. r - 1
                           * -- computed once here as RadixFFT::new does (radix_fft.rs:273-362), in f64 -> f32 */
    int simd;             /* 1: the reference's AVX + FMA butterflies and real <-> complex passes (fft_avx.c) */
    c32* packed_tw[24];   /* simd: the same twiddles packed per SIMD width, [group of 4 butterflies][q - 1][4] (:273-362) */
};
/* fft_avx.c */
size_t orc_avx_stage3(const c32* src, c32* dst, size_t n, size_t stride, const c32* tw);
size_t orc_avx_stage4(const c32* src, c32* dst, size_t n, size_t stride, const c32* tw);
size_t orc_avx_stage5(const c32* src, c32* dst, size_t n, size_t stride, const c32* tw);
size_t orc_avx_stage7(const c32* src, c32* dst, size_t n, size_t stride, const c32* tw);
size_t orc_avx_stage8(const c32* src, c32* dst, size_t n, size_t stride, const c32* tw);
size_t orc_avx_real_complex(c32* lm, c32* right, size_t rm_len, const c32* twiddles, size_t iters, int forward);

/* radix_fft.rs:251-258 */
static c32 twiddle_f32(size_t index, size_t fft_len) {
    double constant = -2.0 * 3.14159265358979323846264338327950288 / (double)fft_len;
    double angle = constant * (double)index;
    return c_new((float)cos(angle), (float)sin(angle));
}

/* radix_fft.rs:222-246 */
static size_t compute_factors(const int* in, size_t n, int* out) {
    if (n == 1) {
        if (in[0] == 2) return 0;
        if (in[0] == 4) { out[0] = 2; return 1; }
        if (in[0] == 8) { out[0] = 4; return 1; }
        return (size_t)-1;
    }
    memcpy(out, in, sizeof(int) * n);
    for (size_t i = 0; i < n; i++) if (out[i] == 2) {
        memmove(out + i, out + i + 1, sizeof(int) * (n - i - 1));
        return n - 1;
    }
    for (size_t i = 0; i < n; i++) if (out[i] == 8) { out[i] = 4; return n; }
    for (size_t i = 0; i < n; i++) if (out[i] == 4) { out[i] = 2; return n; }
    return (size_t)-1;
}

orc_rfft* orc_rfft_new(const int* factors, size_t n_factors, int inverse) { return orc_rfft_new_simd(factors, n_factors, inverse, 0); }

orc_rfft* orc_rfft_new_simd(const int* factors, size_t n_factors, int inverse, int simd) {
    if (n_factors == 0 || n_factors > 24) return NULL;
    size_t n = 1;
    for (size_t i = 0; i < n_factors; i++) {
        int r = factors[i];
        if (!(r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 8)) return NULL;
        n *= (size_t)r;
    }
    if (n % 2) return NULL;                                   /* radix_fft.rs:109-113 */
    orc_rfft* f = (orc_rfft*)calloc(1, sizeof(orc_rfft));
    f->n = n;
    f->n2 = n / 2;
    f->inverse = inverse;
    f->simd = simd && orc_have_avx_fma();
    size_t nf = compute_factors(factors, n_factors, f->factors);
    if (nf == (size_t)-1) { free(f); return NULL; }
    f->n_factors = orc_optimize_factors(f->factors, nf);      /* radix_fft.rs:117-118 */
    size_t count = (n % 4 == 0) ? n / 4 : n / 4 + 1;          /* radix_fft.rs:366-372 */
    f->n_rc = count > 0 ? count - 1 : 0;
    f->rc_twiddles = (c32*)malloc(sizeof(c32) * (f->n_rc + 1));
    for (size_t k = 1; k < count; k++) {
        c32 t = twiddle_f32(k, n);
        f->rc_twiddles[k - 1] = inverse ? c_new(t.re, -t.im)              /* :388-397 */
                                        : c_new(t.re * 0.5f, t.im * 0.5f); /* :377-386 */
    }
    f->scratch = (c32*)calloc(3 * n, sizeof(c32));
    size_t stride = 1;
    for (size_t s = 0; s < f->n_factors; s++) {
        const size_t r = (size_t)f->factors[s];
        if (stride > 1) {
            c32* tw = (c32*)malloc(sizeof(c32) * stride * (r - 1));
            for (size_t k = 0; k < stride; k++)
                for (size_t q = 1; q < r; q++) tw[k * (r - 1) + (q - 1)] = twiddle_f32(k * q, stride * r);
            f->stage_tw[s] = tw;
            if (f->simd && r != 2) {
                const size_t m = f->n2 / r, simd_iters = (m >> 2) << 2;
                c32* pk = (c32*)malloc(sizeof(c32) * (simd_iters ? simd_iters : 1) * (r - 1));
                for (size_t i = 0; i < simd_iters; i++)
                    for (size_t q = 1; q < r; q++)
                        pk[(i / 4) * 4 * (r - 1) + 4 * (q - 1) + (i % 4)] = tw[(i % stride) * (r - 1) + (q - 1)];
                f->packed_tw[s] = pk;
            }
        }
        stride *= r;
    }
    return f;
}

void orc_rfft_free(orc_rfft* f) {
    if (!f) return;
    free(f->rc_twiddles);
    free(f->scratch);
    for (size_t s = 0; s < 24; s++) { free(f->stage_tw[s]); free(f->packed_tw[s]); }
    free(f);
}
size_t orc_rfft_len(const orc_rfft* f) { return f->n; }
size_t orc_rfft_stage_factors(const orc_rfft* f, int* out) {
    memcpy(out, f->factors, sizeof(int) * f->n_factors);
    return f->n_factors;
}

/* One out-of-place Stockham stage of radix r (scalar butterfly specs, butterflyN/mod.rs). */
static inline __attribute__((always_inline)) void stage_body(const c32* src, c32* dst, size_t n, const int r, size_t stride, const c32* tw, size_t i0) {
    const size_t m = n / (size_t)r;
    size_t k = i0 % stride;   /* i % stride */
    for (size_t i = i0; i < m; i++, k = (k + 1 == stride) ? 0 : k + 1) {
        c32 z[8], t[8];
        for (int q = 0; q < r; q++) z[q] = src[i + (size_t)q * m];
        t[0] = z[0];
        for (int q = 1; q < r; q++)
            t[q] = (stride == 1) ? z[q] : c_mul(tw[k * (size_t)(r - 1) + (size_t)(q - 1)], z[q]);
        const size_t j = (size_t)r * i - (size_t)(r - 1) * k;
        c32* o = dst + j;
        switch (r) {
            case 2: {                                          /* butterfly2/mod.rs:233-269 */
                o[0] = c_add(z[0], t[1]);
                o[stride] = c_sub(z[0], t[1]);
            } break;
            case 3: {                                          /* butterfly3/mod.rs:231-349 */
                const float SQRT3_2 = 0.8660254f;
                c32 sum_t = c_add(t[1], t[2]), diff_t = c_sub(t[1], t[2]);
                o[0] = c_add(z[0], sum_t);
                float re_part = z[0].re - 0.5f * sum_t.re, im_part = z[0].im - 0.5f * sum_t.im;
                float sre = SQRT3_2 * diff_t.im, sim = -SQRT3_2 * diff_t.re;
                o[stride] = c_new(re_part + sre, im_part + sim);
                o[2 * stride] = c_new(re_part - sre, im_part - sim);
            } break;
            case 4: {                                          /* butterfly4/mod.rs:233-353 */
                c32 a0 = c_add(z[0], t[2]), a1 = c_sub(z[0], t[2]), a2 = c_add(t[1], t[3]);
                float a3_re = t[1].im - t[3].im, a3_im = t[3].re - t[1].re;
                o[0] = c_add(a0, a2);
                o[2 * stride] = c_sub(a0, a2);
                o[stride] = c_new(a1.re + a3_re, a1.im + a3_im);
                o[3 * stride] = c_new(a1.re - a3_re, a1.im - a3_im);
            } break;
            case 5: {                                          /* butterfly5/mod.rs:234-430 */
                const float C1 = 0.309017f, S1 = 0.95105654f, C2 = -0.809017f, S2 = 0.58778524f;
                c32 sum_all = c_add(c_add(c_add(t[1], t[2]), t[3]), t[4]);
                c32 a1 = c_add(t[1], t[4]), a2 = c_add(t[2], t[3]);
                float b1_re = t[1].im - t[4].im, b1_im = t[4].re - t[1].re;
                float b2_re = t[2].im - t[3].im, b2_im = t[3].re - t[2].re;
                float c1_re = z[0].re + C1 * a1.re + C2 * a2.re, c1_im = z[0].im + C1 * a1.im + C2 * a2.im;
                float c2_re = z[0].re + C2 * a1.re + C1 * a2.re, c2_im = z[0].im + C2 * a1.im + C1 * a2.im;
                float d1_re = S1 * b1_re + S2 * b2_re, d1_im = S1 * b1_im + S2 * b2_im;
                float d2_re = S2 * b1_re - S1 * b2_re, d2_im = S2 * b1_im - S1 * b2_im;
                o[0] = c_add(z[0], sum_all);
                o[stride] = c_new(c1_re + d1_re, c1_im + d1_im);
                o[2 * stride] = c_new(c2_re + d2_re, c2_im + d2_im);
                o[3 * stride] = c_new(c2_re - d2_re, c2_im - d2_im);
                o[4 * stride] = c_new(c1_re - d1_re, c1_im - d1_im);
            } break;
            case 7: {                                          /* butterfly7/mod.rs:237-522 */
                const float C[3] = {0.6234898f, -0.22252093f, -0.90096885f};
                const float S[3] = {0.7818315f, 0.9749279f, 0.43388373f};
                /* (cos index, sin index, sin sign) triples per output, butterfly7/mod.rs:416-436 */
                static const int CI[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {2, 0, 1}, {1, 2, 0}, {0, 1, 2}};
                static const int SS[6][3] = {{1, 1, 1}, {1, -1, -1}, {1, -1, 1}, {-1, 1, -1}, {-1, 1, 1}, {-1, -1, -1}};
                c32 sum_all = c_add(c_add(c_add(c_add(c_add(t[1], t[2]), t[3]), t[4]), t[5]), t[6]);
                c32 a1 = c_add(t[1], t[6]), a2 = c_add(t[2], t[5]), a3 = c_add(t[3], t[4]);
                float b_re[3] = {t[1].im - t[6].im, t[2].im - t[5].im, t[3].im - t[4].im};
                float b_im[3] = {t[6].re - t[1].re, t[5].re - t[2].re, t[4].re - t[3].re};
                o[0] = c_add(z[0], sum_all);
                for (int idx = 1; idx < 7; idx++) {
                    const int* ci = CI[idx - 1];
                    const int* ss = SS[idx - 1];
                    float cos1 = C[ci[0]], cos2 = C[ci[1]], cos3 = C[ci[2]];
                    float sin1 = (float)ss[0] * S[ci[0]], sin2 = (float)ss[1] * S[ci[1]],
                          sin3 = (float)ss[2] * S[ci[2]];
                    float c_re = z[0].re + cos1 * a1.re + cos2 * a2.re + cos3 * a3.re;
                    float c_im = z[0].im + cos1 * a1.im + cos2 * a2.im + cos3 * a3.im;
                    float d_re = sin1 * b_re[0] + sin2 * b_re[1] + sin3 * b_re[2];
                    float d_im = sin1 * b_im[0] + sin2 * b_im[1] + sin3 * b_im[2];
                    o[(size_t)idx * stride] = c_new(c_re + d_re, c_im + d_im);
                }
            } break;
            case 8: {                                          /* butterfly8/mod.rs:239-586 */
                const float H = 0.70710678118654752440f;      /* FRAC_1_SQRT_2 */
                c32 ea0 = c_add(z[0], t[4]), ea1 = c_sub(z[0], t[4]), ea2 = c_add(t[2], t[6]);
                float ea3_re = t[2].im - t[6].im, ea3_im = t[6].re - t[2].re;
                c32 xe0 = c_add(ea0, ea2), xe2 = c_sub(ea0, ea2);
                c32 xe1 = c_new(ea1.re + ea3_re, ea1.im + ea3_im), xe3 = c_new(ea1.re - ea3_re, ea1.im - ea3_im);
                c32 oa0 = c_add(t[1], t[5]), oa1 = c_sub(t[1], t[5]), oa2 = c_add(t[3], t[7]);
                float oa3_re = t[3].im - t[7].im, oa3_im = t[7].re - t[3].re;
                c32 xo0 = c_add(oa0, oa2), xo2 = c_sub(oa0, oa2);
                c32 xo1 = c_new(oa1.re + oa3_re, oa1.im + oa3_im), xo3 = c_new(oa1.re - oa3_re, oa1.im - oa3_im);
                o[0] = c_add(xe0, xo0);
                o[4 * stride] = c_sub(xe0, xo0);
                c32 w1 = c_new(H * (xo1.re + xo1.im), H * (xo1.im - xo1.re));
                o[stride] = c_add(xe1, w1);
                o[5 * stride] = c_sub(xe1, w1);
                c32 w2 = c_new(xo2.im, -xo2.re);
                o[2 * stride] = c_add(xe2, w2);
                o[6 * stride] = c_sub(xe2, w2);
                c32 w3 = c_new(H * (xo3.im - xo3.re), -H * (xo3.re + xo3.im));
                o[3 * stride] = c_add(xe3, w3);
                o[7 * stride] = c_sub(xe3, w3);
            } break;
        }
    }
}

/* One instance per radix (the reference has one function per radix too, butterflies/mod.rs): with `r` a constant the
 * compiler unrolls the loads and drops the switch; the arithmetic and its order are unchanged. */
/* i0: the butterflies [0, i0) have been done already (by the AVX path, whose leftovers are the scalar spec's) */
static void stage_from(const c32* src, c32* dst, size_t n, int r, size_t stride, const c32* tw, size_t i0) {
    switch (r) {
        case 2: stage_body(src, dst, n, 2, stride, tw, i0); break;
        case 3: stage_body(src, dst, n, 3, stride, tw, i0); break;
        case 4: stage_body(src, dst, n, 4, stride, tw, i0); break;
        case 5: stage_body(src, dst, n, 5, stride, tw, i0); break;
        case 7: stage_body(src, dst, n, 7, stride, tw, i0); break;
        case 8: stage_body(src, dst, n, 8, stride, tw, i0); break;
    }
}
static void stage(const c32* src, c32* dst, size_t n, int r, size_t stride, const c32* tw) { stage_from(src, dst, n, r, stride, tw, 0); }
/* a stage of the plan: the AVX + FMA butterflies where the plan asks for them (radix 2 has no stage of its own in the
 * plans of the resampler: scalar) */
static void plan_stage(const orc_rfft* f, size_t s, const c32* src, c32* dst, size_t stride) {
    const int r = f->factors[s];
    size_t done = 0;
    if (f->simd && (stride == 1 || f->packed_tw[s])) {
        const c32* pk = f->packed_tw[s];
        switch (r) {
            case 3: done = orc_avx_stage3(src, dst, f->n2, stride, pk); break;
            case 4: done = orc_avx_stage4(src, dst, f->n2, stride, pk); break;
            case 5: done = orc_avx_stage5(src, dst, f->n2, stride, pk); break;
            case 7: done = orc_avx_stage7(src, dst, f->n2, stride, pk); break;
            case 8: done = orc_avx_stage8(src, dst, f->n2, stride, pk); break;
            default: break;
        }
    }
    stage_from(src, dst, f->n2, r, stride, f->stage_tw[s], done);
}

/* One stage on caller-owned arrays: what the reference's own butterfly tests call (butterfly8/mod.rs:616-697,
 * butterflies/mod.rs:129-290).  tw: (radix - 1) twiddles per column, unused when stride == 1. */
void orc_butterfly_stage(const orc_c32* src, orc_c32* dst, size_t n, int radix, size_t stride, const orc_c32* tw) {
    stage((const c32*)src, (c32*)dst, n, radix, stride, (const c32*)tw);
}

/* stockham_autosort.rs:169-247 (ping-pong; returns 1 when the result is in `scratch`), with the
 * single-factor special case of radix_fft.rs:476-497 (result copied back to data). */
static int stockham(const orc_rfft* f, c32* data, c32* scratch) {
    if (f->n_factors == 0) return 0;
    if (f->n_factors == 1) {
        plan_stage(f, 0, data, scratch, 1);
        memcpy(data, scratch, sizeof(c32) * f->n2);
        return 0;
    }
    c32 *in = data, *out = scratch;
    size_t stride = 1;
    for (size_t s = 0; s < f->n_factors; s++) {
        plan_stage(f, s, in, out, stride);
        c32* tmp = in; in = out; out = tmp;
        stride *= (size_t)f->factors[s];
    }
    return (f->n_factors % 2) ? 1 : 0;
}

/* radix_fft.rs:540-562 + :500-537 + real_complex/mod.rs:37-74 */
void orc_rfft_forward(orc_rfft* f, const float* in, orc_c32* out) {
    c32* data = f->scratch;
    c32* scratch = f->scratch + f->n2;
    memcpy(data, in, sizeof(float) * f->n);          /* reinterpret n reals as n2 complexes */
    const c32* res = stockham(f, data, scratch) ? scratch : data;
    size_t len = f->n2 + 1;
    memcpy(out, res, sizeof(c32) * f->n2);
    size_t split = len / 2;
    c32 *left = out, *right = out + split;
    size_t right_len = len - split;
    if (split == 0) return;
    c32 z0 = left[0];
    left[0] = c_new(z0.re + z0.im, 0.0f);
    right[right_len - 1] = c_new(z0.re - z0.im, 0.0f);
    c32* lm = left + 1;
    size_t lm_len = split - 1, rm_len = right_len - 1;
    size_t iters = lm_len < rm_len ? lm_len : rm_len;
    if (f->n_rc < iters) iters = f->n_rc;
    for (size_t i = f->simd ? orc_avx_real_complex(lm, right, rm_len, f->rc_twiddles, iters, 1) : 0; i < iters; i++) {
        c32 o = lm[i];
        size_t rev = rm_len - 1 - i;
        c32 orv = right[rev];
        c32 tw = f->rc_twiddles[i];
        c32 sum = c_add(o, orv), diff = c_sub(o, orv);
        float tre_sum_im = sum.im * tw.re, tim_sum_im = sum.im * tw.im;
        float tre_diff_re = diff.re * tw.re, tim_diff_re = diff.re * tw.im;
        float half_sum_real = 0.5f * sum.re, half_diff_imag = 0.5f * diff.im;
        float real = tre_sum_im + tim_diff_re;
        float imag = tim_sum_im - tre_diff_re;
        lm[i] = c_new(half_sum_real + real, half_diff_imag + imag);
        right[rev] = c_new(half_sum_real - real, imag - half_diff_imag);
    }
    if (len % 2 == 1) out[len / 2].im = -out[len / 2].im;
}

/* radix_fft.rs:565-589 + :592-670 + real_complex/mod.rs:84-114 */
void orc_rfft_inverse(orc_rfft* f, const orc_c32* in, float* out) {
    c32* data = f->scratch;
    c32* scratch = f->scratch + f->n2;
    size_t len = f->n2 + 1;
    memcpy(data, in, sizeof(c32) * len);   /* data[n2] spills into scratch[0], as the reference */
    size_t split = len / 2;
    c32 *left = data, *right = data + split;
    size_t right_len = len - split;
    if (split > 0) {
        c32 first_sum = c_add(left[0], right[right_len - 1]);
        c32 first_diff = c_sub(left[0], right[right_len - 1]);
        left[0] = c_new(first_sum.re - first_sum.im, first_diff.re - first_diff.im);
        c32* lm = left + 1;
        size_t lm_len = split - 1, rm_len = right_len - 1;
        size_t iters = lm_len < rm_len ? lm_len : rm_len;
        if (f->n_rc < iters) iters = f->n_rc;
        for (size_t i = f->simd ? orc_avx_real_complex(lm, right, rm_len, f->rc_twiddles, iters, 0) : 0; i < iters; i++) {
            c32 a = lm[i];
            size_t rev = rm_len - 1 - i;
            c32 b = right[rev];
            c32 tw = f->rc_twiddles[i];
            c32 sum = c_add(a, b), diff = c_sub(a, b);
            float tre_sum_im = sum.im * tw.re, tim_sum_im = sum.im * tw.im;
            float tre_diff_re = diff.re * tw.re, tim_diff_re = diff.re * tw.im;
            float real = tre_sum_im + tim_diff_re;
            float imag = tim_sum_im - tre_diff_re;
            lm[i] = c_new(sum.re - real, diff.im - imag);
            right[rev] = c_new(sum.re + real, -imag - diff.im);
        }
        if (len % 2 == 1) {
            c32 c = data[len / 2];
            c32 dbl = c_add(c, c);
            data[len / 2] = c_new(dbl.re, -dbl.im);
        }
    }
    for (size_t i = 0; i < f->n2; i++) data[i].im = -data[i].im;           /* :634-637 */
    int in_scratch = stockham(f, data, scratch);
    c32* res = in_scratch ? scratch : data;
    for (size_t i = 0; i < f->n2; i++) res[i].im = -res[i].im;             /* :656-669 */
    memcpy(out, res, sizeof(c32) * f->n2);
}

/* ---- resampler_fft.rs ----------------------------------------------------------------------- */
struct orc_fft_resampler {
    size_t channels, fft_in, fft_out;
    orc_rfft *fft, *ifft;
    c32* filter_spectrum;      /* fft_in + 1 */
    float* overlaps;           /* fft_out * channels */
    float* input_scratch;      /* (chunk_in + fft_in) * channels */
    float* output_scratch;     /* (chunk_out + fft_out) * channels */
    size_t output_scratch_len;
    float *input_buffer, *output_buffer;
    c32 *input_spectrum, *output_spectrum;
};

orc_fft_resampler* orc_fft_new(size_t channels, uint32_t in_hz, uint32_t out_hz) { return orc_fft_new_simd(channels, in_hz, out_hz, 0); }

/* simd != 0: the transforms (and so the filter spectrum) through the reference's AVX + FMA code path (fft_avx.c) */
orc_fft_resampler* orc_fft_new_simd(size_t channels, uint32_t in_hz, uint32_t out_hz, int simd) {
    size_t fft_in, fft_out, ni, no;
    int fin[32], fout[32];
    if (channels == 0 || orc_fft_plan(in_hz, out_hz, &fft_in, &fft_out, fin, &ni, fout, &no, 1)) return NULL;
    orc_fft_resampler* r = (orc_fft_resampler*)calloc(1, sizeof(*r));
    r->channels = channels;
    r->fft_in = fft_in;
    r->fft_out = fft_out;
    fin[ni++] = 2;                                           /* resampler_fft.rs:345-348 */
    fout[no++] = 2;
    r->fft = orc_rfft_new_simd(fin, ni, 0, simd);
    r->ifft = orc_rfft_new_simd(fout, no, 1, simd);
    /* resampler_fft.rs:353-359 */
    double cutoff = fft_in > fft_out
        ? orc_calculate_cutoff_kaiser(fft_out, 10.0) * ((double)fft_out / (double)fft_in)
        : orc_calculate_cutoff_kaiser(fft_in, 10.0);
    float* sincs = (float*)malloc(sizeof(float) * fft_in);
    orc_make_sincs_for_kaiser(fft_in, 1, (float)cutoff, 10.0, ORC_WINDOW_PERIODIC, sincs);
    float* filter_time = (float*)calloc(2 * fft_in, sizeof(float));
    for (size_t i = 0; i < fft_in; i++) filter_time[i] = sincs[i] / (float)(2 * fft_in);  /* :371-373 */
    r->filter_spectrum = (c32*)calloc(fft_in + 1, sizeof(c32));
    orc_rfft_forward(r->fft, filter_time, r->filter_spectrum);
    free(filter_time);
    free(sincs);
    size_t chunk_in = fft_in * channels, chunk_out = fft_out * channels;
    r->overlaps = (float*)calloc(fft_out * channels, sizeof(float));
    r->input_scratch = (float*)calloc((chunk_in + fft_in) * channels, sizeof(float));
    r->output_scratch_len = (chunk_out + fft_out) * channels;
    r->output_scratch = (float*)calloc(r->output_scratch_len, sizeof(float));
    r->input_buffer = (float*)calloc(2 * fft_in, sizeof(float));
    r->output_buffer = (float*)calloc(2 * fft_out, sizeof(float));
    r->input_spectrum = (c32*)calloc(fft_in + 1, sizeof(c32));
    r->output_spectrum = (c32*)calloc(fft_out + 1, sizeof(c32));
    return r;
}

void orc_fft_free(orc_fft_resampler* r) {
    if (!r) return;
    orc_rfft_free(r->fft);
    orc_rfft_free(r->ifft);
    free(r->filter_spectrum); free(r->overlaps); free(r->input_scratch); free(r->output_scratch);
    free(r->input_buffer); free(r->output_buffer); free(r->input_spectrum); free(r->output_spectrum);
    free(r);
}

size_t orc_fft_chunk_size_input(const orc_fft_resampler* r) { return r->fft_in * r->channels; }
size_t orc_fft_chunk_size_output(const orc_fft_resampler* r) { return r->fft_out * r->channels; }
size_t orc_fft_delay(const orc_fft_resampler* r) { return r->fft_in / 2; }
const orc_c32* orc_fft_filter_spectrum(const orc_fft_resampler* r, size_t* len) {
    *len = r->fft_in + 1;
    return r->filter_spectrum;
}

/* FftResampler::resample (resampler_fft.rs:385-424) */
static void block(orc_fft_resampler* r, const float* wave_in, float* wave_out, float* overlap) {
    size_t fi = r->fft_in, fo = r->fft_out;
    memcpy(r->input_buffer, wave_in, sizeof(float) * fi);
    memset(r->input_buffer + fi, 0, sizeof(float) * fi);
    orc_rfft_forward(r->fft, r->input_buffer, r->input_spectrum);
    size_t new_length = fi < fo ? fi + 1 : fo;
    for (size_t i = 0; i < new_length; i++)
        r->input_spectrum[i] = c_mul(r->input_spectrum[i], r->filter_spectrum[i]);
    memcpy(r->output_spectrum, r->input_spectrum, sizeof(c32) * new_length);
    memset(r->output_spectrum + new_length, 0, sizeof(c32) * (fo + 1 - new_length));
    orc_rfft_inverse(r->ifft, r->output_spectrum, r->output_buffer);
    for (size_t i = 0; i < fo; i++) wave_out[i] = r->output_buffer[i] + overlap[i];
    memcpy(overlap, r->output_buffer + fo, sizeof(float) * fo);
}

/* ResamplerFft::resample (resampler_fft.rs:182-240), literal scratch indexing: channel c's block
 * is written at c * output_scratch_ch_size(), where output_scratch_ch_size() returns the INPUT
 * per-channel size (:127-129).  Returns 0 ok, 1/2 buffer-size errors (error.rs), 3 where the
 * reference would index out of range (a panic). */
int orc_fft_resample(orc_fft_resampler* r, const float* in, size_t in_len, float* out,
                     size_t out_len) {
    size_t ch = r->channels, fi = r->fft_in, fo = r->fft_out;
    if (in_len < fi * ch) return 1;
    if (out_len < fo * ch) return 2;
    size_t in_ch_len = fi * ch + fi;
    size_t out_ch_len = fi * ch + fi;                        /* :127-129 (sic) */
    for (size_t f = 0; f < fi; f++)
        for (size_t c = 0; c < ch; c++) r->input_scratch[c * in_ch_len + f] = in[f * ch + c];
    for (size_t c = 0; c < ch; c++) {
        if (c * out_ch_len + fo > r->output_scratch_len) return 3;
        block(r, r->input_scratch + c * in_ch_len, r->output_scratch + c * out_ch_len,
              r->overlaps + fo * c);
    }
    for (size_t f = 0; f < fo; f++)
        for (size_t c = 0; c < ch; c++) {
            if (c * out_ch_len + f >= r->output_scratch_len) return 3;
            out[f * ch + c] = r->output_scratch[c * out_ch_len + f];
        }
    return 0;
}

/* resample/src/main.rs:256-313 (resample_batch): complete chunks straight from the input, a last partial chunk
 * zero padded, the output trimmed to ceil(in_len * chunk_out / chunk_in).  Returns values written (0 on error or
 * when out_cap is too small). */
size_t orc_fft_resample_all(orc_fft_resampler* r, const float* in, size_t in_len, float* out, size_t out_cap) {
    const size_t ci = orc_fft_chunk_size_input(r), co = orc_fft_chunk_size_output(r);
    const size_t complete = in_len / ci, rem = in_len % ci, total = complete + (rem ? 1 : 0);
    if (out_cap < total * co) return 0;
    for (size_t k = 0; k < complete; k++)
        if (orc_fft_resample(r, in + k * ci, ci, out + k * co, co) != 0) return 0;
    if (rem) {
        float* padded = (float*)calloc(ci, sizeof(float));
        memcpy(padded, in + complete * ci, sizeof(float) * rem);
        const int rc = orc_fft_resample(r, padded, ci, out + complete * co, co);
        free(padded);
        if (rc != 0) return 0;
    }
    return (size_t)ceil((double)in_len * (double)co / (double)ci);
}
