/* oracle/fft_avx.c -- TEST INFRASTRUCTURE (the CPU oracle; never linked into the product).
 *
 * The reference's AVX + FMA code path of the FFT resampler, restated with the same intrinsics in the same order:
 *   butterflies/ops/avx.rs:5-65                 complex_mul_avx (moveldup / movehdup / permute / fmaddsub), mul by i, W8 helpers
 *   butterflies/butterfly3/avx.rs:100-207       radix 3 (fnmadd by 0.5, sqrt(3)/2 pattern)
 *   butterflies/butterfly4/avx.rs:100-213       radix 4
 *   butterflies/butterfly5/avx.rs:140-291       radix 5 (FMA chains c1, c2, d1, d2)
 *   butterflies/butterfly7/avx.rs:262-528       radix 7 (compute_output: two FMA chains per output)
 *   butterflies/butterfly8/avx.rs:185-378       radix 8 (two radix-4 + W8 combine)
 *   real_complex/avx.rs:5-125, :129-238         postprocess_fft / preprocess_ifft
 * Four butterflies per iteration, twiddles packed per SIMD width ([group of 4][q - 1][4], radix_fft.rs:273-362), results
 * scattered with 64-bit stores to dst[r i - (r - 1) k + q stride]; the butterflies left over (count % 4) are the scalar
 * spec's (oracle/fft.c).  The reference's stride-1 variants differ from the generic ones only in skipping the identity
 * twiddles and in storing whole vectors: the same values.  Not bit-identical to the scalar path (FMA contraction, the
 * reference says so itself and tests the two against each other at 1e-6: butterflies/mod.rs:129-290); the oracle's tests
 * hold this file to the same tolerance against oracle/fft.c.  bench.py times it as the FFT's CPU baseline ("port-avx").
 */
#include <immintrin.h>
#include <stddef.h>

#include "oracle.h"

typedef orc_c32 c32;
#define AVX __attribute__((target("avx,fma")))

AVX static inline __m256 cmul(__m256 left, __m256 right) {   /* ops/avx.rs:5-11 */
    __m256 right_re = _mm256_moveldup_ps(right);
    __m256 right_im = _mm256_movehdup_ps(right);
    __m256 left_swap = _mm256_permute_ps(left, 0xB1);
    __m256 prod_im = _mm256_mul_ps(left_swap, right_im);
    return _mm256_fmaddsub_ps(left, right_re, prod_im);
}
AVX static inline __m256 neg_imag_mask(void) { return _mm256_set_ps(-0.0f, 0.0f, -0.0f, 0.0f, -0.0f, 0.0f, -0.0f, 0.0f); }
AVX static inline __m256 mul_i(__m256 v, __m256 mask) {       /* ops/avx.rs:31-34: (a + bi) -> (b, -a) */
    return _mm256_xor_ps(_mm256_shuffle_ps(v, v, 0xB1), mask);
}
AVX static inline __m256 w8x(__m256 xy, __m256 mask, __m256 scale) {   /* ops/avx.rs:48-53 */
    __m256 ymx = _mm256_xor_ps(_mm256_shuffle_ps(xy, xy, 0xB1), mask);
    return _mm256_mul_ps(scale, _mm256_add_ps(xy, ymx));
}
AVX static inline __m256 v8x(__m256 xy, __m256 mask, __m256 scale) {   /* ops/avx.rs:60-65 */
    __m256 ymx = _mm256_xor_ps(_mm256_shuffle_ps(xy, xy, 0xB1), mask);
    return _mm256_mul_ps(scale, _mm256_sub_ps(ymx, xy));
}
/* the four complex values of `v` to dst[j0 + off], dst[j1 + off], dst[j2 + off], dst[j3 + off] (64-bit stores) */
AVX static inline void scatter(c32* dst, const size_t* j, size_t off, __m256 v) {
    __m128d lo = _mm_castps_pd(_mm256_castps256_ps128(v)), hi = _mm_castps_pd(_mm256_extractf128_ps(v, 1));
    _mm_storel_pd((double*)(dst + j[0] + off), lo);
    _mm_storeh_pd((double*)(dst + j[1] + off), lo);
    _mm_storel_pd((double*)(dst + j[2] + off), hi);
    _mm_storeh_pd((double*)(dst + j[3] + off), hi);
}

#define LOAD(q) _mm256_loadu_ps((const float*)(src + i + (size_t)(q) * m))
#define TW(q) (stride == 1 ? LOAD(q) : cmul(_mm256_loadu_ps((const float*)(tw + i * (R - 1) + 4 * ((q) - 1))), LOAD(q)))
#define PROLOGUE(RADIX)                                                                         \
    enum { R = RADIX };                                                                         \
    const size_t m = n / R, simd = (m >> 2) << 2;                                               \
    for (size_t i = 0; i < simd; i += 4) {                                                      \
        const size_t k = i % stride;                                                            \
        size_t j[4];                                                                            \
        for (size_t l = 0; l < 4; l++) {                                                        \
            size_t kl = k + l; while (kl >= stride) kl -= stride;                               \
            j[l] = R * (i + l) - (R - 1) * kl;                                                  \
        }

/* Each stage returns the number of butterflies it did (a multiple of 4); `tw`: packed twiddles, unused for stride 1. */
AVX size_t orc_avx_stage3(const c32* src, c32* dst, size_t n, size_t stride, const c32* tw) {
    const float S = 0.8660254f;   /* SQRT3_2, butterfly3/mod.rs:47 */
    const __m256 pat = _mm256_set_ps(-S, S, -S, S, -S, S, -S, S), half = _mm256_set1_ps(0.5f);
    PROLOGUE(3)
        __m256 z0 = LOAD(0), t1 = TW(1), t2 = TW(2);
        __m256 sum_t = _mm256_add_ps(t1, t2), diff_t = _mm256_sub_ps(t1, t2);
        __m256 out0 = _mm256_add_ps(z0, sum_t);
        __m256 part = _mm256_fnmadd_ps(sum_t, half, z0);
        __m256 sd = _mm256_mul_ps(_mm256_shuffle_ps(diff_t, diff_t, 0xB1), pat);
        scatter(dst, j, 0, out0);
        scatter(dst, j, stride, _mm256_add_ps(part, sd));
        scatter(dst, j, 2 * stride, _mm256_sub_ps(part, sd));
    }
    return simd;
}
AVX size_t orc_avx_stage4(const c32* src, c32* dst, size_t n, size_t stride, const c32* tw) {
    const __m256 mask = neg_imag_mask();
    PROLOGUE(4)
        __m256 z0 = LOAD(0), t1 = TW(1), t2 = TW(2), t3 = TW(3);
        __m256 a0 = _mm256_add_ps(z0, t2), a1 = _mm256_sub_ps(z0, t2), a2 = _mm256_add_ps(t1, t3);
        __m256 a3 = mul_i(_mm256_sub_ps(t1, t3), mask);
        scatter(dst, j, 0, _mm256_add_ps(a0, a2));
        scatter(dst, j, stride, _mm256_add_ps(a1, a3));
        scatter(dst, j, 2 * stride, _mm256_sub_ps(a0, a2));
        scatter(dst, j, 3 * stride, _mm256_sub_ps(a1, a3));
    }
    return simd;
}
AVX size_t orc_avx_stage5(const c32* src, c32* dst, size_t n, size_t stride, const c32* tw) {
    /* butterfly5/mod.rs:47-50 */
    const __m256 c25 = _mm256_set1_ps(0.30901699f), s25 = _mm256_set1_ps(0.95105652f);
    const __m256 c45 = _mm256_set1_ps(-0.80901699f), s45 = _mm256_set1_ps(0.58778525f);
    const __m256 mask = neg_imag_mask();
    PROLOGUE(5)
        __m256 z0 = LOAD(0), t1 = TW(1), t2 = TW(2), t3 = TW(3), t4 = TW(4);
        __m256 sum_all = _mm256_add_ps(_mm256_add_ps(_mm256_add_ps(t1, t2), t3), t4);
        __m256 a1 = _mm256_add_ps(t1, t4), a2 = _mm256_add_ps(t2, t3);
        __m256 b1 = mul_i(_mm256_sub_ps(t1, t4), mask), b2 = mul_i(_mm256_sub_ps(t2, t3), mask);
        __m256 c1 = _mm256_fmadd_ps(c45, a2, _mm256_fmadd_ps(c25, a1, z0));
        __m256 c2 = _mm256_fmadd_ps(c25, a2, _mm256_fmadd_ps(c45, a1, z0));
        __m256 d1 = _mm256_fmadd_ps(s45, b2, _mm256_mul_ps(s25, b1));
        __m256 d2 = _mm256_fmsub_ps(s45, b1, _mm256_mul_ps(s25, b2));
        scatter(dst, j, 0, _mm256_add_ps(z0, sum_all));
        scatter(dst, j, stride, _mm256_add_ps(c1, d1));
        scatter(dst, j, 2 * stride, _mm256_add_ps(c2, d2));
        scatter(dst, j, 3 * stride, _mm256_sub_ps(c2, d2));
        scatter(dst, j, 4 * stride, _mm256_sub_ps(c1, d1));
    }
    return simd;
}
#define OUT7(ca, sa, cb, sb, cc, sc)                                                                                  \
    _mm256_add_ps(_mm256_fmadd_ps(cc, a3, _mm256_fmadd_ps(cb, a2, _mm256_fmadd_ps(ca, a1, z0))),                      \
                  _mm256_fmadd_ps(sc, b3, _mm256_fmadd_ps(sb, b2, _mm256_mul_ps(sa, b1))))
AVX size_t orc_avx_stage7(const c32* src, c32* dst, size_t n, size_t stride, const c32* tw) {
    /* butterfly7/mod.rs:47-52 */
    const float C1 = 0.62348980f, S1 = 0.78183148f, C2 = -0.22252093f, S2 = 0.97492791f, C3 = -0.90096887f, S3 = 0.43388374f;
    const __m256 c1 = _mm256_set1_ps(C1), s1 = _mm256_set1_ps(S1), c2 = _mm256_set1_ps(C2), s2 = _mm256_set1_ps(S2);
    const __m256 c3 = _mm256_set1_ps(C3), s3 = _mm256_set1_ps(S3);
    const __m256 n1 = _mm256_set1_ps(-S1), n2 = _mm256_set1_ps(-S2), n3 = _mm256_set1_ps(-S3);
    const __m256 mask = neg_imag_mask();
    PROLOGUE(7)
        __m256 z0 = LOAD(0), t1 = TW(1), t2 = TW(2), t3 = TW(3), t4 = TW(4), t5 = TW(5), t6 = TW(6);
        __m256 sum_all = _mm256_add_ps(_mm256_add_ps(_mm256_add_ps(t1, t2), _mm256_add_ps(t3, t4)), _mm256_add_ps(t5, t6));
        __m256 a1 = _mm256_add_ps(t1, t6), a2 = _mm256_add_ps(t2, t5), a3 = _mm256_add_ps(t3, t4);
        __m256 b1 = mul_i(_mm256_sub_ps(t1, t6), mask), b2 = mul_i(_mm256_sub_ps(t2, t5), mask), b3 = mul_i(_mm256_sub_ps(t3, t4), mask);
        scatter(dst, j, 0, _mm256_add_ps(z0, sum_all));
        scatter(dst, j, stride, OUT7(c1, s1, c2, s2, c3, s3));
        scatter(dst, j, 2 * stride, OUT7(c2, s2, c3, n3, c1, n1));
        scatter(dst, j, 3 * stride, OUT7(c3, s3, c1, n1, c2, s2));
        scatter(dst, j, 4 * stride, OUT7(c3, n3, c1, s1, c2, n2));
        scatter(dst, j, 5 * stride, OUT7(c2, n2, c3, s3, c1, s1));
        scatter(dst, j, 6 * stride, OUT7(c1, n1, c2, n2, c3, n3));
    }
    return simd;
}
AVX size_t orc_avx_stage8(const c32* src, c32* dst, size_t n, size_t stride, const c32* tw) {
    const __m256 mask = neg_imag_mask(), scale = _mm256_set1_ps(0.70710678118654752440f);
    PROLOGUE(8)
        __m256 z0 = LOAD(0), t1 = TW(1), t2 = TW(2), t3 = TW(3), t4 = TW(4), t5 = TW(5), t6 = TW(6), t7 = TW(7);
        __m256 ea0 = _mm256_add_ps(z0, t4), ea1 = _mm256_sub_ps(z0, t4), ea2 = _mm256_add_ps(t2, t6);
        __m256 ea3 = mul_i(_mm256_sub_ps(t2, t6), mask);
        __m256 xe0 = _mm256_add_ps(ea0, ea2), xe2 = _mm256_sub_ps(ea0, ea2), xe1 = _mm256_add_ps(ea1, ea3), xe3 = _mm256_sub_ps(ea1, ea3);
        __m256 oa0 = _mm256_add_ps(t1, t5), oa1 = _mm256_sub_ps(t1, t5), oa2 = _mm256_add_ps(t3, t7);
        __m256 oa3 = mul_i(_mm256_sub_ps(t3, t7), mask);
        __m256 xo0 = _mm256_add_ps(oa0, oa2), xo2 = _mm256_sub_ps(oa0, oa2), xo1 = _mm256_add_ps(oa1, oa3), xo3 = _mm256_sub_ps(oa1, oa3);
        __m256 w1 = w8x(xo1, mask, scale), w2 = mul_i(xo2, mask), w3 = v8x(xo3, mask, scale);
        scatter(dst, j, 0, _mm256_add_ps(xe0, xo0));
        scatter(dst, j, stride, _mm256_add_ps(xe1, w1));
        scatter(dst, j, 2 * stride, _mm256_add_ps(xe2, w2));
        scatter(dst, j, 3 * stride, _mm256_add_ps(xe3, w3));
        scatter(dst, j, 4 * stride, _mm256_sub_ps(xe0, xo0));
        scatter(dst, j, 5 * stride, _mm256_sub_ps(xe1, w1));
        scatter(dst, j, 6 * stride, _mm256_sub_ps(xe2, w2));
        scatter(dst, j, 7 * stride, _mm256_sub_ps(xe3, w3));
    }
    return simd;
}

/* real_complex/avx.rs:5-125 (forward = 1) and :129-238 (forward = 0): pairs (lm[i], right[rm_len - 1 - i]), i < iters;
 * returns how many pairs were done (a multiple of 4). */
AVX size_t orc_avx_real_complex(c32* lm, c32* right, size_t rm_len, const c32* twiddles, size_t iters, int forward) {
    const size_t simd = iters / 4;
    const __m256 half = _mm256_set1_ps(0.5f);
    for (size_t chunk = 0; chunk < simd; chunk++) {
        const size_t i = chunk * 4;
        __m256 a = _mm256_loadu_ps((const float*)(lm + i));
        c32* r0 = right + (rm_len - 1 - i);
        __m128d lo = _mm_loadh_pd(_mm_load_sd((const double*)r0), (const double*)(r0 - 1));
        __m128d hi = _mm_loadh_pd(_mm_load_sd((const double*)(r0 - 2)), (const double*)(r0 - 3));
        __m256 b = _mm256_set_m128(_mm_castpd_ps(hi), _mm_castpd_ps(lo));
        __m256 tw = _mm256_loadu_ps((const float*)(twiddles + i));
        __m256 sum = _mm256_add_ps(a, b), diff = _mm256_sub_ps(a, b);
        __m256 tw_re = _mm256_shuffle_ps(tw, tw, 0xA0), tw_im = _mm256_shuffle_ps(tw, tw, 0xF5);
        __m256 re_sum = _mm256_mul_ps(sum, tw_re), im_sum = _mm256_mul_ps(sum, tw_im);
        __m256 re_diff = _mm256_mul_ps(diff, tw_re), im_diff = _mm256_mul_ps(diff, tw_im);
        __m256 real = _mm256_add_ps(_mm256_shuffle_ps(re_sum, re_sum, 0xF5), _mm256_shuffle_ps(im_diff, im_diff, 0xA0));
        __m256 imag = _mm256_sub_ps(_mm256_shuffle_ps(im_sum, im_sum, 0xF5), _mm256_shuffle_ps(re_diff, re_diff, 0xA0));
        __m256 left_out, right_out;
        if (forward) {
            left_out = _mm256_blend_ps(_mm256_fmadd_ps(half, sum, real), _mm256_fmadd_ps(half, diff, imag), 0xAA);
            right_out = _mm256_blend_ps(_mm256_fmsub_ps(half, sum, real), _mm256_fnmadd_ps(half, diff, imag), 0xAA);
        } else {
            left_out = _mm256_blend_ps(_mm256_sub_ps(sum, real), _mm256_sub_ps(diff, imag), 0xAA);
            right_out = _mm256_blend_ps(_mm256_add_ps(sum, real), _mm256_sub_ps(_mm256_setzero_ps(), _mm256_add_ps(imag, diff)), 0xAA);
        }
        _mm256_storeu_ps((float*)(lm + i), left_out);
        __m128d olo = _mm_castps_pd(_mm256_castps256_ps128(right_out)), ohi = _mm_castps_pd(_mm256_extractf128_ps(right_out, 1));
        _mm_storel_pd((double*)r0, olo);
        _mm_storeh_pd((double*)(r0 - 1), olo);
        _mm_storel_pd((double*)(r0 - 2), ohi);
        _mm_storeh_pd((double*)(r0 - 3), ohi);
    }
    return simd * 4;
}
