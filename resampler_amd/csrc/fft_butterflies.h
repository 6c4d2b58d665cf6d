// fft_butterflies.h -- complex helpers and the radix-2/3/4/5/7/8 butterflies of the FFT path, shared by
// the workgroup-per-transform kernels (fft_kernels.hip) and the wave-per-transform kernel (fft_wave.hip).
// Operation for operation the scalar specs of the reference (src/fft/butterflies/butterflyN/mod.rs).
#pragma once

#include <hip/hip_runtime.h>

namespace rsmp {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// Complex32::mul (fft/mod.rs:52-57)
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// r-point DFT of t[0..R) (t[0] untwiddled), scalar-spec arithmetic of butterflyN/mod.rs.
template <int R> __device__ __forceinline__ void dft(const float2 (&t)[R], float2 (&o)[R]);

template <> __device__ __forceinline__ void dft<2>(const float2 (&t)[2], float2 (&o)[2]) {
    o[0] = cadd(t[0], t[1]);                                   // butterfly2/mod.rs:263-265
    o[1] = csub(t[0], t[1]);
}
template <> __device__ __forceinline__ void dft<3>(const float2 (&t)[3], float2 (&o)[3]) {
    const float SQRT3_2 = 0.8660254f;                          // butterfly3/mod.rs:47
    const float2 sum_t = cadd(t[1], t[2]), diff_t = csub(t[1], t[2]);
    o[0] = cadd(t[0], sum_t);
    const float re_part = t[0].x - 0.5f * sum_t.x, im_part = t[0].y - 0.5f * sum_t.y;
    const float sre = SQRT3_2 * diff_t.y, sim = -SQRT3_2 * diff_t.x;
    o[1] = make_float2(re_part + sre, im_part + sim);
    o[2] = make_float2(re_part - sre, im_part - sim);
}
template <> __device__ __forceinline__ void dft<4>(const float2 (&t)[4], float2 (&o)[4]) {
    const float2 a0 = cadd(t[0], t[2]), a1 = csub(t[0], t[2]), a2 = cadd(t[1], t[3]);   // butterfly4/mod.rs:309-320
    const float a3_re = t[1].y - t[3].y, a3_im = t[3].x - t[1].x;
    o[0] = cadd(a0, a2);
    o[2] = csub(a0, a2);
    o[1] = make_float2(a1.x + a3_re, a1.y + a3_im);
    o[3] = make_float2(a1.x - a3_re, a1.y - a3_im);
}
template <> __device__ __forceinline__ void dft<5>(const float2 (&t)[5], float2 (&o)[5]) {
    const float C1 = 0.309017f, S1 = 0.95105654f, C2 = -0.809017f, S2 = 0.58778524f;   // butterfly5/mod.rs:47-50
    const float2 sum_all = cadd(cadd(cadd(t[1], t[2]), t[3]), t[4]);
    const float2 a1 = cadd(t[1], t[4]), a2 = cadd(t[2], t[3]);
    const float b1_re = t[1].y - t[4].y, b1_im = t[4].x - t[1].x;
    const float b2_re = t[2].y - t[3].y, b2_im = t[3].x - t[2].x;
    const float c1_re = t[0].x + C1 * a1.x + C2 * a2.x, c1_im = t[0].y + C1 * a1.y + C2 * a2.y;
    const float c2_re = t[0].x + C2 * a1.x + C1 * a2.x, c2_im = t[0].y + C2 * a1.y + C1 * a2.y;
    const float d1_re = S1 * b1_re + S2 * b2_re, d1_im = S1 * b1_im + S2 * b2_im;
    const float d2_re = S2 * b1_re - S1 * b2_re, d2_im = S2 * b1_im - S1 * b2_im;
    o[0] = cadd(t[0], sum_all);
    o[1] = make_float2(c1_re + d1_re, c1_im + d1_im);
    o[2] = make_float2(c2_re + d2_re, c2_im + d2_im);
    o[3] = make_float2(c2_re - d2_re, c2_im - d2_im);
    o[4] = make_float2(c1_re - d1_re, c1_im - d1_im);
}
template <> __device__ __forceinline__ void dft<7>(const float2 (&t)[7], float2 (&o)[7]) {
    const float C[3] = {0.6234898f, -0.22252093f, -0.90096885f};   // butterfly7/mod.rs:47-52
    const float S[3] = {0.7818315f, 0.9749279f, 0.43388373f};
    const float2 sum_all = cadd(cadd(cadd(cadd(cadd(t[1], t[2]), t[3]), t[4]), t[5]), t[6]);
    const float2 a1 = cadd(t[1], t[6]), a2 = cadd(t[2], t[5]), a3 = cadd(t[3], t[4]);
    const float b1_re = t[1].y - t[6].y, b1_im = t[6].x - t[1].x;
    const float b2_re = t[2].y - t[5].y, b2_im = t[5].x - t[2].x;
    const float b3_re = t[3].y - t[4].y, b3_im = t[4].x - t[3].x;
    o[0] = cadd(t[0], sum_all);
    // (cos1, sin1, cos2, sin2, cos3, sin3) per output, butterfly7/mod.rs:416-436
#define RSMP_R7(idx, c1, s1, c2, s2, c3, s3)                                                     \
    {                                                                                            \
        const float c_re = t[0].x + (c1) * a1.x + (c2) * a2.x + (c3) * a3.x;                     \
        const float c_im = t[0].y + (c1) * a1.y + (c2) * a2.y + (c3) * a3.y;                     \
        const float d_re = (s1) * b1_re + (s2) * b2_re + (s3) * b3_re;                           \
        const float d_im = (s1) * b1_im + (s2) * b2_im + (s3) * b3_im;                           \
        o[idx] = make_float2(c_re + d_re, c_im + d_im);                                          \
    }
    RSMP_R7(1, C[0], S[0], C[1], S[1], C[2], S[2])
    RSMP_R7(2, C[1], S[1], C[2], -S[2], C[0], -S[0])
    RSMP_R7(3, C[2], S[2], C[0], -S[0], C[1], S[1])
    RSMP_R7(4, C[2], -S[2], C[0], S[0], C[1], -S[1])
    RSMP_R7(5, C[1], -S[1], C[2], S[2], C[0], S[0])
    RSMP_R7(6, C[0], -S[0], C[1], -S[1], C[2], -S[2])
#undef RSMP_R7
}
template <> __device__ __forceinline__ void dft<8>(const float2 (&t)[8], float2 (&o)[8]) {
    const float H = 0.70710678118654752440f;                   // butterfly8/mod.rs:299
    const float2 ea0 = cadd(t[0], t[4]), ea1 = csub(t[0], t[4]), ea2 = cadd(t[2], t[6]);
    const float ea3_re = t[2].y - t[6].y, ea3_im = t[6].x - t[2].x;
    const float2 xe0 = cadd(ea0, ea2), xe2 = csub(ea0, ea2);
    const float2 xe1 = make_float2(ea1.x + ea3_re, ea1.y + ea3_im);
    const float2 xe3 = make_float2(ea1.x - ea3_re, ea1.y - ea3_im);
    const float2 oa0 = cadd(t[1], t[5]), oa1 = csub(t[1], t[5]), oa2 = cadd(t[3], t[7]);
    const float oa3_re = t[3].y - t[7].y, oa3_im = t[7].x - t[3].x;
    const float2 xo0 = cadd(oa0, oa2), xo2 = csub(oa0, oa2);
    const float2 xo1 = make_float2(oa1.x + oa3_re, oa1.y + oa3_im);
    const float2 xo3 = make_float2(oa1.x - oa3_re, oa1.y - oa3_im);
    o[0] = cadd(xe0, xo0);
    o[4] = csub(xe0, xo0);
    const float2 w1 = make_float2(H * (xo1.x + xo1.y), H * (xo1.y - xo1.x));
    o[1] = cadd(xe1, w1);
    o[5] = csub(xe1, w1);
    const float2 w2 = make_float2(xo2.y, -xo2.x);
    o[2] = cadd(xe2, w2);
    o[6] = csub(xe2, w2);
    const float2 w3 = make_float2(H * (xo3.y - xo3.x), -H * (xo3.x + xo3.y));
    o[3] = cadd(xe3, w3);
    o[7] = csub(xe3, w3);
}

}  // namespace rsmp
