// fft_butterflies_pk.h -- the radix-2/3/4/5/7/8 butterflies of fft_butterflies.h on a PACKED complex type.
//
// gfx950 executes v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 on a register PAIR at the rate of a scalar f32
// operation, so a complex value held as (re, im) in an aligned pair costs one instruction per complex add.
// Written on float2 components the compiler's pairing of unrelated scalars (re of one value with im of
// another) needed a v_mov for every other arithmetic instruction; on a two-element vector type the pairs are
// the complex values themselves, and the rotations by -i / conjugations fold into the instructions' op_sel /
// neg modifiers.  Per component these are the operations of fft_butterflies.h in the same order (a - b is
// a + (-b), -(a - b) is b - a: exact identities), so both headers round identically.
#pragma once

#include <hip/hip_runtime.h>

namespace rsmp {

typedef float cf __attribute__((ext_vector_type(2)));   // (re, im)

__device__ __forceinline__ cf cf_make(float re, float im) { cf v; v.x = re; v.y = im; return v; }

// The compiler folds neither a half swap nor a half negation of a packed f32 operand into the instruction
// that consumes it (it materialises them with v_mov / v_xor), so the adds and multiplies whose operand is a
// rotated or conjugated value are spelled with their op_sel / neg modifiers here.  op_sel[i] (op_sel_hi[i])
// picks the half of source i that feeds the low (high) result, neg_lo / neg_hi negate it.
#define RSMP_PK2(name, text)                                                                     \
    __device__ __forceinline__ cf name(cf a, cf b) {                                             \
        cf d;                                                                                    \
        asm(text : "=v"(d) : "v"(a), "v"(b));                                                    \
        return d;                                                                                \
    }
#define RSMP_PK3(name, text)                                                                     \
    __device__ __forceinline__ cf name(cf a, cf b, cf c) {                                       \
        cf d;                                                                                    \
        asm(text : "=v"(d) : "v"(a), "v"(b), "v"(c));                                            \
        return d;                                                                                \
    }
// a + (-i b) = (a.x + b.y, a.y - b.x),  a - (-i b) = (a.x - b.y, a.y + b.x)
RSMP_PK2(cf_add_nrot, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
RSMP_PK2(cf_sub_nrot, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")
// a + conj(b) = (a.x + b.x, a.y - b.y),  a - conj(b) = (a.x - b.x, a.y + b.y)
RSMP_PK2(cf_add_conj, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")
RSMP_PK2(cf_sub_conj, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")
// conj(a) + b = (a.x + b.x, b.y - a.y),  conj(a) + conj(b) = (a.x + b.x, -a.y - b.y),
// conj(a - b) = (a.x - b.x, b.y - a.y)
RSMP_PK2(cf_conj_add, "v_pk_add_f32 %0, %1, %2 neg_hi:[1,0]")
RSMP_PK2(cf_conj_add_conj, "v_pk_add_f32 %0, %1, %2 neg_hi:[1,1]")
RSMP_PK2(cf_conj_sub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[1,0]")
// (a.x b.x, a.x b.y),  (a.y b.x, a.y b.y)
RSMP_PK2(cf_mul_xx, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]")
RSMP_PK2(cf_mul_yy, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]")
// a.y (i b) = (-a.y b.y, a.y b.x),  a.x (-i b) = (a.x b.y, -a.x b.x); the _fma forms add c with one rounding
RSMP_PK2(cf_mul_yy_rot, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]")
RSMP_PK2(cf_mul_xx_nrot, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[0,1]")
RSMP_PK3(cf_fma_yy_rot, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]")
RSMP_PK3(cf_fma_xx_nrot, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]")
#undef RSMP_PK2
#undef RSMP_PK3

__device__ __forceinline__ cf cf_conj(cf a) { return cf_make(a.x, -a.y); }
// Complex32::mul (fft/mod.rs:52-57): (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
__device__ __forceinline__ cf cf_mul(cf a, cf b) {
#ifdef RSMP_FFT_WAVE_EXACT
    return cf_mul_xx(a, b) + cf_mul_yy_rot(a, b);
#else
    // one statement: before an asm statement that reads a register the previous instruction wrote, the
    // compiler (which cannot see what kind of instruction it is) spends an s_nop
    cf d;
#if defined(RSMP_EXP) && (RSMP_EXP & 32)   // (slope experiment: two more packed instructions per complex multiply)
    { cf e; asm volatile("v_pk_mul_f32 %0, %1, %2\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=&v"(e) : "v"(a), "v"(b)); }
#endif
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(d) : "v"(a), "v"(b));
    return d;
#endif
}
// (s.y t.x + d.x t.y, s.y t.y - d.x t.x) for m = (d.x, s.y): the rotation of the real <-> complex passes
// (real_complex/mod.rs:60-63, :104-107)
__device__ __forceinline__ cf cf_rc_rotate(cf m, cf t) {
#ifdef RSMP_FFT_WAVE_EXACT
    return cf_mul_yy(m, t) + cf_mul_xx_nrot(m, t);
#else
    cf d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]"
        : "=&v"(d) : "v"(m), "v"(t));
    return d;
#endif
}

template <int R> __device__ __forceinline__ void pdft(const cf (&t)[R], cf (&o)[R]);

template <> __device__ __forceinline__ void pdft<2>(const cf (&t)[2], cf (&o)[2]) {
    o[0] = t[0] + t[1];                                        // butterfly2/mod.rs:263-265
    o[1] = t[0] - t[1];
}
template <> __device__ __forceinline__ void pdft<3>(const cf (&t)[3], cf (&o)[3]) {
    const float SQRT3_2 = 0.8660254f;                          // butterfly3/mod.rs:47
    const cf sum_t = t[1] + t[2], diff_t = t[1] - t[2];
    o[0] = t[0] + sum_t;
    const cf part = t[0] - 0.5f * sum_t;
    const cf s = SQRT3_2 * diff_t;                             // rotated by -i inside the adds
    o[1] = cf_add_nrot(part, s);
    o[2] = cf_sub_nrot(part, s);
}
template <> __device__ __forceinline__ void pdft<4>(const cf (&t)[4], cf (&o)[4]) {
    const cf a0 = t[0] + t[2], a1 = t[0] - t[2], a2 = t[1] + t[3];   // butterfly4/mod.rs:309-320
    const cf a3 = t[1] - t[3];                                 // rotated by -i inside the adds
    o[0] = a0 + a2;
    o[2] = a0 - a2;
    o[1] = cf_add_nrot(a1, a3);
    o[3] = cf_sub_nrot(a1, a3);
}
template <> __device__ __forceinline__ void pdft<5>(const cf (&t)[5], cf (&o)[5]) {
    const float C1 = 0.309017f, S1 = 0.95105654f, C2 = -0.809017f, S2 = 0.58778524f;   // butterfly5/mod.rs:47-50
    const cf sum_all = ((t[1] + t[2]) + t[3]) + t[4];
    const cf a1 = t[1] + t[4], a2 = t[2] + t[3];
    const cf b1 = t[1] - t[4], b2 = t[2] - t[3];               // (their -i rotation happens in the last adds)
    const cf c1 = (t[0] + C1 * a1) + C2 * a2;
    const cf c2 = (t[0] + C2 * a1) + C1 * a2;
    const cf d1 = S1 * b1 + S2 * b2;
    const cf d2 = S2 * b1 - S1 * b2;
    o[0] = t[0] + sum_all;
    o[1] = cf_add_nrot(c1, d1);
    o[2] = cf_add_nrot(c2, d2);
    o[3] = cf_sub_nrot(c2, d2);
    o[4] = cf_sub_nrot(c1, d1);
}
template <> __device__ __forceinline__ void pdft<7>(const cf (&t)[7], cf (&o)[7]) {
    const float C[3] = {0.6234898f, -0.22252093f, -0.90096885f};   // butterfly7/mod.rs:47-52
    const float S[3] = {0.7818315f, 0.9749279f, 0.43388373f};
    const cf sum_all = ((((t[1] + t[2]) + t[3]) + t[4]) + t[5]) + t[6];
    const cf a1 = t[1] + t[6], a2 = t[2] + t[5], a3 = t[3] + t[4];
    const cf b1 = t[1] - t[6], b2 = t[2] - t[5], b3 = t[3] - t[4];   // (rotated by -i in the last add)
    o[0] = t[0] + sum_all;
    // (cos1, sin1, cos2, sin2, cos3, sin3) per output, butterfly7/mod.rs:416-436.  Outputs idx and 7 - idx
    // have the same cosines and the opposite sines: the second is c - d where the first is c + d, bit for bit
    // what evaluating its own row gives ((-s) b = -(s b)).
#define RSMP_R7(idx, c1, s1, c2, s2, c3, s3)                                \
    {                                                                       \
        const cf c = ((t[0] + (c1) * a1) + (c2) * a2) + (c3) * a3;          \
        const cf d = ((s1) * b1 + (s2) * b2) + (s3) * b3;                   \
        o[idx] = cf_add_nrot(c, d);                                         \
        o[7 - idx] = cf_sub_nrot(c, d);                                     \
    }
    RSMP_R7(1, C[0], S[0], C[1], S[1], C[2], S[2])
    RSMP_R7(2, C[1], S[1], C[2], -S[2], C[0], -S[0])
    RSMP_R7(3, C[2], S[2], C[0], -S[0], C[1], S[1])
#undef RSMP_R7
}
template <> __device__ __forceinline__ void pdft<8>(const cf (&t)[8], cf (&o)[8]) {
    const float H = 0.70710678118654752440f;                   // butterfly8/mod.rs:299
    const cf ea0 = t[0] + t[4], ea1 = t[0] - t[4], ea2 = t[2] + t[6], ea3 = t[2] - t[6];
    const cf xe0 = ea0 + ea2, xe2 = ea0 - ea2, xe1 = cf_add_nrot(ea1, ea3), xe3 = cf_sub_nrot(ea1, ea3);
    const cf oa0 = t[1] + t[5], oa1 = t[1] - t[5], oa2 = t[3] + t[7], oa3 = t[3] - t[7];
    const cf xo0 = oa0 + oa2, xo2 = oa0 - oa2, xo1 = cf_add_nrot(oa1, oa3), xo3 = cf_sub_nrot(oa1, oa3);
    o[0] = xe0 + xo0;
    o[4] = xe0 - xo0;
    const cf w1 = H * cf_add_nrot(xo1, xo1);                   // H (x + y, y - x)
    o[1] = xe1 + w1;
    o[5] = xe1 - w1;
    o[2] = cf_add_nrot(xe2, xo2);                              // w2 = (y, -x)
    o[6] = cf_sub_nrot(xe2, xo2);
    const cf w3 = H * cf_sub_nrot(xo3, xo3);                   // -w3: H (x - y, x + y)
    o[3] = xe3 - w3;
    o[7] = xe3 + w3;
}

// pdft<R> of inputs whose last NZ are zero (they are not read): what pdft computes with them, less the
// operations whose operand is a zero (x + 0, x - 0, c * 0: equal in value; the sign of a zero result may
// differ, which no later operation turns into a different value).
template <int R, int NZ> __device__ __forceinline__ void pdft_tail(cf (&t)[R], cf (&o)[R]) {
    if constexpr (NZ == 0) {
        pdft<R>(t, o);
    } else if constexpr (R == 3 && NZ == 1) {
        const float SQRT3_2 = 0.8660254f;
        o[0] = t[0] + t[1];
        const cf part = t[0] - 0.5f * t[1];
        const cf s = SQRT3_2 * t[1];
        o[1] = cf_add_nrot(part, s);
        o[2] = cf_sub_nrot(part, s);
    } else if constexpr (R == 4 && NZ == 2) {
        o[0] = t[0] + t[1];
        o[2] = t[0] - t[1];
        o[1] = cf_add_nrot(t[0], t[1]);
        o[3] = cf_sub_nrot(t[0], t[1]);
    } else if constexpr (NZ == R - 1) {
#pragma unroll
        for (int q = 0; q < R; ++q) o[q] = t[0];
    } else {
#pragma unroll
        for (int q = R - NZ; q < R; ++q) t[q] = cf_make(0.f, 0.f);
        pdft<R>(t, o);
    }
}

}  // namespace rsmp
