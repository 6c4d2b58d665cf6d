// fft_wave.hip -- ResamplerFft block pipeline, ONE WAVE per transform (gfx950).
//
// Replaces the same reference code as fft_kernels.hip (FftResampler::resample, src/resampler_fft.rs:385-424;
// RadixFFT forward / inverse, src/fft/radix_fft.rs:476-670; the Stockham stages and butterflies; the
// real<->complex passes, src/fft/real_complex/mod.rs:37-114) for the plans it is instantiated for.
//
// Why another mapping: with a workgroup per transform every one of the ~14 phases of a block ends in a
// workgroup barrier and costs about a microsecond however little it computes (43 % of the wave time at
// barriers).  Here a wave owns a (stream, channel) and walks a run of its blocks alone:
//   * the 1176 / 1280 complex points live in ONE private LDS buffer (10 KB per wave); a pass reads all of the
//     wave's inputs into registers, then writes the results back in place -- the LDS executes a wave's
//     operations in order, so no barrier or fence exists anywhere in the block loop;
//   * the first two stages of each transform are one register pass (radix 3x7 / 4x5: wave_fused_first); the
//     forward one takes its inputs straight from HBM and skips the zero padding, the last inverse stage leaves
//     its outputs in registers, where the overlap carry of the (stream, channel) also lives for the whole run:
//     conjugation, overlap-add and the interleaved store happen there;
//   * real-FFT post-process, and filter multiply + truncate / zero-extend + inverse pre-process + the input
//     conjugation of the inverse transform, are two in-place passes over bin pairs;
//   * complex values are a packed two-float vector type (fft_butterflies_pk.h): complex adds are single
//     v_pk_*_f32 instructions, rotations / conjugations / complex multiplies carry op_sel / neg modifiers;
//   * LDS rows are padded where a pass's lane stride would meet on banks, reads are ds_read_b64 only, all of a
//     pass's reads are issued before its first butterfly, radix-7/8 twiddle rows fetch w, w^2, w^4 only.
// Arithmetic: the reference's scalar specs, with a*b + c fused and some twiddles multiplied out (1.45e-7 RMS
// from the CPU path); -DRSMP_FFT_WAVE_EXACT (libresampler_amd_fftexact.so) is operation for operation the
// reference's and bit-identical to it (tests/test_fft_gpu.py).
// The two (or C) waves of a block's channels sit in one workgroup, so their half-line stores of the
// interleaved output meet in the same L2.
#include <cmath>
#include <cstdlib>
#include <type_traits>

// a*b + c may fuse in this file: one rounding fewer per fused pair.  The results then differ from the
// reference's scalar arithmetic (which never fuses) in the last bits -- far inside the 1e-6 RMS gate (measured
// against the oracle: tests/test_fft_gpu.py) -- and the kernel needs 10 % fewer vector instructions.
// -DRSMP_FFT_WAVE_EXACT keeps the reference's operation-for-operation arithmetic (bit-identical to it).
#ifndef RSMP_FFT_WAVE_EXACT
#pragma clang fp contract(fast)
#endif

#include "fft_butterflies_pk.h"
#include "fft_kernels.h"

namespace rsmp {

namespace {


// Lanes of a wave exchange data through the wave's LDS buffer without any barrier: the hardware executes a
// wave's LDS operations in issue order.  The COMPILER, however, reasons per thread and may move a thread's
// store above its own loads of provably different addresses -- which are other lanes' data here (it did,
// in the radix-4 stage).  This pins the program order of memory operations; it emits no instruction.
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// Stream pointers come out of a descriptor in memory, so the compiler knows no address space for them and
// emits FLAT loads and stores -- which count on the LDS counter too, and so tie every wait for an LDS read to
// the block's output stores.  Naming the global address space gives global_load / global_store.
typedef __attribute__((address_space(1))) float GFloat;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f4 GFloat4;
__device__ __forceinline__ const GFloat* as_global(const float* p) { return (const GFloat*)p; }
__device__ __forceinline__ GFloat* as_global(float* p) { return (GFloat*)p; }

// One LDS value by ds_read_b64, which the LDS serves at 256 B/clk.  Left to itself the compiler pairs
// neighbouring loads into ds_read2_b64 / ds_read2st64_b64, which run at HALF that rate (8 LDS cycles for the
// 16 bytes per lane against 2 + 2, MI355X_MICROARCH.md LDS table) in a kernel whose bound is the LDS; a
// volatile access is never merged (and must name the LDS address space: address-space inference skips
// volatile accesses, which would otherwise become flat loads).
__device__ __forceinline__ cf lds_ld(const cf* p) {
    typedef const volatile __attribute__((address_space(3))) cf* LdsPtr;
    return *(LdsPtr)(p);
}

template <int N_, int... Rs> struct WavePlan;
template <int N_, int R0, int R1, int R2, int R3>
struct WavePlan<N_, R0, R1, R2, R3> {
    static constexpr int N = N_;
    static constexpr int kR[4] = {R0, R1, R2, R3};
    static_assert(R0 * R1 * R2 * R3 == N_, "radices");
    // stage twiddles, unique per column: stage s (s >= 1) holds stride_s rows of R_s - 1.  In LDS the rows
    // of stages 2 and 3 are (R - 1) | 1 values apart: lane k reads row k, and an even row length puts lanes
    // 16 apart (radix 7: six values = 12 dwords) on the same banks.
    static constexpr int row(int r) { return (r - 1) | 1; }
    static constexpr int kT1 = 0, kT2 = kT1 + R0 * (R1 - 1), kT3 = kT2 + R0 * R1 * row(R2);
    static constexpr int kTw = kT3 + R0 * R1 * R2 * row(R3);
    static constexpr int kSrc2 = R0 * (R1 - 1), kSrc3 = kSrc2 + R0 * R1 * (R2 - 1);   // offsets in the plan's array
    static constexpr int kRc = N_ / 2 - 1;   // real <-> complex twiddles
    static bool matches(uint32_t n, uint32_t n_stages, const uint32_t* radix) {
        return n == static_cast<uint32_t>(N_) && n_stages == 4 && radix[0] == R0 && radix[1] == R1 &&
               radix[2] == R2 && radix[3] == R3;
    }
};

// One Stockham stage in place in the wave's LDS buffer: butterfly i reads buf[i + q*M], twiddles inputs
// 1..R-1 with w[(i mod STRIDE)*(R-1) + q-1] and writes buf[R*i - (R-1)*k + q*STRIDE]
// (butterfly4/mod.rs:316-320 etc.).  Every read of the stage is issued before its first write.
// The R - 1 twiddles of a butterfly are the powers w, w^2 .. w^(R-1) of one value.  The kernel is bound by
// LDS traffic, of which the twiddle rows were a quarter: radix 7 and 8 fetch w, w^2 and w^4 and multiply
// the others out (one or two roundings more on those twiddles; -DRSMP_FFT_WAVE_EXACT fetches all of them).
// A row is FETCHED (twiddle_fetch: kFetch<R> LDS reads, issued with the stage's data reads) and EXPANDED
// when its butterfly runs.
#if !defined(RSMP_FFT_WAVE_EXACT) && !defined(RSMP_FFT_WAVE_ALL_TWIDDLES)
template <int R> constexpr int kFetch = (R == 7 || R == 8) ? 3 : R - 1;
#else
template <int R> constexpr int kFetch = R - 1;
#endif
template <int R>
__device__ __forceinline__ void twiddle_fetch(const cf* __restrict__ w, cf (&raw)[kFetch<R>]) {
    if constexpr (kFetch<R> != R - 1) {
        raw[0] = lds_ld(w);
        raw[1] = lds_ld(w + 1);
        raw[2] = lds_ld(w + 3);
    } else {
#pragma unroll
        for (int q = 0; q < R - 1; ++q) raw[q] = lds_ld(w + q);
    }
}
template <int R>
__device__ __forceinline__ void twiddle_expand(const cf (&raw)[kFetch<R>], cf (&tw)[R]) {
    if constexpr (kFetch<R> != R - 1) {
        tw[1] = raw[0];
        tw[2] = raw[1];
        tw[4] = raw[2];
        tw[3] = cf_mul(tw[1], tw[2]);
        tw[5] = cf_mul(tw[1], tw[4]);
        tw[6] = cf_mul(tw[2], tw[4]);
        if constexpr (R == 8) tw[7] = cf_mul(tw[3], tw[4]);
    } else {
#pragma unroll
        for (int q = 1; q < R; ++q) tw[q] = raw[q - 1];
    }
}

// QS: distance of a butterfly's inputs in the buffer (N / R, or more when the producer padded its rows).
// OPAD: values of padding after every R * STRIDE outputs (one block of the next stage's columns).  With 21
// columns a half wave of 32 lanes spans two blocks, and 147 values = 294 dwords put the second block's first
// columns on the first block's last banks; two values more (298 = 42 mod 64) and every half wave of the
// stage stores conflict-free.  The next stage then reads its inputs N / R' + OPAD apart (stage_out_pad).
constexpr int stage_out_pad(int r, int stride) { return r == 7 && stride == 21 ? 2 : 0; }
template <int N, int R, int STRIDE, int QS = N / R, int OPAD = stage_out_pad(R, STRIDE)>
__device__ __forceinline__ void wave_stage(cf* buf, const cf* __restrict__ tw, int lane) {
    constexpr int M = N / R;
    constexpr int ITER = (M + 63) / 64;
    constexpr int ROW = (R - 1) | 1;
    // Every LDS read of the stage -- data and twiddle rows, in the order of their use -- is issued before the
    // first butterfly: the wave then waits for a read once per stage, not once per butterfly (LDS operations
    // of a wave complete in order, so butterfly 0 runs while the later reads are still in flight).
    cf t[ITER][R], raw[ITER][kFetch<R>];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = lane + 64 * it;
        if ((it + 1) * 64 <= M || i < M) {
#pragma unroll
            for (int q = 0; q < R; ++q) t[it][q] = lds_ld(buf + i + q * QS);
            twiddle_fetch<R>(tw + (i % STRIDE) * ROW, raw[it]);
        }
    }
    lds_order();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = lane + 64 * it;
        if ((it + 1) * 64 <= M || i < M) {
            const int k = i % STRIDE;
            cf twr[R];
            twiddle_expand<R>(raw[it], twr);
#pragma unroll
            for (int q = 1; q < R; ++q) t[it][q] = cf_mul(twr[q], t[it][q]);
            cf o[R];
            pdft<R>(t[it], o);
            cf* d = buf + R * i - (R - 1) * k + (OPAD ? OPAD * (i / STRIDE) : 0);
#pragma unroll
            for (int q = 0; q < R; ++q) d[q * STRIDE] = o[q];
        }
    }
    lds_order();
}

// Stages 0 (radix RA, stride 1, no twiddles) and 1 (radix RB, stride RA, twiddles W_(RA*RB)^(k q')) of a
// transform in ONE register pass.  The three (RA) stage-1 butterflies 3j, 3j+1, 3j+2 consume exactly the
// outputs of the seven (RB) stage-0 butterflies j + M2*q': unit j therefore takes the RA*RB points
// j + M2*m (m = q' + RB*q), runs RB radix-RA butterflies, the twiddles and RA radix-RB butterflies in
// registers, and writes the contiguous outputs RA*RB*j .. RA*RB*j + RA*RB - 1 -- what the two stages
// would have left in LDS, with one LDS round trip and the stage-1 index arithmetic gone.  Same operations
// on the same values as the separate stages (the unit twiddles of column k = 0 are skipped).
// `load(index)` yields point `index` of the stage-0 input (LDS, or samples straight from HBM).
// A unit's outputs are RA*RB values apart from the next lane's; when that is even (20 values = 40 dwords)
// the 64 lanes of a store meet on 8 bank pairs, so one value of padding follows every fused_pad<>() values
// (160: lanes 8 apart move on by a bank pair) and the next stage reads its inputs fused_qs<>() apart.
template <int RA, int RB> constexpr int fused_pad() { return (RA * RB) % 2 == 0 ? 8 * RA * RB : 0; }
template <int N, int R, int RA, int RB> constexpr int fused_qs() {
    // inputs i + q * (N / R) of the next stage: the padding adds (i + q N/R) / pad, exact when N / R == pad
    static_assert(fused_pad<RA, RB>() == 0 || N / R == fused_pad<RA, RB>(), "padding period = input distance");
    return fused_pad<RA, RB>() ? N / R + 1 : N / R;
}
template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}
// NVALID: points at index >= NVALID of the stage-0 input are zero and are neither fetched nor computed with
// (the zero padding of the forward transform: resampler_fft.rs:387-388).  Butterfly q' takes the points
// j + M2 (q' + RB q): the last NZ of its RA inputs are padding for every j (pdft_tail).
template <int N, int RA, int RB, int NVALID = N, class Load>
__device__ __forceinline__ void wave_fused_first(cf* dst, const cf* __restrict__ tw1, int lane, Load load) {
    constexpr int M2 = N / (RA * RB);
    constexpr int PADJ = fused_pad<RA, RB>() / (RA * RB);   // units per padding value
    constexpr int ITER = (M2 + 63) / 64;
    cf s[ITER][RB][RA];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int j = lane + 64 * it;
        if ((it + 1) * 64 <= M2 || j < M2) {
            static_for<0, RB>([&](auto qp_c) {
                static_for<0, RA>([&](auto q_c) {
                    constexpr int m = decltype(qp_c)::value + RB * decltype(q_c)::value;
                    if constexpr (M2 * m < NVALID) s[it][decltype(qp_c)::value][decltype(q_c)::value] = load(j + M2 * m);
                });
            });
        }
    }
    cf w1[RA][RB];   // (the same for every lane: broadcast reads, fetched with the data)
#pragma unroll
    for (int k = 1; k < RA; ++k)
#pragma unroll
        for (int qp = 1; qp < RB; ++qp) w1[k][qp] = lds_ld(tw1 + k * (RB - 1) + qp - 1);
    lds_order();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int j = lane + 64 * it;
        if ((it + 1) * 64 <= M2 || j < M2) {
            static_for<0, RB>([&](auto qp_c) {
                constexpr int qp = decltype(qp_c)::value;
                // inputs q with M2 (qp + RB q) >= NVALID are zero: count them from the end
                constexpr int first_zero = M2 * qp >= NVALID ? 0 : (NVALID - M2 * qp + M2 * RB - 1) / (M2 * RB);
                constexpr int NZ = first_zero >= RA ? 0 : RA - first_zero;
                cf o[RA];
                pdft_tail<RA, NZ>(s[it][qp], o);
#pragma unroll
                for (int k = 0; k < RA; ++k) s[it][qp][k] = o[k];
            });
#pragma unroll
            for (int k = 0; k < RA; ++k) {
                cf u[RB], o[RB];
                u[0] = s[it][0][k];
#pragma unroll
                for (int qp = 1; qp < RB; ++qp)
                    u[qp] = k == 0 ? s[it][qp][k] : cf_mul(w1[k][qp], s[it][qp][k]);
                pdft<RB>(u, o);
#pragma unroll
                for (int qq = 0; qq < RB; ++qq) dst[RA * RB * j + (PADJ ? j / PADJ : 0) + k + RA * qq] = o[qq];
            }
        }
    }
    lds_order();
}

// postprocess_fft (radix_fft.rs:500-537 + real_complex/mod.rs:37-74), in place on x[0 .. N2].
template <int N2>
__device__ __forceinline__ void wave_postprocess(cf* x, const cf* __restrict__ rc, int lane) {
    constexpr int ITERS = (N2 + 1) / 2 - 1;
    constexpr int TRIPS = (ITERS + 63) / 64;
    if (lane == 0) {
        const cf z0 = x[0];
        x[0] = cf_make(z0.x + z0.y, 0.0f);
        x[N2] = cf_make(z0.x - z0.y, 0.0f);
    }
    // A pair (l, N2 - l) is read and written by one lane, so nothing orders the trips; GROUP trips' reads are
    // issued together and the wave waits once per group.
    constexpr int GROUP = 5;
#pragma unroll
    for (int g = 0; g < TRIPS; g += GROUP) {
        cf o[GROUP], orv[GROUP], tw[GROUP];
#pragma unroll
        for (int u = 0; u < GROUP; ++u) {
            const int i = lane + 64 * (g + u);
            if (g + u < TRIPS && i < ITERS) {
                o[u] = lds_ld(x + 1 + i);
                orv[u] = lds_ld(x + N2 - 1 - i);
                tw[u] = lds_ld(rc + i);
            }
        }
#pragma unroll
        for (int u = 0; u < GROUP; ++u) {
            const int i = lane + 64 * (g + u);
            if (g + u < TRIPS && i < ITERS) {
                // o + conj(orv) = (sum.x, diff.y) and o - conj(orv) = (diff.x, sum.y) of real_complex/mod.rs:52-58
                const cf half = 0.5f * cf_add_conj(o[u], orv[u]);           // (half_sum_real, half_diff_imag)
                const cf ri = cf_rc_rotate(cf_sub_conj(o[u], orv[u]), tw[u]);  // (real, imag)
                x[1 + i] = half + ri;
                x[N2 - 1 - i] = cf_conj_sub(half, ri);                      // (half_sum_real - real, imag - half_diff_imag)
            }
        }
        lds_order();
    }
    if (((N2 + 1) & 1) && lane == 32) x[(N2 + 1) / 2].y = -x[(N2 + 1) / 2].y;
    lds_order();
}

// resampler_fft.rs:401-408 (multiply new_length bins by the filter spectrum, zero the rest up to FO),
// preprocess_ifft (radix_fft.rs:592-624 + real_complex/mod.rs:84-114) and the input conjugation of
// process_inverse_complex (:634-637), fused over the bin pairs (l, FO - l), in place.
// NL = the plan's new_length (a constant of the two sizes: fft_in + 1 or fft_out, resampler_fft.rs:396-399),
// FMAX = the last index of the filter table.
template <int FO, int NL, int FMAX>
__device__ __forceinline__ void wave_filter_preprocess(cf* y, const cf* __restrict__ filter,
                                                       const cf* __restrict__ rc, int lane) {
    constexpr int ITERS = (FO + 1) / 2 - 1;
    constexpr int TRIPS = (ITERS + 63) / 64;
    static_assert(ITERS + 1 <= NL, "the low bin of every pair is multiplied by the filter");
    auto bin = [&](int k) -> cf {
        return k < NL ? cf_mul(lds_ld(y + k), lds_ld(filter + k)) : cf_make(0.f, 0.f);
    };
    cf first = cf_make(0.f, 0.f), mid = cf_make(0.f, 0.f);
    if (lane == 0) {
        const cf a = bin(0), b = bin(FO);
        const cf first_sum = a + b, first_diff = a - b;
        first = cf_make(first_sum.x - first_sum.y, first_diff.x - first_diff.y);
    }
    if (((FO + 1) & 1) && lane == 32) {
        const cf c = bin((FO + 1) / 2);
        const cf dbl = c + c;
        mid = cf_make(dbl.x, -dbl.y);
    }
    // A pair (l, FO - l) is read and written by one lane: GROUP trips' reads are issued together.  Whether the
    // high bins FO - 1 - i of a trip lie below NL is known per trip: all of them (no test), none (zeros, no
    // reads) or some (the reads stay inside the tables, the product is dropped).
    constexpr int GROUP = 3;
#pragma unroll
    for (int g = 0; g < TRIPS; g += GROUP) {
        cf ya[GROUP], fa[GROUP], yb[GROUP], fb[GROUP], tw[GROUP];
#pragma unroll
        for (int u = 0; u < GROUP; ++u) {
            const int trip = g + u, i = lane + 64 * trip;
            const bool none = FO - 1 - (64 * trip + 63) >= NL;
            if (trip < TRIPS && ((trip + 1) * 64 <= ITERS || i < ITERS)) {
                const int l = 1 + i, rr = FO - 1 - i;
                ya[u] = lds_ld(y + l);
                fa[u] = lds_ld(filter + l);
                if (!none) {
                    yb[u] = lds_ld(y + rr);
                    fb[u] = lds_ld(filter + (rr < FMAX ? rr : FMAX));
                }
                tw[u] = lds_ld(rc + i);
            }
        }
#pragma unroll
        for (int u = 0; u < GROUP; ++u) {
            const int trip = g + u, i = lane + 64 * trip;
            const bool none = FO - 1 - (64 * trip + 63) >= NL, all = FO - 1 - 64 * trip < NL;
            if (trip < TRIPS && ((trip + 1) * 64 <= ITERS || i < ITERS)) {
                const int l = 1 + i, rr = FO - 1 - i;
                const cf a = cf_mul(ya[u], fa[u]);
                cf b = cf_make(0.f, 0.f);
                if (!none) {
                    b = cf_mul(yb[u], fb[u]);
                    if (!all) b = rr < NL ? b : cf_make(0.f, 0.f);
                }
                const cf sd = cf_add_conj(a, b);                      // (sum.x, diff.y)
                const cf ri = cf_rc_rotate(cf_sub_conj(a, b), tw[u]); // (real, imag)
                y[l] = cf_conj_sub(sd, ri);                           // (sum.x - real, -(diff.y - imag))
#ifdef RSMP_FFT_WAVE_EXACT
                y[rr] = cf_conj(cf_conj_add_conj(sd, ri));            // (sum.x + real, -(-imag - diff.y)), zero signs included
#else
                y[rr] = sd + ri;
#endif
            }
        }
        lds_order();
    }
    if (lane == 0) y[0] = cf_make(first.x, -first.y);
    if (((FO + 1) & 1) && lane == 32) y[(FO + 1) / 2] = cf_make(mid.x, -mid.y);
    lds_order();
}

// OCC waves per SIMD: 2 = two workgroups of 4 waves per CU (80 KB of LDS each: the tables + 4 buffers),
// 3 = one workgroup of 12 waves per CU (one copy of the tables + 12 buffers = 158 KB; <= 168 registers).
template <class FWD, class INV, bool C2, int OCC>
__global__ __launch_bounds__((OCC == 3 ? 12 : 4) * 64, OCC) void fft_ola_wave_kernel(FftPlanDev plan,
                                                                              const FftStreamDesc* __restrict__ descs,
                                                                              uint32_t run, uint32_t runs_per_stream,
                                                                              uint32_t total_waves) {
    extern __shared__ __attribute__((aligned(16))) cf lds2[];
    constexpr int kWavesPerGroup = OCC == 3 ? 12 : 4;
    constexpr int FI = FWD::N, FO = INV::N;
    constexpr int LDSC = (FI > FO ? FI : FO) + 2 + 8;   // + the padding of a fused first pass (1280 + 8) or of a stage (1176 + 16)
    static_assert(1176 + 16 <= LDSC, "stage padding");
    constexpr int R1 = FWD::kR[0];
    constexpr int RL = INV::kR[3], ML = FO / RL, ITERL = (ML + 63) / 64, HL = RL / 2;
    static_assert(RL % 2 == 0, "the last inverse stage splits its outputs into output half and carry half");
    static_assert(FI % 2 == 0 && FO % 2 == 0, "frame pairs");
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * kWavesPerGroup + wave;
    // Every table of the plan (stage twiddles, real <-> complex twiddles, filter spectrum: 39 KB) is copied
    // to LDS once per workgroup: the per-butterfly twiddle fetches were the kernel's main wait (48 % of the
    // wave time at s_waitcnt, vector-memory instructions in flight 4x the LDS ones).  The only barrier of
    // the kernel follows; after it the waves never meet again.
    constexpr int kTabF = 0, kTabI = kTabF + FWD::kTw, kTabRcF = kTabI + INV::kTw, kTabRcI = kTabRcF + FWD::kRc,
                  kTabFilter = kTabRcI + INV::kRc, kTabEnd = kTabFilter + FI + 1;
    cf* tab = lds2;
    {
        auto copy = [&](cf* dst, const cf* __restrict__ src, int n) {
            for (int i = threadIdx.x; i < n; i += kWavesPerGroup * 64) dst[i] = src[i];
        };
        // stage twiddles: rows of stages 2 and 3 re-spaced (WavePlan::row)
        auto rows = [&](cf* dst, const cf* __restrict__ src, int n_rows, int r) {
            const int len = r - 1, pitch = (r - 1) | 1;
            for (int i = threadIdx.x; i < n_rows * len; i += kWavesPerGroup * 64) dst[(i / len) * pitch + i % len] = src[i];
        };
        auto stage_tables = [&](cf* dst, const cf* __restrict__ src, auto P) {
            typedef decltype(P) PL;
            copy(dst + PL::kT1, src, PL::kSrc2);
            rows(dst + PL::kT2, src + PL::kSrc2, PL::kR[0] * PL::kR[1], PL::kR[2]);
            rows(dst + PL::kT3, src + PL::kSrc3, PL::kR[0] * PL::kR[1] * PL::kR[2], PL::kR[3]);
        };
        stage_tables(tab + kTabF, reinterpret_cast<const cf*>(plan.tw_f), FWD{});
        stage_tables(tab + kTabI, reinterpret_cast<const cf*>(plan.tw_i), INV{});
        copy(tab + kTabRcF, reinterpret_cast<const cf*>(plan.rc_f), FWD::kRc);
        copy(tab + kTabRcI, reinterpret_cast<const cf*>(plan.rc_i), INV::kRc);
        copy(tab + kTabFilter, reinterpret_cast<const cf*>(plan.filter), FI + 1);
    }
    __syncthreads();
    if (gw >= total_waves) return;
    cf* buf = lds2 + kTabEnd + wave * LDSC;
    const cf* tw_f = tab + kTabF;
    const cf* tw_i = tab + kTabI;
    const cf* rc_f = tab + kTabRcF;
    const cf* rc_i = tab + kTabRcI;
    const cf* filter = tab + kTabFilter;

    // wave -> (stream, run of blocks, channel); the channels of a run are neighbouring waves
    const uint32_t stream_idx = gw / (runs_per_stream * (C2 ? 2u : descs[0].channels));
    const FftStreamDesc d = descs[stream_idx];
    const uint32_t C = C2 ? 2u : d.channels;
    const uint32_t in_stream = gw - stream_idx * runs_per_stream * C;
    const uint32_t run_idx = in_stream / C;
    const uint32_t ch = in_stream - run_idx * C;
    const uint32_t first = run_idx * run;
    if (first >= d.n_blocks) return;
    const uint32_t last = first + run < d.n_blocks ? first + run : d.n_blocks;  // exclusive

    // overlap carried into the run: the stream state, or the predecessor block recomputed (not emitted).
    // Held as the unconjugated transform outputs (the conjugation of radix_fft.rs:656-669 is a modifier of
    // the overlap-add below).
    cf carry[ITERL][HL];
#pragma unroll
    for (int it = 0; it < ITERL; ++it) {
        const int i = lane + 64 * it;
#pragma unroll
        for (int q = 0; q < HL; ++q) {
            carry[it][q] = cf_make(0.f, 0.f);
            if (first == 0 && i < ML) {
                const int c = i + q * ML;   // complex index = reals 2c, 2c + 1 of the channel's overlap row
                const GFloat* ov = as_global(d.overlap);
                carry[it][q] = cf_make(ov[ch * FO + 2 * c], -ov[ch * FO + 2 * c + 1]);
            }
        }
    }
    const int64_t b_begin = first == 0 ? 0 : static_cast<int64_t>(first) - 1;

    for (int64_t b = b_begin; b < static_cast<int64_t>(last); ++b) {
        const bool emit = b >= static_cast<int64_t>(first);
        constexpr int S1 = R1, S2 = S1 * FWD::kR[1], S3 = S2 * FWD::kR[2];
        constexpr int T1 = FWD::kT1, T2 = FWD::kT2, T3 = FWD::kT3;
        // ---- forward stages 1 + 2 in one register pass, inputs straight from HBM: complex j of the block's
        // channel = frames 2j, 2j + 1 (j < FI / 2), zero beyond (resampler_fft.rs:387-388)
        {
            const GFloat* xin = as_global(d.in) + static_cast<size_t>(b) * FI * C;
            auto sample = [&](int j) -> cf {
                cf v = cf_make(0.f, 0.f);
                if (j < FI / 2) {
                    if constexpr (C2) {
                        const f4 f = ((const GFloat4*)xin)[j];
                        v = ch == 0 ? cf_make(f.x, f.z) : cf_make(f.y, f.w);
                    } else {
                        v = cf_make(xin[static_cast<size_t>(2 * j) * C + ch], xin[static_cast<size_t>(2 * j + 1) * C + ch]);
                    }
                }
                return v;
            };
            wave_fused_first<FI, FWD::kR[0], FWD::kR[1], FI / 2>(buf, tw_f + T1, lane, sample);
        }
        wave_stage<FI, FWD::kR[2], S2, fused_qs<FI, FWD::kR[2], FWD::kR[0], FWD::kR[1]>()>(buf, tw_f + T2, lane);
        static_assert(stage_out_pad(FWD::kR[2], S2) == 0 || FI / FWD::kR[3] == S3, "padding period = input distance");
        wave_stage<FI, FWD::kR[3], S3, FI / FWD::kR[3] + stage_out_pad(FWD::kR[2], S2)>(buf, tw_f + T3, lane);
        wave_postprocess<FI>(buf, rc_f, lane);
        wave_filter_preprocess<FO, (FI < FO ? FI + 1 : FO), FI>(buf, filter, rc_i, lane);

        constexpr int IS1 = INV::kR[0], IS2 = IS1 * INV::kR[1];
        constexpr int IT1 = INV::kT1, IT2 = INV::kT2, IT3 = INV::kT3;
        // inverse stages 1 + 2 in one register pass, in place
        wave_fused_first<FO, INV::kR[0], INV::kR[1]>(buf, tw_i + IT1, lane, [&](int j) -> cf { return lds_ld(buf + j); });
        wave_stage<FO, INV::kR[2], IS2, fused_qs<FO, INV::kR[2], INV::kR[0], INV::kR[1]>()>(buf, tw_i + IT2, lane);
        // ---- last inverse stage: outputs stay in registers.  Butterfly i (k = i) yields Z[i + q*ML]; the
        // output conjugation (radix_fft.rs:656-669) makes reals 2c, 2c + 1 of the channel out of Z[c]; the
        // first FO reals are overlap-added and stored, the second FO become the next overlap (:416-423).
        GFloat* xout = as_global(d.out) + static_cast<size_t>(b) * FO * C;
        // reads of butterfly it + 1 are issued before butterfly it runs (all of them at once do not fit the
        // 168 registers of three waves per SIMD next to the carry)
        cf tl[2][RL], rawl[2][kFetch<RL>];
        auto fetch = [&](int it) {
            const int i = lane + 64 * it;
            if ((it + 1) * 64 <= ML || i < ML) {
#pragma unroll
                for (int q = 0; q < RL; ++q) tl[it & 1][q] = lds_ld(buf + i + q * (ML + stage_out_pad(INV::kR[2], IS2)));
                twiddle_fetch<RL>(tw_i + IT3 + i * INV::row(RL), rawl[it & 1]);
            }
        };
        fetch(0);
#pragma unroll
        for (int it = 0; it < ITERL; ++it) {
            const int i = lane + 64 * it;
            if (it + 1 < ITERL) fetch(it + 1);
            if ((it + 1) * 64 <= ML || i < ML) {
                cf o[RL], twr[RL];
                twiddle_expand<RL>(rawl[it & 1], twr);
#pragma unroll
                for (int q = 1; q < RL; ++q) tl[it & 1][q] = cf_mul(twr[q], tl[it & 1][q]);
                pdft<RL>(tl[it & 1], o);
#pragma unroll
                for (int q = 0; q < HL; ++q) {
                    const int c = i + q * ML;
                    if (emit) {
                        const cf v = cf_conj_add_conj(o[q], carry[it][q]);
                        xout[static_cast<size_t>(2 * c) * C + ch] = v.x;
                        xout[static_cast<size_t>(2 * c + 1) * C + ch] = v.y;
                    }
                    carry[it][q] = o[q + HL];
                }
            }
        }
        lds_order();
    }
    if (last == d.n_blocks) {
#pragma unroll
        for (int it = 0; it < ITERL; ++it) {
            const int i = lane + 64 * it;
            if (i < ML) {
#pragma unroll
                for (int q = 0; q < HL; ++q) {
                    const int c = i + q * ML;
                    GFloat* ov = as_global(d.overlap_next);
                    ov[ch * FO + 2 * c] = carry[it][q].x;
                    ov[ch * FO + 2 * c + 1] = -carry[it][q].y;
                }
            }
        }
    }
}

typedef WavePlan<1176, 3, 7, 7, 8> W1176;
typedef WavePlan<1280, 4, 5, 8, 8> W1280;

}  // namespace

// Wave-per-transform kernels exist for the 44.1 <-> 48 kHz family (both directions).  Returns
// hipErrorNotSupported when the plan is another one (the caller then uses the workgroup kernels).
hipError_t launch_fft_ola_wave(const FftPlanDev& plan, const FftStreamDesc* d_descs, uint32_t n_streams,
                               uint32_t max_blocks, uint32_t max_channels, uint32_t min_channels,
                               hipStream_t stream) {
    if (max_channels != min_channels) return hipErrorNotSupported;   // one wave layout per launch
    if (plan.n_rc_f != plan.fft_in / 2 - 1 || plan.n_rc_i != plan.fft_out / 2 - 1) return hipErrorNotSupported;
    if (plan.new_length != (plan.fft_in < plan.fft_out ? plan.fft_in + 1 : plan.fft_out)) return hipErrorNotSupported;
    const bool up = W1176::matches(plan.fft_in, plan.n_stages_f, plan.radix_f) &&
                    W1280::matches(plan.fft_out, plan.n_stages_i, plan.radix_i);
    const bool down = W1280::matches(plan.fft_in, plan.n_stages_f, plan.radix_f) &&
                      W1176::matches(plan.fft_out, plan.n_stages_i, plan.radix_i);
    if (!up && !down) return hipErrorNotSupported;
    const uint32_t C = max_channels;
    typedef void (*Kernel)(FftPlanDev, const FftStreamDesc*, uint32_t, uint32_t, uint32_t);
    Kernel fn;
    // Three waves per SIMD (<= 168 registers) where the kernel fits them: the two-channel instantiation.  The
    // any-channel-count one needs ~250 registers (strided sample addressing) and spilled 290 bytes per lane
    // under the cap: two waves per SIMD run it 25-30 % faster (tools/fft_channels_bench.py).
    static const int occ_env = [] { const char* e = getenv("RSMP_FFT_WAVE_OCC"); return e ? atoi(e) : 0; }();
    const int occ = occ_env == 2 || occ_env == 3 ? occ_env : (C == 2 ? 3 : 2);
    if (occ == 3) {
        if (up) fn = C == 2 ? fft_ola_wave_kernel<W1176, W1280, true, 3> : fft_ola_wave_kernel<W1176, W1280, false, 3>;
        else fn = C == 2 ? fft_ola_wave_kernel<W1280, W1176, true, 3> : fft_ola_wave_kernel<W1280, W1176, false, 3>;
    } else {
        if (up) fn = C == 2 ? fft_ola_wave_kernel<W1176, W1280, true, 2> : fft_ola_wave_kernel<W1176, W1280, false, 2>;
        else fn = C == 2 ? fft_ola_wave_kernel<W1280, W1176, true, 2> : fft_ola_wave_kernel<W1280, W1176, false, 2>;
    }
    // tables (stage twiddles, real <-> complex twiddles, filter spectrum fft_in + 1) + one buffer per wave
    const uint32_t kWavesPerGroup = occ == 3 ? 12u : 4u;
    const size_t lds = (static_cast<size_t>(W1176::kTw + W1280::kTw + W1176::kRc + W1280::kRc) + plan.fft_in + 1 +
                        static_cast<size_t>(kWavesPerGroup) * (1280 + 2 + 8)) * sizeof(cf);
    // Blocks per wave: every run after a stream's first recomputes its predecessor block (1 / run extra
    // work), and the launch ends with a partly filled round unless the number of waves is close to a
    // multiple of what the chip holds at once (3 workgroups of 4 waves per CU).
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const double slots = static_cast<double>(cus) * occ * 4;
    uint32_t run = 16;
    double best = -1.0;
    for (uint32_t cand = 6; cand <= 64; ++cand) {
        const double runs = static_cast<double>((max_blocks + cand - 1) / cand);
        const double waves = runs * n_streams * C;
        const double rounds = std::ceil(waves / slots);
        const double useful = static_cast<double>(max_blocks) / (max_blocks + runs - 1.0);   // halo blocks
        const double score = waves / (rounds * slots) * useful;
        if (score > best + 1e-9) { best = score; run = cand; }
    }
    static const char* knob = getenv("RSMP_FFT_RUN");
    if (knob && atoi(knob) > 0) run = static_cast<uint32_t>(atoi(knob));
    const uint32_t runs_per_stream = (max_blocks + run - 1) / run;
    const uint32_t total_waves = runs_per_stream * n_streams * C;
    const dim3 grid((total_waves + kWavesPerGroup - 1) / kWavesPerGroup);
    if (lds > 64 * 1024) {   // dynamic LDS above 64 KiB must be opted into
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, grid, dim3(kWavesPerGroup * 64), lds, stream, plan, d_descs, run, runs_per_stream,
                       total_waves);
    return hipGetLastError();
}

}  // namespace rsmp
