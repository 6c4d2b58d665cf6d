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
#include "common.h"

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
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) f2 GFloat2;
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
// Likewise one ds_write_b64 per value: the ds_write2_b64 the compiler forms of two costs 13 LDS cycles against 6 + 6.
__device__ __forceinline__ void lds_st(cf* p, cf v) {
#ifdef RSMP_FFT_WAVE_MERGED_STORES
    *p = v;
#else
    typedef volatile __attribute__((address_space(3))) cf* LdsPtr;
    *(LdsPtr)(p) = v;
#endif
}

// A transform of N complex points in `Rs...` Stockham stages (2 .. 4 of them), as the reference's planner orders
// them (src/fft/optimizer.rs).  Where the first two radices multiply to at most 21 values per unit (and a third
// stage exists) they run as one register pass (wave_fused_first); every later stage but the inverse's last is a
// wave_stage; the twiddle tables of all stages sit in LDS.
// LDS stores go 16 lanes at a time over 32 banks (MI355X_MICROARCH.md, LDS table; tools/fft_bank_model.py counts the
// array cycles of every pass of a plan pair).  A stage's lane i stores its value q at R (i - k) + k + q stride
// (k = i mod stride): lanes 16 apart in i are in different blocks of `stride` columns unless stride >= 16, and a
// block is (R - 1) stride values further than the lane index says -- two values per 16 lanes of shift keep the
// 16 lanes of a store on distinct banks iff (R - 1) stride + pad is a multiple of 16 values.  (Radix 7, stride 21:
// 147-value blocks, 2 values of padding; radix 8, stride 20: 4.)
constexpr int stage_out_pad(int r, int stride) { return (16 - ((r - 1) * stride) % 16) % 16; }
constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }
// Twiddles a stage keeps per column in LDS: all R - 1 of the row, or -- radix 7 and 8 -- only w, w^2 and w^4 (the
// stage multiplies the others out, see twiddle_expand; the tables of the 1176 <-> 1280 pair shrink from 39 to 29 KB).
#if !defined(RSMP_FFT_WAVE_EXACT) && !defined(RSMP_FFT_WAVE_ALL_TWIDDLES)
constexpr int fetch_count(int r) { return (r == 7 || r == 8) ? 3 : r - 1; }
#else
constexpr int fetch_count(int r) { return r - 1; }
#endif
template <int N_, int... Rs>
struct WavePlan {
    static constexpr int N = N_;
    static constexpr int kStages = sizeof...(Rs);
    static constexpr int kR[sizeof...(Rs)] = {Rs...};
    static_assert(kStages >= 2 && kStages <= 5, "stages");
    static constexpr int stride(int s) { int v = 1; for (int i = 0; i < s; ++i) v *= kR[i]; return v; }
    static_assert(stride(kStages) == N_, "radices");
    static constexpr bool kFused = kStages >= 3 && kR[0] * kR[1] <= 21;
    // Stage twiddles, unique per column: stage s (s >= 1) holds stride(s) rows of R_s - 1.  In LDS the rows of a
    // wave_stage are (R - 1) | 1 values apart: lane k reads row k, and an even row length puts lanes 16 apart
    // (radix 7: six values = 12 dwords) on the same banks.  (The fused pass reads its rows by constant index.)
    static constexpr int row(int r) { return fetch_count(r) | 1; }
    static constexpr int pitch(int s) { return kFused && s == 1 ? kR[1] - 1 : row(kR[s]); }
    static constexpr int tab(int s) { int off = 0; for (int i = 1; i < s; ++i) off += stride(i) * pitch(i); return off; }   // LDS offset of stage s
    static constexpr int src(int s) { int off = 0; for (int i = 1; i < s; ++i) off += stride(i) * (kR[i] - 1); return off; }   // offset in the plan's array
    static constexpr int kTw = tab(kStages);
    static constexpr int kRc = N_ / 2 - 1;   // real <-> complex twiddles
    // Padding between passes (LDS banks).  The first pass (fused or not) writes kUnit values per lane side by side:
    // an even kUnit puts lanes 32 / gcd(2 kUnit, 32) apart on the same banks, so one value of padding follows every
    // kPadJ units (20 values per unit: every 4) where the next stage's input distance is a multiple of that period.
    // After the blocks of a later stage: stage_out_pad, where the stage that follows reads block by block.
    // in_pad(s): what stage s's input distance N / R_s grows by; in_period(s): elements between two padding values
    // inside that distance (0 = none).
    static constexpr int kUnit = kFused ? kR[0] * kR[1] : kR[0];
    static constexpr int kNext = kFused ? 2 : 1;   // the stage that reads the first pass's output
    // (Plans above 2048 points run at the 256-register cap of their wide workgroups: the padded addressing spilled
    // there -- 2352 -> 2560 points 0.80 -> 1.00 ms -- so they keep the plain layout, but for the radix-7 blocks.)
    static constexpr bool kPadded = N_ <= 2048;
    static constexpr int first_padj() {
        if (!kPadded || kUnit % 2 != 0 || kNext >= kStages) return 0;
        const int p = 32 / gcd_c(2 * kUnit, 32);
        return (N_ / kR[kNext < kStages ? kNext : 0]) % (p * kUnit) == 0 ? p : 0;
    }
    static constexpr int kPadJ = first_padj();
    static constexpr int out_pad(int s) {
        if (s < 1 || s + 1 >= kStages || (kFused && s == 1)) return 0;
        if (stride(s) >= N_ / kR[s]) return 0;   // one block
        const int p = kPadded || (kR[s] == 7 && stride(s) == 21) ? stage_out_pad(kR[s], stride(s)) : 0;
        return p != 0 && N_ / kR[s + 1] == stride(s + 1) ? p : 0;
    }
    static constexpr int in_pad(int s) {
        if (s == kNext) return kPadJ ? (N_ / kR[s]) / (kPadJ * kUnit) : 0;
        return s >= 2 ? out_pad(s - 1) : 0;
    }
    static constexpr int in_period(int s) { return s == kNext && kPadJ && N_ / kR[s] > kPadJ * kUnit ? kPadJ * kUnit : 0; }
    static constexpr int buf_values() {   // what the wave's buffer needs: the points + bin N and its neighbour (real <-> complex passes), or the widest padded layout
        int pad = kPadJ ? N_ / (kPadJ * kUnit) : 0;
        for (int s = 1; s + 1 < kStages; ++s) {
            const int p = out_pad(s) * (N_ / stride(s + 1));
            if (p > pad) pad = p;
        }
        return N_ + (pad > 2 ? pad : 2);
    }
    static constexpr int kBuf = buf_values();
    static bool matches(uint32_t n, uint32_t n_stages, const uint32_t* radix) {
        if (n != static_cast<uint32_t>(N_) || n_stages != static_cast<uint32_t>(kStages)) return false;
        for (int s = 0; s < kStages; ++s)
            if (radix[s] != static_cast<uint32_t>(kR[s])) return false;
        return true;
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
template <int R> constexpr int kFetch = fetch_count(R);
template <int R>
__device__ __forceinline__ void twiddle_fetch(const cf* __restrict__ w, cf (&raw)[kFetch<R>]) {
    if constexpr (kFetch<R> != R - 1) {
        raw[0] = lds_ld(w);       // (the LDS row holds w, w^2, w^4)
        raw[1] = lds_ld(w + 1);
        raw[2] = lds_ld(w + 2);
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
// IPP: the producer (the first pass) left one value of padding after every IPP of the stage's inputs (0 = none).
template <int N, int R, int STRIDE, int QS = N / R, int OPAD = 0, int IPP = 0>
__device__ __forceinline__ void wave_stage(cf* buf, const cf* __restrict__ tw, int lane) {
    constexpr int M = N / R;
    constexpr int ITER = (M + 63) / 64;
    constexpr int ROW = fetch_count(R) | 1;
    // Every LDS read of the stage -- data and twiddle rows, in the order of their use -- is issued before the
    // first butterfly: the wave then waits for a read once per stage, not once per butterfly (LDS operations
    // of a wave complete in order, so butterfly 0 runs while the later reads are still in flight).
    if constexpr (STRIDE == M && QS == M && OPAD == 0 && IPP == 0 && ITER >= 4) {
        // A plan's last stage writes every value where it read it (stride = M: the butterfly's own points), so
        // butterflies need not wait for each other's reads: of a long stage (4 or 5 trips: 64-80 values and their
        // twiddles in registers at once, which the plans of 2048 points and more paid with spills) only the next
        // trip's reads are in flight while one runs.
        cf t2[2][R], raw2[2][kFetch<R>];
        auto fetch = [&](int it) {
            const int i = lane + 64 * it;
            if ((it + 1) * 64 <= M || i < M) {
#pragma unroll
                for (int q = 0; q < R; ++q) t2[it & 1][q] = lds_ld(buf + i + q * M);
                twiddle_fetch<R>(tw + i * ROW, raw2[it & 1]);
            }
        };
        fetch(0);
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int i = lane + 64 * it;
            if (it + 1 < ITER) fetch(it + 1);
            if ((it + 1) * 64 <= M || i < M) {
                cf twr[R], o[R];
                twiddle_expand<R>(raw2[it & 1], twr);
#pragma unroll
                for (int q = 1; q < R; ++q) t2[it & 1][q] = cf_mul(twr[q], t2[it & 1][q]);
                pdft<R>(t2[it & 1], o);
#pragma unroll
                for (int q = 0; q < R; ++q) lds_st(buf + i + q * M, o[q]);
            }
        }
        lds_order();
        return;
    }
    cf t[ITER][R], raw[ITER][kFetch<R>];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = lane + 64 * it;
        if ((it + 1) * 64 <= M || i < M) {
#pragma unroll
            for (int q = 0; q < R; ++q) t[it][q] = lds_ld(buf + i + (IPP ? i / (IPP ? IPP : 1) : 0) + q * QS);
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
            for (int q = 0; q < R; ++q) lds_st(d + q * STRIDE, o[q]);
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
// PADJ: one value of padding after every PADJ units (WavePlan::kPadJ; 0 = none).
template <int N, int RA, int RB, int PADJ, int NVALID = N, class Load>
__device__ __forceinline__ void wave_fused_first(cf* dst, const cf* __restrict__ tw1, int lane, Load load) {
    constexpr int M2 = N / (RA * RB);
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
                for (int qq = 0; qq < RB; ++qq) lds_st(dst + RA * RB * j + (PADJ ? j / (PADJ ? PADJ : 1) : 0) + k + RA * qq, o[qq]);
            }
        }
    }
    lds_order();
}

// Stage 0 alone (stride 1, no twiddles) for the plans that do not fuse it with stage 1: butterfly i takes the
// points i + q N / R through `load` (LDS, or samples straight from HBM; points at index >= NVALID are zero) and
// writes R i + q.
template <int N, int R, int PADJ, int NVALID = N, class Load>
__device__ __forceinline__ void wave_first(cf* dst, int lane, Load load) {
    constexpr int M = N / R;
    constexpr int ITER = (M + 63) / 64;
    constexpr int first_zero = (NVALID + M - 1) / M;            // inputs q >= first_zero are zero for every i
    constexpr int NZ = first_zero >= R ? 0 : R - first_zero;
    cf t[ITER][R];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = lane + 64 * it;
        if ((it + 1) * 64 <= M || i < M) {
#pragma unroll
            for (int q = 0; q < R - NZ; ++q) t[it][q] = load(i + q * M);
        }
    }
    lds_order();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = lane + 64 * it;
        if ((it + 1) * 64 <= M || i < M) {
            cf o[R];
            pdft_tail<R, NZ>(t[it], o);
#pragma unroll
            for (int q = 0; q < R; ++q) lds_st(dst + R * i + (PADJ ? i / (PADJ ? PADJ : 1) : 0) + q, o[q]);
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
            const bool none = FO - 1 - (64 * trip + 63) >= NL;   // high bins FO - 1 - i: none / all / some below NL
            const bool lo_none = 1 + 64 * trip >= NL;             // low bins 1 + i likewise (a long up-sampling block)
            if (trip < TRIPS && ((trip + 1) * 64 <= ITERS || i < ITERS)) {
                const int l = 1 + i, rr = FO - 1 - i;
                if (!lo_none) {
                    ya[u] = lds_ld(y + l);
                    fa[u] = lds_ld(filter + (l < FMAX ? l : FMAX));
                }
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
            const bool lo_none = 1 + 64 * trip >= NL, lo_all = 64 + 64 * trip < NL;
            if (trip < TRIPS && ((trip + 1) * 64 <= ITERS || i < ITERS)) {
                const int l = 1 + i, rr = FO - 1 - i;
                cf a = cf_make(0.f, 0.f);
                if (!lo_none) {
                    a = cf_mul(ya[u], fa[u]);
                    if (!lo_all) a = l < NL ? a : cf_make(0.f, 0.f);
                }
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
// 3 = one workgroup of 12 waves per CU (one copy of the tables + 12 buffers = 158 KB; <= 168 registers),
// 1 = one workgroup of as many waves (<= 8) as the CU's LDS holds buffers for (the long plans; the launch
// decides).  (16 waves per CU for the short plans measured within 2 % of 12: the LDS is the bound, not latency.)
constexpr int wave_group_threads(int occ) { return occ == 3 ? 768 : occ == 2 ? 256 : 512; }
// Both transforms above 2048 points (88.2 <-> 96 kHz): four or five trips per stage in registers next to the
// carry do not fit 256 registers -- these pairs run one wave per SIMD with the full register file instead of two
// that spill (88.2 -> 96 kHz: 0.94 -> 0.76 ms).
template <class FWD, class INV> constexpr bool kOneWavePerSimd = (FWD::N > 2048 && INV::N > 2048) || FWD::N > 2560 || INV::N > 2560;
// CHM: 0 = any number of channels (a wave per channel, 4-byte accesses at the frame's stride), 1 = two channels (16-byte
// accesses), 2 = an even number of channels taken as channel pairs (8-byte accesses at the frame's stride)
template <class FWD, class INV, int CHM, int OCC>
__global__ __launch_bounds__((OCC == 1 && kOneWavePerSimd<FWD, INV> ? 256 : wave_group_threads(OCC)),
                             (OCC == 1 ? (kOneWavePerSimd<FWD, INV> ? 1 : 2) : OCC)) void fft_ola_wave_kernel(FftPlanDev plan,
                                                                              const FftStreamDesc* __restrict__ descs,
                                                                              uint32_t run, uint32_t runs_per_stream,
                                                                              uint32_t total_waves, uint32_t pairs) {
    extern __shared__ __attribute__((aligned(16))) cf lds2[];
    constexpr bool C2 = CHM != 0;
    const int kWavesPerGroup = OCC == 1 ? static_cast<int>(blockDim.x >> 6) : wave_group_threads(OCC) / 64;
    constexpr int FI = FWD::N, FO = INV::N;
    constexpr int kFilterLen = FI < FO ? FI + 1 : FO;   // bins the filter multiplies (new_length): the rest of its spectrum stays in HBM
    constexpr int LDSC = FWD::kBuf > INV::kBuf ? FWD::kBuf : INV::kBuf;   // (+ the padding of a fused first pass or of a stage)
    constexpr int SF = FWD::kStages, SI = INV::kStages;
    // The last inverse stage keeps its outputs in registers where its radix is even: outputs q < RL / 2 of a butterfly
    // are the block's frames, the others the overlap carried to the next block.  An odd last radix (the 44.1 kHz
    // family as the output side) runs the stage through LDS like the others and splits in a pass of its own.
    constexpr int RL = INV::kR[SI - 1], ML = FO / RL, ITERL = (ML + 63) / 64;
    constexpr bool kOddLast = RL % 2 != 0;
    constexpr int HL = kOddLast ? 1 : RL / 2;               // carried values per lane and trip
    constexpr int CM = kOddLast ? FO / 2 : ML;              // carry (it, q) <-> complex index lane + 64 it + q CM
    constexpr int CIT = (CM + 63) / 64;
    constexpr int kLastIpp = INV::in_period(SI - 1);   // (a two- or three-stage inverse: its last stage reads the first pass's output)
    static_assert(FI % 2 == 0 && FO % 2 == 0, "frame pairs");
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * kWavesPerGroup + wave;
    // Every table of the plan (stage twiddles, real <-> complex twiddles, the filter bins in use: 29 KB) is copied
    // to LDS once per workgroup: the per-butterfly twiddle fetches were the kernel's main wait (48 % of the
    // wave time at s_waitcnt, vector-memory instructions in flight 4x the LDS ones).  The only barrier of
    // the kernel follows; after it the waves never meet again.
    constexpr int kTabF = 0, kTabI = kTabF + FWD::kTw, kTabRcF = kTabI + INV::kTw, kTabRcI = kTabRcF + FWD::kRc,
                  kTabFilter = kTabRcI + INV::kRc, kTabEnd = kTabFilter + kFilterLen;
    cf* tab = lds2;
    {
        auto copy = [&](cf* dst, const cf* __restrict__ src, int n) {
            for (int i = threadIdx.x; i < n; i += kWavesPerGroup * 64) dst[i] = src[i];
        };
        // stage twiddles: rows re-spaced to the pitch the stage reads them with (WavePlan::pitch); of a row of
        // `len` values the first `keep` are taken, or (keep == 3 < len) w, w^2 and w^4
        auto rows = [&](cf* dst, const cf* __restrict__ src, int n_rows, int len, int keep, int pitch) {
            for (int i = threadIdx.x; i < n_rows * keep; i += kWavesPerGroup * 64) {
                const int r = i / keep, j = i - r * keep;
                dst[r * pitch + j] = src[r * len + (keep < len && j == 2 ? 3 : j)];
            }
        };
        auto stage_tables = [&](cf* dst, const cf* __restrict__ src, auto P) {
            typedef decltype(P) PL;
            static_for<1, PL::kStages>([&](auto s_c) {
                constexpr int s = decltype(s_c)::value;
                constexpr int len = PL::kR[s] - 1;
                rows(dst + PL::tab(s), src + PL::src(s), PL::stride(s), len, PL::kFused && s == 1 ? len : fetch_count(PL::kR[s]), PL::pitch(s));
            });
        };
        stage_tables(tab + kTabF, reinterpret_cast<const cf*>(plan.tw_f), FWD{});
        stage_tables(tab + kTabI, reinterpret_cast<const cf*>(plan.tw_i), INV{});
        copy(tab + kTabRcF, reinterpret_cast<const cf*>(plan.rc_f), FWD::kRc);
        copy(tab + kTabRcI, reinterpret_cast<const cf*>(plan.rc_i), INV::kRc);
        copy(tab + kTabFilter, reinterpret_cast<const cf*>(plan.filter), kFilterLen);
    }
    // Two-channel streams: the two waves of a stream's channels (neighbours, wave ^ 1) exchange half of their final
    // values through LDS so that each stores whole 16-byte pieces of the interleaved output -- frames 2c, 2c + 1 of BOTH
    // channels -- instead of two 4-byte stores per value at a stride of 16 bytes that only sometimes merged with the
    // other wave's in L2 (HBM writes 1.2x the output).  Per wave two words: blocks whose outgoing values are in its
    // buffer (`ready`), blocks whose outgoing values the neighbour has taken (`taken`).  (The long plans have no
    // registers to spare for the values in flight and keep the 4-byte stores.)
    constexpr bool kXch = C2 && !kOddLast && HL % 2 == 0 && FI <= 2048 && FO <= 2048;
    uint32_t* xflags = reinterpret_cast<uint32_t*>(lds2 + kTabEnd + kWavesPerGroup * LDSC);
    if (kXch && threadIdx.x < 2u * kWavesPerGroup) xflags[threadIdx.x] = 0;
    __syncthreads();
    if (gw >= total_waves) return;
    const bool xch = kXch && kWavesPerGroup % 2 == 0;   // (an odd number of waves per workgroup would part a pair)
    uint32_t xseq = 0;                                   // blocks exchanged so far
    cf* buf = lds2 + kTabEnd + wave * LDSC;
    cf* pbuf = lds2 + kTabEnd + (wave ^ 1u) * LDSC;      // the neighbour's buffer
    const cf* tw_f = tab + kTabF;
    const cf* tw_i = tab + kTabI;
    const cf* rc_f = tab + kTabRcF;
    const cf* rc_i = tab + kTabRcI;
    const cf* filter = tab + kTabFilter;

    // wave -> (stream, run of blocks, channel); the channels of a run are neighbouring waves.
    // C2: a stream of `pairs` channel PAIRS is taken as that many two-channel streams whose frames lie 2 * pairs values
    // apart (pairs = 1: stereo) -- `ch` is the wave's channel inside its pair, `chan` in the frame.
    const uint32_t vstream = C2 ? gw / (runs_per_stream * 2u) : gw / (runs_per_stream * descs[0].channels);
    const uint32_t stream_idx = CHM == 2 ? vstream / pairs : vstream;
    const uint32_t pair = CHM == 2 ? vstream - stream_idx * pairs : 0u;
    const FftStreamDesc d = descs[stream_idx];
    const uint32_t C = CHM == 1 ? 2u : CHM == 2 ? 2u * pairs : d.channels;
    const uint32_t CW = C2 ? 2u : C;   // channels that share a run's neighbouring waves
    const uint32_t in_stream = gw - vstream * runs_per_stream * CW;
    const uint32_t run_idx = in_stream / CW;
    const uint32_t ch = in_stream - run_idx * CW;
    const uint32_t chan = C2 ? 2u * pair + ch : ch;
    constexpr bool stereo = CHM == 1;   // (frames of exactly the two channels: 16-byte loads and stores)
    const uint32_t pair2 = 2u * pair;
    const uint32_t first = run_idx * run;
    if (first >= d.n_blocks) return;
    const uint32_t last = first + run < d.n_blocks ? first + run : d.n_blocks;  // exclusive

    // overlap carried into the run: the stream state, or the predecessor block recomputed (not emitted).
    // Held as the unconjugated transform outputs (the conjugation of radix_fft.rs:656-669 is a modifier of
    // the overlap-add below).
    cf carry[CIT][HL];
#pragma unroll
    for (int it = 0; it < CIT; ++it) {
        const int i = lane + 64 * it;
#pragma unroll
        for (int q = 0; q < HL; ++q) {
            carry[it][q] = cf_make(0.f, 0.f);
            if (first == 0 && i < CM) {
                const int c = i + q * CM;   // complex index = reals 2c, 2c + 1 of the channel's overlap row
                const GFloat* ov = as_global(d.overlap);
                carry[it][q] = cf_make(ov[chan * FO + 2 * c], -ov[chan * FO + 2 * c + 1]);
            }
        }
    }
    const int64_t b_begin = first == 0 ? 0 : static_cast<int64_t>(first) - 1;

    for (int64_t b = b_begin; b < static_cast<int64_t>(last); ++b) {
        const bool emit = b >= static_cast<int64_t>(first);
        if constexpr (kXch) {   // the neighbour must have taken the previous block's outgoing values out of this buffer
            if (xch && xseq > 0)
                while (__hip_atomic_load(xflags + 2 * wave + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < xseq) __builtin_amdgcn_s_sleep(1);
        }
        // ---- forward transform: the first pass takes its inputs straight from HBM: complex j of the block's
        // channel = frames 2j, 2j + 1 (j < FI / 2), zero beyond (resampler_fft.rs:387-388)
        {
            const GFloat* xin = as_global(d.in) + static_cast<size_t>(b) * FI * C;
            auto sample = [&](int j) -> cf {
                cf v = cf_make(0.f, 0.f);
                if (j < FI / 2) {
                    if constexpr (C2) {
                        if constexpr (stereo) {
                            const f4 f = ((const GFloat4*)xin)[j];
                            v = ch == 0 ? cf_make(f.x, f.z) : cf_make(f.y, f.w);
                        } else {   // the pair's eight bytes in each of the two frames (32-bit offsets from a uniform base)
                            const uint32_t fo = 2u * static_cast<uint32_t>(j) * C + pair2;
                            const f2 a = *(const GFloat2*)(xin + fo);
                            const f2 bb = *(const GFloat2*)(xin + fo + C);
                            v = ch == 0 ? cf_make(a.x, bb.x) : cf_make(a.y, bb.y);
                        }
                    } else {
                        v = cf_make(xin[static_cast<size_t>(2 * j) * C + ch], xin[static_cast<size_t>(2 * j + 1) * C + ch]);
                    }
                }
                return v;
            };
            auto first_pass = [&](auto&& smp) {
                if constexpr (FWD::kFused) wave_fused_first<FI, FWD::kR[0], FWD::kR[1], FWD::kPadJ, FI / 2>(buf, tw_f + FWD::tab(1), lane, smp);
                else wave_first<FI, FWD::kR[0], FWD::kPadJ, FI / 2>(buf, lane, smp);
            };
            if constexpr (stereo) {
                // WAV samples straight from their PCM bytes (resample/src/main.rs:128-137: `sample as f32 / (1 << (bits - 1)) as
                // f32`; SURVEY 8 f1: "int16/24 -> f32 conversion could be fused into the load kernel"): the conversion pass --
                // PCM read, f32 written, f32 read again -- is gone, the input side of the launch is 2 (16-bit) or 3 bytes a
                // sample instead of 4 + 2 + 4.  Little-endian, two channels a frame; complex j = frames 2j, 2j + 1.
                const uint32_t in_bits = __builtin_amdgcn_readfirstlane(d.in_bits);
                if (in_bits == 0) {
                    first_pass(sample);
                } else {
                    typedef __attribute__((address_space(1))) uint32_t GU32;
                    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
                    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(1))) u2v GU2;
                    typedef __attribute__((address_space(1))) u4v GU4;
                    const GU32* pin = (const GU32*)d.in + (static_cast<size_t>(b) * FI * 2 * (in_bits >> 3)) / 4;   // (FI even: a whole number of words)
                    auto sext = [](uint32_t v, int bits) -> float { return static_cast<float>(static_cast<int32_t>(v << (32 - bits)) >> (32 - bits)); };
                    if (in_bits == 16) {
                        first_pass([&](int j) -> cf {
                            cf v = cf_make(0.f, 0.f);
                            if (j < FI / 2) {
                                const u2v w = ((const GU2*)pin)[j];   // frames 2j, 2j + 1: (ch0 | ch1 << 16) each
                                const uint32_t sh = ch * 16u;
                                v = cf_make(sext(w.x >> sh, 16), sext(w.y >> sh, 16)) * (1.0f / 32768.0f);
                            }
                            return v;
                        });
                    } else if (in_bits == 24) {
                        first_pass([&](int j) -> cf {
                            cf v = cf_make(0.f, 0.f);
                            if (j < FI / 2) {
                                const GU32* p = pin + 3 * j;   // twelve bytes: frame 2j (ch0, ch1), frame 2j + 1 (ch0, ch1), three bytes each
                                const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
                                const uint32_t a = ch == 0 ? w0 : (w0 >> 24) | (w1 << 8);
                                const uint32_t c = ch == 0 ? (w1 >> 16) | (w2 << 16) : w2 >> 8;
                                v = cf_make(sext(a, 24), sext(c, 24)) * (1.0f / 8388608.0f);
                            }
                            return v;
                        });
                    } else {
                        first_pass([&](int j) -> cf {
                            cf v = cf_make(0.f, 0.f);
                            if (j < FI / 2) {
                                const u4v w = ((const GU4*)pin)[j];
                                // (main.rs:131: the divisor `(1 << 31) as f32` is an i32 literal = -2^31: 32-bit files come out with
                                // inverted polarity in the reference, and so here)
                                v = cf_make(static_cast<float>(static_cast<int32_t>(ch == 0 ? w.x : w.y)), static_cast<float>(static_cast<int32_t>(ch == 0 ? w.z : w.w))) * (-1.0f / 2147483648.0f);
                            }
                            return v;
                        });
                    }
                }
            } else {
                first_pass(sample);
            }
        }
        static_for<(FWD::kFused ? 2 : 1), SF>([&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            wave_stage<FI, FWD::kR[s], FWD::stride(s), FI / FWD::kR[s] + FWD::in_pad(s), FWD::out_pad(s), FWD::in_period(s)>(buf, tw_f + FWD::tab(s), lane);
        });
        wave_postprocess<FI>(buf, rc_f, lane);
        wave_filter_preprocess<FO, kFilterLen, kFilterLen - 1>(buf, filter, rc_i, lane);

        // ---- inverse transform, in place; its last stage below
        {
            auto from_lds = [&](int j) -> cf { return lds_ld(buf + j); };
            if constexpr (INV::kFused) wave_fused_first<FO, INV::kR[0], INV::kR[1], INV::kPadJ>(buf, tw_i + INV::tab(1), lane, from_lds);
            else wave_first<FO, INV::kR[0], INV::kPadJ>(buf, lane, from_lds);
        }
        static_for<(INV::kFused ? 2 : 1), (kOddLast ? SI : SI - 1)>([&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            wave_stage<FO, INV::kR[s], INV::stride(s), FO / INV::kR[s] + INV::in_pad(s), INV::out_pad(s), INV::in_period(s)>(buf, tw_i + INV::tab(s), lane);
        });
        // ---- last inverse stage: outputs stay in registers.  Butterfly i (k = i) yields Z[i + q*ML]; the
        // output conjugation (radix_fft.rs:656-669) makes reals 2c, 2c + 1 of the channel out of Z[c]; the
        // first FO reals are overlap-added and stored, the second FO become the next overlap (:416-423).
        GFloat* xout = as_global(d.out) + static_cast<size_t>(b) * FO * C;
        if constexpr (kOddLast) {
#pragma unroll
            for (int it = 0; it < CIT; ++it) {
                const int c = lane + 64 * it;
                if ((it + 1) * 64 <= CM || c < CM) {
                    const cf z = lds_ld(buf + c), z2 = lds_ld(buf + c + CM);
                    if (emit) {
                        const cf v = cf_conj_add_conj(z, carry[it][0]);
                        xout[static_cast<size_t>(2 * c) * C + chan] = v.x;
                        xout[static_cast<size_t>(2 * c + 1) * C + chan] = v.y;
                    }
                    carry[it][0] = z2;
                }
            }
        } else {
        // reads of butterfly it + 1 are issued before butterfly it runs (all of them at once do not fit the
        // 168 registers of three waves per SIMD next to the carry)
        cf tl[2][RL], rawl[2][kFetch<RL>];
        cf vkeep[ITERL][HL / 2 > 0 ? HL / 2 : 1];   // kXch: the values this wave stores itself (q of its channel's parity)
        // where a lane's outgoing value j of trip it goes: the slots its own inputs q = j of that trip came from
        auto xaddr = [&](int i, int j) -> int { return i + (kLastIpp ? i / (kLastIpp ? kLastIpp : 1) : 0) + j * (ML + INV::in_pad(SI - 1)); };
        auto fetch = [&](int it) {
            const int i = lane + 64 * it;
            if ((it + 1) * 64 <= ML || i < ML) {
#pragma unroll
                for (int q = 0; q < RL; ++q) tl[it & 1][q] = lds_ld(buf + i + (kLastIpp ? i / (kLastIpp ? kLastIpp : 1) : 0) + q * (ML + INV::in_pad(SI - 1)));
                twiddle_fetch<RL>(tw_i + INV::tab(SI - 1) + i * INV::row(RL), rawl[it & 1]);
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
                        if (kXch && xch) {
                            if (static_cast<uint32_t>(q & 1) == ch) vkeep[it][q >> 1] = v;
                            else lds_st(buf + xaddr(i, q >> 1), v);
                        } else {
                            xout[static_cast<size_t>(2 * c) * C + chan] = v.x;
                            xout[static_cast<size_t>(2 * c + 1) * C + chan] = v.y;
                        }
                    }
                    carry[it][q] = o[q + HL];
                }
            }
        }
        if constexpr (kXch) {
            if (xch && emit) {
                lds_order();
                if (lane == 0) __hip_atomic_store(xflags + 2 * wave, xseq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (a wave's LDS operations complete in order)
                while (__hip_atomic_load(xflags + 2 * (wave ^ 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < xseq + 1) __builtin_amdgcn_s_sleep(1);
                lds_order();
                cf got[ITERL][HL / 2 > 0 ? HL / 2 : 1];
#pragma unroll
                for (int it = 0; it < ITERL; ++it) {
                    const int i = lane + 64 * it;
                    if ((it + 1) * 64 <= ML || i < ML) {
#pragma unroll
                        for (int j = 0; j < HL / 2; ++j) got[it][j] = lds_ld(pbuf + xaddr(i, j));
                    }
                }
#pragma unroll
                for (int it = 0; it < ITERL; ++it) {
                    const int i = lane + 64 * it;
                    if ((it + 1) * 64 <= ML || i < ML) {
#pragma unroll
                        for (int j = 0; j < HL / 2; ++j) {
                            const int c = i + (2 * j + static_cast<int>(ch)) * ML;   // frames 2c, 2c + 1, both channels: 16 bytes
                            const cf v0 = ch == 0 ? vkeep[it][j] : got[it][j], v1 = ch == 0 ? got[it][j] : vkeep[it][j];
                            if constexpr (stereo) {
                                ((GFloat4*)xout)[c] = f4{v0.x, v1.x, v0.y, v1.y};
                            } else {   // the pair's eight bytes in each of the two frames
                                const uint32_t fo = 2u * static_cast<uint32_t>(c) * C + pair2;
                                *(GFloat2*)(xout + fo) = f2{v0.x, v1.x};
                                *(GFloat2*)(xout + fo + C) = f2{v0.y, v1.y};
                            }
                        }
                    }
                }
                lds_order();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the neighbour's values are in registers
                if (lane == 0) __hip_atomic_store(xflags + 2 * (wave ^ 1u) + 1, xseq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ++xseq;
            }
        }
        }
        lds_order();
    }
    if (last == d.n_blocks) {
#pragma unroll
        for (int it = 0; it < CIT; ++it) {
            const int i = lane + 64 * it;
            if (i < CM) {
#pragma unroll
                for (int q = 0; q < HL; ++q) {
                    const int c = i + q * CM;
                    GFloat* ov = as_global(d.overlap_next);
                    ov[chan * FO + 2 * c] = carry[it][q].x;
                    ov[chan * FO + 2 * c + 1] = -carry[it][q].y;
                }
            }
        }
    }
}

typedef WavePlan<1176, 3, 7, 7, 8> W1176;   // 44.1 kHz side of the 44.1 <-> 48 kHz family
typedef WavePlan<1280, 4, 5, 8, 8> W1280;   // 48 kHz side
typedef WavePlan<512, 8, 8, 8> W512;        // the input block of the power-of-two families (x2, /2, x4, /4, x3, x1.5 ...)
typedef WavePlan<1024, 2, 8, 8, 8> W1024;
typedef WavePlan<256, 4, 8, 8> W256;
typedef WavePlan<128, 2, 8, 8> W128;
typedef WavePlan<64, 8, 8> W64;
typedef WavePlan<768, 3, 4, 8, 8> W768;
typedef WavePlan<1536, 3, 8, 8, 8> W1536;
typedef WavePlan<2048, 4, 8, 8, 8> W2048;
typedef WavePlan<3072, 2, 3, 8, 8, 8> W3072;   // (x6: six trips per stage -- one wave per SIMD with the whole register file)
typedef WavePlan<4096, 8, 8, 8, 8> W4096;      // (x8: three waves per CU are all the LDS holds)
typedef WavePlan<3528, 3, 3, 7, 7, 8> W3528;   // 88.2 kHz against the 16 / 32 kHz families
typedef WavePlan<4704, 3, 4, 7, 7, 8> W4704;   // 176.4 kHz (two waves per CU)
typedef WavePlan<5120, 2, 5, 8, 8, 8> W5120;   // 192 kHz
typedef WavePlan<588, 3, 4, 7, 7> W588;     // 22.05 kHz against the 48 kHz family (input side: the inverse's last radix must be even)
typedef WavePlan<882, 2, 3, 3, 7, 7> W882;
typedef WavePlan<1764, 3, 3, 4, 7, 7> W1764;
typedef WavePlan<2352, 2, 3, 7, 7, 8> W2352; // 88.2 kHz
typedef WavePlan<2560, 5, 8, 8, 8> W2560;    // 96 kHz
typedef WavePlan<640, 2, 5, 8, 8> W640;        // 16 kHz against the 44.1 kHz family

typedef void (*WaveKernel)(FftPlanDev, const FftStreamDesc*, uint32_t, uint32_t, uint32_t, uint32_t);
struct WaveChoice {
    WaveKernel fn = nullptr;
    size_t lds = 0;          // bytes of a workgroup
    uint32_t waves = 0;      // waves per workgroup
    uint32_t resident = 0;   // waves a CU holds at once
    int occ = 0;
};

// The instantiation for (FWD, INV) if the plan is that pair.  Waves per CU, by what the CU's LDS holds (one copy
// of the tables per workgroup + a buffer per wave) and what the registers allow: two-channel streams run 12
// (<= 168 registers) or 2 x 4 waves; streams of 4, 6, 8 .. channels run as channel pairs on the same code with 8-byte
// accesses (2 x 4 waves); the any-channel-count build (odd counts) needs ~250 registers (strided
// sample addressing; it spilled 290 bytes per lane under the 168 cap and ran 25-30 % slower,
// tools/fft_channels_bench.py) and runs 2 x 4.  Plans too long for that run one workgroup of up to 8 waves.
template <class FWD, class INV>
bool wave_choice(const FftPlanDev& plan, uint32_t channels, int occ_env, WaveChoice* out) {
    if (!FWD::matches(plan.fft_in, plan.n_stages_f, plan.radix_f) || !INV::matches(plan.fft_out, plan.n_stages_i, plan.radix_i))
        return false;
    constexpr size_t tables = static_cast<size_t>(FWD::kTw + INV::kTw + FWD::kRc + INV::kRc + (FWD::N < INV::N ? FWD::N + 1 : INV::N));   // (+ the filter bins in use)
    constexpr size_t buf = FWD::kBuf > INV::kBuf ? FWD::kBuf : INV::kBuf;
    constexpr size_t kCu = 160 * 1024 / sizeof(cf);
    constexpr size_t kFlags = 16;   // cf-sized words of exchange flags behind the buffers (two 32-bit words per wave, up to 16 waves)
    constexpr bool fit12 = tables + 12 * buf + kFlags <= kCu, fit4 = 2 * (tables + 4 * buf + kFlags) <= kCu;
    constexpr uint32_t wide_fit = (kCu - tables - kFlags) / buf < 8 ? static_cast<uint32_t>((kCu - tables - kFlags) / buf) : 8u;
    constexpr uint32_t wide = kOneWavePerSimd<FWD, INV> && wide_fit > 4 ? 4u : wide_fit;
    // (fewer than two waves per CU -- the longest plans in the exact build, whose twiddle rows are whole -- belong to
    // the workgroup kernels)
    if constexpr (!fit4 && wide < 2) {
        (void)channels; (void)occ_env; (void)out;
        return false;
    } else {
    static const bool no_c2 = rsmp::knob("RSMP_FFT_WAVE_NOC2") != nullptr;   // A/B: the any-channel-count build for two channels
    // (an even number of channels: channel pairs on the two-channel build)
    const bool paired = channels % 2 == 0 && !no_c2;
    // (the pairs build addresses its frames at a run-time stride: under the 168-register cap of twelve waves per CU it
    // spills 25 registers and runs 13 % slower than 2 x 4 waves with all of them -- 8 channels 0.84 against 0.73 ms)
    int occ = paired ? (fit12 && channels == 2 ? 3 : fit4 ? 2 : 1) : (fit4 ? 2 : 1);
    if (paired && ((occ_env == 3 && fit12) || (occ_env == 2 && fit4))) occ = occ_env;
    out->occ = occ;
    out->fn = nullptr;
    if (paired && channels == 2) {
        if constexpr (fit12) if (occ == 3) out->fn = fft_ola_wave_kernel<FWD, INV, 1, 3>;
        if constexpr (fit4) if (occ == 2) out->fn = fft_ola_wave_kernel<FWD, INV, 1, 2>;
        if constexpr (!fit4) if (occ == 1) out->fn = fft_ola_wave_kernel<FWD, INV, 1, 1>;
    } else if (paired) {
        if constexpr (fit12) if (occ == 3) out->fn = fft_ola_wave_kernel<FWD, INV, 2, 3>;
        if constexpr (fit4) if (occ == 2) out->fn = fft_ola_wave_kernel<FWD, INV, 2, 2>;
        if constexpr (!fit4) if (occ == 1) out->fn = fft_ola_wave_kernel<FWD, INV, 2, 1>;
    } else {
        if constexpr (fit4) out->fn = fft_ola_wave_kernel<FWD, INV, 0, 2>;
        else out->fn = fft_ola_wave_kernel<FWD, INV, 0, 1>;
    }
    static const uint32_t wide_knob = [] { const char* e = rsmp::knob("RSMP_FFT_WAVE_WIDE"); return e ? static_cast<uint32_t>(atoi(e)) : 0u; }();
    out->waves = occ == 3 ? 12u : occ == 2 ? 4u : (wide_knob >= 1 && wide_knob <= wide ? wide_knob : wide);
    out->resident = occ == 2 ? 8u : out->waves;
    out->lds = (tables + out->waves * buf + kFlags) * sizeof(cf);
    return out->fn != nullptr;
    }
}
template <class FWD, class... INVS>
bool wave_choices(const FftPlanDev& plan, uint32_t channels, int occ_env, WaveChoice* out) {
    return (wave_choice<FWD, INVS>(plan, channels, occ_env, out) || ...);
}

}  // namespace

// Wave-per-transform kernels exist for the 44.1 <-> 48 kHz family (both directions) and for the families whose
// input block is 512 frames (x2, /2, /4, /8, x3, x1.5 ...).  Returns hipErrorNotSupported when the plan is another
// one (the caller then uses the workgroup kernels).
hipError_t launch_fft_ola_wave(const FftPlanDev& plan, const FftStreamDesc* d_descs, uint32_t n_streams,
                               uint32_t max_blocks, uint32_t max_channels, uint32_t min_channels,
                               hipStream_t stream) {
    if (max_channels != min_channels) return hipErrorNotSupported;   // one wave layout per launch
    if (plan.n_rc_f != plan.fft_in / 2 - 1 || plan.n_rc_i != plan.fft_out / 2 - 1) return hipErrorNotSupported;
    if (plan.new_length != (plan.fft_in < plan.fft_out ? plan.fft_in + 1 : plan.fft_out)) return hipErrorNotSupported;
    const uint32_t C = max_channels;
    constexpr int occ_env = 0;
    WaveChoice wc;
    const bool found = wave_choices<W1176, W1280>(plan, C, occ_env, &wc) || wave_choices<W1280, W1176>(plan, C, occ_env, &wc) ||
                       wave_choices<W512, W64, W128, W256, W768, W1024, W1536, W2048, W3072, W4096>(plan, C, occ_env, &wc) ||
                       wave_choices<W768, W64, W128, W256, W512>(plan, C, occ_env, &wc) ||
                       wave_choices<W1536, W64, W128>(plan, C, occ_env, &wc) ||
                       wave_choices<W588, W1280, W2560>(plan, C, occ_env, &wc) || wave_choices<W882, W640, W1280>(plan, C, occ_env, &wc) ||
                       wave_choices<W1764, W640, W1280>(plan, C, occ_env, &wc) || wave_choices<W2352, W1280, W2560>(plan, C, occ_env, &wc) ||
                       wave_choice<W1176, W2560>(plan, C, occ_env, &wc) || wave_choice<W1280, W2352>(plan, C, occ_env, &wc) ||
                       wave_choices<W2560, W2352, W1176, W588>(plan, C, occ_env, &wc) || wave_choices<W640, W882, W1764, W3528>(plan, C, occ_env, &wc) || wave_choices<W3528, W640, W1280>(plan, C, occ_env, &wc) ||
                       wave_choices<W1280, W588, W882, W1764, W3528, W4704>(plan, C, occ_env, &wc) ||
                       wave_choices<W4704, W1280, W2560>(plan, C, occ_env, &wc) || wave_choices<W5120, W1176, W2352>(plan, C, occ_env, &wc) ||
                       wave_choice<W2560, W4704>(plan, C, occ_env, &wc) || wave_choice<W1176, W5120>(plan, C, occ_env, &wc) ||
                       wave_choice<W2352, W5120>(plan, C, occ_env, &wc) || wave_choice<W588, W5120>(plan, C, occ_env, &wc);
    if (!found) return hipErrorNotSupported;
    const uint32_t kWavesPerGroup = wc.waves;
    const size_t lds = wc.lds;
    WaveKernel fn = wc.fn;
    // Blocks per wave: every run after a stream's first recomputes its predecessor block (1 / run extra
    // work), and the launch ends with a partly filled round unless the number of waves is close to a
    // multiple of what the chip holds at once (3 workgroups of 4 waves per CU).
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const double slots = static_cast<double>(cus) * wc.resident;
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
    const uint32_t runs_per_stream = (max_blocks + run - 1) / run;
    const uint32_t total_waves = runs_per_stream * n_streams * C;
    const dim3 grid((total_waves + kWavesPerGroup - 1) / kWavesPerGroup);
    if (lds > 64 * 1024) {   // dynamic LDS above 64 KiB must be opted into
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, grid, dim3(kWavesPerGroup * 64), lds, stream, plan, d_descs, run, runs_per_stream,
                       total_waves, C / 2);   // (channel pairs: read by the two-channel build only)
    return hipGetLastError();
}

}  // namespace rsmp
