// fft_pair.hip -- ResamplerFft block pipeline for TWO-CHANNEL streams: ONE WAVE per stream, the two channels of a frame
// taken as ONE complex sample (gfx950).
//
// Replaces the same reference code as fft_wave.hip (FftResampler::resample, src/resampler_fft.rs:385-424; RadixFFT
// forward / inverse, src/fft/radix_fft.rs:476-670; the Stockham stages and butterflies) -- and with it the two
// real <-> complex passes (radix_fft.rs:500-537, :592-624; real_complex/mod.rs:37-114), which here do not exist.
//
// Why: the wave-per-channel kernel's time is the SUM of its vector instructions, its LDS reads and its LDS stores
// (profiles/r05/fft_slopes.txt: 0.175 / 0.67 / 0.93 us of launch per instruction and block-channel; nothing hides behind
// anything), and a third of its LDS traffic is not transform at all: the real-FFT post-process, the filter + inverse
// pre-process pass, the exchange that makes 16-byte stores out of two channels' values.  The resampler is linear with a
// REAL impulse response (resampler_fft.rs:361-376: the filter's spectrum H is the transform of real taps), so
//     resample(L + i R) = resample(L) + i resample(R):
// the interleaved stereo frame (L, R) IS the complex sample z = L + i R.  Per block of FI frames:
//   Z = DFT_2FI(z, zero padded)          -- as two FI-point transforms, decimation in frequency:
//       Z[2k'] = DFT_FI(z)[k'],  Z[2k' + 1] = DFT_FI(z w)[k'],  w[n] = exp(-2 pi i n / 2FI)   (the padding makes the
//       first radix-2 step trivial)
//   V[k] = H[k] Z[k] (k < NL),  V[2FO - k] = conj(H[k]) Z[2FI - k] (0 < k < NL),  0 elsewhere
//       (L's and R's spectra are Hermitian, their sum Z is not: both halves are carried; the truncation / zero extension of
//       resampler_fft.rs:396-408 keeps the parity of a bin, so even bins feed even bins)
//   v = IDFT_2FO(V)                      -- as two FO-point transforms, decimation in time:
//       F[n] = A[n] + u[n] B[n],  F[n + FO] = A[n] - u[n] B[n],  A = DFT_FO(conj V even), B = DFT_FO(conj V odd),
//       u[n] = exp(-2 pi i n / 2FO),  v = conj(F)
//   frame n of the output = v[n] + overlap[n]; v[FO + n] is the next overlap (resampler_fft.rs:416-423).
// So a wave runs the EVEN-bin chain start to finish (forward transform, filter, inverse transform: A in registers), then
// the ODD-bin chain, and the last inverse stage of the second chain combines, overlap-adds and stores whole frames.
// Per stream and block: 230 LDS stores instead of 322, ~440 reads instead of 514, ~2850 vector instructions instead of ~2860
// + no exchange between waves, every input frame one 8-byte complex load, every output frame one 8-byte store.  The
// transforms are the FI / FO-point plans of the wave-per-channel kernel (same stages, same padded layouts:
// fft_wave_core.h).
// Around that: every channel of every block is brought to its own level before the two share butterflies (PairScale: a
// quiet or silent channel beside a loud one, a NaN in one channel); a CU holds eight waves -- two per SIMD, served oldest
// first -- and a stream's blocks are cut into a LONG run for an old wave and a SHORT one for a young wave (the kernel body's
// run arithmetic, launch_fft_ola_pair); the output stores and the samples' second read are non-temporal (the block's samples
// stay in L2 between the chains); WAV PCM is converted where the frames are loaded (BITS).  DESIGN.md 4.4.
// Arithmetic: the reference's butterflies on other operands -- equal to the CPU path within rounding (tests/test_fft_gpu.py
// holds it to the gate of 1e-6 RMS; measured ~1.5e-7 like the wave-per-channel kernel), not bit for bit; the exact build
// (libresampler_amd_fftexact.so) never takes this kernel.
#include <cmath>
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(fast)

#include "fft_butterflies_pk.h"
#include "fft_kernels.h"
#include "common.h"

#ifndef RSMP_EXP
#define RSMP_EXP 0
#endif
#define RSMP_FEAT (RSMP_EXP & 63)   // A/B builds (make exp EXPFILE=fft_pair.hip)

// Diagnostic builds (tools/fft_trace.py, RSMP_EXP >> 6): 1 = every wave's start / end on the constant 100 MHz clock and where it
// ran; 2 = also the shader-clock cycles a wave spends in each phase of its blocks.
#if (RSMP_EXP >> 6) != 0
#define RSMP_FFT_TRACE 1
__device__ unsigned long long rsmp_fft_trace_buf[4096 * 16];
extern "C" int rsmp_debug_fft_trace(unsigned long long* out, size_t words) {
    return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(rsmp_fft_trace_buf), words * 8));
}
#endif
#if (RSMP_EXP >> 6) == 2
#define RSMP_TR(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); tr_ph[i] += t_ - tr_last; tr_last = t_; } while (0)
#else
#define RSMP_TR(i) do { } while (0)
#endif

namespace rsmp {

namespace {

#include "fft_wave_core.h"

// conj(a) * b = (a.x b.x + a.y b.y, a.x b.y - a.y b.x)
__device__ __forceinline__ cf cf_mul_cn(cf a, cf b) {
    cf d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]"
        : "=&v"(d) : "v"(a), "v"(b));
    return d;
}
// conj(a) * conj(b) = conj(a b) = (a.x b.x - a.y b.y, -(a.x b.y + a.y b.x))
__device__ __forceinline__ cf cf_mul_cc(cf a, cf b) {
    cf d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]"
        : "=&v"(d) : "v"(a), "v"(b));
    return d;
}

// conj(a) u + conj(c), component by component: (a.x u.x + c.x, -(a.y u.y) - c.y)
__device__ __forceinline__ cf cf_conj_scale_add_conj(cf a, cf u, cf c) {
    cf d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(u), "v"(c));
    return d;
}

// A value reduced over the wave's 64 lanes (row-wise DPP steps, then the rows' last lanes handed on): the sum, or the
// largest of non-negative values.  The result is uniform (read out of lane 63).
template <bool MAX>
__device__ __forceinline__ float wave_reduce(float v) {
    auto step = [](float x, auto ctrl_c, auto mask_c) {
        const int xi = __builtin_bit_cast(int, x);
        // (rows a step leaves out get 0 / themselves: bound_ctrl off, `old` = the neutral element)
        const int m = __builtin_amdgcn_update_dpp(MAX ? xi : 0, xi, decltype(ctrl_c)::value, decltype(mask_c)::value, 0xf, false);
        return MAX ? fmaxf(x, __builtin_bit_cast(float, m)) : x + __builtin_bit_cast(float, m);
    };
    v = step(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});   // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});   // row_mirror
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 into rows 1, 3
    v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 into rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// The two channels of a block share every butterfly, so what one channel's rounding leaves behind is relative to the
// LARGER channel.  Each channel of a block is therefore brought to a level near 1 by a power of two on the way in (exact)
// and taken back on the way out (exact): a channel 2^-20 below its partner, or silent, comes out as if it had been
// transformed alone.  `scale` multiplies the samples, `unscale` the outputs; a silent channel's `unscale` is zero (its
// outputs are exactly the overlap it carries), a non-finite channel's is NaN (bit 0 / 1 of `dead`: its samples are taken
// as zeros, and what it puts out -- this block and the overlap into the next -- is NaN like the reference's,
// resampler_fft.rs:385-424 with a NaN anywhere in the block).  The level is the block's ENERGY per channel, summed by the
// even chain's first pass, which has the whole block in registers (one packed multiply-add per sample); an energy that is
// not finite -- a NaN, an infinity, or samples beyond 1e17 -- or below 2^-80 -- a silent channel, or samples below 1e-14 --
// sends the block through the careful path (largest magnitude, NaN / infinity told from large by a sum of products with
// zero; a channel is silent only if its largest magnitude is zero).
struct PairScale {
    cf scale, unscale;
    uint32_t dead;
    // h = the power of two the channel's level is near: scale = 2^-h, unscale = 2^h (0 for a silent, NaN for a dead channel)
    static __device__ __forceinline__ void pow2(int h, bool silent, bool is_dead, float* sc, float* un) {
        h = h < -126 ? -126 : h > 126 ? 126 : h;
        *sc = __builtin_bit_cast(float, (127 - h) << 23);
        *un = is_dead ? __builtin_nanf("") : silent ? 0.f : __builtin_bit_cast(float, (127 + h) << 23);
    }
    // from the energies of the block's `log2n`-ish many samples (both finite)
    __device__ __forceinline__ void from_energy(float el, float er, int log2n) {
        auto h_of = [&](float e) { return (((__builtin_bit_cast(int, e) >> 23) & 0xff) - 127 - log2n) >> 1; };
        float a, b, c, d2;
        pow2(h_of(el), el == 0.f, false, &a, &b);
        pow2(h_of(er), er == 0.f, false, &c, &d2);
        scale = cf_make(a, c);
        unscale = cf_make(b, d2);
    }
    // from the largest magnitudes (of the channels that are not dead)
    __device__ __forceinline__ void from_max(float ml, float mr) {
        auto h_of = [](float m) { return ((__builtin_bit_cast(int, m) >> 23) & 0xff) - 127; };
        float a, b, c, d2;
        pow2(h_of(ml), ml == 0.f, (dead & 1u) != 0, &a, &b);
        pow2(h_of(mr), mr == 0.f, (dead & 2u) != 0, &c, &d2);
        scale = cf_make(a, c);
        unscale = cf_make(b, d2);
    }
    __device__ __forceinline__ cf kill(cf z) const {   // a dead channel's samples are zeros
        return cf_make((dead & 1u) ? 0.f : z.x, (dead & 2u) ? 0.f : z.y);
    }
};
struct PairPrep {
    PairScale* ps;
    template <int ITER, int K, int M> __device__ __forceinline__ void run(cf (&v)[ITER][K], int lane) const {
        cf en = cf_make(0.f, 0.f);
#pragma unroll
        for (int it = 0; it < ITER; ++it)
            if ((it + 1) * 64 <= M || lane + 64 * it < M)
#pragma unroll
                for (int k = 0; k < K; ++k) en = v[it][k] * v[it][k] + en;
        const float el = wave_reduce<false>(en.x), er = wave_reduce<false>(en.y);
        ps->dead = 0;
        constexpr int kLog2N = 31 - __builtin_clz(static_cast<unsigned>(K * M));
        // (an energy of zero is a silent channel OR samples below 1e-22, whose squares are lost: told apart by the careful path)
        constexpr float kTiny = 8.2718061e-25f;   // 2^-80
        if (__builtin_isfinite(el) && __builtin_isfinite(er) && el >= kTiny && er >= kTiny) {
            ps->from_energy(el, er, kLog2N);
        } else {   // (a wave-uniform branch few blocks take)
            asm volatile("; a channel of this block is silent or tiny, not finite, or beyond 1e17");
            float ml = 0.f, mr = 0.f;
            cf nf = cf_make(0.f, 0.f), zero = cf_make(0.f, 0.f);   // nf += v * 0: NaN from the first NaN or infinity of a channel on
            asm volatile("" : "+v"(zero));
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                if ((it + 1) * 64 <= M || lane + 64 * it < M) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        ml = __builtin_fmaxf(ml, __builtin_fabsf(v[it][k].x));
                        mr = __builtin_fmaxf(mr, __builtin_fabsf(v[it][k].y));
                        nf = v[it][k] * zero + nf;
                    }
                }
            }
            const uint32_t dead = (__builtin_amdgcn_ballot_w64(nf.x != nf.x) != 0 ? 1u : 0u) | (__builtin_amdgcn_ballot_w64(nf.y != nf.y) != 0 ? 2u : 0u);
            ps->dead = dead;
#pragma unroll
            for (int it = 0; it < ITER; ++it)
#pragma unroll
                for (int k = 0; k < K; ++k) v[it][k] = ps->kill(v[it][k]);
            ps->from_max((dead & 1u) ? 0.f : wave_reduce<true>(ml), (dead & 2u) ? 0.f : wave_reduce<true>(mr));
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int k = 0; k < K; ++k) v[it][k] = v[it][k] * ps->scale;
    }
};

// The last stage of a plan (stride = M: a butterfly's points are its own) with its outputs left in registers: butterfly
// i = lane + 64 it yields o[it][q] = X[i + q M].  Every read is issued before the first butterfly (what follows overwrites
// the buffer).  QS / IPP: where the stage's inputs lie (wave_stage).
template <int N, int R, int QS, int IPP>
__device__ __forceinline__ void wave_last_regs(const cf* buf, const cf* __restrict__ tw, int lane, cf (&o)[(N / R + 63) / 64][R]) {
    constexpr int M = N / R;
    constexpr int ITER = (M + 63) / 64;
    constexpr int ROW = fetch_count(R) | 1;
    cf raw[ITER][kFetch<R>];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = lane + 64 * it;
        if ((it + 1) * 64 <= M || i < M) {
#pragma unroll
            for (int q = 0; q < R; ++q) o[it][q] = lds_ld(buf + i + (IPP ? i / (IPP ? IPP : 1) : 0) + q * QS);
            twiddle_fetch<R>(tw + i * ROW, raw[it]);
        }
    }
    lds_order();
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = lane + 64 * it;
        if ((it + 1) * 64 <= M || i < M) {
            cf twr[R], t[R];
            twiddle_expand<R>(raw[it], twr);
            t[0] = o[it][0];
#pragma unroll
            for (int q = 1; q < R; ++q) t[q] = cf_mul(twr[q], o[it][q]);
            pdft<R>(t, o[it]);
        }
    }
}

// Where the bins of the forward transform go (header): bin k = 2 k' + par of Z is kept as a positive frequency for
// k <= kPosMax and as a negative one for k >= kNegMin (up-sampling: every bin, and bin FI -- the input's Nyquist bin -- as
// both; down-sampling: bins of NL - 1 and below either way, resampler_fft.rs:396-408); a kept bin lands at k' (positive) or
// k' + FO - FI (negative) of the chain's FO-point inverse transform, what lies between reads as zero.
template <int FI, int FO>
struct PairBins {
    static constexpr bool kUp = FI < FO;
    static constexpr int kNL = kUp ? FI + 1 : FO;
    static constexpr int kPosMax = kNL - 1;
    static constexpr int kNegMin = 2 * FI - (kNL - 1);
    static constexpr int kShift = FO - FI;
    static constexpr int zero_lo(int par) { return (kPosMax - par) / 2 + 1; }                       // first index nobody writes
    static constexpr int zero_hi(int par) {                                                           // last one
        const int kp = (kNegMin - par + 1) / 2;                                                       // smallest k' with 2 k' + par >= kNegMin
        const int lo_neg = kp + kShift;
        const int nyq = (kUp && par == 0 && FI % 2 == 0) ? FI / 2 + kShift : lo_neg;                  // (bin FI as a negative frequency)
        return (nyq < lo_neg ? nyq : lo_neg) - 1;
    }
};

// BITS: 0 = f32 frames; 16 / 24 / 32 = little-endian WAV PCM frames of that width, converted where they are loaded
// exactly as resample/src/main.rs:128-137 converts a sample (`sample as f32 / (1 << (bits - 1)) as f32`, the 32-bit divisor
// an i32 literal = -2^31): bit for bit what rsmp_pcm_to_stereo_f32_device + the f32 launch give.
// Waves per CU: eight (two per SIMD, an old and a young one) where the LDS holds the tables and eight buffers, else four.
template <class FWD, class INV>
constexpr int pair_waves() {
    constexpr size_t buf = FWD::kBuf > INV::kBuf ? FWD::kBuf : INV::kBuf;
    constexpr size_t tables = static_cast<size_t>(FWD::kTw + INV::kTw + PairBins<FWD::N, INV::N>::kNL + FWD::N + INV::N);
    // (plans of up to 512 points need ~105 registers: sixteen waves, four per SIMD, hide more of their many short passes)
    if (FWD::N <= 768 && INV::N <= 512 && (tables + 16 * buf) * sizeof(cf) <= 160 * 1024) return 16;
    // (twelve waves -- 168 registers -- for the plans of up to 1024 points: 26-55 spilled registers, 512 -> 1024 frames 0.72 ->
    // 0.77 ms, 640 -> 882 0.76 -> 0.87: measured, not kept)
    return (tables + 8 * buf) * sizeof(cf) <= 160 * 1024 ? 8 : (tables + 4 * buf) * sizeof(cf) <= 160 * 1024 ? 4 : 0;
}
template <class FWD, class INV, int BITS>
__global__ __launch_bounds__((pair_waves<FWD, INV>() ? pair_waves<FWD, INV>() : 4) * 64, 1) void fft_ola_pair_kernel(FftPlanDev plan, const FftStreamDesc* __restrict__ descs,
                                                              uint32_t run0, uint32_t run1, uint32_t run2, uint32_t run3,
                                                              uint32_t pairs_per_stream, uint32_t total_waves) {
    extern __shared__ __attribute__((aligned(16))) cf lds2[];
    constexpr int kWaves = pair_waves<FWD, INV>() ? pair_waves<FWD, INV>() : 4;
    constexpr int FI = FWD::N, FO = INV::N;
    typedef PairBins<FI, FO> Bins;
    constexpr int NL = Bins::kNL;
    constexpr int LDSC = FWD::kBuf > INV::kBuf ? FWD::kBuf : INV::kBuf;
    constexpr int SF = FWD::kStages, SI = INV::kStages;
    constexpr int RLF = FWD::kR[SF - 1], MF = FI / RLF, ITF = (MF + 63) / 64;   // last forward stage
    constexpr int RLI = INV::kR[SI - 1], MI = FO / RLI, ITI = (MI + 63) / 64;   // last inverse stage
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t gw = blockIdx.x * kWaves + wave;
    (void)gw;
    // tables, once per workgroup: stage twiddles (rows re-spaced as the stages read them), the filter bins in use, and
    // the two chirps w (FI values) and u (FO values)
    constexpr int kFilterEven = (NL + 1) / 2;   // even bins 0, 2, ..
    constexpr int kTabF = 0, kTabI = kTabF + FWD::kTw, kTabFilter = kTabI + INV::kTw, kTabW = kTabFilter + NL,
                  kTabU = kTabW + FI, kTabEnd = kTabU + FO;
    cf* tab = lds2;
    {
        auto copy = [&](cf* dst, const cf* __restrict__ src, int n) {
            for (int i = threadIdx.x; i < n; i += kWaves * 64) dst[i] = src[i];
        };
        auto rows = [&](cf* dst, const cf* __restrict__ src, int n_rows, int len, int keep, int pitch) {
            for (int i = threadIdx.x; i < n_rows * keep; i += kWaves * 64) {
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
        // (the filter bins a chain multiplies are every other one: kept by parity -- even bins, then odd bins -- so that a
        // chain's lanes read neighbouring values, not every second one: two lanes per bank pair otherwise)
        for (int i = threadIdx.x; i < NL; i += kWaves * 64)
            tab[kTabFilter + (i & 1) * kFilterEven + (i >> 1)] = reinterpret_cast<const cf*>(plan.filter)[i];
        copy(tab + kTabW, reinterpret_cast<const cf*>(plan.chirp_f), FI);
        copy(tab + kTabU, reinterpret_cast<const cf*>(plan.chirp_i), FO);
    }
    __syncthreads();
    if ((blockIdx.x * 4u + (wave & 3u)) * static_cast<uint32_t>(kWaves / 4) >= total_waves) return;   // (a pair beyond the launch's last)
#ifdef RSMP_FFT_TRACE
    const unsigned long long tr_t0 = wall_clock64();
    unsigned long long tr_ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tr_last = __builtin_readcyclecounter();
    (void)tr_ph; (void)tr_last;
#endif
    cf* buf = lds2 + kTabEnd + wave * LDSC;
    const cf* tw_f = tab + kTabF;
    const cf* tw_i = tab + kTabI;
    const cf* filter = tab + kTabFilter;
    const cf* chirp_w = tab + kTabW;
    const cf* chirp_u = tab + kTabU;

    // A SIMD serves its two waves oldest first: the first four waves of a workgroup run nearly unimpeded, the last four in
    // their gaps, and with equal runs the launch ended with one wave per SIMD for its last quarter (tools/fft_trace.py:
    // ends at 323 / 429 us; evening the two out by priority gave nothing -- two equals get in each other's way).  So a
    // stream is cut into PAIRS of runs, a long one for an old wave and a short one for a young wave.
    // (four waves per CU: one per SIMD, every run a long one)
    const uint32_t kind = wave >> 2;                                  // 0: old wave, long run; 1: young wave, short run
    const uint32_t pair_idx = blockIdx.x * 4u + (wave & 3u);
    const uint32_t stream_idx = pair_idx / pairs_per_stream;
    const uint32_t in_stream = pair_idx - stream_idx * pairs_per_stream;
    const FftStreamDesc d = descs[stream_idx];
    // (sixteen waves per CU: four ages, four run lengths)
    const uint32_t before = kind == 0 ? 0u : kind == 1 ? run0 : kind == 2 ? run0 + run1 : run0 + run1 + run2;
    const uint32_t first = in_stream * (run0 + run1 + run2 + run3) + before;
    const uint32_t run = kind == 0 ? run0 : kind == 1 ? run1 : kind == 2 ? run2 : run3;
    if (first >= d.n_blocks || run == 0) return;
    const uint32_t last = first + run < d.n_blocks ? first + run : d.n_blocks;  // exclusive

    // The overlap carried into the run: the stream state ([2][FO] reals, resampler_fft.rs:51), or the predecessor block
    // recomputed (not emitted).  Held as F[FO + n] = conj(v[FO + n]), n = lane + 64 it + q MI.
    cf carry[ITI][RLI];
    {
        // (every load issued before the first is waited for: a branch per value made them wait for each other, 2 x 24 HBM
        // round trips in a row for the wave that starts a stream)
        const GFloat* ov = as_global(d.overlap);
        const bool have = first == 0;
        float ca[ITI][RLI], cb[ITI][RLI];
#pragma unroll
        for (int it = 0; it < ITI; ++it) {
            const int i = lane + 64 * it < MI ? lane + 64 * it : MI - 1;
#pragma unroll
            for (int q = 0; q < RLI; ++q) {
                ca[it][q] = have ? ov[i + q * MI] : 0.f;
                cb[it][q] = have ? ov[FO + i + q * MI] : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < ITI; ++it)
#pragma unroll
            for (int q = 0; q < RLI; ++q) carry[it][q] = lane + 64 * it < MI ? cf_make(ca[it][q], -cb[it][q]) : cf_make(0.f, 0.f);
    }
    const int64_t b_begin = first == 0 ? 0 : static_cast<int64_t>(first) - 1;

    for (int64_t b = b_begin; b < static_cast<int64_t>(last); ++b) {
        const bool emit = b >= static_cast<int64_t>(first);
        // frame j of the block as the complex sample (channel 0, channel 1)
        typedef __attribute__((address_space(1))) uint32_t GU32;
        typedef uint32_t u2v __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(1))) u2v GU2;
        const __attribute__((address_space(1))) char* xraw = (const __attribute__((address_space(1))) char*)d.in +
                                                             static_cast<size_t>(b) * FI * (BITS == 0 ? 8 : 2 * (BITS / 8));
        auto frame = [&](int j) -> cf {
            if constexpr (BITS == 0) {
                const f2 v = ((const GFloat2*)xraw)[j];
                return cf_make(v.x, v.y);
            } else if constexpr (BITS == 16) {
                const uint32_t w = ((const GU32*)xraw)[j];
                return cf_make(static_cast<float>(static_cast<int32_t>(w << 16) >> 16), static_cast<float>(static_cast<int32_t>(w) >> 16)) * (1.0f / 32768.0f);
            } else if constexpr (BITS == 24) {   // six bytes at an even address: the eight bytes at the word boundary below it
                static_assert(BITS != 24 || FI % 2 == 0, "a block's last frame starts at an odd half word: its load ends with the block");
                typedef u2v __attribute__((address_space(1), aligned(4))) GU2a4;
                const uint32_t byte = 6u * static_cast<uint32_t>(j);
                const u2v w = *(const GU2a4*)(xraw + (byte & ~3u));
                const bool odd = (byte & 2u) != 0;
                const uint32_t a = odd ? (w.x >> 16) | (w.y << 16) : w.x;
                const uint32_t c = odd ? w.y >> 8 : (w.x >> 24) | (w.y << 8);
                return cf_make(static_cast<float>(static_cast<int32_t>(a << 8) >> 8), static_cast<float>(static_cast<int32_t>(c << 8) >> 8)) * (1.0f / 8388608.0f);
            } else {
                const u2v w = ((const GU2*)xraw)[j];
                return cf_make(static_cast<float>(static_cast<int32_t>(w.x)), static_cast<float>(static_cast<int32_t>(w.y))) * (-1.0f / 2147483648.0f);
            }
        };
        // (the odd chain's read is the samples' last: non-temporal, like the output stores -- HBM traffic 1.19x -> 1.05x
        // algorithmic with both, profiles/r05/fft_traffic_nt.txt)
        auto frame_last = [&](int j) -> cf {
#if !(RSMP_FEAT & 2)
            if constexpr (BITS == 0) {
                const f2 v = __builtin_nontemporal_load((const GFloat2*)xraw + j);
                return cf_make(v.x, v.y);
            } else
#endif
            return frame(j);
        };
        GFloat2* xout = (GFloat2*)(as_global(d.out) + static_cast<size_t>(b) * FO * 2);
        cf A[ITI][RLI];   // the even chain's outputs, kept while the odd chain runs
        PairScale ps;
        ps.dead = 0;
        static_for<0, 2>([&](auto par_c) {
            constexpr int par = decltype(par_c)::value;
            // ---- forward FI-point transform of z (even bins) or z w (odd bins)
            {
                auto first_pass = [&](auto&& smp, auto prep) {
                    if constexpr (FWD::kFused) wave_fused_first<FI, FWD::kR[0], FWD::kR[1], FWD::kPadJ>(buf, tw_f + FWD::tab(1), lane, smp, prep);
                    else wave_first<FI, FWD::kR[0], FWD::kPadJ>(buf, lane, smp, prep);
                };
                if constexpr (par == 0) {
                    first_pass(frame, PairPrep{&ps});
                } else if (ps.dead == 0) {
                    first_pass([&](int j) -> cf { return cf_mul(lds_ld(chirp_w + j), frame_last(j) * ps.scale); }, NoPrep{});
                } else {   // (a pass of its own for the block with a NaN in it: no select per sample in everybody's path)
                    asm volatile("; a channel of this block is not finite");
                    first_pass([&](int j) -> cf { return cf_mul(lds_ld(chirp_w + j), ps.kill(frame_last(j)) * ps.scale); }, NoPrep{});
                }
            }
            RSMP_TR(6 * par + 0);
            static_for<(FWD::kFused ? 2 : 1), SF - 1>([&](auto s_c) {
                constexpr int s = decltype(s_c)::value;
                wave_stage<FI, FWD::kR[s], FWD::stride(s), FI / FWD::kR[s] + FWD::in_pad(s), FWD::out_pad(s), FWD::in_period(s)>(buf, tw_f + FWD::tab(s), lane);
            });
            RSMP_TR(6 * par + 1);
            // ---- last forward stage in registers; times the filter; conj(V) of the chain into the buffer, in index order
            {
                cf z[ITF][RLF], hv[ITF][RLF];
                wave_last_regs<FI, RLF, MF + FWD::in_pad(SF - 1), FWD::in_period(SF - 1)>(buf, tw_f + FWD::tab(SF - 1), lane, z);
#pragma unroll
                for (int it = 0; it < ITF; ++it) {
                    const int i = lane + 64 * it;
                    if ((it + 1) * 64 <= MF || i < MF) {
#pragma unroll
                        for (int q = 0; q < RLF; ++q) {
                            const int k = 2 * (i + q * MF) + par;
                            const int hi = k <= Bins::kPosMax ? k : 2 * FI - k;   // (of the chain's parity either way)
                            const int hc = hi < NL ? hi : NL - 1 - ((NL - 1 - par) & 1);
                            hv[it][q] = lds_ld(filter + par * kFilterEven + (hc >> 1));
                        }
                    }
                }
                lds_order();
#pragma unroll
                for (int it = 0; it < ITF; ++it) {
                    const int i = lane + 64 * it;
                    if ((it + 1) * 64 <= MF || i < MF) {
#pragma unroll
                        for (int q = 0; q < RLF; ++q) {
                            const int kp = i + q * MF, k = 2 * kp + par;
                            // what the bins of this q are, for every lane of the trip (k runs over [klo, khi])
                            const int klo = 2 * (64 * it + q * MF) + par;
                            const int khi = 2 * (((it + 1) * 64 <= MF ? 64 * it + 63 : MF - 1) + q * MF) + par;
                            const bool all_pos = khi <= Bins::kPosMax, no_pos = klo > Bins::kPosMax;
                            const bool all_neg = klo >= Bins::kNegMin, no_neg = khi < Bins::kNegMin;
                            // conj(V) = conj(H Z) (positive), = H conj(Z) (negative: V = conj(H) Z)
                            if (!no_pos) {
                                const cf u = cf_mul_cc(z[it][q], hv[it][q]);
                                if (all_pos || k <= Bins::kPosMax) lds_st(buf + kp, u);
                            }
                            if (!no_neg) {
                                // (bin FI of an up-sampling plan is both: its filter value is the one fetched above)
                                const cf u = cf_mul_cn(z[it][q], hv[it][q]);
                                if (all_neg || k >= Bins::kNegMin) lds_st(buf + kp + Bins::kShift, u);
                            }
                        }
                    }
                }
                lds_order();
            }
            RSMP_TR(6 * par + 2);
            // ---- inverse FO-point transform of the chain's bins (forward butterflies on conjugated input)
            {
                constexpr int ZLO = Bins::zero_lo(par), ZHI = Bins::zero_hi(par);
                auto from_lds = [&](int j) -> cf {
                    cf v = lds_ld(buf + j);
                    if (ZLO <= ZHI && j >= ZLO && j <= ZHI) v = cf_make(0.f, 0.f);
                    return v;
                };
                if constexpr (INV::kFused) wave_fused_first<FO, INV::kR[0], INV::kR[1], INV::kPadJ>(buf, tw_i + INV::tab(1), lane, from_lds);
                else wave_first<FO, INV::kR[0], INV::kPadJ>(buf, lane, from_lds);
            }
            RSMP_TR(6 * par + 3);
            static_for<(INV::kFused ? 2 : 1), SI - 1>([&](auto s_c) {
                constexpr int s = decltype(s_c)::value;
                wave_stage<FO, INV::kR[s], INV::stride(s), FO / INV::kR[s] + INV::in_pad(s), INV::out_pad(s), INV::in_period(s)>(buf, tw_i + INV::tab(s), lane);
            });
            RSMP_TR(6 * par + 4);
            if constexpr (par == 0) {
                wave_last_regs<FO, RLI, MI + INV::in_pad(SI - 1), INV::in_period(SI - 1)>(buf, tw_i + INV::tab(SI - 1), lane, A);
            } else {
                cf B[ITI][RLI], uv[ITI][RLI];
#pragma unroll
                for (int it = 0; it < ITI; ++it) {
                    const int i = lane + 64 * it;
                    if ((it + 1) * 64 <= MI || i < MI) {
#pragma unroll
                        for (int q = 0; q < RLI; ++q) uv[it][q] = lds_ld(chirp_u + i + q * MI);
                    }
                }
                wave_last_regs<FO, RLI, MI + INV::in_pad(SI - 1), INV::in_period(SI - 1)>(buf, tw_i + INV::tab(SI - 1), lane, B);
                // F[n] = A + u B -> frame n = conj(F[n]) + conj(carry[n]); F[n + FO] = A - u B is the next carry
#pragma unroll
                for (int it = 0; it < ITI; ++it) {
                    const int i = lane + 64 * it;
                    if ((it + 1) * 64 <= MI || i < MI) {
#pragma unroll
                        for (int q = 0; q < RLI; ++q) {
                            const cf t = cf_mul(uv[it][q], B[it][q]);
                            if (emit) {
                                const cf v = cf_conj_scale_add_conj(A[it][q] + t, ps.unscale, carry[it][q]);
                                // (non-temporal: the output does not push the block's samples out of L2 before the odd chain reads
                                // them again -- HBM reads 1.34x -> 1.11x the input, profiles/r05/fft_traffic_nt.txt; the time is the same)
#if RSMP_FEAT & 1
                                xout[i + q * MI] = f2{v.x, v.y};
#else
                                __builtin_nontemporal_store(f2{v.x, v.y}, xout + i + q * MI);
#endif
                            }
                            carry[it][q] = (A[it][q] - t) * ps.unscale;
                        }
                    }
                }
            }
            lds_order();
            RSMP_TR(6 * par + 5);
        });
    }
#ifdef RSMP_FFT_TRACE
    if (lane == 0 && gw < 4096) {
        uint32_t hw_id, xcc_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
        unsigned long long* t = rsmp_fft_trace_buf + static_cast<size_t>(gw) * 16;
        t[0] = tr_t0;
        t[1] = wall_clock64();
        t[2] = (static_cast<unsigned long long>(xcc_id) << 32) | hw_id;
        t[3] = static_cast<unsigned long long>(last - b_begin);
        for (int i = 0; i < 12; ++i) t[4 + i] = tr_ph[i];
    }
#endif
    if (last == d.n_blocks) {
#pragma unroll
        for (int it = 0; it < ITI; ++it) {
            const int i = lane + 64 * it;
            if (i < MI) {
#pragma unroll
                for (int q = 0; q < RLI; ++q) {
                    const int n = i + q * MI;
                    GFloat* ov = as_global(d.overlap_next);
                    ov[n] = carry[it][q].x;
                    ov[FO + n] = -carry[it][q].y;
                }
            }
        }
    }
}

typedef WavePlan<1176, 3, 7, 7, 8> W1176;   // 44.1 kHz side of the 44.1 <-> 48 kHz family
typedef WavePlan<1280, 4, 5, 8, 8> W1280;   // 48 kHz side
// (the other plans of up to 2048 points, as in fft_wave.hip)
typedef WavePlan<512, 8, 8, 8> W512;
typedef WavePlan<1024, 2, 8, 8, 8> W1024;
typedef WavePlan<256, 4, 8, 8> W256;
typedef WavePlan<128, 2, 8, 8> W128;
typedef WavePlan<64, 8, 8> W64;
typedef WavePlan<768, 3, 4, 8, 8> W768;
typedef WavePlan<1536, 3, 8, 8, 8> W1536;
typedef WavePlan<588, 3, 4, 7, 7> W588;
typedef WavePlan<882, 2, 3, 3, 7, 7> W882;
typedef WavePlan<1764, 3, 3, 4, 7, 7> W1764;
typedef WavePlan<640, 2, 5, 8, 8> W640;

typedef void (*PairKernel)(FftPlanDev, const FftStreamDesc*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t);

struct PairChoice {
    PairKernel fn = nullptr;
    size_t lds = 0;
    uint32_t waves = 0;   // per workgroup = per CU
};
template <class FWD, class INV>
bool pair_choice(const FftPlanDev& plan, uint32_t pcm_bits, PairChoice* out) {
    if (!FWD::matches(plan.fft_in, plan.n_stages_f, plan.radix_f) || !INV::matches(plan.fft_out, plan.n_stages_i, plan.radix_i))
        return false;
    constexpr int waves = pair_waves<FWD, INV>();
    if constexpr (waves == 0) {
        (void)pcm_bits; (void)out;
        return false;
    } else {
        constexpr size_t buf = FWD::kBuf > INV::kBuf ? FWD::kBuf : INV::kBuf;
        constexpr size_t tables = static_cast<size_t>(FWD::kTw + INV::kTw + PairBins<FWD::N, INV::N>::kNL + FWD::N + INV::N);
        out->fn = pcm_bits == 16 ? fft_ola_pair_kernel<FWD, INV, 16> : pcm_bits == 24 ? fft_ola_pair_kernel<FWD, INV, 24>
                  : pcm_bits == 32 ? fft_ola_pair_kernel<FWD, INV, 32> : fft_ola_pair_kernel<FWD, INV, 0>;
        out->lds = (tables + waves * buf) * sizeof(cf);
        out->waves = waves;
        return true;
    }
}

template <class FWD, class... INVS>
bool pair_choices(const FftPlanDev& plan, uint32_t pcm_bits, PairChoice* out) {
    return (pair_choice<FWD, INVS>(plan, pcm_bits, out) || ...);
}

}  // namespace

// One wave per two-channel stream and run of blocks.  hipErrorNotSupported when the plan is not one of the pairs above
// (the caller then uses the wave-per-channel kernels).
hipError_t launch_fft_ola_pair(const FftPlanDev& plan, const FftStreamDesc* d_descs, uint32_t n_streams, uint32_t max_blocks,
                               hipStream_t stream, uint32_t pcm_bits) {
    if (pcm_bits != 0 && pcm_bits != 16 && pcm_bits != 24 && pcm_bits != 32) return hipErrorNotSupported;
    if (plan.chirp_f == nullptr || plan.chirp_i == nullptr) return hipErrorNotSupported;
    if (plan.new_length != (plan.fft_in < plan.fft_out ? plan.fft_in + 1 : plan.fft_out)) return hipErrorNotSupported;
    PairChoice pc;
    const bool found = pair_choice<W1176, W1280>(plan, pcm_bits, &pc) || pair_choice<W1280, W1176>(plan, pcm_bits, &pc)
#if RSMP_EXP != 0   // (timing experiments instantiate the 44.1 <-> 48 kHz pair alone)
                       ;
#else
                       // (by tools/fft_pairs_bench.py, both kernels in one lease -- profiles/r05/fft_pairs_pair_vs_wave.txt: the
                       // down-sampling pairs gain 8 - 21 %, 512 -> 1024 frames 4 %; a 1764-point inverse, 512 -> 1536 / 2048,
                       // 768 -> 256 / 512, 882 -> 1280 and 1764 -> 1280 frames spill or run four waves and stay with fft_wave.hip)
                       || pair_choices<W512, W64, W128, W256, W768, W1024>(plan, pcm_bits, &pc) || pair_choices<W768, W64, W128, W256, W512>(plan, pcm_bits, &pc) ||
                       pair_choices<W1536, W64, W128>(plan, pcm_bits, &pc) || pair_choices<W588, W1280>(plan, pcm_bits, &pc) ||
                       pair_choices<W882, W640>(plan, pcm_bits, &pc) || pair_choices<W1764, W640>(plan, pcm_bits, &pc) ||
                       pair_choices<W640, W882>(plan, pcm_bits, &pc) || pair_choices<W1280, W588, W882>(plan, pcm_bits, &pc);
#endif
    if (!found) return hipErrorNotSupported;
    // Pairs of runs per stream: every run after a stream's first recomputes its predecessor block (1 / run extra work), and
    // the launch ends with a partly filled round unless the number of waves is close to a multiple of what the chip holds
    // (8 per CU).  Of a pair's blocks the old wave takes kLongShare (its share of a SIMD while both waves run).
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t classes = pc.waves / 4;   // waves per SIMD: ages
    const double slots = static_cast<double>(cus) * pc.waves;
    static const double share_knob = [] { const char* e = rsmp::knob("RSMP_FFT_PAIR_SHARE"); return e ? atof(e) : 0.0; }();   // A/B
    // the share of a SIMD each age gets while all of them run (two ages: by a sweep on the 44.1 -> 48 kHz launch; four: the
    // same falling series, RSMP_FFT_PAIR_SHARE = its ratio)
    double share[4] = {1.0, 0.0, 0.0, 0.0};
    if (classes == 2) {
        share[0] = share_knob > 0.0 && share_knob < 1.0 ? share_knob : 0.6;
        share[1] = 1.0 - share[0];
    } else if (classes >= 3) {
        const double r = share_knob > 0.0 && share_knob <= 1.0 ? share_knob : 0.8;
        double sum = 0.0;
        for (uint32_t c = 0; c < classes; ++c) sum += std::pow(r, static_cast<double>(c));
        for (uint32_t c = 0; c < classes; ++c) share[c] = std::pow(r, static_cast<double>(c)) / sum;
    }
    uint32_t both = 32;
    double best = -1.0;
    for (uint32_t cand = 6 * classes; cand <= 64 * classes; ++cand) {   // blocks of a group of runs (one per age)
        const double pairs = static_cast<double>((max_blocks + cand - 1) / cand);
        const double waves = classes * pairs * n_streams;
        const double rounds = std::ceil(waves / slots);
        const double useful = static_cast<double>(max_blocks) / (max_blocks + classes * pairs - 1.0);   // halo blocks
        const double score = waves / (rounds * slots) * useful;
        if (score > best + 1e-9) { best = score; both = cand; }
    }
    const uint32_t pairs_per_stream = (max_blocks + both - 1) / both;
    // (the halo block is part of a wave's work: the shares are of both + classes)
    uint32_t runs[4] = {0, 0, 0, 0}, given = 0;
    for (uint32_t c = 0; c + 1 < classes; ++c) {
        const long v = std::lround(share[c] * (both + static_cast<double>(classes)) - 1.0);
        runs[c] = static_cast<uint32_t>(v < 1 ? 1 : v);
        if (given + runs[c] > both) runs[c] = both - given;
        given += runs[c];
    }
    runs[classes - 1] = both - given;
    const uint32_t total_waves = pairs_per_stream * n_streams * classes;
    const dim3 grid((total_waves + pc.waves - 1) / pc.waves);
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pc.fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pc.fn, grid, dim3(pc.waves * 64), pc.lds, stream, plan, d_descs, runs[0], runs[1], runs[2], runs[3], pairs_per_stream, total_waves);
    return hipGetLastError();
}

}  // namespace rsmp
