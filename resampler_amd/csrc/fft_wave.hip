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
// Round 5 (tools/fft_trace.py, profiles/r05/fft_slopes.txt): a SIMD serves its waves oldest first -- three waves with equal
// runs ended at 66 / 80 / 97 % of the launch -- so the waves of a SIMD publish their block counts in LDS and the one behind
// raises its priority; the two waves of a stream touch the next block's lines into L2 a block ahead.  Two-channel streams
// of the plan pairs it is built for go to fft_pair.hip (a wave per stream, the frame as one complex sample) instead.
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

#ifndef RSMP_EXP
#define RSMP_EXP 0   // A/B builds (make exp EXPFILE=fft_wave.hip): timing experiments, never shipped
#endif
// RSMP_EXP = features + 64 * trace.  Features (bits): 1 = NO priority feedback between the waves of a SIMD; 4 = NO touch of
// the next block's samples into L2 (the two things tools/fft_trace.py found, switched off again for an A/B);
// 8 / 16 = every LDS store / read issued twice, 32 = two more packed instructions per complex multiply (what a store, a read,
// a vector instruction costs the launch: profiles/r05/fft_slopes.txt).  Trace (tools/fft_trace.py): 1 = every wave's start /
// end on the constant 100 MHz clock and where it ran; 2 = also the shader-clock cycles a wave spends in each phase of its
// blocks (the reads of the clock drain the LDS queue at every phase boundary: the phases' shares are what it is for, not the
// total).
#define RSMP_FEAT (RSMP_EXP & 63)
#define RSMP_PRIO (!(RSMP_FEAT & 1))
#define RSMP_TOUCH (!(RSMP_FEAT & 4))
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
#if RSMP_PRIO
    // Waves of one SIMD are served oldest first: of the three waves a SIMD holds the oldest ends its run a third earlier
    // than the youngest (tools/fft_trace.py), and the launch ends with one wave per SIMD.  Each wave publishes its block
    // count; a wave behind the others of its SIMD raises its priority (s_setprio), the one ahead lowers it.
    uint32_t* wsimd = xflags + 32;   // SIMD id of each wave of the workgroup
    uint32_t* wprog = xflags + 48;   // blocks done
    uint32_t my_simd;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 4, 2)" : "=s"(my_simd));
    if (lane == 0) { wsimd[wave] = my_simd; wprog[wave] = 0; }
#endif
    __syncthreads();
#if RSMP_PRIO
    uint32_t peer_a = wave, peer_b = wave;
    for (uint32_t w = 0; w < static_cast<uint32_t>(kWavesPerGroup); ++w) {
        const uint32_t sd = __builtin_amdgcn_readfirstlane(wsimd[w]);
        if (w != wave && sd == my_simd) {
            if (peer_a == wave) peer_a = w; else peer_b = w;
        }
    }
#endif
    if (gw >= total_waves) {
#if RSMP_PRIO
        if (lane == 0) wprog[wave] = 0xffffffffu;
#endif
        return;
    }
#ifdef RSMP_FFT_TRACE
    const unsigned long long tr_t0 = wall_clock64();
    unsigned long long tr_ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tr_last = __builtin_readcyclecounter();
    (void)tr_ph; (void)tr_last;
#endif
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
    if (first >= d.n_blocks) {
#if RSMP_PRIO
        if (lane == 0) wprog[wave] = 0xffffffffu;
#endif
        return;
    }
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

#if RSMP_TOUCH
    float pf = 0.f;   // the touch of the next block's lines (a load nobody reads: it is waited for a block later, behind younger stores)
#endif
    for (int64_t b = b_begin; b < static_cast<int64_t>(last); ++b) {
        const bool emit = b >= static_cast<int64_t>(first);
#if RSMP_TOUCH
        asm volatile("" :: "v"(pf));
#endif
        if constexpr (kXch) {   // the neighbour must have taken the previous block's outgoing values out of this buffer
            if (xch && xseq > 0)
                while (__hip_atomic_load(xflags + 2 * wave + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < xseq) __builtin_amdgcn_s_sleep(1);
        }
        RSMP_TR(0);
#if (RSMP_EXP >> 6) == 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the previous block's stores: the first pass's loads wait behind them -- vmcnt counts in order)
        RSMP_TR(10);
#endif
#if RSMP_PRIO
        {
            const uint32_t mine = static_cast<uint32_t>(b - b_begin);
            if (lane == 0) __hip_atomic_store(wprog + wave, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t pa = __builtin_amdgcn_readfirstlane(__hip_atomic_load(wprog + peer_a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            const uint32_t pb = __builtin_amdgcn_readfirstlane(__hip_atomic_load(wprog + peer_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            const uint32_t ahead = (peer_a != wave && pa > mine ? 1u : 0u) + (peer_b != wave && pb > mine ? 1u : 0u);   // waves of this SIMD that have done more
            if (ahead == 0) __builtin_amdgcn_s_setprio(0);
            else if (ahead == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(2);
        }
#endif
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
#if RSMP_TOUCH
        if constexpr (stereo) {
            // The first pass of every block waits for HBM (28 % of a wave's time, tools/fft_trace.py).  The block after this
            // one is touched into L2 now: the two waves of a stream take every other 128-byte line of its 9.4 KB.
            // A block is FI frames of two samples of 4 bytes (f32) or in_bits / 8 bytes (PCM): the touch stays inside it.
            if (b + 1 < static_cast<int64_t>(d.n_blocks)) {
                const uint32_t bits = __builtin_amdgcn_readfirstlane(d.in_bits);
                const uint32_t block_bytes = static_cast<uint32_t>(FI) * 2u * (bits ? bits >> 3 : 4u);   // (FI even: a multiple of 4)
                const GFloat* nx = (const GFloat*)((const __attribute__((address_space(1))) char*)as_global(d.in) + static_cast<size_t>(b + 1) * block_bytes);
                const uint32_t fl = (2u * static_cast<uint32_t>(lane) + ch) * 32u;   // in 4-byte words: every other 128-byte line
                if (fl * 4u < block_bytes) pf = nx[fl];
            }
        }
#endif
        RSMP_TR(1);
        static_for<(FWD::kFused ? 2 : 1), SF>([&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            wave_stage<FI, FWD::kR[s], FWD::stride(s), FI / FWD::kR[s] + FWD::in_pad(s), FWD::out_pad(s), FWD::in_period(s)>(buf, tw_f + FWD::tab(s), lane);
            RSMP_TR(s);   // (2, 3)
        });
        wave_postprocess<FI>(buf, rc_f, lane);
        RSMP_TR(4);
        wave_filter_preprocess<FO, kFilterLen, kFilterLen - 1>(buf, filter, rc_i, lane);
        RSMP_TR(5);

        // ---- inverse transform, in place; its last stage below
        {
            auto from_lds = [&](int j) -> cf { return lds_ld(buf + j); };
            if constexpr (INV::kFused) wave_fused_first<FO, INV::kR[0], INV::kR[1], INV::kPadJ>(buf, tw_i + INV::tab(1), lane, from_lds);
            else wave_first<FO, INV::kR[0], INV::kPadJ>(buf, lane, from_lds);
        }
        RSMP_TR(6);
        static_for<(INV::kFused ? 2 : 1), (kOddLast ? SI : SI - 1)>([&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            wave_stage<FO, INV::kR[s], INV::stride(s), FO / INV::kR[s] + INV::in_pad(s), INV::out_pad(s), INV::in_period(s)>(buf, tw_i + INV::tab(s), lane);
        });
        RSMP_TR(7);
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
        RSMP_TR(8);
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
        RSMP_TR(9);
    }
#if RSMP_PRIO
    if (lane == 0) __hip_atomic_store(wprog + wave, 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
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
    constexpr size_t kFlags = 32;   // cf-sized words behind the buffers: exchange flags (two 32-bit words per wave, up to 16 waves), then the waves' SIMD ids and block counts
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

bool fft_wave_is_exact() {
#ifdef RSMP_FFT_WAVE_EXACT
    return true;
#else
    return false;
#endif
}

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
    const bool found = wave_choices<W1176, W1280>(plan, C, occ_env, &wc) || wave_choices<W1280, W1176>(plan, C, occ_env, &wc)
#if RSMP_EXP != 0   // (timing experiments instantiate the 44.1 <-> 48 kHz pair alone: 25 s instead of 160 s per build)
                       ;
#else
                       ||
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
#endif
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
