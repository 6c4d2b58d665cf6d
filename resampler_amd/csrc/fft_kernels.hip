// fft_kernels.hip -- ResamplerFft block pipeline for gfx950, entirely in LDS.
//
// Replaces FftResampler::resample (src/resampler_fft.rs:385-424) and everything below it:
// RadixFFT::process forward / inverse (src/fft/radix_fft.rs:476-670), the Stockham stage loop
// (src/fft/stockham_autosort.rs:169-247), the radix-2/3/4/5/7/8 butterflies
// (src/fft/butterflies/butterflyN/mod.rs, scalar specs) and the real<->complex passes
// (src/fft/real_complex/mod.rs:37-114), plus the deinterleave / interleave copies of
// ResamplerFft::resample (resampler_fft.rs:197-202, :232-237), which disappear into the first
// load and the last store.
//
// One workgroup = one stream x a run of consecutive blocks; per block and channel:
//   load (zero padded, 2 reals = 1 complex) -> forward Stockham stages ping-ponging between two
//   LDS arrays -> real-FFT post-process -> x filter spectrum, truncate / zero-extend ->
//   inverse-real pre-process + conjugate -> inverse Stockham stages -> conjugate, overlap-add with
//   the carry kept in LDS, interleaved store.
// HBM traffic is the algorithmic minimum: fft_in reads + fft_out writes per block-channel (the
// overlap row only at the ends of a launch); twiddles and the filter spectrum (~30 KB) stay in L2.
// Channels are computed independently (what the reference yields per channel when its scratch
// regions do not collide).
#include "fft_kernels.h"
#include "common.h"
#include "fft_butterflies.h"

#include <cmath>

#include <cstdlib>

namespace rsmp {

namespace {

constexpr int kFftThreads = 256;

// One out-of-place Stockham stage: butterfly i reads src[i + q*m], twiddles inputs 1..R-1 with
// w[(i mod stride)*(R-1) + q-1] (none when stride == 1) and writes dst[R*i - (R-1)*k + q*stride]
// (butterfly4/mod.rs:316-320 etc.).
template <int R, int THREADS = kFftThreads>
__device__ __forceinline__ void stage(const float2* __restrict__ src, float2* __restrict__ dst,
                                      uint32_t n, uint32_t stride, const float2* __restrict__ tw) {
    const uint32_t m = n / R;
    for (uint32_t i = threadIdx.x; i < m; i += THREADS) {
        const uint32_t k = stride == 1 ? 0u : i % stride;
        float2 t[R], o[R];
#pragma unroll
        for (int q = 0; q < R; ++q) t[q] = src[i + q * m];
        if (stride != 1) {
            const float2* w = tw + k * (R - 1);
#pragma unroll
            for (int q = 1; q < R; ++q) t[q] = cmul(w[q - 1], t[q]);
        }
        dft<R>(t, o);
        float2* d = dst + R * i - (R - 1) * k;
#pragma unroll
        for (int q = 0; q < R; ++q) d[q * stride] = o[q];
    }
}

// stockham_autosort (stockham_autosort.rs:169-247): returns the buffer holding the result.
template <int THREADS = kFftThreads>
__device__ float2* stockham(float2* a, float2* b, uint32_t n, uint32_t n_stages,
                            const uint32_t* radix, const uint32_t* tw_off,
                            const float2* __restrict__ tw) {
    uint32_t stride = 1;
    for (uint32_t s = 0; s < n_stages; ++s) {
        const uint32_t r = radix[s];
        const float2* w = tw + tw_off[s];
        switch (r) {
            case 2: stage<2, THREADS>(a, b, n, stride, w); break;
            case 3: stage<3, THREADS>(a, b, n, stride, w); break;
            case 4: stage<4, THREADS>(a, b, n, stride, w); break;
            case 5: stage<5, THREADS>(a, b, n, stride, w); break;
            case 7: stage<7, THREADS>(a, b, n, stride, w); break;
            default: stage<8, THREADS>(a, b, n, stride, w); break;
        }
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
        stride *= r;
    }
    return a;
}

// postprocess_fft (radix_fft.rs:500-537 + real_complex/mod.rs:37-74), in place on x[0 .. n2].
template <int THREADS = kFftThreads>
__device__ void postprocess_forward(float2* x, uint32_t n2, const float2* __restrict__ rc, uint32_t n_rc) {
    const uint32_t len = n2 + 1, split = len / 2;
    uint32_t iters = split - 1;                       // left middle
    const uint32_t rm_len = len - split - 1;          // right middle
    if (rm_len < iters) iters = rm_len;
    if (n_rc < iters) iters = n_rc;
    if (threadIdx.x == 0) {
        const float2 z0 = x[0];
        x[0] = make_float2(z0.x + z0.y, 0.0f);
        x[n2] = make_float2(z0.x - z0.y, 0.0f);
    }
    for (uint32_t i = threadIdx.x; i < iters; i += THREADS) {
        const uint32_t l = 1 + i, rr = n2 - 1 - i;
        const float2 o = x[l], orv = x[rr], tw = rc[i];
        const float2 sum = cadd(o, orv), diff = csub(o, orv);
        const float half_sum_real = 0.5f * sum.x, half_diff_imag = 0.5f * diff.y;
        const float real = sum.y * tw.x + diff.x * tw.y;
        const float imag = sum.y * tw.y - diff.x * tw.x;
        x[l] = make_float2(half_sum_real + real, half_diff_imag + imag);
        x[rr] = make_float2(half_sum_real - real, imag - half_diff_imag);
    }
    if ((len & 1u) && threadIdx.x == 32) x[len / 2].y = -x[len / 2].y;
}

// preprocess_ifft (radix_fft.rs:592-624 + real_complex/mod.rs:84-114) followed by the input
// conjugation of process_inverse_complex (:634-637), in place on y[0 .. n2].
template <int THREADS = kFftThreads>
__device__ void preprocess_inverse(float2* y, uint32_t n2, const float2* __restrict__ rc, uint32_t n_rc) {
    const uint32_t len = n2 + 1, split = len / 2;
    uint32_t iters = split - 1;
    const uint32_t rm_len = len - split - 1;
    if (rm_len < iters) iters = rm_len;
    if (n_rc < iters) iters = n_rc;
    if (threadIdx.x == 0) {
        const float2 a = y[0], b = y[n2];
        const float2 first_sum = cadd(a, b), first_diff = csub(a, b);
        y[0] = make_float2(first_sum.x - first_sum.y, first_diff.x - first_diff.y);
    }
    for (uint32_t i = threadIdx.x; i < iters; i += THREADS) {
        const uint32_t l = 1 + i, rr = n2 - 1 - i;
        const float2 a = y[l], b = y[rr], tw = rc[i];
        const float2 sum = cadd(a, b), diff = csub(a, b);
        const float real = sum.y * tw.x + diff.x * tw.y;
        const float imag = sum.y * tw.y - diff.x * tw.x;
        y[l] = make_float2(sum.x - real, diff.y - imag);
        y[rr] = make_float2(sum.x + real, -imag - diff.y);
    }
    if ((len & 1u) && threadIdx.x == 32) {
        const float2 c = y[len / 2];
        const float2 dbl = cadd(c, c);
        y[len / 2] = make_float2(dbl.x, -dbl.y);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n2; i += THREADS) y[i].y = -y[i].y;
}

// THREADS = 256, PER_CHANNEL = false: a workgroup walks the channels of a block one after the other.
// THREADS = 64, PER_CHANNEL = true: a one-wave workgroup per (stream, run of blocks, channel) -- its barriers cost
// nothing, and as many of them share a CU as the LDS holds (blocks of 512 -> 256 frames: 16): transforms too small
// to occupy 256 threads (64 radix-8 butterflies per stage) run up to twice as fast this way.
template <int THREADS, bool PER_CHANNEL>
__global__ __launch_bounds__(THREADS) void fft_ola_kernel(FftPlanDev plan,
                                                          const FftStreamDesc* __restrict__ descs,
                                                          uint32_t run) {
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    const FftStreamDesc d = descs[blockIdx.y];
    const uint32_t first = blockIdx.x * run;
    if (first >= d.n_blocks) return;
    const uint32_t last = first + run < d.n_blocks ? first + run : d.n_blocks;  // exclusive
    const uint32_t C = d.channels, fi = plan.fft_in, fo = plan.fft_out;
    if (PER_CHANNEL && blockIdx.z >= C) return;
    const uint32_t c_begin = PER_CHANNEL ? blockIdx.z : 0u, c_end = PER_CHANNEL ? blockIdx.z + 1u : C;
    float2* bufA = lds2;
    float2* bufB = lds2 + plan.lds_complex;
    float* carry = reinterpret_cast<float*>(lds2 + 2 * plan.lds_complex);   // [C][fo], or [fo] of the workgroup's channel
    const uint32_t carry_base = PER_CHANNEL ? c_begin * fo : 0u;           // index of carry[0] in the stream's overlap rows
    const uint32_t carry_len = (c_end - c_begin) * fo;

    // overlap carried into the run: the stream state, or the predecessor block recomputed
    if (first == 0)
        for (uint32_t e = threadIdx.x; e < carry_len; e += THREADS) carry[e] = d.overlap[carry_base + e];
    const int64_t b_begin = first == 0 ? 0 : static_cast<int64_t>(first) - 1;
    __syncthreads();

    for (int64_t b = b_begin; b < static_cast<int64_t>(last); ++b) {
        const bool emit = b >= static_cast<int64_t>(first);
        const float* __restrict__ xin = d.in + static_cast<size_t>(b) * fi * C;
        float* __restrict__ xout = d.out + static_cast<size_t>(b) * fo * C;
        for (uint32_t c = c_begin; c < c_end; ++c) {
            // resampler_fft.rs:387-388: fi reals + fi zeros, viewed as fi complexes (radix_fft.rs:552-554)
            for (uint32_t i = threadIdx.x; i < fi; i += THREADS) {
                float2 v = make_float2(0.f, 0.f);
                if (2 * i + 1 < fi) v = make_float2(xin[(2 * i) * C + c], xin[(2 * i + 1) * C + c]);
                else if (2 * i < fi) v = make_float2(xin[(2 * i) * C + c], 0.f);
                bufA[i] = v;
            }
            __syncthreads();
            float2* X = stockham<THREADS>(bufA, bufB, fi, plan.n_stages_f, plan.radix_f, plan.tw_off_f, plan.tw_f);
            float2* Y = X == bufA ? bufB : bufA;
            postprocess_forward<THREADS>(X, fi, plan.rc_f, plan.n_rc_f);
            __syncthreads();
            // resampler_fft.rs:401-408: multiply new_length bins, zero the rest up to fo
            for (uint32_t k = threadIdx.x; k <= fo; k += THREADS)
                Y[k] = k < plan.new_length ? cmul(X[k], plan.filter[k]) : make_float2(0.f, 0.f);
            __syncthreads();
            preprocess_inverse<THREADS>(Y, fo, plan.rc_i, plan.n_rc_i);
            __syncthreads();
            float2* Z = stockham<THREADS>(Y, X, fo, plan.n_stages_i, plan.radix_i, plan.tw_off_i, plan.tw_i);
            // output conjugation (radix_fft.rs:656-669), reals 2i, 2i+1 <- Z[i]; overlap-add (:416-423)
            float* ov = carry + (c - c_begin) * fo;
            if (emit)
                for (uint32_t t = threadIdx.x; t < fo; t += THREADS) {
                    const float2 z = Z[t >> 1];
                    const float y = (t & 1u) ? -z.y : z.x;
                    xout[t * C + c] = y + ov[t];
                }
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < fo; t += THREADS) {
                const float2 z = Z[(t + fo) >> 1];
                ov[t] = ((t + fo) & 1u) ? -z.y : z.x;
            }
            __syncthreads();
        }
    }
    if (last == d.n_blocks)
        for (uint32_t e = threadIdx.x; e < carry_len; e += THREADS) d.overlap_next[carry_base + e] = carry[e];
}

// ---- plan-specialised kernel -----------------------------------------------------------------------
// The same pipeline with the transform lengths and stage radices as template parameters: trip
// counts, strides, divisors and twiddle offsets fold into immediates and the stage loop disappears.
// The generic kernel above issues one instruction per 4 cycles per SIMD, 30 % of them scalar loop /
// dispatch / spill traffic -- it is bound by instruction issue, not by LDS, HBM or the ALUs -- so
// what a plan-specific build removes is exactly what it is short of.  Instantiated for the
// 44.1 <-> 48 kHz family (1176 = 3.7.7.8 and 1280 = 4.5.8.8 complex points); every other plan runs
// on the generic kernel.  Arithmetic, operation order and tables are identical.
template <int N, int R, int STRIDE>
__device__ __forceinline__ void stage_ct(const float2* __restrict__ src, float2* __restrict__ dst,
                                         const float2* __restrict__ tw) {
    constexpr int M = N / R;
    constexpr int ITER = (M + kFftThreads - 1) / kFftThreads;
#pragma unroll 1   // unrolled, the two butterflies of a 2-trip stage cost a workgroup per CU in registers
    for (int it = 0; it < ITER; ++it) {
        const int i = static_cast<int>(threadIdx.x) + it * kFftThreads;
        if ((it + 1) * kFftThreads <= M || i < M) {
            const int k = STRIDE == 1 ? 0 : i % STRIDE;
            float2 t[R], o[R];
#pragma unroll
            for (int q = 0; q < R; ++q) t[q] = src[i + q * M];
            if constexpr (STRIDE != 1) {
                const float2* w = tw + k * (R - 1);
#pragma unroll
                for (int q = 1; q < R; ++q) t[q] = cmul(w[q - 1], t[q]);
            }
            dft<R>(t, o);
            float2* d = dst + R * i - (R - 1) * k;
#pragma unroll
            for (int q = 0; q < R; ++q) d[q * STRIDE] = o[q];
        }
    }
}

template <int N, int STRIDE, int TWOFF, int R, int... Rest>
struct StagesCt {
    static __device__ __forceinline__ float2* run(float2* a, float2* b, const float2* __restrict__ tw) {
        stage_ct<N, R, STRIDE>(a, b, tw + TWOFF);
        __syncthreads();
        if constexpr (sizeof...(Rest) == 0) {
            return b;
        } else {
            return StagesCt<N, STRIDE * R, TWOFF + (STRIDE == 1 ? 0 : STRIDE * (R - 1)), Rest...>::run(b, a, tw);
        }
    }
};
template <int N, int... Rs> struct PlanCt {
    static constexpr int kN = N;
    static constexpr int kStages = sizeof...(Rs);
    static __device__ __forceinline__ float2* run(float2* a, float2* b, const float2* __restrict__ tw) {
        return StagesCt<N, 1, 0, Rs...>::run(a, b, tw);
    }
    static bool matches(uint32_t n, uint32_t n_stages, const uint32_t* radix) {
        const uint32_t want[] = {static_cast<uint32_t>(Rs)...};
        if (n != static_cast<uint32_t>(N) || n_stages != sizeof...(Rs)) return false;
        for (uint32_t s = 0; s < n_stages; ++s)
            if (radix[s] != want[s]) return false;
        return true;
    }
};

// postprocess_forward / preprocess_inverse with the length known (every plan the resampler builds
// has n2 / 2 - 1 twiddles: the pair loop covers all bins)
template <int N2>
__device__ __forceinline__ void postprocess_forward_ct(float2* x, const float2* __restrict__ rc) {
    constexpr int LEN = N2 + 1, SPLIT = LEN / 2, ITERS = SPLIT - 1;
    constexpr int TRIPS = (ITERS + kFftThreads - 1) / kFftThreads;
    if (threadIdx.x == 0) {
        const float2 z0 = x[0];
        x[0] = make_float2(z0.x + z0.y, 0.0f);
        x[N2] = make_float2(z0.x - z0.y, 0.0f);
    }
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
                if (it > 0) __builtin_amdgcn_sched_barrier(0);
        const int i = static_cast<int>(threadIdx.x) + it * kFftThreads;
        if (i < ITERS) {
            const int l = 1 + i, rr = N2 - 1 - i;
            const float2 o = x[l], orv = x[rr], tw = rc[i];
            const float2 sum = cadd(o, orv), diff = csub(o, orv);
            const float half_sum_real = 0.5f * sum.x, half_diff_imag = 0.5f * diff.y;
            const float real = sum.y * tw.x + diff.x * tw.y;
            const float imag = sum.y * tw.y - diff.x * tw.x;
            x[l] = make_float2(half_sum_real + real, half_diff_imag + imag);
            x[rr] = make_float2(half_sum_real - real, imag - half_diff_imag);
        }
    }
    if ((LEN & 1) && threadIdx.x == 32) x[LEN / 2].y = -x[LEN / 2].y;
}

template <int N2>
__device__ __forceinline__ void preprocess_inverse_ct(float2* y, const float2* __restrict__ rc) {
    constexpr int LEN = N2 + 1, SPLIT = LEN / 2, ITERS = SPLIT - 1;
    constexpr int TRIPS = (ITERS + kFftThreads - 1) / kFftThreads;
    if (threadIdx.x == 0) {
        const float2 a = y[0], b = y[N2];
        const float2 first_sum = cadd(a, b), first_diff = csub(a, b);
        y[0] = make_float2(first_sum.x - first_sum.y, first_diff.x - first_diff.y);
    }
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
                if (it > 0) __builtin_amdgcn_sched_barrier(0);
        const int i = static_cast<int>(threadIdx.x) + it * kFftThreads;
        if (i < ITERS) {
            const int l = 1 + i, rr = N2 - 1 - i;
            const float2 a = y[l], b = y[rr], tw = rc[i];
            const float2 sum = cadd(a, b), diff = csub(a, b);
            const float real = sum.y * tw.x + diff.x * tw.y;
            const float imag = sum.y * tw.y - diff.x * tw.x;
            y[l] = make_float2(sum.x - real, diff.y - imag);
            y[rr] = make_float2(sum.x + real, -imag - diff.y);
        }
    }
    if ((LEN & 1) && threadIdx.x == 32) {
        const float2 c = y[LEN / 2];
        const float2 dbl = cadd(c, c);
        y[LEN / 2] = make_float2(dbl.x, -dbl.y);
    }
    __syncthreads();
    constexpr int T2 = (N2 + kFftThreads - 1) / kFftThreads;
#pragma unroll
    for (int it = 0; it < T2; ++it) {
                if (it > 0) __builtin_amdgcn_sched_barrier(0);
        const int i = static_cast<int>(threadIdx.x) + it * kFftThreads;
        if (i < N2) y[i].y = -y[i].y;
    }
}

template <class FWD, class INV>
__global__ __launch_bounds__(kFftThreads) void fft_ola_kernel_ct(FftPlanDev plan,
                                                                 const FftStreamDesc* __restrict__ descs,
                                                                 uint32_t run) {
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    constexpr int FI = FWD::kN, FO = INV::kN;
    constexpr int LDSC = (FI > FO ? FI : FO) + 1;
    const FftStreamDesc d = descs[blockIdx.y];
    const uint32_t first = blockIdx.x * run;
    if (first >= d.n_blocks) return;
    const uint32_t last = first + run < d.n_blocks ? first + run : d.n_blocks;  // exclusive
    const uint32_t C = d.channels;
    float2* bufA = lds2;
    float2* bufB = lds2 + LDSC;
    float* carry = reinterpret_cast<float*>(lds2 + 2 * LDSC);   // [C][FO]

    if (first == 0)
        for (uint32_t e = threadIdx.x; e < C * FO; e += kFftThreads) carry[e] = d.overlap[e];
    const int64_t b_begin = first == 0 ? 0 : static_cast<int64_t>(first) - 1;
    __syncthreads();

    for (int64_t b = b_begin; b < static_cast<int64_t>(last); ++b) {
        const bool emit = b >= static_cast<int64_t>(first);
        const float* __restrict__ xin = d.in + static_cast<size_t>(b) * FI * C;
        float* __restrict__ xout = d.out + static_cast<size_t>(b) * FO * C;
        for (uint32_t c = 0; c < C; ++c) {
            constexpr int TI = (FI + kFftThreads - 1) / kFftThreads;
#pragma unroll
            for (int it = 0; it < TI; ++it) {
                if (it > 0) __builtin_amdgcn_sched_barrier(0);
                const int i = static_cast<int>(threadIdx.x) + it * kFftThreads;
                if (i < FI) {
                    float2 v = make_float2(0.f, 0.f);
                    if (2 * i + 1 < FI) v = make_float2(xin[(2 * i) * C + c], xin[(2 * i + 1) * C + c]);
                    else if (2 * i < FI) v = make_float2(xin[(2 * i) * C + c], 0.f);
                    bufA[i] = v;
                }
            }
            __syncthreads();
            float2* X = FWD::run(bufA, bufB, plan.tw_f);
            float2* Y = X == bufA ? bufB : bufA;
            postprocess_forward_ct<FI>(X, plan.rc_f);
            __syncthreads();
            constexpr int TK = (FO + 1 + kFftThreads - 1) / kFftThreads;
#pragma unroll
            for (int it = 0; it < TK; ++it) {
                if (it > 0) __builtin_amdgcn_sched_barrier(0);
                const uint32_t k = threadIdx.x + it * kFftThreads;
                if (k <= FO) Y[k] = k < plan.new_length ? cmul(X[k], plan.filter[k]) : make_float2(0.f, 0.f);
            }
            __syncthreads();
            preprocess_inverse_ct<FO>(Y, plan.rc_i);
            __syncthreads();
            float2* Z = INV::run(Y, X, plan.tw_i);
            float* ov = carry + c * FO;
            constexpr int TO = (FO + kFftThreads - 1) / kFftThreads;
            if (emit) {
#pragma unroll
                for (int it = 0; it < TO; ++it) {
                if (it > 0) __builtin_amdgcn_sched_barrier(0);
                    const uint32_t t = threadIdx.x + it * kFftThreads;
                    if (t < FO) {
                        const float2 z = Z[t >> 1];
                        const float y = (t & 1u) ? -z.y : z.x;
                        xout[t * C + c] = y + ov[t];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < TO; ++it) {
                if (it > 0) __builtin_amdgcn_sched_barrier(0);
                const uint32_t t = threadIdx.x + it * kFftThreads;
                if (t < FO) {
                    const float2 z = Z[(t + FO) >> 1];
                    ov[t] = ((t + FO) & 1u) ? -z.y : z.x;
                }
            }
            __syncthreads();
        }
    }
    if (last == d.n_blocks)
        for (uint32_t e = threadIdx.x; e < C * FO; e += kFftThreads) d.overlap_next[e] = carry[e];
}

// ---- two channels per phase --------------------------------------------------------------------------
// A barrier-separated phase of this pipeline costs about a microsecond however little it computes
// (per-wave trace), so the stereo build does the work of both channels between two barriers: half
// the barriers per transform, two independent butterflies per thread to overlap, and the
// interleaved frames are loaded and stored as whole float4 (two frames x two channels) instead of
// channel-strided scalars.  Arithmetic per channel is unchanged.
template <int N, int STRIDE, int TWOFF, int R, int... Rest>
struct StagesCt2 {
    static __device__ __forceinline__ bool run(float2* a0, float2* b0, float2* a1, float2* b1,
                                               const float2* __restrict__ tw) {
        stage_ct<N, R, STRIDE>(a0, b0, tw + TWOFF);
        stage_ct<N, R, STRIDE>(a1, b1, tw + TWOFF);
        __syncthreads();
        if constexpr (sizeof...(Rest) == 0) {
            return true;   // result in the b buffers
        } else {
            return !StagesCt2<N, STRIDE * R, TWOFF + (STRIDE == 1 ? 0 : STRIDE * (R - 1)), Rest...>::run(b0, a0, b1, a1, tw);
        }
    }
};

template <int N2>
__device__ __forceinline__ void conj_ct(float2* y) {
    constexpr int T2 = (N2 + kFftThreads - 1) / kFftThreads;
#pragma unroll
    for (int it = 0; it < T2; ++it) {
        const int i = static_cast<int>(threadIdx.x) + it * kFftThreads;
        if (i < N2) y[i].y = -y[i].y;
    }
}

// preprocess_inverse_ct without its trailing barrier + conjugation (the caller runs both channels, then
// one barrier, then conj_ct on both)
template <int N2>
__device__ __forceinline__ void preprocess_inverse_head_ct(float2* y, const float2* __restrict__ rc) {
    constexpr int LEN = N2 + 1, SPLIT = LEN / 2, ITERS = SPLIT - 1;
    constexpr int TRIPS = (ITERS + kFftThreads - 1) / kFftThreads;
    if (threadIdx.x == 0) {
        const float2 a = y[0], b = y[N2];
        const float2 first_sum = cadd(a, b), first_diff = csub(a, b);
        y[0] = make_float2(first_sum.x - first_sum.y, first_diff.x - first_diff.y);
    }
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
        const int i = static_cast<int>(threadIdx.x) + it * kFftThreads;
        if (i < ITERS) {
            const int l = 1 + i, rr = N2 - 1 - i;
            const float2 a = y[l], b = y[rr], tw = rc[i];
            const float2 sum = cadd(a, b), diff = csub(a, b);
            const float real = sum.y * tw.x + diff.x * tw.y;
            const float imag = sum.y * tw.y - diff.x * tw.x;
            y[l] = make_float2(sum.x - real, diff.y - imag);
            y[rr] = make_float2(sum.x + real, -imag - diff.y);
        }
    }
    if ((LEN & 1) && threadIdx.x == 32) {
        const float2 c = y[LEN / 2];
        const float2 dbl = cadd(c, c);
        y[LEN / 2] = make_float2(dbl.x, -dbl.y);
    }
}

template <class P> struct Stages2Of;
template <int N, int... Rs> struct Stages2Of<PlanCt<N, Rs...>> { typedef StagesCt2<N, 1, 0, Rs...> type; };

template <class FWD, class INV>
__global__ __launch_bounds__(kFftThreads) void fft_ola_kernel_ct2(FftPlanDev plan,
                                                                  const FftStreamDesc* __restrict__ descs,
                                                                  uint32_t run) {
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    constexpr int FI = FWD::kN, FO = INV::kN;
    constexpr int LDSC = (FI > FO ? FI : FO) + 1;
    static_assert(FI % 2 == 0 && FO % 2 == 0, "frame pairs");
    const FftStreamDesc d = descs[blockIdx.y];   // d.channels == 2
    const uint32_t first = blockIdx.x * run;
    if (first >= d.n_blocks) return;
    const uint32_t last = first + run < d.n_blocks ? first + run : d.n_blocks;  // exclusive
    float2* A0 = lds2;
    float2* B0 = lds2 + LDSC;
    float2* A1 = lds2 + 2 * LDSC;
    float2* B1 = lds2 + 3 * LDSC;
    float* carry = reinterpret_cast<float*>(lds2 + 4 * LDSC);   // [2][FO]

    if (first == 0)
        for (uint32_t e = threadIdx.x; e < 2 * FO; e += kFftThreads) carry[e] = d.overlap[e];
    const int64_t b_begin = first == 0 ? 0 : static_cast<int64_t>(first) - 1;
    __syncthreads();

    for (int64_t b = b_begin; b < static_cast<int64_t>(last); ++b) {
        const bool emit = b >= static_cast<int64_t>(first);
        const float4* __restrict__ xin4 = reinterpret_cast<const float4*>(d.in + static_cast<size_t>(b) * FI * 2);
        float4* __restrict__ xout4 = reinterpret_cast<float4*>(d.out + static_cast<size_t>(b) * FO * 2);
        // resampler_fft.rs:387-388: FI reals + FI zeros per channel, viewed as FI complexes; complex i =
        // frames 2i, 2i+1 (i < FI / 2), zero beyond.  One float4 = both frames of both channels.
        constexpr int TI = (FI + kFftThreads - 1) / kFftThreads;
#pragma unroll
        for (int it = 0; it < TI; ++it) {
            const int i = static_cast<int>(threadIdx.x) + it * kFftThreads;
            if (i < FI) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < FI / 2) v = xin4[i];
                A0[i] = make_float2(v.x, v.z);
                A1[i] = make_float2(v.y, v.w);
            }
        }
        __syncthreads();
        const bool in_b = Stages2Of<FWD>::type::run(A0, B0, A1, B1, plan.tw_f);
        float2* X0 = in_b ? B0 : A0;
        float2* X1 = in_b ? B1 : A1;
        float2* Y0 = in_b ? A0 : B0;
        float2* Y1 = in_b ? A1 : B1;
        postprocess_forward_ct<FI>(X0, plan.rc_f);
        postprocess_forward_ct<FI>(X1, plan.rc_f);
        __syncthreads();
        constexpr int TK = (FO + 1 + kFftThreads - 1) / kFftThreads;
#pragma unroll
        for (int it = 0; it < TK; ++it) {
            const uint32_t k = threadIdx.x + it * kFftThreads;
            if (k <= FO) {
                const bool on = k < plan.new_length;
                const float2 f = on ? plan.filter[k] : make_float2(0.f, 0.f);
                Y0[k] = on ? cmul(X0[k], f) : make_float2(0.f, 0.f);
                Y1[k] = on ? cmul(X1[k], f) : make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
        preprocess_inverse_head_ct<FO>(Y0, plan.rc_i);
        preprocess_inverse_head_ct<FO>(Y1, plan.rc_i);
        __syncthreads();
        conj_ct<FO>(Y0);
        conj_ct<FO>(Y1);
        __syncthreads();
        const bool z_in_x = Stages2Of<INV>::type::run(Y0, X0, Y1, X1, plan.tw_i);
        const float2* Z0 = z_in_x ? X0 : Y0;
        const float2* Z1 = z_in_x ? X1 : Y1;
        // output conjugation (radix_fft.rs:656-669), reals 2i, 2i+1 <- Z[i]; the first FO reals of a
        // channel are overlap-added and stored, the second FO become its next overlap (:416-423).
        float2* ov0 = reinterpret_cast<float2*>(carry);
        float2* ov1 = reinterpret_cast<float2*>(carry + FO);
        constexpr int TO = (FO / 2 + kFftThreads - 1) / kFftThreads;
#pragma unroll
        for (int it = 0; it < TO; ++it) {
            const uint32_t i = threadIdx.x + it * kFftThreads;
            if (i < FO / 2) {
                const float2 z0 = Z0[i], z1 = Z1[i], n0 = Z0[i + FO / 2], n1 = Z1[i + FO / 2];
                const float2 o0 = ov0[i], o1 = ov1[i];
                if (emit) xout4[i] = make_float4(z0.x + o0.x, z1.x + o1.x, -z0.y + o0.y, -z1.y + o1.y);
                ov0[i] = make_float2(n0.x, -n0.y);
                ov1[i] = make_float2(n1.x, -n1.y);
            }
        }
        __syncthreads();
    }
    if (last == d.n_blocks)
        for (uint32_t e = threadIdx.x; e < 2 * FO; e += kFftThreads) d.overlap_next[e] = carry[e];
}

typedef PlanCt<1176, 3, 7, 7, 8> Plan1176;
typedef PlanCt<1280, 4, 5, 8, 8> Plan1280;

__global__ __launch_bounds__(kFftThreads) void fft_filter_kernel(FftPlanDev plan,
                                                                 const float* __restrict__ filter_time,
                                                                 float2* __restrict__ spectrum) {
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    float2* bufA = lds2;
    float2* bufB = lds2 + plan.lds_complex;
    const uint32_t fi = plan.fft_in;
    for (uint32_t i = threadIdx.x; i < fi; i += kFftThreads)
        bufA[i] = make_float2(filter_time[2 * i], filter_time[2 * i + 1]);
    __syncthreads();
    float2* X = stockham(bufA, bufB, fi, plan.n_stages_f, plan.radix_f, plan.tw_off_f, plan.tw_f);
    postprocess_forward(X, fi, plan.rc_f, plan.n_rc_f);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k <= fi; k += kFftThreads) spectrum[k] = X[k];
}


// ---- plans whose two transform buffers do not fit the LDS (blocks of 7056 .. 12288 frames: the rate pairs with
// 384 kHz, or 176.4 kHz against 16 / 32 kHz) -------------------------------------------------------------------
// ONE buffer: a Stockham stage reads all of its butterflies into registers, a barrier, then writes the results
// back in place (1024 threads: at most 6 butterflies of 16 values per thread).  The overlap rows stay in HBM
// (the stream's next-state rows serve as the carry: one workgroup walks a whole stream).  Same operations in
// the same order as the kernels above.
constexpr int kBigThreads = 1024;
constexpr uint32_t kBigMaxN = 12288;

template <int R>
__device__ __forceinline__ void stage_inplace(float2* buf, uint32_t n, uint32_t stride, const float2* __restrict__ tw) {
    constexpr int KMAX = (kBigMaxN / R + kBigThreads - 1) / kBigThreads;
    const uint32_t m = n / R;
    float2 t[KMAX][R];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const uint32_t i = threadIdx.x + k * kBigThreads;
        if (i < m) {
#pragma unroll
            for (int q = 0; q < R; ++q) t[k][q] = buf[i + q * m];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const uint32_t i = threadIdx.x + k * kBigThreads;
        if (i < m) {
            const uint32_t kk = stride == 1 ? 0u : i % stride;
            float2 o[R];
            if (stride != 1) {
                const float2* w = tw + kk * (R - 1);
#pragma unroll
                for (int q = 1; q < R; ++q) t[k][q] = cmul(w[q - 1], t[k][q]);
            }
            dft<R>(t[k], o);
            float2* d = buf + R * i - (R - 1) * kk;
#pragma unroll
            for (int q = 0; q < R; ++q) d[q * stride] = o[q];
        }
    }
    __syncthreads();
}

__device__ void stockham_inplace(float2* buf, uint32_t n, uint32_t n_stages, const uint32_t* radix,
                                 const uint32_t* tw_off, const float2* __restrict__ tw) {
    uint32_t stride = 1;
    for (uint32_t s = 0; s < n_stages; ++s) {
        const uint32_t r = radix[s];
        const float2* w = tw + tw_off[s];
        switch (r) {
            case 2: stage_inplace<2>(buf, n, stride, w); break;
            case 3: stage_inplace<3>(buf, n, stride, w); break;
            case 4: stage_inplace<4>(buf, n, stride, w); break;
            case 5: stage_inplace<5>(buf, n, stride, w); break;
            case 7: stage_inplace<7>(buf, n, stride, w); break;
            default: stage_inplace<8>(buf, n, stride, w); break;
        }
        stride *= r;
    }
}

__global__ __launch_bounds__(kBigThreads) void fft_ola_big_kernel(FftPlanDev plan, const FftStreamDesc* __restrict__ descs,
                                                                  uint32_t run) {
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    const FftStreamDesc d = descs[blockIdx.y];
    const uint32_t first = blockIdx.x * run;
    if (first >= d.n_blocks) return;
    const uint32_t last = first + run < d.n_blocks ? first + run : d.n_blocks;  // exclusive
    const uint32_t C = d.channels, c = blockIdx.z, fi = plan.fft_in, fo = plan.fft_out;
    if (c >= C) return;
    float2* buf = lds2;
    float* carry = reinterpret_cast<float*>(lds2 + plan.lds_complex);   // [fo]: the channel's overlap
    // overlap carried into the run: the stream state, or the predecessor block recomputed (not emitted)
    if (first == 0)
        for (uint32_t e = threadIdx.x; e < fo; e += kBigThreads) carry[e] = d.overlap[c * fo + e];
    const int64_t b_begin = first == 0 ? 0 : static_cast<int64_t>(first) - 1;
    __syncthreads();
    for (int64_t b = b_begin; b < static_cast<int64_t>(last); ++b) {
        const bool emit = b >= static_cast<int64_t>(first);
        const float* __restrict__ xin = d.in + static_cast<size_t>(b) * fi * C;
        float* __restrict__ xout = d.out + static_cast<size_t>(b) * fo * C;
        // resampler_fft.rs:387-388: fi reals + fi zeros, viewed as fi complexes (radix_fft.rs:552-554)
        for (uint32_t i = threadIdx.x; i < fi; i += kBigThreads) {
            float2 v = make_float2(0.f, 0.f);
            if (2 * i + 1 < fi) v = make_float2(xin[(2 * i) * C + c], xin[(2 * i + 1) * C + c]);
            else if (2 * i < fi) v = make_float2(xin[(2 * i) * C + c], 0.f);
            buf[i] = v;
        }
        __syncthreads();
        stockham_inplace(buf, fi, plan.n_stages_f, plan.radix_f, plan.tw_off_f, plan.tw_f);
        postprocess_forward<kBigThreads>(buf, fi, plan.rc_f, plan.n_rc_f);
        __syncthreads();
        // resampler_fft.rs:401-408: multiply new_length bins, zero the rest up to fo
        for (uint32_t k = threadIdx.x; k <= fo; k += kBigThreads)
            buf[k] = k < plan.new_length ? cmul(buf[k], plan.filter[k]) : make_float2(0.f, 0.f);
        __syncthreads();
        preprocess_inverse<kBigThreads>(buf, fo, plan.rc_i, plan.n_rc_i);
        __syncthreads();
        stockham_inplace(buf, fo, plan.n_stages_i, plan.radix_i, plan.tw_off_i, plan.tw_i);
        // output conjugation (radix_fft.rs:656-669), reals 2i, 2i+1 <- Z[i]; overlap-add (:416-423)
        for (uint32_t t = threadIdx.x; t < fo; t += kBigThreads) {
            const float2 z = buf[t >> 1];
            const float y = (t & 1u) ? -z.y : z.x;
            if (emit) xout[t * C + c] = y + carry[t];
            const float2 z2 = buf[(t + fo) >> 1];
            carry[t] = ((t + fo) & 1u) ? -z2.y : z2.x;
        }
        __syncthreads();
    }
    if (last == d.n_blocks)
        for (uint32_t e = threadIdx.x; e < fo; e += kBigThreads) d.overlap_next[c * fo + e] = carry[e];
}


__global__ __launch_bounds__(kBigThreads) void fft_filter_big_kernel(FftPlanDev plan, const float* __restrict__ filter_time,
                                                                     float2* __restrict__ spectrum) {
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    float2* buf = lds2;
    const uint32_t fi = plan.fft_in;
    for (uint32_t i = threadIdx.x; i < fi; i += kBigThreads) buf[i] = make_float2(filter_time[2 * i], filter_time[2 * i + 1]);
    __syncthreads();
    stockham_inplace(buf, fi, plan.n_stages_f, plan.radix_f, plan.tw_off_f, plan.tw_f);
    postprocess_forward<kBigThreads>(buf, fi, plan.rc_f, plan.n_rc_f);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k <= fi; k += kBigThreads) spectrum[k] = buf[k];
}

}  // namespace

// one buffer (the stages run in place) + the overlap row of the workgroup's channel
size_t fft_big_lds_bytes(const FftPlanDev& plan) {
    return static_cast<size_t>(plan.lds_complex) * sizeof(float2) + static_cast<size_t>(plan.fft_out) * sizeof(float);
}

size_t fft_ola_lds_bytes(const FftPlanDev& plan, uint32_t channels) {
    return 2 * static_cast<size_t>(plan.lds_complex) * sizeof(float2) +
           static_cast<size_t>(channels) * plan.fft_out * sizeof(float);
}

hipError_t launch_fft_ola(const FftPlanDev& plan, const FftStreamDesc* d_descs, uint32_t n_streams,
                          uint32_t max_blocks, uint32_t max_channels, uint32_t min_channels,
                          hipStream_t stream, uint32_t pcm_bits) {
    if (n_streams == 0 || max_blocks == 0) return hipSuccess;
    static const bool no_wave = rsmp::knob("RSMP_FFT_WAVE") != nullptr && atoi(rsmp::knob("RSMP_FFT_WAVE")) == 0;   // A/B
    // (PCM input is read by the two-channel wave kernel only: FftStreamDesc::in_bits)
    if (pcm_bits != 0 && (no_wave || max_channels != 2 || min_channels != 2)) return hipErrorNotSupported;
    // two-channel f32 streams: a wave per stream, the frame as one complex sample (not in the exact build)
    static const bool no_pair = rsmp::knob("RSMP_FFT_PAIR") != nullptr && atoi(rsmp::knob("RSMP_FFT_PAIR")) == 0;   // A/B
    // (a launch of a block or two per stream is a streaming call: there the wave-per-channel kernel's two waves per stream
    // finish sooner than one wave running both chains -- 32.6 against 36.3 us per one-block call, tools/fft_call_latency.py)
    if (!no_wave && !no_pair && max_channels == 2 && min_channels == 2 && max_blocks >= 4 && !fft_wave_is_exact()) {
        const hipError_t e = launch_fft_ola_pair(plan, d_descs, n_streams, max_blocks, stream, pcm_bits);
        if (e != hipErrorNotSupported) return e;
    }
    if (!no_wave) {
        const hipError_t e = launch_fft_ola_wave(plan, d_descs, n_streams, max_blocks, max_channels, min_channels, stream);
        if (e != hipErrorNotSupported || pcm_bits != 0) return e;
    }
    // Blocks per workgroup: every run after a stream's first recomputes its predecessor block (1 / run
    // extra work), and the launch ends with a partly filled round of workgroups unless their number
    // is close to a multiple of what the chip holds at once.  Pick the run length (8..64) that
    // maximises useful work per occupied slot.
    auto pick_run = [&](const void* fn, uint32_t threads, size_t lds_bytes, uint32_t grid_z) {
        uint32_t run = kFftRun;
        int dev = 0, cus = 256, per_cu = 4;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 4;
        const double slots = static_cast<double>(cus) * per_cu;
        double best = -1.0;
        for (uint32_t cand = 8; cand <= 64; ++cand) {
            const double runs = static_cast<double>((max_blocks + cand - 1) / cand);
            const double wgs = runs * n_streams * grid_z;
            const double rounds = std::ceil(wgs / slots);
            const double useful = static_cast<double>(max_blocks) / (max_blocks + runs - 1.0);   // halo blocks
            const double score = wgs / (rounds * slots) * useful;
            if (score > best + 1e-9) { best = score; run = cand; }
        }
        return run;
    };
    size_t lds = fft_ola_lds_bytes(plan, max_channels);
    constexpr int big_knob = 160 * 1024;
    if (lds > static_cast<size_t>(big_knob)) {   // the two-buffer kernels do not fit: one buffer, in place, a workgroup per channel
        const size_t big = fft_big_lds_bytes(plan);
        if (big > 160 * 1024) return hipErrorInvalidValue;
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_ola_big_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        const uint32_t run = pick_run(reinterpret_cast<const void*>(fft_ola_big_kernel), kBigThreads, big, max_channels);
        hipLaunchKernelGGL(fft_ola_big_kernel, dim3((max_blocks + run - 1) / run, n_streams, max_channels), dim3(kBigThreads), big,
                           stream, plan, d_descs, run);
        return hipGetLastError();
    }
    bool all_stereo = max_channels == 2 && min_channels == 2;
    static const bool generic_only = rsmp::knob("RSMP_FFT_GENERIC") != nullptr;   // A/B: skip the specialised builds
    const bool rc_full = plan.n_rc_f == plan.fft_in / 2 - 1 && plan.n_rc_i == plan.fft_out / 2 - 1;
    typedef void (*Kernel)(FftPlanDev, const FftStreamDesc*, uint32_t);
    Kernel fn = fft_ola_kernel<kFftThreads, false>;
    uint32_t threads = kFftThreads, grid_z = 1;
    constexpr bool no_stereo = false;
    const bool stereo = all_stereo && !no_stereo;
    if (!generic_only && rc_full && Plan1176::matches(plan.fft_in, plan.n_stages_f, plan.radix_f) &&
        Plan1280::matches(plan.fft_out, plan.n_stages_i, plan.radix_i))
        fn = stereo ? fft_ola_kernel_ct2<Plan1176, Plan1280> : fft_ola_kernel_ct<Plan1176, Plan1280>;
    else if (!generic_only && rc_full && Plan1280::matches(plan.fft_in, plan.n_stages_f, plan.radix_f) &&
             Plan1176::matches(plan.fft_out, plan.n_stages_i, plan.radix_i))
        fn = stereo ? fft_ola_kernel_ct2<Plan1280, Plan1176> : fft_ola_kernel_ct<Plan1280, Plan1176>;
    else {
        all_stereo = false;
        // the generic pipeline: for blocks up to 512 frames (both sides) a one-wave workgroup per channel (see the
        // kernel; 96 -> 48 kHz 1.56 -> 1.18 ms, 192 -> 48 kHz 1.48 -> 0.76 ms per 64 x 2^20 frames); above that the
        // four-wave workgroups keep more waves on a CU for the same LDS and win (tools/fft_pairs_bench.py)
        constexpr int wave_knob = -1;
        const size_t lds_wave = 2 * static_cast<size_t>(plan.lds_complex) * sizeof(float2) + static_cast<size_t>(plan.fft_out) * sizeof(float);
        const bool per_channel = wave_knob >= 0 ? wave_knob != 0 : plan.lds_complex <= 513;
        constexpr int threads_knob = 0;
        if (per_channel && lds_wave <= 160 * 1024) {
            fn = fft_ola_kernel<64, true>;
            threads = 64;
            grid_z = max_channels;
            lds = lds_wave;
        } else if (threads_knob ? threads_knob == 1024 : lds > 80 * 1024) {
            // the long plans: one workgroup per CU is all the LDS holds, so it is 16 waves wide, not 4
            fn = fft_ola_kernel<1024, false>;
            threads = 1024;
        } else if (threads_knob ? threads_knob == 512 : lds > 160 * 1024 / 3) {
            fn = fft_ola_kernel<512, false>;   // two workgroups per CU
            threads = 512;
        }
    }
    if (fn == static_cast<Kernel>(fft_ola_kernel_ct2<Plan1176, Plan1280>) ||
        fn == static_cast<Kernel>(fft_ola_kernel_ct2<Plan1280, Plan1176>))
        lds = 4 * static_cast<size_t>(plan.lds_complex) * sizeof(float2) + 2 * static_cast<size_t>(plan.fft_out) * sizeof(float);
    if (lds > 64 * 1024) {   // dynamic LDS above 64 KiB must be opted into
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    const uint32_t run = pick_run(reinterpret_cast<const void*>(fn), threads, lds, grid_z);
    const dim3 grid((max_blocks + run - 1) / run, n_streams, grid_z);
    hipLaunchKernelGGL(fn, grid, dim3(threads), lds, stream, plan, d_descs, run);
    return hipGetLastError();
}

hipError_t launch_fft_filter_spectrum(const FftPlanDev& plan, const float* d_filter_time,
                                      float2* d_filter_spectrum, hipStream_t stream) {
    const size_t lds = 2 * static_cast<size_t>(plan.lds_complex) * sizeof(float2);
    if (lds > 160 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_filter_big_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fft_filter_big_kernel, dim3(1), dim3(kBigThreads), fft_big_lds_bytes(plan), stream, plan,
                           d_filter_time, d_filter_spectrum);
        return hipGetLastError();
    }
    if (lds > 64 * 1024) {   // dynamic LDS above 64 KiB must be opted into
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fft_filter_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fft_filter_kernel, dim3(1), dim3(kFftThreads), lds, stream, plan,
                       d_filter_time, d_filter_spectrum);
    return hipGetLastError();
}

}  // namespace rsmp
