// fir_split.hip -- periodic FIR on the bf16 matrix cores with three-way split operands (gfx950).
//
// Replaces the same reference code as the other periodic kernels (src/resampler_fir.rs:542-590 +
// src/fir/avx.rs:5-61) for two-channel streams whose rate pair has 16..160 classes (44.1 <-> 48 kHz).
//
// Why: 128 taps per output value is 16.6 FMA per byte of HBM traffic, more than the f32 pipes can
// retire per byte (78 TFMA/s vs 8 TB/s): an exact-f32 kernel cannot get past ~50 % of the HBM roofline.
// The bf16 matrix pipe is 16x faster per product, and an f32 value is EXACTLY the sum of three bf16
// values (8 + 8 + 8 significant bits, by truncation).  With x = x1 + x2 + x3 and c = c1 + c2 + c3 the
// six products c1x1, c2x1, c3x1, c1x2, c2x2, c1x3 (each exact in f32, accumulated in f32 by the MFMA)
// leave out only terms below 2^-24 of a product: the result is as close to the f64 sum as the
// reference's own f32 FMA chain is (measured: DESIGN.md section 4.1).
//
// Layout of the computation (same classes / tiles / shifted zero-padded windows as fir_periodic.h):
//   D[class 16][period 16] += A[class][k 32] * B[k][period]      v_mfma_f32_16x16x32_bf16
//   * one workgroup per CU, 12 waves: 2 producers + 10 consumers; consumer T owns class tile T for the
//     whole launch and keeps its coefficient tile -- 3 planes x window/32 steps x 4 registers -- in
//     VGPRs: the class table is read once per workgroup, not once per work unit;
//   * an LDS image holds 16 periods of both channels as three bf16 planes, TRANSPOSED: row = frame
//     inside the period (0 .. first frame of the last tile's window + window), 16 periods side by
//     side (32 bytes).  Any window start is then a row address (no alignment constraint), and
//     ds_read_b64_tr_b16 delivers the B operand -- 4 consecutive frames x 16 periods per 16 lanes --
//     at the full 256 B/clk.  Rows beyond the period repeat the next period's first frames;
//   * producers load frames from HBM (coalesced along the frame index), split them into the three
//     planes with 4 VALU operations per value, pair two periods into a dword and write the planes
//     with ds_write_b32 (8-byte chunks XOR-swizzled by the row so that the writes spread over the banks);
//   * two images (ring), monotonic LDS counters instead of barriers; the wrap variant of class 0
//     (row 1023 on the previous frame, :562-564) is computed by a producer in f32 from global memory.
#include "fir_periodic.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "common.h"

namespace rsmp {

namespace {

constexpr uint32_t kProducers = 2, kConsumers = 10, kWaves = kProducers + kConsumers;
constexpr uint32_t kCtrlBytes = 256;                 // staged[2], done[2], wflag[2]
constexpr uint32_t kWrapBytes = 2 * 16 * 16;         // two slots x 16 periods x (ch0, ch1, take, -)
constexpr uint32_t kImageBase = kCtrlBytes + kWrapBytes;
constexpr uint32_t kLdsLimit = 160 * 1024;
constexpr int kBatch = 8;                            // (row block, period pair) combos in flight per producer

struct SplitArgs {
    uint32_t a, b, taps, n_tiles, rows, blocks_per_stream, total_items, debug;
};

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const float __attribute__((address_space(1)))* gconst_f32_ptr;
typedef const v2f __attribute__((address_space(1)))* gconst_f2_ptr;
typedef const v4u __attribute__((address_space(1)))* gconst_u4_ptr;
typedef const uint32_t __attribute__((address_space(1)))* gconst_u32_ptr;
typedef float __attribute__((address_space(1)))* g_f32_ptr;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ uint32_t lds_load_acquire(uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store_release(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// One count per WAVE (a wave-wide atomic would add one per active lane): everything the wave did in
// LDS before is complete when the count becomes visible.
__device__ __forceinline__ void lds_signal(uint32_t* p) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0)
        (void)__hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The wave-uniform part of a stream descriptor that the kernel needs, held in scalar registers.
struct StreamCtx {
    const float* in;
    const float* hist;
    float* out;
    const float* coeffs;
    const float* class_coef;
    const uint32_t* wrap_bits;
    uint32_t n_out, hist_frames, in_frames;
    uint64_t abs_out, abs_consumed, wrap_k0;
};

__device__ __forceinline__ StreamCtx load_stream(const FirStreamDesc* descs, uint32_t s) {
    const FirStreamDesc& d = descs[s];
    StreamCtx c;
    c.in = d.in;
    c.hist = d.hist;
    c.out = d.out;
    c.coeffs = d.coeffs;
    c.class_coef = d.class_coef;
    c.wrap_bits = d.wrap_bits;
    c.n_out = d.n_out;
    c.hist_frames = d.hist_frames;
    c.in_frames = d.in_frames;
    c.abs_out = d.abs_out;
    c.abs_consumed = d.abs_consumed;
    c.wrap_k0 = d.wrap_k0;
    return c;
}

struct Item {
    uint64_t q0;          // first period of the image
    int32_t n_block0;     // launch-relative output index of (period q0, class 0)
    int32_t k_block0;     // wrap-bitmap index of (period q0, class 0)
    bool valid;
};

__device__ __forceinline__ Item item_of(const SplitArgs& g, const StreamCtx& d, uint32_t block) {
    Item it;
    it.q0 = d.abs_out / g.b + static_cast<uint64_t>(block) * 16u;
    it.valid = d.n_out != 0 && it.q0 * g.b < d.abs_out + d.n_out;
    it.n_block0 = static_cast<int32_t>(static_cast<int64_t>(it.q0 * g.b) - static_cast<int64_t>(d.abs_out));
    it.k_block0 = static_cast<int32_t>(static_cast<int64_t>(it.q0) - static_cast<int64_t>(d.wrap_k0));
    return it;
}

// f32 -> three bf16 planes by truncation: x == p1 + p2 + p3 exactly (24 significant bits = 8 + 8 + 8;
// both subtractions are exact).  Returned as f32 bit patterns whose low halves are don't-care.
__device__ __forceinline__ void split3(float x, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = __float_as_uint(x);
    const float r1 = x - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(p2 & 0xFFFF0000u);
    p3 = __float_as_uint(r2);
}
// (high half of hi) : (high half of lo)
__device__ __forceinline__ uint32_t pack_hi16(uint32_t hi, uint32_t lo) {
    return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}

struct Combo {
    v2f x0, x1;   // (ch0, ch1) of row k in periods 2pp and 2pp + 1
};

template <int NK>
__global__ __launch_bounds__(768) void fir_split_kernel(const FirStreamDesc* __restrict__ descs,
                                                        const SplitArgs g) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint32_t* ctrl = reinterpret_cast<uint32_t*>(lds);
    uint32_t* staged = ctrl;        // [slot]: producers that finished staging, cumulative
    uint32_t* done = ctrl + 2;      // [slot]: consumers that finished reading, cumulative
    uint32_t* wflag = ctrl + 4;     // [slot]: round + 1 whose wrap results are in the wrap area
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < kCtrlBytes / 4) ctrl[threadIdx.x] = 0;
    __syncthreads();

    const uint32_t R = g.rows;
    const uint32_t image_bytes = 6u * R * 32u;
    const uint32_t n_active = g.n_tiles < kConsumers ? g.n_tiles : kConsumers;
    const uint32_t item_begin = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x) * g.total_items / gridDim.x);
    const uint32_t item_end = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x + 1) * g.total_items / gridDim.x);
    if (item_begin == item_end) return;

    uint32_t cur_stream = 0xFFFFFFFFu;
    StreamCtx d{};
    uint32_t rnd = 0;

    if (wave < kProducers) {
        // ---- producer ------------------------------------------------------------------------------
        const uint32_t n_rb = (R + 63) / 64;                       // 64-row blocks
        const uint32_t n_combos = n_rb * 8;                        // x 8 period pairs
        const uint32_t mine = (n_combos - wave + kProducers - 1) / kProducers;   // combos wave, wave + P, ...
        for (uint32_t item = item_begin; item < item_end; ++item) {
            const uint32_t stream = item / g.blocks_per_stream;
            if (stream != cur_stream) {
                d = load_stream(descs, stream);
                cur_stream = stream;
            }
            const Item it = item_of(g, d, item - stream * g.blocks_per_stream);
            if (!it.valid) continue;
            const uint32_t slot = rnd & 1, use = rnd >> 1;
            char* img = lds + kImageBase + slot * image_bytes;
            const int64_t f0 = static_cast<int64_t>(it.q0 * g.a) - static_cast<int64_t>(d.abs_consumed);
            const int64_t hist_frames = d.hist_frames;
            const int64_t total = hist_frames + static_cast<int64_t>(d.in_frames);
            const bool interior = f0 >= hist_frames && f0 + static_cast<int64_t>(15u * g.a + R) <= total;
            gconst_f2_ptr in2 = (gconst_f2_ptr)d.in;
            gconst_f2_ptr hist2 = (gconst_f2_ptr)d.hist;

            auto store_combo = [&](uint32_t c, const Combo& cb) {
                const uint32_t rb = c >> 3, pp = c & 7;
                const uint32_t k = rb * 64 + lane;
                if (k >= R) return;
                uint32_t a1, a2, a3, b1, b2, b3;
                char* row = img + k * 32 + ((((pp >> 1) ^ ((k >> 2) & 3)) << 3) | ((pp & 1) << 2));
                // channel 0
                split3(cb.x0.x, a1, a2, a3);
                split3(cb.x1.x, b1, b2, b3);
                *reinterpret_cast<uint32_t*>(row) = pack_hi16(b1, a1);
                *reinterpret_cast<uint32_t*>(row + R * 32) = pack_hi16(b2, a2);
                *reinterpret_cast<uint32_t*>(row + 2 * R * 32) = pack_hi16(b3, a3);
                // channel 1
                split3(cb.x0.y, a1, a2, a3);
                split3(cb.x1.y, b1, b2, b3);
                *reinterpret_cast<uint32_t*>(row + 3 * R * 32) = pack_hi16(b1, a1);
                *reinterpret_cast<uint32_t*>(row + 4 * R * 32) = pack_hi16(b2, a2);
                *reinterpret_cast<uint32_t*>(row + 5 * R * 32) = pack_hi16(b3, a3);
            };
            if (interior) {
                // the whole image lies inside `in`: batches of loads in flight, the first two before the
                // slot is known to be free
                gconst_f2_ptr src = in2 + (f0 - hist_frames) + (lane < R ? lane : 0);
                auto load_combo = [&](uint32_t c) -> Combo {
                    const uint32_t rb = c >> 3, pp = c & 7;
                    uint32_t k = rb * 64;
                    if (k + lane >= R) k = 0;   // rows past the image: any valid address (not stored)
                    gconst_f2_ptr s0 = src + ((2 * pp) * g.a + k);
                    Combo cb;
                    cb.x0 = s0[0];
                    cb.x1 = s0[g.a];
                    return cb;
                };
                Combo xa[kBatch], xb[kBatch];
                const uint32_t n_batches = (mine + kBatch - 1) / kBatch;
                auto load_batch = [&](Combo (&x)[kBatch], uint32_t bt) {
#pragma unroll
                    for (int i = 0; i < kBatch; ++i) {
                        const uint32_t m = bt * kBatch + i;
                        if (m < mine) x[i] = load_combo(wave + m * kProducers);
                    }
                };
                auto store_batch = [&](const Combo (&x)[kBatch], uint32_t bt) {
#pragma unroll
                    for (int i = 0; i < kBatch; ++i) {
                        const uint32_t m = bt * kBatch + i;
                        if (m < mine) store_combo(wave + m * kProducers, x[i]);
                    }
                };
                load_batch(xa, 0);
                if (n_batches > 1) load_batch(xb, 1);
                while (lds_load_acquire(done + slot) < n_active * use) __builtin_amdgcn_s_sleep(1);
                for (uint32_t bt = 0; bt < n_batches; bt += 2) {
                    store_batch(xa, bt);
                    if (bt + 2 < n_batches) load_batch(xa, bt + 2);
                    if (bt + 1 < n_batches) store_batch(xb, bt + 1);
                    if (bt + 3 < n_batches) load_batch(xb, bt + 3);
                }
            } else {
                // stream edges: frames outside [hist|in] read as zero; one combo at a time
                while (lds_load_acquire(done + slot) < n_active * use) __builtin_amdgcn_s_sleep(1);
                for (uint32_t m = 0; m < mine; ++m) {
                    const uint32_t c = wave + m * kProducers;
                    const uint32_t rb = c >> 3, pp = c & 7;
                    uint32_t k = rb * 64 + lane;
                    if (k >= R) k = R - 1;
                    auto fetch = [&](int64_t f) -> v2f {
                        const bool ok = f >= 0 && f < total;
                        const int64_t fc = f < 0 ? 0 : (f >= total ? total - 1 : f);
                        v2f v = fc < hist_frames ? hist2[fc] : in2[fc - hist_frames];
                        if (!ok) v = v2f{0.f, 0.f};
                        return v;
                    };
                    const int64_t f = f0 + static_cast<int64_t>((2 * pp) * g.a + k);
                    Combo cb;
                    cb.x0 = fetch(f);
                    cb.x1 = fetch(f + g.a);
                    store_combo(c, cb);
                }
            }
            lds_signal(staged + slot);

            // wrap variant of class 0 for the 16 periods: row 1023 on the window one frame earlier
            // (resampler_fir.rs:544, :562-565), f32 FMA from global memory; lane = (quarter of the taps, period)
            if (wave == kProducers - 1) {
                const uint32_t p = lane & 15, part = lane >> 4;
                const int32_t nw = it.n_block0 + static_cast<int32_t>(p * g.b);
                uint32_t take = 0;
                if (nw >= 0 && nw < static_cast<int32_t>(d.n_out)) {
                    const uint32_t K = static_cast<uint32_t>(it.k_block0 + static_cast<int32_t>(p));
                    take = (((gconst_u32_ptr)d.wrap_bits)[K >> 5] >> (K & 31)) & 1u;
                }
                float* wv = reinterpret_cast<float*>(lds + kCtrlBytes + slot * (kWrapBytes / 2));
                v2f acc = v2f{0.f, 0.f};
                if (__any(take != 0) && !(g.debug & 1024)) {
                    gconst_f32_ptr wrow = (gconst_f32_ptr)d.coeffs + static_cast<size_t>(1023) * g.taps;
                    const uint32_t per = (g.taps + 3) / 4;
                    const uint32_t t0 = part * per, t1 = t0 + per < g.taps ? t0 + per : g.taps;
                    const int64_t fw = f0 + static_cast<int64_t>(p * g.a) - 1;
                    for (uint32_t t = t0; t < t1; ++t) {
                        const int64_t f = fw + t;
                        const bool ok = f >= 0 && f < total;
                        const int64_t fc = f < 0 ? 0 : (f >= total ? total - 1 : f);
                        v2f v = fc < hist_frames ? hist2[fc] : in2[fc - hist_frames];
                        if (!ok) v = v2f{0.f, 0.f};
                        const float w = wrow[t];
                        acc.x = fmaf(w, v.x, acc.x);
                        acc.y = fmaf(w, v.y, acc.y);
                    }
                    acc.x += __shfl_xor(acc.x, 16, 64);
                    acc.y += __shfl_xor(acc.y, 16, 64);
                    acc.x += __shfl_xor(acc.x, 32, 64);
                    acc.y += __shfl_xor(acc.y, 32, 64);
                }
                if (lane < 16) *reinterpret_cast<v4f*>(wv + lane * 4) = v4f{acc.x, acc.y, __uint_as_float(take), 0.f};
                lds_store_release(wflag + slot, rnd + 1);
            }
            ++rnd;
        }
        return;
    }

    // ---- consumer ----------------------------------------------------------------------------------
    const uint32_t T = wave - kProducers;
    if (T >= g.n_tiles) return;
    const uint32_t grp = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
    const uint32_t ob = (T * 16u * g.a) / g.b;        // first frame of the tile's window
    const uint32_t row0 = ob + 4 * grp + q;
    const uint32_t lane_off = row0 * 32 + ((pc ^ ((row0 >> 2) & 3)) << 3);
    const uint32_t j0 = T * 16u + 4 * grp;             // the lane's four classes (D rows)
    const uint32_t pl = lane & 15;                     // the lane's period (D column)

    bf16x8 A[NK][3];
    const float* cur_table = nullptr;

    for (uint32_t item = item_begin; item < item_end; ++item) {
        const uint32_t stream = item / g.blocks_per_stream;
        if (stream != cur_stream) {
            d = load_stream(descs, stream);
            cur_stream = stream;
        }
        const Item it = item_of(g, d, item - stream * g.blocks_per_stream);
        if (!it.valid) continue;
        if (d.class_coef != cur_table) {   // streams of one launch may differ in drift
            cur_table = d.class_coef;
            gconst_u4_ptr tp = (gconst_u4_ptr)(cur_table) + static_cast<size_t>(T) * (NK * 3 * 64) + lane;
#pragma unroll
            for (int s = 0; s < NK; ++s)
#pragma unroll
                for (int p = 0; p < 3; ++p) A[s][p] = __builtin_bit_cast(bf16x8, tp[(s * 3 + p) * 64]);
        }
        const uint32_t slot = rnd & 1, use = rnd >> 1;
        const uint32_t base = kImageBase + slot * image_bytes + lane_off;
        while (lds_load_acquire(staged + slot) < kProducers * (use + 1)) __builtin_amdgcn_s_sleep(1);

        v4f acc0 = v4f{0.f, 0.f, 0.f, 0.f}, acc1 = v4f{0.f, 0.f, 0.f, 0.f};
        auto frag = [&](uint32_t plane_ch, int s) -> bf16x8 {
            const uint32_t addr = base + plane_ch * (R * 32u);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + addr + s * 1024));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + addr + s * 1024 + 512));
            const s16x8 t = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(bf16x8, t);
        };
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            const bf16x8 x1 = frag(0, s), x2 = frag(1, s), x3 = frag(2, s);
            const bf16x8 y1 = frag(3, s), y2 = frag(4, s), y3 = frag(5, s);
            // smallest products first
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][0], x3, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][0], y3, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][1], x2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][1], y2, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][2], x1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][2], y1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][0], x2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][0], y2, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][1], x1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][1], y1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][0], x1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][0], y1, acc1, 0, 0, 0);
        }
        // class 0 may take the wrap variant a producer computed (tile 0, D row 0)
        if (T == 0) {
            while (lds_load_acquire(wflag + slot) != rnd + 1) __builtin_amdgcn_s_sleep(1);
            const v4f w = *reinterpret_cast<const v4f*>(lds + kCtrlBytes + slot * (kWrapBytes / 2) + pl * 16);
            if (grp == 0 && __float_as_uint(w.z) != 0u && !(g.debug & 2048)) {
                acc0.x = w.x;
                acc1.x = w.y;
            }
        }
        lds_signal(done + slot);

        // lane = (period, 4 consecutive classes), both channels: 32 contiguous bytes
        const int32_t n0 = it.n_block0 + static_cast<int32_t>(pl * g.b + j0);
        const int32_t n_limit = static_cast<int32_t>(d.n_out);
        g_f32_ptr o = (g_f32_ptr)d.out + static_cast<int64_t>(n0) * 2;
        const v4f lo = v4f{acc0.x, acc1.x, acc0.y, acc1.y};
        const v4f hi = v4f{acc0.z, acc1.z, acc0.w, acc1.w};
        if (!(g.debug & 16)) {
            if (j0 + 4 <= g.b && n0 >= 0 && n0 + 4 <= n_limit) {
                typedef v4f __attribute__((address_space(1), aligned(8)))* g_f4a8_ptr;
                ((g_f4a8_ptr)o)[0] = lo;
                ((g_f4a8_ptr)o)[1] = hi;
            } else {
                const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int32_t n = n0 + r;
                    if (j0 + r < g.b && n >= 0 && n < n_limit) {
                        typedef v2f __attribute__((address_space(1)))* g_f2_ptr;
                        *((g_f2_ptr)(o + 2 * r)) = v2f{v[2 * r], v[2 * r + 1]};
                    }
                }
            }
        }
        ++rnd;
    }
}

inline uint32_t split_class_offset(uint32_t a, uint32_t b, uint32_t j) {
    return static_cast<uint32_t>((static_cast<uint64_t>(j) * a) / b);
}

}  // namespace

// Geometry of the split-bf16 kernel for num/den, or !ok: two channels, one true period per class
// pattern (b = den: 16..160 classes = at most one tile per consumer wave), window of <= 160 taps,
// two images within the LDS.
PeriodicGeometry split_geometry(uint64_t num, uint64_t den, uint32_t taps, uint32_t channels) {
    PeriodicGeometry g;
    if (channels != 2 || num == 0 || num > 4096 || den < 16 || den > 16 * kConsumers) return g;
    const uint32_t a = static_cast<uint32_t>(num), b = static_cast<uint32_t>(den);
    const uint32_t n_tiles = (b + 15) / 16;
    uint32_t shift = 0, ob_max = 0;
    for (uint32_t t = 0; t < n_tiles; ++t) {
        const uint32_t ob = split_class_offset(a, b, 16 * t);
        if (ob > ob_max) ob_max = ob;
        for (uint32_t i = 0; i < 16 && 16 * t + i < b; ++i) {
            const uint32_t s = split_class_offset(a, b, 16 * t + i) - ob;
            if (s > shift) shift = s;
        }
    }
    const uint32_t kpad = (taps + shift + 31) / 32 * 32;
    if (kpad / 32 < 1 || kpad / 32 > 5) return g;
    const uint32_t rows = ob_max + kpad;
    const uint32_t lds = kImageBase + 2u * 6u * rows * 32u;
    if (lds > kLdsLimit) return g;
    g.a = a;
    g.b = b;
    g.den = b;
    g.taps = taps;
    g.row_len = kpad;
    g.n_tiles = n_tiles;
    g.n_units = n_tiles;
    g.cg = 2;
    g.lp = 1;
    g.pw = 16;
    g.row_stride = rows;       // rows of an image (frames of a period + window reach)
    g.waves = kWaves;
    g.producers = kProducers;
    g.images = 2;
    g.mfma = 3;
    g.lds_bytes = lds;
    g.inline_wraps = true;
    g.ok = true;
    return g;
}

// Class-table image for the split kernel: [tile][k step][plane][lane][8 bf16]; lane (class m =
// lane & 15, k group = lane >> 4) element j holds window position 32 s + 16 (j >> 2) + 4 (lane >> 4) +
// (j & 3) -- the order in which the transposed LDS reads deliver the frames.
void split_store_class(std::vector<float>& coef, const PeriodicGeometry& g, uint32_t tile, uint32_t m,
                       uint32_t shift, const std::vector<float>& mixed) {
    const uint32_t nk = g.row_len / 32;
    uint32_t* words = reinterpret_cast<uint32_t*>(coef.data());
    for (uint32_t s = 0; s < nk; ++s)
        for (uint32_t grp = 0; grp < 4; ++grp)
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t pos = 32 * s + 16 * (j >> 2) + 4 * grp + (j & 3);
                float c = 0.f;
                if (pos >= shift && pos - shift < g.taps) c = mixed[pos - shift];
                uint32_t p[3];
                uint32_t u;
                std::memcpy(&u, &c, 4);
                p[0] = u >> 16;
                float h;
                uint32_t hu = u & 0xFFFF0000u;
                std::memcpy(&h, &hu, 4);
                const float r1 = c - h;
                std::memcpy(&u, &r1, 4);
                p[1] = u >> 16;
                hu = u & 0xFFFF0000u;
                std::memcpy(&h, &hu, 4);
                const float r2 = r1 - h;
                std::memcpy(&u, &r2, 4);
                p[2] = u >> 16;
                const uint32_t lane = 16 * grp + m;
                for (uint32_t pl = 0; pl < 3; ++pl) {
                    const size_t dword = ((((static_cast<size_t>(tile) * nk + s) * 3 + pl) * 64 + lane) * 4) + (j >> 1);
                    const uint32_t sh = (j & 1) * 16;
                    words[dword] = (words[dword] & ~(0xFFFFu << sh)) | (p[pl] << sh);
                }
            }
}

size_t split_table_floats(const PeriodicGeometry& g) {
    return static_cast<size_t>(g.n_tiles) * (g.row_len / 32) * 3 * 64 * 4;
}

hipError_t launch_fir_split(const FirStreamDesc* d_descs, uint32_t n_streams, const PeriodicGeometry& geo,
                            uint32_t max_blocks, uint32_t cus, hipStream_t stream) {
    static const uint32_t debug = [] {
        const char* e = getenv("RSMP_FIR_DEBUG");
        return e ? static_cast<uint32_t>(atoi(e)) : 0u;
    }();
    SplitArgs args{geo.a, geo.b, geo.taps, geo.n_tiles, geo.row_stride, max_blocks, max_blocks * n_streams, debug};
    const void* fns[5] = {reinterpret_cast<const void*>(fir_split_kernel<1>),
                          reinterpret_cast<const void*>(fir_split_kernel<2>),
                          reinterpret_cast<const void*>(fir_split_kernel<3>),
                          reinterpret_cast<const void*>(fir_split_kernel<4>),
                          reinterpret_cast<const void*>(fir_split_kernel<5>)};
    const uint32_t nk = geo.row_len / 32;
    if (nk < 1 || nk > 5) return hipErrorInvalidValue;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    static std::mutex mu;
    static std::map<std::pair<int, uint32_t>, bool> granted;
    {
        std::lock_guard<std::mutex> lock(mu);
        bool& have = granted[{device, nk}];
        if (!have) {
            e = hipFuncSetAttribute(fns[nk - 1], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
            if (e != hipSuccess) return e;
            have = true;
        }
    }
    const dim3 grid(args.total_items < cus ? args.total_items : cus);
    static const bool verbose = getenv("RSMP_FIR_VERBOSE") != nullptr;
    if (verbose)
        fprintf(stderr, "[rsmp] split launch: a=%u b=%u window=%u tiles=%u rows=%u lds=%u items=%u grid=%u\n",
                geo.a, geo.b, geo.row_len, geo.n_tiles, geo.row_stride, geo.lds_bytes, args.total_items, grid.x);
    void* kargs[2] = {&d_descs, &args};
    e = hipLaunchKernel(fns[nk - 1], grid, dim3(kWaves * 64), kargs, geo.lds_bytes, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace rsmp
