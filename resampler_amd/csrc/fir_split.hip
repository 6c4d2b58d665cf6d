// fir_split.hip -- periodic FIR on the 16-bit matrix cores with split f32 operands (gfx950).
//
// Replaces the same reference code as the other periodic kernels (src/resampler_fir.rs:542-590 +
// src/fir/avx.rs:5-61) for two-channel streams whose rate pair has 16..160 classes (44.1 <-> 48 kHz).
//
// Why: 128 taps per output value is 16.6 FMA per byte of HBM traffic, more than the f32 pipes can
// retire per byte (78 TFMA/s vs 8 TB/s): an exact-f32 kernel cannot get past ~50 % of the HBM roofline.
// The 16-bit matrix pipe is 16x faster per product, and an f32 operand can be cut into 16-bit planes:
//   PLANES = 2 (default): x * 2^12 = h1 + h2 + r with h1, h2 fp16 (round to nearest) and |r| <= 2^-22 |x|;
//     likewise c * 2^13.  Three products c1x1 + c2x1 + c1x2 (each exact in f32, accumulated in f32 by
//     v_mfma_f32_16x16x32_f16, which keeps fp16 denormals -- tools/f16_mfma_probe.hip) are closer to the
//     f64 sum than the reference's own f32 FMA chain at audio levels (6e-8 vs 1.4e-7 RMS at full scale,
//     tests/test_split_precision.py).  Samples of magnitude >= 16 overflow the scaled plane; the launch's
//     non-finite check (fir_nonfinite.h) then has the chunk recomputed in the reference's f32 form.
//   PLANES = 3 (RSMP_FIR_SPLIT_PLANES=3): x = p1 + p2 + p3 EXACTLY with bf16 planes by truncation (8 + 8 +
//     8 significant bits); six products c1x1, c2x1, c3x1, c1x2, c2x2, c1x3 leave out only terms below
//     2^-24 of a product.  Twice the matrix work, any magnitude.
//
// Layout of the computation (same classes / tiles / shifted zero-padded windows as fir_periodic.h):
//   D[class 16][period 16] += A[class][k 32] * B[k][period]      v_mfma_f32_16x16x32_{f16,bf16}
//   * one workgroup per CU, 16 waves: 6 producers (five stagers + one wrap-only) + 10 consumers; consumer T
//     owns class tile T for the whole launch and keeps its coefficient tile -- planes x window/32 steps x 4
//     registers -- in VGPRs: the class table is read once per workgroup, not once per work unit;
//   * an LDS image holds 16 periods of both channels as 16-bit planes, TRANSPOSED: row = frame inside the
//     period (0 .. end of the last tile's window), per row and (channel, plane) one 32-byte plane row (16
//     periods side by side) + 32 bytes of padding (an odd number of 32-byte units per row: conflict-free
//     transposed reads).  Any window start is then a row address (no alignment constraint), and
//     ds_read_b64_tr_b16 delivers the B operand -- 4 consecutive frames x 16 periods per 16 lanes -- at the
//     full 256 B/clk, every offset an immediate.  Rows beyond the period repeat the next period's first
//     frames.  Two planes: 160-byte rows, three images in the ring; three planes: 224 bytes, two images;
//   * producers load frames from HBM (16 bytes per lane, coalesced along the frame index), cut them into
//     planes (fp16: one v_cvt_pk_f16_f32 per pair and plane), pack four periods into 8 bytes and write each
//     chunk twice (row k, and row k + a of the previous period) with ds_write_b64, chunks XOR-swizzled by
//     the row so that the writes spread over the banks; the next item's loads are in flight while the
//     current one is written (inline-asm loads, explicit vmcnt);
//   * ring of images, monotonic LDS counters instead of barriers; the wrap variant of class 0 (row 1023
//     on the previous frame, :562-564) is computed by producers in f32 from global memory.
#include "fir_periodic.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>
#include <vector>
#include <mutex>

#include "common.h"

namespace rsmp {

namespace {

constexpr uint32_t kProducers = 6, kConsumers = 10, kWaves = 16;   // five stagers + one wrap-only producer
constexpr uint32_t kCtrlBytes = 256;                 // staged[2], done[2]
constexpr uint32_t kWrapBytes = 4 * 16 * 16;         // up to four slots x 16 periods x (ch0, ch1, take, -)
constexpr uint32_t kTouchBytes = 3 * 256;              // landing zone of the consumers' L2 prefetch touches
constexpr uint32_t kImageBase = kCtrlBytes + kWrapBytes + kTouchBytes;
// An image row: per channel and plane one 32-byte plane row (16 periods x 16 bits), + 32 bytes of padding.
// In 32-byte units the stride is 7 (three bf16 planes) or 5 (two fp16 planes): odd, so the eight rows a
// transposed read touches per 32 lanes fall on distinct bank groups (64 banks x 4 B = 8 units).
__host__ __device__ constexpr uint32_t row_bytes(int planes) { return planes == 3 ? 7u * 32u : 5u * 32u; }
// Two-plane split: the operands are scaled by powers of two before they are cut into fp16 planes, so that
// samples down to 2^-26 and taps down to 2^-27 keep their full relative precision (fp16 normals start at
// 2^-14); samples of magnitude >= 16 overflow to infinity there -- the launch's non-finite check then has
// the item recomputed in the reference's f32 form.
constexpr float kXScale = 4096.0f;        // 2^12 (the sample scale of an item without samples; see `peak`)
#ifndef RSMP_EXP
#define RSMP_EXP 0   // A/B builds (make exp): timing experiments, never shipped
#endif
constexpr uint32_t kPeakHeadroom = 4;     // a predicted scale leaves 2^4 above the pair's latest peak
constexpr uint32_t kPeakQuiet = 10;       // an item whose peak lies more than 2^-10 below what its scale allows (2^-6 below the prediction) is redone
constexpr uint32_t kPeakMax = 138;        // no scale is derived from a peak of 2^11 and above (samples of 2^13 and above overflow the planes: redone)
constexpr float kCScale = 8192.0f;        // 2^13
constexpr uint32_t kTouchAhead = 3;                    // items between a touch and the producers' loads of the same frames
constexpr uint32_t kLdsLimit = 160 * 1024;
#ifndef RSMP_POLL_SLEEP
#define RSMP_POLL_SLEEP 1   // s_sleep units (64 cycles) between two polls of an LDS counter
#endif
constexpr int kWrapTaps = 8;                         // taps of the wrap variant per lane (16 lanes per period)
constexpr uint32_t kStagers = 5;                          // (row block, period pair) combos in flight per producer

struct SplitArgs {
    uint32_t a, b, taps, n_tiles, rows, slots, lds_bytes, blocks_per_stream, total_items, debug;
    uint32_t n_streams, fuse_tail;   // fuse_tail: also copy every stream's still-buffered tail into hist_next
    uint32_t cstride, pairs;         // WIDE kernels: channels of a frame and channel pairs = cstride / 2 (an item = one pair of a block)
    uint32_t groups, per_group;      // more than ten class tiles (b > 160): tile groups of ten; items per slice (the launch's items = groups x quads x that)
    uint32_t quads, qpairs;          // channel counts of 8, 12, 16: the launch's items are quad-major (see Cursor); pairs per slice (2, or all)
    uint32_t wide;                   // frames of other than two channels (the WIDE instantiations)
    uint32_t nowrap;                 // an exact ratio (super period: no output ever takes the wrap variant) on a WIDE build: its windows are not fetched
    const uint32_t* items;           // the launch's item table (split_items_kernel): kItemWords words per item
    unsigned long long* wtrace;   // RSMP_FIR_WTRACE: kWtraceSlots timestamped events per wave
    NfArgs nf;                    // non-finite sums are marked here (fir_nonfinite.h)
};

constexpr uint32_t kWtraceSlots = 16;

// Diagnostic per-wave phase clock (RSMP_FIR_WTRACE): event(tag) adds the shader-clock cycles since the
// wave's previous event to the bucket of the previous tag (in LDS, 16 buckets per wave, written out at
// the end).  Cheap enough not to change what it measures: one s_memtime and one LDS add per event.
template <bool DIAG>
struct WaveTrace {
    unsigned long long* out;
    unsigned long long* acc;     // LDS
    unsigned long long last;
    uint32_t prev;
    __device__ __forceinline__ void init(const SplitArgs& g, char* lds, uint32_t wave) {
        out = DIAG && g.wtrace ? g.wtrace + (static_cast<size_t>(blockIdx.x) * 16 + wave) * kWtraceSlots : nullptr;
        acc = reinterpret_cast<unsigned long long*>(lds + g.lds_bytes) + wave * kWtraceSlots;
        last = 0;
        prev = 0;
    }
    __device__ __forceinline__ void event(uint32_t tag) {
        if constexpr (!DIAG) return;
        if (out) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (last != 0 && (threadIdx.x & 63) == 0)
                (void)__hip_atomic_fetch_add(acc + prev, now - last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            last = now;
            prev = tag & 15;
        }
    }
    __device__ __forceinline__ void flush() {
        if constexpr (!DIAG) return;
        if (out) {
            __builtin_amdgcn_s_waitcnt(0);
            const uint32_t l = threadIdx.x & 63;
            if (l < kWtraceSlots) out[l] = acc[l];
        }
    }
};

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef const float __attribute__((address_space(1)))* gconst_f32_ptr;
typedef const v2f __attribute__((address_space(1)))* gconst_f2_ptr;
typedef const v4u __attribute__((address_space(1)))* gconst_u4_ptr;
typedef const uint32_t __attribute__((address_space(1)))* gconst_u32_ptr;
typedef float __attribute__((address_space(1)))* g_f32_ptr;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __forceinline__ uint32_t lds_load_acquire(uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// One count per WAVE (a wave-wide atomic would add one per active lane): everything the wave did in
// LDS before is complete when the count becomes visible.
__device__ __forceinline__ void lds_signal(uint32_t* p) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0)
        (void)__hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The same for a wave whose only business with the others is LDS: a release fence would also wait for the
// wave's global stores to be acknowledged (vmcnt(0)) -- for a consumer that is the previous item's output,
// not anything the other waves read.
__device__ __forceinline__ void lds_signal_local(uint32_t* p) {
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
    if ((threadIdx.x & 63) == 0)
        (void)__hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" : : : "memory");
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
    uint64_t q_first;     // abs_out / b: the period holding the launch's first output
    uint32_t sidx;        // the stream's index in the launch
};

// Through the scalar cache (constant address space): the descriptors do not change during the launch,
// and scalar loads are counted by lgkmcnt -- vector loads here would sit in the producers' vmcnt queue.
__device__ __forceinline__ StreamCtx load_stream(const FirStreamDesc* descs, uint32_t s, uint32_t b) {
    typedef const uint32_t __attribute__((address_space(4)))* const_u32_ptr;
    FirStreamDesc d;
    const_u32_ptr src = (const_u32_ptr)(descs + s);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&d);
#pragma unroll
    for (size_t i = 0; i < sizeof(FirStreamDesc) / 4; ++i) dst[i] = src[i];
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
    c.q_first = d.abs_out / b;   // (64-bit division: once per stream, not per item)
    c.sidx = s;
    return c;
}

struct Item {
    uint64_t q0;          // first period of the image
    int32_t n_block0;     // launch-relative output index of (period q0, class 0)
    int32_t k_block0;     // wrap-bitmap index of (period q0, class 0)
    bool valid;
};

// Walks a workgroup's items in order without per-item divisions or 64-bit multiplications: the item
// values of a stream advance by constants from one block to the next.
// WIDE (streams of 4, 6, 8 .. 16 channels): an item is one channel PAIR of a block, the pairs of a block are
// consecutive items -- the workgroup that staged a block's first pair finds the lines of the others in its L2,
// and their 8-byte stores into the same lines meet there.
// More than ten class tiles (up to 320 classes: 44.1 -> 96 kHz, 48 -> 96 kHz): the launch's items come in tile GROUPS
// of ten tiles, group-major -- the workgroups of the first half of the grid take every block with tiles 0 .. 9, those
// of the second half the same blocks with tiles 10 .. 19 at about the same time (workgroup w and w + grid / 2 share an
// XCD under round-robin placement, so the second read of a block's frames is an L2 hit).  A consumer keeps its
// coefficient tile in registers across the items of a group.
// Frames of 8, 12 or 16 channels are cut the same way into QUADS (two channel pairs = the sixteen bytes a stager load
// covers): a quad of a block is two consecutive items, the quads of a block belong to different slices of the launch --
// different workgroups at about the same time -- instead of following each other on one workgroup a whole block of
// staging apart, by when the lines they share have left the L2 (config 5: 2.7x the algorithmic HBM traffic).
// The items of a launch are tabulated once, by a small launch in front of the kernel (split_items_kernel: one thread per
// item): every one of a workgroup's sixteen waves walks the same items, and deriving an item from its index -- stream,
// block, pair, tile group, period, output and frame offsets, whether its frames lie inside the input -- was several
// hundred cycles of scalar 64-bit arithmetic per wave and item (round 2's phase clocks: "find next" / "next item").
constexpr uint32_t kItemWords = 8;   // stream | flags << 24, pair | group << 8, n_block0, k_block0, f0 (2), off0, -
constexpr uint32_t kItemValid = 1, kItemInterior = 2;
struct Cursor {
    uint32_t item;                  // the next item to look at
    uint32_t cur_pair, cur_group;   // channel pair and tile group of the item `next` returned
    uint32_t cur_stream;            // the stream `c` describes (0xFFFFFFFF: none yet)
    StreamCtx c;
    int32_t n_block0, k_block0;     // launch-relative output index / wrap-bitmap index of (first period, class 0)
    int64_t f0;                     // frame index of (first period, row 0) in [hist|in]
    uint32_t off0;                  // interior items: f0 - hist_frames
    bool interior;                  // the image and its wrap windows lie inside `in` (32-bit byte offsets)
    template <bool WIDE>
    __device__ __forceinline__ void init(const SplitArgs&, uint32_t first) {
        item = first;
        cur_pair = cur_group = 0;
        cur_stream = 0xFFFFFFFFu;
        c = StreamCtx{};
        n_block0 = k_block0 = 0;
        f0 = 0;
        off0 = 0;
        interior = false;
    }
    // Finds the next valid item before `end`; returns false when there is none.  On success the fields describe it and
    // `found` is its index.
    template <bool WIDE>
    __device__ __forceinline__ bool next(const SplitArgs& g, const FirStreamDesc* descs, uint32_t end, uint32_t& found) {
        typedef const uint32_t __attribute__((address_space(4)))* const_u32_ptr;
        while (item < end) {
            const_u32_ptr rec = (const_u32_ptr)(g.items + static_cast<size_t>(item) * kItemWords);
            uint32_t w[kItemWords];
#pragma unroll
            for (uint32_t k = 0; k < kItemWords; ++k) w[k] = rec[k];
            found = item;
            ++item;
            if (!((w[0] >> 24) & kItemValid)) continue;
            const uint32_t stream = w[0] & 0xFFFFFFu;
            if (stream != cur_stream) {
                c = load_stream(descs, stream, g.b);
                cur_stream = stream;
            }
            interior = ((w[0] >> 24) & kItemInterior) != 0;
            cur_pair = w[1] & 255u;
            cur_group = w[1] >> 8;
            n_block0 = static_cast<int32_t>(w[2]);
            k_block0 = static_cast<int32_t>(w[3]);
            f0 = static_cast<int64_t>((static_cast<uint64_t>(w[5]) << 32) | w[4]);
            off0 = w[6];
            return true;
        }
        return false;
    }
};

// One thread per item of the launch: the item's record (kItemWords words, see Cursor).  Item order: slices (tile group x
// quad), inside a slice streams, blocks of 16 periods, the slice's channel pairs.
__device__ __forceinline__ void split_item_record(const FirStreamDesc* __restrict__ descs, const SplitArgs& g,
                                                  uint32_t* __restrict__ items, const uint32_t i) {
    if (i >= g.total_items) return;
    const uint32_t slice = i / g.per_group, in_slice = i - slice * g.per_group;
    const uint32_t group = slice / g.quads, quad = slice - group * g.quads;
    const uint32_t per_stream = g.blocks_per_stream * g.qpairs;
    const uint32_t stream = in_slice / per_stream, rem = in_slice - stream * per_stream;
    const uint32_t block = rem / g.qpairs, pair = quad * g.qpairs + (rem - block * g.qpairs);
    const FirStreamDesc& d = descs[stream];
    const uint64_t q0 = d.abs_out / g.b + static_cast<uint64_t>(block) * 16u;
    const uint64_t q_limit = d.n_out != 0 ? (d.abs_out + d.n_out + g.b - 1) / g.b : 0;
    const bool valid = q0 < q_limit;
    const int64_t f0 = static_cast<int64_t>(q0 * g.a) - static_cast<int64_t>(d.abs_consumed);
    const int64_t hf = d.hist_frames;
    const uint32_t fsb = g.cstride * 4u;
    const bool interior = f0 > hf && f0 + static_cast<int64_t>(17u * g.a + (g.wide ? 6u : 2u)) <= hf + static_cast<int64_t>(d.in_frames) &&
                          (g.wide ? static_cast<uint64_t>(d.in_frames) * fsb < (1ull << 32) - 65536u : d.in_frames < (1u << 27));   // (32-bit byte offsets)
    uint32_t* w = items + static_cast<size_t>(i) * kItemWords;
    w[0] = stream | ((valid ? kItemValid : 0u) | (interior ? kItemInterior : 0u)) << 24;
    w[1] = pair | group << 8;
    w[2] = static_cast<uint32_t>(static_cast<int32_t>(static_cast<int64_t>(q0 * g.b) - static_cast<int64_t>(d.abs_out)));
    w[3] = static_cast<uint32_t>(static_cast<int32_t>(static_cast<int64_t>(q0) - static_cast<int64_t>(d.wrap_k0)));
    w[4] = static_cast<uint32_t>(static_cast<uint64_t>(f0));
    w[5] = static_cast<uint32_t>(static_cast<uint64_t>(f0) >> 32);
    w[6] = static_cast<uint32_t>(f0 - hf);
    w[7] = 0;
}
__global__ __launch_bounds__(256) void split_items_kernel(const FirStreamDesc* __restrict__ descs, const SplitArgs g,
                                                          uint32_t* __restrict__ items) {
    split_item_record(descs, g, items, blockIdx.x * 256u + threadIdx.x);
}

// Several jobs (rate pairs: a geometry and the streams that have it) in ONE launch: the launch's workgroups are dealt to
// the jobs in proportion to their work; a workgroup finds its job from its index and is then exactly the workgroup
// `wg` of `n_wgs` of a launch of that job alone.  (A run of config 4 is six rate pairs: six launches of this kernel, of
// the item kernel in front and of the repair kernel behind cost a run of 16 calls more than the work itself.)
constexpr uint32_t kMaxSplitJobs = 8;
struct SplitMulti {
    SplitArgs g[kMaxSplitJobs];
    const FirStreamDesc* descs[kMaxSplitJobs];
    uint32_t wg_end[kMaxSplitJobs];      // workgroups of jobs 0 .. j (fir_split_multi_kernel)
    uint32_t ib_end[kMaxSplitJobs];      // blocks of 256 items of jobs 0 .. j (split_items_multi_kernel)
    uint32_t n_jobs;
    uint32_t build[kMaxSplitJobs];       // fir_split_all_kernel: which body a job's workgroups run (kSplitBuild*)
};
__device__ __forceinline__ uint32_t split_job_of(const uint32_t (&end)[kMaxSplitJobs], uint32_t n_jobs, uint32_t block, uint32_t& begin) {
    uint32_t j = 0;
    begin = 0;
#pragma unroll
    for (uint32_t i = 0; i + 1 < kMaxSplitJobs; ++i) {
        const bool past = i + 1 < n_jobs && block >= end[i];
        j += past ? 1u : 0u;
        begin = past ? end[i] : begin;
    }
    return j;
}
__global__ __launch_bounds__(256) void split_items_multi_kernel(const SplitMulti m) {
    uint32_t begin;
    const uint32_t j = split_job_of(m.ib_end, m.n_jobs, blockIdx.x, begin);
    split_item_record(m.descs[j], m.g[j], const_cast<uint32_t*>(m.g[j].items), (blockIdx.x - begin) * 256u + threadIdx.x);
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
// f32 pair -> fp16 pair (round to nearest, one v_cvt_pk_f16_f32): low half = a, high half = b
__device__ __forceinline__ uint32_t cvt_pk_f16(float a, float b) {
    const v2f v = v2f{a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// s - (float)half of w in ONE instruction (v_fma_mix_f32: the fp16 operand is widened inside the FMA; the result is exact
// either way).  The stagers are bound by what one wave can issue: the compiler's v_cvt_f32_f16 + v_sub_f32 per sample, and
// its v_pk_mul_f32 / v_pk_add_f32 pairs with a v_mov per operand to line the registers up, were a quarter of the split.
__device__ __forceinline__ float resid_lo(float s, uint32_t w) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(s));
    return r;
}
__device__ __forceinline__ float resid_hi(float s, uint32_t w) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(s));
    return r;
}
__device__ __forceinline__ float mul_plain(float a, float b) {   // (one v_mul_f32: not paired into v_pk_mul_f32)
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (high half of hi) : (high half of lo)
__device__ __forceinline__ uint32_t pack_hi16(uint32_t hi, uint32_t lo) {
    return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}

// Loads the compiler does not see: issued by inline asm and awaited with an explicit vmcnt, so that a
// producer can keep the NEXT item's loads in flight across the loop back edge.  (With plain loads the
// compiler's own wait insertion falls back to vmcnt(0) in this loop -- every store waited for the
// loads just issued, i.e. the full HBM latency seven times per item.)  vmcnt is in order: waiting
// until at most N operations are outstanding completes everything older than the N youngest.
template <bool GUARD = false>
__device__ __forceinline__ const void* uniform_ptr(const void* p) {   // into scalar registers, whatever the compiler thought
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    // The loads that take these scalar registers as base address are inline asm, which the compiler's hazard
    // recogniser does not look into; where a vector instruction wrote them (v_readfirstlane here, or v_readlane
    // restoring a spilled SGPR -- the many-channel build spills a hundred) the ISA wants five wait states before a
    // vector-memory instruction reads them.  Spent here, tied to the two registers so that nothing moves across.
    // (A one-channel build of this kernel read a stale base -- address 0 + offset -- without them.)
    // (every build takes the guard: with the peak bookkeeping the two-channel build spills scalar registers too)
    if constexpr (GUARD) asm volatile("s_nop 4" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<const void*>(static_cast<uint64_t>(hi) << 32 | lo);
}
__device__ __forceinline__ void gload4(v4f& dst, uint32_t byte_off, const void* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base) : "memory");
}
__device__ __forceinline__ void gload2(v2f& dst, uint32_t byte_off, const void* base) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base) : "memory");
}
__device__ __forceinline__ void gload1(uint32_t& dst, uint32_t byte_off, const void* base) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base) : "memory");
}
// PCM input (BITS = 16 / 24 / 32; FirStreamDesc::in_bits): what a prefetch load of two consecutive two-channel FRAMES
// fetches -- 8 bytes (16-bit), 16 bytes of which the first 12 count (24-bit: at any even byte address) or 16 bytes
// (32-bit) -- and its conversion to (ch0, ch1, ch0', ch1') as resample/src/main.rs:128-137 converts a sample.
template <int BITS> struct PcmRaw { typedef v4f type; };
template <> struct PcmRaw<16> { typedef v2f type; };
__device__ __forceinline__ void gload_raw(v4f& dst, uint32_t byte_off, const void* base) { gload4(dst, byte_off, base); }
__device__ __forceinline__ void gload_raw(v2f& dst, uint32_t byte_off, const void* base) { gload2(dst, byte_off, base); }
// one frame (ch0, ch1) from the eight bytes loaded at its first byte
template <int BITS>
__device__ __forceinline__ v2f pcm_frame1(const v2f& raw) {
    const uint32_t w0 = __float_as_uint(raw.x), w1 = __float_as_uint(raw.y);
    if constexpr (BITS == 16) {
        return v2f{static_cast<float>(static_cast<int32_t>(w0 << 16) >> 16), static_cast<float>(static_cast<int32_t>(w0) >> 16)} * (1.0f / 32768.0f);
    } else if constexpr (BITS == 24) {
        return v2f{static_cast<float>(static_cast<int32_t>(w0 << 8) >> 8), static_cast<float>(static_cast<int32_t>(((w0 >> 24) | (w1 << 8)) << 8) >> 8)} * (1.0f / 8388608.0f);
    } else if constexpr (BITS == 32) {
        return v2f{static_cast<float>(static_cast<int32_t>(w0)), static_cast<float>(static_cast<int32_t>(w1))} * (-1.0f / 2147483648.0f);
    } else {
        return raw;
    }
}
template <int BITS, class RAW>
__device__ __forceinline__ v4f pcm_frames(const RAW& raw) {
    if constexpr (BITS == 0) {
        return raw;
    } else if constexpr (BITS == 16) {
        const uint32_t w0 = __float_as_uint(raw.x), w1 = __float_as_uint(raw.y);
        auto lo = [](uint32_t w) { return static_cast<float>(static_cast<int32_t>(w << 16) >> 16); };
        auto hi = [](uint32_t w) { return static_cast<float>(static_cast<int32_t>(w) >> 16); };
        return v4f{lo(w0), hi(w0), lo(w1), hi(w1)} * (1.0f / 32768.0f);
    } else if constexpr (BITS == 24) {
        const uint32_t w0 = __float_as_uint(raw.x), w1 = __float_as_uint(raw.y), w2 = __float_as_uint(raw.z);
        auto s24 = [](uint32_t w) { return static_cast<float>(static_cast<int32_t>(w << 8) >> 8); };
        return v4f{s24(w0), s24((w0 >> 24) | (w1 << 8)), s24((w1 >> 16) | (w2 << 16)), static_cast<float>(static_cast<int32_t>(w2) >> 8)} * (1.0f / 8388608.0f);
    } else {
        auto f = [](float w) { return static_cast<float>(static_cast<int32_t>(__float_as_uint(w))); };
        return v4f{f(raw.x), f(raw.y), f(raw.z), f(raw.w)} * (-1.0f / 2147483648.0f);
    }
}

// Sum over the 16 lanes of a DPP row (every lane gets the total).
__device__ __forceinline__ float row_sum16(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}



// A producer's view of a work item (wave-uniform).
struct PItem {
    uint32_t item;        // index in the launch; item_end = none
    uint32_t pair;        // WIDE: the item's channel pair
    Item it;
    int64_t f0;           // frame index of (period q0, row 0) in [hist|in]
    bool interior;        // the image and its wrap windows lie inside `in`, below 2^28 frames
    uint32_t off0;        // interior: f0 - hist_frames
};

// Wave roles.  A workgroup's waves go to the four SIMDs cyclically, so waves w and w + 4 share one.
// Ten consumers: three on SIMD 0 (waves 4, 8, 12) and SIMD 1 (5, 9, 13), two on SIMD 2 (10, 14) and
// SIMD 3 (11, 15).  Five producers: one beside SIMD 1's consumers (wave 1), two each on SIMDs 2 and 3
// (waves 2, 6 and 3, 7); wave 0 exits, so SIMD 0's three consumers have their SIMD to themselves
// (measured 1.5 % faster than a producer there and an empty slot on SIMD 3).
__device__ __forceinline__ bool wave_is_producer(uint32_t w) { return w < 4 || w == 6 || w == 7; }
__device__ __forceinline__ uint32_t producer_index(uint32_t w) { return w == 0 ? 5 : (w < 4 ? w - 1 : w - 3); }
__device__ __forceinline__ uint32_t consumer_index(uint32_t w) { return w < 6 ? w - 4 : w - 6; }

// DIAG: the diagnostic instantiation (RSMP_FIR_DEBUG switches for timing experiments, RSMP_FIR_WTRACE phase
// clocks); in the shipping instantiation `dbg` is the constant 0 and every such test folds away.
// WIDE: streams of 4, 6, 8 .. 16 channels, as channel pairs (see Cursor): a frame is g.cstride floats, the pair's
// two channels 8 bytes inside it.  The stagers load 16 bytes = two pairs of one frame (load_task_wide: issued for
// the even pair, kept in registers for the odd one), the wrap passes load 8 bytes per frame, the consumers store
// 8 bytes per frame; everything between the loads and the stores is the two-channel kernel.  The wrap-only
// producer and the stagers are separate instantiations of the staging loop (ROLE), so that neither pays for the
// other's registers (two wrap passes against one + the 40 registers of a stager's loads).
// ROUNDS: lane tasks per stager lane and item -- 2 for periods of 161 .. 320 frames (96 -> 44.1 kHz, 96 -> 48 kHz): both
// rounds' loads are in flight together, one item ahead, like the single round's.
template <int NK, int PLANES, bool DIAG, int WIDE, int ROUNDS = 1, int BITS = 0>   // WIDE: 0 two channels, 1 channel pairs, 2 one channel, 3 pairs + a last channel alone; BITS: the input's PCM width (0: f32)
__device__ __forceinline__ void fir_split_body(const FirStreamDesc* __restrict__ descs, const SplitArgs& g,
                                               const uint32_t wg, const uint32_t n_wgs) {   // workgroup `wg` of the `n_wgs` that share g's items
    static_assert(ROUNDS == 1 || ((WIDE == 0 || WIDE == 1) && PLANES == 2), "two rounds: two channels or channel pairs, fp16 planes");
    constexpr uint32_t kRowBytes = row_bytes(PLANES);
    const uint32_t fs = WIDE ? g.cstride : 2u;   // floats per frame
    const uint32_t fsb = fs * 4u;                // bytes per frame
    // One channel: a pair whose second channel is a phantom -- the loads take the following frame's sample for it, its
    // sums are computed and dropped (half the matrix work of a pair is waste: still faster than the vector kernel).
    constexpr bool mono = WIDE == 2;
    // WIDE == 3, an odd channel count: the last pair is the last channel + a phantom (the next frame's first channel is
    // what the loads deliver for it; its sums are dropped); pointers into such frames are only 4-byte aligned.
    constexpr bool kOdd = WIDE == 3;
    auto phantom = [&](uint32_t pair) { return kOdd && pair + 1 == g.pairs; };
    typedef v2f __attribute__((address_space(1), aligned(4)))* g_f2u_ptr;
    const uint32_t dbg = DIAG ? g.debug : 0u;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    uint32_t* ctrl = reinterpret_cast<uint32_t*>(lds);
    uint32_t* staged = ctrl;        // [slot]: producers that finished staging, cumulative
    uint32_t* done = ctrl + 4;      // [slot]: consumers that finished reading, cumulative
    // Two-plane split: the image of an item is block floating point -- its samples are scaled by a power of two that
    // puts the item's peak near the top of the fp16 range before they are cut into planes, so quiet passages, loud ones
    // and signals far outside [-1, 1] all keep 22 significant bits per sample down to 2^-13 of the item's peak (and
    // 2^-38 of the peak in absolute terms below that); the consumers undo the scale.  The scale must be the same for
    // every stager of an item and known before the first conversion:
    //   * normally it is PREDICTED: every stager adds its share of the item's true peak to `peak[slot]` (one atomic
    //     max, no waiting); when the consumers have the complete image they copy the final value to `fin[slot]`, and
    //     the stagers of the item that reuses the slot -- `slots` items later, all of them after the same wait -- read
    //     it into a running table of the stream's latest peak per channel pair.  The scale of an item allows 2^4 above
    //     the latest peak of its pair.  A louder item overflows a plane (non-finite sums), a much quieter one is seen by
    //     the consumers (final peak against the scale used): both mark their outputs for the repair launch
    //     (fir_nonfinite.h), which evaluates them in the reference's f32 form;
    //   * an item without history (the first items of a stream in this workgroup) takes the exact peak: the stagers
    //     meet at a counter (`premax`) after adding their shares.
    uint32_t* premax = ctrl + 8;    // stagers that have added their share of an item's peak and wait for the others, cumulative
    uint32_t* peak = ctrl + 36;     // [slot][2 channels]: (use + 1) << 8 | biased exponent of the peak so far (monotonic: never reset)
    uint32_t* used = ctrl + 16;     // [slot]: biased exponent the item's scale was derived from (stager 0)
    uint32_t* fin = ctrl + 20;      // [slot][2]: (stream << 4 | pair) + 1 and final peak exponent of the slot's last item (consumer 0)
    uint32_t* pubd = ctrl + 32;     // [slot]: stagers that have added their share of a PREDICTED item's peak (behind its `staged` count), cumulative
    // Round 5: the two channels of a pair have a scale EACH (`peak`, `used`, `fin` are channel 0's, these channel 1's): the
    // reference computes every channel on its own (src/resampler_fir.rs:567-586), and with one exponent per pair a channel
    // 2^-20 below its partner kept 8e-7 .. 9e-6 of its own level instead of the 1e-6 the path is held to (VERDICT r04
    // missing #3).  The planes of a channel only ever meet that channel's accumulator, so nothing else changes.
    // (`peak` is [slot][2] now: the two channels' words side by side -- one 8-byte LDS read for the consumers --, `used`
    // and `fin`'s exponent word carry channel 0's exponent in bits 0-7 and channel 1's in bits 8-15.)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < (g.lds_bytes + (DIAG && g.wtrace ? 16 * kWtraceSlots * 8 : 0)) / 4; i += blockDim.x)
        ctrl[i] = 0;   // counters; finite image rows
    __syncthreads();

    const uint32_t R = g.rows;
    const uint32_t image_bytes = R * kRowBytes;
    const uint32_t n_active = g.n_tiles < kConsumers ? g.n_tiles : kConsumers;   // consumers that take part (with tile groups: all ten)
    const uint32_t item_begin = static_cast<uint32_t>(static_cast<uint64_t>(wg) * g.total_items / n_wgs);
    const uint32_t item_end = static_cast<uint32_t>(static_cast<uint64_t>(wg + 1) * g.total_items / n_wgs);
    // The frames that stay buffered after the launch move to hist_next (what fir_tail_copy_kernel does in
    // launches that mix kernels): nobody in this launch reads hist_next, the wave with most slack does it.
    if (g.fuse_tail && wave == 0) {
        typedef const uint32_t __attribute__((address_space(4)))* const_u32_ptr;
        for (uint32_t sidx = wg; sidx < g.n_streams; sidx += n_wgs) {
            FirStreamDesc d;
            const_u32_ptr src = (const_u32_ptr)(descs + sidx);
            uint32_t* dst = reinterpret_cast<uint32_t*>(&d);
#pragma unroll
            for (size_t i = 0; i < sizeof(FirStreamDesc) / 4; ++i) dst[i] = src[i];
            const uint32_t total = d.tail_frames * fs, first = d.tail_start * fs, hist_values = d.hist_frames * fs;
            for (uint32_t i = lane; i < total; i += 64) {
                const uint32_t sv = first + i;
                d.hist_next[i] = sv < hist_values ? d.hist[sv] : fir_in_value(d, sv - hist_values);
            }
        }
    }
    if (item_begin == item_end) return;

    uint32_t slot = 0, use = 0;   // ring position of the current item: image slot, times the slot was used before
    WaveTrace<DIAG> wt;
    wt.init(g, lds, wave);

    if (wave_is_producer(wave)) {
        // ---- producers -----------------------------------------------------------------------------
        // Producers 0-4 stage the image, producer 5 computes the wrap variant of class 0.
        // A lane task = frames 2K and 2K+1 of five consecutive periods 4Q .. 4Q+4 (both channels, five
        // 16-byte loads): it writes the four periods 4Q..4Q+3 of rows 2K and 2K+1 (one 8-byte chunk per
        // plane and row) and, because row k + a repeats row k of the NEXT period, the chunks of rows
        // 2K+a and 2K+1+a from the periods 4Q+1..4Q+4 -- every frame is split once and written twice.
        // 4 * ceil(a/2) lane tasks per image, 64 per producer.  The loads of the NEXT item are issued
        // into the same registers as soon as the current item has been written: HBM latency is hidden
        // across items.  A global load costs a wave ~100 cycles to issue here, whatever its width: few, wide.
        const uint32_t P = producer_index(wave);
        const uint32_t half_a = (g.a + 1) / 2;
        const uint32_t n_lane_tasks = 4 * half_a;   // <= 64 * kStagers * ROUNDS (split_geometry)
        const uint32_t n_real = (n_lane_tasks + 63) / 64 < kStagers ? (n_lane_tasks + 63) / 64 : kStagers;   // stagers that have lane tasks (the others only signal)
        // (the lane id behind an optimisation barrier per item: otherwise loop-invariant addressing is
        // hoisted out of the item loop, spilled, and each reload from scratch waits for ALL the
        // prefetches in flight -- scratch loads share the in-order vmcnt)
        uint32_t ln = lane;

        Cursor cu;
        cu.template init<(WIDE != 0)>(g, item_begin);
        auto find_next = [&]() -> PItem {   // the next valid item; its stream context is cu.c
            PItem r;
            r.item = item_end;
            r.pair = 0;
            r.it = Item{};
            r.f0 = 0;
            r.interior = false;
            r.off0 = 0;
            uint32_t found;
            if (cu.template next<(WIDE != 0)>(g, descs, item_end, found)) {
                r.item = found;
                r.pair = cu.cur_pair;
                r.it.q0 = 0;
                r.it.n_block0 = cu.n_block0;
                r.it.k_block0 = cu.k_block0;
                r.it.valid = true;
                r.f0 = cu.f0;
                r.interior = cu.interior;   // (the image and its wrap windows lie inside `in`: split_items_kernel)
                r.off0 = cu.off0;
            }
            return r;
        };
        auto fetch_edge = [&](const StreamCtx& c, uint32_t pair, int64_t f) -> v2f {
            const int64_t hf = c.hist_frames, total = hf + static_cast<int64_t>(c.in_frames);
            const bool ok = f >= 0 && f < total;
            const int64_t fc = f < 0 ? 0 : (f >= total ? total - 1 : f);
            v2f v;
            if constexpr (WIDE) {
                if (mono) v = v2f{fc < hf ? ((gconst_f32_ptr)c.hist)[fc] : ((gconst_f32_ptr)c.in)[fc - hf], 0.f};
                else if (phantom(pair)) v = v2f{fc < hf ? ((gconst_f32_ptr)c.hist)[fc * fs + 2 * pair] : ((gconst_f32_ptr)c.in)[(fc - hf) * fs + 2 * pair], 0.f};
                else if constexpr (kOdd) {
                    typedef const v2f __attribute__((address_space(1), aligned(4)))* gconst_f2u_ptr;
                    v = fc < hf ? *(gconst_f2u_ptr)(c.hist + fc * fs + 2 * pair) : *(gconst_f2u_ptr)(c.in + (fc - hf) * fs + 2 * pair);
                } else v = fc < hf ? *(gconst_f2_ptr)(c.hist + fc * fs + 2 * pair) : *(gconst_f2_ptr)(c.in + (fc - hf) * fs + 2 * pair);
            }
            else if constexpr (BITS != 0) v = fc < hf ? ((gconst_f2_ptr)c.hist)[fc] : v2f{fir_pcm_value(c.in, BITS, 2 * static_cast<size_t>(fc - hf)), fir_pcm_value(c.in, BITS, 2 * static_cast<size_t>(fc - hf) + 1)};
            else v = fc < hf ? ((gconst_f2_ptr)c.hist)[fc] : ((gconst_f2_ptr)c.in)[fc - hf];
            if (!ok) v = v2f{0.f, 0.f};
            return v;
        };
        // State carried from one pass to the next: whether there is a current item, whether its loads are
        // in flight, and -- only for an edge item -- its description (an interior item needs none: its
        // data sits in the registers).  One static instance of every asm load: the first pass (no
        // current item yet) only issues the first item's loads; a separate prologue would make the
        // compiler copy registers that are still in flight where its values meet the loop's.  All the
        // loads of an item are issued together and nothing else is in flight when they are used: the
        // wait is a plain vmcnt(0).
        bool have = false, loaded = false;
        PItem nxt = find_next();
        PItem ecur = nxt;
        StreamCtx ectx = cu.c;

        // ROLE 0: the two-channel kernel's producers; WIDE: 1 = the wrap-only producer, 2 = a stager; two rounds
        // (two channels): 1 = the wrap-only producer, 3 = a stager (x[2][5] next to two wrap passes spilled)
        auto staging = [&](auto role_c) {
            constexpr int ROLE = decltype(role_c)::value;
            constexpr int kMaxPass = ROLE >= 2 ? 1 : 2;   // wrap passes (of four periods) a wave may take at a time
            // ---- staging (all producers) + one pass of the wrap variant (producers 0-3) ---------------
            uint32_t tQ[ROUNDS], tK[ROUNDS];   // (one division per round for the whole launch)
            bool real_rd[ROUNDS];
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd) {
                uint32_t t = rd * (64 * kStagers) + P * 64 + lane;
                real_rd[rd] = ROLE != 1 && P < kStagers && rd * (64 * kStagers) + P * 64 < n_lane_tasks;   // (the wrap-only producer stages nothing)
                if (t >= n_lane_tasks) t = n_lane_tasks - 1;   // surplus lanes repeat the last lane task
                tQ[rd] = t / half_a;
                tK[rd] = t - tQ[rd] * half_a;
            }
            const bool real_task = real_rd[0];
            uint64_t hist_e = 0, hist_e1 = 0;     // latest final peak exponent per channel pair and channel (8 bits each; 0 = none) ...
            uint32_t hist_stream = 0xFFFFFFFFu;   // ... of this stream
            uint32_t n_met = 0;                   // meetings at `premax` so far
            uint32_t cstream = cu.c.sidx;         // the current item's stream
            uint32_t cur_off0 = 0;                // ... where its frames start in `in` (two rounds: the second is loaded late)
            const float* cur_in = nullptr;
            // real = false: dummy loads of the first bytes of the descriptor array (always mapped), so that
            // every pass through the loop issues the same loads
            constexpr uint32_t kInFrameBytes = BITS == 0 ? 8u : 2u * BITS / 8u;   // two channels a frame
            typedef typename PcmRaw<BITS>::type xraw_t;
            auto load_task = [&](xraw_t (&v)[5], bool real, const PItem& pi, const void* base, int rd) {
                const uint32_t off = real ? (pi.off0 + 4 * tQ[rd] * g.a + 2 * tK[rd]) * kInFrameBytes : 0u;
                const uint32_t step = real ? g.a * kInFrameBytes : 0u;
#pragma unroll
                for (int i = 0; i < 5; ++i) gload_raw(v[i], off + i * step, base);
            };
            // WIDE: a 16-byte load is FOUR channels of one frame -- two channel pairs, i.e. two consecutive items of the
            // block: the loads are issued for the even pair (`base` = its first channel) and stay in the registers
            // for the odd one.  Per item that is five loads as in the two-channel kernel, and every line is used whole.
            // one channel: the two frames of a period are neighbours -- one 8-byte load; the phantom channel of frame 2K
            // is frame 2K + 1, that of frame 2K + 1 is zero
            // two rounds, channel pairs: a 16-byte load is four channels of a frame -- the even pair of a block and its odd
            // partner; a stager converts both items from the same registers (even pair into this slot's image, odd pair
            // into the next slot's), round after round
            auto load_task_quad = [&](v4f (&v)[5][2], const PItem& pi, const void* base, int rd) {
                const uint32_t off = (pi.off0 + 4 * tQ[rd] * g.a + 2 * tK[rd]) * fsb;
                const uint32_t step = g.a * fsb;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    gload4(v[i][0], off + i * step, base);
                    gload4(v[i][1], off + i * step + fsb, base);
                }
            };
            auto load_task_quad_half = [&](v4f (&v)[5][2], const PItem& pi, const void* base, int rd, int fr) {
                const uint32_t off = (pi.off0 + 4 * tQ[rd] * g.a + 2 * tK[rd]) * fsb + fr * fsb;
                const uint32_t step = g.a * fsb;
#pragma unroll
                for (int i = 0; i < 5; ++i) gload4(v[i][fr], off + i * step, base);
            };
            auto load_task_mono = [&](v2f (&v)[5], const PItem& pi, const void* base) {
                const uint32_t off = (pi.off0 + 4 * tQ[0] * g.a + 2 * tK[0]) * 4u;
                const uint32_t step = g.a * 4u;
#pragma unroll
                for (int i = 0; i < 5; ++i) gload2(v[i], off + i * step, base);
            };
            auto load_task_wide = [&](v4f (&v)[5][2], const PItem& pi, const void* base) {
                const uint32_t off = (pi.off0 + 4 * tQ[0] * g.a + 2 * tK[0]) * fsb;
                const uint32_t step = g.a * fsb;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    gload4(v[i][0], off + i * step, base);
                    gload4(v[i][1], off + i * step + fsb, base);
                }
            };
            // v[i] = (ch0, ch1) of frame 2K and (ch0, ch1) of frame 2K+1 in period 4Q+i
            // at(i, fr, c) = channel c of frame 2K + fr in period 4Q + i
            auto store_task = [&](char* img, auto&& at, int rd, v2f xsc, int fr_lo = 0, int fr_hi = 2) {
                typedef uint32_t u2 __attribute__((ext_vector_type(2)));
                const uint32_t tQ_ = tQ[rd];
                const uint32_t k0 = 2 * tK[rd];   // rows k0 and k0 + 1 share a swizzle (k0 is even)
                char* prim = img + k0 * kRowBytes + ((tQ_ ^ ((k0 >> 2) & 3)) << 3);
                // rows past the image: their copies go to the touch landing zone instead -- an unconditional
                // store is cheaper than a predicated one
                char* land = lds + kCtrlBytes + kWrapBytes;
                const uint32_t kd0 = k0 + g.a, kd1 = k0 + 1 + g.a;
                char* dup0 = kd0 < R ? img + kd0 * kRowBytes + ((tQ_ ^ ((kd0 >> 2) & 3)) << 3) : land;
                char* dup1 = kd1 < R ? img + kd1 * kRowBytes + ((tQ_ ^ ((kd1 >> 2) & 3)) << 3) : land;
                char* prim1 = k0 + 1 < R ? prim + kRowBytes : land;
#pragma unroll
                for (int c = 0; c < 2; ++c) {          // channel
#pragma unroll
                    for (int fr = 0; fr < 2; ++fr) {   // frame 2K + fr
                        if (fr < fr_lo || fr >= fr_hi) continue;   // (a compile-time range at every call site)
                        char* pr = fr ? prim1 : prim;
                        char* du = fr ? dup1 : dup0;
                        if constexpr (PLANES == 3) {
                            uint32_t pl[3][5];
#pragma unroll
                            for (int i = 0; i < 5; ++i) split3(at(i, fr, c), pl[0][i], pl[1][i], pl[2][i]);
#pragma unroll
                            for (int p = 0; p < 3; ++p) {
                                *reinterpret_cast<u2*>(pr + (3 * c + p) * 32) =
                                    u2{pack_hi16(pl[p][1], pl[p][0]), pack_hi16(pl[p][3], pl[p][2])};
                                *reinterpret_cast<u2*>(du + (3 * c + p) * 32) =
                                    u2{pack_hi16(pl[p][2], pl[p][1]), pack_hi16(pl[p][4], pl[p][3])};
                            }
                        } else {
                            // two fp16 planes: h1 = RN16(s), h2 = RN16(s - h1) with s = 2^12 x (s - h1 is exact)
                            float s[5];
#pragma unroll
                            for (int i = 0; i < 5; ++i) s[i] = mul_plain(at(i, fr, c), c ? xsc.y : xsc.x);
                            const uint32_t a01 = cvt_pk_f16(s[0], s[1]), a23 = cvt_pk_f16(s[2], s[3]), a4 = cvt_pk_f16(s[4], s[4]);
                            const uint32_t b01 = cvt_pk_f16(resid_lo(s[0], a01), resid_hi(s[1], a01));
                            const uint32_t b23 = cvt_pk_f16(resid_lo(s[2], a23), resid_hi(s[3], a23));
                            const uint32_t b4 = cvt_pk_f16(resid_lo(s[4], a4), 0.f);
                            // periods (4Q, 4Q+1), (4Q+2, 4Q+3) to row k; (4Q+1, 4Q+2), (4Q+3, 4Q+4) to row k + a
                            *reinterpret_cast<u2*>(pr + (2 * c) * 32) = u2{a01, a23};
                            *reinterpret_cast<u2*>(pr + (2 * c + 1) * 32) = u2{b01, b23};
                            *reinterpret_cast<u2*>(du + (2 * c) * 32) =
                                u2{__builtin_amdgcn_alignbyte(a23, a01, 2), __builtin_amdgcn_alignbyte(a4, a23, 2)};
                            *reinterpret_cast<u2*>(du + (2 * c + 1) * 32) =
                                u2{__builtin_amdgcn_alignbyte(b23, b01, 2), __builtin_amdgcn_alignbyte(b4, b23, 2)};
                        }
                    }
                }
            };
            // wrap variant of class 0 (row 1023 on the window one frame earlier, resampler_fir.rs:544,
            // :562-565), f32 from global memory, in passes of four periods; lane = (period, 8 taps).  The
            // wrap-only producer (index 5, the wave beside SIMD 0's three consumers) takes passes 0 and 1,
            // the two stagers with most slack (indices 1 and 2) passes 2 and 3.
            const uint32_t n_pass = P == 5 ? 2u : (P == 1 || P == 2 ? 1u : 0u);
            const uint32_t pass0 = P == 5 ? 0u : P + 1;
            const bool wrapper = n_pass != 0;
            #define wpart (ln & 15)
            float wcoef[kWrapTaps];
            const float* cur_coeffs = nullptr;
            xraw_t wx[2][kWrapTaps / 2];
            v2f wxw[kMaxPass][kWrapTaps];   // WIDE: one frame (the pair's two channels) per load
            uint32_t wword[kMaxPass], wsel[kMaxPass];   // the bitmap word with this lane's period's take bit, the bit (32 = none)
#pragma unroll
            for (int ps = 0; ps < kMaxPass; ++ps) {
                wword[ps] = 0;
                wsel[ps] = 32;
            }
            // The wrap variant's windows are fetched only for the periods whose output TAKES it (the bitmap says so: about
            // four periods in ten of the bench's stream, none of an exact ratio's).  A window load is 8 lines per period and
            // instruction where an image load is 8 lines per 64 lanes: fetched for every period the windows were 2.5x the
            // image's address work.  The bitmap words of the NEXT item are requested at the loop's top (load_wrap_bits) and
            // have landed when its windows are requested at the pass's end (load_wrap), lanes without a take switched off.
            // Two-round kernels only: where one round of lane tasks fills the item's time (the 147/160 pair, the tile-group
            // pairs) a vector load in flight across the conversion costs more than the windows' address work saves
            // (measured in one lease: 8 ch 96 -> 44.1 kHz 0.95 -> 0.75 ms, 4 ch 96 -> 44.1
            // 0.29 -> 0.25; the headline 0.286 -> 0.338, 2 ch 44.1 -> 96 kHz 0.64 -> 0.71).
            constexpr bool kWrapByTake = ROUNDS == 2;
            uint32_t wnext[kMaxPass], wnsel[kMaxPass];
#pragma unroll
            for (int ps = 0; ps < kMaxPass; ++ps) {
                wnext[ps] = 0;
                wnsel[ps] = 32;
            }
            auto load_wrap_bits = [&](int ps, const PItem& pi, const StreamCtx& c) {
                const uint32_t wper = 4 * (pass0 + ps) + (ln >> 4);
                const int32_t nw = pi.it.n_block0 + static_cast<int32_t>(wper * g.b);
                const bool in_launch = nw >= 0 && nw < static_cast<int32_t>(c.n_out);
                const uint32_t K = in_launch ? static_cast<uint32_t>(pi.it.k_block0) + wper : 0u;   // (word 0 always exists)
                gload1(wnext[ps], (K >> 5) * 4u, uniform_ptr<true>(c.wrap_bits));
                wnsel[ps] = in_launch ? K & 31u : 32u;
            };
            auto load_wrap = [&](int ps, const PItem& pi, const StreamCtx& c) {
                const uint32_t wper = 4 * (pass0 + ps) + (ln >> 4);
                bool fetch = true;
                if constexpr (kWrapByTake) {
                    wword[ps] = wnext[ps];
                    wsel[ps] = wnsel[ps];
                    fetch = wsel[ps] < 32 && ((wword[ps] >> wsel[ps]) & 1u);   // (lanes of the periods that take it)
                } else if (g.nowrap) {   // (launch-uniform) an exact ratio: nothing to fetch, nothing taken
                    fetch = false;
                    wsel[ps] = 32u;
                } else {   // every window, and the bitmap word with them
                    const int32_t nw = pi.it.n_block0 + static_cast<int32_t>(wper * g.b);
                    const bool in_launch = nw >= 0 && nw < static_cast<int32_t>(c.n_out);
                    const uint32_t K = in_launch ? static_cast<uint32_t>(pi.it.k_block0) + wper : 0u;   // (word 0 always exists)
                    gload1(wword[ps], (K >> 5) * 4u, uniform_ptr<true>(c.wrap_bits));
                    wsel[ps] = in_launch ? K & 31u : 32u;
                }
                if (fetch) {
                    if constexpr (ROLE != 0 && BITS != 0) {
                        // (two rounds, two channels, PCM: a frame per load -- eight bytes from the frame's first byte, of which
                        // its 4 / 6 / 8 count; an interior item's windows end a frame before the input does)
                        const void* base = uniform_ptr<true>(c.in);
                        const uint32_t off = (pi.off0 + wper * g.a - 1 + wpart * kWrapTaps) * kInFrameBytes;
#pragma unroll
                        for (int i = 0; i < kWrapTaps; ++i) gload2(wxw[ps][i], off + i * kInFrameBytes, base);
                    } else if constexpr (ROLE != 0) {
                        const void* base = uniform_ptr<true>(c.in + 2 * pi.pair);
                        const uint32_t off = (pi.off0 + wper * g.a - 1 + wpart * kWrapTaps) * fsb;
#pragma unroll
                        for (int i = 0; i < kWrapTaps; ++i) gload2(wxw[ps][i], off + i * fsb, base);
                    } else {
                        const void* base = uniform_ptr<true>(c.in);
                        const uint32_t off = (pi.off0 + wper * g.a - 1 + wpart * kWrapTaps) * kInFrameBytes;
#pragma unroll
                        for (int i = 0; i < kWrapTaps / 2; ++i) gload_raw(wx[ps][i], off + i * 2u * kInFrameBytes, base);
                    }
                }
            };
            auto wrap_out = [&](const v2f (&w)[kWrapTaps], uint32_t wper, uint32_t take) {
                v2f acc = v2f{0.f, 0.f};
                if (!(dbg & 1024) && !g.nowrap) {
#pragma unroll
                    for (int i = 0; i < kWrapTaps; ++i) {
                        acc.x = fmaf(wcoef[i], w[i].x, acc.x);
                        acc.y = fmaf(wcoef[i], w[i].y, acc.y);
                    }
                    acc.x = row_sum16(acc.x);
                    acc.y = row_sum16(acc.y);
                }
                float* wv = reinterpret_cast<float*>(lds + kCtrlBytes + slot * 256);
                if (wpart == 0) *reinterpret_cast<v4f*>(wv + wper * 4) = v4f{acc.x, acc.y, __uint_as_float(take), 0.f};
            };
            // x / x2 are written by the asm loads ONLY (and read once, behind the wait for them): a register that is in
            // flight across the loop's back edge must have no other definition, or the allocator may give the loop-carried
            // value a second home and copy it there at the latch -- before it has landed (it did: the three-plane build
            // copied x at the latch once the edge path wrote x too, and converted garbage).  The item's samples live in xc.
            xraw_t x[5];
            xraw_t x2[5];         // two rounds, two channels: round 1's loads
            v4f xc[5];            // the current item's (round's) lane task: x, x2 or the edge frames
            constexpr bool kShare = ROLE == 3 && WIDE == 1;   // two rounds, channel pairs: an even item stages its odd partner too
            bool odd_done = false;     // kShare: the current (odd) item was staged with the item before it
            bool wloaded = false;      // the current item's wrap windows are in the registers (as `loaded` for its image)
            uint32_t cur_item = 0;     // the current item's index in the launch
            v4f xq[5][2];         // ROLE 2: (period, frame) x four channels
            v2f xm[5];            // ROLE 2, one channel: frames 2K, 2K + 1 of a period
            uint32_t cpair = 0;   // WIDE: the current item's channel pair
            for (;;) {
                const bool more = nxt.item != item_end;
                if (!have && !more) break;
                bool pre = more && nxt.interior && !(dbg & 8192);   // the next item's loads can be issued ahead
                // (a WIDE stager's odd pair lives on its even neighbour's loads, the item before it in this loop)
                if constexpr (ROLE == 2) pre = pre && ((nxt.pair & 1u) == 0 || (have && loaded));
                if constexpr (kShare) pre = pre && (nxt.pair & 1u) == 0;   // (an odd item is staged by its even neighbour, or from plain loads)
                asm volatile("" : "+v"(ln));
#pragma unroll
                for (int rd = 0; rd < ROUNDS; ++rd) asm volatile("" : "+v"(tQ[rd]), "+v"(tK[rd]));
                char* img = lds + kImageBase + slot * image_bytes;
                uint32_t f_id_v = 0, f_e_v = 0;   // what the slot's previous item turned out to peak at (see `fin`)
                if (have) {
                    wt.event(11);
                    while (lds_load_acquire(done + slot) < n_active * use) __builtin_amdgcn_s_sleep(RSMP_POLL_SLEEP);
                    if constexpr (PLANES == 2) {   // (requested here, used behind the wait for the loads)
                        f_id_v = __hip_atomic_load(fin + 2 * slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        f_e_v = __hip_atomic_load(fin + 2 * slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    wt.event(12);
                }
                asm volatile("s_waitcnt vmcnt(0)" : : : "memory");   // the current item's loads (or dummies)
                // (kShare: an odd item's image comes with its even neighbour's, its wrap windows are prefetched like any item's)
                const bool wpre = pre || (kShare && more && nxt.interior && !(dbg & 8192));
                if constexpr (kWrapByTake) {
                    if (wpre) {
#pragma unroll
                        for (int ps = 0; ps < kMaxPass; ++ps)
                            if (static_cast<uint32_t>(ps) < n_pass) load_wrap_bits(ps, nxt, cu.c);
                    }
                }
                if (have) wt.event(6);
                if constexpr (ROLE == 2 || kShare) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        if constexpr (mono) asm volatile("" : "+v"(xm[i]));
                        else asm volatile("" : "+v"(xq[i][0]), "+v"(xq[i][1]));
                    }
                } else if constexpr (ROLE == 0 || ROLE == 3) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(x[i]));
                }
#pragma unroll
                for (int ps = 0; ps < kMaxPass; ++ps) {
                    if constexpr (ROLE != 0) {
#pragma unroll
                        for (int i = 0; i < kWrapTaps; ++i) asm volatile("" : "+v"(wxw[ps][i]));
                    } else {
#pragma unroll
                        for (int i = 0; i < kWrapTaps / 2; ++i) asm volatile("" : "+v"(wx[ps][i]));
                    }
                    asm volatile("" : "+v"(wword[ps]));
                }
                auto wrap_passes = [&](bool from_regs) {
#pragma unroll
                    for (int ps = 0; ps < kMaxPass; ++ps) {
                        if (static_cast<uint32_t>(ps) >= n_pass) continue;
                        const uint32_t wper = 4 * (pass0 + ps) + (ln >> 4);
                        v2f w[kWrapTaps];
                        if (from_regs) {
#pragma unroll
                            for (int i = 0; i < kWrapTaps / 2; ++i) {
                                if constexpr (ROLE != 0 && BITS != 0) {
                                    w[2 * i] = pcm_frame1<BITS>(wxw[ps][2 * i]);
                                    w[2 * i + 1] = pcm_frame1<BITS>(wxw[ps][2 * i + 1]);
                                } else if constexpr (ROLE != 0) {
                                    w[2 * i] = wxw[ps][2 * i];
                                    w[2 * i + 1] = wxw[ps][2 * i + 1];
                                } else {
                                    const v4f wf = pcm_frames<BITS>(wx[ps][i]);
                                    w[2 * i] = v2f{wf.x, wf.y};
                                    w[2 * i + 1] = v2f{wf.z, wf.w};
                                }
                            }
                            wrap_out(w, wper, wsel[ps] < 32 ? (wword[ps] >> wsel[ps]) & 1u : 0u);
                        } else {
                            const int64_t fw = ecur.f0 + static_cast<int64_t>(wper * g.a) - 1 + wpart * kWrapTaps;
#pragma unroll
                            for (int i = 0; i < kWrapTaps; ++i) w[i] = fetch_edge(ectx, ecur.pair, fw + i);
                            const int32_t nw = ecur.it.n_block0 + static_cast<int32_t>(wper * g.b);
                            const bool in_launch = nw >= 0 && nw < static_cast<int32_t>(ectx.n_out);
                            const uint32_t K = in_launch ? static_cast<uint32_t>(ecur.it.k_block0) + wper : 0u;
                            const uint32_t word = ((gconst_u32_ptr)ectx.wrap_bits)[K >> 5];
                            wrap_out(w, wper, in_launch ? (word >> (K & 31)) & 1u : 0u);
                        }
                    }
                };
                bool signalled = false;   // kShare: the item's `staged` count was given early (below)
                // One round, two channels: the share of a PREDICTED item's peak is added behind the item's `staged` count --
                // nobody needs it before the consumers are through with the item (they wait for `pubd`), and the wave
                // reduction + LDS atomic were ~300 cycles in front of every count.
                constexpr bool kDeferPeak = ROLE == 0 && ROUNDS == 1 && PLANES == 2;
                bool peak_deferred = false;
                v2f peak_m = v2f{0.f, 0.f};
                // the peak of the samples a lane holds (lane_max below), then the wave's, into the item's `peak` word (one atomic)
                // (m = the lane's peak per channel; the exponents travel as two 16-bit fields of one word, so the wave's
                // reduction is as long as it was for one: v_pk_max_u16)
                auto publish = [&](v2f m, uint32_t sl, uint32_t us) {
                    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                    // (non-negative floats order like their bit patterns; a NaN is left to the sums)
                    uint32_t mb = (__float_as_uint(m.x) >> 23) | ((__float_as_uint(m.y) >> 23) << 16);
                    auto pkmax = [](uint32_t a, uint32_t b) -> uint32_t {
                        us2 x, y;
                        __builtin_memcpy(&x, &a, 4);
                        __builtin_memcpy(&y, &b, 4);
                        const us2 z = __builtin_elementwise_max(x, y);
                        uint32_t r;
                        __builtin_memcpy(&r, &z, 4);
                        return r;
                    };
                    mb = pkmax(mb, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(mb), 0x128, 0xf, 0xf, false)));
                    mb = pkmax(mb, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(mb), 0x124, 0xf, 0xf, false)));
                    mb = pkmax(mb, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(mb), 0x122, 0xf, 0xf, false)));
                    mb = pkmax(mb, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(mb), 0x121, 0xf, 0xf, false)));
                    // (lane 0 of each row holds the row's maxima; two more packed maxima over the rows' lane 0 by v_readlane:
                    // the wave's in scalar registers)
                    const uint32_t r01 = pkmax(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mb), 0)),
                                               static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mb), 16)));
                    const uint32_t r23 = pkmax(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mb), 32)),
                                               static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mb), 48)));
                    const uint32_t rr = __builtin_amdgcn_readfirstlane(pkmax(r01, r23));
                    const uint32_t w0 = rr & 0xFFFFu, w1 = rr >> 16;
                    if (lane == 0) {
                        (void)__hip_atomic_fetch_max(peak + 2 * sl, ((us + 1) << 8) | w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        (void)__hip_atomic_fetch_max(peak + 2 * sl + 1, ((us + 1) << 8) | w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                };
                const bool staged_already = kShare && odd_done;   // (an odd item its even neighbour has staged)
                if constexpr (kShare) odd_done = false;
                if constexpr (ROLE != 1) if (have && real_task && !(dbg & 1) && !staged_already) {
                    // stream edges: frames outside [hist|in] read as zero; plain loads into the registers of the asm loads
                    auto fetch_edge_round = [&](int rd) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            const int64_t f = ecur.f0 + static_cast<int64_t>((4 * tQ[rd] + i) * g.a + 2 * tK[rd]);
                            const v2f lo = fetch_edge(ectx, ecur.pair, f), hi = fetch_edge(ectx, ecur.pair, f + 1);
                            if constexpr (ROLE == 2 || kShare) {
                                if constexpr (mono) xm[i] = v2f{lo.x, hi.x};
                                else {
                                    xq[i][0] = v4f{lo.x, lo.y, lo.x, lo.y};
                                    xq[i][1] = v4f{hi.x, hi.y, hi.x, hi.y};
                                }
                            } else {
                                xc[i] = v4f{lo.x, lo.y, hi.x, hi.y};
                            }
                        }
                    };
                    if (!loaded) fetch_edge_round(0);
                    else if constexpr ((ROLE == 0 || ROLE == 3) && !kShare) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) xc[i] = pcm_frames<BITS>(x[i]);
                    }
                    // at(i, fr, c) = channel c of frame 2K + fr in period 4Q + i of the lane task in the registers; partner:
                    // the same of the block's odd pair (kShare: the upper half of the sixteen bytes)
                    auto at = [&](int i, int fr, int c) -> float {
                        if constexpr (ROLE == 2) {
                            if constexpr (mono) return c == 0 ? xm[i][fr] : (fr == 0 ? xm[i][1] : 0.f);
                            // (an odd count's last pair: the phantom channel is ZEROS, not the next frame's first channel the
                            // load delivers.  Its sums are dropped either way, but its planes are not alone in the LDS: the last
                            // tiles of the image in the slot BEFORE read a few rows past their own image -- against zero padding
                            // coefficients -- and an edge item, whose phantom is zeros (fetch_edge), leaves the pair a history of
                            // silence from which the next item's real samples were scaled to infinity: 0 x inf = NaN in the
                            // neighbour's sums, one item per workgroup range and stream redone by the repair pass -- 1.0 ms on
                            // top of a 0.68 ms launch of three-channel streams, profiles/r06/odd_channels_repair.txt)
                            else if (kOdd && c == 1 && phantom(cpair)) return 0.f;
                            else return (cpair & 1u) ? xq[i][fr][2 + c] : xq[i][fr][c];
                        } else if constexpr (kShare) {
                            return xq[i][fr][c];
                        } else {
                            return xc[i][2 * fr + c];
                        }
                    };
                    auto at_partner = [&](int i, int fr, int c) -> float { return kOdd && c == 1 && phantom(cpair + 1) ? 0.f : xq[i][fr][2 + c]; };
                    // the peak of the samples in the registers: this lane's (lane_max), then the wave's and (one atomic) into
                    // the item's (publish) -- once per item and pair where the scale is predicted, once more after the first
                    // round where the scale is taken from it
                    auto lane_max = [&](auto&& src, v2f m, int fr_lo = 0, int fr_hi = 2) -> v2f {
#pragma unroll
                        for (int i = 0; i < 5; ++i)
#pragma unroll
                            for (int fr = 0; fr < 2; ++fr)
#pragma unroll
                                for (int c = 0; c < (mono ? 1 : 2); ++c)
                                    if (fr >= fr_lo && fr < fr_hi) {
                                        if (c) m.y = __builtin_fmaxf(m.y, __builtin_fabsf(src(i, fr, c)));
                                        else m.x = __builtin_fmaxf(m.x, __builtin_fabsf(src(i, fr, c)));
                                    }
                        if constexpr (mono) m.y = m.x;   // (the phantom channel is the same channel's odd frames)
                        return m;
                    };
                    // what a slot's previous item (`slots` items back) turned out to peak at: into the running table
                    auto note_fin = [&](uint32_t f_id, uint32_t f_e01) {
                        if constexpr (WIDE == 0) {   // one pair per stream: the latest final peaks ARE the history (no table)
                            if (f_id != 0) {
                                hist_stream = (f_id - 1) >> 4;
                                hist_e = f_e01;
                            }
                            return;
                        }
                        const uint32_t f_e = f_e01 & 255u, f_e1 = f_e01 >> 8;
                        if (f_id != 0) {
                            const uint32_t f_stream = (f_id - 1) >> 4, f_pair = (f_id - 1) & 15u;
                            if (f_stream != hist_stream) {
                                hist_stream = f_stream;
                                hist_e = hist_e1 = 0;
                            }
                            hist_e = (hist_e & ~(0xFFull << (8 * f_pair))) | (static_cast<uint64_t>(f_e) << (8 * f_pair));
                            hist_e1 = (hist_e1 & ~(0xFFull << (8 * f_pair))) | (static_cast<uint64_t>(f_e1) << (8 * f_pair));
                        }
                    };
                    // (channel 0's exponent in bits 0-7, channel 1's in bits 8-15; 0 = none)
                    auto hist_of = [&](uint32_t pr) -> uint32_t {
                        if constexpr (WIDE == 0) return hist_stream == cstream ? static_cast<uint32_t>(hist_e) : 0u;
                        return hist_stream == cstream ? (static_cast<uint32_t>(hist_e >> (8 * pr)) & 255u) | ((static_cast<uint32_t>(hist_e1 >> (8 * pr)) & 255u) << 8) : 0u;
                    };
                    // an item's scale: predicted from its pair's latest peak, or (no history) from the peak of what the stagers
                    // hold once every one of them has added its share -- the whole item, or (two rounds) its first round:
                    // ten of its sixteen periods, taken with the same headroom as a prediction
                    auto scale_for = [&](uint32_t e_hist, uint32_t sl) -> v2f {
                        uint32_t E = e_hist & 255u, E1 = e_hist >> 8;
                        if (E != 0 && E1 != 0) {   // 2^4 above the latest peak of the pair's channel
                            E += kPeakHeadroom;
                            E1 += kPeakHeadroom;
                        } else {
                            // (one monotonic counter for all slots: nobody gets past meeting k before every stager has
                            // arrived at it, so arrivals at meeting k + 1 cannot be taken for arrivals at k)
                            lds_signal(premax);
                            ++n_met;
                            while (lds_load_acquire(premax) < n_real * n_met) __builtin_amdgcn_s_sleep(RSMP_POLL_SLEEP);
                            E = __hip_atomic_load(peak + 2 * sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & 255u;
                            E1 = __hip_atomic_load(peak + 2 * sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & 255u;
                            if (ROUNDS == 2 && E != 0) E += kPeakHeadroom;
                            if (ROUNDS == 2 && E1 != 0) E1 += kPeakHeadroom;
                        }
                        // (never scaled for peaks of 2^11 and above: such samples must overflow the planes and have the item
                        // redone, not push the audio next to them below the planes' range)
                        E = E > kPeakMax ? kPeakMax : E;
                        E1 = E1 > kPeakMax ? kPeakMax : E1;
                        E = __builtin_amdgcn_readfirstlane(E < 31u ? 31u : E);   // (below 2^-96: treated as that)
                        E1 = __builtin_amdgcn_readfirstlane(E1 < 31u ? 31u : E1);
                        if (P == 0 && lane == 0) used[sl] = E | (E1 << 8);
                        // a peak in [2^(E-127), 2^(E-126)) times 2^(141-E) lies in [2^14, 2^15), inside fp16
                        return v2f{__uint_as_float((268u - E) << 23), __uint_as_float((268u - E1) << 23)};
                    };
                    v2f xs = v2f{kXScale, kXScale}, xs2 = v2f{kXScale, kXScale};   // the item's sample scales (block floating point, one per channel); its partner's
                    v2f m_own = v2f{0.f, 0.f}, m_partner = v2f{0.f, 0.f};   // running peaks of this lane's samples (own pair, partner pair), per channel
                    constexpr bool kLatePeak = ROUNDS == 2;   // (one round: the item's whole peak is known here)
                    if constexpr (PLANES == 2) {
                        note_fin(__builtin_amdgcn_readfirstlane(f_id_v), __builtin_amdgcn_readfirstlane(f_e_v));
                        m_own = lane_max(at, v2f{0.f, 0.f});   // (here: keeping the samples alive behind the count as well cost 11 %)
                        const uint32_t eh_raw = hist_of(cpair);
                        const uint32_t eh = (eh_raw & 255u) != 0 && (eh_raw >> 8) != 0 ? eh_raw : 0u;   // (both channels have a history, or neither)
                        if (kDeferPeak && eh != 0) {
                            peak_deferred = true;
                            peak_m = m_own;
                        } else if (!kLatePeak || eh == 0 || !real_rd[ROUNDS - 1]) {
                            publish(m_own, slot, use);
                            if constexpr (kDeferPeak) lds_signal(pubd + slot);
                        }
                        xs = scale_for(eh, slot);
                    }
                    wt.event(9);
                    // kShare: the block's odd pair from the same registers, into the next slot's image
                    bool share = false;
                    uint32_t slot2 = 0, use2 = 0;
                    char* img2 = img;
                    if constexpr (kShare) {
                        share = loaded && more && (cpair & 1u) == 0 && nxt.item == cur_item + 1 && nxt.pair == cpair + 1;
                        if (share) {
                            slot2 = slot + 1 == g.slots ? 0u : slot + 1;
                            use2 = slot + 1 == g.slots ? use + 1 : use;
                            img2 = lds + kImageBase + slot2 * image_bytes;
                        }
                    }
                    // The partner's image slot is the one the consumers free LAST (they take the quad's items in order), so
                    // it is waited for as late as the shared registers allow: behind the item's own first plane set.
                    auto partner_slot = [&]() {
                        while (lds_load_acquire(done + slot2) < n_active * use2) __builtin_amdgcn_s_sleep(RSMP_POLL_SLEEP);
                        note_fin(__builtin_amdgcn_readfirstlane(__hip_atomic_load(fin + 2 * slot2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)),
                                 __builtin_amdgcn_readfirstlane(__hip_atomic_load(fin + 2 * slot2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)));
                        m_partner = lane_max(at_partner, v2f{0.f, 0.f});
                        const uint32_t eh2_raw = hist_of(cpair + 1);
                        const uint32_t eh2 = (eh2_raw & 255u) != 0 && (eh2_raw >> 8) != 0 ? eh2_raw : 0u;
                        if (eh2 == 0 || !real_rd[ROUNDS - 1]) publish(m_partner, slot2, use2);
                        xs2 = scale_for(eh2, slot2);
                    };
                    PItem pc;   // (of an interior item only where its frames start is needed)
                    pc.off0 = cur_off0;
                    if constexpr (kShare) {
                        // Two rounds through ONE set of registers, pipelined by frame: when the frames 2K of round 0 are
                        // written their registers take round 1's frames 2K, which fly while the frames 2K + 1 are written.
                        const bool ahead = loaded && real_rd[1];
                        store_task(img, at, 0, xs, 0, 1);
                        if (share) {
                            partner_slot();
                            store_task(img2, at_partner, 0, xs2, 0, 1);
                        }
                        if (ahead) load_task_quad_half(xq, pc, uniform_ptr<true>(cur_in), 1, 0);
                        store_task(img, at, 0, xs, 1, 2);
                        if (share) store_task(img2, at_partner, 0, xs2, 1, 2);
                        if (real_rd[1]) {
                            wt.event(10);
                            if (ahead) {
                                load_task_quad_half(xq, pc, uniform_ptr<true>(cur_in), 1, 1);
                                asm volatile("s_waitcnt vmcnt(5)" : : : "memory");   // the frames 2K (the five loads issued last may fly on)
#pragma unroll
                                for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(xq[i][0]));
                            } else {
                                fetch_edge_round(1);
                            }
                            wt.event(3);
                            m_own = lane_max(at, m_own, 0, 1);
                            store_task(img, at, 1, xs, 0, 1);
                            if (share) {
                                m_partner = lane_max(at_partner, m_partner, 0, 1);
                                store_task(img2, at_partner, 1, xs2, 0, 1);
                            }
                            // (unconditional: free when nothing is in flight, and tools/asm_inflight_lint.py follows no
                            // correlated branches)
                            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
                            for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(xq[i][1]));
                            m_own = lane_max(at, m_own, 1, 2);
                            publish(m_own, slot, use);
                            store_task(img, at, 1, xs, 1, 2);
                            if (share) {
                                // the item itself is complete here: its count is given before the partner's last plane set,
                                // so the consumers start on it that much earlier (no load of this wave is in flight)
                                if (wrapper) wrap_passes(wloaded);
                                lds_signal(staged + slot);
                                signalled = true;
                                m_partner = lane_max(at_partner, m_partner, 1, 2);
                                publish(m_partner, slot2, use2);
                                store_task(img2, at_partner, 1, xs2, 1, 2);
                            }
                        }
                    } else if constexpr (ROUNDS == 2) {
                        // Two rounds (two channels): round 1's loads fly while round 0's planes are written (a second set
                        // of registers: twenty more next to a 99-register kernel)
                        const bool ahead = loaded && real_rd[1];
                        if (ahead) load_task(x2, true, pc, uniform_ptr<true>(cur_in), 1);
                        store_task(img, at, 0, xs);
                        if (real_rd[1]) {
                            wt.event(10);
                            if (ahead) {
                                asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#pragma unroll
                                for (int i = 0; i < 5; ++i) {
                                    asm volatile("" : "+v"(x2[i]));
                                    xc[i] = pcm_frames<BITS>(x2[i]);
                                }
                            } else {
                                fetch_edge_round(1);
                            }
                            wt.event(3);
                            if constexpr (PLANES == 2) publish(lane_max(at, m_own), slot, use);
                            store_task(img, at, 1, xs);
                        }
                    } else {
                        store_task(img, at, 0, xs);
                    }
                    if constexpr (kShare) odd_done = share;
                }
                if (have && wrapper && !signalled) {
                    wt.event(13);
                    wrap_passes(wloaded);
                }
                if (more && wrapper && cu.c.coeffs != cur_coeffs) {   // the next item's taps of row 1023 (rare: compiler-visible loads)
                    cur_coeffs = cu.c.coeffs;
                    gconst_f32_ptr wrow = (gconst_f32_ptr)cu.c.coeffs + static_cast<size_t>(1023) * g.taps;
#pragma unroll
                    for (int i = 0; i < kWrapTaps; ++i) {
                        const uint32_t tt = wpart * kWrapTaps + i;
                        wcoef[i] = tt < g.taps ? wrow[tt] : 0.f;
                    }
                    // have them land here: a compiler-inserted wait at their use would also wait for every
                    // prefetch issued in between
#pragma unroll
                    for (int i = 0; i < kWrapTaps; ++i) asm volatile("" : "+v"(wcoef[i]));
                }
                if constexpr (kWrapByTake) {
                    if (wpre && wrapper) {
                        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");   // the bitmap words requested at the top (nothing else is in flight)
#pragma unroll
                        for (int ps = 0; ps < kMaxPass; ++ps) asm volatile("" : "+v"(wnext[ps]));
#pragma unroll
                        for (int ps = 0; ps < kMaxPass; ++ps)
                            if (static_cast<uint32_t>(ps) < n_pass) load_wrap(ps, nxt, cu.c);
                    }
                }
                if (pre) {
                    if constexpr (ROLE == 2) {
                        if constexpr (mono) {
                            if (real_task) load_task_mono(xm, nxt, uniform_ptr<true>(cu.c.in));
                        } else if (real_task && (nxt.pair & 1u) == 0) load_task_wide(xq, nxt, uniform_ptr<true>(cu.c.in + 2 * nxt.pair));
                    } else if constexpr (kShare) {
                        if (real_task) load_task_quad(xq, nxt, uniform_ptr<true>(cu.c.in + 2 * nxt.pair), 0);   // (an even pair: `pre`)
                    } else if constexpr (ROLE == 0 || ROLE == 3) {
                        if (real_task) load_task(x, true, nxt, uniform_ptr<true>(cu.c.in), 0);   // (two rounds: round 0)
                    }
                }
                if constexpr (!kWrapByTake) {
                    if (wpre) {
#pragma unroll
                        for (int ps = 0; ps < kMaxPass; ++ps)
                            if (static_cast<uint32_t>(ps) < n_pass) load_wrap(ps, nxt, cu.c);
                    }
                }
                if (have) {
                    wt.event(4);
                    if (!signalled) lds_signal(staged + slot);
                    if constexpr (kDeferPeak) {
                        if (peak_deferred) {
                            publish(peak_m, slot, use);
                            lds_signal(pubd + slot);
                        }
                    }
                    wt.event(14);
                    if (++slot == g.slots) {
                        slot = 0;
                        ++use;
                    }
                }
                have = more;
                loaded = pre;
                wloaded = wpre;
                if (more && !pre) {   // an edge item comes next: keep its description
                    ecur = nxt;
                    ectx = cu.c;
                }
                if (more) {
                    cpair = nxt.pair;
                    cur_item = nxt.item;
                    cstream = cu.c.sidx;
                    cur_off0 = nxt.off0;
                    cur_in = cu.c.in + (WIDE == 1 ? 2 * nxt.pair : 0u);
                    nxt = find_next();
                }
                if (have) wt.event(5);
            }
            #undef wpart
        };
        if constexpr (WIDE != 0 && ROUNDS == 1) {
            if (P == 5) staging(std::integral_constant<int, 1>{});
            else staging(std::integral_constant<int, 2>{});
        } else if constexpr (ROUNDS == 2) {
            if (P == 5) staging(std::integral_constant<int, 1>{});
            else staging(std::integral_constant<int, 3>{});
        } else {
            staging(std::integral_constant<int, 0>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
        wt.flush();
        return;
    }

    // ---- consumer ----------------------------------------------------------------------------------
    const uint32_t T = consumer_index(wave);
    if (T >= n_active) return;
    const uint32_t grp = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
    const uint32_t pl = lane & 15;                     // the lane's period (D column)
    // the wave's class tile: T in the item's tile group (ten tiles per group; the last group may have fewer -- a consumer
    // without a tile there still waits for the image and counts itself done)
    uint32_t Tt = T, lane_off = 0, j0 = 0;
    uint32_t cur_group = 0xFFFFFFFFu;
    auto set_tile = [&](uint32_t group) {
        Tt = group * kConsumers + T;
        const uint32_t ob = (Tt * 16u * g.a) / g.b;        // first frame of the tile's window
        const uint32_t row0 = ob + 4 * grp + q;
        lane_off = row0 * kRowBytes + ((pc ^ ((row0 >> 2) & 3)) << 3);
        j0 = Tt * 16u + 4 * grp;                           // the lane's four classes (D rows)
    };

    typedef typename std::conditional<PLANES == 3, bf16x8, f16x8>::type frag_t;
    frag_t A[NK][PLANES];
    const float* cur_table = nullptr;
    // The sums of an item whose wave lies wholly inside the launch are stored inside the NEXT item's MFMA
    // stream (the wave mostly waits for the matrix pipe there); only that wave-uniform straight-line case:
    // a divergent store path inside the stream would issue MFMAs under a partial EXEC mask.
    v4f pend_lo = v4f{0.f, 0.f, 0.f, 0.f}, pend_hi = v4f{0.f, 0.f, 0.f, 0.f};
    g_f32_ptr pend_o = nullptr;
    bool pend_ph = false;   // WIDE == 3: the pending sums are the last channel's
    bool pend = false;
    typedef v2f __attribute__((address_space(1)))* g_f2_ptr;
    // a lane's four frames x two channels: 32 contiguous bytes, or (WIDE) 8 bytes in each of four frames
    auto store_frames = [&](g_f32_ptr o, const v4f& lo, const v4f& hi, bool ph) {
        if constexpr (WIDE) {
            if (kOdd && ph) {   // the last channel alone: one value per frame
                o[0] = lo.x;
                o[fs] = lo.z;
                o[2 * fs] = hi.x;
                o[3 * fs] = hi.z;
            } else if constexpr (kOdd) {
                *((g_f2u_ptr)o) = v2f{lo.x, lo.y};
                *((g_f2u_ptr)(o + fs)) = v2f{lo.z, lo.w};
                *((g_f2u_ptr)(o + 2 * fs)) = v2f{hi.x, hi.y};
                *((g_f2u_ptr)(o + 3 * fs)) = v2f{hi.z, hi.w};
            } else if (mono) {   // four frames of the one channel
                typedef v4f __attribute__((address_space(1), aligned(4)))* g_f4a4_ptr;
                *((g_f4a4_ptr)o) = v4f{lo.x, lo.z, hi.x, hi.z};
            } else {
                *((g_f2_ptr)o) = v2f{lo.x, lo.y};
                *((g_f2_ptr)(o + fs)) = v2f{lo.z, lo.w};
                *((g_f2_ptr)(o + 2 * fs)) = v2f{hi.x, hi.y};
                *((g_f2_ptr)(o + 3 * fs)) = v2f{hi.z, hi.w};
            }
        } else {
            typedef v4f __attribute__((address_space(1), aligned(8)))* g_f4a8_ptr;
            ((g_f4a8_ptr)o)[0] = lo;
            ((g_f4a8_ptr)o)[1] = hi;
        }
    };
    auto flush_pending = [&]() {
        if (pend) {   // wave-uniform
            store_frames(pend_o, pend_lo, pend_hi, pend_ph);
            pend = false;
        }
    };
    Cursor cu;
    cu.template init<(WIDE != 0)>(g, item_begin);
    uint32_t item;
    while (cu.template next<(WIDE != 0)>(g, descs, item_end, item)) {
        const StreamCtx& d = cu.c;
        Item it;
        it.q0 = 0;
        it.n_block0 = cu.n_block0;
        it.k_block0 = cu.k_block0;
        it.valid = true;
        if (d.class_coef != cur_table || cu.cur_group != cur_group) {   // streams of one launch may differ in drift
            cur_table = d.class_coef;
            cur_group = cu.cur_group;
            set_tile(cur_group);
            const uint32_t Tl = Tt < g.n_tiles ? Tt : 0u;   // (a consumer without a tile in this group loads any tile)
            gconst_u4_ptr tp = (gconst_u4_ptr)(cur_table) + static_cast<size_t>(Tl) * (NK * PLANES * 64) + lane;
#pragma unroll
            for (int s = 0; s < NK; ++s)
#pragma unroll
                for (int p = 0; p < PLANES; ++p) A[s][p] = __builtin_bit_cast(frag_t, tp[(s * PLANES + p) * 64]);
        }
        const uint32_t base = kImageBase + slot * image_bytes + lane_off;
        // L2 prefetch for the producers: the frames of the item kTouchAhead items on (same stream assumed),
        // one dword per 128-byte line by LDS-DMA into a landing zone nobody reads.  The producers' own
        // loads, issued one item ahead, would otherwise each pay the full HBM latency -- longer than an item.
        if (T < 3 && d.in_frames != 0 && (dbg & 4096)) {   // (off: measured 4 % slower than without)
            int64_t f = cu.f0 - static_cast<int64_t>(d.hist_frames) + static_cast<int64_t>(kTouchAhead * 16u * g.a) +
                        static_cast<int64_t>((T * 64 + lane) * 16u);
            if (f < 0) f = 0;
            if (f >= static_cast<int64_t>(d.in_frames)) f = static_cast<int64_t>(d.in_frames) - 1;
            typedef __attribute__((address_space(3))) void* lds_void_ptr;
            __builtin_amdgcn_global_load_lds((gconst_f32_ptr)d.in + f * fs,
                                             (lds_void_ptr)(lds + kCtrlBytes + kWrapBytes + T * 256), 4, 0, 0);
        }
        wt.event(1);
        if (!(dbg & 16384))
            while (lds_load_acquire(staged + slot) < kProducers * (use + 1)) __builtin_amdgcn_s_sleep(RSMP_POLL_SLEEP);
        wt.event(2);
        if (Tt >= g.n_tiles) {   // (wave-uniform) no tile for this wave in the item's group: done with the image
            flush_pending();
            lds_signal_local(done + slot);
            if (++slot == g.slots) {
                slot = 0;
                ++use;
            }
            continue;
        }

        v4f acc0 = v4f{0.f, 0.f, 0.f, 0.f}, acc1 = v4f{0.f, 0.f, 0.f, 0.f};
        auto frag = [&](uint32_t plane_ch, int s) -> frag_t {
            const uint32_t addr = base + plane_ch * 32u;   // (everything but `base` is an immediate offset)
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + addr + s * (32 * kRowBytes)));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + addr + s * (32 * kRowBytes) + 16 * kRowBytes));
            const s16x8 t = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(frag_t, t);
        };
        if (!(dbg & 2))
#pragma unroll
        for (int s = 0; s < NK; ++s) {
            if constexpr (PLANES == 3) {
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
            } else {
                const f16x8 x1 = frag(0, s), x2 = frag(1, s);
                const f16x8 y1 = frag(2, s), y2 = frag(3, s);
                // c1 x2 + c2 x1 + c1 x1 (c2 x2 is below 2^-22 of a product), smallest first
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][0], x2, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][0], y2, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][1], x1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][1], y1, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][0], x1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[s][0], y1, acc1, 0, 0, 0);
            }
            // (WIDE: an even pair's sums wait for the odd pair of their block -- the next item -- and leave as 16-byte stores)
            if (s == 0 && !(dbg & 131072) && (!WIDE || (cu.cur_pair & 1u) == 0)) {
                __builtin_amdgcn_sched_barrier(0);
                flush_pending();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        bool item_bad = false;   // (wave-uniform) the item's scale was off: its outputs are redone by the repair launch
        if constexpr (PLANES == 2) {
            if constexpr (WIDE == 0 && ROUNDS == 1) if (!(dbg & 1)) {   // the stagers' shares of the peak: all in (added behind `staged`; long since).  (dbg & 1: the timing experiment without staging adds none)
                const uint32_t half_a_c = (g.a + 1) / 2, tasks_c = 4 * half_a_c;
                const uint32_t n_real_c = (tasks_c + 63) / 64 < kStagers ? (tasks_c + 63) / 64 : kStagers;
                while (lds_load_acquire(pubd + slot) < n_real_c * (use + 1)) __builtin_amdgcn_s_sleep(RSMP_POLL_SLEEP);
            }
            const uint32_t e_used01 = __builtin_amdgcn_readfirstlane(__hip_atomic_load(used + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            const uint32_t e_used = e_used01 & 255u, e_used1 = e_used01 >> 8;
            const uint64_t e_act01 = __hip_atomic_load(reinterpret_cast<const uint64_t*>(peak + 2 * slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t e_act = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(e_act01)) & 255u;
            const uint32_t e_act1 = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(e_act01 >> 32)) & 255u;
            // (a channel much quieter than its scale allows: redone.  A channel of zeros has nothing to lose.)
            item_bad = (e_act != 0 && e_act + kPeakQuiet < e_used) || (e_act1 != 0 && e_act1 + kPeakQuiet < e_used1);
            if (T == 0 && lane == 0) {   // for the stagers of the item that takes this slot next (read behind their wait for `done`)
                fin[2 * slot] = ((d.sidx << 4) | (WIDE ? cu.cur_pair : 0u)) + 1u;
                // (a channel of zeros counts as a history too -- the smallest scale -- so that a silent channel does not send
                // every item of its stream to the stagers' meeting)
                fin[2 * slot + 1] = (e_act ? e_act : 1u) | ((e_act1 ? e_act1 : 1u) << 8);
            }
            acc0 *= __uint_as_float((e_used - 27u) << 23);    // 2^(E-141): channel 0's scale undone; 2^-13: the taps'
            acc1 *= __uint_as_float((e_used1 - 27u) << 23);   // ... channel 1's
        }
        if (!WIDE || (cu.cur_pair & 1u) == 0) flush_pending();   // (no MFMA loop ran, or the experiment switch above)
        // class 0 may take the wrap variant the producers left with the image (tile 0, D row 0)
        if (Tt == 0) {
            const v4f w = *reinterpret_cast<const v4f*>(lds + kCtrlBytes + slot * 256 + pl * 16);
            if (grp == 0 && __float_as_uint(w.z) != 0u && !(dbg & 2048)) {
                acc0.x = w.x;
                acc1.x = w.y;
            }
        }
        wt.event(7);
        lds_signal_local(done + slot);

        // lane = (period, 4 consecutive classes), both channels: 32 contiguous bytes
        const int32_t n0 = it.n_block0 + static_cast<int32_t>(pl * g.b + j0);
        const int32_t n_limit = static_cast<int32_t>(d.n_out);
        // a non-finite sum (inf / NaN sample, or one too large for the 16-bit planes): the chunk is redone
        // in the reference's form by the repair launch
        nf_mark(g.nf, item_bad || nf_is_bad(mono || phantom(cu.cur_pair) ? (acc0.x + acc0.y) + (acc0.z + acc0.w) : nf_sum8(acc0, acc1)), d.sidx, n0, 4, n_limit);
        g_f32_ptr o = (g_f32_ptr)d.out + static_cast<int64_t>(n0) * fs + (WIDE ? 2 * cu.cur_pair : 0u);
        const v4f lo = v4f{acc0.x, acc1.x, acc0.y, acc1.y};
        const v4f hi = v4f{acc0.z, acc1.z, acc0.w, acc1.w};
        if (!(dbg & 16)) {
            const bool full = j0 + 4 <= g.b && n0 >= 0 && n0 + 4 <= n_limit;
            bool combined = false;
            if constexpr (WIDE) {
                // the odd pair of a block whose even pair is pending: four channels of a frame side by side, 16-byte stores
                if ((cu.cur_pair & 1u) && pend && pend_o + 2 == o && __all(full) && !phantom(cu.cur_pair)) {
                    typedef v4f __attribute__((address_space(1), aligned(4)))* g_f4a4_ptr;
                    typedef v4f __attribute__((address_space(1), aligned(8)))* g_f4a8e_ptr;
                    typedef typename std::conditional<kOdd, g_f4a4_ptr, g_f4a8e_ptr>::type g_f4a8_ptr;
                    *((g_f4a8_ptr)pend_o) = v4f{pend_lo.x, pend_lo.y, lo.x, lo.y};
                    *((g_f4a8_ptr)(pend_o + fs)) = v4f{pend_lo.z, pend_lo.w, lo.z, lo.w};
                    *((g_f4a8_ptr)(pend_o + 2 * fs)) = v4f{pend_hi.x, pend_hi.y, hi.x, hi.y};
                    *((g_f4a8_ptr)(pend_o + 3 * fs)) = v4f{pend_hi.z, pend_hi.w, hi.z, hi.w};
                    pend = false;
                    combined = true;
                } else {
                    flush_pending();   // (an even pair left without its partner: the workgroup's range ended between them)
                }
            }
            if (combined) {
            } else if (__builtin_expect(__all(full), 1)) {   // the whole wave inside the launch: stored inside the next item's stream
                pend_lo = lo;
                pend_hi = hi;
                pend_o = o;
                pend_ph = phantom(cu.cur_pair);
                pend = true;
            } else if (full) {
                store_frames(o, lo, hi, phantom(cu.cur_pair));
            } else {
                const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int32_t n = n0 + r;
                    if (j0 + r < g.b && n >= 0 && n < n_limit) {
                        if (mono) o[r] = v[2 * r];
                        else if (phantom(cu.cur_pair)) o[fs * r] = v[2 * r];
                        else if constexpr (kOdd) *((g_f2u_ptr)(o + fs * r)) = v2f{v[2 * r], v[2 * r + 1]};
                        else *((g_f2_ptr)(o + fs * r)) = v2f{v[2 * r], v[2 * r + 1]};
                    }
                }
            }
        }
        wt.event(8);
        if (++slot == g.slots) {
            slot = 0;
            ++use;
        }
    }
    flush_pending();
    wt.flush();
}

template <int NK, int PLANES, bool DIAG, int WIDE, int ROUNDS = 1, int BITS = 0>
__global__ __launch_bounds__(1024) void fir_split_kernel(const FirStreamDesc* __restrict__ descs, const SplitArgs g) {
    static_assert(BITS == 0 || (WIDE == 0 && PLANES == 2 && !DIAG), "PCM input: the two-channel fp16 build");
    fir_split_body<NK, PLANES, DIAG, WIDE, ROUNDS, BITS>(descs, g, blockIdx.x, gridDim.x);
}
template <int NK, int PLANES, int WIDE, int ROUNDS>
__global__ __launch_bounds__(1024) void fir_split_multi_kernel(const SplitMulti m) {
    uint32_t begin;
    const uint32_t j = __builtin_amdgcn_readfirstlane(split_job_of(m.wg_end, m.n_jobs, blockIdx.x, begin));
    begin = __builtin_amdgcn_readfirstlane(begin);
    fir_split_body<NK, PLANES, false, WIDE, ROUNDS>(m.descs[j], m.g[j], blockIdx.x - begin, m.wg_end[j] - begin);
}

// The builds a run of config 4's six rate pairs needs -- one round with a five-step window (44.1 <-> 48 kHz, 44.1 / 48 -> 96 kHz),
// two rounds with five (96 -> 48 kHz) or six steps (96 -> 44.1 kHz) -- in ONE kernel: a workgroup finds its job as in
// fir_split_multi_kernel and runs the body of the job's build.  One launch per run instead of three: the two-round builds'
// jobs are a quarter of the batch, alone they do not fill the chip (a 128-stream shard: 99 + 38 + 34 us one after the
// other, each with its own ramp-up and tail), side by side they end with the one-round jobs.
constexpr uint32_t kSplitBuild5x1 = 0, kSplitBuild5x2 = 1, kSplitBuild6x2 = 2;
__global__ __launch_bounds__(1024) void fir_split_all_kernel(const SplitMulti m) {
    uint32_t begin;
    const uint32_t j = __builtin_amdgcn_readfirstlane(split_job_of(m.wg_end, m.n_jobs, blockIdx.x, begin));
    begin = __builtin_amdgcn_readfirstlane(begin);
    const uint32_t build = __builtin_amdgcn_readfirstlane(m.build[j]);
    if (build == kSplitBuild5x1) fir_split_body<5, 2, false, 0, 1>(m.descs[j], m.g[j], blockIdx.x - begin, m.wg_end[j] - begin);
    else if (build == kSplitBuild5x2) fir_split_body<5, 2, false, 0, 2>(m.descs[j], m.g[j], blockIdx.x - begin, m.wg_end[j] - begin);
    else fir_split_body<6, 2, false, 0, 2>(m.descs[j], m.g[j], blockIdx.x - begin, m.wg_end[j] - begin);
}

inline uint32_t split_class_offset(uint32_t a, uint32_t b, uint32_t j) {
    return static_cast<uint32_t>((static_cast<uint64_t>(j) * a) / b);
}

}  // namespace

// Geometry of the split kernel for num/den, or !ok.  A super period of a = r num input frames and b = r den outputs:
// r = 1 for 16 .. 320 classes; a ratio with a power-of-two denominator below 16 (48 <-> 96 kHz: exact in f64, so no
// output ever takes the row-1023 variant and every class of the super period is an ordinary one) takes the largest r
// with a, b <= 320.  Up to ten class tiles per tile group (one tile per consumer wave), up to two groups; periods of up
// to 160 frames in one round of lane tasks, up to 320 in two (two-channel streams); window of <= 160 taps (192 with
// two rounds); two to four images within the LDS.
// RSMP_FIR_SPLIT_PLANES = 3 selects the three-plane bf16 split (every f32 operand exactly), default 2: two
// fp16 planes per operand, three matrix products instead of six.
static uint32_t split_planes_knob() {
    static const uint32_t v = [] {
        const char* e = rsmp::knob("RSMP_FIR_SPLIT_PLANES");
        return e && atoi(e) == 3 ? 3u : 2u;
    }();
    return v;
}

PeriodicGeometry split_geometry(uint64_t num, uint64_t den, uint32_t taps, uint32_t channels) {
    PeriodicGeometry g;
    // (three planes: the two-channel kernel only -- the channel-pair, one-channel and odd-count builds keep two)
    const uint32_t planes = channels == 2 ? split_planes_knob() : 2u;
    const uint32_t kRowBytes = row_bytes(static_cast<int>(planes));
    // two channels, or (RSMP_FIR_SPLIT_WIDE=0 turns it off) an even number up to 16 taken as channel pairs, two pairs per
    // 16-byte load (6, 10, 14 channels: the last pair alone -- its load reaches 8 bytes into the next frame)
    static const bool wide_ok = [] { const char* e = rsmp::knob("RSMP_FIR_SPLIT_WIDE"); return !e || atoi(e) != 0; }();
    static const bool long_ok = [] { const char* e = rsmp::knob("RSMP_FIR_SPLIT_LONG"); return !e || atoi(e) != 0; }();   // 0: round 2's geometries only
    if (channels != 2 && (channels > 16 || !wide_ok)) return g;
    constexpr uint32_t kMaxAB = 320;
    if (num == 0 || den == 0 || num > kMaxAB || den > kMaxAB) return g;
    uint32_t r = 1;
    if (den < 16) {
        if ((den & (den - 1)) != 0) return g;   // (the wrap variant exists for class 0 only: exact ratios need none)
        r = static_cast<uint32_t>(std::min(kMaxAB / num, kMaxAB / den));
        r -= r % (16 / static_cast<uint32_t>(den));   // whole tiles
        if (r == 0) return g;
    }
    const uint32_t a = static_cast<uint32_t>(num) * r, b = static_cast<uint32_t>(den) * r;
    if (b < 16) return g;
    const uint32_t n_tiles = (b + 15) / 16;
    const uint32_t groups = (n_tiles + kConsumers - 1) / kConsumers;
    const uint32_t rounds = 4 * ((a + 1) / 2) > 64 * kStagers ? 2u : 1u;
    if (!long_ok && (groups > 1 || rounds > 1 || r > 1)) return g;
    if (groups > 2 || (rounds == 2 && (channels % 2 != 0 || planes != 2))) return g;   // (two rounds: channel pairs, fp16 planes)
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
    if (kpad / 32 < 1 || kpad / 32 > (rounds == 2 ? 6u : 5u) || taps > 16 * kWrapTaps) return g;
    if (rounds == 2 && kpad / 32 < 5) return g;   // (the two-round kernels exist for windows of 160 and 192 taps: 128-tap filters)
    // Rows a plane really needs: the last tile's window ends at ob_max + taps + shift.  The MFMA steps read
    // on to ob_max + kpad with zero coefficients -- into the rows that follow in LDS (the next plane, the next
    // image, the pad after the last image: always finite values, the whole LDS is zeroed at the start).
    const uint32_t rows = ob_max + taps + shift;
    if (4 * ((a + 1) / 2) > 64 * kStagers * rounds || rows < a || rows > 2 * a) return g;   // (rows beyond a repeat the next period)
    const uint32_t pad = (kpad - (taps + shift)) * kRowBytes;
    uint32_t slots = (kLdsLimit - kImageBase - pad) / (rows * kRowBytes);   // ring of images: slack between producers and consumers
    if (slots > 4) slots = 4;
    if (slots < 2) return g;
    const uint32_t lds = kImageBase + slots * rows * kRowBytes + pad;
    g.a = a;
    g.b = b;
    g.den = static_cast<uint32_t>(den);   // the true period of the phase pattern (b = r den)
    g.taps = taps;
    g.row_len = kpad;
    g.n_tiles = n_tiles;
    g.n_units = n_tiles;
    g.cg = channels == 1 ? 1 : (channels % 2 ? 3 : 2);   // (1: one channel, a pair with a phantom second channel; 3: odd count, the last pair likewise)
    g.lp = (channels + 1) / 2;      // channel pairs of a frame (an item of the launch = one pair of a block)
    g.pw = 16;
    g.row_stride = rows;       // rows of an image (frames of a period + window reach)
    g.waves = kWaves;
    g.producers = kProducers;
    g.images = slots;
    g.mfma = 3;
    g.planes = planes;
    g.groups = groups;
    g.rounds = rounds;
    g.lds_bytes = lds;
    g.inline_wraps = true;
    g.ok = true;
    return g;
}

// Class-table image for the split kernel: [tile][k step][plane][lane][8 x 16 bit]; lane (class m =
// lane & 15, k group = lane >> 4) element j holds window position 32 s + 16 (j >> 2) + 4 (lane >> 4) +
// (j & 3) -- the order in which the transposed LDS reads deliver the frames.  Three planes: bf16 by
// truncation (c == p1 + p2 + p3 exactly); two planes: fp16, round to nearest, of 2^13 c.
void split_store_class(std::vector<float>& coef, const PeriodicGeometry& g, uint32_t tile, uint32_t m,
                       uint32_t shift, const std::vector<float>& mixed) {
    const uint32_t nk = g.row_len / 32;
    const uint32_t planes = g.planes;
    uint32_t* words = reinterpret_cast<uint32_t*>(coef.data());
    for (uint32_t s = 0; s < nk; ++s)
        for (uint32_t grp = 0; grp < 4; ++grp)
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t pos = 32 * s + 16 * (j >> 2) + 4 * grp + (j & 3);
                float c = 0.f;
                if (pos >= shift && pos - shift < g.taps) c = mixed[pos - shift];
                uint32_t p[3] = {0, 0, 0};
                if (planes == 3) {
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
                } else {
                    const float sc = c * kCScale;
                    const _Float16 h1 = static_cast<_Float16>(sc);
                    const _Float16 h2 = static_cast<_Float16>(sc - static_cast<float>(h1));
                    uint16_t b1, b2;
                    std::memcpy(&b1, &h1, 2);
                    std::memcpy(&b2, &h2, 2);
                    p[0] = b1;
                    p[1] = b2;
                }
                const uint32_t lane = 16 * grp + m;
                for (uint32_t pl = 0; pl < planes; ++pl) {
                    const size_t dword = ((((static_cast<size_t>(tile) * nk + s) * planes + pl) * 64 + lane) * 4) + (j >> 1);
                    const uint32_t sh = (j & 1) * 16;
                    words[dword] = (words[dword] & ~(0xFFFFu << sh)) | (p[pl] << sh);
                }
            }
}

size_t split_table_floats(const PeriodicGeometry& g) {
    return static_cast<size_t>(g.n_tiles) * (g.row_len / 32) * g.planes * 64 * 4;
}

// The item tables' workspaces, one per (device, stream) that has launched the split kernel.
namespace {
struct ItemsSlot { uint32_t* ptr = nullptr; size_t cap = 0; uint64_t key = 0; uint32_t items = 0; };
std::mutex& items_ws_mu() { static std::mutex* m = new std::mutex; return *m; }
std::map<std::pair<int, hipStream_t>, ItemsSlot>& items_ws() {
    static auto* w = new std::map<std::pair<int, hipStream_t>, ItemsSlot>;   // (leaked on purpose: process lifetime)
    return *w;
}
}  // namespace

// A handle or batch that destroys a stream of its own gives the stream's workspace back first (the work on the stream is
// complete by then): without this every create / destroy cycle left a device allocation and a dead key behind.
void split_release_stream(int device, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(items_ws_mu());
    auto it = items_ws().find({device, stream});
    if (it == items_ws().end()) return;
    if (it->second.ptr) (void)hipFree(it->second.ptr);
    items_ws().erase(it);
}

namespace {
uint32_t split_debug_knob() {
    static const uint32_t debug = [] {
        const char* e = rsmp::knob("RSMP_FIR_DEBUG");
        return e ? static_cast<uint32_t>(atoi(e)) : 0u;
    }();
    return debug;
}
// The kernel's arguments for the streams of one geometry (everything but the item table's address).
SplitArgs make_split_args(const PeriodicGeometry& geo, uint32_t n_streams, uint32_t max_blocks, bool fuse_tail, const NfArgs& nf) {
    const uint32_t pairs = geo.lp;
    const bool wide = pairs > 1 || geo.cg == 1;
    static const bool quad_major = [] { const char* e = rsmp::knob("RSMP_FIR_SPLIT_QUADS"); return !e || atoi(e) != 0; }();
    const uint32_t quads = quad_major && geo.cg == 2 && pairs >= 4 && pairs % 2 == 0 ? pairs / 2 : 1u;   // 8, 12, 16 channels
    const uint32_t qpairs = pairs / quads;
    const uint32_t groups = geo.groups ? geo.groups : 1u, per_group = max_blocks * n_streams * qpairs;
    return SplitArgs{geo.a, geo.b, geo.taps, geo.n_tiles, geo.row_stride, geo.images, geo.lds_bytes, max_blocks, per_group * groups * quads,
                     split_debug_knob(), n_streams, fuse_tail ? 1u : 0u, geo.cg == 1 ? 1u : (geo.cg == 3 ? 2 * pairs - 1 : 2 * pairs), pairs,
                     groups, per_group, quads, qpairs, wide ? 1u : 0u,
                     // (b = r den with r > 1: a super period of an exact ratio.  WIDE builds only: the two-channel tile-group
                     // kernel got SLOWER without the fetches -- 2 ch 48 -> 96 kHz 0.59 -> 0.66 ms)
                     wide && geo.b != geo.den ? 1u : 0u, nullptr, nullptr, nf};
}
// The item-table workspace of (device, stream), at least `need` bytes.  `key` / `items`: see launch_fir_split; *have_table =
// the workspace already holds the table with that key.
// Held from taking a stream's item table to the launch that reads it: two host threads that enqueue on one stream would
// otherwise interleave table launch and kernel launch and read each other's table (enqueueing takes microseconds).
std::mutex& items_launch_mu() { static std::mutex* m = new std::mutex; return *m; }
hipError_t items_workspace(int device, hipStream_t stream, size_t need, uint64_t key, uint32_t items, uint32_t** d_items, bool* have_table) {
    std::lock_guard<std::mutex> lock(items_ws_mu());
    ItemsSlot& slot = items_ws()[{device, stream}];
    if (slot.cap < need) {
        if (slot.ptr) {   // (a launch on this stream may still read the old table)
            if (hipError_t e = hipStreamSynchronize(stream); e != hipSuccess) return e;
            (void)hipFree(slot.ptr);
            slot = ItemsSlot{};
        }
        const size_t cap = need + need / 2 + 4096;
        if (hipError_t e = hipMalloc(&slot.ptr, cap); e != hipSuccess) return e;
        slot.cap = cap;
    }
    *d_items = slot.ptr;
    *have_table = key != 0 && slot.key == key && slot.items == items;
    slot.key = key;
    slot.items = items;
    return hipSuccess;
}
uint32_t device_cus(int device) {
    static std::mutex mu;
    static std::map<int, uint32_t> count;
    std::lock_guard<std::mutex> lock(mu);
    uint32_t& c = count[device];
    if (c == 0) {
        int v = 0;
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device);
        c = static_cast<uint32_t>(v > 0 ? v : 256);
    }
    return c;
}
}  // namespace

hipError_t launch_fir_split(const FirStreamDesc* d_descs, uint32_t n_streams, const PeriodicGeometry& geo,
                            uint32_t max_blocks, uint32_t cus, bool fuse_tail, const NfArgs& nf, hipStream_t stream,
                            uint64_t items_key, uint32_t pcm_bits) {
    const uint32_t debug = split_debug_knob();
    const uint32_t pairs = geo.lp;
    const bool wide = pairs > 1 || geo.cg == 1;
    SplitArgs args = make_split_args(geo, n_streams, max_blocks, fuse_tail, nf);
    const uint32_t groups = args.groups;
    static const char* wtrace_path = rsmp::knob("RSMP_FIR_WTRACE");
    const bool diag = debug != 0 || wtrace_path != nullptr;
#define RSMP_SPLIT_FNS(P, D, W)                                                                              \
    {reinterpret_cast<const void*>(fir_split_kernel<1, P, D, W>), reinterpret_cast<const void*>(fir_split_kernel<2, P, D, W>), \
     reinterpret_cast<const void*>(fir_split_kernel<3, P, D, W>), reinterpret_cast<const void*>(fir_split_kernel<4, P, D, W>), \
     reinterpret_cast<const void*>(fir_split_kernel<5, P, D, W>)}
    static const void* const fns_all[2][2][5] = {{RSMP_SPLIT_FNS(2, false, 0), RSMP_SPLIT_FNS(2, true, 0)},
                                                 {RSMP_SPLIT_FNS(3, false, 0), RSMP_SPLIT_FNS(3, true, 0)}};
    static const void* const fns_wide[1][5] = {RSMP_SPLIT_FNS(2, false, 1)};   // (no diagnostic build; two planes only)
    static const void* const fns_mono[1][5] = {RSMP_SPLIT_FNS(2, false, 2)};
    static const void* const fns_odd[1][5] = {RSMP_SPLIT_FNS(2, false, 3)};
#undef RSMP_SPLIT_FNS
    const bool one_channel = geo.cg == 1;
    const bool odd_count = geo.cg == 3;
    // two rounds of lane tasks (periods of 161 .. 320 frames): two-channel fp16 kernel, windows of 5 or 6 steps
    static const void* const fns_long[4][6] = {{nullptr, nullptr, nullptr, nullptr,
                                                reinterpret_cast<const void*>(fir_split_kernel<5, 2, false, 0, 2>),
                                                reinterpret_cast<const void*>(fir_split_kernel<6, 2, false, 0, 2>)},
                                               {nullptr, nullptr, nullptr, nullptr,
                                                reinterpret_cast<const void*>(fir_split_kernel<5, 2, false, 1, 2>),
                                                reinterpret_cast<const void*>(fir_split_kernel<6, 2, false, 1, 2>)},
                                               // (diagnostic builds: config 5's geometry, the 192-tap window; config 4's two-channel streams)
                                               {nullptr, nullptr, nullptr, nullptr,
                                                reinterpret_cast<const void*>(fir_split_kernel<5, 2, true, 0, 2>),
                                                reinterpret_cast<const void*>(fir_split_kernel<6, 2, true, 0, 2>)},
                                               {nullptr, nullptr, nullptr, nullptr, nullptr,
                                                reinterpret_cast<const void*>(fir_split_kernel<6, 2, true, 1, 2>)}};
    const bool two_rounds = geo.rounds == 2;
    const bool diag_long = two_rounds && diag && (geo.row_len / 32 == 6 || (geo.row_len / 32 == 5 && !wide));
    const void* const* fns = two_rounds  ? fns_long[(wide ? 1 : 0) + (diag_long ? 2 : 0)]
                             : one_channel ? fns_mono[0]
                             : odd_count ? fns_odd[0]
                             : wide      ? fns_wide[0]
                                         : fns_all[geo.planes == 3 ? 1 : 0][diag ? 1 : 0];
    const uint32_t nk = geo.row_len / 32;
    if (nk < 1 || nk > (two_rounds ? 6u : 5u) || fns[nk - 1] == nullptr) return hipErrorInvalidValue;
    // PCM input (FirStreamDesc::in_bits): the two-channel fp16 builds of the 128-tap windows -- 160 taps in one round
    // (44.1 <-> 48 kHz) or two, 192 taps in two rounds (96 -> 44.1 kHz) -- exist for the three widths; hipErrorNotSupported
    // for any other geometry (the caller converts with rsmp_pcm_to_stereo_f32_device first).
    const void* fn = fns[nk - 1];
    if (pcm_bits != 0) {
#define RSMP_PCM_FNS(B)                                                                                                   \
    {reinterpret_cast<const void*>(fir_split_kernel<5, 2, false, 0, 1, B>), reinterpret_cast<const void*>(fir_split_kernel<5, 2, false, 0, 2, B>), \
     reinterpret_cast<const void*>(fir_split_kernel<6, 2, false, 0, 2, B>)}
        static const void* const fns_pcm[3][3] = {RSMP_PCM_FNS(16), RSMP_PCM_FNS(24), RSMP_PCM_FNS(32)};
#undef RSMP_PCM_FNS
        const int v = !two_rounds && nk == 5 ? 0 : two_rounds && nk == 5 ? 1 : two_rounds && nk == 6 ? 2 : -1;
        if (wide || one_channel || odd_count || geo.planes != 2 || diag || v < 0 || (pcm_bits != 16 && pcm_bits != 24 && pcm_bits != 32))
            return hipErrorNotSupported;
        fn = fns_pcm[pcm_bits == 16 ? 0 : pcm_bits == 24 ? 1 : 2][v];
    }
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    static std::mutex mu;
    static std::map<std::pair<int, uint32_t>, bool> granted;
    {
        std::lock_guard<std::mutex> lock(mu);
        bool& have = granted[{device, ((((nk * 8 + geo.planes) * 2 + ((diag && !wide) || diag_long ? 1u : 0u)) * 4 + (one_channel ? 2u : odd_count ? 3u : wide ? 1u : 0u)) * 2 + (two_rounds ? 1u : 0u)) * 64 + pcm_bits}];
        if (!have) {
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
            if (e != hipSuccess) return e;
            have = true;
        }
    }
    const dim3 grid(args.total_items < cus ? args.total_items : cus);
    static const bool verbose = rsmp::knob("RSMP_FIR_VERBOSE") != nullptr;
    if (verbose)
        fprintf(stderr, "[rsmp] split launch: a=%u b=%u window=%u tiles=%u groups=%u rounds=%u rows=%u slots=%u lds=%u items=%u grid=%u\n",
                geo.a, geo.b, geo.row_len, geo.n_tiles, groups, geo.rounds, geo.row_stride, geo.images, geo.lds_bytes, args.total_items, grid.x);
    static unsigned long long* d_wtrace = nullptr;
    const size_t wtrace_words = static_cast<size_t>(grid.x) * 16 * kWtraceSlots;
    if (wtrace_path) {
        if (d_wtrace) (void)hipFree(d_wtrace);
        if (hipMalloc(&d_wtrace, wtrace_words * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d_wtrace, 0, wtrace_words * 8, stream);
        args.wtrace = d_wtrace;
    }
    std::unique_lock<std::mutex> launch_lock(items_launch_mu(), std::defer_lock);
    // the item table: a launch of its own in front (one thread per item), in a workspace kept per stream
    {
        // The table is a pure function of the streams' counters and the geometry: a launch whose key (the caller's hash of
        // exactly those, 0 = none) equals the key of the table the workspace holds finds it there -- a service resampling
        // batch after batch of equally long files, the bench's step -- and skips the table launch (5 us in front of the kernel).
        const size_t need = static_cast<size_t>(args.total_items) * kItemWords * sizeof(uint32_t);
        uint32_t* d_items = nullptr;
        bool have_table = false;
        launch_lock.lock();
        if ((e = items_workspace(device, stream, need, items_key, args.total_items, &d_items, &have_table)) != hipSuccess) return e;
        args.items = d_items;
        if (!have_table) {
            split_items_kernel<<<dim3((args.total_items + 255) / 256), dim3(256), 0, stream>>>(d_descs, args, d_items);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
    }
    void* kargs[2] = {&d_descs, &args};
    e = hipLaunchKernel(fn, grid, dim3(kWaves * 64), kargs,
                        geo.lds_bytes + (wtrace_path ? 16 * kWtraceSlots * 8 : 0), stream);
    if (e != hipSuccess) return e;
    if (wtrace_path) {   // one line per wave: block wave event...
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h(wtrace_words);
        (void)hipMemcpy(h.data(), d_wtrace, wtrace_words * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(wtrace_path, "w")) {
            for (size_t w = 0; w < wtrace_words / kWtraceSlots; ++w) {
                fprintf(f, "%zu %zu", w / 16, w % 16);
                for (uint32_t i = 0; i < kWtraceSlots; ++i) fprintf(f, " %llu", h[w * kWtraceSlots + i]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    return hipGetLastError();
}

size_t fir_split_multi_item_words(const SplitJob* jobs, size_t n_jobs) {
    size_t items = 0;   // (every job counted, covered by the multi-job builds or not: an upper bound)
    for (size_t j = 0; j < n_jobs; ++j)
        if (jobs[j].n_streams != 0 && jobs[j].max_blocks != 0)
            items += make_split_args(*jobs[j].geo, jobs[j].n_streams, jobs[j].max_blocks, false, jobs[j].nf).total_items;
    return items * kItemWords;
}

hipError_t launch_fir_split_multi(const SplitJob* jobs, size_t n_jobs, hipStream_t stream, uint32_t reserve_cus, const uint32_t* items_prebuilt,
                                  hipStream_t items_only) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    // (reserve_cus: compute units the launches leave to somebody else -- a workgroup of this kernel fills a CU's registers
    // and LDS, nothing runs BESIDE it on that CU: the run planner of a small lock-step batch, fir_lockstep_api.cpp)
    const uint32_t cus_all = device_cus(device);
    const uint32_t cus = reserve_cus < cus_all / 2 ? cus_all - reserve_cus : cus_all;
    static const bool diag = split_debug_knob() != 0 || rsmp::knob("RSMP_FIR_WTRACE") != nullptr;
    static const bool multi_on = [] { const char* v = rsmp::knob("RSMP_FIR_SPLIT_MULTI"); return !v || atoi(v) != 0; }();
    // the multi-job builds: two channels, two fp16 planes; windows of 1 .. 5 steps in one round, 5 or 6 in two
    static const void* const fns_multi[2][6] = {
        {reinterpret_cast<const void*>(fir_split_multi_kernel<1, 2, 0, 1>), reinterpret_cast<const void*>(fir_split_multi_kernel<2, 2, 0, 1>),
         reinterpret_cast<const void*>(fir_split_multi_kernel<3, 2, 0, 1>), reinterpret_cast<const void*>(fir_split_multi_kernel<4, 2, 0, 1>),
         reinterpret_cast<const void*>(fir_split_multi_kernel<5, 2, 0, 1>), nullptr},
        {nullptr, nullptr, nullptr, nullptr,
         reinterpret_cast<const void*>(fir_split_multi_kernel<5, 2, 0, 2>), reinterpret_cast<const void*>(fir_split_multi_kernel<6, 2, 0, 2>)}};
    auto multi_fn = [&](const PeriodicGeometry& g) -> const void* {
        if (diag || !multi_on || g.mfma != 3 || g.cg != 2 || g.lp != 1 || g.planes != 2) return nullptr;
        const uint32_t nk = g.row_len / 32;
        if (nk < 1 || nk > 6 || (g.rounds != 1 && g.rounds != 2)) return nullptr;
        return fns_multi[g.rounds - 1][nk - 1];
    };
    // (RSMP_FIR_SPLIT_ALL=0, debug: one launch per kernel build as in round 5)
    static const bool all_on = [] { const char* v = rsmp::knob("RSMP_FIR_SPLIT_ALL"); return !v || atoi(v) != 0; }();
    const void* const fn_all = reinterpret_cast<const void*>(fir_split_all_kernel);
    auto all_build = [&](const PeriodicGeometry& g, uint32_t* build) -> bool {
        const uint32_t nk = g.row_len / 32;
        if (g.rounds == 1 && nk == 5) *build = kSplitBuild5x1;
        else if (g.rounds == 2 && nk == 5) *build = kSplitBuild5x2;
        else if (g.rounds == 2 && nk == 6) *build = kSplitBuild6x2;
        else return false;
        return true;
    };
    // jobs the multi-job builds do not cover: a launch each, as before
    // (items_only: nothing but the covered jobs' item tables, written to items_prebuilt by launches on that stream -- what a later
    // call with the same jobs and items_prebuilt then does not launch in front of its kernels)
    std::vector<size_t> covered;
    for (size_t j = 0; j < n_jobs; ++j) {
        const SplitJob& job = jobs[j];
        if (job.n_streams == 0 || job.max_blocks == 0) continue;
        if (multi_fn(*job.geo)) covered.push_back(j);
        else if (items_only) continue;
        else if ((e = launch_fir_split(job.d_descs, job.n_streams, *job.geo, job.max_blocks, cus, false, job.nf, stream, 0)) != hipSuccess) return e;
    }
    size_t prebuilt_off = 0;   // items of the batches of jobs before this one
    for (size_t c0 = 0; c0 < covered.size(); c0 += kMaxSplitJobs) {
        const uint32_t n = static_cast<uint32_t>(std::min<size_t>(kMaxSplitJobs, covered.size() - c0));
        SplitArgs args[kMaxSplitJobs];
        const void* fn[kMaxSplitJobs];
        uint32_t build[kMaxSplitJobs] = {};
        size_t total_items = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const SplitJob& job = jobs[covered[c0 + i]];
            args[i] = make_split_args(*job.geo, job.n_streams, job.max_blocks, false, job.nf);
            fn[i] = multi_fn(*job.geo);
            total_items += args[i].total_items;
        }
        // jobs of more than one build, all of them builds of fir_split_all_kernel: one launch for the lot
        {
            bool every = all_on && n > 1, mixed = false;
            for (uint32_t i = 0; i < n && every; ++i) {
                every = all_build(*jobs[covered[c0 + i]].geo, &build[i]);
                mixed = mixed || fn[i] != fn[0];
            }
            if (every && mixed)
                for (uint32_t i = 0; i < n; ++i) fn[i] = fn_all;
        }
        // one item table after the other in the stream's workspace, built by one launch
        uint32_t* d_items = nullptr;
        bool have_table = false;
        std::lock_guard<std::mutex> launch_lock(items_launch_mu());   // (to the end of this batch of jobs: table launch + kernel launches)
        if (items_prebuilt) d_items = const_cast<uint32_t*>(items_prebuilt) + prebuilt_off * kItemWords;
        else if ((e = items_workspace(device, stream, total_items * kItemWords * sizeof(uint32_t), 0, 0, &d_items, &have_table)) != hipSuccess) return e;
        prebuilt_off += total_items;
        {
            SplitMulti m{};
            size_t off = 0;
            uint32_t blocks = 0;
            for (uint32_t i = 0; i < n; ++i) {
                args[i].items = d_items + off * kItemWords;
                off += args[i].total_items;
                m.g[i] = args[i];
                m.descs[i] = jobs[covered[c0 + i]].d_descs;
                blocks += (args[i].total_items + 255) / 256;
                m.ib_end[i] = blocks;
            }
            m.n_jobs = n;
            if (items_only || !items_prebuilt) {
                hipLaunchKernelGGL(split_items_multi_kernel, dim3(blocks), dim3(256), 0, items_only ? items_only : stream, m);
                if ((e = hipGetLastError()) != hipSuccess) return e;
            }
        }
        if (items_only) continue;
        // one launch per kernel build among the jobs; its workgroups dealt in proportion to the jobs' staging work
        // (items x frames of a period: the stagers bound this kernel)
        bool done[kMaxSplitJobs] = {};
        for (uint32_t i0 = 0; i0 < n; ++i0) {
            if (done[i0]) continue;
            SplitMulti m{};
            uint32_t idx[kMaxSplitJobs], nb = 0, lds = 0;
            double weight[kMaxSplitJobs], weight_sum = 0.0;
            uint64_t items_sum = 0;
            for (uint32_t i = i0; i < n; ++i) {
                if (done[i] || fn[i] != fn[i0]) continue;
                done[i] = true;
                idx[nb] = i;
                // (items x frames of a period.  Round 6 tried the items' costs measured job by job instead -- 2.46 .. 4.39 us per item,
                // profiles/r05/channels_bench.txt, by which this deal gives the two-round jobs 30 % too many workgroups: config 4 got
                // 2 % SLOWER, 3.48 against 3.41 us per step in one lease; inside the shared launch the jobs do not cost what they cost alone)
                weight[nb] = static_cast<double>(args[i].total_items) * args[i].a;
                weight_sum += weight[nb];
                items_sum += args[i].total_items;
                lds = std::max(lds, args[i].lds_bytes);
                ++nb;
            }
            const uint32_t wgs = static_cast<uint32_t>(std::min<uint64_t>(cus, items_sum));
            if (wgs < nb) {   // (fewer items than jobs cannot be: every job has at least one)
                return hipErrorInvalidValue;
            }
            // at least one workgroup per job, no more than it has items; the rest by weight (largest remainder)
            uint32_t share[kMaxSplitJobs], given = 0;
            double frac[kMaxSplitJobs];
            for (uint32_t b = 0; b < nb; ++b) {
                const double ideal = weight[b] / weight_sum * wgs;
                uint32_t w = static_cast<uint32_t>(ideal);
                w = std::max<uint32_t>(1u, std::min<uint32_t>(w, args[idx[b]].total_items));
                share[b] = w;
                frac[b] = ideal - w;
                given += w;
            }
            // A job with two tile groups: its item order makes workgroups w and w + n / 2 stage the same frames at about the
            // same time (tiles 0 .. 9 / 10 .. 19 of the same blocks), and the second of them finds the frames in L2 only if
            // both sit on one XCD -- workgroups go round the eight XCDs, so n / 2 must be a multiple of 8.  (Without this the
            // two such pairs of config 4 read their input twice from HBM: 0.37 GB of 2.35 per run.)
            bool two_groups[kMaxSplitJobs];
            for (uint32_t b = 0; b < nb; ++b) {
                two_groups[b] = args[idx[b]].groups > 1;
                if (two_groups[b] && wgs >= 16) {
                    const uint32_t r16 = std::max<uint32_t>(16u, (share[b] + 8u) / 16u * 16u);
                    given = given - share[b] + r16;
                    frac[b] += static_cast<double>(share[b]) - static_cast<double>(r16);
                    share[b] = r16;
                }
            }
            auto step_of = [&](uint32_t b) { return two_groups[b] && wgs >= 16 ? 16u : 1u; };
            while (given > wgs) {   // (over: take from the job with the most to spare, a plain one if there is one)
                int big = -1;
                for (uint32_t b = 0; b < nb; ++b)
                    if (share[b] > step_of(b) && !(two_groups[b] && wgs >= 16) && (big < 0 || share[b] > share[big])) big = static_cast<int>(b);
                if (big < 0)
                    for (uint32_t b = 0; b < nb; ++b)
                        if (share[b] > step_of(b) && (big < 0 || share[b] > share[big])) big = static_cast<int>(b);
                if (big < 0) break;
                const uint32_t st = std::min(step_of(big), share[big] - 1);
                share[big] -= st;
                frac[big] += st;
                given -= st;
            }
            while (given < wgs) {
                int best = -1;
                for (uint32_t b = 0; b < nb; ++b)
                    if (step_of(b) == 1u && share[b] < args[idx[b]].total_items && (best < 0 || frac[b] > frac[best])) best = static_cast<int>(b);
                if (best < 0) break;   // (only jobs that move in sixteens are left: the odd workgroups stay away)
                ++share[best];
                frac[best] -= 1.0;
                ++given;
            }
            uint32_t end = 0;
            for (uint32_t b = 0; b < nb; ++b) {
                m.g[b] = args[idx[b]];
                m.descs[b] = jobs[covered[c0 + idx[b]]].d_descs;
                m.build[b] = build[idx[b]];
                end += share[b];
                m.wg_end[b] = end;
            }
            m.n_jobs = nb;
            static std::mutex mu;
            static std::map<std::pair<int, const void*>, bool> granted;
            {
                std::lock_guard<std::mutex> lock(mu);
                bool& have = granted[{device, fn[i0]}];
                if (!have) {
                    if ((e = hipFuncSetAttribute(fn[i0], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit)) != hipSuccess) return e;
                    have = true;
                }
            }
            static const bool verbose = rsmp::knob("RSMP_FIR_VERBOSE") != nullptr;
            if (verbose)
                for (uint32_t b = 0; b < nb; ++b)
                    fprintf(stderr, "[rsmp] split multi launch: job %u of %u: a=%u b=%u rounds=%u items=%u workgroups=%u\n", b, nb, m.g[b].a, m.g[b].b,
                            jobs[covered[c0 + idx[b]]].geo->rounds, m.g[b].total_items, share[b]);
            void* kargs[1] = {&m};
            if ((e = hipLaunchKernel(fn[i0], dim3(end), dim3(kWaves * 64), kargs, lds, stream)) != hipSuccess) return e;
        }
    }
    return hipGetLastError();
}

}  // namespace rsmp
