// fir_lockstep.hip -- one kernel launch = one lock-step step of a batch of ResamplerFir streams
// (see fir_lockstep.h).  Per stream and step it is exactly one reference resample() call
// (src/resampler_fir.rs:509-621): same frames accepted, same outputs, same frames retired.
//
// A workgroup (8 waves) owns a few streams of one rate pair:
//   A  wave 0, one lane per stream: the step's plan -- from the stream's plan record (written one step ahead
//      by wave 7, below) or, when the record is stale, the reference's control flow in line
//      (fir_mirror_core.h) -> n_out, frames consumed, the outputs that take the row-1023 variant, the exact
//      position runs -- and the column table.  Waves 1..6 stage the frames meanwhile: two-channel streams as
//      a transposed fp16 image built straight from HBM (the split variant: operands cut into two fp16 planes,
//      fir_split.hip's arithmetic; the columns derived from the plan records by the staging waves themselves),
//      everything else as f32 spans by LDS-DMA (zeroed guards around them).  Wave 7 plans the NEXT step from the
//      kernel's first cycle on and joins the units afterwards.  The barrier behind this phase is a count in LDS.
//   B  the buffered tail goes to the stream's other history buffer (history alternates by step parity: no
//      step reads what it writes); then the outputs: D[16 classes][16 columns] += A[class][tap] * B[tap][column]
//      with v_mfma_f32_16x16x32_f16 on the image (three products per 32 taps) or v_mfma_f32_16x16x4_f32 on the
//      spans (exact f32, an fmaf chain over the taps), a column being one (stream, super period) pair -- "row =
//      stream": the window of class j of period q starts at frame q*a + off(j) of ITS stream, wherever that
//      stream stands; coefficients are the class tables of fir_periodic.h (the two phase rows pre-mixed with
//      the class's frac, shifted to the tile's common window, zero padded) in the operand order of the MFMA;
//   C  outputs whose f64 position fell just below an integer (previous frame, row 1023, :562-564);
//   D  streams that cannot use class tables (irrational ratio, drifted position) and streams whose
//      step saw a non-finite sum (inf / NaN, or a sample beyond the fp16 planes' range) are evaluated in the
//      reference's own form (two phase rows, eight partial sums lerped per lane, src/fir/avx.rs:25-58): a zero
//      padding coefficient times an infinity would otherwise turn finite reference outputs into NaN.
#include "fir_lockstep.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

#include "common.h"

namespace rsmp {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(1)))* gconst_f32_ptr;
typedef const v4f __attribute__((address_space(1)))* gconst_f4_ptr;
typedef float __attribute__((address_space(1)))* g_f32_ptr;
typedef const uint32_t __attribute__((address_space(4)))* const_u32_ptr;
// split variant (the operand types and scales of fir_split.hip)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef const v4u __attribute__((address_space(1)))* gconst_u4_ptr;
typedef const v2f __attribute__((address_space(1)))* gconst_f2_ptr;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
// samples: block floating point per column (`colpeak`), as fir_split.hip per item; taps: 2^13, in the table
constexpr uint32_t kLsPeakMax = 138;                               // no scale is derived from a peak of 2^11 and above (samples of 2^13 and above overflow: reference form)
constexpr uint32_t kLsPlanner = kLsWaves - 1;                      // the wave that plans the next step
constexpr uint32_t kLsStagers = kLsWaves - 2;                      // waves 1 .. kLsStagers stage the frames
constexpr uint32_t kLsSyncBytes = 32;                              // n_cols, unit counter, image counter, early flag, ready counter
constexpr uint32_t kLsMaxK32 = 6;                                  // 32-tap steps of a tile window (row_len <= 192)

template <class T>
__device__ __forceinline__ T load_uniform(const T* p) {   // wave-uniform POD through the scalar cache
    static_assert(sizeof(T) % 4 == 0, "dword-sized PODs only");
    T v;
    const_u32_ptr src = (const_u32_ptr)p;
    uint32_t* dst = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) dst[i] = src[i];
    return v;
}

constexpr uint32_t kFlagNonFinite = 1, kFlagReference = 2, kFlagRunOverflow = 4;

struct PlanLds {             // one stream's step, in LDS
    uint32_t n_out;          // output frames of the step
    uint32_t hist_frames;    // frames buffered before the step
    uint32_t accepted;       // frames taken from `in`
    uint32_t consumed;       // frames retired by the step
    uint32_t tail_frames;    // frames buffered after it
    uint32_t n_segs, n_wraps;
    uint32_t flags;
    uint64_t abs_out, abs_consumed;   // absolute counters before the step
    float* out;              // where the step's first output frame goes
    const void* runs;        // the step's exact position runs (SegLds[n_segs]): in LDS, or in the plan record
};
static_assert(sizeof(PlanLds) == 64, "PlanLds layout");

static_assert(sizeof(FirMirrorState) == 88, "the stash holds 16 states of 88 bytes in 16 x 96 bytes: the last 128 bytes are the column peaks");
constexpr uint32_t kLsPeakOff = kLsMaxSlots * 64 + kLsSyncBytes + kLsMaxSlots * 88;   // colpeak[16] (channel 0), peak counter, the columns' scales: channel 0 (4 words), channel 1 (4 words)
constexpr uint32_t kLsPeak1Off = kLsMaxSlots * 64 + kLsSyncBytes + kLsMaxSlots * 96;  // colpeak1[16]: channel 1 (round 5: a scale per channel, as fir_split.hip)

struct ColLds {              // one column of the matrix product: super period q of a stream
    int32_t frame0;          // span-relative frame of absolute input frame q * a
    int32_t n0;              // step-relative output index of (period q, class 0); may be negative
    uint32_t slot;
    uint32_t pad;
};

struct SegLds {
    uint32_t first, count;
    double p0, inc;
};

struct LsLayout {
    uint32_t ptrs, colsrc, cols, segs, wbits, wlist, spans, total;   // byte offsets
};
// data_bytes: the spans of the streams (slots x region_frames x channels f32) or the split image (rows x 160 B)
__host__ __device__ inline LsLayout ls_layout(uint32_t slots, uint32_t max_cols, uint32_t wrap_words,
                                              uint32_t wrap_cap, uint32_t data_bytes) {
    LsLayout l;
    l.ptrs = kLsMaxSlots * 64 + kLsSyncBytes + kLsMaxSlots * 96 + 64;  // PlanLds[16], sync words, state stash[16], channel 1's column peaks
    l.colsrc = l.ptrs + kLsMaxSlots * 32;                         // (hist, in, hist_next) pointers per slot
    l.cols = l.colsrc + 16 * 32;                                  // split: where each of the 16 columns' frames come from
    l.segs = (l.cols + max_cols * 16 + 7) & ~7u;
    l.wbits = l.segs + slots * kLsSegCap * 24;
    l.wlist = l.wbits + slots * wrap_words * 4;
    l.spans = (l.wlist + slots * wrap_cap * 4 + 15) & ~15u;
    l.total = l.spans + data_bytes;
    return l;
}
__host__ __device__ inline uint32_t ls_data_bytes(bool split, uint32_t rows, uint32_t row_bytes, uint32_t slots,
                                                  uint32_t region_frames, uint32_t channels) {
    return split ? rows * row_bytes : slots * region_frames * channels * 4u;
}

// mirror_call sink writing into LDS.
struct LdsSink {
    SegLds* segs;
    uint32_t* bits;
    uint32_t* list;
    uint32_t n_segs, n_wraps, wrap_cap;
    bool periodic, overflow;
    __host__ __device__ bool want_wraps() const { return periodic; }
    __host__ __device__ void run(uint64_t first, uint64_t count, double p0, double inc) {
        if (n_segs < kLsSegCap) {
            segs[n_segs].first = static_cast<uint32_t>(first);
            segs[n_segs].count = static_cast<uint32_t>(count);
            segs[n_segs].p0 = p0;
            segs[n_segs].inc = inc;
            ++n_segs;
        } else {
            overflow = true;
        }
    }
    __host__ __device__ void wrap(uint64_t index) {
        if (bits) bits[index >> 5] |= 1u << (index & 31);
        if (n_wraps < wrap_cap) list[n_wraps] = static_cast<uint32_t>(index);
        ++n_wraps;
    }
};

__device__ __forceinline__ float group_sum8(float v) {
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// The MFMA stream of one unit and one channel pair (NCH = 2: both channels of a frame with one 8-byte
// LDS read) or single channel (NCH = 1): the B operands of block b + 1 are read while block b's MFMAs
// issue; the A operands (the tile's coefficients) are already in registers.
template <int NCH>
__device__ __forceinline__ void unit_mfma(const v4f (&a_reg)[kLsMaxBlk], uint32_t nblk, const float* xb,
                                          uint32_t C, v4f& acc0, v4f& acc1) {
    v2f xc[4], xn[4];
    auto load4 = [&](v2f (&x)[4], const float* p) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (NCH == 2) x[s] = *reinterpret_cast<const v2f*>(p + 4 * s * C);
            else x[s] = v2f{p[4 * s * C], 0.f};
        }
    };
    load4(xc, xb);
#pragma unroll
    for (uint32_t blk = 0; blk < kLsMaxBlk; ++blk) {
        if (blk < nblk) {
            load4(xn, xb + 16 * (blk + 1 < nblk ? blk + 1 : blk) * C);
            const float av[4] = {a_reg[blk].x, a_reg[blk].y, a_reg[blk].z, a_reg[blk].w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], xc[s].x, acc0, 0, 0, 0);
                if constexpr (NCH == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], xc[s].y, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) xc[s] = xn[s];
        }
    }
}

// ---- split variant (two fp16 planes per f32 operand: fir_split.hip's arithmetic and image layout) --------
// f32 pair -> fp16 pair (round to nearest, one v_cvt_pk_f16_f32): low half = a, high half = b
__device__ __forceinline__ uint32_t ls_cvt_pk_f16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v2f{a, b}, f16x2));
}
__device__ __forceinline__ float ls_f16_lo(uint32_t w) { return static_cast<float>(__builtin_bit_cast(f16x2, w)[0]); }
__device__ __forceinline__ float ls_f16_hi(uint32_t w) { return static_cast<float>(__builtin_bit_cast(f16x2, w)[1]); }

// The MFMA stream of one unit on the image: row r of the image holds frame r of every column (a column's
// frame 0 is its super period's first frame), per row four 32-byte plane rows (channel 0 high / low plane,
// channel 1 high / low plane: 16 columns x 16 bits, 8-byte chunks XOR-swizzled by the row) + 32 bytes of
// padding.  A tile's window starts at row `base_row`: ds_read_b64_tr_b16 delivers 4 consecutive frames of
// the lane's column, the B operand of v_mfma_f32_16x16x32_f16 being two of those (frames 32 s + 4 grp .. and
// 32 s + 16 + 4 grp ..), in the order the split table (split_store_class) holds the taps.  Per 32 taps:
// c1 x2 + c2 x1 + c1 x1, smallest products first (fir_split.hip).
template <uint32_t NK, uint32_t ROWB>   // 32-tap steps of the tile window and the row pitch: compile-time, so that step s + 1's
                                        // reads are in flight under step s's MFMAs and every offset is an immediate
__device__ __forceinline__ void unit_mfma_split_nk(const v4f (&a_reg)[kLsMaxBlk], char* lds, uint32_t image_off,
                                                   uint32_t base_row, uint32_t lane, float os, float os1, v4f& acc0, v4f& acc1) {
    const uint32_t grp = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
    const uint32_t row0 = base_row + 4 * grp + q;
    const uint32_t base = image_off + row0 * ROWB + ((pc ^ ((row0 >> 2) & 3)) << 3);
    auto frag = [&](uint32_t plane_ch, uint32_t s) -> f16x8 {
        const uint32_t addr = base + plane_ch * 32u + s * (32u * ROWB);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + addr));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + addr + 16u * ROWB));
        return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
#pragma unroll
    for (uint32_t s = 0; s < NK; ++s) {
        const f16x8 c1 = __builtin_bit_cast(f16x8, a_reg[2 * s]), c2 = __builtin_bit_cast(f16x8, a_reg[2 * s + 1]);
        const f16x8 x1 = frag(0, s), x2 = frag(1, s), y1 = frag(2, s), y2 = frag(3, s);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1, x2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1, y2, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c2, x1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c2, y1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1, x1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1, y1, acc1, 0, 0, 0);
    }
    acc0 *= os;    // the lane's column's scale (channel 0) and the taps' 2^13, undone
    acc1 *= os1;   // ... channel 1's
}
template <uint32_t ROWB>
__device__ __forceinline__ void unit_mfma_split_rb(const v4f (&a_reg)[kLsMaxBlk], uint32_t nk, char* lds, uint32_t image_off,
                                                   uint32_t base_row, uint32_t lane, float os, float os1, v4f& acc0, v4f& acc1) {
    switch (nk) {   // (workgroup-uniform)
        case 1: unit_mfma_split_nk<1, ROWB>(a_reg, lds, image_off, base_row, lane, os, os1, acc0, acc1); break;
        case 2: unit_mfma_split_nk<2, ROWB>(a_reg, lds, image_off, base_row, lane, os, os1, acc0, acc1); break;
        case 3: unit_mfma_split_nk<3, ROWB>(a_reg, lds, image_off, base_row, lane, os, os1, acc0, acc1); break;
        case 4: unit_mfma_split_nk<4, ROWB>(a_reg, lds, image_off, base_row, lane, os, os1, acc0, acc1); break;
        case 5: unit_mfma_split_nk<5, ROWB>(a_reg, lds, image_off, base_row, lane, os, os1, acc0, acc1); break;
        default: unit_mfma_split_nk<kLsMaxK32, ROWB>(a_reg, lds, image_off, base_row, lane, os, os1, acc0, acc1); break;
    }
}
__device__ __forceinline__ void unit_mfma_split(const v4f (&a_reg)[kLsMaxBlk], uint32_t nk, uint32_t row_bytes, char* lds,
                                                uint32_t image_off, uint32_t base_row, uint32_t lane, float os, float os1, v4f& acc0, v4f& acc1) {
    if (row_bytes == kLsImageRowBytes) unit_mfma_split_rb<kLsImageRowBytes>(a_reg, nk, lds, image_off, base_row, lane, os, os1, acc0, acc1);
    else unit_mfma_split_rb<kLsImageRowBytesPacked>(a_reg, nk, lds, image_off, base_row, lane, os, os1, acc0, acc1);
}
// the scale of a column whose largest sample has biased exponent e (clamped: see colpeak): 2^(141 - e) puts that
// sample into [2^14, 2^15), inside fp16; and the factor that takes the column's sums back (2^13: the taps)
__device__ __forceinline__ uint32_t ls_peak_clamp(uint32_t e) { return e < 31u ? 31u : (e > kLsPeakMax ? kLsPeakMax : e); }
__device__ __forceinline__ float ls_col_unscale(uint32_t e) { return __uint_as_float((e - 27u) << 23); }   // (e: clamped)

// Frame f (relative to the first buffered frame) of a stream's [buffered | new] frames, channel c: from the
// LDS span (zeroed guards around it), or -- split variant, which keeps no f32 copy in LDS -- from HBM.
struct SpanView {
    const float* lds_span;       // f32 variant: frame 0 of the span in LDS
    const float* hist;           // split variant
    const float* in;
    uint32_t hist_frames, span_frames, C;
    __device__ __forceinline__ float at(int64_t f, uint32_t c) const {
        if (lds_span) return lds_span[f * static_cast<int64_t>(C) + c];
        if (f < 0 || f >= static_cast<int64_t>(span_frames)) return 0.f;
        return f < static_cast<int64_t>(hist_frames) ? hist[f * C + c] : in[(f - hist_frames) * C + c];
    }
};

// TRACE: diagnostic instantiation (RSMP_LS_TRACE=path): per workgroup the shader clock at the phase
// boundaries of wave 0 and of wave 1 goes to args.trace; the shipping instantiation has no trace code.
template <bool TRACE>
__global__ __launch_bounds__(kLsWaves * 64, kLsWaves / 2) void fir_lockstep_kernel(LockstepArgs args) {   // two workgroups per CU
    extern __shared__ __attribute__((aligned(16))) char lds[];
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ua[5] = {0, 0, 0, 0, 0};   // TRACE: unit setup, tile wait, MFMA stream, epilogue (cycles), units
    if constexpr (TRACE) tr[0] = __builtin_amdgcn_s_memtime();
    const LockstepGroup g = load_uniform(args.groups + blockIdx.x);
    const uint32_t C = g.channels;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool split = g.split != 0;   // (workgroup-uniform)
    const LsLayout lay = ls_layout(g.slots, g.max_cols, g.wrap_words, g.wrap_cap,
                                   ls_data_bytes(split, g.rows, g.row_bytes, g.slots, g.region_frames, C));
    struct SlotPtrs { const float* hist; const float* in; float* hist_next; uint64_t pad; };   // this step's buffered frames, its new frames, where its tail goes
    SlotPtrs* ptrs = reinterpret_cast<SlotPtrs*>(lds + lay.ptrs);   // split: the streams' frames are read from HBM
    struct ColSrc { const float* hist; const float* in; int32_t frame0; uint32_t hist_frames, span_frames, pad; };   // 32 B
    ColSrc* colsrc = reinterpret_cast<ColSrc*>(lds + lay.colsrc);
    PlanLds* plan = reinterpret_cast<PlanLds*>(lds);
    uint32_t* n_cols_p = reinterpret_cast<uint32_t*>(lds + kLsMaxSlots * 64);
    FirMirrorState* stash = reinterpret_cast<FirMirrorState*>(lds + kLsMaxSlots * 64 + kLsSyncBytes);   // new states until every reader of the old ones is done
    ColLds* cols = reinterpret_cast<ColLds*>(lds + lay.cols);
    SegLds* segs = reinterpret_cast<SegLds*>(lds + lay.segs);
    uint32_t* wbits = reinterpret_cast<uint32_t*>(lds + lay.wbits);
    uint32_t* wlist = reinterpret_cast<uint32_t*>(lds + lay.wlist);
    float* spans = reinterpret_cast<float*>(lds + lay.spans);
    const uint32_t region_dw = g.region_frames * C;

    // The coefficient tile of a unit (row_len / 16 blocks of 1 KB, L2 resident) is fetched whole into
    // registers before the unit's first MFMA: one L2 round trip per unit.  (Requesting a wave's first
    // tile before the planning phase would hide that too, but keeps 48 registers live across the
    // planner: it spilled under the 128-register cap that two workgroups per CU need.)
    const uint32_t nblk = g.row_len / 16;
    v4f a_reg[kLsMaxBlk];
    auto fetch_tile = [&](uint32_t t) {
        gconst_f4_ptr gA = (gconst_f4_ptr)(g.class_coef + static_cast<size_t>(t) * nblk * 256) + lane;
#pragma unroll
        for (uint32_t b = 0; b < kLsMaxBlk; ++b)
            if (b < nblk) a_reg[b] = gA[b * 64];
    };

    // ---- split variant: the image -------------------------------------------------------------------
    // Entry (row r, column c) = frame frame0(c) + r of the column's stream, cut into two fp16 planes of
    // 2^12 x (x * 2^12 = h1 + h2 + r, |r| <= 2^-22 |x|); frames outside the stream's [buffered | new] span and
    // unused columns are zero.  Lanes run along the rows: a wave instruction reads 64 consecutive frames.
    // src: lane c < 16 holds column c's ColSrc (8 words).
    const uint32_t image_off = lay.spans;
    const bool aligned8 = args.in_aligned8 != 0;
    // Block floating point per stream: a stream's columns are cut into planes scaled by a power of two that puts the
    // stream's level near the top of the fp16 range -- a quiet stream next to a loud one, or one far outside [-1, 1], keeps
    // 22 significant bits per sample.  The level is the largest magnitude in the stream's [buffered | new] span, which the
    // staging waves find with one scan (coalesced 8-byte loads, an atomic max per column in LDS):
    //   * normally the scale is PREDICTED from the peak the previous step's scan left in `peaks` (2^4 of headroom) and the
    //     image is written at once; the scan follows (its loads hit L2 then) and leaves this step's peak for the next
    //     step.  A stream that got louder than the headroom overflows a plane (non-finite sums), one that fell more than
    //     2^-6 below the prediction is seen by the units (peak against scale): both are redone in the reference's f32 form
    //     inside this launch (kFlagNonFinite), as inf / NaN samples are;
    //   * a step without a prediction for every stream of the workgroup (the first step) scans first: the staging waves
    //     meet at a count and take the exact peaks.
    uint32_t* colpeak = reinterpret_cast<uint32_t*>(lds + kLsPeakOff);   // [16]: bits of the largest |sample| of the column's stream; [16]: a count; [17..20]: the exponents the columns were scaled for
    uint8_t* colscale = reinterpret_cast<uint8_t*>(colpeak + 17);    // [16] channel 0, [16] channel 1
    uint32_t* colpeak1 = reinterpret_cast<uint32_t*>(lds + kLsPeak1Off);   // [16]: channel 1 (the reference computes every channel on its own, src/resampler_fir.rs:567-586)
    auto write_image = [&](uint32_t (&src)[8]) {
        const uint32_t t0 = threadIdx.x - 64;
        // the previous step's peak of column c's stream (lane c < 16; src[7] = stream index + 1, 0 = no stream)
        uint32_t pred_e = 0, pred_e1 = 0;
        bool pred_ok = true;
        if (lane < 16 && src[7] != 0) {
            const uint32_t* rec = args.peaks + 4 * static_cast<size_t>(src[7] - 1);
            pred_ok = rec[0] == args.epoch && rec[1] == args.step;
            pred_e = rec[2] >> 23;
            pred_e1 = rec[3] >> 23;
        }
        const bool predicted = __builtin_amdgcn_readfirstlane(__all(pred_ok) ? 1 : 0) != 0;
        // one scan per stream (the columns of a stream are neighbours); this wave's share of every column's peak by one atomic each
        auto scan = [&]() {
            uint32_t colpk = 0, colpk1 = 0;   // lane c < 16: this wave's share of column c's stream's peak, per channel
            uint32_t prev_lo = 0, prev_hi = 0, pv = 0, pv1 = 0;
#pragma unroll 1
            for (uint32_t c = 0; c < 16; ++c) {
                auto word = [&](int k) -> uint32_t { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(src[k]), static_cast<int>(c))); };
                const uint32_t h_lo = word(0), h_hi = word(1);
                const uint32_t span_fr = word(6);
                if (c == 0 || h_lo != prev_lo || h_hi != prev_hi) {
                    prev_lo = h_lo;
                    prev_hi = h_hi;
                    gconst_f32_ptr hist = (gconst_f32_ptr) reinterpret_cast<const float*>((static_cast<uint64_t>(h_hi) << 32) | h_lo);
                    gconst_f32_ptr in = (gconst_f32_ptr) reinterpret_cast<const float*>((static_cast<uint64_t>(word(3)) << 32) | word(2));
                    const uint32_t hist_fr = word(5);
                    float m = 0.f, m1 = 0.f;
                    for (uint32_t f = t0; f < span_fr; f += kLsStagers * 64) {
                        gconst_f32_ptr p = f < hist_fr ? hist + 2 * f : in + 2 * (f - hist_fr);
                        float a, b;
                        if (aligned8) {
                            const v2f v = *(gconst_f2_ptr)p;
                            a = v.x;
                            b = v.y;
                        } else {
                            a = p[0];
                            b = p[1];
                        }
                        m = __builtin_fmaxf(m, __builtin_fabsf(a));   // (a NaN is left to the sums)
                        m1 = __builtin_fmaxf(m1, __builtin_fabsf(b));
                    }
                    // (non-negative floats order like their bit patterns; the two channels' exponents travel as the 16-bit
                    // halves of one word: the wave's reduction is as long as for one)
                    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                    auto pkmax = [](uint32_t x, uint32_t y) -> uint32_t {
                        us2 a2, b2;
                        __builtin_memcpy(&a2, &x, 4);
                        __builtin_memcpy(&b2, &y, 4);
                        const us2 z = __builtin_elementwise_max(a2, b2);
                        uint32_t r;
                        __builtin_memcpy(&r, &z, 4);
                        return r;
                    };
                    uint32_t mb = (__float_as_uint(m) >> 23) | ((__float_as_uint(m1) >> 23) << 16);
                    mb = pkmax(mb, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(mb), 0x128, 0xf, 0xf, false)));
                    mb = pkmax(mb, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(mb), 0x124, 0xf, 0xf, false)));
                    mb = pkmax(mb, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(mb), 0x122, 0xf, 0xf, false)));
                    mb = pkmax(mb, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(mb), 0x121, 0xf, 0xf, false)));
                    const uint32_t r0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mb), 0)),
                                   r1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mb), 16)),
                                   r2 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mb), 32)),
                                   r3 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(mb), 48));
                    pv = max(max(r0 & 0xFFFFu, r1 & 0xFFFFu), max(r2 & 0xFFFFu, r3 & 0xFFFFu)) << 23;   // (kept as the bits of 2^e: the records' form)
                    pv1 = max(max(r0 >> 16, r1 >> 16), max(r2 >> 16, r3 >> 16)) << 23;
                }
                colpk = lane == c ? pv : colpk;
                colpk1 = lane == c ? pv1 : colpk1;
            }
            if (lane < 16 && colpk != 0)
                (void)__hip_atomic_fetch_max(colpeak + lane, colpk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane < 16 && colpk1 != 0)
                (void)__hip_atomic_fetch_max(colpeak1 + lane, colpk1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        // this step's peaks to the next step's predictions: by the staging wave that counts in last (every share is in then)
        auto count_in_and_publish = [&]() -> bool {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            uint32_t old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(colpeak + 16, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            const bool last = __builtin_amdgcn_readfirstlane(old) == kLsStagers - 1;
            if (last && lane < 16 && src[7] != 0) {
                uint32_t* rec = args.peaks + 4 * static_cast<size_t>(src[7] - 1);
                const uint32_t bits = __hip_atomic_load(colpeak + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t bits1 = __hip_atomic_load(colpeak1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                rec[0] = args.epoch;
                rec[1] = args.step + 1u;
                rec[2] = bits;
                rec[3] = bits1;
            }
            return last;
        };
        uint32_t e_used, e_used1;   // lane c < 16: the exponents column c's two channels are scaled for
        if (predicted) {
            e_used = pred_e == 0 ? 127u : pred_e + 4u;   // (a silent channel: as full-scale audio; a first sample overflows, is redone)
            e_used1 = pred_e1 == 0 ? 127u : pred_e1 + 4u;
        } else {
            scan();
            (void)count_in_and_publish();
            while (__hip_atomic_load(colpeak + 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < kLsStagers) __builtin_amdgcn_s_sleep(1);
            e_used = __hip_atomic_load(colpeak + (lane & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 23;
            e_used1 = __hip_atomic_load(colpeak1 + (lane & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 23;
        }
        e_used = ls_peak_clamp(e_used);
        e_used1 = ls_peak_clamp(e_used1);
        if (wave == 1 && lane < 16) {
            colscale[lane] = static_cast<uint8_t>(e_used);
            colscale[16 + lane] = static_cast<uint8_t>(e_used1);
        }
        uint32_t xsv = (268u - e_used) << 23, xsv1 = (268u - e_used1) << 23;
        asm volatile("" : "+v"(xsv), "+v"(xsv1));   // lane c < 16: column c's scales (defined under the full EXEC mask: read by lane index below)
        for (uint32_t r = t0; r < g.rows; r += kLsStagers * 64) {
            // every column's frame of this row is requested before the first is converted: one memory
            // latency per row block, not one per column
            float x0[16], x1[16];
            bool ok[16];
#pragma unroll
            for (uint32_t c = 0; c < 16; ++c) {
                // branch-free: an entry outside its stream's frames (or of an unused column) loads the first
                // buffered value instead and drops it -- a guarded load would end the run of loads in flight.
                // Column c's description sits in lane c of `src` and is the same for every lane: scalar registers.
                auto word = [&](int k) -> uint32_t { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(src[k]), c)); };
                gconst_f32_ptr hist = (gconst_f32_ptr) reinterpret_cast<const float*>((static_cast<uint64_t>(word(1)) << 32) | word(0));
                gconst_f32_ptr in = (gconst_f32_ptr) reinterpret_cast<const float*>((static_cast<uint64_t>(word(3)) << 32) | word(2));
                const int32_t frame0 = static_cast<int32_t>(word(4));
                const uint32_t hist_fr = word(5), span_fr = word(6);
                const int32_t f = frame0 + static_cast<int32_t>(r);
                ok[c] = f >= 0 && static_cast<uint32_t>(f) < span_fr;
                const uint32_t fu = ok[c] ? static_cast<uint32_t>(f) : 0u;
                gconst_f32_ptr p = fu < hist_fr ? hist + 2 * fu : in + 2 * (fu - hist_fr);
                if (!ok[c]) p = hist;   // (the history buffer always exists)
                if (aligned8) {   // (uniform) one 8-byte load per frame: half the load instructions of the phase
                    const v2f v = *(gconst_f2_ptr)p;
                    x0[c] = v.x;
                    x1[c] = v.y;
                } else {
                    x0[c] = p[0];
                    x1[c] = p[1];
                }
            }
#pragma unroll
            for (uint32_t c = 0; c < 16; ++c) {
                x0[c] = ok[c] ? x0[c] : 0.f;
                x1[c] = ok[c] ? x1[c] : 0.f;
            }
            char* row = lds + image_off + r * g.row_bytes;
            const uint32_t sw = (r >> 2) & 3;
#pragma unroll
            for (uint32_t c = 0; c < 16; ++c) {
                const float xs = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(xsv), c)));
                const float xs1 = __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(xsv1), c)));
                const float s0 = x0[c] * xs, s1 = x1[c] * xs1;
                const uint32_t hi = ls_cvt_pk_f16(s0, s1);
                const uint32_t lo = ls_cvt_pk_f16(s0 - ls_f16_lo(hi), s1 - ls_f16_hi(hi));
                char* e = row + ((((c >> 2) ^ sw) << 3) + (c & 3) * 2);
                *reinterpret_cast<uint16_t*>(e) = static_cast<uint16_t>(hi);            // channel 0, high plane
                *reinterpret_cast<uint16_t*>(e + 32) = static_cast<uint16_t>(lo);       // channel 0, low plane
                *reinterpret_cast<uint16_t*>(e + 64) = static_cast<uint16_t>(hi >> 16);  // channel 1, high plane
                *reinterpret_cast<uint16_t*>(e + 96) = static_cast<uint16_t>(lo >> 16);  // channel 1, low plane
            }
        }
        if (predicted) {   // the step's own peaks: for the units' check and the next step's predictions
            scan();
            (void)count_in_and_publish();
        }
    };

    // ---- A: plan (wave 0) | stage (waves 1..) ---------------------------------------------------
    uint32_t my_in_fr = 0;      // wave 0, lanes < count: frames offered to the lane's stream in this step
    bool from_record = false;   // ... and whether its plan came from the record (then the planner wave plans the next step)
    // The sync words start at zero (unit counter, image counter, early flag, ready counter): the only
    // workgroup barrier before the final one -- every wave is here at once.
    if (threadIdx.x < kLsSyncBytes / 4) n_cols_p[threadIdx.x] = 0;
    if (threadIdx.x >= 64 && threadIdx.x < 64 + 25) reinterpret_cast<uint32_t*>(lds + kLsPeakOff)[threadIdx.x - 64] = 0;   // column peaks, their count, the scales
    if (threadIdx.x >= 128 && threadIdx.x < 128 + 16) reinterpret_cast<uint32_t*>(lds + kLsPeak1Off)[threadIdx.x - 128] = 0;   // channel 1's column peaks
    __syncthreads();

    // The NEXT step's plan (the reference's control flow for step k + 1, ~20 k cycles of serial f64 arithmetic
    // on one lane per stream) goes to the stream's plan record.  A wave of its own does that from the kernel's
    // first cycle on -- the state it starts from is in the record of THIS step (`after`) -- and takes part in
    // nothing else before the units: it used to sit on wave 0 behind the barrier, where it was the longest
    // chain of the step.  (When this step has no usable record, wave 0 plans in line and then does it itself.)
    auto plan_ahead = [&](uint32_t gs, FirMirrorState st, uint32_t offered) {
        char* nrec = args.recs + (static_cast<size_t>((args.step + 1u) & 1u) * args.n_streams + gs) * args.rec_stride;
        uint32_t in_fr = offered;
        const uint32_t room = g.span_frames > st.available ? g.span_frames - static_cast<uint32_t>(st.available) : 0u;
        if (in_fr > room) in_fr = room;
        LdsSink sink{reinterpret_cast<SegLds*>(nrec + kLsRecSegs), nullptr, reinterpret_cast<uint32_t*>(nrec + kLsRecWraps),
                     0u, 0u, g.wrap_cap, g.periodic != 0 && st.periodic_ok != 0, false};
        LsPlanHeader hd;
        hd.hist_frames = static_cast<uint32_t>(st.available);
        hd.abs_out = st.abs_out;
        hd.abs_consumed = st.abs_consumed;
        const FirCallCounts c = mirror_call(st, in_fr, args.streams[gs].out_cap_frames, sink);
        hd.epoch = args.epoch;
        hd.step = args.step + 1u;
        hd.in_frames = offered;
        hd.n_out = static_cast<uint32_t>(c.produced);
        hd.accepted = static_cast<uint32_t>(c.accepted);
        hd.consumed = static_cast<uint32_t>(c.consumed);
        hd.tail_frames = static_cast<uint32_t>(st.available);
        hd.n_segs = sink.n_segs;
        hd.n_wraps = sink.n_wraps < g.wrap_cap ? sink.n_wraps : g.wrap_cap;
        hd.flags = (sink.periodic && st.periodic_ok != 0 ? 0u : kFlagReference) | (sink.overflow ? kFlagRunOverflow : 0u);
        hd.pad0 = 0;
        hd.pad1 = 0;
        hd.after = st;
        *reinterpret_cast<LsPlanHeader*>(nrec) = hd;
    };
    if (wave == kLsPlanner && lane < g.count) {
        const uint32_t gs = g.first + lane;
        const uint32_t offered = args.in_frames_per_stream ? args.in_frames_per_stream[args.order[gs]] : args.in_frames;
        const LsPlanHeader* hd = reinterpret_cast<const LsPlanHeader*>(
            args.recs + (static_cast<size_t>(args.step & 1u) * args.n_streams + gs) * args.rec_stride);
        if (hd->epoch == args.epoch && hd->step == args.step && hd->in_frames == offered) plan_ahead(gs, hd->after, offered);
    }

    // Split variant, steady state: the staging waves do not wait for wave 0's column table.  Each reads the
    // streams' plan records itself (the same few words wave 0 reads), derives the columns in registers and
    // writes its share of the image while wave 0 is still planning: the units start right behind the barrier.
    // (Not when a record is stale -- first step, changed frame count -- or frames are offered per stream: then
    // the table wave 0 leaves in LDS is used, after the barrier.)
    bool early = false;
    const bool stager = wave >= 1 && wave <= kLsStagers;
    if (split && stager && args.in_frames_per_stream == nullptr) {
        bool valid = true;
        uint32_t mc = 0, f_hist_fr = 0, f_span_fr = 0;
        int32_t f_base = 0;             // frame0 of the stream's first column
        uint64_t f_hist = 0, f_in = 0;
        if (lane < g.count) {
            const uint32_t gs = g.first + lane;
            const LsPlanHeader* hd = reinterpret_cast<const LsPlanHeader*>(
                args.recs + (static_cast<size_t>(args.step & 1u) * args.n_streams + gs) * args.rec_stride);
            const uint32_t h_epoch = hd->epoch, h_step = hd->step, h_in = hd->in_frames, h_n_out = hd->n_out;
            const uint32_t h_hist = hd->hist_frames, h_acc = hd->accepted, h_flags = hd->flags;
            const uint64_t h_abs_out = hd->abs_out, h_abs_consumed = hd->abs_consumed;
            const LockstepStream& ls = args.streams[gs];
            f_hist = reinterpret_cast<uint64_t>((args.hist_parity != 0) ? ls.hist_alt : ls.hist);
            f_in = reinterpret_cast<uint64_t>(ls.in + args.in_offset * C);
            valid = h_epoch == args.epoch && h_step == args.step && h_in == args.in_frames;
            if (valid && h_n_out != 0 && !(h_flags & kFlagReference)) {
                const uint64_t q_first = h_abs_out / g.b;
                mc = static_cast<uint32_t>((h_abs_out + h_n_out - 1) / g.b - q_first) + 1;
                f_base = static_cast<int32_t>(static_cast<int64_t>(q_first * g.a) - static_cast<int64_t>(h_abs_consumed));
            }
            f_hist_fr = h_hist;
            f_span_fr = h_hist + h_acc;
        }
        if (__builtin_amdgcn_readfirstlane(__all(valid) ? 1 : 0)) {
            // column c = lane c: find its stream, take that stream's fields
            uint32_t src[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            uint32_t acc = 0;
            const uint32_t hist0_lo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(f_hist)), 0));
            const uint32_t hist0_hi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(f_hist >> 32)), 0));
            src[0] = hist0_lo;   // a column without a stream: no frames, a pointer that can be dereferenced
            src[1] = hist0_hi;
            for (uint32_t s = 0; s < g.count; ++s) {
                auto bc = [&](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), static_cast<int>(s))); };
                const uint32_t n = bc(mc);
                const uint32_t room = g.max_cols > acc ? g.max_cols - acc : 0u;
                const uint32_t take = n < room ? n : room;
                const uint32_t w0 = bc(static_cast<uint32_t>(f_hist)), w1 = bc(static_cast<uint32_t>(f_hist >> 32));
                const uint32_t w2 = bc(static_cast<uint32_t>(f_in)), w3 = bc(static_cast<uint32_t>(f_in >> 32));
                const uint32_t w4 = bc(static_cast<uint32_t>(f_base)), w5 = bc(f_hist_fr), w6 = bc(f_span_fr);
                if (lane >= acc && lane < acc + take) {
                    src[0] = w0; src[1] = w1; src[2] = w2; src[3] = w3;
                    src[4] = w4 + (lane - acc) * g.a;
                    src[5] = w5; src[6] = w6;
                    src[7] = g.first + s + 1u;   // the column's stream (+ 1: 0 = a column without one)
                }
                acc += take;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(src[k]));   // (defined under the full EXEC mask: see below)
            early = true;
            if (acc != 0) write_image(src);
            if constexpr (TRACE) tr[1] = __builtin_amdgcn_s_memtime();
        }
    }
    if (split && stager && lane == 0) n_cols_p[3] = early ? 1u : 0u;   // (every staging wave writes the same value, before the barrier)
    if (wave == 0) {
        if (lane < g.count) {
            const uint32_t gs = g.first + lane;
            const LockstepStream ls = args.streams[gs];
            const uint32_t in_off = args.in_frames_per_stream ? args.in_frames_per_stream[args.order[gs]] : args.in_frames;
            my_in_fr = in_off;
            uint32_t* bits = wbits + lane * g.wrap_words;
            for (uint32_t w = 0; w < g.wrap_words; ++w) bits[w] = 0;
            const char* rec = args.recs + (static_cast<size_t>(args.step & 1u) * args.n_streams + gs) * args.rec_stride;
            const LsPlanHeader hd = *reinterpret_cast<const LsPlanHeader*>(rec);
            PlanLds pl;
            FirMirrorState st;
            from_record = hd.epoch == args.epoch && hd.step == args.step && hd.in_frames == in_off;
            if (from_record) {
                // planned a step ahead: take the record
                pl.n_out = hd.n_out;
                pl.hist_frames = hd.hist_frames;
                pl.accepted = hd.accepted;
                pl.consumed = hd.consumed;
                pl.tail_frames = hd.tail_frames;
                pl.n_segs = hd.n_segs;
                pl.n_wraps = hd.n_wraps;
                pl.flags = hd.flags;
                pl.abs_out = hd.abs_out;
                pl.abs_consumed = hd.abs_consumed;
                pl.runs = rec + kLsRecSegs;
                st = hd.after;
                const uint32_t* wl = reinterpret_cast<const uint32_t*>(rec + kLsRecWraps);
                for (uint32_t i = 0; i < hd.n_wraps; ++i) {
                    const uint32_t n = wl[i];
                    wlist[lane * g.wrap_cap + i] = n;
                    bits[n >> 5] |= 1u << (n & 31);
                }
            } else {
                st = args.states[gs];
                uint32_t in_fr = in_off;
                const uint32_t room = g.span_frames > st.available ? g.span_frames - static_cast<uint32_t>(st.available) : 0u;
                if (in_fr > room) in_fr = room;   // (cannot happen when out_cap >= buffer_size_output: available < taps)
                LdsSink sink{segs + lane * kLsSegCap, bits, wlist + lane * g.wrap_cap, 0u, 0u, g.wrap_cap,
                             g.periodic != 0 && st.periodic_ok != 0, false};
                pl.hist_frames = static_cast<uint32_t>(st.available);
                pl.abs_out = st.abs_out;
                pl.abs_consumed = st.abs_consumed;
                const FirCallCounts c = mirror_call(st, in_fr, ls.out_cap_frames, sink);
                pl.n_out = static_cast<uint32_t>(c.produced);
                pl.accepted = static_cast<uint32_t>(c.accepted);
                pl.consumed = static_cast<uint32_t>(c.consumed);
                pl.tail_frames = static_cast<uint32_t>(st.available);
                pl.n_segs = sink.n_segs;
                pl.n_wraps = sink.n_wraps < g.wrap_cap ? sink.n_wraps : g.wrap_cap;
                pl.flags = (sink.periodic && st.periodic_ok != 0 ? 0u : kFlagReference) |
                           (sink.overflow ? kFlagRunOverflow : 0u);
                pl.runs = segs + lane * kLsSegCap;
            }
            uint64_t cursor = 0;
            if (args.append) {
                cursor = args.out_cursor[gs];
                args.out_cursor[gs] = cursor + static_cast<uint64_t>(pl.n_out) * C;
            }
            pl.out = ls.out + cursor;
            plan[lane] = pl;
            stash[lane] = st;
            ptrs[lane] = SlotPtrs{(args.hist_parity != 0) ? ls.hist_alt : ls.hist, ls.in + args.in_offset * C,
                                  (args.hist_parity != 0) ? ls.hist : ls.hist_alt, 0};
            args.counts[2 * gs] = static_cast<uint64_t>(pl.accepted) * C;
            args.counts[2 * gs + 1] = static_cast<uint64_t>(pl.n_out) * C;
        }
        // the lanes of this wave read each other's PlanLds next: LDS operations of a wave complete in order, so
        // only the compiler has to keep the order (a release fence would also wait for the stores to HBM above)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if constexpr (TRACE) tr[1] = __builtin_amdgcn_s_memtime();   // planned
        // The column table: every (stream, super period) pair with outputs in this step.  Each planner
        // lane places its own stream's columns behind those of the lanes before it.
        uint32_t my_cols = 0;
        uint64_t q_first = 0;
        if (lane < g.count && g.periodic) {
            const PlanLds& pl = plan[lane];
            if (pl.n_out != 0 && !(pl.flags & kFlagReference)) {
                q_first = pl.abs_out / g.b;
                my_cols = static_cast<uint32_t>((pl.abs_out + pl.n_out - 1) / g.b - q_first) + 1;
            }
        }
        uint32_t before = 0, total = 0;
        for (uint32_t s = 0; s < g.count; ++s) {
            const uint32_t n = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(my_cols), static_cast<int>(s)));   // (s is uniform: no LDS round trip)
            if (s < lane) before += n;
            total += n;
        }
        // (split: what the staging waves need per column goes to a flat table -- no chain of dependent LDS reads
        // there; a column without a stream has no frames and a history pointer that can be dereferenced)
        if (split && lane < 16) colsrc[lane] = ColSrc{ptrs[0].hist, ptrs[0].in, 0, 0u, 0u, 0u};
        if (lane < g.count) {
            const PlanLds& pl = plan[lane];
            for (uint32_t i = 0; i < my_cols && before + i < g.max_cols; ++i) {
                const uint64_t q = q_first + i;
                ColLds cl;
                cl.frame0 = static_cast<int32_t>(static_cast<int64_t>(q * g.a) - static_cast<int64_t>(pl.abs_consumed));
                cl.n0 = static_cast<int32_t>(static_cast<int64_t>(q * g.b) - static_cast<int64_t>(pl.abs_out));
                cl.slot = lane;
                cl.pad = 0;
                cols[before + i] = cl;
                if (split)
                    colsrc[before + i] = ColSrc{ptrs[lane].hist, ptrs[lane].in, cl.frame0, pl.hist_frames,
                                                pl.hist_frames + pl.accepted, g.first + lane + 1u};   // (pad: the stream's index + 1)
            }
        }
        if (lane == 0) {
            n_cols_p[0] = total < g.max_cols ? total : g.max_cols;
        }
    } else if (!split && stager) {
        // Stage [buffered | new] of every stream with LDS-DMA (global_load_lds, 256 B per wave instruction,
        // no VGPR round trip: every piece of every stream is in flight at once); everything else of the
        // region is zeroed (the guards are read by masked columns and by zero padding coefficients: they
        // must be finite).  The dwords a stream's last, partial piece writes beyond its span are zeroed by
        // the wave that issued it, after the piece has landed.
        typedef __attribute__((address_space(3))) void* lds_void_ptr;
        const uint32_t t0 = threadIdx.x - 64;
        const uint32_t nt = kLsStagers * 64;
        // lane s of every staging wave fetches stream s's parameters: one round trip for all streams
        uint32_t my_hist_dw = 0, my_span_dw = 0;
        unsigned long long my_hist = 0, my_in = 0;
        if (lane < g.count) {
            const uint32_t gs = g.first + lane;
            // the frames this step accepts (the first lines of mirror_call)
            const uint64_t avail = args.states[gs].available;
            const uint64_t readp = args.states[gs].read_position;
            uint32_t in_fr = args.in_frames_per_stream ? args.in_frames_per_stream[args.order[gs]] : args.in_frames;
            const uint32_t room = g.span_frames > avail ? g.span_frames - static_cast<uint32_t>(avail) : 0u;
            if (in_fr > room) in_fr = room;
            const uint64_t wp = readp + avail;
            const uint64_t rem = kMirrorBufferSize > wp ? kMirrorBufferSize - wp : 0;
            uint64_t acc = in_fr < rem ? in_fr : rem;
            if (acc > kMirrorInputCapacity - avail) acc = kMirrorInputCapacity - avail;
            my_hist_dw = static_cast<uint32_t>(avail) * C;
            my_span_dw = my_hist_dw + static_cast<uint32_t>(acc) * C;
            my_hist = reinterpret_cast<unsigned long long>((args.hist_parity != 0) ? args.streams[gs].hist_alt : args.streams[gs].hist);
            my_in = reinterpret_cast<unsigned long long>(args.streams[gs].in + args.in_offset * C);
        }
        const uint32_t guard_dw = g.guard_frames * C;
        for (uint32_t s = 0; s < g.count; ++s) {
            const uint32_t hist_dw = __shfl(my_hist_dw, s, 64);
            const uint32_t span_dw = __shfl(my_span_dw, s, 64);
            gconst_f32_ptr hist = (gconst_f32_ptr) reinterpret_cast<const float*>(__shfl(my_hist, s, 64));
            gconst_f32_ptr in = (gconst_f32_ptr) reinterpret_cast<const float*>(__shfl(my_in, s, 64));
            float* region = spans + s * region_dw;
            const uint32_t pieces = (span_dw + 63) / 64;
            for (uint32_t p = wave - 1; p < pieces; p += kLsStagers) {
                const uint32_t k = p * 64 + lane;
                const uint32_t kc = k < span_dw ? k : span_dw - 1;
                gconst_f32_ptr src = kc < hist_dw ? hist + kc : in + (kc - hist_dw);
                __builtin_amdgcn_global_load_lds(src, (lds_void_ptr)(region + guard_dw + p * 64), 4, 0, 0);
            }
            for (uint32_t i = t0; i < guard_dw; i += nt) region[i] = 0.f;
            for (uint32_t i = guard_dw + pieces * 64 + t0; i < region_dw; i += nt) region[i] = 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces have landed
        for (uint32_t s = 0; s < g.count; ++s) {
            const uint32_t span_dw = __shfl(my_span_dw, s, 64);
            const uint32_t pieces = (span_dw + 63) / 64;
            if (pieces != 0 && (pieces - 1) % kLsStagers == wave - 1) {   // this wave issued the last piece
                const uint32_t k = (pieces - 1) * 64 + lane;
                if (k >= span_dw) spans[s * region_dw + guard_dw + k] = 0.f;
            }
        }
    }
    if constexpr (TRACE) tr[2] = __builtin_amdgcn_s_memtime();   // wave 0: columns built; others: staged
    // Wave 0 and the staging waves count in (their LDS writes first), every wave waits for the seven counts:
    // a barrier the planner wave does not hold up.
    if (wave == 0 || stager) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) (void)__hip_atomic_fetch_add(n_cols_p + 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (LDS operations of a wave complete in order)
    }
    while (__hip_atomic_load(n_cols_p + 4, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 1 + kLsStagers)
        __builtin_amdgcn_s_sleep(1);
    if constexpr (TRACE) tr[3] = __builtin_amdgcn_s_memtime();
    if (wave == 0 && lane < g.count) {
        const uint32_t gs = g.first + lane;
        const FirMirrorState st = stash[lane];
        args.states[gs] = st;   // every reader of the old state has counted in
        if (!from_record) plan_ahead(gs, st, my_in_fr);
    }

    // ---- split variant, records not usable: the image from the column table wave 0 left in LDS ----------
    if (split && stager && *n_cols_p != 0 && !early) {
        uint32_t src[8];   // lane c < 16: column c's ColSrc
        {
            const uint32_t* cs = reinterpret_cast<const uint32_t*>(colsrc) + (lane & 15) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) src[k] = cs[k];
            // pinned here, where every lane is active: v_readlane below reads lanes 0..15 whatever the loop's
            // EXEC mask is, and a wave's last row block may have fewer than 16 rows (the compiler sank the loads
            // into the loop: lanes without a row then never loaded their column)
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(src[k]));
        }
        write_image(src);
        // one count per wave: its share of the image is in LDS when the count becomes visible
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) (void)__hip_atomic_fetch_add(n_cols_p + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (TRACE) tr[1] = __builtin_amdgcn_s_memtime();   // (staging waves: their share of the image written)
    }

    // ---- B: retire: the still-buffered tail goes to the front of the stream's OTHER history buffer (the
    // next step reads that one: nothing of this step reads what is written here, so no ordering is needed)
    for (uint32_t s = 0; s < g.count; ++s) {
        const PlanLds& pl = plan[s];
        const uint32_t tail_dw = pl.tail_frames * C;
        g_f32_ptr dst = (g_f32_ptr)ptrs[s].hist_next;
        if (!split) {
            const float* src = spans + s * region_dw + (g.guard_frames + pl.consumed) * C;
            for (uint32_t i = threadIdx.x; i < tail_dw; i += kLsWaves * 64) dst[i] = src[i];
        } else {
            const uint32_t hist_dw = pl.hist_frames * C, first = pl.consumed * C;
            gconst_f32_ptr hist = (gconst_f32_ptr)ptrs[s].hist, in = (gconst_f32_ptr)ptrs[s].in;
            for (uint32_t i = threadIdx.x; i < tail_dw; i += kLsWaves * 64) {
                const uint32_t e = first + i;
                dst[i] = e < hist_dw ? hist[e] : in[e - hist_dw];
            }
        }
    }

    // ---- B: matrix-core units (16 columns x one 16-class tile) ------------------------------------
    const uint32_t n_cols = *n_cols_p;
    if (n_cols) {
        const uint32_t n_chunks = (n_cols + 15) / 16;
        const uint32_t n_units = n_chunks * g.n_tiles;
        const bool pair_ok = (C & 1u) == 0;   // both channels of a frame with one 8-byte LDS read
        if (split && n_cols_p[3] == 0)   // (word 3: the image was written before the barrier)
            // the image is complete once the seven staging waves have counted in
            while (__hip_atomic_load(n_cols_p + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < kLsStagers)
                __builtin_amdgcn_s_sleep(1);
        // units are claimed from an LDS counter: the first wave joins late (it has planned the next step)
        for (;;) {
            uint32_t u_claim = 0;
            if (lane == 0) u_claim = atomicAdd(n_cols_p + 1, 1u);
            const uint32_t u = __builtin_amdgcn_readfirstlane(u_claim);
            if (u >= n_units) break;
            unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
            if constexpr (TRACE) ts0 = __builtin_amdgcn_s_memtime();
            const uint32_t chunk = u / g.n_tiles;
            const uint32_t t = u - chunk * g.n_tiles;
            const uint32_t tile_base = static_cast<uint32_t>((static_cast<uint64_t>(t) * 16u * g.a) / g.b);   // TileMeta::base = class_offset(16 t)
            const uint32_t col = chunk * 16 + (lane & 15);
            const bool on = col < n_cols;
            const ColLds cl = cols[on ? col : chunk * 16];
            const uint32_t n_out = plan[cl.slot].n_out;
            float* out = plan[cl.slot].out;
            const uint32_t* bits = wbits + cl.slot * g.wrap_words;
            const float* xb = spans + cl.slot * region_dw +
                              static_cast<int32_t>(static_cast<int32_t>(g.guard_frames) + cl.frame0 +
                                                   static_cast<int32_t>(tile_base) + static_cast<int32_t>(lane >> 4)) *
                                  static_cast<int32_t>(C);
            fetch_tile(t);
            if constexpr (TRACE) {
                ts1 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ts2 = __builtin_amdgcn_s_memtime();
            }
            const uint32_t j = t * 16 + 4 * (lane >> 4);          // first of the lane's four classes
            const uint32_t jw = (t * 16 + g.den - 1) / g.den * g.den;   // first class of the tile at an integer position
            for (uint32_t c0 = 0; c0 < C; c0 += 2) {
                const bool two = c0 + 1 < C;
                v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                if (split) {
                    unit_mfma_split(a_reg, g.row_len / 32, g.row_bytes, lds, image_off, tile_base, lane,
                                    ls_col_unscale(colscale[lane & 15u]), ls_col_unscale(colscale[16u + (lane & 15u)]), acc0, acc1);
                } else if (two && pair_ok) {
                    unit_mfma<2>(a_reg, nblk, xb + c0, C, acc0, acc1);
                } else {
                    v4f unused = {0.f, 0.f, 0.f, 0.f};
                    unit_mfma<1>(a_reg, nblk, xb + c0, C, acc0, unused);
                    if (two) unit_mfma<1>(a_reg, nblk, xb + c0 + 1, C, acc1, unused);
                }
                if constexpr (TRACE) {
                    asm volatile("" :: "v"(acc0), "v"(acc1));
                    ts3 = __builtin_amdgcn_s_memtime();
                }
                // a non-finite sum anywhere in the tile: the stream's step is redone in reference form
                const float chk = (acc0.x + acc0.y) + (acc0.z + acc0.w) + (acc1.x + acc1.y) + (acc1.z + acc1.w);
                // ... and so is one whose level fell more than 2^-10 below what its planes were scaled for (2^-6 below the
                // prediction): the planes no longer hold its samples to 22 bits
                bool redo = !(fabsf(chk) <= FLT_MAX);
                if (split) {
                    const uint32_t e_act = colpeak[lane & 15u] >> 23, e_sc = colscale[lane & 15u];
                    const uint32_t e_act1 = colpeak1[lane & 15u] >> 23, e_sc1 = colscale[16u + (lane & 15u)];
                    redo = redo || (e_act != 0 && e_act + 10u < e_sc) || (e_act1 != 0 && e_act1 + 10u < e_sc1);
                }
                if (on && redo)
                    (void)__hip_atomic_fetch_or(&plan[cl.slot].flags, kFlagNonFinite, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_WORKGROUP);
                const float v0[4] = {acc0.x, acc0.y, acc0.z, acc0.w};
                const float v1[4] = {acc1.x, acc1.y, acc1.z, acc1.w};
                bool ok[4];
                bool all = on;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t jr = j + r;
                    const int32_t n = cl.n0 + static_cast<int32_t>(jr);
                    bool valid = on && jr < g.b && n >= 0 && n < static_cast<int32_t>(n_out);
                    if (valid) {
                        const bool at_integer = g.den >= 16 ? jr == jw : jr % g.den == 0;
                        if (at_integer && ((bits[static_cast<uint32_t>(n) >> 5] >> (n & 31)) & 1u)) valid = false;   // phase C
                    }
                    ok[r] = valid;
                    all = all && valid;
                }
                const int32_t nl = cl.n0 + static_cast<int32_t>(j);
                if (all && C == 2) {
                    typedef v4f __attribute__((address_space(1), aligned(8)))* g_f4a8_ptr;
                    g_f4a8_ptr o = (g_f4a8_ptr)(out + static_cast<size_t>(nl) * 2);
                    o[0] = v4f{v0[0], v1[0], v0[1], v1[1]};
                    o[1] = v4f{v0[2], v1[2], v0[3], v1[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ok[r]) {
                            g_f32_ptr o = (g_f32_ptr)(out + static_cast<size_t>(nl + r) * C + c0);
                            o[0] = v0[r];
                            if (two) o[1] = v1[r];
                        }
                }
            }
            if constexpr (TRACE) {
                const unsigned long long ts4 = __builtin_amdgcn_s_memtime();
                ua[0] += ts1 - ts0; ua[1] += ts2 - ts1; ua[2] += ts3 - ts2; ua[3] += ts4 - ts3; ua[4] += 1;
            }
        }
    }

    if constexpr (TRACE) tr[4] = __builtin_amdgcn_s_memtime();   // units done
    // ---- C: outputs just below an integer position: previous frame, row 1023, frac 0 ----------------
    if (split) {
        // The frames come from HBM here: every load of an output (its 128 taps x 2 channels over 8 lanes) is
        // requested before the first multiply, and the wraps of all streams form one list -- one memory
        // latency for the whole phase.  Same sums in the same order as the f32 variant below.
        const uint32_t grp = threadIdx.x >> 3, ngrp = kLsWaves * 8, gl = threadIdx.x & 7;
        uint32_t total = 0;
        for (uint32_t s = 0; s < g.count; ++s)
            if (!(plan[s].flags & kFlagReference)) total += plan[s].n_wraps;
        for (uint32_t item = grp; item < (total + ngrp - 1) / ngrp * ngrp; item += ngrp) {
            const bool live = item < total;
            uint32_t s = 0, e = live ? item : 0;
            for (; s + 1 < g.count; ++s) {
                const uint32_t nw = (plan[s].flags & kFlagReference) ? 0u : plan[s].n_wraps;
                if (e < nw) break;
                e -= nw;
            }
            const PlanLds& pl = plan[s];
            const uint32_t n = live ? wlist[s * g.wrap_cap + e] : 0;
            const uint64_t m = pl.abs_out + n;
            const int64_t v0 = static_cast<int64_t>((m / g.den) * g.num) - 1 - static_cast<int64_t>(pl.abs_consumed);
            const float4* row = reinterpret_cast<const float4*>(args.streams[g.first + s].coeffs + static_cast<size_t>(1023) * g.taps);
            gconst_f32_ptr hist = (gconst_f32_ptr)ptrs[s].hist, in = (gconst_f32_ptr)ptrs[s].in;
            const int64_t hist_fr = pl.hist_frames, span_fr = pl.hist_frames + pl.accepted;
            float4 k[4];
            float x[4][4][2];
#pragma unroll
            for (uint32_t it = 0; it < 4; ++it) {
                const uint32_t q = gl + 8 * it;
                const bool have = live && q < g.taps / 4;
                k[it] = have ? row[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    const int64_t f = v0 + 4 * q + j;
                    const bool ok = have && f >= 0 && f < span_fr;
                    const int64_t fc = ok ? f : hist_fr;   // (a frame that exists whenever any does: in[0])
                    gconst_f32_ptr p = fc < hist_fr ? hist + 2 * fc : in + 2 * (fc - hist_fr);
                    const float a0 = ok ? p[0] : 0.f, a1 = ok ? p[1] : 0.f;
                    x[it][j][0] = a0;
                    x[it][j][1] = a1;
                }
            }
#pragma unroll
            for (uint32_t c = 0; c < 2; ++c) {
                float a = 0.f;
#pragma unroll
                for (uint32_t it = 0; it < 4; ++it)
                    if (gl + 8 * it < g.taps / 4) {
                        a = fmaf(k[it].x, x[it][0][c], a);
                        a = fmaf(k[it].y, x[it][1][c], a);
                        a = fmaf(k[it].z, x[it][2][c], a);
                        a = fmaf(k[it].w, x[it][3][c], a);
                    }
                a = group_sum8(a);
                if (live && gl == 0) pl.out[static_cast<size_t>(n) * 2 + c] = a;
            }
        }
    } else {
        const uint32_t grp = threadIdx.x >> 3, ngrp = kLsWaves * 8, gl = threadIdx.x & 7;
        for (uint32_t s = 0; s < g.count; ++s) {
            const PlanLds& pl = plan[s];
            if (pl.flags & kFlagReference) continue;
            const uint32_t nw = pl.n_wraps;
            if (nw == 0) continue;
            const float4* row = reinterpret_cast<const float4*>(args.streams[g.first + s].coeffs +
                                                                static_cast<size_t>(1023) * g.taps);
            const SpanView sv{split ? nullptr : spans + s * region_dw + g.guard_frames * C, ptrs[s].hist, ptrs[s].in,
                              pl.hist_frames, pl.hist_frames + pl.accepted, C};
            for (uint32_t e = grp; e < (nw + ngrp - 1) / ngrp * ngrp; e += ngrp) {
                const bool live = e < nw;
                const uint32_t n = live ? wlist[s * g.wrap_cap + e] : 0;
                const uint64_t m = pl.abs_out + n;
                const int64_t v0 = static_cast<int64_t>((m / g.den) * g.num) - 1 -
                                   static_cast<int64_t>(pl.abs_consumed);
                for (uint32_t c = 0; c < C; ++c) {
                    float a = 0.f;
                    if (live)
                        for (uint32_t q = gl; q < g.taps / 4; q += 8) {
                            const float4 k = row[q];
                            const int64_t f = v0 + 4 * q;
                            a = fmaf(k.x, sv.at(f, c), a);
                            a = fmaf(k.y, sv.at(f + 1, c), a);
                            a = fmaf(k.z, sv.at(f + 2, c), a);
                            a = fmaf(k.w, sv.at(f + 3, c), a);
                        }
                    a = group_sum8(a);
                    if (live && gl == 0) pl.out[static_cast<size_t>(n) * C + c] = a;
                }
            }
        }
    }

    // ---- D: reference form for the streams that need it -------------------------------------------
    if constexpr (TRACE) tr[5] = __builtin_amdgcn_s_memtime();
    __syncthreads();   // (every wave's units are done: the flags they may have set are final)
    if constexpr (TRACE) tr[6] = __builtin_amdgcn_s_memtime();
    bool any_reference = false;   // (workgroup-uniform) some stream is redone in the reference's form: rare
    for (uint32_t s = 0; s < g.count; ++s)
        any_reference = any_reference || ((plan[s].flags & (kFlagReference | kFlagNonFinite)) && plan[s].n_out != 0);
    if (any_reference) {
        // phase B/C stores are acknowledged before D overwrites them -- a wait (an HBM store's round trip) and a second
        // barrier that only the workgroups with such a stream pay
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const uint32_t grp = threadIdx.x >> 3, ngrp = kLsWaves * 8, gl = threadIdx.x & 7;
        for (uint32_t s = 0; s < g.count; ++s) {
            const PlanLds& pl = plan[s];
            if (!(pl.flags & (kFlagReference | kFlagNonFinite)) || pl.n_out == 0) continue;
            const float* coeffs = args.streams[g.first + s].coeffs;
            const SpanView sv{split ? nullptr : spans + s * region_dw + g.guard_frames * C, ptrs[s].hist, ptrs[s].in,
                              pl.hist_frames, pl.hist_frames + pl.accepted, C};
            const SegLds* sg = static_cast<const SegLds*>(pl.runs);
            const uint32_t n_round = (pl.n_out + ngrp - 1) / ngrp * ngrp;
            for (uint32_t n = grp; n < n_round; n += ngrp) {
                const bool live = n < pl.n_out;
                double p = 0.0;
                if (live) {
                    uint32_t i = 0;
                    while (i + 1 < pl.n_segs && n >= sg[i].first + sg[i].count) ++i;
                    p = fma(static_cast<double>(n - sg[i].first), sg[i].inc, sg[i].p0);
                }
                const double fl = floor(p);                                   // resampler_fir.rs:544
                const double fract = p - fl;                                  // :558
                double phase_f = fract * 1024.0;                              // :562
                phase_f = phase_f < 1023.0 ? phase_f : 1023.0;
                const uint32_t phase1 = static_cast<uint32_t>(phase_f);       // :563
                const uint32_t phase2 = phase1 + 1 < 1023u ? phase1 + 1 : 1023u;  // :564
                const float frac = static_cast<float>(phase_f - static_cast<double>(phase1));  // :565
                const float omf = 1.0f - frac;                                // avx.rs:42
                const int64_t v0 = static_cast<int64_t>(fl);
                const float4* row1 = reinterpret_cast<const float4*>(coeffs + static_cast<size_t>(phase1) * g.taps);
                const float4* row2 = reinterpret_cast<const float4*>(coeffs + static_cast<size_t>(phase2) * g.taps);
                for (uint32_t c = 0; c < C; ++c) {
                    float a1 = 0.f, a2 = 0.f;
                    if (live)
                        for (uint32_t q = gl; q < g.taps / 4; q += 8) {
                            const float4 k1 = row1[q];
                            const float4 k2 = row2[q];
                            const int64_t f = v0 + 4 * q;
                            const float x0 = sv.at(f, c), x1 = sv.at(f + 1, c), x2 = sv.at(f + 2, c), x3 = sv.at(f + 3, c);
                            a1 = fmaf(k1.x, x0, a1); a2 = fmaf(k2.x, x0, a2);
                            a1 = fmaf(k1.y, x1, a1); a2 = fmaf(k2.y, x1, a2);
                            a1 = fmaf(k1.z, x2, a1); a2 = fmaf(k2.z, x2, a2);
                            a1 = fmaf(k1.w, x3, a1); a2 = fmaf(k2.w, x3, a2);
                        }
                    const float part = a1 * omf + a2 * frac;                   // per-lane lerp (avx.rs:41-45)
                    const float y = group_sum8(part);
                    if (live && gl == 0) pl.out[static_cast<size_t>(n) * C + c] = y;
                }
            }
        }
    }
    if (wave == 0 && lane < g.count) {
        const uint32_t f = plan[lane].flags;   // (phase B may have added the non-finite flag)
        const uint32_t status = ((f & kFlagRunOverflow) ? kLsStatusRunOverflow : 0u) |
                                ((f & kFlagNonFinite) ? kLsStatusNonFinite : 0u) |
                                (((f & kFlagReference) && g.periodic) ? kLsStatusAperiodic : 0u);
        if (status) args.status[g.first + lane] |= status;
    }
    if constexpr (TRACE) {
        tr[7] = __builtin_amdgcn_s_memtime();
        if (args.trace && wave < 2 && lane == 0) {
            for (int i = 0; i < 8; ++i) args.trace[(static_cast<size_t>(blockIdx.x) * 2 + wave) * 8 + i] = tr[i];
            if (wave == 1)   // wave 1's unit phases ride in the slots of wave 0 that are unused: after the stamps
                for (int i = 0; i < 5; ++i) args.trace[static_cast<size_t>(gridDim.x) * 16 + static_cast<size_t>(blockIdx.x) * 5 + i] = ua[i];
        }
    }
}

}  // namespace

LockstepGeometry lockstep_geometry(uint64_t num, uint64_t den, double ratio, uint32_t taps,
                                   uint32_t channels, uint32_t step_frames, bool allow_split) {
    static const bool exact_knob = [] { const char* e = rsmp::knob("RSMP_LS_EXACT"); return e && atoi(e) != 0; }();
    LockstepGeometry g;
    g.taps = taps;
    g.num = static_cast<uint32_t>(num);
    g.den = static_cast<uint32_t>(den);
    // With out_cap >= buffer_size_output a step never leaves more than taps - 1 frames buffered, and
    // it produces at most (buffered + new - taps + 1) / ratio + 1 frames.
    g.span_frames = taps + step_frames + 8;
    g.max_out = static_cast<uint32_t>(std::ceil(static_cast<double>(step_frames + 8) / ratio)) + 2;
    g.wrap_words = (g.max_out + 31) / 32;
    auto finish = [&](bool periodic) -> bool {
        g.periodic = periodic;
        if (!periodic) {
            g.r = g.a = 0;
            g.b = 1;
            g.row_len = g.n_tiles = 0;
            g.guard_frames = 0;
            g.region_frames = g.span_frames + 64;   // (the last staging piece may run 63 dwords past the span)
            g.cols_per_stream = 1;
            g.wrap_cap = 1;
        }
        const uint32_t want = periodic ? std::max(1u, 16u / g.cols_per_stream) : 4u;
        for (uint32_t s = std::min(want, kLsMaxSlots); s >= 1; --s) {
            const uint32_t bytes = ls_layout(s, s * g.cols_per_stream, g.wrap_words, g.wrap_cap,
                                             ls_data_bytes(g.split, g.rows, g.row_bytes, s, g.region_frames, channels)).total;
            // Two workgroups per CU: 80 KB each.  One dynamic LDS size serves the whole launch, so a group above
            // that would halve the occupancy of every group: a split image that does not fit makes way for the
            // exact-f32 layout (which drops to one stream per workgroup before it gives up on that).
            if (g.split && s * g.cols_per_stream > 16) continue;   // one image = 16 columns (a long step of a high ratio has more: f32 layout)
            if (bytes <= (s > 1 || g.split ? kLsLdsPerWorkgroup : kLsLdsLimit)) {
                g.slots = s;
                g.max_cols = s * g.cols_per_stream;
                g.lds_bytes = bytes;
                return true;
            }
        }
        return false;
    };
    if (num != 0 && den != 0 && num <= (1u << 20) && den <= (1u << 20)) {
        const uint32_t shift = static_cast<uint32_t>((15 * num + den - 1) / den);
        g.split = allow_split && !exact_knob && channels == 2;
        g.row_len = g.split ? (taps + shift + 31) / 32 * 32 : (taps + shift + 15) / 16 * 16;
        uint64_t r = (96 + den - 1) / den;
        if (r == 0) r = 1;
        const uint64_t a = num * r, b = den * r;
        if (a <= 8192 && b <= 65536 && g.row_len <= 16 * kLsMaxBlk) {
            g.r = static_cast<uint32_t>(r);
            g.a = static_cast<uint32_t>(a);
            g.b = static_cast<uint32_t>(b);
            g.n_tiles = (g.b + 15) / 16;
            g.guard_frames = g.a + (g.a & 1u);
            g.region_frames = g.guard_frames + g.span_frames + g.a + g.row_len;
            g.region_frames += g.region_frames & 1u;
            g.cols_per_stream = (g.max_out - 1) / g.b + 2;
            g.wrap_cap = g.max_out / g.den + 2;
            g.rows = static_cast<uint32_t>((static_cast<uint64_t>(g.n_tiles - 1) * 16 * g.a) / g.b) + g.row_len;
            g.row_bytes = kLsImageRowBytes;
            if (finish(true)) return g;
            if (g.split) {   // without the rows' padding (transposed reads then meet on banks: 2-4x the LDS time of a unit, still far below f32 products)
                g.row_bytes = kLsImageRowBytesPacked;
                if (finish(true)) return g;
            }
            if (g.split) {   // the image does not fit: exact-f32 layout
                g.split = false;
                g.row_len = (taps + shift + 15) / 16 * 16;
                if (g.row_len <= 16 * kLsMaxBlk && finish(true)) return g;
            }
        }
    }
    g.split = false;
    if (!finish(false)) g.lds_bytes = 0;   // caller reports the failure
    return g;
}

PeriodicGeometry lockstep_class_geometry(const LockstepGeometry& g) {
    PeriodicGeometry p;
    p.ok = g.periodic;
    p.a = g.a;
    p.b = g.b;
    p.den = g.den;
    p.taps = g.taps;
    p.row_len = g.row_len;
    p.n_tiles = g.n_tiles;
    p.mfma = g.split ? 3 : 1;   // A-operand order of v_mfma_f32_16x16x4_f32, or the split table of fir_split.hip
    p.planes = g.split ? 2 : 0;
    p.inline_wraps = false;
    return p;
}

hipError_t launch_fir_lockstep(const LockstepArgs& args, uint32_t n_groups, uint32_t max_lds_bytes,
                               hipStream_t stream) {
    if (n_groups == 0) return hipSuccess;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    static std::mutex mu;
    static std::map<int, bool> granted;
    {
        std::lock_guard<std::mutex> lock(mu);
        bool& have = granted[device];
        if (!have) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(fir_lockstep_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kLsLdsLimit);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(fir_lockstep_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kLsLdsLimit);
            if (e != hipSuccess) return e;
            have = true;
        }
    }
    static const char* trace_path = rsmp::knob("RSMP_LS_TRACE");
    if (trace_path) {   // diagnostic: one synchronous traced step, phase clocks written to the file
        LockstepArgs a = args;
        const size_t words = static_cast<size_t>(n_groups) * 21;
        unsigned long long* d = nullptr;
        if (hipMalloc(&d, words * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemset(d, 0, words * 8);
        a.trace = d;
        hipLaunchKernelGGL(fir_lockstep_kernel<true>, dim3(n_groups), dim3(kLsWaves * 64), max_lds_bytes, stream, a);
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h(words);
        (void)hipMemcpy(h.data(), d, words * 8, hipMemcpyDeviceToHost);
        (void)hipFree(d);
        if (FILE* f = fopen(trace_path, "a")) {
            for (uint32_t b = 0; b < n_groups; ++b) {
                fprintf(f, "%u", b);
                for (int w = 0; w < 2; ++w)
                    for (int i = 1; i < 8; ++i) fprintf(f, " %lld", (long long)(h[(b * 2 + w) * 8 + i] - h[(b * 2 + w) * 8]));
                for (int i = 0; i < 5; ++i) fprintf(f, " %lld", (long long)h[static_cast<size_t>(n_groups) * 16 + b * 5 + i]);
                fprintf(f, "\n");
            }
            fclose(f);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(fir_lockstep_kernel<false>, dim3(n_groups), dim3(kLsWaves * 64), max_lds_bytes, stream, args);
    return hipGetLastError();
}

}  // namespace rsmp
