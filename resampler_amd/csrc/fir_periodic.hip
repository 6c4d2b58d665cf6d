// fir_periodic.hip -- throughput FIR kernel for rational rate pairs on gfx950 (see fir_periodic.h
// for the idea).  Replaces the same reference code as fir_generic.hip
// (src/resampler_fir.rs:542-590 + src/fir/avx.rs:5-61) for launches long enough to fill waves
// with whole periods.
//
// Workgroup = `waves` wave64s sharing one staged input span of `pw` periods (two workgroups per
// CU: one stages / stores while the other computes):
//   stage   : [hist|in] frames (q0*a ... (q0+pw)*a + row_len) -> LDS, one row per period with an
//             odd frame stride (the per-lane strided reads below are then bank-conflict free),
//             zero filled outside the stream; branch-free so the loads of a row are all in flight;
//   compute : wave w takes class tiles w, w+waves, ...; lane = (period, channel group).  Per tap:
//             one ds_read of the lane's sample(s), 8 wave-uniform coefficients through the scalar
//             cache (s_load_dwordx16 = 2 taps), 8 v_pk_fma_f32 with an SGPR operand (packed fp32
//             is the only way to the 128 FMA/clk/CU peak on gfx950).  No cross-lane traffic;
//   wrap    : tiles holding a class whose exact position is an integer carry a 9th accumulator
//             (row 1023, previous frame) and pick per lane from the launch's wrap bitmap;
//   store   : a 4x4 DPP transpose inside each lane quad turns "lane = period, 64 B of output
//             each" into 64 B-contiguous quads, so one store instruction issues 16 x 64 B requests
//             instead of 64 x 16 B (L2 request rate, not bytes, limits scattered stores).
// HBM traffic = input span once per workgroup (+ row_len halo) + output once; the class table
// (<= a few hundred KB) stays in L2 / scalar cache.
#include "fir_periodic.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#include "common.h"
#include "filter_design.h"

namespace rsmp {

namespace {

constexpr uint32_t kLdsTwoPerCu = 80 * 1024;   // two workgroups per CU
constexpr uint32_t kLdsMax = 160 * 1024;

struct GeoArgs {
    uint32_t a, b, r, row_len, n_tiles, lp, pw, row_stride, waves, channels, xprev_len;
    uint32_t inline_wraps;
    uint32_t debug;  // RSMP_FIR_DEBUG: bit0 skip staging, bit1 skip the tap loops (timing only)
    uint32_t stagger_ticks;  // one-time start delay of the second workgroup slot (100 MHz ticks)
    unsigned long long* trace;  // RSMP_FIR_TRACE diagnostic build only: 6 u64 per workgroup
    unsigned long long* wtrace; // RSMP_FIR_WTRACE: kWtraceSlots timestamped events per wave
    uint32_t blocks_per_stream, total_items;
    unsigned long long* work_counter;   // launch-wide item queue (monotonic)
    unsigned long long work_base;       // its value before this launch
};

constexpr uint32_t kWtraceSlots = 160, kWtraceWaves = 12;

typedef const float __attribute__((address_space(4)))* const_f32_ptr;   // scalar-cache loads
typedef const float __attribute__((address_space(1)))* gconst_f32_ptr;  // global (not flat) loads
typedef float __attribute__((address_space(1)))* g_f32_ptr;
typedef const uint32_t __attribute__((address_space(1)))* gconst_u32_ptr;
typedef const uint32_t __attribute__((address_space(4)))* const_u32_ptr;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Copies a wave-uniform, read-only POD through the scalar cache (s_load) into registers.
template <class T>
__device__ __forceinline__ T load_uniform(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized PODs only");
    T v;
    const_u32_ptr src = (const_u32_ptr)p;
    uint32_t* dst = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) dst[i] = src[i];
    return v;
}

template <int CG> struct Acc {
    float v[kClassTile][CG];
    float w[CG];  // wrap variant of one column
};

template <int CG>
__device__ __forceinline__ void load_x(float (&xs)[CG], const float* p) {
    if constexpr (CG == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        xs[0] = t.x;
        xs[1] = t.y;
    } else {
        xs[0] = p[0];
    }
}

// `count` taps: sample(s) from LDS (stride `cstride` dwords per frame), 8 coefficients per tap
// (+1 for the wrap variant) from the class table through scalar loads.
template <int CG, bool WRAP>
__device__ __forceinline__ void accumulate(Acc<CG>& acc, const float* __restrict__ x,
                                           uint32_t cstride, const_f32_ptr g, const_f32_ptr gw,
                                           uint32_t count) {
    uint32_t m = 0;
    for (; m + 4 <= count; m += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float xs[CG];
            load_x<CG>(xs, x + (m + u) * cstride);
#pragma unroll
            for (int i = 0; i < (int)kClassTile; ++i) {
                const float c = g[(m + u) * kClassTile + i];
#pragma unroll
                for (int k = 0; k < CG; ++k) acc.v[i][k] = fmaf(c, xs[k], acc.v[i][k]);
            }
            if constexpr (WRAP) {
                const float c = gw[m + u];
#pragma unroll
                for (int k = 0; k < CG; ++k) acc.w[k] = fmaf(c, xs[k], acc.w[k]);
            }
        }
    }
    for (; m < count; ++m) {
        float xs[CG];
        load_x<CG>(xs, x + m * cstride);
#pragma unroll
        for (int i = 0; i < (int)kClassTile; ++i) {
            const float c = g[m * kClassTile + i];
#pragma unroll
            for (int k = 0; k < CG; ++k) acc.v[i][k] = fmaf(c, xs[k], acc.v[i][k]);
        }
        if constexpr (WRAP) {
            const float c = gw[m];
#pragma unroll
            for (int k = 0; k < CG; ++k) acc.w[k] = fmaf(c, xs[k], acc.w[k]);
        }
    }
}

__device__ __forceinline__ float dpp_quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}

// ---- fast path (2 channels, both per lane): packed-FMA chunks of 8 taps ---------------------------
typedef const v2f __attribute__((address_space(4)))* const_v2f_ptr;

// acc[2p] += c_p.lo * x, acc[2p+1] += c_p.hi * x for the four coefficient pairs of one tap: src0 is
// an SGPR pair whose low / high half is broadcast to both lanes of the packed FMA by op_sel.
__device__ __forceinline__ void pk_fma8(v2f (&acc)[8], v2f c0, v2f c1, v2f c2, v2f c3, v2f x) {
    asm("v_pk_fma_f32 %0, %8, %12, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %8, %12, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %2, %9, %12, %2 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %3, %9, %12, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %4, %10, %12, %4 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %5, %10, %12, %5 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %6, %11, %12, %6 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %7, %11, %12, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),
          "+v"(acc[6]), "+v"(acc[7])
        : "s"(c0), "s"(c1), "s"(c2), "s"(c3), "v"(x));
}

// NT taps (8 or 4): 4*NT coefficient pairs (NT/2 x s_load_dwordx16) against NT frames.  One wait
// covers 8*NT packed FMAs; that is what lets a handful of waves per SIMD hide the scalar-cache miss
// latency (each line of the table is touched by one wave only).  NT = 8 needs 64 SGPRs for the
// coefficients (12-wave workgroups, 6 waves per SIMD); NT = 4 halves that and leaves room for
// 16-wave workgroups at 8 waves per SIMD.
template <bool WRAP, int NT>
__device__ __forceinline__ void taps(v2f (&acc)[8], v2f& accw, const v2f (&x)[NT],
                                     const_v2f_ptr gc, const_f32_ptr gwc) {
    v2f c[4 * NT];
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) c[i] = gc[i];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        pk_fma8(acc, c[4 * u], c[4 * u + 1], c[4 * u + 2], c[4 * u + 3], x[u]);
        if constexpr (WRAP) {
            const float w = gwc[u];
            accw.x = fmaf(w, x[u].x, accw.x);
            accw.y = fmaf(w, x[u].y, accw.y);
        }
    }
}

template <bool WRAP, int NT>
__device__ __forceinline__ void tile_taps_c2(v2f (&acc)[8], v2f& accw, const float* rowA,
                                             const float* rowB, uint32_t n1, uint32_t row_len,
                                             const_f32_ptr g, const_f32_ptr gw) {
    const uint32_t n_chunks = row_len / NT;
    const uint32_t chunks_a = n1 / NT;          // chunks entirely inside the lane's own row
    const_v2f_ptr gc = (const_v2f_ptr)g;
    uint32_t c = 0;
    for (; c < chunks_a; ++c) {
        v2f x[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) x[u] = *reinterpret_cast<const v2f*>(rowA + 2 * NT * c + 2 * u);
        taps<WRAP, NT>(acc, accw, x, gc + 4 * NT * c, gw + NT * c);
    }
    if (c < n_chunks && (n1 % NT)) {            // the chunk that straddles the two rows
        v2f x[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const uint32_t m = NT * c + u;
            const float* px = m < n1 ? rowA + 2 * m : rowB + 2 * (m - n1);
            x[u] = *reinterpret_cast<const v2f*>(px);
        }
        taps<WRAP, NT>(acc, accw, x, gc + 4 * NT * c, gw + NT * c);
        ++c;
    }
    for (; c < n_chunks; ++c) {
        const float* pb = rowB + 2 * (NT * c - n1);
        v2f x[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) x[u] = *reinterpret_cast<const v2f*>(pb + 2 * u);
        taps<WRAP, NT>(acc, accw, x, gc + 4 * NT * c, gw + NT * c);
    }
}

// Up to 768 threads = 12 waves (3 per SIMD); two workgroups per CU -> 6 waves per SIMD -> at
// most 80 VGPRs.
// C2 = true: exactly two channels, both handled by one lane (CG == 2) -- the headline config.
template <int CG, bool C2, int NT>
__global__ __launch_bounds__(NT == 8 ? 768 : 1024, NT == 8 ? 6 : 8) void fir_periodic_kernel(const FirStreamDesc* __restrict__ descs,
                                                              GeoArgs geo) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // Persistent workgroups: the grid is two workgroups per CU and each walks the launch's work
    // items (stream, period block) with the grid as stride.  A fresh dispatch per block cost
    // ~10 us of empty LDS slot between workgroups (measured with RSMP_FIR_TRACE) -- a third of
    // each slot's time.
    //
    // All workgroups start together and take equally long, so the two workgroups sharing a CU
    // would stage (HBM busy, VALU idle) and compute (VALU busy, HBM idle) in lockstep.  Delaying
    // the second dispatch round once puts the pairs out of phase for the rest of the launch:
    // one streams while the other computes.  Dispatch order only affects speed, never results.
    unsigned long long t_trace[4] = {0, 0, 0, 0};
    if (geo.trace) t_trace[0] = __builtin_amdgcn_s_memrealtime();
    if (geo.stagger_ticks && blockIdx.x >= gridDim.x / 2) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < geo.stagger_ticks) __builtin_amdgcn_s_sleep(16);
    }

    const uint32_t C = C2 ? 2u : geo.channels;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // RSMP_FIR_WTRACE (diagnostic): per-wave event log, (100 MHz timestamp << 8) | tag
    uint32_t ev_cursor = 0;
    auto event = [&](uint32_t tag) {
        if (geo.wtrace && ev_cursor < kWtraceSlots && wave < kWtraceWaves) {
            const unsigned long long v = (__builtin_amdgcn_s_memrealtime() << 8) | tag;
            if (lane == 0)
                geo.wtrace[(static_cast<size_t>(blockIdx.x) * kWtraceWaves + wave) * kWtraceSlots + ev_cursor] = v;
            ++ev_cursor;
        }
    };
    uint32_t* tile_counter = reinterpret_cast<uint32_t*>(lds);   // next unclaimed class tile
    float* __restrict__ xprev = lds + 4;               // [pw][C]: the frame in front of each period
    float* __restrict__ rows = lds + geo.xprev_len;    // [pw + 1][row_stride]

  // Work items are claimed from a launch-wide queue (one 64-bit counter that only ever grows; the
  // host passes the value it had before this launch).  Static striding left the younger of the
  // two workgroups of a CU -- which loses VALU arbitration to the older one -- with a third of
  // its items still to do after its neighbour had finished.  The claim for the next item is made
  // while the current one is being staged, so it is never on the critical path.
  uint32_t* next_item = reinterpret_cast<uint32_t*>(lds) + 1;
  auto claim = [&]() -> uint32_t {
      const unsigned long long t = atomicAdd(geo.work_counter, 1ull) - geo.work_base;
      return t < geo.total_items ? static_cast<uint32_t>(t) : 0xFFFFFFFFu;
  };
  if (threadIdx.x == 0) *next_item = claim();
  __syncthreads();
  uint32_t item = *next_item;
  while (item != 0xFFFFFFFFu) {
    const uint32_t stream_idx = item / geo.blocks_per_stream;
    const uint32_t block_idx = item - stream_idx * geo.blocks_per_stream;
    // The descriptor is wave-uniform and read-only: fetch it through the scalar cache.
    const FirStreamDesc d = load_uniform(descs + stream_idx);
    const uint32_t n_out = d.n_out;
    const uint64_t abs_out = d.abs_out;
    const uint64_t q_first = abs_out / geo.b;
    const uint64_t q0 = q_first + static_cast<uint64_t>(block_idx) * geo.pw;
    const bool valid = n_out != 0 && q0 * geo.b < abs_out + n_out;
    // launch-relative index of output (period q0, class 0); fits int32 (n_out < 2^31)
    const int32_t n_block0 = static_cast<int32_t>(static_cast<int64_t>(q0 * geo.b) -
                                                  static_cast<int64_t>(abs_out));
    event(1);          // item begins (waiting for the other waves)
    __syncthreads();   // every wave has read `item` and is done with the previous LDS image
    event(2);
    if (threadIdx.x == 0) *next_item = claim();
    if (!valid) {      // padding item of a ragged batch
        __syncthreads();
        item = *next_item;
        continue;
    }

    // ---- stage ---------------------------------------------------------------------------------
    // LDS-DMA (global_load_lds, 4 B per lane): the rows region is filled 256 B per wave
    // instruction straight from [hist|in] with no VGPR round trip, so every piece of a wave
    // (~30) is in flight at once and the other resident workgroup computes meanwhile.  Source
    // addresses are clamped into the stream; an edge workgroup zeroes the out-of-stream part
    // afterwards.
    if (!(geo.debug & 1)) {
        const int64_t hist_values = static_cast<int64_t>(d.hist_frames) * C;
        const int64_t total_values = hist_values + static_cast<int64_t>(d.in_frames) * C;  // > 0
        gconst_f32_ptr hist = (gconst_f32_ptr)d.hist;
        gconst_f32_ptr in = (gconst_f32_ptr)d.in;
        // virtual value index of the span's first sample: absolute frame q0*a minus the frames
        // retired before this launch, times C
        const int64_t w_span =
            (static_cast<int64_t>(q0 * geo.a) - static_cast<int64_t>(d.abs_consumed)) * C;
        const uint32_t row_values = geo.a * C;
        const bool flat = geo.row_stride == row_values;   // odd a: rows are back to back
        const uint32_t region = (geo.pw + 1) * geo.row_stride;
        // virtual value index feeding LDS dword L of the rows region (pad dwords of even-a rows
        // re-read the next row's first frame; they are never used)
        auto w_of = [&](uint32_t L) -> int64_t {
            if (flat) return w_span + L;
            const uint32_t p = L / geo.row_stride;
            return w_span + static_cast<int64_t>(p) * row_values + (L - p * geo.row_stride);
        };
        auto src_of = [&](int64_t w) -> gconst_f32_ptr {
            const int64_t wc = w < 0 ? 0 : (w >= total_values ? total_values - 1 : w);
            return wc < hist_values ? hist + wc : in + (wc - hist_values);
        };
        typedef __attribute__((address_space(3))) void* lds_void_ptr;
        const int64_t span_values = static_cast<int64_t>(geo.pw + 1) * row_values;
        const bool edge = w_span < 0 || w_span + span_values > total_values;
        if (flat && w_span >= hist_values && !edge) {
            // interior workgroup: the span is one contiguous piece of `in`
            gconst_f32_ptr src = in + (w_span - hist_values) + lane;
            for (uint32_t base = wave * 64; base < region; base += geo.waves * 64)
                if (base + lane < region)
                    __builtin_amdgcn_global_load_lds(src + base, (lds_void_ptr)(rows + base), 4, 0, 0);
        } else {
            for (uint32_t base = wave * 64; base < region; base += geo.waves * 64) {
                const uint32_t L = base + lane;
                if (L < region)
                    __builtin_amdgcn_global_load_lds(src_of(w_of(L)), (lds_void_ptr)(rows + base), 4, 0, 0);
            }
        }
        for (uint32_t e = threadIdx.x; e < geo.pw * C; e += blockDim.x) {
            const uint32_t p = C == 1 ? e : e / C;
            const int64_t w = w_span + static_cast<int64_t>(p) * row_values - C + (e - p * C);
            float val = *src_of(w);
            if (w < 0 || w >= total_values) val = 0.f;
            xprev[e] = val;
        }
        if (edge) {   // workgroup-uniform
            __builtin_amdgcn_s_waitcnt(0);   // own DMA pieces have landed
            __syncthreads();
            for (uint32_t L = threadIdx.x; L < region; L += blockDim.x) {
                const int64_t w = w_of(L);
                if (w < 0 || w >= total_values) rows[L] = 0.f;
            }
        }
    }
    if (threadIdx.x == 0) *tile_counter = 0;
    event(3);          // staging issued
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0): LDS-DMA is tracked by vmcnt
    event(4);          // own pieces landed
    __syncthreads();
    event(5);          // everyone's pieces landed

    if (geo.trace) t_trace[1] = __builtin_amdgcn_s_memrealtime();
    // ---- compute -------------------------------------------------------------------------------
    const uint32_t pl = C2 ? lane : lane / geo.lp;   // period of this lane inside the block
    const uint32_t gi = C2 ? 0u : lane - pl * geo.lp;  // channel group of this lane
    const bool lane_on = pl < geo.pw;
    const uint32_t pl_c = lane_on ? pl : 0;          // idle lanes shadow lane 0 (no stores)
    const float* __restrict__ lane_row = rows + pl_c * geo.row_stride + gi * CG;
    const_f32_ptr table = (const_f32_ptr)(d.class_coef);
    const_f32_ptr wtable = (const_f32_ptr)(d.class_wrap_coef);
    const TileMeta* metas = static_cast<const TileMeta*>(d.class_meta);
    gconst_u32_ptr wrap_bits = (gconst_u32_ptr)d.wrap_bits;
    g_f32_ptr out = (g_f32_ptr)d.out;
    // bit index of (period q0, class 0) in the wrap bitmap
    const int32_t k_block0 = static_cast<int32_t>(static_cast<int64_t>(q0 * geo.r) -
                                                  static_cast<int64_t>(d.wrap_k0));

    // Class tiles are claimed dynamically: a workgroup's waves are spread unevenly over the four
    // SIMDs (and share them with the other resident workgroup), so a static split leaves the
    // least loaded SIMD idle while the most loaded one finishes.
    // The claim of the next tile (an LDS atomic) and the fetch of its descriptor are issued while
    // the current tile computes, so a tile switch exposes neither latency.
    uint32_t t_claim = 0;
    if (lane == 0) t_claim = atomicAdd(tile_counter, 1u);
    uint32_t t = __builtin_amdgcn_readfirstlane(t_claim);
    TileMeta tm_cur = load_uniform(metas + (t < geo.n_tiles ? t : 0));
    while (t < geo.n_tiles) {
        event(6);      // tile begins
        if (lane == 0) t_claim = atomicAdd(tile_counter, 1u);   // next tile, consumed at the bottom
        const uint32_t j0 = t * kClassTile;
        const TileMeta tm = tm_cur;
        const uint32_t ob = tm.base;
        const_f32_ptr g = table + static_cast<size_t>((geo.debug & 32) ? 0 : t) * geo.row_len * kClassTile;
        const_f32_ptr gw = wtable + static_cast<size_t>(t) * geo.row_len;
        const bool has_wrap = geo.inline_wraps && tm.wrap_col >= 0;
        // window [ob, ob+row_len) of the lane's period row, spilling into the next row
        const uint32_t n1 = geo.a - ob < geo.row_len ? geo.a - ob : geo.row_len;
        const int32_t n_lane0 = n_block0 + static_cast<int32_t>(pl_c * geo.b + j0);  // class j0
        const int32_t n_limit = static_cast<int32_t>(n_out);

        float av[kClassTile][CG];
        float aw[CG];
        if constexpr (C2) {
            v2f acc[8], accw = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = v2f{0.f, 0.f};
            if (!(geo.debug & 2)) {
                if (has_wrap)
                    tile_taps_c2<true, NT>(acc, accw, lane_row + ob * 2, lane_row + geo.row_stride, n1,
                                       geo.row_len, g, gw);
                else
                    tile_taps_c2<false, NT>(acc, accw, lane_row + ob * 2, lane_row + geo.row_stride, n1,
                                        geo.row_len, g, gw);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { av[i][0] = acc[i].x; av[i][1] = acc[i].y; }
            aw[0] = accw.x;
            aw[1] = accw.y;
        } else {
            Acc<CG> acc;
#pragma unroll
            for (int i = 0; i < (int)kClassTile; ++i)
#pragma unroll
                for (int k = 0; k < CG; ++k) acc.v[i][k] = 0.f;
#pragma unroll
            for (int k = 0; k < CG; ++k) acc.w[k] = 0.f;
            if (!(geo.debug & 2)) {
                if (has_wrap) {
                    accumulate<CG, true>(acc, lane_row + ob * C, C, g, gw, n1);
                    if (n1 < geo.row_len)
                        accumulate<CG, true>(acc, lane_row + geo.row_stride, C, g + n1 * kClassTile,
                                             gw + n1, geo.row_len - n1);
                } else {
                    accumulate<CG, false>(acc, lane_row + ob * C, C, g, gw, n1);
                    if (n1 < geo.row_len)
                        accumulate<CG, false>(acc, lane_row + geo.row_stride, C, g + n1 * kClassTile,
                                              gw + n1, geo.row_len - n1);
                }
            }
#pragma unroll
            for (int i = 0; i < (int)kClassTile; ++i)
#pragma unroll
                for (int k = 0; k < CG; ++k) av[i][k] = acc.v[i][k];
#pragma unroll
            for (int k = 0; k < CG; ++k) aw[k] = acc.w[k];
        }

        event(7);      // taps done
        const uint32_t t_next = __builtin_amdgcn_readfirstlane(t_claim);
        const TileMeta tm_next = load_uniform(metas + (t_next < geo.n_tiles ? t_next : 0));
        if (has_wrap) {
            if (tm.extra_col != -2) {
                float xs[CG];
                load_x<CG>(xs, tm.extra_col >= 0 ? lane_row + tm.extra_col * C
                                                 : xprev + pl_c * C + gi * CG);
#pragma unroll
                for (int k = 0; k < CG; ++k) aw[k] = fmaf(tm.extra_coef, xs[k], aw[k]);
            }
            const int32_t nw = n_lane0 + tm.wrap_col;
            bool take = false;
            if (nw >= 0 && nw < n_limit) {
                const uint32_t K = static_cast<uint32_t>(k_block0 + static_cast<int32_t>(pl_c * geo.r + tm.wrap_jd));
                take = (wrap_bits[K >> 5] >> (K & 31)) & 1u;
            }
#pragma unroll
            for (int i = 0; i < (int)kClassTile; ++i)
                if (i == tm.wrap_col && take) {
#pragma unroll
                    for (int k = 0; k < CG; ++k) av[i][k] = aw[k];
                }
        }

        // ---- store ---------------------------------------------------------------------------
        bool done = false;
        if (geo.debug & 16) {   // timing only: keep the sums alive with one conditional store
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < (int)kClassTile; ++i)
#pragma unroll
                for (int k = 0; k < CG; ++k) s += av[i][k];
            if (s == 12345.678f) out[0] = s;
            done = true;
        }
        if constexpr (C2) {
            // Quad transpose: lane r of a quad ends up with quarter r (2 frames = 16 B) of the four
            // periods of the quad; store s then covers period quad_base + s contiguously.
            const bool full_tile = j0 + kClassTile <= geo.b;
            const bool mine_full = n_lane0 >= 0 && n_lane0 + (int32_t)kClassTile <= n_limit;
            const bool mine_none = n_lane0 >= n_limit || n_lane0 + (int32_t)kClassTile <= 0 || !lane_on;
            const bool partial = !(mine_full || mine_none);
            if (full_tile && !__any(partial)) {
                v4f B[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    B[k] = v4f{av[2 * k][0], av[2 * k][1], av[2 * k + 1][0], av[2 * k + 1][1]};
                const bool odd = lane & 1, hi = lane & 2;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const v4f s = odd ? B[2 * p] : B[2 * p + 1];
                    const v4f r4 = v4f{dpp_quad_xor1(s.x), dpp_quad_xor1(s.y), dpp_quad_xor1(s.z),
                                       dpp_quad_xor1(s.w)};
                    if (odd) B[2 * p] = r4; else B[2 * p + 1] = r4;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const v4f s = hi ? B[k] : B[k + 2];
                    const v4f r4 = v4f{dpp_quad_xor2(s.x), dpp_quad_xor2(s.y), dpp_quad_xor2(s.z),
                                       dpp_quad_xor2(s.w)};
                    if (hi) B[k] = r4; else B[k + 2] = r4;
                }
                const uint32_t quad_base = lane & ~3u, r = lane & 3u;
                const int32_t n_quad0 = n_block0 + static_cast<int32_t>(quad_base * geo.b + j0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int32_t ns = n_quad0 + s * static_cast<int32_t>(geo.b);
                    if (quad_base + s < geo.pw && ns >= 0 && ns + (int32_t)kClassTile <= n_limit) {
                        typedef v4f __attribute__((address_space(1), aligned(8)))* g_f4a8_ptr;
                        *((g_f4a8_ptr)(out + static_cast<size_t>(ns) * 2 + r * 4)) = B[s];
                    }
                }
                done = true;
            }
        }
        if (!done && lane_on && !(geo.debug & 16)) {
#pragma unroll
            for (int i = 0; i < (int)kClassTile; ++i) {
                const int32_t n = n_lane0 + i;
                if (j0 + i < geo.b && n >= 0 && n < n_limit) {
                    g_f32_ptr o = out + static_cast<size_t>(n) * C + gi * CG;
                    if constexpr (CG == 2) {
                        typedef v2f __attribute__((address_space(1)))* g_f2_ptr;
                        *((g_f2_ptr)o) = v2f{av[i][0], av[i][1]};
                    } else {
                        o[0] = av[i][0];
                    }
                }
            }
        }
        t = t_next;
        tm_cur = tm_next;
        event(8);      // stores issued
    }
    item = *next_item;   // written before the barrier that preceded the tile loop
  }   // items
    if (geo.trace) {
        t_trace[2] = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0);   // stores acknowledged
        t_trace[3] = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            unsigned long long* rec = geo.trace + 6ull * blockIdx.x;
            unsigned hw_id, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            rec[0] = t_trace[0]; rec[1] = t_trace[1]; rec[2] = t_trace[2]; rec[3] = t_trace[3];
            rec[4] = hw_id; rec[5] = xcc;
        }
    }
}

// Outputs whose f64 position fell just below an integer: previous frame, row 1023, frac 0
// (resampler_fir.rs:544, :562-565).  8 lanes per output, as fir_generic.  Used only when the
// geometry cannot take the wrap variant inline (den < 8).
__global__ __launch_bounds__(256) void fir_wrap_fixup_kernel(const FirStreamDesc* __restrict__ descs) {
    const FirStreamDesc& d = descs[blockIdx.y];
    const uint32_t g = threadIdx.x & 7;
    const uint32_t entry = blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool live = entry < d.n_wraps;
    const uint32_t n = live ? d.wraps[entry] : 0;
    const uint64_t m = d.abs_out + n;
    const int64_t exact = static_cast<int64_t>((m / d.den) * d.num);  // m % den == 0
    const int64_t v0 = exact - 1 - static_cast<int64_t>(d.abs_consumed);
    const uint32_t taps = d.taps, C = d.channels;
    const float4* __restrict__ row =
        reinterpret_cast<const float4*>(d.coeffs + static_cast<size_t>(1023) * taps);
    const int64_t hist_frames = d.hist_frames;
    for (uint32_t c = 0; c < C; ++c) {
        float a = 0.f;
        if (live) {
            for (uint32_t qd = g; qd < taps / 4; qd += 8) {
                const float4 k = row[qd];
                const float kk[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t v = v0 + 4 * qd + u;
                    const float x = v < hist_frames
                                        ? d.hist[static_cast<size_t>(v) * C + c]
                                        : d.in[static_cast<size_t>(v - hist_frames) * C + c];
                    a = fmaf(kk[u], x, a);
                }
            }
        }
        a += __shfl_xor(a, 4, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 1, 64);
        if (live && g == 0) d.out[static_cast<size_t>(n) * C + c] = a;
    }
}

// LDS prefix: [4 dwords: dynamic tile counter] [pw][C] previous-frame samples, 16-byte multiple.
uint32_t xprev_len_of(uint32_t pw, uint32_t channels) { return 4 + (pw * channels + 3) / 4 * 4; }

GeoArgs to_args(const PeriodicGeometry& g) {
    static const uint32_t debug = [] {
        const char* e = getenv("RSMP_FIR_DEBUG");
        return e ? static_cast<uint32_t>(atoi(e)) : 0u;
    }();
    static const uint32_t stagger = [] {
        const char* e = getenv("RSMP_FIR_STAGGER_US");
        return static_cast<uint32_t>((e ? atof(e) : 12.0) * 100.0);
    }();
    const uint32_t channels = g.lp * g.cg;
    return GeoArgs{g.a, g.b, g.b / g.den, g.row_len, g.n_tiles, g.lp, g.pw, g.row_stride, g.waves,
                   channels, xprev_len_of(g.pw, channels), g.inline_wraps ? 1u : 0u, debug, stagger, nullptr, nullptr, 0u, 0u, nullptr, 0ull};
}

// Device class tables, shared by every stream on a device with the same polyphase table, rate
// pair, geometry and drift.
struct ClassTableKey {
    int device;
    const void* table;
    uint32_t den, a, b, row_len;
    uint64_t drift_bits;
    bool operator<(const ClassTableKey& o) const {
        return std::tie(device, table, den, a, b, row_len, drift_bits) <
               std::tie(o.device, o.table, o.den, o.a, o.b, o.row_len, o.drift_bits);
    }
};
struct ClassTableCache {
    std::mutex mu;
    std::map<ClassTableKey, ClassTable> tables;
};
ClassTableCache& class_cache() {
    static ClassTableCache* c = new ClassTableCache;
    return *c;
}

constexpr double kDriftQuantum = 2e-9;  // positions this close share a class table

inline uint32_t class_offset(const PeriodicGeometry& g, uint32_t j) {
    return static_cast<uint32_t>((static_cast<uint64_t>(j) * g.a) / g.b);
}

}  // namespace

PeriodicGeometry periodic_geometry(uint64_t num, uint64_t den, uint32_t taps, uint32_t channels) {
    PeriodicGeometry g;
    if (num == 0 || den == 0 || channels == 0 || channels > 64) return g;
    if (num > (1u << 20) || den > (1u << 20)) return g;
    // max in-tile shift: off(j) = floor(j*num/den); tiles start at multiples of 8.
    const uint32_t shift = static_cast<uint32_t>((7 * num + den - 1) / den);
    g.taps = taps;
    g.den = static_cast<uint32_t>(den);
    g.row_len = (taps + shift + 7) / 8 * 8;   // whole 8-tap chunks (fir_periodic_kernel)
    // super period: a >= row_len (a window spans at most two rows) and b >= 8
    uint64_t r = (g.row_len + num - 1) / num;
    if (den * r < kClassTile) r = (kClassTile + den - 1) / den;
    const uint64_t a = num * r, b = den * r;
    if (a > 4096 || b > (1u << 16)) return g;
    g.a = static_cast<uint32_t>(a);
    g.b = static_cast<uint32_t>(b);
    g.n_tiles = (g.b + kClassTile - 1) / kClassTile;
    g.inline_wraps = den >= kClassTile;

    auto fit = [&](uint32_t cg) -> bool {
        if (channels % cg != 0) return false;
        const uint32_t lp = channels / cg;
        if (lp > 64) return false;
        const uint32_t pw_max = 64 / lp;
        // odd number of frames per row: the lane stride then hits every LDS bank once
        const uint32_t stride = (g.a | 1u) * channels;
        const uint32_t row_bytes = stride * 4;
        const uint32_t fixed = (64 * channels + 16) * 4;  // xprev
        auto rows_in = [&](uint32_t budget) -> uint32_t {
            if (budget <= fixed + 2 * row_bytes) return 0;
            return (budget - fixed) / row_bytes - 1;
        };
        uint32_t pw = rows_in(kLdsTwoPerCu);
        if (pw * 4 < pw_max * 3) pw = rows_in(kLdsMax);  // < 75% of the lanes: use the whole LDS
        if (pw > pw_max) pw = pw_max;
        if (pw * 2 < pw_max || pw == 0) return false;
        g.cg = cg;
        g.lp = lp;
        g.pw = pw;
        g.row_stride = stride;
        g.lds_bytes = (xprev_len_of(pw, channels) + (pw + 1) * stride) * 4;
        return true;
    };
    if (!fit(2) && !fit(1)) return g;
    // waves per workgroup: a multiple of the 4 SIMDs, at most 12 (__launch_bounds__(768, 6));
    // tiles are claimed dynamically, so the count need not divide n_tiles
    uint32_t best = g.n_tiles >= 12 ? 12u : (g.n_tiles >= 8 ? 8u : 4u);
    if (const char* e = getenv("RSMP_FIR_WAVES")) {   // tuning knob (4..12)
        const int w = atoi(e);
        if (w >= 1 && w <= 12) best = static_cast<uint32_t>(w);
    }
    g.waves = best;
    g.ok = true;
    return g;
}

bool periodic_supported(const FirMirror& m, size_t channels, size_t taps, int kernel_mode) {
    if (kernel_mode == RSMP_FIR_KERNEL_GENERIC) return false;
    if (!m.periodic_ok()) return false;
    return periodic_geometry(m.num(), m.den(), static_cast<uint32_t>(taps),
                             static_cast<uint32_t>(channels)).ok;
}

bool periodic_worthwhile(const FirMirror& planned, size_t produced_frames, int kernel_mode) {
    if (kernel_mode == RSMP_FIR_KERNEL_PERIODIC) return produced_frames > 0;
    // AUTO: a launch shorter than a few workgroup spans leaves most lanes idle.
    (void)planned;
    return produced_frames >= 16384;
}

uint32_t periodic_blocks(const PeriodicGeometry& geo, uint64_t abs_out, uint32_t n_out) {
    if (n_out == 0) return 0;
    const uint64_t q_first = abs_out / geo.b;
    const uint64_t q_last = (abs_out + n_out - 1) / geo.b;
    return static_cast<uint32_t>((q_last - q_first) / geo.pw + 1);
}

size_t periodic_wrap_words(uint64_t abs_out, uint32_t n_out, uint64_t den) {
    if (n_out == 0) return 1;
    const uint64_t k0 = abs_out / den, k1 = (abs_out + n_out - 1) / den;
    return static_cast<size_t>((k1 - k0) / 32 + 1);
}

void periodic_fill_wrap_bits(const std::vector<uint32_t>& wraps, uint64_t abs_out, uint64_t den,
                             uint32_t* words, size_t n_words) {
    std::memset(words, 0, n_words * sizeof(uint32_t));
    const uint64_t k0 = abs_out / den;
    for (uint32_t n : wraps) {
        const uint64_t k = (abs_out + n) / den - k0;
        words[k >> 5] |= 1u << (k & 31);
    }
}

HostClassTable build_class_table(const std::vector<float>& coeffs, const PeriodicGeometry& g,
                                 double drift) {
    const uint32_t taps = g.taps;
    HostClassTable out;
    out.coef.assign(static_cast<size_t>(g.n_tiles) * g.row_len * kClassTile, 0.0f);
    out.wrap_coef.assign(static_cast<size_t>(g.n_tiles) * g.row_len, 0.0f);
    out.meta.resize(g.n_tiles);
    std::vector<float> mixed(taps);
    const float* row1023 = coeffs.data() + (kPhases - 1) * taps;
    for (uint32_t t = 0; t < g.n_tiles; ++t) {
        TileMeta& tm = out.meta[t];
        std::memset(&tm, 0, sizeof tm);
        const uint32_t j0 = t * kClassTile;
        tm.base = class_offset(g, j0);
        tm.wrap_col = -1;
        tm.extra_col = -2;
        float* base = out.coef.data() + static_cast<size_t>(t) * g.row_len * kClassTile;
        for (uint32_t i = 0; i < kClassTile && j0 + i < g.b; ++i) {
            const uint32_t j = j0 + i;
            // exact fractional position of class j, plus the stream's current f64 drift
            const uint64_t rem = (static_cast<uint64_t>(j) * g.a) % g.b;
            double fract = static_cast<double>(rem) / static_cast<double>(g.b) + drift;
            if (j % g.den == 0) fract = drift > 0.0 ? drift : 0.0;  // below-integer: wrap variant
            if (fract < 0.0) fract = 0.0;
            // resampler_fir.rs:562-565
            double phase_f = fract * static_cast<double>(kPhases);
            if (phase_f > static_cast<double>(kPhases - 1)) phase_f = static_cast<double>(kPhases - 1);
            const size_t phase1 = static_cast<size_t>(phase_f);
            const size_t phase2 = phase1 + 1 < kPhases - 1 ? phase1 + 1 : kPhases - 1;
            const float frac = static_cast<float>(phase_f - static_cast<double>(phase1));
            const float* c1 = coeffs.data() + phase1 * taps;
            const float* c2 = coeffs.data() + phase2 * taps;
            const float omf = 1.0f - frac;
            for (uint32_t k = 0; k < taps; ++k) mixed[k] = c1[k] * omf + c2[k] * frac;  // avx.rs:41-45
            const uint32_t shift = class_offset(g, j) - tm.base;
            for (uint32_t k = 0; k < taps; ++k) base[(k + shift) * kClassTile + i] = mixed[k];

            if (g.inline_wraps && j % g.den == 0) {
                // wrap variant of class j: row 1023 on the window one frame earlier (:544, :562-564)
                tm.wrap_col = static_cast<int32_t>(i);
                tm.wrap_jd = j / g.den;
                float* wc = out.wrap_coef.data() + static_cast<size_t>(t) * g.row_len;
                const int64_t w = static_cast<int64_t>(class_offset(g, j)) - 1;
                if (w >= static_cast<int64_t>(tm.base)) {
                    const uint32_t ws = static_cast<uint32_t>(w - tm.base);
                    for (uint32_t k = 0; k < taps; ++k) wc[k + ws] = row1023[k];
                } else {  // one sample in front of the tile window
                    for (uint32_t k = 1; k < taps; ++k) wc[k - 1] = row1023[k];
                    tm.extra_col = static_cast<int32_t>(tm.base) - 1;
                    tm.extra_coef = row1023[0];
                }
            }
        }
    }
    return out;
}

int periodic_bind(PeriodicState& st, int device, const std::vector<float>& table,
                  const FirMirror& planned, uint32_t channels, hipStream_t stream) {
    (void)stream;
    if (!st.geo_valid) {
        st.geo = periodic_geometry(planned.num(), planned.den(), static_cast<uint32_t>(planned.taps()),
                                   channels);
        st.geo_valid = true;
        st.table_valid = false;
    }
    if (!st.geo.ok) return fail(RSMP_ERR_INVALID_ARGUMENT, "periodic kernel: unsupported geometry");
    const double drift = std::round(planned.drift() / kDriftQuantum) * kDriftQuantum;
    if (st.table_valid && drift == st.table_drift) return RSMP_OK;
    ClassTableCache& cache = class_cache();
    std::lock_guard<std::mutex> lock(cache.mu);
    uint64_t bits;
    std::memcpy(&bits, &drift, sizeof bits);
    const ClassTableKey key{device, table.data(), st.geo.den, st.geo.a, st.geo.b, st.geo.row_len, bits};
    auto it = cache.tables.find(key);
    if (it == cache.tables.end()) {
        const HostClassTable host = build_class_table(table, st.geo, drift);
        const size_t coef_bytes = host.coef.size() * sizeof(float);
        const size_t wrap_bytes = host.wrap_coef.size() * sizeof(float);
        const size_t meta_bytes = host.meta.size() * sizeof(TileMeta);
        char* dptr = nullptr;
        RSMP_HIP_CHECK(hipMalloc(&dptr, coef_bytes + wrap_bytes + meta_bytes));
        RSMP_HIP_CHECK(hipMemcpy(dptr, host.coef.data(), coef_bytes, hipMemcpyHostToDevice));
        RSMP_HIP_CHECK(hipMemcpy(dptr + coef_bytes, host.wrap_coef.data(), wrap_bytes,
                                 hipMemcpyHostToDevice));
        RSMP_HIP_CHECK(hipMemcpy(dptr + coef_bytes + wrap_bytes, host.meta.data(), meta_bytes,
                                 hipMemcpyHostToDevice));
        ClassTable ct;
        ct.d_coef = reinterpret_cast<const float*>(dptr);
        ct.d_wrap_coef = reinterpret_cast<const float*>(dptr + coef_bytes);
        ct.d_meta = reinterpret_cast<const TileMeta*>(dptr + coef_bytes + wrap_bytes);
        it = cache.tables.emplace(key, ct).first;
    }
    st.table = it->second;
    st.table_valid = true;
    st.table_drift = drift;
    return RSMP_OK;
}

hipError_t launch_fir_periodic(const FirStreamDesc* d_descs, uint32_t n_streams,
                               const PeriodicGeometry& geo, uint32_t max_blocks,
                               unsigned long long* d_work_counter, unsigned long long* work_base,
                               hipStream_t stream) {
    if (n_streams == 0 || max_blocks == 0) return hipSuccess;
    const dim3 block(geo.waves * 64);
    GeoArgs args = to_args(geo);
    args.blocks_per_stream = max_blocks;
    args.total_items = max_blocks * n_streams;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    static std::map<int, uint32_t> cu_count;
    static std::mutex cu_mu;
    uint32_t cus;
    {
        std::lock_guard<std::mutex> lock(cu_mu);
        uint32_t& c = cu_count[device];
        if (c == 0) {
            int v = 0;
            e = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device);
            if (e != hipSuccess) return e;
            c = static_cast<uint32_t>(v > 0 ? v : 256);
        }
        cus = c;
    }
    const uint32_t slots = cus * (geo.lds_bytes > kLdsTwoPerCu ? 1u : 2u);
    const dim3 grid(args.total_items < slots ? args.total_items : slots);
    args.work_counter = d_work_counter;
    args.work_base = *work_base;
    *work_base += args.total_items + grid.x;   // every workgroup makes exactly one failing claim
    static const char* trace_path = getenv("RSMP_FIR_TRACE");
    static unsigned long long* d_trace = nullptr;
    const size_t trace_words = 6ull * grid.x;
    if (trace_path) {
        if (d_trace) (void)hipFree(d_trace);
        if (hipMalloc(&d_trace, trace_words * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemset(d_trace, 0, trace_words * 8);
        args.trace = d_trace;
    }
    static const char* wtrace_path = getenv("RSMP_FIR_WTRACE");
    static unsigned long long* d_wtrace = nullptr;
    const size_t wtrace_words = static_cast<size_t>(grid.x) * kWtraceWaves * kWtraceSlots;
    if (wtrace_path) {
        if (d_wtrace) (void)hipFree(d_wtrace);
        if (hipMalloc(&d_wtrace, wtrace_words * 8) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemset(d_wtrace, 0, wtrace_words * 8);
        args.wtrace = d_wtrace;
    }
    // Dynamic LDS above 64 KiB must be opted into, once per kernel and device.
    static std::mutex mu;
    static std::map<std::pair<int, int>, bool> granted;
    // variants: 0 = two channels, one lane per period; 1 = CG 2, any even channel count; 2 = CG 1.
    // (4-tap chunks with 16-wave workgroups at 8 waves per SIMD measured 13 % slower than 8-tap
    // chunks with 12 waves: the 64-VGPR cap spills.)
    const int variant = geo.cg == 2 ? (geo.lp == 1 ? 0 : 1) : 2;
    const void* fns[3] = {reinterpret_cast<const void*>(fir_periodic_kernel<2, true, 8>),
                          reinterpret_cast<const void*>(fir_periodic_kernel<2, false, 8>),
                          reinterpret_cast<const void*>(fir_periodic_kernel<1, false, 8>)};
    {
        std::lock_guard<std::mutex> lock(mu);
        bool& have = granted[{device, variant}];
        if (!have) {
            e = hipFuncSetAttribute(fns[variant], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsMax);
            if (e != hipSuccess) return e;
            have = true;
        }
    }
    static const bool verbose = getenv("RSMP_FIR_VERBOSE") != nullptr;
    if (verbose) {
        int blocks = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fns[variant], geo.waves * 64,
                                                           geo.lds_bytes);
        fprintf(stderr,
                "[rsmp] periodic launch: a=%u b=%u row_len=%u tiles=%u cg=%u lp=%u pw=%u stride=%u "
                "waves=%u lds=%u items=%u grid=%u occupancy=%d blocks/CU\n",
                geo.a, geo.b, geo.row_len, geo.n_tiles, geo.cg, geo.lp, geo.pw, geo.row_stride,
                geo.waves, geo.lds_bytes, args.total_items, grid.x, blocks);
    }
    if (variant == 0)
        hipLaunchKernelGGL((fir_periodic_kernel<2, true, 8>), grid, block, geo.lds_bytes, stream, d_descs, args);
    else if (variant == 1)
        hipLaunchKernelGGL((fir_periodic_kernel<2, false, 8>), grid, block, geo.lds_bytes, stream, d_descs, args);
    else
        hipLaunchKernelGGL((fir_periodic_kernel<1, false, 8>), grid, block, geo.lds_bytes, stream, d_descs, args);
    if (trace_path) {
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h(trace_words);
        (void)hipMemcpy(h.data(), d_trace, trace_words * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(trace_path, "w")) {
            for (size_t i = 0; i < trace_words / 6; ++i)
                fprintf(f, "%zu %llu %llu %llu %llu %llu %llu\n", i, h[6 * i], h[6 * i + 1], h[6 * i + 2],
                        h[6 * i + 3], h[6 * i + 4], h[6 * i + 5]);
            fclose(f);
        }
    }
    if (wtrace_path) {   // one line per wave: block wave event...
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h(wtrace_words);
        (void)hipMemcpy(h.data(), d_wtrace, wtrace_words * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(wtrace_path, "w")) {
            for (size_t w = 0; w < wtrace_words / kWtraceSlots; ++w) {
                fprintf(f, "%zu %zu", w / kWtraceWaves, w % kWtraceWaves);
                for (uint32_t i = 0; i < kWtraceSlots && h[w * kWtraceSlots + i]; ++i)
                    fprintf(f, " %llu:%llu", h[w * kWtraceSlots + i] >> 8, h[w * kWtraceSlots + i] & 255);
                fprintf(f, "\n");
            }
            fclose(f);
        }
    }
    return hipGetLastError();
}

hipError_t launch_fir_wrap_fixup(const FirStreamDesc* d_descs, uint32_t n_streams,
                                 uint32_t max_wraps, hipStream_t stream) {
    if (n_streams == 0 || max_wraps == 0) return hipSuccess;
    hipLaunchKernelGGL(fir_wrap_fixup_kernel, dim3((max_wraps + 31) / 32, n_streams), dim3(256), 0,
                       stream, d_descs);
    return hipGetLastError();
}

}  // namespace rsmp
