// fir_generic_bulk.hip -- the reference-form polyphase FIR kernel for LONG launches of streams whose ratio has no short period
// (ResamplerFir::new_from_hz with arbitrary rates, src/resampler_fir.rs:295-301: 44100 -> 47999 Hz, a ratio drifted out of the
// periodic analysis' bound ...), gfx950.
//
// Replaces the same reference code as fir_generic.hip -- the per-output-frame loop body (src/resampler_fir.rs:542-590) and the
// convolution leaf (src/fir/avx.rs:5-61) in its own two-row form: two dot products against the adjacent phase rows, lerped per
// lane by `frac`, then summed -- with the same exact f64 position per frame (p = p0 + k * inc from the host mirror's run
// descriptors).  What differs is where the operands come from.  fir_generic_kernel reads, per output frame and channel, both
// 512-byte phase rows from L2 and its 128-frame window from global memory: ~1 KB of L2 reads per frame for the rows alone,
// 0.7 % of the HBM roofline.  Consecutive outputs of such a ratio never share a row -- the phase moves
// by ~-83 rows per output at 44100 -> 47999 -- but the outputs of a long tile revisit every row: so a workgroup takes a TILE of
// up to 4096 consecutive output frames of one stream and
//   A. computes every output's (window start, phase row, frac) once, exactly as the reference derives them (:544-565);
//   B. SORTS the tile's outputs by phase row (a counting sort over the 1024 rows in LDS), so that the outputs that read rows
//      p and p + 1 are neighbours -- a wave walks a contiguous stretch of the sorted list and finds the row of the output before
//      in L1 (a row is fetched from L2 about once per tile: 128 B per frame instead of 1 KB);
//   C. stages the tile's input window -- [hist | in], every channel, PCM converted (fir_in_value) -- in LDS once;
//   D. eight lanes per output frame as in fir_generic_kernel (lane g owns the float4 chunks g, g + 8, ... of both rows, held in
//      registers for ALL the frame's channels), the samples of a channel PAIR as one 8-byte LDS read per tap and the two
//      channels' products as one packed FMA per row and tap, per-lane lerp, three DPP steps, one store per frame and pair.
// Not bit-identical to the CPU path (eight lanes with another tap assignment than the AVX registers'; tests hold it to the same
// 1e-6 RMS as every other kernel), identical counts by construction (the counts are the host mirror's).
#include <algorithm>

#include "fir_kernels.h"

namespace rsmp {

namespace {

constexpr int kBulkBlock = 1024;                     // sixteen waves
constexpr uint32_t kBulkTileMax = 4096;              // output frames per workgroup
constexpr uint32_t kBulkWindowBytes = 64u * 1024u;   // LDS for the input window
constexpr uint32_t kRows = 1024;                     // phases (resampler_fir.rs:17)

struct BulkMeta {      // one output frame of the tile
    uint16_t rel;      // its window's first frame, relative to the tile's
    uint16_t phase;    // phase1 (:563)
    float frac;        // (:565)
};
static_assert(sizeof(BulkMeta) == 8, "BulkMeta layout");

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bulk_sum8(float v) {
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// One output frame's exact position in its run -> window start in [hist | in], phase row, frac (:544-565, as fir_generic_kernel).
__device__ __forceinline__ void bulk_position(const rsmp_fir_segment& sg, uint32_t n, int64_t* v0, uint32_t* phase1, float* frac) {
    const double p = fma(static_cast<double>(n - sg.out_start), sg.inc, sg.p0);
    const double fl = floor(p);                                   // :544
    const double fract = p - fl;                                  // :558 (p >= 0)
    double phase_f = fract * 1024.0;                              // :562
    phase_f = phase_f < 1023.0 ? phase_f : 1023.0;
    const uint32_t ph = static_cast<uint32_t>(phase_f);           // :563
    *phase1 = ph;
    *frac = static_cast<float>(phase_f - static_cast<double>(ph));  // :565
    *v0 = sg.in_base + static_cast<int64_t>(fl);
}
constexpr uint32_t kBulkSegs = 256;   // run descriptors of a tile kept in LDS (a 512-frame call is ~10 runs: a tile of 4096 outputs ~80)

// Descriptor pointers come out of a struct in memory: without a named address space the compiler emits FLAT loads for them
// (which also count on the LDS counter and wait with it).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4f* GRow;
typedef __attribute__((address_space(1))) float* GOut;
typedef const __attribute__((address_space(1))) float* GIn;
typedef __attribute__((address_space(1))) v2f* GOut2;

// acc += k.lo * x / acc += k.hi * x on both halves of a packed pair (x = the two channels of a frame, k = two neighbouring taps of
// a row): the tap is broadcast by the instruction's op_sel bits -- written as v2f{k, k} the compiler builds the pair with two moves
// per tap, 64 moves beside 64 multiply-adds per pass.
__device__ __forceinline__ void pk_fma_lo(v2f& acc, v2f k, v2f x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(k), "v"(x));
}
__device__ __forceinline__ void pk_fma_hi(v2f& acc, v2f k, v2f x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(k), "v"(x));
}

// What a pass of eight output frames needs before its arithmetic: which output each eight-lane group has, its meta word and the
// lane's sixteen taps of both phase rows.  Fetched a pass AHEAD (two sets of registers, the loop below alternates them): the
// chain sorted[] -> meta[] -> row address -> L2 is ~2 us of latency that a pass's own arithmetic (~1 us) otherwise waits for
// (without it: 13.9 ms per 64 x 2^20 frames; with it 4.4).  Measured and dropped: a group OWNING a phase row for a pass -- the rows
// fetched once for all the row's outputs, four on average -- is slower (5.6 ms): the longest of eight rows sets the trips, half the
// lanes idle, and the kernel is bound by the instructions it issues (profiles/r06/generic_bulk.txt), not by the rows' path.
struct BulkPass {
    v4f k1[4], k2[4];
    BulkMeta m;
    uint32_t i;
    bool live;
};

template <int CH, bool FULL>   // FULL: 128 taps (every lane's four chunks exist: no tests, every LDS offset an immediate)
__device__ __forceinline__ void bulk_outputs(const FirStreamDesc& d, const BulkMeta* meta, const uint16_t* sorted, const float* win, uint32_t n0,
                                             uint32_t nt, uint32_t lane, uint32_t wave) {
    const uint32_t ch = CH ? static_cast<uint32_t>(CH) : d.channels, taps = d.taps;
    const uint32_t g = lane & 7u, slot = lane >> 3;
    const uint32_t per_wave = ((nt + 15u) / 16u + 7u) & ~7u;   // (a multiple of eight: a wave's passes are whole)
    const uint32_t begin = wave * per_wave;
    const uint32_t end = begin + per_wave < nt ? begin + per_wave : nt;
    if (begin >= end) return;
    const uint32_t chunks = FULL ? 32u : taps / 4;   // float4 chunks per row (4 .. 32); lane g owns g, g + 8, g + 16, g + 24
    const GRow coeffs = (GRow)d.coeffs;
    const GOut out = (GOut)d.out;
    auto fetch = [&](uint32_t base, BulkPass& p) {
        p.live = base + slot < end;
        p.i = sorted[p.live ? base + slot : begin];
        p.m = meta[p.i];
        const uint32_t phase2 = p.m.phase + 1u < 1023u ? p.m.phase + 1u : 1023u;   // :564
        const GRow row1 = coeffs + static_cast<size_t>(p.m.phase) * chunks, row2 = coeffs + static_cast<size_t>(phase2) * chunks;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t q = g + 8u * j;
            p.k1[j] = FULL || q < chunks ? row1[q] : v4f{0.f, 0.f, 0.f, 0.f};
            p.k2[j] = FULL || q < chunks ? row2[q] : v4f{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto compute = [&](const BulkPass& p) {
        const float frac = p.m.frac, one_minus_frac = 1.0f - frac;                // avx.rs:42
        const float* wbase = win + static_cast<uint32_t>(p.m.rel) * ch;
        const size_t n = static_cast<size_t>(n0) + p.i;
        if constexpr (CH == 2 && FULL) {   // two channels, 128 taps: frame f of the window is one 8-byte value at an immediate offset
            const v2f* xp = reinterpret_cast<const v2f*>(win) + static_cast<uint32_t>(p.m.rel) + 4u * g;
            v2f a1 = v2f{0.f, 0.f}, a2 = v2f{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v2f x0 = xp[32 * j], x1 = xp[32 * j + 1], x2 = xp[32 * j + 2], x3 = xp[32 * j + 3];
                const v2f k1a = v2f{p.k1[j].x, p.k1[j].y}, k1b = v2f{p.k1[j].z, p.k1[j].w};
                const v2f k2a = v2f{p.k2[j].x, p.k2[j].y}, k2b = v2f{p.k2[j].z, p.k2[j].w};
                pk_fma_lo(a1, k1a, x0); pk_fma_lo(a2, k2a, x0);
                pk_fma_hi(a1, k1a, x1); pk_fma_hi(a2, k2a, x1);
                pk_fma_lo(a1, k1b, x2); pk_fma_lo(a2, k2b, x2);
                pk_fma_hi(a1, k1b, x3); pk_fma_hi(a2, k2b, x3);
            }
            const v2f part = a1 * one_minus_frac + a2 * frac;
            const float y0 = bulk_sum8(part.x), y1 = bulk_sum8(part.y);
            if (p.live && g == 0) *(GOut2)(out + n * 2) = v2f{y0, y1};
            return;
        }
        uint32_t c = 0;
        for (; c + 2 <= ch; c += 2) {   // a channel pair: the two channels' samples of a frame are neighbours
            v2f a1 = v2f{0.f, 0.f}, a2 = v2f{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t q = g + 8u * j;
                if (FULL || q < chunks) {
                    const float* x = wbase + (4u * q) * ch + c;
                    v2f x0, x1, x2, x3;
                    if constexpr (CH == 2) {   // (frames of exactly the pair: 8-byte aligned)
                        const v2f* xp = reinterpret_cast<const v2f*>(x);
                        x0 = xp[0]; x1 = xp[1]; x2 = xp[2]; x3 = xp[3];
                    } else {
                        x0 = v2f{x[0], x[1]}; x1 = v2f{x[ch], x[ch + 1]}; x2 = v2f{x[2 * ch], x[2 * ch + 1]}; x3 = v2f{x[3 * ch], x[3 * ch + 1]};
                    }
                    a1 = __builtin_elementwise_fma(v2f{p.k1[j].x, p.k1[j].x}, x0, a1); a2 = __builtin_elementwise_fma(v2f{p.k2[j].x, p.k2[j].x}, x0, a2);
                    a1 = __builtin_elementwise_fma(v2f{p.k1[j].y, p.k1[j].y}, x1, a1); a2 = __builtin_elementwise_fma(v2f{p.k2[j].y, p.k2[j].y}, x1, a2);
                    a1 = __builtin_elementwise_fma(v2f{p.k1[j].z, p.k1[j].z}, x2, a1); a2 = __builtin_elementwise_fma(v2f{p.k2[j].z, p.k2[j].z}, x2, a2);
                    a1 = __builtin_elementwise_fma(v2f{p.k1[j].w, p.k1[j].w}, x3, a1); a2 = __builtin_elementwise_fma(v2f{p.k2[j].w, p.k2[j].w}, x3, a2);
                }
            }
            // per-lane lerp, then the horizontal sum (avx.rs:41-58)
            const float y0 = bulk_sum8(a1.x * one_minus_frac + a2.x * frac);
            const float y1 = bulk_sum8(a1.y * one_minus_frac + a2.y * frac);
            if (p.live && g == 0) {
                if constexpr (CH == 2) *(GOut2)(out + n * 2) = v2f{y0, y1};
                else { out[n * ch + c] = y0; out[n * ch + c + 1] = y1; }
            }
        }
        if (CH != 2 && c < ch) {   // one channel, or an odd count's last
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t q = g + 8u * j;
                if (FULL || q < chunks) {
                    const float* x = wbase + (4u * q) * ch + c;
                    a1 = fmaf(p.k1[j].x, x[0], a1); a2 = fmaf(p.k2[j].x, x[0], a2);
                    a1 = fmaf(p.k1[j].y, x[ch], a1); a2 = fmaf(p.k2[j].y, x[ch], a2);
                    a1 = fmaf(p.k1[j].z, x[2 * ch], a1); a2 = fmaf(p.k2[j].z, x[2 * ch], a2);
                    a1 = fmaf(p.k1[j].w, x[3 * ch], a1); a2 = fmaf(p.k2[j].w, x[3 * ch], a2);
                }
            }
            const float y = bulk_sum8(a1 * one_minus_frac + a2 * frac);
            if (p.live && g == 0) out[n * ch + c] = y;
        }
    };
    if constexpr (CH == 0 || !FULL) {   // (any channel count / fewer taps: one set of registers -- with two the general loop spills)
        BulkPass p;
        for (uint32_t base = begin; base < end; base += 8u) {
            fetch(base, p);
            compute(p);
        }
        return;
    }
    BulkPass pa, pb;
    fetch(begin, pa);
    for (uint32_t base = begin; base < end; base += 16u) {
        if (base + 8u < end) fetch(base + 8u, pb);
        compute(pa);
        if (base + 8u >= end) break;
        if (base + 16u < end) fetch(base + 16u, pa);
        compute(pb);
    }
}

// CH: 1 / 2 = every stream of the launch has that many channels; 0 = any counts (a kernel each: side by side in one kernel the three
// bodies shared a register allocation and spilled)
template <int CH, bool FULL>
__global__ __launch_bounds__(kBulkBlock) void fir_generic_bulk_kernel(const FirStreamDesc* __restrict__ descs, uint32_t tile_frames,
                                                                      uint32_t window_cap_frames) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const FirStreamDesc d = descs[blockIdx.y];
    const uint32_t n0 = blockIdx.x * tile_frames;
    if (n0 >= d.n_out) return;
    const uint32_t nt = d.n_out - n0 < tile_frames ? d.n_out - n0 : tile_frames;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t ch = CH ? static_cast<uint32_t>(CH) : d.channels, taps = d.taps;

    uint32_t* counts = reinterpret_cast<uint32_t*>(lds);                    // [1024]: outputs per phase row, then the scatter's cursors
    uint32_t* offsets = counts + kRows;                                     // [1024]: first slot of a row's outputs in `sorted`
    uint32_t* wsum = offsets + kRows;                                       // [16]: the scan's wave totals
    rsmp_fir_segment* segs = reinterpret_cast<rsmp_fir_segment*>(wsum + 32);   // [kBulkSegs]: the tile's run descriptors
    BulkMeta* meta = reinterpret_cast<BulkMeta*>(segs + kBulkSegs);         // [tile_frames]
    uint16_t* sorted = reinterpret_cast<uint16_t*>(meta + tile_frames);     // [tile_frames]: the tile's outputs by phase row
    float* win = reinterpret_cast<float*>(lds + ((reinterpret_cast<char*>(sorted + tile_frames) - lds + 15) & ~15));   // [window][ch]

    // ---- the tile's run descriptors into LDS (the runs are sorted by out_start; tile_seg[n / kFirTile] is the run of that
    // 32-frame group's first frame): two rounds of global latency for the whole tile instead of a walk per output
    const uint32_t s_lo = d.tile_seg[n0 / kFirTile];
    uint32_t s_hi = d.tile_seg[(n0 + nt - 1) / kFirTile] + kFirTile;   // (the last group's frames lie at most a run per frame further)
    if (s_hi >= d.n_segs) s_hi = d.n_segs - 1;
    const uint32_t n_seg = s_hi - s_lo + 1 < kBulkSegs ? s_hi - s_lo + 1 : kBulkSegs;
    for (uint32_t j = tid; j < n_seg; j += kBulkBlock) segs[j] = d.segs[s_lo + j];
    counts[tid] = 0;
    __syncthreads();
    // the run of output n: the last one that starts at or before it (binary search in LDS; beyond kBulkSegs runs -- never seen --
    // the walk goes on in global memory)
    auto locate = [&](uint32_t n, int64_t* v0, uint32_t* phase1, float* frac) {
        uint32_t lo = 0, hi = n_seg - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (segs[mid].out_start <= n) lo = mid;
            else hi = mid - 1;
        }
        rsmp_fir_segment sg = segs[lo];
        uint32_t s = s_lo + lo;
        while (n >= sg.out_start + sg.count) sg = d.segs[++s];
        bulk_position(sg, n, v0, phase1, frac);
    };
    // ---- the tile's window: from the first output's window start to the last one's end (window starts never go back)
    int64_t w0, v_last;
    {
        uint32_t ph;
        float fr;
        locate(n0, &w0, &ph, &fr);
        locate(n0 + nt - 1, &v_last, &ph, &fr);
    }
    uint32_t wlen = static_cast<uint32_t>(v_last - w0) + taps;
    if (wlen > window_cap_frames) wlen = window_cap_frames;   // (cannot happen: the host sizes the tile for the launch's largest ratio)
    // ---- C: the window, all channels ([hist | in]; frames beyond the input read as zero -- no output's window reaches them)
    {
        const int64_t hist_frames = d.hist_frames, total = hist_frames + static_cast<int64_t>(d.in_frames);
        const uint32_t n_values = wlen * ch;
        for (uint32_t j = tid; j < n_values; j += kBulkBlock) {
            const uint32_t f = j / ch, c = j - f * ch;
            const int64_t v = w0 + f;
            float x = 0.f;
            if (v >= 0 && v < total) {
                if (v < hist_frames) x = ((GIn)d.hist)[static_cast<size_t>(v) * ch + c];
                else if (d.in_bits == 0) x = ((GIn)d.in)[static_cast<size_t>(v - hist_frames) * ch + c];
                else x = fir_pcm_value(d.in, d.in_bits, static_cast<size_t>(v - hist_frames) * ch + c);
            }
            win[j] = x;
        }
    }
    // ---- A: every output's window start / phase row / frac; the histogram of the rows (behind the window's loads: their latency
    // passes under this arithmetic)
    for (uint32_t i = tid; i < nt; i += kBulkBlock) {
        int64_t v;
        uint32_t ph;
        float fr;
        locate(n0 + i, &v, &ph, &fr);
        BulkMeta m;
        m.rel = static_cast<uint16_t>(v - w0);
        m.phase = static_cast<uint16_t>(ph);
        m.frac = fr;
        meta[i] = m;
        atomicAdd(counts + ph, 1u);
    }
    __syncthreads();
    // ---- B: exclusive scan of the 1024 counts (a value per thread: six shuffle steps, the waves' totals through LDS), scatter
    {
        const uint32_t c = counts[tid];
        uint32_t incl = c;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t up = __shfl_up(incl, sft, 64);
            if (lane >= static_cast<uint32_t>(sft)) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        offsets[tid] = before + incl - c;
        counts[tid] = 0;
    }
    __syncthreads();
    for (uint32_t i = tid; i < nt; i += kBulkBlock) {
        const uint32_t ph = meta[i].phase;
        sorted[offsets[ph] + atomicAdd(counts + ph, 1u)] = static_cast<uint16_t>(i);
    }
    __syncthreads();

    // ---- D: the outputs in phase order, a contiguous stretch of the list per wave, eight lanes per output frame
    bulk_outputs<CH, FULL>(d, meta, sorted, win, n0, nt, lane, wave);
}

}  // namespace

// Tiles of `tile` output frames whose windows fit the LDS for every stream of the launch: the largest multiple of 64 up to
// kBulkTileMax with ceil(tile * max_ratio) + taps + 2 frames of max_channels channels inside kBulkWindowBytes; 0 = none (many
// channels at a high ratio: the caller keeps fir_generic_kernel).
uint32_t fir_generic_bulk_tile(uint32_t max_channels, uint32_t max_taps, double max_ratio) {
    if (max_channels == 0 || max_ratio <= 0.0) return 0;
    const uint32_t cap = kBulkWindowBytes / (4u * max_channels);
    if (cap < max_taps + 8u) return 0;
    double t = (static_cast<double>(cap) - max_taps - 4.0) / max_ratio;
    if (t > kBulkTileMax) t = kBulkTileMax;
    const uint32_t tile = static_cast<uint32_t>(t) / 64u * 64u;
    return tile >= 512u ? tile : 0u;
}

hipError_t launch_fir_generic_bulk(const FirStreamDesc* d_descs, uint32_t n_streams, uint32_t max_out, uint32_t max_channels,
                                   uint32_t max_taps, double max_ratio, hipStream_t stream, uint32_t uniform_channels, uint32_t uniform_taps) {
    if (n_streams == 0 || max_out == 0) return hipSuccess;
    const uint32_t tile = fir_generic_bulk_tile(max_channels, max_taps, max_ratio);
    if (tile == 0) return hipErrorNotSupported;
    const uint32_t cap = kBulkWindowBytes / (4u * max_channels);
    const size_t lds = (2 * kRows + 32) * sizeof(uint32_t) + kBulkSegs * sizeof(rsmp_fir_segment) + static_cast<size_t>(tile) * (sizeof(BulkMeta) + sizeof(uint16_t)) + 16 +
                       static_cast<size_t>(cap) * max_channels * sizeof(float);
    const bool full = uniform_taps == 128;
    const int ci = uniform_channels == 2 ? 2 : uniform_channels == 1 ? 1 : 0;
    static const void* const fns[3][2] = {
        {reinterpret_cast<const void*>(fir_generic_bulk_kernel<0, false>), reinterpret_cast<const void*>(fir_generic_bulk_kernel<0, true>)},
        {reinterpret_cast<const void*>(fir_generic_bulk_kernel<1, false>), reinterpret_cast<const void*>(fir_generic_bulk_kernel<1, true>)},
        {reinterpret_cast<const void*>(fir_generic_bulk_kernel<2, false>), reinterpret_cast<const void*>(fir_generic_bulk_kernel<2, true>)}};
    const void* fn = fns[ci][full ? 1 : 0];
    static bool granted[3][2] = {};   // (the attribute is per function and process: set once)
    bool& have = granted[ci][full ? 1 : 0];
    if (!have) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        have = true;
    }
    const dim3 grid((max_out + tile - 1) / tile, n_streams);
    uint32_t tile_arg = tile, cap_arg = cap;
    void* kargs[3] = {&d_descs, &tile_arg, &cap_arg};
    if (hipError_t e = hipLaunchKernel(fn, grid, dim3(kBulkBlock), kargs, lds, stream); e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace rsmp
