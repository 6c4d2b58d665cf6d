// fir_periodic.h -- host side of the periodic (rational-ratio) FIR throughput kernel.
//
// For in_hz/out_hz = num/den the exact position of output m is m*num/den: the phase pattern
// repeats every `den` outputs and the reference's f64 position stays within ~1e-9 of it
// (measured per launch by FirMirror).  The kernel exploits that:
//   * outputs are grouped in *classes* j = m mod b and classes in tiles of 8; b = r*den outputs
//     consume a = r*num input frames (r = the smallest multiplier with a >= the padded row
//     length, so a window never spans more than two period rows in LDS);
//   * per class the two phase rows are pre-mixed with the class's frac (the reference's lerp,
//     resampler_fir.rs:562-565 + fir/avx.rs:41-45, hoisted out of the per-sample loop -> `taps`
//     FMAs per value instead of 2*taps), shifted by the class's offset inside its tile and
//     zero padded, so all 8 classes of a tile read the SAME input samples: a register-tiled
//     8 x channels outer product per tap;
//   * the 64 lanes of a wave are 64 different periods, so the 8 coefficients of a tap are
//     wave-uniform and arrive through the scalar cache (s_load) while the samples come from LDS;
//   * the only outputs whose discrete choices depend on the sign of the f64 drift are those with
//     m*num/den integer: position just below the integer picks the previous frame and row 1023
//     (:562-564).  FirMirror lists them (`wraps`); the kernel carries a 9th accumulator for that
//     variant in the tiles that contain such a class and selects per lane from a bitmap
//     (den >= 8), or a fix-up kernel recomputes them (den < 8).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <vector>

#include "fir_kernels.h"
#include "fir_plan.h"

namespace rsmp {

constexpr uint32_t kClassTile = 8;
constexpr uint32_t kMfmaClassTile = 16;   // classes per tile of the matrix-core kernel (M of 16x16x4)

struct PeriodicGeometry {
    bool ok = false;
    uint32_t a = 0, b = 0;       // super period: a input frames -> b output frames
    uint32_t den = 0;            // true period of the phase pattern (b = r * den)
    uint32_t taps = 0;
    uint32_t row_len = 0;        // taps + max in-tile shift, rounded up to whole chunks (8; mfma 16)
    uint32_t n_tiles = 0;        // ceil(b / class tile); class tile = 8 (vector kernels) or 16 (mfma)
    uint32_t cg = 0;             // channels per lane (1 or 2)
    uint32_t lp = 0;             // lanes per period = channels / cg
    uint32_t pw = 0;             // periods per workgroup (<= 64 / lp)
    uint32_t row_stride = 0;     // LDS dwords between period rows (odd frame count: conflict-free)
    uint32_t waves = 0;          // waves per workgroup
    uint32_t producers = 0;      // > 0: double-buffered kernel, this many waves only stage
    uint32_t images = 0;         // double-buffered kernels: LDS images in the ring (2 or 4)
    uint32_t mfma = 0;           // > 0: matrix-core kernel (16-class tiles); period groups of 16 per work unit;
                                 // 3: split kernel (fir_split.hip)
    uint32_t planes = 0;         // split kernel: 16-bit planes per f32 operand (3: bf16, exact; 2: fp16)
    uint32_t groups = 0;         // split kernel: tile groups of ten class tiles (1 or 2)
    uint32_t rounds = 0;         // split kernel: rounds of lane tasks per stager and item (1: periods <= 160 frames; 2: <= 320)
    uint32_t n_units = 0;        // work units per item: n_tiles (vector kernels) or tiles x unit splits (mfma)
    uint32_t lds_bytes = 0;
    bool inline_wraps = false;   // den >= 8: wrap variant computed inside the kernel
    bool operator==(const PeriodicGeometry& o) const {
        return a == o.a && b == o.b && den == o.den && taps == o.taps && row_len == o.row_len &&
               cg == o.cg && lp == o.lp && pw == o.pw && row_stride == o.row_stride &&
               waves == o.waves && producers == o.producers && mfma == o.mfma && images == o.images &&
               planes == o.planes;
    }
};

// allow_matrix = false: vector kernels only (RSMP_FIR_KERNEL_PERIODIC_VECTOR); allow_split = false:
// never the split-bf16 kernel (RSMP_FIR_KERNEL_PERIODIC_F32)
PeriodicGeometry periodic_geometry(uint64_t num, uint64_t den, uint32_t taps, uint32_t channels,
                                   bool allow_matrix = true, bool allow_split = true);

// Per class tile: where its window starts and what its wrap variant (if any) needs.
struct TileMeta {
    uint32_t base;         // first input frame of the tile's window, relative to the period start
    int32_t wrap_col;      // column (0..7) whose class has an integer exact position, or -1
    uint32_t wrap_jd;      // (class index of that column) / den
    int32_t extra_col;     // frame (relative to the period start, may be -1) of the one sample
                           // the wrap window has in front of the tile window; -2 = none
    float extra_coef;      // its coefficient (row 1023, tap 0)
    uint32_t pad[3];
};
static_assert(sizeof(TileMeta) == 32, "TileMeta is read with one s_load_dwordx8");

// Device image of one class table: [tile][row_len][8] coefficients, [tile][row_len] wrap-variant
// coefficients, [tile] TileMeta.
// `hold` keeps the device allocation alive: the cache is bounded (a stream's drift moves on for as long as it runs, and
// every drift step is a new table), and a table that has left it is freed once nobody holds it any more AND the device
// has been waited for (kernels already enqueued may still read it): class_table_for.
struct ClassTable {
    const float* d_coef = nullptr;
    const float* d_wrap_coef = nullptr;
    const TileMeta* d_meta = nullptr;
    std::shared_ptr<void> hold;
};

// Per-handle periodic state: the class table currently bound to the stream.
struct PeriodicState {
    PeriodicGeometry geo;
    bool geo_valid = false;
    int geo_mode = -1;           // kernel mode the geometry was derived for
    ClassTable table;
    bool table_valid = false;
    double table_drift = 0.0;
};

bool periodic_supported(const FirMirror& m, size_t channels, size_t taps, int kernel_mode);
bool periodic_worthwhile(const FirMirror& planned, size_t produced_frames, int kernel_mode);

// Makes sure `st` holds the geometry and the device class table matching the stream's rate pair
// and its current f64 drift (host build + one upload, cached per device and shared by every
// stream with the same polyphase table, rate pair and drift).
// `drift`: what the launch's coefficient rows are mixed for -- the middle between the stream's drift before the launch and
// behind it (the drift moves by ~1e-14 of a frame per output: a launch is half as far from its table that way).
int periodic_bind(PeriodicState& st, int device, const std::vector<float>& table, int kernel_mode,
                  const FirMirror& planned, double drift, uint32_t channels, hipStream_t stream);

// Device class table for a geometry and drift (built on the host once, cached per device).  `prebuilt`: the host image
// for exactly these arguments, from build_class_table run ahead of time (on another thread: it touches no device): only
// the allocation and the upload, 0.06 ms, are left for the caller.
struct HostClassTable;
int class_table_for(int device, const std::vector<float>& table, const PeriodicGeometry& g, double drift,
                    ClassTable* out, const HostClassTable* prebuilt = nullptr);

// Bitmap of wrapped outputs for one launch: bit K <-> the output with absolute index
// (abs_out / den + K) * den.  Returns the number of 32-bit words.
size_t periodic_wrap_words(uint64_t abs_out, uint32_t n_out, uint64_t den);
void periodic_fill_wrap_bits(const std::vector<uint32_t>& wraps, uint64_t abs_out, uint64_t den,
                             uint32_t* words, size_t n_words);

// One launch per geometry: d_descs[0..n_streams) all use `geo`; grid = (max_blocks, n_streams).
// `d_work_counter` is a zero-initialised 64-bit device word owned by the caller; the kernel leaves it
// at zero again (launches sharing it must be ordered, which launch_jobs enforces per handle).
// `nf`: where non-finite sums are marked (fir_nonfinite.h); the caller follows up with launch_fir_repair.
// items_key: a hash of everything the split kernel's item table depends on (the streams' counters in launch order; 0 =
// none): a launch with the key of the table already in the stream's workspace does not rebuild it.
// pcm_bits != 0: the streams' `in` is PCM of that width (FirStreamDesc::in_bits): the split kernel's two-channel builds
// read it; hipErrorNotSupported for any other kernel.
hipError_t launch_fir_periodic(const FirStreamDesc* d_descs, uint32_t n_streams,
                               const PeriodicGeometry& geo, uint32_t max_blocks,
                               unsigned long long* d_work_counter, const NfArgs& nf, hipStream_t stream,
                               bool fuse_tail = false, uint64_t items_key = 0, uint32_t pcm_bits = 0);
// Recomputes the outputs listed in each stream's `wraps` with row 1023 / previous frame
// (only for geometries without inline wraps).
hipError_t launch_fir_wrap_fixup(const FirStreamDesc* d_descs, uint32_t n_streams,
                                 uint32_t max_wraps, hipStream_t stream);
// Number of period blocks (grid.x) a stream's launch needs.
uint32_t periodic_blocks(const PeriodicGeometry& geo, uint64_t abs_out, uint32_t n_out);

// Split-bf16 matrix kernel (fir_split.hip): geometry (mfma == 3; row_stride = rows of an LDS image),
// class-table image and launch.
PeriodicGeometry split_geometry(uint64_t num, uint64_t den, uint32_t taps, uint32_t channels);
size_t split_table_floats(const PeriodicGeometry& g);
void split_store_class(std::vector<float>& coef, const PeriodicGeometry& g, uint32_t tile, uint32_t m,
                       uint32_t shift, const std::vector<float>& mixed);
// fuse_tail: the kernel also copies every stream's still-buffered tail into hist_next (no tail-copy launch)
hipError_t launch_fir_split(const FirStreamDesc* d_descs, uint32_t n_streams, const PeriodicGeometry& geo,
                            uint32_t max_blocks, uint32_t cus, bool fuse_tail, const NfArgs& nf, hipStream_t stream,
                            uint64_t items_key = 0, uint32_t pcm_bits = 0);

// Several rate pairs in as few launches as their geometries allow (one item-table launch for all of them, then one
// launch of the kernel per window length among them): `jobs[j]` = the streams d_descs[0 .. n_streams) of geometry
// `geo`, as for launch_fir_split.  Jobs whose geometry the multi-job build does not cover (other than two channels)
// get a launch of their own.  The caller follows up with launch_fir_repair_multi over the same jobs' `nf`.
struct SplitJob {
    const FirStreamDesc* d_descs;
    uint32_t n_streams;
    const PeriodicGeometry* geo;
    uint32_t max_blocks;
    NfArgs nf;
};
// (items_prebuilt: the covered jobs' item tables, one behind the other, built before by a call with items_only = the stream to build
// them on -- which launches nothing else; fir_split_multi_item_words: how many words they take)
hipError_t launch_fir_split_multi(const SplitJob* jobs, size_t n_jobs, hipStream_t stream, uint32_t reserve_cus = 0,
                                  const uint32_t* items_prebuilt = nullptr, hipStream_t items_only = nullptr);
size_t fir_split_multi_item_words(const SplitJob* jobs, size_t n_jobs);

// Gives back the split kernel's item-table workspace of a stream that is about to be destroyed.
void split_release_stream(int device, hipStream_t stream);

// Host build of the class table (exposed for tests).
struct HostClassTable {
    std::vector<float> coef;       // [tile][row_len][8]; mfma: [tile][row_len / 16][64 lanes][4 steps]
    std::vector<float> wrap_coef;  // [tile][row_len]
    std::vector<TileMeta> meta;    // [tile]
};
HostClassTable build_class_table(const std::vector<float>& coeffs, const PeriodicGeometry& g,
                                 double drift);

}  // namespace rsmp
