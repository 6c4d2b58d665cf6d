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
//     (:562-564).  FirMirror lists them (`wraps`) and a fix-up kernel recomputes them.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "fir_kernels.h"
#include "fir_plan.h"

namespace rsmp {

constexpr uint32_t kClassTile = 8;

struct PeriodicGeometry {
    bool ok = false;
    uint32_t a = 0, b = 0;       // super period: a input frames -> b output frames
    uint32_t taps = 0;
    uint32_t row_len = 0;        // taps + max in-tile shift, rounded up to a multiple of 4
    uint32_t n_tiles = 0;        // ceil(b / 8)
    uint32_t cg = 0;             // channels per lane (1 or 2)
    uint32_t lp = 0;             // lanes per period = channels / cg
    uint32_t pw = 0;             // periods per workgroup (<= 64 / lp)
    uint32_t row_stride = 0;     // LDS dwords between period rows (padded: conflict-free)
    uint32_t waves = 0;          // waves per workgroup
    uint32_t lds_bytes = 0;
    bool operator==(const PeriodicGeometry& o) const {
        return a == o.a && b == o.b && taps == o.taps && row_len == o.row_len && cg == o.cg &&
               lp == o.lp && pw == o.pw && row_stride == o.row_stride && waves == o.waves;
    }
};

PeriodicGeometry periodic_geometry(uint64_t num, uint64_t den, uint32_t taps, uint32_t channels);

// Per-handle periodic state: the class table currently bound to the stream.
struct PeriodicState {
    PeriodicGeometry geo;
    bool geo_valid = false;
    const float* d_table = nullptr;  // device class table (owned by the global cache)
    double table_drift = 0.0;
};

bool periodic_supported(const FirMirror& m, size_t channels, size_t taps, int kernel_mode);
bool periodic_worthwhile(const FirMirror& planned, size_t produced_frames, int kernel_mode);

// Makes sure `st` holds the geometry and the device class table matching the stream's rate pair
// and its current f64 drift (host build + one upload, cached per device and shared by every
// stream with the same polyphase table, rate pair and drift).
int periodic_bind(PeriodicState& st, int device, const std::vector<float>& table,
                  const FirMirror& planned, uint32_t channels, hipStream_t stream);

// One launch per geometry: d_descs[0..n_streams) all use `geo`; grid = (max_blocks, n_streams).
hipError_t launch_fir_periodic(const FirStreamDesc* d_descs, uint32_t n_streams,
                               const PeriodicGeometry& geo, uint32_t max_blocks,
                               hipStream_t stream);
// Recomputes the outputs listed in each stream's `wraps` with row 1023 / previous frame.
hipError_t launch_fir_wrap_fixup(const FirStreamDesc* d_descs, uint32_t n_streams,
                                 uint32_t max_wraps, hipStream_t stream);
// Number of period blocks (grid.x) a stream's launch needs.
uint32_t periodic_blocks(const PeriodicGeometry& geo, uint64_t abs_out, uint32_t n_out);

// Host build of the class table (exposed for tests): layout [tile][row_len][8].
std::vector<float> build_class_table(const std::vector<float>& coeffs, const PeriodicGeometry& g,
                                     uint64_t den, double drift);

}  // namespace rsmp
