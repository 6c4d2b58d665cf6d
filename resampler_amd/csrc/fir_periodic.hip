// fir_periodic.hip -- throughput FIR kernel for rational rate pairs on gfx950 (see fir_periodic.h
// for the idea).  Replaces the same reference code as fir_generic.hip
// (src/resampler_fir.rs:542-590 + src/fir/avx.rs:5-61) for launches long enough to fill waves
// with whole periods.
//
// Workgroup = `waves` wave64s sharing one staged input span of `pw` periods:
//   stage   : [hist|in] frames (q0*a ... (q0+pw)*a + row_len) -> LDS, one padded row per period
//             (row stride == lanes-per-period mod 32 read units -> the strided per-lane reads below
//             are bank-conflict free), zero filled outside the stream;
//   compute : wave w takes class tiles w, w+waves, ...; lane = (period, channel group).  Per tap:
//             one ds_read of the lane's sample(s), 8 wave-uniform coefficients through the scalar
//             cache, 8 x CG v_fma with an SGPR operand.  No cross-lane traffic at all;
//   store   : each lane writes its 8 consecutive output frames (interleaved), masked to the launch.
// HBM traffic = input span once per workgroup (+ row_len halo) + output once; the class table
// (<= a few hundred KB) stays in L2 / scalar cache.
#include "fir_periodic.h"

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#include "common.h"
#include "filter_design.h"

namespace rsmp {

namespace {

struct GeoArgs {
    uint32_t a, b, row_len, n_tiles, lp, pw, row_stride, waves, channels, taps;
};

typedef const float __attribute__((address_space(4)))* const_f32_ptr;

template <int CG> struct Acc { float v[kClassTile][CG]; };

// `count` taps: sample(s) from LDS (stride `cstride` dwords per frame), 8 coefficients per tap
// from the class table through scalar loads.
template <int CG>
__device__ __forceinline__ void accumulate(Acc<CG>& acc, const float* __restrict__ x,
                                           uint32_t cstride, const_f32_ptr g, uint32_t count) {
    uint32_t m = 0;
    for (; m + 4 <= count; m += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float xs[CG];
            if constexpr (CG == 2) {
                const float2 t = *reinterpret_cast<const float2*>(x + (m + u) * cstride);
                xs[0] = t.x;
                xs[1] = t.y;
            } else {
                xs[0] = x[(m + u) * cstride];
            }
#pragma unroll
            for (int i = 0; i < (int)kClassTile; ++i) {
                const float c = g[(m + u) * kClassTile + i];
#pragma unroll
                for (int k = 0; k < CG; ++k) acc.v[i][k] = fmaf(c, xs[k], acc.v[i][k]);
            }
        }
    }
    for (; m < count; ++m) {
        float xs[CG];
        if constexpr (CG == 2) {
            const float2 t = *reinterpret_cast<const float2*>(x + m * cstride);
            xs[0] = t.x;
            xs[1] = t.y;
        } else {
            xs[0] = x[m * cstride];
        }
#pragma unroll
        for (int i = 0; i < (int)kClassTile; ++i) {
            const float c = g[m * kClassTile + i];
#pragma unroll
            for (int k = 0; k < CG; ++k) acc.v[i][k] = fmaf(c, xs[k], acc.v[i][k]);
        }
    }
}

template <int CG>
__global__ void fir_periodic_kernel(const FirStreamDesc* __restrict__ descs, GeoArgs geo) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FirStreamDesc& d = descs[blockIdx.y];
    const uint32_t n_out = d.n_out;
    if (n_out == 0) return;
    const uint64_t abs_out = d.abs_out;
    const uint64_t q_first = abs_out / geo.b;
    const uint64_t q0 = q_first + static_cast<uint64_t>(blockIdx.x) * geo.pw;
    const uint64_t m_end = abs_out + n_out;  // one past the last absolute output index
    if (q0 * geo.b >= m_end) return;

    const uint32_t C = geo.channels;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- stage ---------------------------------------------------------------------------------
    {
        const int64_t hist_frames = d.hist_frames;
        const int64_t total_frames = hist_frames + d.in_frames;
        const float* __restrict__ hist = d.hist;
        const float* __restrict__ in = d.in;
        // virtual index of the span's first frame: absolute frame q0*a minus frames retired so far
        const int64_t v_span = static_cast<int64_t>(q0 * geo.a) - static_cast<int64_t>(d.abs_consumed);
        const uint32_t row_values = geo.a * C;
        for (uint32_t p = wave; p <= geo.pw; p += geo.waves) {
            const uint32_t values = (p < geo.pw) ? row_values
                                                 : (geo.row_len < geo.a ? geo.row_len : geo.a) * C;
            const int64_t v_row = v_span + static_cast<int64_t>(p) * geo.a;
            float* __restrict__ dst = lds + static_cast<size_t>(p) * geo.row_stride;
            if ((C & 1) == 0) {
                // frames are 8-byte aligned: move float2 units
                const uint32_t half_c = C >> 1;
                for (uint32_t e = lane; e < values / 2; e += 64) {
                    const int64_t v = v_row + (half_c == 1 ? e : e / half_c);
                    const uint32_t within = half_c == 1 ? 0u : (e % half_c) * 2u;
                    float2 val = make_float2(0.f, 0.f);
                    if (v >= 0 && v < total_frames) {
                        const float* src = v < hist_frames
                                               ? hist + static_cast<size_t>(v) * C + within
                                               : in + static_cast<size_t>(v - hist_frames) * C + within;
                        val = *reinterpret_cast<const float2*>(src);
                    }
                    *reinterpret_cast<float2*>(dst + 2 * e) = val;
                }
            } else {
                for (uint32_t e = lane; e < values; e += 64) {
                    const int64_t v = v_row + (C == 1 ? e : e / C);
                    const uint32_t within = C == 1 ? 0u : e % C;
                    float val = 0.f;
                    if (v >= 0 && v < total_frames)
                        val = v < hist_frames ? hist[static_cast<size_t>(v) * C + within]
                                              : in[static_cast<size_t>(v - hist_frames) * C + within];
                    dst[e] = val;
                }
            }
        }
    }
    __syncthreads();

    // ---- compute -------------------------------------------------------------------------------
    const uint32_t pl = lane / geo.lp;           // period of this lane inside the block
    const uint32_t gi = lane - pl * geo.lp;      // channel group of this lane
    const bool lane_on = pl < geo.pw;
    const uint32_t pl_c = lane_on ? pl : 0;      // idle lanes shadow lane 0 (no stores)
    const uint32_t lane_base = pl_c * geo.row_stride + gi * CG;
    const_f32_ptr table = (const_f32_ptr)(d.mixed);
    float* __restrict__ out = d.out;
    const uint64_t q = q0 + pl_c;

    for (uint32_t t = wave; t < geo.n_tiles; t += geo.waves) {
        const uint32_t j0 = t * kClassTile;
        const uint32_t ob = static_cast<uint32_t>((static_cast<uint64_t>(j0) * geo.a) / geo.b);
        const_f32_ptr g = table + static_cast<size_t>(t) * geo.row_len * kClassTile;
        Acc<CG> acc;
#pragma unroll
        for (int i = 0; i < (int)kClassTile; ++i)
#pragma unroll
            for (int k = 0; k < CG; ++k) acc.v[i][k] = 0.f;

        // window [ob, ob+row_len) of the lane's period row, spilling into the next row
        const uint32_t n1 = geo.a - ob < geo.row_len ? geo.a - ob : geo.row_len;
        accumulate<CG>(acc, lds + lane_base + ob * C, C, g, n1);
        if (n1 < geo.row_len)
            accumulate<CG>(acc, lds + lane_base + geo.row_stride, C, g + n1 * kClassTile,
                           geo.row_len - n1);

        if (lane_on) {
            const uint64_t m0 = q * geo.b + j0;
#pragma unroll
            for (int i = 0; i < (int)kClassTile; ++i) {
                const uint64_t m = m0 + i;
                if (j0 + i < geo.b && m >= abs_out && m < m_end) {
                    float* o = out + (m - abs_out) * C + gi * CG;
                    if constexpr (CG == 2) {
                        *reinterpret_cast<float2*>(o) = make_float2(acc.v[i][0], acc.v[i][1]);
                    } else {
                        o[0] = acc.v[i][0];
                    }
                }
            }
        }
    }
}

// Outputs whose f64 position fell just below an integer: previous frame, row 1023, frac 0
// (resampler_fir.rs:544, :562-565).  8 lanes per output, as fir_generic.
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

GeoArgs to_args(const PeriodicGeometry& g, uint32_t channels) {
    return GeoArgs{g.a, g.b, g.row_len, g.n_tiles, g.lp, g.pw, g.row_stride, g.waves, channels, g.taps};
}

// Device class tables, shared by every stream on a device with the same polyphase table, rate
// pair, geometry and drift.
struct ClassTableKey {
    int device;
    const void* table;
    uint64_t den;
    uint32_t a, b, row_len;
    uint64_t drift_bits;
    bool operator<(const ClassTableKey& o) const {
        return std::tie(device, table, den, a, b, row_len, drift_bits) <
               std::tie(o.device, o.table, o.den, o.a, o.b, o.row_len, o.drift_bits);
    }
};
struct ClassTableCache {
    std::mutex mu;
    std::map<ClassTableKey, float*> tables;
};
ClassTableCache& class_cache() {
    static ClassTableCache* c = new ClassTableCache;
    return *c;
}

constexpr double kDriftQuantum = 2e-9;  // positions this close share a class table

}  // namespace

PeriodicGeometry periodic_geometry(uint64_t num, uint64_t den, uint32_t taps, uint32_t channels) {
    PeriodicGeometry g;
    if (num == 0 || den == 0 || channels == 0 || channels > 64) return g;
    if (num > (1u << 20) || den > (1u << 20)) return g;
    // max in-tile shift: off(j) = floor(j*num/den); classes of a tile share the first one's base
    // (the pattern repeats every den classes, and tiles of the super period start at multiples
    // of 8, so scanning lcm-many tiles covers all of them; den*8 classes always do).
    uint32_t shift = 0;
    const uint64_t scan = den * kClassTile;
    for (uint64_t j0 = 0; j0 < scan; j0 += kClassTile) {
        const uint64_t s = ((j0 + kClassTile - 1) * num) / den - (j0 * num) / den;
        if (s > shift) shift = static_cast<uint32_t>(s);
        if (j0 > (1u << 16)) break;  // long enough: the bound ceil(7*num/den) is reached early
    }
    const uint32_t bound = static_cast<uint32_t>((7 * num + den - 1) / den);
    if (shift < bound) shift = bound;
    g.taps = taps;
    g.row_len = (taps + shift + 3) / 4 * 4;
    // super period: a >= row_len (a window spans at most two rows) and b >= 8
    uint64_t r = (g.row_len + num - 1) / num;
    if (den * r < kClassTile) r = (kClassTile + den - 1) / den;
    const uint64_t a = num * r, b = den * r;
    if (a > 4096 || b > (1u << 16)) return g;
    g.a = static_cast<uint32_t>(a);
    g.b = static_cast<uint32_t>(b);
    g.n_tiles = (g.b + kClassTile - 1) / kClassTile;

    constexpr uint32_t kLdsTwoPerCu = 80 * 1024;   // two workgroups per CU
    constexpr uint32_t kLdsMax = 160 * 1024;
    auto fit = [&](uint32_t cg) -> bool {
        if (channels % cg != 0) return false;
        const uint32_t lp = channels / cg;
        if (lp > 64) return false;
        const uint32_t pw_max = 64 / lp;
        uint32_t units = g.a * lp;  // read units (cg dwords) per period row
        units += ((lp + 32 - units % 32) % 32);  // units == lp (mod 32): conflict-free lane stride
        const uint32_t stride = units * cg;
        const uint32_t row_bytes = stride * 4;
        uint32_t pw = kLdsTwoPerCu / row_bytes;
        pw = pw > 0 ? pw - 1 : 0;
        if (pw * 4 < pw_max * 3) {  // < 75% of the lanes: take the whole LDS instead
            pw = kLdsMax / row_bytes;
            pw = pw > 0 ? pw - 1 : 0;
        }
        if (pw > pw_max) pw = pw_max;
        if (pw * 2 < pw_max || pw == 0) return false;
        g.cg = cg;
        g.lp = lp;
        g.pw = pw;
        g.row_stride = stride;
        g.lds_bytes = (pw + 1) * row_bytes;
        return true;
    };
    if (!fit(2) && !fit(1)) return g;
    // waves per workgroup: balance the class tiles, keep >= 4 waves
    const uint32_t max_waves = g.lds_bytes > kLdsTwoPerCu ? 16 : 10;
    uint32_t best = 4;
    double best_cost = 1e9;
    for (uint32_t w = 4; w <= max_waves; ++w) {
        const double cost = static_cast<double>((g.n_tiles + w - 1) / w * w) / g.n_tiles;
        if (cost <= best_cost + 1e-9) { best_cost = cost; best = w; }
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

std::vector<float> build_class_table(const std::vector<float>& coeffs, const PeriodicGeometry& g,
                                     uint64_t den, double drift) {
    const uint32_t taps = g.taps;
    std::vector<float> tab(static_cast<size_t>(g.n_tiles) * g.row_len * kClassTile, 0.0f);
    std::vector<float> mixed(taps);
    for (uint32_t j = 0; j < g.b; ++j) {
        // exact fractional position of class j, plus the stream's current f64 drift
        const uint64_t rem = (static_cast<uint64_t>(j) * g.a) % g.b;
        double fract = static_cast<double>(rem) / static_cast<double>(g.b) + drift;
        if (j % den == 0) fract = drift > 0.0 ? drift : 0.0;  // below-integer cases: fix-up kernel
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
        const uint32_t t = j / kClassTile, i = j % kClassTile;
        const uint32_t j0 = t * kClassTile;
        const uint32_t shift = static_cast<uint32_t>((static_cast<uint64_t>(j) * g.a) / g.b -
                                                     (static_cast<uint64_t>(j0) * g.a) / g.b);
        float* base = tab.data() + static_cast<size_t>(t) * g.row_len * kClassTile;
        for (uint32_t k = 0; k < taps; ++k) base[(k + shift) * kClassTile + i] = mixed[k];
    }
    return tab;
}

int periodic_bind(PeriodicState& st, int device, const std::vector<float>& table,
                  const FirMirror& planned, uint32_t channels, hipStream_t stream) {
    (void)stream;
    if (!st.geo_valid) {
        st.geo = periodic_geometry(planned.num(), planned.den(), static_cast<uint32_t>(planned.taps()),
                                   channels);
        st.geo_valid = true;
        st.d_table = nullptr;
    }
    if (!st.geo.ok) return fail(RSMP_ERR_INVALID_ARGUMENT, "periodic kernel: unsupported geometry");
    const double drift = std::round(planned.drift() / kDriftQuantum) * kDriftQuantum;
    if (st.d_table && drift == st.table_drift) return RSMP_OK;
    ClassTableCache& cache = class_cache();
    std::lock_guard<std::mutex> lock(cache.mu);
    uint64_t bits;
    std::memcpy(&bits, &drift, sizeof bits);
    const ClassTableKey key{device, table.data(), planned.den(), st.geo.a, st.geo.b,
                            st.geo.row_len, bits};
    auto it = cache.tables.find(key);
    if (it == cache.tables.end()) {
        const std::vector<float> host = build_class_table(table, st.geo, planned.den(), drift);
        float* dptr = nullptr;
        RSMP_HIP_CHECK(hipMalloc(&dptr, host.size() * sizeof(float)));
        RSMP_HIP_CHECK(hipMemcpy(dptr, host.data(), host.size() * sizeof(float),
                                 hipMemcpyHostToDevice));
        it = cache.tables.emplace(key, dptr).first;
    }
    st.d_table = it->second;
    st.table_drift = drift;
    return RSMP_OK;
}

hipError_t launch_fir_periodic(const FirStreamDesc* d_descs, uint32_t n_streams,
                               const PeriodicGeometry& geo, uint32_t max_blocks,
                               hipStream_t stream) {
    if (n_streams == 0 || max_blocks == 0) return hipSuccess;
    const dim3 grid(max_blocks, n_streams);
    const dim3 block(geo.waves * 64);
    const GeoArgs args = to_args(geo, geo.lp * geo.cg);
    hipError_t e;
    if (geo.cg == 2) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(fir_periodic_kernel<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, geo.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fir_periodic_kernel<2>, grid, block, geo.lds_bytes, stream, d_descs, args);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(fir_periodic_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, geo.lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fir_periodic_kernel<1>, grid, block, geo.lds_bytes, stream, d_descs, args);
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
