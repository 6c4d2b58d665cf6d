// fir_lockstep_api.cpp -- C ABI of the lock-step batch (rsmp_fir_lockstep_*): BASELINE config 4's
// shape, a fixed set of ResamplerFir instances (src/resampler_fir.rs:179-643) that are each fed one
// chunk per step.  Creation sorts the streams by rate pair / table, builds the per-workgroup groups
// and moves the streams' reference state to HBM; a step is one launch of fir_lockstep_kernel with
// constant arguments -- no per-stream host work, no upload, no host-side state machine.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

#include "common.h"
#include "device_util.h"
#include "fir_handle.h"
#include "fir_lockstep.h"

using rsmp::DeviceBuffer;
using rsmp::DeviceGuard;
using rsmp::FirMirrorState;
using rsmp::LockstepGroup;
using rsmp::LockstepStream;

struct rsmp_fir_lockstep {
    int device = 0;
    uint32_t step_frames = 0;
    std::vector<rsmp_fir*> rs;        // caller order
    std::vector<uint32_t> order;      // internal index -> caller index
    std::vector<LockstepGroup> groups;
    std::vector<LockstepStream> streams;   // internal order
    std::vector<uint32_t> channels;        // internal order
    bool in_aligned8 = false;
    DeviceBuffer d_groups, d_streams, d_states, d_cursor, d_counts, d_status, d_order, d_recs, d_peaks;
    uint32_t rec_stride = 0, epoch = 1, step = 0;   // plan-ahead records (fir_lockstep.h)
    uint32_t max_lds = 0;
    bool bound = false;
    hipStream_t last_stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::vector<uint64_t> h_counts;
    std::vector<FirMirrorState> h_states;
    // optional timing of the step launches (rsmp_fir_lockstep_set_profiling): ring of event pairs
    static constexpr int kProfRing = 64;
    bool profiling = false;
    hipEvent_t prof_start[kProfRing] = {}, prof_stop[kProfRing] = {};
    size_t prof_count = 0;
};

namespace {

// A stream's buffered frames alternate between its two history buffers (fir_lockstep.h, LockstepStream):
// the next step (index ls->step) reads `hist` when its index is even.  After any number of steps the handle's
// `cur` is made to name the buffer that holds the frames.
void refresh_history_index(rsmp_fir_lockstep* ls) {
    if (!ls->bound) return;
    for (size_t k = 0; k < ls->rs.size(); ++k) {
        rsmp_fir* r = ls->rs[ls->order[k]];
        const float* live = (ls->step & 1u) ? ls->streams[k].hist_alt : ls->streams[k].hist;
        r->cur = live == r->d_hist[0] ? 0 : 1;
    }
}

int upload_states(rsmp_fir_lockstep* ls) {
    const size_t n = ls->rs.size();
    ls->h_states.resize(n);
    for (size_t k = 0; k < n; ++k) ls->h_states[k] = ls->rs[ls->order[k]]->mirror.state();
    RSMP_HIP_CHECK(hipMemcpy(ls->d_states.get(), ls->h_states.data(), n * sizeof(FirMirrorState),
                             hipMemcpyHostToDevice));
    return RSMP_OK;
}

}  // namespace

extern "C" rsmp_fir_lockstep* rsmp_fir_lockstep_new(rsmp_fir* const* rs, size_t n, size_t step_frames) {
    if (!rs || n == 0 || step_frames == 0 || step_frames > rsmp::kMirrorInputCapacity) {
        rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_new: need streams and 1..4096 frames per step");
        return nullptr;
    }
    for (size_t i = 0; i < n; ++i)
        if (!rs[i] || rs[i]->device != rs[0]->device) {
            rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "lock-step streams must share one device");
            return nullptr;
        }
    {
        std::vector<rsmp_fir*> sorted(rs, rs + n);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
            rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "lock-step batch lists the same stream twice");
            return nullptr;
        }
    }
    DeviceGuard guard(rs[0]->device);
    std::unique_ptr<rsmp_fir_lockstep> ls(new rsmp_fir_lockstep);
    ls->device = rs[0]->device;
    ls->step_frames = static_cast<uint32_t>(step_frames);
    ls->rs.assign(rs, rs + n);
    // Streams that share a polyphase table, a rate pair and a channel count share a class table and
    // a geometry: they become neighbours, then workgroups of `slots` streams.
    // (a stream set to RSMP_FIR_KERNEL_PERIODIC_F32 keeps every product in f32: its own groups)
    typedef std::tuple<const void*, uint32_t, uint32_t, size_t, size_t, bool> Key;
    auto exact_of = [](const rsmp_fir* r) { return r->kernel_mode != RSMP_FIR_KERNEL_AUTO; };
    auto key_of = [&](const rsmp_fir* r) {
        return Key(static_cast<const void*>(r->table.get()), r->in_hz, r->out_hz, r->channels, r->taps, exact_of(r));
    };
    ls->order.resize(n);
    for (size_t i = 0; i < n; ++i) ls->order[i] = static_cast<uint32_t>(i);
    std::stable_sort(ls->order.begin(), ls->order.end(),
                     [&](uint32_t x, uint32_t y) { return key_of(rs[x]) < key_of(rs[y]); });
    ls->streams.resize(n);
    ls->channels.resize(n);
    size_t k = 0;
    while (k < n) {
        const rsmp_fir* r0 = rs[ls->order[k]];
        size_t e = k;
        while (e < n && key_of(rs[ls->order[e]]) == key_of(r0)) ++e;
        const rsmp::LockstepGeometry geo =
            rsmp::lockstep_geometry(r0->mirror.num(), r0->mirror.den(), r0->mirror.ratio(),
                                    static_cast<uint32_t>(r0->taps), static_cast<uint32_t>(r0->channels),
                                    ls->step_frames, !exact_of(r0));
        if (geo.lds_bytes == 0) {
            rsmp::fail(RSMP_ERR_INVALID_ARGUMENT,
                       "lock-step batch: %zu channels x %zu frames per step do not fit the LDS", r0->channels,
                       step_frames);
            return nullptr;
        }
        rsmp::ClassTable ct;
        if (geo.periodic) {
            if (rsmp::class_table_for(ls->device, *r0->table, rsmp::lockstep_class_geometry(geo), 0.0, &ct) != RSMP_OK)
                return nullptr;
        }
        for (size_t first = k; first < e; first += geo.slots) {
            LockstepGroup g;
            std::memset(&g, 0, sizeof g);
            g.first = static_cast<uint32_t>(first);
            g.count = static_cast<uint32_t>(std::min<size_t>(geo.slots, e - first));
            g.channels = static_cast<uint32_t>(r0->channels);
            g.taps = static_cast<uint32_t>(r0->taps);
            g.periodic = geo.periodic ? 1u : 0u;
            g.num = geo.num;
            g.den = geo.den ? geo.den : 1u;
            g.a = geo.a;
            g.b = geo.b ? geo.b : 1u;
            g.row_len = geo.row_len;
            g.n_tiles = geo.n_tiles;
            g.guard_frames = geo.guard_frames;
            g.span_frames = geo.span_frames;
            g.region_frames = geo.region_frames;
            g.max_out = geo.max_out;
            g.wrap_words = geo.wrap_words;
            g.wrap_cap = geo.wrap_cap;
            g.max_cols = geo.max_cols;
            g.class_coef = ct.d_coef;
            g.class_meta = ct.d_meta;
            g.lds_bytes = geo.lds_bytes;
            g.slots = geo.slots;
            g.split = geo.split ? 1u : 0u;
            g.rows = geo.rows;
            g.row_bytes = geo.row_bytes;
            ls->groups.push_back(g);
            if (geo.lds_bytes > ls->max_lds) ls->max_lds = geo.lds_bytes;
            if (rsmp::lockstep_rec_stride(geo.wrap_cap) > ls->rec_stride) ls->rec_stride = rsmp::lockstep_rec_stride(geo.wrap_cap);
        }
        for (size_t i = k; i < e; ++i) {
            const rsmp_fir* r = rs[ls->order[i]];
            if (r->mirror.available() >= r->taps + 8) {
                rsmp::fail(RSMP_ERR_INVALID_ARGUMENT,
                           "lock-step batch: stream %u holds %zu buffered frames (an output-capped call left "
                           "them); drain it first", ls->order[i], r->mirror.available());
                return nullptr;
            }
            ls->channels[i] = static_cast<uint32_t>(r->channels);
        }
        k = e;
    }
    // Workgroup order = dispatch order: with more workgroups than CUs (two fit a CU) number k + CUs becomes the second
    // tenant of the CU that took number k.  The slow geometries first and the quick ones last pairs each slow workgroup
    // with a quick one (or leaves it alone, see below) instead of with its own kind -- a step ends with its slowest workgroup, and
    // two slow tenants slow each other (`tools/ls_trace.py`: the 20-tile and the 505-row images end at 46-52 k cycles, the
    // one-stream 48 -> 96 kHz ones at 27 k).  Cost: matrix units + rows to stage, a packed image's bank conflicts on top.
    {
        auto cost = [](const LockstepGroup& g) {
            const double units = static_cast<double>(g.n_tiles) * ((g.max_cols + 15) / 16);
            return units + g.count * (g.split ? g.rows : g.region_frames) / 64.0 + (g.split && g.row_bytes == rsmp::kLsImageRowBytesPacked ? 10.0 : 0.0);
        };
        std::stable_sort(ls->groups.begin(), ls->groups.end(),
                         [&](const LockstepGroup& x, const LockstepGroup& y) { return cost(x) > cost(y); });
        // ... and the slowest of all ALONE: with n workgroups on c CUs the indices n - c .. c - 1 get no second tenant, so the
        // order is [next slowest: first tenants][slowest: alone][quickest: second tenants] (0.0184 -> 0.0181 ms per step)
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ls->device);
        const size_t n = ls->groups.size(), c = static_cast<size_t>(cus);
        if (n > c && n < 2 * c) {
            const size_t second = n - c, alone = c - second;
            std::vector<LockstepGroup> o;
            o.insert(o.end(), ls->groups.begin() + alone, ls->groups.begin() + alone + second);
            o.insert(o.end(), ls->groups.begin(), ls->groups.begin() + alone);
            o.insert(o.end(), ls->groups.begin() + alone + second, ls->groups.end());
            ls->groups.swap(o);
        }
    }
    if (ls->d_groups.reserve(ls->groups.size() * sizeof(LockstepGroup)) != hipSuccess ||
        ls->d_streams.reserve(n * sizeof(LockstepStream)) != hipSuccess ||
        ls->d_states.reserve(n * sizeof(FirMirrorState)) != hipSuccess ||
        ls->d_cursor.reserve(n * sizeof(uint64_t)) != hipSuccess ||
        ls->d_counts.reserve(2 * n * sizeof(uint64_t)) != hipSuccess ||
        ls->d_status.reserve(n * sizeof(uint32_t)) != hipSuccess ||
        ls->d_order.reserve(n * sizeof(uint32_t)) != hipSuccess ||
        ls->d_recs.reserve(2 * n * static_cast<size_t>(ls->rec_stride)) != hipSuccess ||
        ls->d_peaks.reserve(n * 16) != hipSuccess ||
        hipStreamCreateWithFlags(&ls->own_stream, hipStreamNonBlocking) != hipSuccess) {
        rsmp::fail(RSMP_ERR_HIP, "lock-step batch: cannot allocate device state");
        return nullptr;
    }
    // every stream's earlier launches (which wrote its buffered frames) must be complete
    for (size_t i = 0; i < n; ++i) {
        (void)hipStreamSynchronize(rs[i]->stream);
        if (rs[i]->last_stream_valid) (void)hipStreamSynchronize(rs[i]->last_stream);
    }
    if (hipMemcpy(ls->d_groups.get(), ls->groups.data(), ls->groups.size() * sizeof(LockstepGroup),
                  hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ls->d_order.get(), ls->order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(ls->d_recs.get(), 0, 2 * n * static_cast<size_t>(ls->rec_stride)) != hipSuccess ||
        hipMemset(ls->d_peaks.get(), 0, n * 16) != hipSuccess ||
        hipMemset(ls->d_cursor.get(), 0, n * sizeof(uint64_t)) != hipSuccess ||
        hipMemset(ls->d_counts.get(), 0, 2 * n * sizeof(uint64_t)) != hipSuccess ||
        hipMemset(ls->d_status.get(), 0, n * sizeof(uint32_t)) != hipSuccess ||
        upload_states(ls.get()) != RSMP_OK) {
        rsmp::fail(RSMP_ERR_HIP, "lock-step batch: cannot initialise device state");
        return nullptr;
    }
    return ls.release();
}

extern "C" void rsmp_fir_lockstep_free(rsmp_fir_lockstep* ls) {
    if (!ls) return;
    DeviceGuard guard(ls->device);
    (void)rsmp_fir_lockstep_sync(ls);
    if (ls->own_stream) (void)hipStreamDestroy(ls->own_stream);
    for (hipEvent_t e : ls->prof_start) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ls->prof_stop) if (e) (void)hipEventDestroy(e);
    delete ls;
}

extern "C" size_t rsmp_fir_lockstep_size(const rsmp_fir_lockstep* ls) { return ls ? ls->rs.size() : 0; }
extern "C" size_t rsmp_fir_lockstep_workgroups(const rsmp_fir_lockstep* ls) { return ls ? ls->groups.size() : 0; }
extern "C" size_t rsmp_fir_lockstep_split_workgroups(const rsmp_fir_lockstep* ls) {
    size_t n = 0;
    if (ls) for (const LockstepGroup& g : ls->groups) n += g.split ? 1 : 0;
    return n;
}

extern "C" int rsmp_fir_lockstep_bind(rsmp_fir_lockstep* ls, const float* const* d_in, float* const* d_out,
                                      const size_t* out_caps) {
    if (!ls || !d_in || !d_out || !out_caps)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_bind: null argument");
    DeviceGuard guard(ls->device);
    const size_t n = ls->rs.size();
    if (ls->last_stream) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    refresh_history_index(ls);
    bool aligned8 = true;
    for (size_t k = 0; k < n; ++k) {
        const uint32_t i = ls->order[k];
        const rsmp_fir* r = ls->rs[i];
        if (out_caps[i] % r->channels != 0)
            return rsmp::fail(RSMP_ERR_INVALID_OUTPUT_BUFFER_SIZE, "Output buffer size is invalid");
        // the reference's documented sizing (resampler_fir.rs:456-465); with it a step never leaves more
        // than taps - 1 frames buffered, which is what bounds a stream's LDS span
        if (out_caps[i] < rsmp_fir_buffer_size_output(r))
            return rsmp::fail(RSMP_ERR_INVALID_OUTPUT_BUFFER_SIZE,
                              "lock-step batch: stream %u needs room for buffer_size_output() = %zu values per step",
                              i, rsmp_fir_buffer_size_output(r));
        LockstepStream& s = ls->streams[k];
        if (reinterpret_cast<uintptr_t>(d_in[i]) % 8 != 0) aligned8 = false;
        s.in = d_in[i];
        s.out = d_out[i];
        s.hist = r->d_hist[(ls->step & 1u) ? r->cur ^ 1 : r->cur];       // the next step reads the handle's current buffer
        s.hist_alt = r->d_hist[(ls->step & 1u) ? r->cur : r->cur ^ 1];
        s.coeffs = r->d_coeffs;
        s.out_cap_frames = out_caps[i] / r->channels;
    }
    if (ls->last_stream) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    RSMP_HIP_CHECK(hipMemcpy(ls->d_streams.get(), ls->streams.data(), n * sizeof(LockstepStream),
                             hipMemcpyHostToDevice));
    RSMP_HIP_CHECK(hipMemset(ls->d_cursor.get(), 0, n * sizeof(uint64_t)));   // nothing has been appended to the new buffers
    ls->bound = true;
    ls->in_aligned8 = aligned8;
    ++ls->epoch;   // plans made ahead assumed the previous output capacities
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_step(rsmp_fir_lockstep* ls, size_t in_frames, size_t in_offset_frames,
                                      const uint32_t* d_in_frames, int append, void* stream) {
    if (!ls || !ls->bound)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_step: bind buffers first");
    if (in_frames > ls->step_frames)
        return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE,
                          "lock-step batch: %zu frames offered, created for %u per step", in_frames,
                          ls->step_frames);
    DeviceGuard guard(ls->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ls->own_stream;
    if (ls->last_stream && ls->last_stream != s) {
        // steps of one batch are ordered: a change of stream waits for the previous step
        RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    }
    rsmp::LockstepArgs a;
    a.groups = ls->d_groups.as<LockstepGroup>();
    a.streams = ls->d_streams.as<LockstepStream>();
    a.states = ls->d_states.as<FirMirrorState>();
    a.out_cursor = ls->d_cursor.as<uint64_t>();
    a.counts = ls->d_counts.as<uint64_t>();
    a.status = ls->d_status.as<uint32_t>();
    a.order = ls->d_order.as<uint32_t>();
    a.in_frames_per_stream = d_in_frames;
    a.in_offset = in_offset_frames;
    a.in_frames = static_cast<uint32_t>(in_frames);
    a.append = append ? 1u : 0u;
    a.in_aligned8 = ls->in_aligned8 ? 1u : 0u;
    a.trace = nullptr;
    a.recs = ls->d_recs.as<char>();
    a.peaks = ls->d_peaks.as<uint32_t>();
    a.rec_stride = ls->rec_stride;
    a.n_streams = static_cast<uint32_t>(ls->rs.size());
    a.epoch = ls->epoch;
    a.step = ls->step++;
    if (ls->profiling)
        RSMP_HIP_CHECK(hipEventRecord(ls->prof_start[ls->prof_count % rsmp_fir_lockstep::kProfRing], s));
    RSMP_HIP_CHECK(rsmp::launch_fir_lockstep(a, static_cast<uint32_t>(ls->groups.size()), ls->max_lds, s));
    if (ls->profiling) {
        RSMP_HIP_CHECK(hipEventRecord(ls->prof_stop[ls->prof_count % rsmp_fir_lockstep::kProfRing], s));
        ++ls->prof_count;
    }
    ls->last_stream = s;
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_counts(rsmp_fir_lockstep* ls, size_t* consumed, size_t* produced) {
    if (!ls) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_counts: null batch");
    DeviceGuard guard(ls->device);
    const size_t n = ls->rs.size();
    if (ls->last_stream) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    ls->h_counts.resize(2 * n);
    RSMP_HIP_CHECK(hipMemcpy(ls->h_counts.data(), ls->d_counts.get(), 2 * n * sizeof(uint64_t),
                             hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n; ++k) {
        const uint32_t i = ls->order[k];
        if (consumed) consumed[i] = static_cast<size_t>(ls->h_counts[2 * k]);
        if (produced) produced[i] = static_cast<size_t>(ls->h_counts[2 * k + 1]);
    }
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_status(rsmp_fir_lockstep* ls, uint32_t* status) {
    if (!ls || !status) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_status: null argument");
    DeviceGuard guard(ls->device);
    const size_t n = ls->rs.size();
    if (ls->last_stream) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    std::vector<uint32_t> h(n);
    RSMP_HIP_CHECK(hipMemcpy(h.data(), ls->d_status.get(), n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n; ++k) status[ls->order[k]] = h[k];
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_sync(rsmp_fir_lockstep* ls) {
    if (!ls) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_sync: null batch");
    DeviceGuard guard(ls->device);
    const size_t n = ls->rs.size();
    if (ls->last_stream) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    ls->h_states.resize(n);
    RSMP_HIP_CHECK(hipMemcpy(ls->h_states.data(), ls->d_states.get(), n * sizeof(FirMirrorState),
                             hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n; ++k) ls->rs[ls->order[k]]->mirror.set_state(ls->h_states[k]);
    refresh_history_index(ls);
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_reset(rsmp_fir_lockstep* ls) {
    if (!ls) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_reset: null batch");
    DeviceGuard guard(ls->device);
    const size_t n = ls->rs.size();
    if (ls->last_stream) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    for (rsmp_fir* r : ls->rs) r->mirror.reset();   // resampler_fir.rs:638-642
    RSMP_HIP_CHECK(hipMemset(ls->d_cursor.get(), 0, n * sizeof(uint64_t)));
    RSMP_HIP_CHECK(hipMemset(ls->d_status.get(), 0, n * sizeof(uint32_t)));
    ++ls->epoch;   // plans made ahead belong to the old states
    return upload_states(ls);
}

extern "C" int rsmp_fir_lockstep_set_profiling(rsmp_fir_lockstep* ls, int enable) {
    if (!ls) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_set_profiling: null batch");
    DeviceGuard guard(ls->device);
    if (enable && !ls->prof_start[0])
        for (int i = 0; i < rsmp_fir_lockstep::kProfRing; ++i) {
            RSMP_HIP_CHECK(hipEventCreate(&ls->prof_start[i]));
            RSMP_HIP_CHECK(hipEventCreate(&ls->prof_stop[i]));
        }
    ls->profiling = enable != 0;
    ls->prof_count = 0;
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_mean_kernel_ms(rsmp_fir_lockstep* ls, float* ms, size_t* launches) {
    if (!ls || !ms || ls->prof_count == 0)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_mean_kernel_ms: no profiled step");
    DeviceGuard guard(ls->device);
    const size_t ring = rsmp_fir_lockstep::kProfRing;
    const size_t n = ls->prof_count < ring ? ls->prof_count : ring;
    RSMP_HIP_CHECK(hipEventSynchronize(ls->prof_stop[(ls->prof_count - 1) % ring]));
    double sum = 0.0;
    for (size_t k = 0; k < n; ++k) {
        const size_t i = (ls->prof_count - 1 - k) % ring;
        float t = 0.f;
        RSMP_HIP_CHECK(hipEventElapsedTime(&t, ls->prof_start[i], ls->prof_stop[i]));
        sum += t;
    }
    *ms = static_cast<float>(sum / static_cast<double>(n));
    if (launches) *launches = n;
    return RSMP_OK;
}
