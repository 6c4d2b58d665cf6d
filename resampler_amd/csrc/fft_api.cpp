// fft_api.cpp -- ResamplerFft front-end on the GPU: handles, state, launch assembly, C ABI.
//
// Mirrors src/resampler_fft.rs of the reference: ResamplerFft::new (:75-119), chunk_size_input /
// chunk_size_output (:135-145), delay (:151-153), resample (:182-240).  Stream state is the
// per-channel overlap row (`overlaps`, :51), resident in HBM; everything else the reference keeps
// per instance (scratch pads, spectra) lives in LDS for the duration of a block.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_set>
#include <vector>

#include "common.h"
#include "device_util.h"
#include "fft_kernels.h"
#include "fft_plan.h"

using rsmp::DeviceBuffer;
using rsmp::DeviceGuard;
using rsmp::FftPlanDev;
using rsmp::FftStreamDesc;
using rsmp::PinnedBuffer;

namespace {

// Device-side half of the reference's FFT_CACHE (resampler_fft.rs:35-36, :306-318): one plan
// image per (device, rate pair), shared by every instance.
struct DevicePlan {
    rsmp::FftResamplerPlan host;
    FftPlanDev dev;
};
struct PlanCache {
    std::mutex mu;
    std::map<std::tuple<int, uint32_t, uint32_t>, std::shared_ptr<DevicePlan>> plans;
};
PlanCache& plan_cache() {
    static PlanCache* c = new PlanCache;
    return *c;
}

template <class T>
int upload(const std::vector<T>& v, const T** out) {
    T* d = nullptr;
    const size_t bytes = (v.empty() ? 1 : v.size()) * sizeof(T);
    RSMP_HIP_CHECK(hipMalloc(&d, bytes));
    if (!v.empty()) RSMP_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = d;
    return RSMP_OK;
}

int get_plan(int device, uint32_t in_hz, uint32_t out_hz, std::shared_ptr<DevicePlan>* out) {
    PlanCache& c = plan_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    const auto key = std::make_tuple(device, in_hz, out_hz);
    auto it = c.plans.find(key);
    if (it != c.plans.end()) { *out = it->second; return RSMP_OK; }
    // a plan that fails half way releases what it had uploaded (a cached plan lives for the process)
    auto release = [](DevicePlan* dp) {
        const FftPlanDev& dv = dp->dev;
        for (const void* q : {static_cast<const void*>(dv.tw_f), static_cast<const void*>(dv.tw_i),
                              static_cast<const void*>(dv.rc_f), static_cast<const void*>(dv.rc_i),
                              static_cast<const void*>(dv.filter), static_cast<const void*>(dv.chirp_f),
                              static_cast<const void*>(dv.chirp_i)})
            if (q) (void)hipFree(const_cast<void*>(q));
        delete dp;
    };
    std::shared_ptr<DevicePlan> p(new DevicePlan, release);
    std::memset(&p->dev, 0, sizeof p->dev);
    p->host = rsmp::make_fft_resampler_plan(in_hz, out_hz);
    if (!p->host.ok)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "ResamplerFft: unsupported rate pair %u -> %u", in_hz, out_hz);
    const rsmp::FftResamplerPlan& h = p->host;
    if (h.forward.stages.size() > rsmp::kMaxFftStages || h.inverse.stages.size() > rsmp::kMaxFftStages)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "ResamplerFft: too many FFT stages");
    FftPlanDev& d = p->dev;
    d.fft_in = static_cast<uint32_t>(h.fft_in);
    d.fft_out = static_cast<uint32_t>(h.fft_out);
    d.n_stages_f = static_cast<uint32_t>(h.forward.stages.size());
    d.n_stages_i = static_cast<uint32_t>(h.inverse.stages.size());
    for (size_t s = 0; s < h.forward.stages.size(); ++s) {
        d.radix_f[s] = static_cast<uint32_t>(h.forward.stages[s]);
        d.tw_off_f[s] = h.forward.stage_twiddle_offset[s];
    }
    for (size_t s = 0; s < h.inverse.stages.size(); ++s) {
        d.radix_i[s] = static_cast<uint32_t>(h.inverse.stages[s]);
        d.tw_off_i[s] = h.inverse.stage_twiddle_offset[s];
    }
    static_assert(sizeof(rsmp::Complex32) == sizeof(float2), "Complex32 is float2");
    int rc;
    const rsmp::Complex32* ptr = nullptr;
    if ((rc = upload(h.forward.stage_twiddles, &ptr)) != RSMP_OK) return rc;
    d.tw_f = reinterpret_cast<const float2*>(ptr);
    if ((rc = upload(h.inverse.stage_twiddles, &ptr)) != RSMP_OK) return rc;
    d.tw_i = reinterpret_cast<const float2*>(ptr);
    if ((rc = upload(h.forward.rc_twiddles, &ptr)) != RSMP_OK) return rc;
    d.rc_f = reinterpret_cast<const float2*>(ptr);
    if ((rc = upload(h.inverse.rc_twiddles, &ptr)) != RSMP_OK) return rc;
    d.rc_i = reinterpret_cast<const float2*>(ptr);
    d.n_rc_f = static_cast<uint32_t>(h.forward.rc_twiddles.size());
    d.n_rc_i = static_cast<uint32_t>(h.inverse.rc_twiddles.size());
    d.new_length = static_cast<uint32_t>(h.new_length);
    // exp(-2 pi i n / 2N) for the two block lengths (f64 -> f32 like every twiddle, radix_fft.rs:251-258): the odd-bin
    // chain of the two-channel kernel (fft_pair.hip)
    auto chirp = [](size_t n2) {
        std::vector<rsmp::Complex32> v(n2);
        for (size_t n = 0; n < n2; ++n) {
            const double a = -3.14159265358979323846264338327950288 * static_cast<double>(n) / static_cast<double>(n2);
            v[n].re = static_cast<float>(std::cos(a));
            v[n].im = static_cast<float>(std::sin(a));
        }
        return v;
    };
    if ((rc = upload(chirp(h.fft_in), &ptr)) != RSMP_OK) return rc;
    d.chirp_f = reinterpret_cast<const float2*>(ptr);
    if ((rc = upload(chirp(h.fft_out), &ptr)) != RSMP_OK) return rc;
    d.chirp_i = reinterpret_cast<const float2*>(ptr);
    d.lds_complex = static_cast<uint32_t>((h.fft_in > h.fft_out ? h.fft_in : h.fft_out) + 1);
    if (rsmp::fft_ola_lds_bytes(d, 1) > 160 * 1024 && (rsmp::fft_big_lds_bytes(d) > 160 * 1024 || d.lds_complex > 12288 + 1))
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT,
                          "ResamplerFft: blocks of %zu -> %zu frames do not fit the 160 KiB LDS",
                          h.fft_in, h.fft_out);
    // Filter spectrum = forward transform of the windowed sinc, by the same device transform the
    // resampler uses (resampler_fft.rs:361-376).
    const float* d_time = nullptr;
    if ((rc = upload(h.filter_time, &d_time)) != RSMP_OK) return rc;
    float2* d_spec = nullptr;
    RSMP_HIP_CHECK(hipMalloc(&d_spec, (h.fft_in + 1) * sizeof(float2)));
    RSMP_HIP_CHECK(rsmp::launch_fft_filter_spectrum(d, d_time, d_spec, nullptr));
    RSMP_HIP_CHECK(hipDeviceSynchronize());
    RSMP_HIP_CHECK(hipFree(const_cast<float*>(d_time)));
    d.filter = d_spec;
    c.plans.emplace(key, p);
    *out = p;
    return RSMP_OK;
}

}  // namespace

struct LaunchEvent {
    hipEvent_t ev = nullptr;
    ~LaunchEvent() { if (ev) (void)hipEventDestroy(ev); }
};

struct rsmp_fft {
    int device = 0;
    size_t channels = 0;
    uint32_t in_hz = 0, out_hz = 0;
    std::shared_ptr<DevicePlan> plan;
    float* d_overlap = nullptr;   // 2 x [channels][fft_out] (ping-pong per launch), zero at construction (:81)
    int cur = 0;
    hipStream_t stream = nullptr;
    hipStream_t last_stream = nullptr;   // the stream of the handle's previous launch (launches of one handle are ordered)
    bool last_stream_valid = false;
    // recorded behind the handle's most recent launch, on that launch's stream: the LEADER's event of that launch, shared
    // by the batch's handles (one record per launch, not one per stream; it outlives the leader through the reference)
    std::shared_ptr<LaunchEvent> last_launch, launch_ev;
    hipEvent_t desc_copied = nullptr;
    bool desc_pending = false;
    PinnedBuffer h_desc;
    DeviceBuffer d_desc;
    // Small descriptor tables (up to 16 KB: ~340 streams) are not uploaded at all: the kernels read them out of mapped,
    // coherent host memory -- a copy-engine operation and its wait less in front of every launch (a launch of the bench's 64
    // streams: ms per step - kernel ms 14 -> 8 us).  A ring of slots, each reusable once the launch that read it has passed
    // its event.
    static constexpr int kDescSlots = 4;
    PinnedBuffer h_desc_ring[kDescSlots];
    hipEvent_t desc_read[kDescSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool desc_slot_used[kDescSlots] = {false, false, false, false};
    int desc_slot = 0;
    DeviceBuffer d_stage_in, d_stage_out;
    rsmp::PinnedBuffer h_stage_in, h_stage_out;   // small calls: mapped host memory instead of copy-engine transfers
    bool profiling = false;
    hipEvent_t prof_start = nullptr, prof_stop = nullptr;
    bool prof_valid = false;
};

namespace {

struct FftJob {
    rsmp_fft* r;
    const float* d_in;
    float* d_out;
    size_t n_blocks;
};

int launch_fft_jobs(rsmp_fft* leader, const std::vector<FftJob>& jobs, hipStream_t stream, uint32_t pcm_bits = 0) {
    const size_t n = jobs.size();
    const size_t bytes = n * sizeof(FftStreamDesc);
    // A handle's launches are ordered (&mut self in the reference): launch k + 1 reads the overlap rows launch k
    // writes, so a caller that changes streams between two calls waits here for the first -- as the FIR path does.
    // A handle listed twice in one batch would read one old state through both descriptors: refused.
    std::unordered_set<const rsmp_fft*> seen;
    for (size_t i = 0; i < n; ++i)
        if (!seen.insert(jobs[i].r).second)
            return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "ResamplerFft batch: handle %zu is listed twice", i);
    for (size_t i = 0; i < n; ++i) {
        rsmp_fft* h = jobs[i].r;
        // (an event, not the previous stream's handle: the caller may have destroyed that stream by now, and a wait on an
        // event neither blocks the host nor breaks a stream capture)
        if (h->last_stream_valid && h->last_stream != stream && h->last_launch)
            RSMP_HIP_CHECK(rsmp::stream_wait_event(stream, h->last_launch->ev));
        h->last_stream = stream;
        h->last_stream_valid = true;
    }
    // (a launch of a block or two per stream -- a streaming call -- keeps the upload: the table read over the host link at the
    // kernel's start is on its critical path there, 33.6 against 32.1 us per call)
    size_t most_blocks = 0;
    for (size_t i = 0; i < n; ++i) most_blocks = std::max(most_blocks, jobs[i].n_blocks);
    const bool direct = bytes <= 16 * 1024 && most_blocks >= 4;
    int slot = 0;
    FftStreamDesc* h = nullptr;
    if (direct) {
        slot = leader->desc_slot;
        leader->desc_slot = (slot + 1) % rsmp_fft::kDescSlots;
        if (!leader->desc_read[slot]) RSMP_HIP_CHECK(hipEventCreateWithFlags(&leader->desc_read[slot], hipEventDisableTiming));
        if (leader->desc_slot_used[slot]) {   // the launch that read this slot (four launches ago) is through
            RSMP_HIP_CHECK(hipEventSynchronize(leader->desc_read[slot]));
            leader->desc_slot_used[slot] = false;
        }
        RSMP_HIP_CHECK(leader->h_desc_ring[slot].reserve(16 * 1024));
        h = leader->h_desc_ring[slot].as<FftStreamDesc>();
    } else {
        if (leader->desc_pending) {
            RSMP_HIP_CHECK(hipEventSynchronize(leader->desc_copied));
            leader->desc_pending = false;
        }
        RSMP_HIP_CHECK(leader->h_desc.reserve(bytes));
        if (bytes > leader->d_desc.capacity()) {
            RSMP_HIP_CHECK(hipStreamSynchronize(stream));
            RSMP_HIP_CHECK(leader->d_desc.reserve(bytes));
        }
        h = leader->h_desc.as<FftStreamDesc>();
    }
    uint32_t max_blocks = 0, max_channels = 0, min_channels = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) {
        const FftJob& j = jobs[i];
        h[i].in = j.d_in;
        h[i].out = j.d_out;
        const size_t ov = j.r->channels * j.r->plan->host.fft_out;
        h[i].overlap = j.r->d_overlap + j.r->cur * ov;
        h[i].overlap_next = j.r->d_overlap + (j.r->cur ^ 1) * ov;
        h[i].n_blocks = static_cast<uint32_t>(j.n_blocks);
        h[i].channels = static_cast<uint32_t>(j.r->channels);
        h[i].in_bits = pcm_bits;
        h[i].pad = 0;
        if (h[i].n_blocks > max_blocks) max_blocks = h[i].n_blocks;
        if (h[i].channels > max_channels) max_channels = h[i].channels;
        if (h[i].channels < min_channels) min_channels = h[i].channels;
    }
    const FftStreamDesc* d_descs = h;   // (mapped host memory: the same address on the device)
    if (!direct) {
        RSMP_HIP_CHECK(hipMemcpyAsync(leader->d_desc.get(), h, bytes, hipMemcpyHostToDevice, stream));
        RSMP_HIP_CHECK(rsmp::event_record(leader->desc_copied, stream));
        leader->desc_pending = true;
        d_descs = leader->d_desc.as<FftStreamDesc>();
    }
    if (leader->profiling) RSMP_HIP_CHECK(rsmp::event_record(leader->prof_start, stream));
    {
        const hipError_t e = rsmp::launch_fft_ola(leader->plan->dev, d_descs, static_cast<uint32_t>(n),
                                                  max_blocks, max_channels, min_channels, stream, pcm_bits);
        if (e == hipErrorNotSupported && pcm_bits != 0)
            return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "ResamplerFft: PCM input is read in place by the two-channel wave kernel only "
                                                       "(this rate pair / channel count has none): convert with rsmp_pcm_to_stereo_f32_device first");
        RSMP_HIP_CHECK(e);
    }
    if (leader->profiling) {
        RSMP_HIP_CHECK(rsmp::event_record(leader->prof_stop, stream));
        leader->prof_valid = true;
    }
    if (!leader->launch_ev) {
        leader->launch_ev = std::make_shared<LaunchEvent>();
        RSMP_HIP_CHECK(hipEventCreateWithFlags(&leader->launch_ev->ev, hipEventDisableTiming));
    }
    RSMP_HIP_CHECK(rsmp::event_record(leader->launch_ev->ev, stream));
    if (direct) {   // the slot may be rewritten once this launch's kernels have read it
        RSMP_HIP_CHECK(rsmp::event_record(leader->desc_read[slot], stream));
        leader->desc_slot_used[slot] = true;
    }
    for (const FftJob& j : jobs) {
        if (j.n_blocks != 0) j.r->cur ^= 1;   // (a stream without blocks in this launch keeps its state where it is)
        j.r->last_launch = leader->launch_ev;
    }
    return RSMP_OK;
}

int check_lds(const rsmp_fft* r) {
    // (what does not fit the two-buffer kernels with the overlap rows in LDS runs on the one-buffer kernel)
    if (rsmp::fft_ola_lds_bytes(r->plan->dev, static_cast<uint32_t>(r->channels)) > 160 * 1024 &&
        (rsmp::fft_big_lds_bytes(r->plan->dev) > 160 * 1024 || r->plan->dev.lds_complex > 12288 + 1))
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "ResamplerFft: %zu channels of this block size exceed the LDS",
                          r->channels);
    return RSMP_OK;
}

}  // namespace

// ================================ C ABI ==========================================================
extern "C" rsmp_fft* rsmp_fft_new(size_t channels, int input_rate, int output_rate, int device) {
    const uint32_t in_hz = rsmp_sample_rate_hz(input_rate), out_hz = rsmp_sample_rate_hz(output_rate);
    if (!in_hz || !out_hz || channels == 0 || channels > 4096) {
        rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "ResamplerFft::new: invalid channels / SampleRate");
        return nullptr;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        rsmp::fail(RSMP_ERR_NO_DEVICE, "ResamplerFft: no HIP device (this engine has no CPU path)");
        return nullptr;
    }
    if (device < 0 || device >= n_dev) {
        rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "ResamplerFft: device %d out of range", device);
        return nullptr;
    }
    DeviceGuard guard(device);
    std::unique_ptr<rsmp_fft> r(new rsmp_fft);
    r->device = device;
    r->channels = channels;
    r->in_hz = in_hz;
    r->out_hz = out_hz;
    if (get_plan(device, in_hz, out_hz, &r->plan) != RSMP_OK) return nullptr;
    if (check_lds(r.get()) != RSMP_OK) return nullptr;
    const size_t ov_bytes = 2 * channels * r->plan->host.fft_out * sizeof(float);
    if (hipMalloc(&r->d_overlap, ov_bytes) != hipSuccess || hipMemset(r->d_overlap, 0, ov_bytes) != hipSuccess ||
        hipStreamSynchronize(nullptr) != hipSuccess ||   // (the handle's stream is non-blocking: no implicit order)
        hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&r->desc_copied, hipEventDisableTiming) != hipSuccess) {
        rsmp::fail(RSMP_ERR_HIP, "ResamplerFft: cannot allocate stream state");
        if (r->d_overlap) (void)hipFree(r->d_overlap);
        return nullptr;
    }
    return r.release();
}

extern "C" void rsmp_fft_free(rsmp_fft* r) {
    if (!r) return;
    DeviceGuard guard(r->device);
    (void)hipDeviceSynchronize();
    if (r->d_overlap) (void)hipFree(r->d_overlap);
    if (r->desc_copied) (void)hipEventDestroy(r->desc_copied);
    for (hipEvent_t e : r->desc_read)
        if (e) (void)hipEventDestroy(e);
    if (r->prof_start) (void)hipEventDestroy(r->prof_start);
    if (r->prof_stop) (void)hipEventDestroy(r->prof_stop);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

extern "C" size_t rsmp_fft_chunk_size_input(const rsmp_fft* r) { return r->plan->host.fft_in * r->channels; }
extern "C" size_t rsmp_fft_chunk_size_output(const rsmp_fft* r) { return r->plan->host.fft_out * r->channels; }
extern "C" size_t rsmp_fft_delay(const rsmp_fft* r) { return r->plan->host.fft_in / 2; }
extern "C" size_t rsmp_fft_channels(const rsmp_fft* r) { return r->channels; }

extern "C" int rsmp_fft_set_profiling(rsmp_fft* r, int enable) {
    DeviceGuard guard(r->device);
    if (enable && !r->prof_start) {
        RSMP_HIP_CHECK(hipEventCreate(&r->prof_start));
        RSMP_HIP_CHECK(hipEventCreate(&r->prof_stop));
    }
    r->profiling = enable != 0;
    r->prof_valid = false;
    return RSMP_OK;
}

extern "C" int rsmp_fft_last_kernel_ms(rsmp_fft* r, float* ms) {
    DeviceGuard guard(r->device);
    if (!r->prof_valid || !ms)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fft_last_kernel_ms: no profiled launch");
    RSMP_HIP_CHECK(hipEventSynchronize(r->prof_stop));
    RSMP_HIP_CHECK(hipEventElapsedTime(ms, r->prof_start, r->prof_stop));
    return RSMP_OK;
}

// n_chunks consecutive chunks == n_chunks calls of ResamplerFft::resample (resampler_fft.rs:182-240).
extern "C" int rsmp_fft_resample_bulk_device(rsmp_fft* r, const float* d_in, size_t in_len,
                                             float* d_out, size_t out_len, size_t n_chunks,
                                             void* stream) {
    DeviceGuard guard(r->device);
    if (n_chunks == 0) return RSMP_OK;
    // :186-192 -- `>=`, extra values are ignored
    if (in_len < n_chunks * rsmp_fft_chunk_size_input(r))
        return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE, "Input buffer size is invalid");
    if (out_len < n_chunks * rsmp_fft_chunk_size_output(r))
        return rsmp::fail(RSMP_ERR_INVALID_OUTPUT_BUFFER_SIZE, "Output buffer size is invalid");
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : r->stream;
    std::vector<FftJob> jobs{FftJob{r, d_in, d_out, n_chunks}};
    return launch_fft_jobs(r, jobs, s);
}

extern "C" int rsmp_fft_resample_device(rsmp_fft* r, const float* d_in, size_t in_len, float* d_out,
                                        size_t out_len, void* stream) {
    return rsmp_fft_resample_bulk_device(r, d_in, in_len, d_out, out_len, 1, stream);
}

extern "C" int rsmp_fft_resample_bulk(rsmp_fft* r, const float* in, size_t in_len, float* out,
                                      size_t out_len, size_t n_chunks) {
    DeviceGuard guard(r->device);
    const size_t need_in = n_chunks * rsmp_fft_chunk_size_input(r);
    const size_t need_out = n_chunks * rsmp_fft_chunk_size_output(r);
    if (in_len < need_in)
        return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE, "Input buffer size is invalid");
    if (out_len < need_out)
        return rsmp::fail(RSMP_ERR_INVALID_OUTPUT_BUFFER_SIZE, "Output buffer size is invalid");
    if (n_chunks == 0) return RSMP_OK;
    RSMP_HIP_CHECK(hipStreamSynchronize(r->stream));
    // (small calls through mapped host memory, as rsmp_fir_resample: fir_api.cpp)
    static const size_t zero_copy_max = [] {
        const char* e = getenv("RSMP_FIR_ZEROCOPY_MAX");
        return e ? static_cast<size_t>(atoll(e)) : static_cast<size_t>(256 * 1024);
    }();
    if (need_in * sizeof(float) <= zero_copy_max && need_out * sizeof(float) <= zero_copy_max) {
        RSMP_HIP_CHECK(r->h_stage_in.reserve(need_in * sizeof(float), false));
        RSMP_HIP_CHECK(r->h_stage_out.reserve(need_out * sizeof(float), false));
        std::memcpy(r->h_stage_in.get(), in, need_in * sizeof(float));
        float* d_i = nullptr;
        float* d_o = nullptr;
        RSMP_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_i), r->h_stage_in.get(), 0));
        RSMP_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_o), r->h_stage_out.get(), 0));
        const int rc = rsmp_fft_resample_bulk_device(r, d_i, need_in, d_o, need_out, n_chunks, r->stream);
        if (rc != RSMP_OK) return rc;
        RSMP_HIP_CHECK(hipStreamSynchronize(r->stream));
        std::memcpy(out, r->h_stage_out.get(), need_out * sizeof(float));
        return RSMP_OK;
    }
    RSMP_HIP_CHECK(r->d_stage_in.reserve(need_in * sizeof(float)));
    RSMP_HIP_CHECK(r->d_stage_out.reserve(need_out * sizeof(float)));
    RSMP_HIP_CHECK(hipMemcpyAsync(r->d_stage_in.get(), in, need_in * sizeof(float),
                                  hipMemcpyHostToDevice, r->stream));
    const int rc = rsmp_fft_resample_bulk_device(r, r->d_stage_in.as<float>(), need_in,
                                                 r->d_stage_out.as<float>(), need_out, n_chunks,
                                                 r->stream);
    if (rc != RSMP_OK) return rc;
    RSMP_HIP_CHECK(hipMemcpyAsync(out, r->d_stage_out.get(), need_out * sizeof(float),
                                  hipMemcpyDeviceToHost, r->stream));
    RSMP_HIP_CHECK(hipStreamSynchronize(r->stream));
    return RSMP_OK;
}

extern "C" int rsmp_fft_resample(rsmp_fft* r, const float* in, size_t in_len, float* out,
                                 size_t out_len) {
    return rsmp_fft_resample_bulk(r, in, in_len, out, out_len, 1);
}

extern "C" int rsmp_fft_batch_resample_bulk_device(rsmp_fft* const* rs, size_t n,
                                                   const float* const* d_in, float* const* d_out,
                                                   const size_t* n_chunks, void* stream) {
    if (n == 0) return RSMP_OK;
    if (!rs || !d_in || !d_out || !n_chunks)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fft_batch_resample_bulk_device: null argument");
    for (size_t i = 0; i < n; ++i)
        if (!rs[i] || rs[i]->device != rs[0]->device || rs[i]->plan != rs[0]->plan)
            return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT,
                              "batch streams must share one device and one rate pair");
    DeviceGuard guard(rs[0]->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : rs[0]->stream;
    std::vector<FftJob> jobs;
    jobs.reserve(n);
    for (size_t i = 0; i < n; ++i) jobs.push_back(FftJob{rs[i], d_in[i], d_out[i], n_chunks[i]});
    return launch_fft_jobs(rs[0], jobs, s);
}

// The same with the WAV file's samples as they are: little-endian PCM of `bits` (16 / 24 / 32) per sample, two channels a
// frame, converted inside the kernel's first load as resample/src/main.rs:128-137 converts them (`sample as f32 /
// (1 << (bits - 1)) as f32`, the 32-bit divisor's sign included) -- rsmp_pcm_to_stereo_f32_device + the f32 entry point
// give the same samples bit for bit, with one more pass over HBM (PCM read, f32 written, f32 read).
extern "C" int rsmp_fft_batch_resample_bulk_pcm_device(rsmp_fft* const* rs, size_t n, const void* const* d_pcm, int bits,
                                                       float* const* d_out, const size_t* n_chunks, void* stream) {
    if (n == 0) return RSMP_OK;
    if (!rs || !d_pcm || !d_out || !n_chunks || (bits != 16 && bits != 24 && bits != 32))
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fft_batch_resample_bulk_pcm_device: null argument, or bits not 16 / 24 / 32");
    for (size_t i = 0; i < n; ++i) {
        if (!rs[i] || rs[i]->device != rs[0]->device || rs[i]->plan != rs[0]->plan || rs[i]->channels != 2)
            return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "PCM batch streams must share one device and one rate pair and have two channels");
        if (reinterpret_cast<uintptr_t>(d_pcm[i]) % (bits == 32 ? 16 : bits == 16 ? 8 : 4) != 0)
            return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "PCM input must be aligned to 8 (16-bit), 4 (24-bit) or 16 (32-bit) bytes");
    }
    DeviceGuard guard(rs[0]->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : rs[0]->stream;
    std::vector<FftJob> jobs;
    jobs.reserve(n);
    for (size_t i = 0; i < n; ++i) jobs.push_back(FftJob{rs[i], static_cast<const float*>(d_pcm[i]), d_out[i], n_chunks[i]});
    return launch_fft_jobs(rs[0], jobs, s, static_cast<uint32_t>(bits));
}
