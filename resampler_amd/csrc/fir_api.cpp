// fir_api.cpp -- ResamplerFir front-end on the GPU: handles, state, launch assembly, C ABI.
//
// Mirrors src/resampler_fir.rs of the reference: construction (:295-404), buffer_size_output
// (:456-465), resample (:509-621), delay (:630-632), reset (:638-642).  The per-call control
// flow (how many frames are accepted / produced / retired) runs on the host in FirMirror; the
// arithmetic runs in one launch of fir_generic / fir_periodic per call, bulk buffer or batch.
//
// Device-side stream state: instead of the reference's planar double-size ring
// (input_buffers, :187, :329) each stream keeps only the frames still buffered
// (available_frames <= 4096) as an interleaved `hist` array, ping-ponged between two HBM
// buffers; a launch reads the virtual concatenation [hist | new input] and a small tail-copy
// kernel writes the next hist.  There is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <algorithm>
#include <vector>

#include "common.h"
#include "device_util.h"
#include "filter_design.h"
#include "fir_handle.h"
#include "fir_kernels.h"
#include "fir_periodic.h"
#include "fir_plan.h"

using rsmp::DeviceBuffer;
using rsmp::DeviceGuard;
using rsmp::FirMirror;
using rsmp::FirStreamDesc;
using rsmp::PinnedBuffer;

namespace {

// Per-device cache of uploaded polyphase tables (the device-side half of the reference's
// FIR_CACHE, resampler_fir.rs:164-166): key = (device, host table identity).
struct DeviceTableCache {
    std::mutex mu;
    std::map<std::pair<int, const void*>, float*> tables;
    // Keeps host tables alive for as long as their device copies are cached.
    std::vector<std::shared_ptr<const std::vector<float>>> pins;
};
DeviceTableCache& table_cache() {
    static DeviceTableCache* c = new DeviceTableCache;  // leaked on purpose (process lifetime)
    return *c;
}

int upload_table(int device, const std::shared_ptr<const std::vector<float>>& host, float** out) {
    DeviceTableCache& c = table_cache();
    std::lock_guard<std::mutex> lock(c.mu);
    const auto key = std::make_pair(device, static_cast<const void*>(host.get()));
    auto it = c.tables.find(key);
    if (it != c.tables.end()) { *out = it->second; return RSMP_OK; }
    float* d = nullptr;
    RSMP_HIP_CHECK(hipMalloc(&d, host->size() * sizeof(float)));
    RSMP_HIP_CHECK(hipMemcpy(d, host->data(), host->size() * sizeof(float), hipMemcpyHostToDevice));
    c.tables.emplace(key, d);
    c.pins.push_back(host);
    *out = d;
    return RSMP_OK;
}

}  // namespace

namespace {

// ---- batches in many different states: planned on the device (see rsmp_fir_batch_resample_bulk_device_ex) ---------------------------
// The lock-step batch a routed launch runs on is kept per list of handles, from launch to launch: its plan stream, its class tables and
// the run it plans ahead are what make the second and later launches cheap.  Its states are written back into the handles before a
// routed call returns, so the handles are always current and the batch can be thrown away at any time WITHOUT a write-back
// (rsmp_fir_lockstep_discard) -- which is what happens when a handle has been touched through another entry since, when the cache is
// full, and when one of its handles is destroyed.
struct RoutedBatch {
    rsmp_fir_lockstep* ls = nullptr;
    size_t frames = 0;                     // max_step_frames it was made for
    std::vector<const void*> bound;        // d_in / d_out it is bound to
    uint64_t used = 0;
};
std::mutex& routed_mu() { static std::mutex* m = new std::mutex; return *m; }
std::map<std::vector<rsmp_fir*>, RoutedBatch>& routed_cache() { static auto* c = new std::map<std::vector<rsmp_fir*>, RoutedBatch>; return *c; }
uint64_t routed_clock = 0;
constexpr size_t kRoutedCacheSize = 8;
constexpr size_t kRoutedMinStates = 16;    // fewer different states than this (and than streams): the host's shared plans are cheaper
constexpr size_t kRoutedMaxCallFrames = 2048, kRoutedMinCalls = 8;

void routed_forget(const rsmp_fir* r) {
    std::lock_guard<std::mutex> lock(routed_mu());
    auto& cache = routed_cache();
    for (auto it = cache.begin(); it != cache.end();) {
        if (std::find(it->first.begin(), it->first.end(), r) != it->first.end()) {
            rsmp_fir_lockstep_discard(it->second.ls);
            it = cache.erase(it);
        } else {
            ++it;
        }
    }
}

// Releases everything a (possibly half-built) handle owns.
void fir_destroy(rsmp_fir* r) {
    if (!r) return;
    routed_forget(r);
    DeviceGuard guard(r->device);
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 2; ++i) if (r->d_hist[i]) (void)hipFree(r->d_hist[i]);
    if (r->d_work_counter) (void)hipFree(r->d_work_counter);
    for (hipEvent_t e : r->plan_copied) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : r->prof_start) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : r->prof_stop) if (e) (void)hipEventDestroy(e);
    if (r->stream) {
        rsmp::split_release_stream(r->device, r->stream);
        (void)hipStreamDestroy(r->stream);
    }
    delete r;
}

rsmp_fir* fir_create(size_t channels, uint32_t in_hz, uint32_t out_hz, int latency,
                     int attenuation, int device) {
    const size_t taps = rsmp::latency_taps(latency);
    const double beta = rsmp::attenuation_beta(attenuation);
    if (channels == 0 || channels > 4096 || taps == 0 || beta < 0.0) {
        rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "ResamplerFir: invalid channels/latency/attenuation");
        return nullptr;
    }
    if (in_hz == 0) {  // resampler_fir.rs:302-305 panics
        rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "input sample rate must be greater than zero");
        return nullptr;
    }
    if (out_hz == 0) {  // resampler_fir.rs:306-309 panics
        rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "output sample rate must be greater than zero");
        return nullptr;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        rsmp::fail(RSMP_ERR_NO_DEVICE, "ResamplerFir: no HIP device (this engine has no CPU path)");
        return nullptr;
    }
    if (device < 0 || device >= n_dev) {
        rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "ResamplerFir: device %d out of range (%d devices)",
                   device, n_dev);
        return nullptr;
    }
    DeviceGuard guard(device);
    std::unique_ptr<rsmp_fir> r(new rsmp_fir(in_hz, out_hz, taps));
    r->device = device;
    r->channels = channels;
    r->taps = taps;
    r->attenuation = attenuation;
    r->in_hz = in_hz;
    r->out_hz = out_hz;
    const rsmp::FirDesign design = rsmp::fir_design(in_hz, out_hz, taps, beta);
    r->table = rsmp::get_or_create_fir_coeffs(design.cutoff, taps, attenuation);
    if (upload_table(device, r->table, &r->d_coeffs) != RSMP_OK) return nullptr;
    const size_t hist_bytes = rsmp::kInputCapacity * channels * sizeof(float);
    for (int i = 0; i < 2; ++i) {
        if (hipMalloc(&r->d_hist[i], hist_bytes) != hipSuccess ||
            hipMemset(r->d_hist[i], 0, hist_bytes) != hipSuccess ||
            hipStreamSynchronize(nullptr) != hipSuccess) {   // (the handle's stream is non-blocking: no implicit order)
            rsmp::fail(RSMP_ERR_HIP, "ResamplerFir: cannot allocate stream state");
            fir_destroy(r.release());
            return nullptr;
        }
    }
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&r->plan_copied[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->plan_copied[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->plan_copied[2], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&r->plan_copied[3], hipEventDisableTiming) != hipSuccess) {
        rsmp::fail(RSMP_ERR_HIP, "ResamplerFir: cannot create stream/event");
        fir_destroy(r.release());
        return nullptr;
    }
    return r.release();
}

// The host-side result of replaying a reference call sequence: shared between the streams of a
// batch that are in the same state and are fed the same amount of input (their control flow is
// data independent, so one replay serves all of them).
struct Plan {
    FirMirror planned;                      // mirror state after the launch
    std::vector<rsmp_fir_segment> segs;     // generic kernel: exact position runs
    std::vector<uint32_t> wraps;            // periodic kernel: row-1023 fix-ups
    std::vector<uint32_t> wrap_bits;        // ... as the bitmap the kernels with inline wraps read (built with the plan, by its worker)
    std::vector<size_t> calls;              // (consumed, produced) per reference call, in values
    size_t accepted_frames = 0;
    size_t produced_frames = 0;
    size_t consumed_frames = 0;
    size_t hist_frames = 0;
    bool periodic = false;
    explicit Plan(const FirMirror& m) : planned(m) {}
};

struct PlanKey {
    uint32_t in_hz, out_hz;
    size_t taps, channels, read_position, available;
    uint64_t position_bits, abs_out, abs_consumed;
    size_t in_len, out_cap_or_zero, chunk_len;
    int kernel_mode;
    bool operator==(const PlanKey& o) const { return std::memcmp(this, &o, sizeof o) == 0; }
};

// One stream's part of a launch.
struct Job {
    rsmp_fir* r;
    const float* d_in;
    size_t in_len;     // f32 values offered
    float* d_out;
    size_t out_cap;    // f32 values of room
    size_t chunk_len;  // 0: one reference call with output capacity out_cap; else bulk loop
    std::shared_ptr<Plan> plan;
    size_t consumed() const { return plan->accepted_frames * r->channels; }
    size_t produced() const { return plan->produced_frames * r->channels; }
};

PlanKey make_key(const Job& j) {
    PlanKey k;
    std::memset(&k, 0, sizeof k);
    const rsmp_fir* r = j.r;
    k.in_hz = r->in_hz;
    k.out_hz = r->out_hz;
    k.taps = r->taps;
    k.channels = r->channels;
    k.read_position = r->mirror.read_position();
    k.available = r->mirror.available();
    const double pos = r->mirror.position();
    std::memcpy(&k.position_bits, &pos, sizeof pos);
    k.abs_out = r->mirror.abs_out();
    k.abs_consumed = r->mirror.abs_consumed();
    k.in_len = j.in_len;
    k.out_cap_or_zero = j.chunk_len == 0 ? j.out_cap : 0;
    k.chunk_len = j.chunk_len;
    k.kernel_mode = r->kernel_mode;
    return k;
}

// Process-wide plan cache.  A plan is a pure function of (configuration, stream state, amount of
// input, chunking) -- no sample values -- so, like an FFT plan, it is built once and reused: a
// service converting many files replays the same few plans over and over (every fresh stream
// of a given length starts in the same state).  Bounded; oldest entry evicted first.
struct PlanCache {
    std::mutex mu;
    std::vector<std::pair<PlanKey, std::shared_ptr<Plan>>> entries;
    size_t next_evict = 0;
    static constexpr size_t kMaxEntries = 64;

    std::shared_ptr<Plan> find(const PlanKey& k) {
        std::lock_guard<std::mutex> lock(mu);
        for (auto& e : entries) if (e.first == k) return e.second;
        return nullptr;
    }
    void insert(const PlanKey& k, const std::shared_ptr<Plan>& p) {
        std::lock_guard<std::mutex> lock(mu);
        if (entries.size() < kMaxEntries) { entries.emplace_back(k, p); return; }
        entries[next_evict] = std::make_pair(k, p);
        next_evict = (next_evict + 1) % kMaxEntries;
    }
};
PlanCache& plan_cache() {
    static PlanCache* c = new PlanCache;
    return *c;
}

// Replays the reference call sequence on a copy of the mirror (committed only on success).
int plan_job_uncached(Job& j, bool with_segments) {
    rsmp_fir* r = j.r;
    const size_t ch = r->channels;
    if (j.in_len % ch != 0)
        return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE, "Input buffer size is invalid");
    auto plan = std::make_shared<Plan>(r->mirror);
    Plan& pl = *plan;
    pl.hist_frames = pl.planned.available();
    const bool want_periodic =
        !with_segments && rsmp::periodic_supported(pl.planned, ch, r->taps, r->kernel_mode);
    std::vector<uint32_t>* wraps = want_periodic ? &pl.wraps : nullptr;
    std::vector<rsmp_fir_segment>* segs = want_periodic ? nullptr : &pl.segs;
    const size_t in_frames_total = j.in_len / ch;
    if (j.chunk_len == 0) {
        if (j.out_cap % ch != 0)
            return rsmp::fail(RSMP_ERR_INVALID_OUTPUT_BUFFER_SIZE, "Output buffer size is invalid");
        const rsmp::FirCallResult c = pl.planned.call(in_frames_total, j.out_cap / ch, 0, 0, segs, wraps);
        pl.accepted_frames = c.accepted;
        pl.produced_frames = c.produced;
        pl.consumed_frames = c.consumed;
        pl.calls.push_back(c.accepted * ch);
        pl.calls.push_back(c.produced * ch);
    } else {
        // resample/src/main.rs:226-254 with CHUNK_SIZE = chunk_len values
        if (j.chunk_len % ch != 0)
            return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE,
                              "Input buffer size is invalid (chunk_len not a multiple of channels)");
        const size_t chunk_frames = j.chunk_len / ch;
        const size_t cap_frames = pl.planned.buffer_size_output_frames();
        size_t offset = 0;
        while (offset < in_frames_total) {
            const size_t remaining = in_frames_total - offset;
            const size_t take = remaining < chunk_frames ? remaining : chunk_frames;
            if (pl.produced_frames > 0x7FF00000ull - cap_frames)
                return rsmp::fail(RSMP_ERR_CAPACITY, "bulk launch exceeds 2^31 output frames");
            const rsmp::FirCallResult c =
                pl.planned.call(take, cap_frames, static_cast<int64_t>(pl.consumed_frames),
                                static_cast<uint32_t>(pl.produced_frames), segs, wraps);
            pl.calls.push_back(c.accepted * ch);
            pl.calls.push_back(c.produced * ch);
            pl.produced_frames += c.produced;
            pl.consumed_frames += c.consumed;
            pl.accepted_frames += c.accepted;
            offset += c.accepted;
            if (c.accepted == 0) break;
        }
        if (pl.produced_frames * ch > j.out_cap)
            return rsmp::fail(RSMP_ERR_CAPACITY, "bulk output needs %zu values, room for %zu",
                              pl.produced_frames * ch, j.out_cap);
    }
    if (want_periodic) {
        if (!pl.planned.periodic_ok() ||
            !rsmp::periodic_worthwhile(pl.planned, pl.produced_frames, r->kernel_mode))
            return plan_job_uncached(j, true);  // replay once more, keeping the position runs
        pl.periodic = true;
        // (one 64-bit division per wrapped output: on the planning worker, not on the thread that builds the launch)
        pl.wrap_bits.resize(rsmp::periodic_wrap_words(r->mirror.abs_out(), static_cast<uint32_t>(pl.produced_frames), r->mirror.den()));
        rsmp::periodic_fill_wrap_bits(pl.wraps, r->mirror.abs_out(), r->mirror.den(), pl.wrap_bits.data(), pl.wrap_bits.size());
    }
    j.plan = plan;
    return RSMP_OK;
}

int plan_job(Job& j) {
    if (j.chunk_len == 0) return plan_job_uncached(j, false);  // one call: cheaper than a lookup
    const PlanKey key = make_key(j);
    if (auto hit = plan_cache().find(key)) {
        if (hit->produced_frames * j.r->channels > j.out_cap)
            return rsmp::fail(RSMP_ERR_CAPACITY, "bulk output needs %zu values, room for %zu",
                              hit->produced_frames * j.r->channels, j.out_cap);
        j.plan = hit;
        return RSMP_OK;
    }
    const int rc = plan_job_uncached(j, false);
    if (rc == RSMP_OK) plan_cache().insert(key, j.plan);
    return rc;
}

// Process-wide workers for planning batches of streams in distinct states (rsmp_fir_batch_resample_bulk_device): the
// items of a run are claimed from an atomic counter by the pool's threads and by the caller; one run at a time.
// Everything a run shares with the workers lives in ONE object per run (its function, its item count, its claim
// counter, its completion count), handed over under the mutex: a worker that wakes up late holds either the finished
// run (nothing left to claim) or the current one -- never one run's counter with another run's bounds.
class PlanPool {
public:
    PlanPool() { ensure_threads(64); }   // (at library load: creating up to 63 threads inside the first timed launch cost it milliseconds)
    ~PlanPool() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (std::thread& t : threads_) t.join();
    }
    template <class F>
    void run(size_t total, F&& fn) {
        std::lock_guard<std::mutex> one_run(run_mu_);
        auto job = std::make_shared<Job>();
        job->fn = fn;
        job->total = total;
        {
            std::lock_guard<std::mutex> lock(mu_);
            job_ = job;
            ++generation_;
        }
        cv_work_.notify_all();
        drain(*job);
        std::unique_lock<std::mutex> lock(mu_);
        cv_done_.wait(lock, [&] { return job->done == job->total; });
        job_.reset();
    }

private:
    struct Job {
        std::function<void(size_t)> fn;
        size_t total = 0;
        std::atomic<size_t> next{0};
        size_t done = 0;   // (under mu_)
    };
    void ensure_threads(size_t total) {
        unsigned hw = std::thread::hardware_concurrency();
        const size_t want = std::min<size_t>(total - 1, std::min<size_t>(hw > 1 ? hw - 1 : 0, 63));
        while (threads_.size() < want) threads_.emplace_back([this] { loop(); });
    }
    void drain(Job& job) {
        for (;;) {
            const size_t m = job.next.fetch_add(1, std::memory_order_relaxed);
            if (m >= job.total) break;
            job.fn(m);
            std::lock_guard<std::mutex> lock(mu_);
            if (++job.done == job.total) cv_done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_work_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                job = job_;   // (null: the run this wake-up was for has finished)
            }
            if (job) drain(*job);
        }
    }
    std::mutex mu_, run_mu_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> threads_;
    std::shared_ptr<Job> job_;
    uint64_t generation_ = 0;
    bool stop_ = false;
};
PlanPool& plan_pool() {
    static PlanPool* pool = new PlanPool();   // (leaked on purpose: joining workers from a static destructor at exit races the runtime's teardown)
    return *pool;
}
// (the workers must not be created inside the first launch that needs them; ADVICE r04: nor from a
// static initialiser at dlopen, for every user of the library.  They are created by the first ResamplerFir constructor
// of the process -- rsmp_fir_new_from_hz below -- and idle on a condition variable from then on: at most 63 threads,
// hardware_concurrency() - 1 if that is less.)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Assembles and enqueues the launches for a set of planned jobs on one device / stream.
// `leader` owns the launch workspace.
// pcm_bits != 0: every job's d_in is a WAV file's PCM of that width, read in place (FirStreamDesc::in_bits): two-channel
// streams on the generic kernel (short launches) or the split kernel's PCM builds.
int launch_jobs(rsmp_fir* leader, std::vector<Job>& jobs, hipStream_t stream, uint32_t pcm_bits = 0) {
    // Plans are shared, immutable once built; where a plan's arrays sit in THIS launch's workspace is
    // launch-local (streams of a batch that share a plan share its arrays too).
    struct Placement { size_t seg_off = 0, tile_off = 0, wrap_off = 0; bool written = false; };
    std::map<const Plan*, Placement> place;
    const size_t n = jobs.size();
    // Launches that touch a handle (its buffered frames, its plan slots, the leader's item queue) are
    // ordered: the ABI lets every call name a stream, so a handle that was last used on another stream
    // waits for that stream first (rare; a caller that keeps one stream per handle never blocks here).
    auto order_after = [&](rsmp_fir* h) -> int {
        if (h->last_stream_valid && h->last_stream != stream) RSMP_HIP_CHECK(hipStreamSynchronize(h->last_stream));
        h->last_stream = stream;
        h->last_stream_valid = true;
        return RSMP_OK;
    };
    if (int rc = order_after(leader)) return rc;
    for (Job& j : jobs)
        if (int rc = order_after(j.r)) return rc;
    // Bind class tables first (may upload), then order: generic jobs, then periodic jobs grouped
    // by geometry (one launch per geometry).
    std::vector<size_t> order;
    size_t n_generic = 0;
    for (size_t i = 0; i < n; ++i)
        if (!jobs[i].plan->periodic) { order.push_back(i); ++n_generic; }
    struct Group { rsmp::PeriodicGeometry geo; std::vector<size_t> members; };
    std::vector<Group> groups;
    for (size_t i = 0; i < n; ++i) {
        Job& j = jobs[i];
        if (!j.plan->periodic) continue;
        const int rc = rsmp::periodic_bind(j.r->periodic, j.r->device, *j.r->table, j.r->kernel_mode, j.plan->planned,
                                           0.5 * (j.r->mirror.drift() + j.plan->planned.drift()),
                                           static_cast<uint32_t>(j.r->channels), stream);
        if (rc != RSMP_OK) return rc;
        bool found = false;
        for (Group& g : groups)
            if (g.geo == j.r->periodic.geo) { g.members.push_back(i); found = true; break; }
        if (!found) groups.push_back(Group{j.r->periodic.geo, {i}});
    }
    for (const Group& g : groups) for (size_t i : g.members) order.push_back(i);
    if (pcm_bits != 0) {
        for (const Job& j : jobs)
            if (j.r->channels != 2) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "PCM input: two-channel streams only");
        for (const Group& g : groups) {
            const uint32_t nk = g.geo.row_len / 32;
            const bool ok = g.geo.mfma == 3 && g.geo.planes == 2 && g.geo.lp == 1 && g.geo.cg == 2 &&
                            ((g.geo.rounds == 1 && nk == 5) || (g.geo.rounds == 2 && (nk == 5 || nk == 6)));
            if (!ok)
                return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT,
                                  "PCM input is read in place by the two-channel split kernel of the 128-tap rate pairs only "
                                  "(44.1 <-> 48, 96 -> 44.1 / 48 kHz ...): convert with rsmp_pcm_to_stereo_f32_device first");
        }
    }

    // Workspace layout: [descs] then per distinct plan: [runs][tile index] or [wraps].
    size_t bytes = align_up(n * sizeof(FirStreamDesc), 256);
    for (Job& j : jobs) {
        const Plan& pl = *j.plan;
        if (place.count(&pl)) continue;
        Placement& pp = place[&pl];
        if (!pl.periodic) {
            pp.seg_off = bytes;
            bytes = align_up(bytes + pl.segs.size() * sizeof(rsmp_fir_segment), 256);
            pp.tile_off = bytes;
            const size_t tiles = (pl.produced_frames + rsmp::kFirTile - 1) / rsmp::kFirTile;
            bytes = align_up(bytes + tiles * sizeof(uint32_t), 256);
        } else {
            pp.wrap_off = bytes;
            const rsmp::PeriodicGeometry& geo = j.r->periodic.geo;
            const size_t words = geo.inline_wraps
                                     ? rsmp::periodic_wrap_words(j.r->mirror.abs_out(),
                                                                 static_cast<uint32_t>(pl.produced_frames),
                                                                 geo.den)
                                     : pl.wraps.size();
            bytes = align_up(bytes + words * sizeof(uint32_t), 256);
        }
    }
    const int slot = leader->plan_slot;
    leader->plan_slot = (slot + 1) % rsmp_fir::kPlanSlots;
    // Small plans (a single call, a handful of streams) are not uploaded at all: the kernels read
    // them from mapped, coherent host memory.  That removes a copy-engine operation and its
    // cross-queue synchronisation (~50 us) from every streaming call.
    const bool direct = bytes <= 16 * 1024;
    if (direct) {
        if (leader->plan_pending[slot]) {  // kernels of the slot's previous launch are done with it
            RSMP_HIP_CHECK(hipEventSynchronize(leader->plan_copied[slot]));
            leader->plan_pending[slot] = false;
        }
        RSMP_HIP_CHECK(leader->h_plan[slot].reserve(16 * 1024));
    } else if (bytes > leader->d_plan[slot].capacity()) {
        RSMP_HIP_CHECK(hipStreamSynchronize(stream));
        RSMP_HIP_CHECK(leader->d_plan[slot].reserve(bytes));
        leader->plan_image[slot].clear();
    }
    // The image is assembled in ordinary host memory first: a launch that repeats an earlier one
    // of this slot (same streams, buffers and state -- e.g. a service resampling batch after batch
    // of equally long files) finds its image already in HBM and skips the upload.
    leader->plan_scratch.assign(bytes, 0);
    char* h = leader->plan_scratch.data();
    char* d = direct ? leader->h_plan[slot].as<char>() : leader->d_plan[slot].as<char>();
    FirStreamDesc* descs = reinterpret_cast<FirStreamDesc*>(h);

    uint32_t max_out_generic = 0, max_ch_generic = 0, max_tail_values = 0, max_wraps = 0, max_taps_generic = 0, min_ch_generic = 0xFFFFFFFFu, min_taps_generic = 0xFFFFFFFFu;
    double max_ratio_generic = 0.0;
    for (size_t slot = 0; slot < n; ++slot) {
        Job& j = jobs[order[slot]];
        const Plan& pl = *j.plan;
        Placement& pp = place[&pl];
        rsmp_fir* r = j.r;
        const uint32_t ch = static_cast<uint32_t>(r->channels);
        FirStreamDesc& ds = descs[slot];
        std::memset(&ds, 0, sizeof ds);
        ds.in = j.d_in;
        ds.hist = r->d_hist[r->cur];
        ds.hist_next = r->d_hist[r->cur ^ 1];
        ds.out = j.d_out;
        ds.coeffs = r->d_coeffs;
        ds.n_out = static_cast<uint32_t>(pl.produced_frames);
        ds.hist_frames = static_cast<uint32_t>(pl.hist_frames);
        ds.in_frames = static_cast<uint32_t>(pl.accepted_frames);
        ds.tail_start = static_cast<uint32_t>(pl.consumed_frames);
        ds.tail_frames = static_cast<uint32_t>(pl.planned.available());
        ds.channels = ch;
        ds.taps = static_cast<uint32_t>(r->taps);
        ds.num = static_cast<uint32_t>(r->mirror.num());
        ds.den = static_cast<uint32_t>(r->mirror.den());
        ds.abs_out = r->mirror.abs_out();
        ds.abs_consumed = r->mirror.abs_consumed();
        ds.in_bits = pcm_bits;
        if (ds.tail_frames * ch > max_tail_values) max_tail_values = ds.tail_frames * ch;
        if (!pl.periodic) {
            ds.segs = reinterpret_cast<const rsmp_fir_segment*>(d + pp.seg_off);
            ds.n_segs = static_cast<uint32_t>(pl.segs.size());
            ds.tile_seg = reinterpret_cast<const uint32_t*>(d + pp.tile_off);
            if (!pp.written) {
                std::memcpy(h + pp.seg_off, pl.segs.data(), pl.segs.size() * sizeof(rsmp_fir_segment));
                uint32_t* ts = reinterpret_cast<uint32_t*>(h + pp.tile_off);
                size_t s = 0;
                for (size_t t = 0; t * rsmp::kFirTile < pl.produced_frames; ++t) {
                    const size_t first = t * rsmp::kFirTile;
                    while (first >= static_cast<size_t>(pl.segs[s].out_start) + pl.segs[s].count) ++s;
                    ts[t] = static_cast<uint32_t>(s);
                }
            }
            if (ds.n_out > max_out_generic) max_out_generic = ds.n_out;
            if (ch > max_ch_generic) max_ch_generic = ch;
            if (ch < min_ch_generic) min_ch_generic = ch;
            if (ds.taps > max_taps_generic) max_taps_generic = ds.taps;
            if (ds.taps < min_taps_generic) min_taps_generic = ds.taps;
            max_ratio_generic = std::max(max_ratio_generic, static_cast<double>(r->in_hz) / static_cast<double>(r->out_hz));
        } else {
            const rsmp::PeriodicGeometry& geo = r->periodic.geo;
            ds.drift = r->periodic.table_drift;
            ds.class_coef = r->periodic.table.d_coef;
            ds.class_wrap_coef = r->periodic.table.d_wrap_coef;
            ds.class_meta = r->periodic.table.d_meta;
            if (geo.inline_wraps) {
                ds.wrap_bits = reinterpret_cast<const uint32_t*>(d + pp.wrap_off);
                // (the split kernel counts periods of b outputs; b = den unless its super period spans several true
                // periods -- exact ratios only, whose streams have no wrapped outputs: an all-zero bitmap, any indexing)
                ds.wrap_k0 = r->mirror.abs_out() / (geo.mfma == 3 ? geo.b : geo.den);
                if (geo.mfma == 3 && geo.b != geo.den && !pl.wraps.empty())
                    return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "split kernel: a stream of an exact ratio has wrapped outputs");
                if (!pp.written) {
                    const size_t words = rsmp::periodic_wrap_words(r->mirror.abs_out(), ds.n_out, geo.den);
                    if (pl.wrap_bits.size() == words && geo.den == r->mirror.den())
                        std::memcpy(h + pp.wrap_off, pl.wrap_bits.data(), words * sizeof(uint32_t));
                    else
                        rsmp::periodic_fill_wrap_bits(pl.wraps, r->mirror.abs_out(), geo.den,
                                                      reinterpret_cast<uint32_t*>(h + pp.wrap_off), words);
                }
            } else {
                ds.wraps = reinterpret_cast<const uint32_t*>(d + pp.wrap_off);
                ds.n_wraps = static_cast<uint32_t>(pl.wraps.size());
                if (!pp.written)
                    std::memcpy(h + pp.wrap_off, pl.wraps.data(), pl.wraps.size() * sizeof(uint32_t));
                if (ds.n_wraps > max_wraps) max_wraps = ds.n_wraps;
            }
        }
        pp.written = true;
    }
    if (direct) {
        std::memcpy(d, h, bytes);
    } else if (leader->plan_image[slot] != leader->plan_scratch) {
        if (leader->plan_pending[slot]) {  // this slot's previous upload must have left pinned memory
            RSMP_HIP_CHECK(hipEventSynchronize(leader->plan_copied[slot]));
            leader->plan_pending[slot] = false;
        }
        RSMP_HIP_CHECK(leader->h_plan[slot].reserve(bytes));
        std::memcpy(leader->h_plan[slot].get(), h, bytes);
        RSMP_HIP_CHECK(hipMemcpyAsync(d, leader->h_plan[slot].get(), bytes, hipMemcpyHostToDevice, stream));
        RSMP_HIP_CHECK(rsmp::event_record(leader->plan_copied[slot], stream));
        leader->plan_pending[slot] = true;
        leader->plan_image[slot].swap(leader->plan_scratch);
    }

    const FirStreamDesc* d_descs = reinterpret_cast<const FirStreamDesc*>(d);
    if (leader->profiling)
        RSMP_HIP_CHECK(rsmp::event_record(leader->prof_start[leader->prof_count % rsmp_fir::kProfRing], stream));
    // a launch made of generic-kernel streams only (a streaming call, a batch of them) lets that kernel copy
    // the tails as well: one launch per call instead of two
    bool tail_fused = n_generic == n && max_out_generic != 0;
    // Long launches of streams without a short period (arbitrary rates, src/resampler_fir.rs:295-301) take the tiled kernel,
    // whose workgroups sort a tile's outputs by phase row and stage its window in LDS (fir_generic_bulk.hip); streaming calls
    // keep the one-launch latency path.  (RSMP_FIR_GENERIC_BULK=0, debug: the latency kernel for everything.)
    static const bool bulk_on = [] { const char* e = rsmp::knob("RSMP_FIR_GENERIC_BULK"); return !e || atoi(e) != 0; }();
    const bool generic_bulk = bulk_on && n_generic != 0 && max_out_generic >= rsmp::kFirBulkMinOut &&
                              rsmp::fir_generic_bulk_tile(max_ch_generic, max_taps_generic, max_ratio_generic) != 0;
    if (generic_bulk) {
        tail_fused = false;
        RSMP_HIP_CHECK(rsmp::launch_fir_generic_bulk(d_descs, static_cast<uint32_t>(n_generic), max_out_generic, max_ch_generic,
                                                     max_taps_generic, max_ratio_generic, stream,
                                                     min_ch_generic == max_ch_generic ? max_ch_generic : 0u,
                                                     min_taps_generic == max_taps_generic ? max_taps_generic : 0u));
    } else if (n_generic)
        RSMP_HIP_CHECK(rsmp::launch_fir_generic(d_descs, static_cast<uint32_t>(n_generic),
                                                max_out_generic, max_ch_generic, stream, tail_fused));
    size_t first = n_generic;
    // Where the periodic launches mark non-finite sums: one bit per stream and 1024-frame chunk
    // (fir_nonfinite.h), one region of the buffer per launch; the repair launches follow the timed ones.
    struct Repair { size_t first; uint32_t count; rsmp::NfArgs nf; };
    std::vector<Repair> repairs;
    size_t nf_words_total = 0;
    for (const Group& g : groups) {
        uint32_t max_out = 0;
        for (size_t i : g.members)
            if (jobs[i].plan->produced_frames > max_out) max_out = static_cast<uint32_t>(jobs[i].plan->produced_frames);
        Repair rp;
        rp.first = 0;
        rp.count = static_cast<uint32_t>(g.members.size());
        rp.nf.chunks = (max_out >> rsmp::kNfChunkShift) + 1;
        rp.nf.words = reinterpret_cast<uint32_t*>(nf_words_total * sizeof(uint32_t));   // offset for now
        rp.nf.tag = 0;
        nf_words_total += 1 + (static_cast<size_t>(rp.count) * rp.nf.chunks + 31) / 32;
        repairs.push_back(rp);
    }
    if (nf_words_total * sizeof(uint32_t) > leader->d_nf.capacity()) {
        RSMP_HIP_CHECK(hipStreamSynchronize(stream));
        RSMP_HIP_CHECK(leader->d_nf.reserve(nf_words_total * sizeof(uint32_t)));
        RSMP_HIP_CHECK(hipMemsetAsync(leader->d_nf.get(), 0, leader->d_nf.capacity(), stream));
    }
    size_t gi = 0;
    std::vector<rsmp::SplitJob> split_jobs;
    for (const Group& g : groups) {
        uint32_t max_blocks = 0;
        for (size_t i : g.members) {
            const uint32_t b = rsmp::periodic_blocks(g.geo, jobs[i].r->mirror.abs_out(),
                                                     static_cast<uint32_t>(jobs[i].plan->produced_frames));
            if (b > max_blocks) max_blocks = b;
        }
        if (!leader->d_work_counter) {
            RSMP_HIP_CHECK(hipMalloc(&leader->d_work_counter, sizeof(unsigned long long)));
            // on the launch stream: a null-stream memset is not ordered with a non-blocking stream and
            // could land after the first kernel had started claiming
            RSMP_HIP_CHECK(hipMemsetAsync(leader->d_work_counter, 0, sizeof(unsigned long long), stream));
        }
        // a launch made of split-kernel streams only lets that kernel copy the tails as well
        tail_fused = n_generic == 0 && groups.size() == 1 && g.geo.mfma == 3 && max_blocks != 0;
        Repair& rp = repairs[gi++];
        rp.first = first;
        rp.nf.words = leader->d_nf.as<uint32_t>() + reinterpret_cast<size_t>(rp.nf.words) / sizeof(uint32_t);
        if (++leader->nf_tag == 0) leader->nf_tag = 1;
        rp.nf.tag = leader->nf_tag;
        // what the split kernel's item table is a function of (fir_split.hip, split_items_kernel): FNV-1a over it
        uint64_t key = 1469598103934665603ull;
        auto mix = [&](uint64_t v) { for (int b = 0; b < 8; ++b) { key ^= (v >> (8 * b)) & 0xFFu; key *= 1099511628211ull; } };
        mix(g.geo.a); mix(g.geo.b); mix(g.geo.lp); mix(g.geo.groups); mix(max_blocks); mix(g.members.size());
        for (size_t i : g.members) {
            const Job& j = jobs[i];
            mix(j.r->mirror.abs_out()); mix(j.r->mirror.abs_consumed()); mix(j.plan->produced_frames);
            mix(j.plan->hist_frames); mix(j.plan->accepted_frames); mix(j.r->channels);
        }
        if (key == 0) key = 1;
        if (groups.size() > 1 && g.geo.mfma == 3 && pcm_bits == 0) {
            // several rate pairs in one batch: those of the split kernel share launches (launch_fir_split_multi: one item
            // table launch, one kernel launch per kernel build among them), as in rsmp_fir_lockstep_run
            split_jobs.push_back(rsmp::SplitJob{d_descs + first, static_cast<uint32_t>(g.members.size()), &g.geo, max_blocks, rp.nf});
        } else {
            RSMP_HIP_CHECK(rsmp::launch_fir_periodic(d_descs + first,
                                                     static_cast<uint32_t>(g.members.size()), g.geo,
                                                     max_blocks, leader->d_work_counter, rp.nf, stream, tail_fused, key, pcm_bits));
        }
        first += g.members.size();
    }
    if (!split_jobs.empty()) RSMP_HIP_CHECK(rsmp::launch_fir_split_multi(split_jobs.data(), split_jobs.size(), stream));
    if (leader->profiling) {
        RSMP_HIP_CHECK(rsmp::event_record(leader->prof_stop[leader->prof_count % rsmp_fir::kProfRing], stream));
        ++leader->prof_count;
    }
    // (RSMP_FIR_NO_REPAIR, debug: what the periodic kernels wrote, without the repair pass -- tools/repair_probe.py)
    static const bool no_repair = rsmp::knob("RSMP_FIR_NO_REPAIR") != nullptr;
    static const bool count_marks = rsmp::knob("RSMP_FIR_COUNT_MARKS") != nullptr;   // (debug: how many chunks the launch marked, per launch group)
    if (count_marks) {
        RSMP_HIP_CHECK(hipStreamSynchronize(stream));
        for (const Repair& rp : repairs) {
            const size_t words = 1 + (static_cast<size_t>(rp.count) * rp.nf.chunks + 31) / 32;
            std::vector<uint32_t> h(words);
            RSMP_HIP_CHECK(hipMemcpy(h.data(), rp.nf.words, words * sizeof(uint32_t), hipMemcpyDeviceToHost));
            size_t bits = 0;
            for (size_t w = 1; w < words; ++w) bits += static_cast<size_t>(__builtin_popcount(h[w]));
            fprintf(stderr, "[rsmp] launch group of %u streams x %u chunks: tag word %u (this launch's %u), %zu chunks marked\n", rp.count, rp.nf.chunks,
                    h[0], rp.nf.tag, bits);
            for (uint32_t st = 0; st < rp.count && st < 3; ++st) {   // (which: the first streams' chunk numbers)
                fprintf(stderr, "[rsmp]   stream %u:", st);
                for (uint32_t c = 0; c < rp.nf.chunks; ++c) {
                    const size_t bit = static_cast<size_t>(st) * rp.nf.chunks + c;
                    if (h[1 + (bit >> 5)] >> (bit & 31) & 1u) fprintf(stderr, " %u", c);
                }
                fprintf(stderr, "\n");
            }
        }
    }
    if (no_repair) repairs.clear();
    if (repairs.size() > 1) {
        std::vector<rsmp::RepairJob> rj;
        for (const Repair& rp : repairs) rj.push_back(rsmp::RepairJob{d_descs + rp.first, rp.count, rp.nf});
        RSMP_HIP_CHECK(rsmp::launch_fir_repair_multi(rj.data(), rj.size(), stream));
    } else {
        for (const Repair& rp : repairs)
            RSMP_HIP_CHECK(rsmp::launch_fir_repair(d_descs + rp.first, rp.count, rp.nf, stream));
    }
    if (n > n_generic && max_wraps > 0)
        RSMP_HIP_CHECK(rsmp::launch_fir_wrap_fixup(d_descs + n_generic,
                                                   static_cast<uint32_t>(n - n_generic), max_wraps,
                                                   stream));
    if (!tail_fused)
        RSMP_HIP_CHECK(rsmp::launch_fir_tail_copy(d_descs, static_cast<uint32_t>(n), max_tail_values,
                                                  stream));
    if (direct) {   // the slot may be rewritten once these kernels have read it
        RSMP_HIP_CHECK(rsmp::event_record(leader->plan_copied[slot], stream));
        leader->plan_pending[slot] = true;
        leader->plan_image[slot].clear();
    }
    // Commit: the mirrors advance, the hist buffers swap.
    for (Job& j : jobs) {
        j.r->last_periodic = j.plan->periodic;
        j.r->mirror = j.plan->planned;
        j.r->cur ^= 1;
    }
    return RSMP_OK;
}

void report_calls(const Plan& pl, size_t* calls, size_t max_calls, size_t* n_calls) {
    const size_t nc = pl.calls.size() / 2;
    if (n_calls) *n_calls = nc;
    if (calls)
        for (size_t i = 0; i < nc && i < max_calls; ++i) {
            calls[2 * i] = pl.calls[2 * i];
            calls[2 * i + 1] = pl.calls[2 * i + 1];
        }
}

int run_single_piece(rsmp_fir* r, const float* d_in, size_t in_len, float* d_out, size_t out_cap,
                     size_t chunk_len, size_t* consumed, size_t* produced, size_t* calls,
                     size_t max_calls, size_t* n_calls, hipStream_t stream);

// A launch's coefficient rows are mixed for ONE drift, the stream's f64 drift moves by ~1e-14 of a frame per output: a
// bulk call of more than kMaxLaunchOutputs outputs is cut into launches of at most that many (at call boundaries: the
// reference's loop, resample/src/main.rs:226-254, does not know the difference), each with the table of its own middle --
// 2e-7 of a frame from either end, 3e-7 of a full-scale sample.  (Config 5's 26.5 M outputs stay one launch.)
constexpr uint64_t kMaxLaunchOutputs = 46000000ull;

int run_single(rsmp_fir* r, const float* d_in, size_t in_len, float* d_out, size_t out_cap,
               size_t chunk_len, size_t* consumed, size_t* produced, size_t* calls,
               size_t max_calls, size_t* n_calls, hipStream_t stream) {
    const size_t ch = r->channels;
    if (chunk_len == 0 || chunk_len % ch != 0 || in_len % ch != 0)
        return run_single_piece(r, d_in, in_len, d_out, out_cap, chunk_len, consumed, produced, calls, max_calls, n_calls, stream);
    const size_t chunk_frames = chunk_len / ch;
    size_t piece_chunks = static_cast<size_t>(static_cast<double>(kMaxLaunchOutputs) * r->mirror.ratio() / static_cast<double>(chunk_frames));
    if (piece_chunks == 0) piece_chunks = 1;
    const size_t piece_len = piece_chunks * chunk_len;
    if (in_len <= piece_len)
        return run_single_piece(r, d_in, in_len, d_out, out_cap, chunk_len, consumed, produced, calls, max_calls, n_calls, stream);
    size_t off = 0, made = 0, n_total = 0;
    while (off < in_len) {
        const size_t take = std::min(piece_len, in_len - off);
        size_t c = 0, p = 0, nc = 0;
        const size_t room = n_total < max_calls ? max_calls - n_total : 0;
        if (int rc = run_single_piece(r, d_in + off, take, d_out + made, out_cap - made, chunk_len, &c, &p,
                                      calls && room ? calls + 2 * n_total : nullptr, room, &nc, stream))
            return rc;
        off += c;
        made += p;
        n_total += nc;
        if (c != take) break;   // (cannot happen in a bulk call: every call accepts what it is offered)
    }
    if (consumed) *consumed = off;
    if (produced) *produced = made;
    if (n_calls) *n_calls = n_total;
    return RSMP_OK;
}

int run_single_piece(rsmp_fir* r, const float* d_in, size_t in_len, float* d_out, size_t out_cap,
                     size_t chunk_len, size_t* consumed, size_t* produced, size_t* calls,
                     size_t max_calls, size_t* n_calls, hipStream_t stream) {
    std::vector<Job> jobs;
    jobs.push_back(Job{r, d_in, in_len, d_out, out_cap, chunk_len, nullptr});
    int rc = plan_job(jobs[0]);
    if (rc != RSMP_OK) return rc;
    rc = launch_jobs(r, jobs, stream);
    if (rc != RSMP_OK) return rc;
    if (consumed) *consumed = jobs[0].consumed();
    if (produced) *produced = jobs[0].produced();
    report_calls(*jobs[0].plan, calls, max_calls, n_calls);
    return RSMP_OK;
}

}  // namespace

// ================================ C ABI ==========================================================
extern "C" int rsmp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" rsmp_fir* rsmp_fir_new_from_hz(size_t channels, uint32_t input_rate_hz,
                                          uint32_t output_rate_hz, int latency, int attenuation,
                                          int device) {
    (void)plan_pool();   // (the planning workers exist from the process's first ResamplerFir on: see PlanPool)
    return fir_create(channels, input_rate_hz, output_rate_hz, latency, attenuation, device);
}

extern "C" rsmp_fir* rsmp_fir_new(size_t channels, int input_rate, int output_rate, int latency,
                                  int attenuation, int device) {
    const uint32_t in_hz = rsmp_sample_rate_hz(input_rate);
    const uint32_t out_hz = rsmp_sample_rate_hz(output_rate);
    if (!in_hz || !out_hz) {
        rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "ResamplerFir::new: invalid SampleRate value");
        return nullptr;
    }
    (void)plan_pool();
    return fir_create(channels, in_hz, out_hz, latency, attenuation, device);
}

extern "C" void rsmp_fir_free(rsmp_fir* r) { fir_destroy(r); }

extern "C" size_t rsmp_fir_buffer_size_output(const rsmp_fir* r) {
    return r->mirror.buffer_size_output_frames() * r->channels;
}
extern "C" size_t rsmp_fir_delay(const rsmp_fir* r) { return r->taps / 2; }
extern "C" size_t rsmp_fir_channels(const rsmp_fir* r) { return r->channels; }
extern "C" size_t rsmp_fir_taps(const rsmp_fir* r) { return r->taps; }
extern "C" size_t rsmp_fir_phases(const rsmp_fir* r) { (void)r; return rsmp::kPhases; }

extern "C" void rsmp_fir_state(const rsmp_fir* r, size_t* read_position, size_t* available_frames,
                               double* position) {
    if (read_position) *read_position = r->mirror.read_position();
    if (available_frames) *available_frames = r->mirror.available();
    if (position) *position = r->mirror.position();
}

// Puts the stream where a host-only plan stands: the plan's state becomes the handle's and the frames the
// reference would hold buffered at that point (available_frames of them) are taken from the END of
// `history`, the input that precedes the point.  With rsmp_fir_plan_bulk this starts a resampler in the
// middle of a stream -- a long stream is cut at call boundaries and each piece runs on its own GPU with
// the results of the unsharded run (exact position state from the host mirror plus a halo).
extern "C" int rsmp_fir_seek(rsmp_fir* r, const rsmp_fir_plan* p, const float* history, size_t history_len,
                             int history_on_device, void* stream_v) {
    if (!r || !p) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_seek: null argument");
    const rsmp::FirMirrorState& s = p->mirror.state();
    if (s.num != r->mirror.num() || s.den != r->mirror.den() || s.taps != r->mirror.taps())
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_seek: the plan is for another rate pair or latency");
    const size_t need = static_cast<size_t>(s.available) * r->channels;
    if (history_len < need || (need && !history))
        return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE, "rsmp_fir_seek: %zu values buffered at this point, history holds %zu",
                          need, history_len);
    DeviceGuard guard(r->device);
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : r->stream;
    if (r->last_stream_valid && r->last_stream != stream) RSMP_HIP_CHECK(hipStreamSynchronize(r->last_stream));
    r->last_stream = stream;
    r->last_stream_valid = true;
    if (need) {
        RSMP_HIP_CHECK(hipMemcpyAsync(r->d_hist[r->cur], history + (history_len - need), need * sizeof(float),
                                      history_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        if (!history_on_device) RSMP_HIP_CHECK(hipStreamSynchronize(stream));   // the caller's buffer is free on return
    }
    r->mirror.set_state(s);
    return RSMP_OK;
}

extern "C" void rsmp_fir_reset(rsmp_fir* r) {
    // resampler_fir.rs:638-642: only the three scalars; stale frames are unreachable.
    r->mirror.reset();
}

extern "C" void rsmp_fir_batch_reset(rsmp_fir* const* rs, size_t n) {
    for (size_t i = 0; i < n; ++i) rs[i]->mirror.reset();
}

extern "C" int rsmp_fir_set_kernel(rsmp_fir* r, int kernel) {
    if (kernel < RSMP_FIR_KERNEL_AUTO || kernel > RSMP_FIR_KERNEL_PERIODIC_F32)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_set_kernel: unknown kernel %d", kernel);
    r->kernel_mode = kernel;
    return RSMP_OK;
}

extern "C" int rsmp_fir_set_profiling(rsmp_fir* r, int enable) {
    DeviceGuard guard(r->device);
    if (enable && !r->prof_start[0])
        for (int i = 0; i < rsmp_fir::kProfRing; ++i) {
            RSMP_HIP_CHECK(hipEventCreate(&r->prof_start[i]));
            RSMP_HIP_CHECK(hipEventCreate(&r->prof_stop[i]));
        }
    r->profiling = enable != 0;
    r->prof_count = 0;
    return RSMP_OK;
}

extern "C" int rsmp_fir_last_kernel_ms(rsmp_fir* r, float* ms) {
    DeviceGuard guard(r->device);
    if (r->prof_count == 0 || !ms)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_last_kernel_ms: no profiled launch");
    const size_t i = (r->prof_count - 1) % rsmp_fir::kProfRing;
    RSMP_HIP_CHECK(hipEventSynchronize(r->prof_stop[i]));
    RSMP_HIP_CHECK(hipEventElapsedTime(ms, r->prof_start[i], r->prof_stop[i]));
    return RSMP_OK;
}

extern "C" int rsmp_fir_kernel_variant(const rsmp_fir* r) {
    if (!r) return -1;
    if (!r->last_periodic || !r->periodic.geo_valid || !r->periodic.geo.ok) return 0;
    const rsmp::PeriodicGeometry& g = r->periodic.geo;
    return g.mfma == 3 ? (g.planes == 3 ? 4 : 5) : (g.mfma ? 3 : (g.producers ? 2 : 1));
}

extern "C" int rsmp_fir_mean_kernel_ms(rsmp_fir* r, float* ms, size_t* launches) {
    DeviceGuard guard(r->device);
    if (r->prof_count == 0 || !ms)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_mean_kernel_ms: no profiled launch");
    const size_t n = r->prof_count < static_cast<size_t>(rsmp_fir::kProfRing) ? r->prof_count : rsmp_fir::kProfRing;
    RSMP_HIP_CHECK(hipEventSynchronize(r->prof_stop[(r->prof_count - 1) % rsmp_fir::kProfRing]));
    double sum = 0.0;
    for (size_t k = 0; k < n; ++k) {
        const size_t i = (r->prof_count - 1 - k) % rsmp_fir::kProfRing;
        float t = 0.f;
        RSMP_HIP_CHECK(hipEventElapsedTime(&t, r->prof_start[i], r->prof_stop[i]));
        sum += t;
    }
    *ms = static_cast<float>(sum / static_cast<double>(n));
    if (launches) *launches = n;
    return RSMP_OK;
}

extern "C" int rsmp_fir_resample_device(rsmp_fir* r, const float* d_in, size_t in_len, float* d_out,
                                        size_t out_len, size_t* consumed, size_t* produced,
                                        void* stream) {
    DeviceGuard guard(r->device);
    // resampler_fir.rs:514-519: input is validated before output.
    if (in_len % r->channels != 0)
        return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE, "Input buffer size is invalid");
    if (out_len % r->channels != 0)
        return rsmp::fail(RSMP_ERR_INVALID_OUTPUT_BUFFER_SIZE, "Output buffer size is invalid");
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : r->stream;
    return run_single(r, d_in, in_len, d_out, out_len, 0, consumed, produced, nullptr, 0, nullptr, s);
}

extern "C" int rsmp_fir_resample(rsmp_fir* r, const float* in, size_t in_len, float* out,
                                 size_t out_len, size_t* consumed, size_t* produced) {
    DeviceGuard guard(r->device);
    if (in_len % r->channels != 0)
        return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE, "Input buffer size is invalid");
    if (out_len % r->channels != 0)
        return rsmp::fail(RSMP_ERR_INVALID_OUTPUT_BUFFER_SIZE, "Output buffer size is invalid");
    // At most INPUT_CAPACITY frames can be accepted and buffer_size_output-ish produced per
    // call, so the staging buffers are bounded no matter how large the caller's slices are.
    const size_t max_in = rsmp::kInputCapacity * r->channels;
    const size_t stage_in = in_len < max_in ? in_len : max_in;
    const size_t max_out =
        (static_cast<size_t>(static_cast<double>(rsmp::kInputCapacity) / r->mirror.ratio()) + 8) *
        r->channels;
    const size_t stage_out = out_len < max_out ? out_len : max_out;
    RSMP_HIP_CHECK(hipStreamSynchronize(r->stream));
    // A streaming call is a few kilobytes: two copy-engine transfers and their synchronisation cost more than
    // the kernel.  Small calls therefore go through mapped host memory (cached on the device, visible at kernel
    // boundaries): the CPU copies the caller's slice in, the kernel reads it over the link (once: re-reads hit
    // L2) and writes the output back the same way.
    static const size_t zero_copy_max = [] {
        const char* e = getenv("RSMP_FIR_ZEROCOPY_MAX");   // bytes per direction; 0 disables
        return e ? static_cast<size_t>(atoll(e)) : static_cast<size_t>(256 * 1024);
    }();
    const bool zero_copy = (stage_in + 4) * sizeof(float) <= zero_copy_max && (stage_out + 4) * sizeof(float) <= zero_copy_max;
    float* d_in_stage = nullptr;
    float* d_out_stage = nullptr;
    if (zero_copy) {
        RSMP_HIP_CHECK(r->h_stage_in.reserve((stage_in + 4) * sizeof(float), false));
        RSMP_HIP_CHECK(r->h_stage_out.reserve((stage_out + 4) * sizeof(float), false));
        if (stage_in) std::memcpy(r->h_stage_in.get(), in, stage_in * sizeof(float));
        RSMP_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_in_stage), r->h_stage_in.get(), 0));
        RSMP_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_out_stage), r->h_stage_out.get(), 0));
    } else {
        RSMP_HIP_CHECK(r->d_stage_in.reserve((stage_in + 4) * sizeof(float)));
        RSMP_HIP_CHECK(r->d_stage_out.reserve((stage_out + 4) * sizeof(float)));
        if (stage_in)
            RSMP_HIP_CHECK(hipMemcpyAsync(r->d_stage_in.get(), in, stage_in * sizeof(float),
                                          hipMemcpyHostToDevice, r->stream));
        d_in_stage = r->d_stage_in.as<float>();
        d_out_stage = r->d_stage_out.as<float>();
    }
    size_t c = 0, p = 0;
    const int rc = run_single(r, d_in_stage, stage_in, d_out_stage, stage_out, 0, &c, &p, nullptr, 0, nullptr, r->stream);
    if (rc != RSMP_OK) return rc;
    if (p && !zero_copy)
        RSMP_HIP_CHECK(hipMemcpyAsync(out, r->d_stage_out.get(), p * sizeof(float),
                                      hipMemcpyDeviceToHost, r->stream));
    RSMP_HIP_CHECK(hipStreamSynchronize(r->stream));
    if (p && zero_copy) std::memcpy(out, r->h_stage_out.get(), p * sizeof(float));
    if (consumed) *consumed = c;
    if (produced) *produced = p;
    return RSMP_OK;
}

extern "C" size_t rsmp_fir_bulk_output_bound(const rsmp_fir* r, size_t in_len, size_t chunk_len) {
    (void)chunk_len;
    // Everything buffered plus everything offered, resampled, plus one frame of slack per call
    // boundary effect; generous but O(in_len).
    const size_t frames = in_len / r->channels + r->mirror.available();
    const size_t out_frames = static_cast<size_t>(static_cast<double>(frames) / r->mirror.ratio()) + 8;
    return out_frames * r->channels;
}

extern "C" int rsmp_fir_resample_bulk_device(rsmp_fir* r, const float* d_in, size_t in_len,
                                             size_t chunk_len, float* d_out, size_t out_cap,
                                             size_t* consumed, size_t* produced, size_t* calls,
                                             size_t max_calls, size_t* n_calls, void* stream) {
    DeviceGuard guard(r->device);
    if (chunk_len == 0)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_resample_bulk: chunk_len must be > 0");
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : r->stream;
    return run_single(r, d_in, in_len, d_out, out_cap, chunk_len, consumed, produced, calls,
                      max_calls, n_calls, s);
}

extern "C" int rsmp_fir_resample_bulk(rsmp_fir* r, const float* in, size_t in_len, size_t chunk_len,
                                      float* out, size_t out_cap, size_t* consumed, size_t* produced,
                                      size_t* calls, size_t max_calls, size_t* n_calls) {
    DeviceGuard guard(r->device);
    if (chunk_len == 0)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_resample_bulk: chunk_len must be > 0");
    RSMP_HIP_CHECK(hipStreamSynchronize(r->stream));
    RSMP_HIP_CHECK(r->d_stage_in.reserve((in_len + 4) * sizeof(float)));
    RSMP_HIP_CHECK(r->d_stage_out.reserve((out_cap + 4) * sizeof(float)));
    if (in_len)
        RSMP_HIP_CHECK(hipMemcpyAsync(r->d_stage_in.get(), in, in_len * sizeof(float),
                                      hipMemcpyHostToDevice, r->stream));
    size_t c = 0, p = 0;
    const int rc = run_single(r, r->d_stage_in.as<float>(), in_len, r->d_stage_out.as<float>(),
                              out_cap, chunk_len, &c, &p, calls, max_calls, n_calls, r->stream);
    if (rc != RSMP_OK) return rc;
    if (p)
        RSMP_HIP_CHECK(hipMemcpyAsync(out, r->d_stage_out.get(), p * sizeof(float),
                                      hipMemcpyDeviceToHost, r->stream));
    RSMP_HIP_CHECK(hipStreamSynchronize(r->stream));
    if (consumed) *consumed = c;
    if (produced) *produced = p;
    return RSMP_OK;
}

static int batch_bulk_piece(rsmp_fir* const* rs, size_t n, const float* const* d_in, const size_t* in_lens,
                            size_t chunk_len, float* const* d_out, const size_t* out_caps, size_t* consumed,
                            size_t* produced, void* stream, uint32_t pcm_bits = 0);

// The launch through the device planner, if the batch is one for it.  *took = 1: done (rc is the call's result).
static int batch_bulk_routed(rsmp_fir* const* rs, size_t n, const float* const* d_in, const size_t* in_lens, size_t chunk_len,
                             float* const* d_out, const size_t* out_caps, size_t* consumed, size_t* produced, void* stream, int planner,
                             int* took) {
    *took = 0;
    if (planner == 0 || n < 2) return RSMP_OK;
    const size_t ch = rs[0]->channels, length = in_lens[0];
    if (ch == 0 || chunk_len % ch != 0 || length % ch != 0) return RSMP_OK;
    const size_t frames = chunk_len / ch;
    // calls every stream accepts whole (rsmp_fir_lockstep_run_bulk), at least a handful of them, the same buffer length for all
    if (frames > kRoutedMaxCallFrames || length / ch < kRoutedMinCalls * frames) return RSMP_OK;
    for (size_t i = 0; i < n; ++i) {
        if (!rs[i] || rs[i]->channels != ch || rs[i]->device != rs[0]->device || in_lens[i] != length) return RSMP_OK;
        // room for what the launch will produce: the outputs below the limit once `length` more values are accepted, in exact arithmetic
        // (fir_mirror_fast.h: mirror_predict's m1), + 2 for an output that f64 puts a hair below it.  (rsmp_fir_bulk_output_bound is
        // no test here: it grows with the frames a stream has buffered, and a buffer sized by it before the stream's first launch
        // would fail it ever after.)  Anything else: the host path, which checks the room exactly and says so.
        {
            const rsmp::FirMirrorState st = rs[i]->mirror.state();
            const uint64_t a_now = st.abs_consumed + st.available + length / ch;
            if (st.num == 0 || st.den == 0 || st.den >= (1ull << 21) || st.num >= (1ull << 21) || a_now >= (1ull << 40)) return RSMP_OK;
            // ceil(x den / num): the outputs m >= 0 with m num / den < x
            const uint64_t m1 = a_now + 1 > st.taps ? ((a_now + 1 - st.taps) * st.den + st.num - 1) / st.num : 0;
            const uint64_t made = (m1 > st.abs_out ? m1 - st.abs_out : 0) + 2;
            if (out_caps[i] / ch < made) return RSMP_OK;
        }
        for (size_t k = 0; k < i; ++k)
            if (rs[k] == rs[i]) return RSMP_OK;
    }
    if (planner < 0) {
        size_t distinct = 0;
        if (rsmp_fir_batch_distinct_states(rs, n, &distinct) != RSMP_OK || distinct < std::min(kRoutedMinStates, n)) return RSMP_OK;
    }
    std::lock_guard<std::mutex> lock(routed_mu());
    auto& cache = routed_cache();
    const std::vector<rsmp_fir*> key(rs, rs + n);
    auto it = cache.find(key);
    if (it != cache.end() && it->second.ls == nullptr) {   // (a batch the lock-step entry has refused before: the host planner's)
        it->second.used = ++routed_clock;
        return RSMP_OK;
    }
    if (it != cache.end()) {
        int in_sync = 0;
        static const bool trace = rsmp::knob("RSMP_ROUTE_TRACE") != nullptr;
        const int rc_sync = rsmp_fir_lockstep_in_sync(it->second.ls, &in_sync);
        if (trace) fprintf(stderr, "[rsmp] routed batch: in_sync rc %d -> %d, frames %zu / %zu\n", rc_sync, in_sync, it->second.frames, frames);
        if (rc_sync != RSMP_OK || !in_sync || it->second.frames < frames) {
            rsmp_fir_lockstep_discard(it->second.ls);   // (the handles have moved on: they hold the newer state)
            cache.erase(it);
            it = cache.end();
        }
    }
    if (it == cache.end()) {
        // (a handle of this batch in ANOTHER cached batch: that one's device states go stale with this launch, which its own next use
        // finds out -- rsmp_fir_lockstep_in_sync --, nothing to do here)
        if (cache.size() >= kRoutedCacheSize) {
            auto oldest = cache.begin();
            for (auto jt = cache.begin(); jt != cache.end(); ++jt)
                if (jt->second.used < oldest->second.used) oldest = jt;
            rsmp_fir_lockstep_discard(oldest->second.ls);
            cache.erase(oldest);
        }
        rsmp_fir_lockstep* ls = rsmp_fir_lockstep_new(rs, n, frames);
        if (!ls) {   // (streams a lock-step batch does not take: the host planner's, without an error of this call's -- and remembered)
            rsmp::last_error_slot().clear();
            RoutedBatch none;
            none.used = ++routed_clock;
            cache.emplace(key, std::move(none));
            return RSMP_OK;
        }
        RoutedBatch rb;
        rb.ls = ls;
        rb.frames = frames;
        it = cache.emplace(key, std::move(rb)).first;
    }
    RoutedBatch& rb = it->second;
    rb.used = ++routed_clock;
    auto fail_and_drop = [&](int rc) {   // (whatever state the batch is in now: not one to keep)
        const std::string msg = rsmp::last_error_slot();
        rsmp_fir_lockstep_discard(rb.ls);
        cache.erase(it);
        rsmp::last_error_slot() = msg;
        *took = 1;
        return rc;
    };
    std::vector<const void*> bound;
    bound.reserve(2 * n);
    for (size_t i = 0; i < n; ++i) bound.push_back(d_in[i]);
    for (size_t i = 0; i < n; ++i) bound.push_back(d_out[i]);
    {
        static const bool trace = rsmp::knob("RSMP_ROUTE_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[rsmp] routed batch: %s\n", bound != rb.bound ? "bind" : "bound already");
    }
    if (bound != rb.bound) {
        if (rb.bound.empty()) {
            std::vector<size_t> caps(n);
            for (size_t i = 0; i < n; ++i) caps[i] = rsmp_fir_buffer_size_output(rs[i]);   // per CALL, as the reference sizes a call's buffer
            if (int rc = rsmp_fir_lockstep_bind(rb.ls, d_in, d_out, caps.data())) return fail_and_drop(rc);
        } else {   // (fresh buffers for this launch: the run planned ahead for it stays)
            if (int rc = rsmp_fir_lockstep_rebind_buffers(rb.ls, d_in, d_out, stream)) return fail_and_drop(rc);
        }
        rb.bound = bound;
    }
    if (int rc = rsmp_fir_lockstep_run_bulk(rb.ls, length / ch, frames, 0, 0, stream)) return fail_and_drop(rc);
    uint32_t flags = 0;
    std::vector<size_t> acc(n), made(n);
    if (int rc = rsmp_fir_lockstep_sync_totals(rb.ls, acc.data(), made.data(), &flags)) return fail_and_drop(rc);
    if (flags & (1u | 8u | 16u)) {
        rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "bulk batch planned on the device: status flags %u", flags);
        return fail_and_drop(RSMP_ERR_INVALID_ARGUMENT);
    }
    for (size_t i = 0; i < n; ++i) {
        if (consumed) consumed[i] = acc[i];
        if (produced) produced[i] = made[i];
    }
    *took = 1;
    return RSMP_OK;
}

extern "C" int rsmp_fir_batch_resample_bulk_device(rsmp_fir* const* rs, size_t n,
                                                   const float* const* d_in, const size_t* in_lens,
                                                   size_t chunk_len, float* const* d_out,
                                                   const size_t* out_caps, size_t* consumed,
                                                   size_t* produced, void* stream) {
    return rsmp_fir_batch_resample_bulk_device_ex(rs, n, d_in, in_lens, chunk_len, d_out, out_caps, consumed, produced, stream, -1, nullptr);
}

extern "C" int rsmp_fir_batch_resample_bulk_device_ex(rsmp_fir* const* rs, size_t n,
                                                      const float* const* d_in, const size_t* in_lens,
                                                      size_t chunk_len, float* const* d_out,
                                                      const size_t* out_caps, size_t* consumed,
                                                      size_t* produced, void* stream, int planner, int* planned_on_device) {
    if (planned_on_device) *planned_on_device = 0;
    if (n == 0) return RSMP_OK;
    if (!rs || !d_in || !in_lens || !d_out || !out_caps || chunk_len == 0)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_batch_resample_bulk_device: null/zero argument");
    {
        bool any_null = false;
        for (size_t i = 0; i < n; ++i) any_null = any_null || !rs[i];
        int took = 0;
        if (!any_null) {
            const int rc = batch_bulk_routed(rs, n, d_in, in_lens, chunk_len, d_out, out_caps, consumed, produced, stream, planner, &took);
            if (took) {
                if (planned_on_device) *planned_on_device = 1;
                return rc;
            }
        }
    }
    // A launch's coefficient rows are mixed for one drift (run_single): a stream offered more than kMaxLaunchOutputs outputs'
    // worth of input takes part in several launches, cut at call boundaries; the others are through after the first.
    std::vector<size_t> piece(n, 0);
    bool cut = false;
    for (size_t i = 0; i < n; ++i) {
        if (!rs[i]) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_batch_resample_bulk_device: null stream");
        const size_t ch = rs[i]->channels;
        piece[i] = in_lens[i];
        if (chunk_len % ch != 0 || in_lens[i] % ch != 0) continue;
        size_t chunks = static_cast<size_t>(static_cast<double>(kMaxLaunchOutputs) * rs[i]->mirror.ratio() / static_cast<double>(chunk_len / ch));
        if (chunks == 0) chunks = 1;
        if (in_lens[i] > chunks * chunk_len) {
            piece[i] = chunks * chunk_len;
            cut = true;
        }
    }
    if (!cut) return batch_bulk_piece(rs, n, d_in, in_lens, chunk_len, d_out, out_caps, consumed, produced, stream);
    std::vector<size_t> off(n, 0), made(n, 0);
    std::vector<char> stopped(n, 0);
    for (;;) {
        std::vector<rsmp_fir*> sub_rs;
        std::vector<const float*> sub_in;
        std::vector<float*> sub_out;
        std::vector<size_t> sub_len, sub_cap, idx;
        for (size_t i = 0; i < n; ++i) {
            if (stopped[i] || off[i] >= in_lens[i]) continue;   // (through; a stream offered nothing takes part in no launch)
            idx.push_back(i);
            sub_rs.push_back(rs[i]);
            sub_in.push_back(d_in[i] + off[i]);
            sub_len.push_back(std::min(piece[i], in_lens[i] - off[i]));
            sub_out.push_back(d_out[i] + made[i]);
            sub_cap.push_back(out_caps[i] - made[i]);
        }
        if (idx.empty()) break;
        std::vector<size_t> c(idx.size(), 0), p(idx.size(), 0);
        if (int rc = batch_bulk_piece(sub_rs.data(), idx.size(), sub_in.data(), sub_len.data(), chunk_len, sub_out.data(),
                                      sub_cap.data(), c.data(), p.data(), stream))
            return rc;
        bool progress = false;
        for (size_t k = 0; k < idx.size(); ++k) {
            off[idx[k]] += c[k];
            made[idx[k]] += p[k];
            if (c[k] != sub_len[k]) stopped[idx[k]] = 1;   // (cannot happen in a bulk call; the rest is not offered again)
            progress = progress || c[k] != 0;
        }
        if (!progress) break;
    }
    for (size_t i = 0; i < n; ++i) {
        if (consumed) consumed[i] = off[i];
        if (produced) produced[i] = made[i];
    }
    return RSMP_OK;
}

static int batch_bulk_piece(rsmp_fir* const* rs, size_t n, const float* const* d_in, const size_t* in_lens,
                            size_t chunk_len, float* const* d_out, const size_t* out_caps, size_t* consumed,
                            size_t* produced, void* stream, uint32_t pcm_bits) {
    for (size_t i = 0; i < n; ++i) {
        if (!rs[i] || rs[i]->device != rs[0]->device)
            return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "batch streams must share one device");
        for (size_t k = 0; k < i; ++k)
            if (rs[k] == rs[i])
                return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "batch lists the same stream twice");
    }
    DeviceGuard guard(rs[0]->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : rs[0]->stream;
    static const bool verbose_t = rsmp::knob("RSMP_FIR_VERBOSE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<Job> jobs;
    jobs.reserve(n);
    // Streams in the same state that are fed the same amount share one replay of the reference
    // call sequence (the control flow does not depend on the sample values).
    std::vector<std::pair<PlanKey, std::shared_ptr<Plan>>> memo;
    std::vector<size_t> rep;        // per job: index into memo
    std::vector<size_t> memo_job;   // per memo entry: the first job with that key
    for (size_t i = 0; i < n; ++i) {
        jobs.push_back(Job{rs[i], d_in[i], in_lens[i], d_out[i], out_caps[i], chunk_len, nullptr});
        const PlanKey key = make_key(jobs.back());
        size_t m = 0;
        while (m < memo.size() && !(memo[m].first == key)) ++m;
        if (m == memo.size()) {
            memo.emplace_back(key, nullptr);
            memo_job.push_back(i);
        }
        rep.push_back(m);
    }
    // Distinct keys are planned in parallel: replaying a long stream's control flow is ~0.5 ms of serial
    // f64 arithmetic on one core, and a batch of streams in different states has one replay per stream.  The
    // workers are a process-wide pool (creating a thread costs as much as a tenth of a replay).
    {
        const size_t todo = memo.size();
        std::vector<int> rcs(todo, RSMP_OK);
        std::vector<std::string> msgs(todo);
        auto work = [&](size_t m) {
            Job& j = jobs[memo_job[m]];
            rcs[m] = plan_job(j);
            if (rcs[m] != RSMP_OK) msgs[m] = rsmp::last_error_slot();   // (the slot is thread local)
            memo[m].second = j.plan;
        };
        if (todo <= 1) {
            if (todo == 1) work(0);
        } else {
            plan_pool().run(todo, work);
        }
        for (size_t m = 0; m < todo; ++m)
            if (rcs[m] != RSMP_OK) {
                rsmp::last_error_slot() = msgs[m];
                return rcs[m];
            }
    }
    for (size_t i = 0; i < n; ++i) {
        Job& j = jobs[i];
        j.plan = memo[rep[i]].second;
        if (j.plan->produced_frames * rs[i]->channels > out_caps[i])
            return rsmp::fail(RSMP_ERR_CAPACITY, "bulk output needs %zu values, room for %zu",
                              j.plan->produced_frames * rs[i]->channels, out_caps[i]);
    }
    const auto t_planned = std::chrono::steady_clock::now();
    const int rc = launch_jobs(rs[0], jobs, s, pcm_bits);
    if (rc != RSMP_OK) return rc;
    if (verbose_t) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "[rsmp] bulk batch of %zu streams (%zu distinct plans): planning %.3f ms, building and enqueueing the launch %.3f ms\n", n,
                memo.size(), std::chrono::duration<double, std::milli>(t_planned - t_begin).count(),
                std::chrono::duration<double, std::milli>(t_end - t_planned).count());
    }
    for (size_t i = 0; i < n; ++i) {
        if (consumed) consumed[i] = jobs[i].consumed();
        if (produced) produced[i] = jobs[i].produced();
    }
    return RSMP_OK;
}

// The bulk driver loop over a WAV file's samples as they are in the file (resample/src/main.rs:128-137 + :226-254):
// d_pcm[i] = little-endian PCM of `bits` (16 / 24 / 32) per sample, two channels a frame, in_lens[i] SAMPLES; the
// conversion happens where the kernels read their input -- the split kernel's prefetch loads, its edge and wrap-window
// paths, the tail copy, the repair pass -- so the launch reads the PCM alone: rsmp_pcm_to_stereo_f32_device + the f32
// entry point give the same samples with one more pass over HBM (PCM read, f32 written, f32 read).
extern "C" int rsmp_fir_batch_resample_bulk_pcm_device(rsmp_fir* const* rs, size_t n, const void* const* d_pcm, int bits,
                                                       const size_t* in_lens, size_t chunk_len, float* const* d_out,
                                                       const size_t* out_caps, size_t* consumed, size_t* produced, void* stream) {
    if (n == 0) return RSMP_OK;
    if (!rs || !d_pcm || !in_lens || !d_out || !out_caps || chunk_len == 0 || (bits != 16 && bits != 24 && bits != 32))
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_batch_resample_bulk_pcm_device: null / zero argument, or bits not 16 / 24 / 32");
    std::vector<const float*> in(n);
    for (size_t i = 0; i < n; ++i) {
        if (!rs[i] || rs[i]->channels != 2)
            return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_batch_resample_bulk_pcm_device: two-channel streams only");
        if (reinterpret_cast<uintptr_t>(d_pcm[i]) % 4 != 0)
            return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "PCM input must be 4-byte aligned");
        // (a launch's coefficient rows are mixed for one drift: a stream offered more than one launch's worth of input
        // goes through the f32 entry point, which cuts it -- 16 minutes of audio)
        const size_t chunks = static_cast<size_t>(static_cast<double>(kMaxLaunchOutputs) * rs[i]->mirror.ratio() / static_cast<double>(chunk_len / 2 ? chunk_len / 2 : 1));
        if (chunk_len % 2 == 0 && in_lens[i] % 2 == 0 && in_lens[i] > (chunks ? chunks : 1) * chunk_len)
            return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_batch_resample_bulk_pcm_device: more than one launch's worth of input (46 M outputs)");
        in[i] = static_cast<const float*>(d_pcm[i]);
    }
    return batch_bulk_piece(rs, n, in.data(), in_lens, chunk_len, d_out, out_caps, consumed, produced, stream, static_cast<uint32_t>(bits));
}
