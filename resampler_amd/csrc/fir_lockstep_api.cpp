// fir_lockstep_api.cpp -- C ABI of the lock-step batch (rsmp_fir_lockstep_*): BASELINE config 4's
// shape, a fixed set of ResamplerFir instances (src/resampler_fir.rs:179-643) that are each fed one
// chunk per step.  Creation sorts the streams by rate pair / table, builds the per-workgroup groups
// and moves the streams' reference state to HBM; a step is one launch of fir_lockstep_kernel with
// constant arguments -- no per-stream host work, no upload, no host-side state machine.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <tuple>
#include <vector>

#include "common.h"
#include "device_util.h"
#include "fir_handle.h"
#include "fir_lockstep.h"
#include "fir_table_refresher.h"

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
    bool rebased = false;   // the buffers changed under a run planned ahead (rsmp_fir_lockstep_rebind_buffers): its descriptors are patched when it is taken over
    hipStream_t last_stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::vector<uint64_t> h_counts;
    std::vector<FirMirrorState> h_states;
    uint32_t hist_parity = 0;   // 0: the next step / run reads `hist` of LockstepStream and leaves its tail in `hist_alt`
    // rsmp_fir_lockstep_run (k calls per stream and launch): the bulk kernels' geometry per rate pair, the run's
    // descriptors and what the device-side planner leaves for them (fir_lockstep_run.hip)
    struct RunGroup { rsmp::PeriodicGeometry geo; size_t first = 0, count = 0; uint32_t max_out_step = 0; };
    int run_state = 0;          // 0: not looked at yet, 1: every rate pair has a bulk kernel, -1: runs are loops of steps
    std::vector<RunGroup> run_groups;
    // The run's descriptors, bitmaps, per-call counts and call records exist twice ("slots", used alternately): the NEXT run
    // is planned ahead on a stream of its own while the current one computes (plan_ahead below) and must not overwrite
    // what the current run's kernels and the caller (run_counts) still read.
    struct RunSlot { DeviceBuffer descs, bits, counts, recs; hipEvent_t compute_done = nullptr; hipStream_t compute_stream = nullptr; bool used = false, compute_recorded = false;
                     // the split kernel's item tables of the run planned ahead into this slot, built on the plan stream behind its plan
                     DeviceBuffer items; uint64_t items_seq = 0, items_ops = 0; bool items_valid = false; };
    RunSlot slot[2];
    int next_slot = 0, last_slot = 0;
    DeviceBuffer d_run_rs, d_run_nf, d_run_work, d_run_preds, d_run_states0;
    // planned ahead: states / append positions / last counts / status flags of the run AFTER the current one, in scratch
    // copies until the run is really asked for (then committed by one small kernel), or dropped
    DeviceBuffer sp_states, sp_cursor, sp_last, sp_status;
    hipStream_t plan_stream = nullptr;   // the candidate picked for the caller's stream of the last run (pick_plan_stream)
    hipStream_t plan_candidates[2] = {nullptr, nullptr};
    // caller's stream -> the candidate that runs beside it, found out by a probe that the DEVICE decides and the host
    // never waits for (launch_fir_lockstep_probe_wait): until it is known, runs on that stream are not planned ahead
    struct PlanPick { int pick = -1; bool decided = false, probing = false; int cand = 0, tries = 0; };
    std::map<hipStream_t, PlanPick> plan_pick;
    DeviceBuffer d_probe;                // the probes' flag word
    rsmp::PinnedBuffer h_probe;          // ... and their result
    hipEvent_t probe_ev = nullptr;
    hipStream_t probe_owner = nullptr;   // the caller's stream whose probe is in flight (one at a time)
    uint32_t probe_token = 0;
    hipEvent_t ev_ready = nullptr, plan_done = nullptr, ev_commit = nullptr;
    hipStream_t ahead_q = nullptr;       // the plan stream the run planned ahead was enqueued on
    bool ahead_waited = false;           // the caller's stream `ahead_waited_on` already waits for plan_done (rsmp_fir_lockstep_run)
    hipStream_t ahead_waited_on = nullptr;
    uint64_t stat_table_ops = 0;         // patch launches + table uploads enqueued on a caller's stream (poll_drift, flush_tables)
    uint64_t stat_commits_on_plan_stream = 0;
    struct RunKey { uint32_t k = 0, in_frames = 0, append = 0, parity = 0; uint64_t in_offset = 0, seq = 0; int slot = 0; bool valid = false; };
    RunKey ahead;               // what the plan stream was asked to plan
    bool ahead_inflight = false;   // ... and has not been waited for since
    RunKey prev;                // the previous run (the pattern the next one is guessed from)
    uint64_t run_seq = 0;
    uint32_t run_wrap_words = 0, run_k = 0, run_nf_tag = 0;
    bool run_planned = false;   // the most recent run went through the device planner
    size_t run_counts_k = 0;    // calls of the most recent run whose counts are in slot[last_slot].counts
    std::vector<uint32_t> h_run_counts;
    // The class tables follow the streams' f64 drift.  The reference's position (src/resampler_fir.rs:589) moves away from
    // the exact rational one by ~1e-14 of a frame per output for as long as a stream runs (every add rounds on the grid of
    // its binade): 1e-6 of a frame after half an hour of audio -- 2e-6 of a full-scale sample with coefficient rows mixed
    // for another drift.  Streams of one key whose drifts lie together form a class; the class's tables are built for the
    // drift of its first stream, which is read back from the device now and then (asynchronously: a tiny kernel, a copy
    // into pinned memory, an event looked at when the next step or run is enqueued); when it has moved by more than
    // kLsDriftTolerance, the tables are replaced (class_table_for: cached per device, built on the host otherwise).
    struct DriftClass {
        uint32_t rep = 0;                 // internal index of the stream that stands for the class
        size_t first = 0, count = 0;      // its streams, internal order
        double table_drift = 0.0;         // what the bound tables were built for
        const rsmp_fir* r0 = nullptr;
        bool has_step = false, has_run = false;
        rsmp::PeriodicGeometry step_geo, run_geo;
        rsmp::ClassTable step_table, run_table;
        // Tables of the process-wide cache this class has bound (creation, reset: the host knows the states and may
        // wait) stay held until nothing enqueued or planned ahead can read them (bind / reset, behind their waits) --
        // the cache is bounded, and a table it has evicted lives by its holders alone.
        std::vector<std::shared_ptr<void>> holds;
        // The NEXT tables: owned double-buffered device images that the batch's worker thread fills when the drift has
        // covered most of the way to the tolerance (TableRefresher: host arithmetic, allocation, upload and the wait
        // for it all happen there); the crossing swaps pointers.
        rsmp::TableRefresher::Table* step_next = nullptr;
        rsmp::TableRefresher::Table* run_next = nullptr;
        bool next_pending = false;        // a request is out (or its result is waiting to be taken)
        double next_drift = 0.0;
        double seen_drift = 0.0;          // the class's drift as last read back ...
        double rate = 0.0;                // ... and how fast it moves per input frame (from the last two readings)
        bool late = false;                // past the tolerance, the next tables not there yet
    };
    std::vector<DriftClass> classes;
    std::unique_ptr<rsmp::TableRefresher> refresher;
    double drift_tolerance = 0.0;         // (set at creation: kLsDriftTolerance; rsmp_fir_lockstep_set_drift_policy)
    uint64_t drift_check_frames = 0;
    size_t n_late = 0;                    // classes currently `late`
    // A reading tells where the DEVICE was when the gather kernel ran; what the host enqueues now runs later -- by as much
    // as the host is ahead of the device (a caller that never waits: thousands of launches, tens of millions of frames per
    // stream: several tolerances of drift).  Decisions are made for the drift a launch enqueued NOW will see: the last
    // reading + the measured rate x the frames enqueued since that reading was asked for.
    uint64_t frames_total = 0;            // input frames per stream enqueued through this batch so far
    uint64_t frames_at_inflight = 0;      // ... when the reading in flight was asked for
    uint64_t frames_at_seen = 0;          // ... when the latest completed reading was asked for
    bool have_seen = false;
    uint64_t frames_at_eval = 0;          // (when the classes were last looked at)
    std::vector<rsmp::TableRefresher::Table*> guards_due;   // images unbound by this call's replacements (record_guards)
    // diagnostics (rsmp_fir_lockstep_stats)
    uint64_t stat_ahead_hits = 0, stat_ahead_misses = 0, stat_late_polls = 0, stat_table_waits = 0, stat_probes = 0;
    std::vector<rsmp::LsRunStream> h_run_rs;
    DeviceBuffer d_drift_reps;
    rsmp::PinnedBuffer h_drift, h_stage;            // the drifts read back; staging of the group / stream tables when they change
    hipEvent_t drift_ev = nullptr, stage_ev = nullptr;
    bool drift_inflight = false, stage_inflight = false, groups_dirty = false, rs_dirty = false;
    uint64_t frames_since_drift = 0;
    size_t table_rebinds = 0;                 // times a class got new tables (diagnostic)
    // optional timing of the step launches (rsmp_fir_lockstep_set_profiling): ring of event pairs
    static constexpr int kProfRing = 256;
    bool profiling = false;
    hipEvent_t prof_start[kProfRing] = {}, prof_stop[kProfRing] = {};
    size_t prof_count = 0;
};

namespace {

int drop_plan_ahead(rsmp_fir_lockstep* ls, hipStream_t s);   // (rsmp_fir_lockstep_run, below)

constexpr double kLsDriftQuantum = 1e-8;      // tables are built for drifts on this grid (frames)
constexpr double kLsDriftClass = 2e-8;        // streams of one key whose drifts round to the same multiple share a class
// A class's tables are replaced when its drift is further from theirs than this: 2e-7 of a full-scale sample at worst,
// a fifth of the 1e-6 the path is allowed.  (Tighter costs: a replacement is a table built on the host per class, ~0.5 ms;
// config 4 on one GPU runs 0.13 M frames of every stream per millisecond, and at 4e-8 its six classes were rebuilt every
// 35 ms -- 10 % of the bench's step.  A real-time stream crosses 1.2e-7 every five minutes.)
constexpr double kLsDriftTolerance = 1.2e-7;
constexpr uint64_t kLsDriftCheckFrames = 1u << 19;   // input frames per stream between two looks at the drifts (~5e-9 of drift)

double quantized_drift(double d) { return std::round(d / kLsDriftQuantum) * kLsDriftQuantum; }

// Class `c` takes `step` / `run` as its tables (either may be null: not replaced), built for drift `t`: the host copies
// of the group and stream tables are changed; the caller moves them to the device (flush_tables, or a patch kernel).
void bind_class_tables(rsmp_fir_lockstep* ls, size_t c, const rsmp::ClassTable* step, const rsmp::ClassTable* run, double t) {
    rsmp_fir_lockstep::DriftClass& cl = ls->classes[c];
    if (step) {
        if (cl.step_table.hold) cl.holds.push_back(cl.step_table.hold);   // (a run planned ahead may still name it)
        cl.step_table = *step;
        for (LockstepGroup& g : ls->groups)
            if (g.periodic && g.pad0 == c) {
                g.class_coef = step->d_coef;
                g.class_meta = step->d_meta;
            }
    }
    if (run) {
        if (cl.run_table.hold) cl.holds.push_back(cl.run_table.hold);
        cl.run_table = *run;
        for (size_t i = cl.first; i < cl.first + cl.count; ++i) {
            ls->h_run_rs[i].class_coef = run->d_coef;
            ls->h_run_rs[i].class_wrap_coef = run->d_wrap_coef;
            ls->h_run_rs[i].class_meta = run->d_meta;
            ls->h_run_rs[i].drift = t;
        }
    }
    cl.table_drift = t;
    ++ls->table_rebinds;
}

// New tables for class `c` from the process-wide cache, built for drift `d` -- where the HOST knows the states and may
// wait (creation, reset): class_table_for builds on this thread, allocates and copies synchronously.
int rebind_class_blocking(rsmp_fir_lockstep* ls, size_t c, double d) {
    rsmp_fir_lockstep::DriftClass& cl = ls->classes[c];
    const double t = quantized_drift(d);
    rsmp::ClassTable st, rt;
    const bool want_run = cl.has_run && ls->run_state == 1;
    if (cl.has_step)
        if (int rc = rsmp::class_table_for(ls->device, *cl.r0->table, cl.step_geo, t, &st)) return rc;
    if (want_run)
        if (int rc = rsmp::class_table_for(ls->device, *cl.r0->table, cl.run_geo, t, &rt)) return rc;
    bind_class_tables(ls, c, cl.has_step ? &st : nullptr, want_run ? &rt : nullptr, t);
    if (cl.has_step) ls->groups_dirty = true;
    if (want_run) ls->rs_dirty = true;
    if (cl.late) { cl.late = false; --ls->n_late; }
    return RSMP_OK;
}

// Images unbound by a replacement may be overwritten behind everything enqueued on `s` so far.  An image whose guard
// could not be recorded stays in the list (the next call tries again before it asks for anything).
int record_due_guards(rsmp_fir_lockstep* ls, hipStream_t s) {
    while (!ls->guards_due.empty()) {
        if (int rc = ls->refresher->record_guard(ls->guards_due.back(), s)) return rc;
        ls->guards_due.pop_back();
    }
    return RSMP_OK;
}

// The launch path's side of a replacement.  Where the drifts that have come back from the device say so, a class's next
// tables are ASKED FOR (most of the way to the tolerance: one event record), and a class past the tolerance TAKES the
// tables the worker has left for it (pointer swaps + one patch kernel for all classes of this look).  Nothing here
// builds, allocates or copies.  It WAITS for the worker only where a class is three tolerances past its tables without new
// ones: back-pressure on a caller that enqueues without ever waiting -- the image a replacement overwrites was bound
// until the replacement before it, and the device must have passed that point (TableRefresher's guard), so the host
// can be about two table generations ahead of the device and no more (tools/soak_lockstep.py --hours 24 at ~25 k
// launches per second of host time: 374 waits in 29 k runs, none in the bench's 64 launches or behind a caller that
// synchronises now and then; counted in stat_table_waits).
int poll_drift(rsmp_fir_lockstep* ls, hipStream_t s) {
    // Guards a previous call took but did not get to record (it failed between its replacement and request_drift): the
    // images it unbound must not be refilled before everything enqueued so far has passed -- recorded here, in front of
    // any request() below, they are later than needed and never stale.
    if (int rc = record_due_guards(ls, s)) return rc;
    bool fresh = false;
    if (ls->drift_inflight) {
        if (hipEventQuery(ls->drift_ev) == hipSuccess) {
            ls->drift_inflight = false;
            fresh = true;
            const double* d = ls->h_drift.as<double>();
            const bool rate_ok = ls->have_seen && ls->frames_at_inflight > ls->frames_at_seen;
            const double span = rate_ok ? static_cast<double>(ls->frames_at_inflight - ls->frames_at_seen) : 1.0;
            for (size_t c = 0; c < ls->classes.size(); ++c) {
                rsmp_fir_lockstep::DriftClass& cl = ls->classes[c];
                if (rate_ok) cl.rate = (d[c] - cl.seen_drift) / span;
                cl.seen_drift = d[c];
            }
            ls->frames_at_seen = ls->frames_at_inflight;
            ls->have_seen = true;
        } else {
            (void)hipGetLastError();   // (hipErrorNotReady is not an error here)
        }
    }
    // (looked at when a reading has come in, while a class is late, and every 2^17 frames in between: the host's lead grows)
    if (!fresh && ls->n_late == 0 && ls->frames_total - ls->frames_at_eval < (1u << 17)) return RSMP_OK;
    ls->frames_at_eval = ls->frames_total;
    const double lead = ls->have_seen ? static_cast<double>(ls->frames_total - ls->frames_at_seen) : 0.0;
    using TR = rsmp::TableRefresher;
    const double tol = ls->drift_tolerance;
    rsmp::LsPatchArgs pa;
    pa.groups = ls->d_groups.as<LockstepGroup>();
    pa.rs = ls->run_state == 1 ? ls->d_run_rs.as<rsmp::LsRunStream>() : nullptr;
    pa.n_groups = static_cast<uint32_t>(ls->groups.size());
    pa.n_streams = static_cast<uint32_t>(ls->rs.size());
    pa.n_patches = 0;
    pa.pad = 0;
    auto flush_patches = [&]() -> int {
        if (pa.n_patches) {
            RSMP_HIP_CHECK(rsmp::launch_fir_lockstep_patch_tables(pa, s));
            ++ls->stat_table_ops;
        }
        pa.n_patches = 0;
        return RSMP_OK;
    };
    for (size_t c = 0; c < ls->classes.size(); ++c) {
        rsmp_fir_lockstep::DriftClass& cl = ls->classes[c];
        const double now_drift = cl.seen_drift + cl.rate * lead;   // what a launch enqueued now will see
        const double off = now_drift - cl.table_drift;
        const bool want_step = cl.has_step, want_run = cl.has_run && ls->run_state == 1;
        if (!want_step && !want_run) continue;
        auto state_of = [](TR::Table* t) { return t ? t->state.load(std::memory_order_acquire) : static_cast<int>(TR::kReady); };
        auto ask = [&](double nd) -> int {
            cl.next_drift = nd;
            if (want_step) if (int rc = ls->refresher->request(cl.step_next, nd)) return rc;
            if (want_run) if (int rc = ls->refresher->request(cl.run_next, nd)) return rc;
            cl.next_pending = true;
            return RSMP_OK;
        };
        if (std::fabs(off) > tol) {
            if (!cl.late) { cl.late = true; ++ls->n_late; }
            int st = want_step ? state_of(cl.step_next) : TR::kReady, rt = want_run ? state_of(cl.run_next) : TR::kReady;
            if (cl.next_pending && (st == TR::kRequested || rt == TR::kRequested) && std::fabs(off) > 3.0 * tol) {
                if (want_step) ls->refresher->wait(cl.step_next);
                if (want_run) ls->refresher->wait(cl.run_next);
                ++ls->stat_table_waits;
                st = want_step ? state_of(cl.step_next) : TR::kReady;
                rt = want_run ? state_of(cl.run_next) : TR::kReady;
            }
            if (cl.next_pending && (st == TR::kRequested || rt == TR::kRequested)) {
                static const bool verbose = rsmp::knob("RSMP_FIR_VERBOSE") != nullptr;
                if (verbose) fprintf(stderr, "[rsmp] class %zu: drift %.3g past its tables' %.3g, the next ones (%.3g) on their way\n", c, now_drift, cl.table_drift, cl.next_drift);
                ++ls->stat_late_polls;   // on their way: the old tables serve a little longer (a fifth of the bound per tolerance)
                continue;
            }
            if (cl.next_pending && (st == TR::kFailed || rt == TR::kFailed))
                return rsmp::fail(RSMP_ERR_HIP, "lock-step batch: the replacement class tables could not be made");
            if (cl.next_pending && std::fabs(now_drift - cl.next_drift) <= 0.5 * tol) {
                rsmp::ClassTable stt, rtt;
                if (want_step) { stt = ls->refresher->take(cl.step_next); ls->guards_due.push_back(cl.step_next); }
                if (want_run) { rtt = ls->refresher->take(cl.run_next); ls->guards_due.push_back(cl.run_next); }
                cl.next_pending = false;
                static const bool verbose = rsmp::knob("RSMP_FIR_VERBOSE") != nullptr;
                if (verbose) fprintf(stderr, "[rsmp] class %zu: drift %.3g (read %.3g + lead), tables %.3g -> %.3g\n", c, now_drift, cl.seen_drift, cl.table_drift, cl.next_drift);
                bind_class_tables(ls, c, want_step ? &stt : nullptr, want_run ? &rtt : nullptr, cl.next_drift);
                cl.late = false;
                --ls->n_late;
                rsmp::LsTablePatch& q = pa.p[pa.n_patches++];
                q.cls = static_cast<uint32_t>(c);
                q.first = static_cast<uint32_t>(cl.first);
                q.count = static_cast<uint32_t>(cl.count);
                q.flags = (want_step ? 1u : 0u) | (want_run ? 2u : 0u);
                q.step_coef = stt.d_coef;
                q.step_meta = stt.d_meta;
                q.run_coef = rtt.d_coef;
                q.run_wrap_coef = rtt.d_wrap_coef;
                q.run_meta = rtt.d_meta;
                q.drift = cl.next_drift;
                if (pa.n_patches == rsmp::kLsMaxPatches) if (int rc = flush_patches()) return rc;
                continue;
            }
            // nothing asked for yet, or what was prepared is for another drift (a jump): ask now
            if (cl.next_pending) {   // (both results are in: drop them, the images are free again)
                if (want_step) ls->refresher->discard(cl.step_next);
                if (want_run) ls->refresher->discard(cl.run_next);
                cl.next_pending = false;
            }
            ++ls->stat_late_polls;
            static const bool verbose = rsmp::knob("RSMP_FIR_VERBOSE") != nullptr;
            if (verbose) fprintf(stderr, "[rsmp] class %zu: drift %.3g past its tables' %.3g with nothing asked for\n", c, now_drift, cl.table_drift);
            if (int rc = ask(quantized_drift(now_drift))) return rc;
        } else {
            if (cl.late) { cl.late = false; --ls->n_late; }   // (the extrapolation came back inside the tolerance: no more polling on its account)
            if (std::fabs(off) > 0.6 * tol && !cl.next_pending) {
                // most of the way: the tables the class will want at the crossing are made now, beside everything else
                if (int rc = ask(quantized_drift(cl.table_drift + (off > 0.0 ? tol : -tol)))) return rc;
            }
        }
    }
    return flush_patches();
}

// Changed group / stream tables go to the device as a whole, in stream order in front of what is enqueued next: the
// blocking paths' way (reset, the first run).  (Called where no planner of the batch is running: the plan stream has
// been waited for.)
int flush_tables(rsmp_fir_lockstep* ls, hipStream_t s) {
    if (!ls->groups_dirty && !ls->rs_dirty) return RSMP_OK;
    ++ls->stat_table_ops;
    const size_t gb = ls->groups.size() * sizeof(LockstepGroup), rb = ls->h_run_rs.size() * sizeof(rsmp::LsRunStream);
    if (ls->stage_inflight) {   // (the staging memory of the previous change: long since read)
        RSMP_HIP_CHECK(hipEventSynchronize(ls->stage_ev));
        ls->stage_inflight = false;
    }
    RSMP_HIP_CHECK(ls->h_stage.reserve(gb + rb));
    char* h = ls->h_stage.as<char>();
    if (ls->groups_dirty) {
        std::memcpy(h, ls->groups.data(), gb);
        RSMP_HIP_CHECK(hipMemcpyAsync(ls->d_groups.get(), h, gb, hipMemcpyHostToDevice, s));
    }
    if (ls->rs_dirty && ls->run_state == 1) {
        std::memcpy(h + gb, ls->h_run_rs.data(), rb);
        RSMP_HIP_CHECK(hipMemcpyAsync(ls->d_run_rs.get(), h + gb, rb, hipMemcpyHostToDevice, s));
    }
    RSMP_HIP_CHECK(rsmp::event_record(ls->stage_ev, s));
    ls->stage_inflight = true;
    ls->groups_dirty = ls->rs_dirty = false;
    return RSMP_OK;
}

// After a step or run of `frames` input frames per stream: now and then the classes' drifts start their way to the host.
// Images this call's replacements have unbound may be overwritten behind everything enqueued so far.
int request_drift(rsmp_fir_lockstep* ls, hipStream_t s, uint64_t frames) {
    if (int rc = record_due_guards(ls, s)) return rc;
    ls->frames_since_drift += frames;
    ls->frames_total += frames;
    if (ls->drift_inflight || ls->frames_since_drift < ls->drift_check_frames || ls->classes.empty()) return RSMP_OK;
    const uint32_t nc = static_cast<uint32_t>(ls->classes.size());
    // (the kernel stores straight into the mapped, coherent host buffer: a copy-engine operation in the stream costs the
    // stream ~0.1 ms of cross-queue synchronisation, 6-9 % of config 4's step when done every fourth run)
    RSMP_HIP_CHECK(rsmp::launch_fir_lockstep_gather_drift(ls->d_states.as<FirMirrorState>(), ls->d_drift_reps.as<uint32_t>(),
                                                          ls->h_drift.as<double>(), nc, s));
    RSMP_HIP_CHECK(rsmp::event_record(ls->drift_ev, s));
    ls->drift_inflight = true;
    ls->frames_at_inflight = ls->frames_total;
    ls->frames_since_drift = 0;
    return RSMP_OK;
}

// The host knows the states (creation, reset): every class gets the tables of its first stream's drift at once.  What
// the worker was asked for belongs to the old states: waited for and dropped.
int rebind_from_host_states(rsmp_fir_lockstep* ls) {
    if (ls->drift_inflight) {   // (what is on its way belongs to the old states)
        RSMP_HIP_CHECK(hipEventSynchronize(ls->drift_ev));
        ls->drift_inflight = false;
    }
    ls->frames_since_drift = 0;
    ls->have_seen = false;
    ls->frames_at_seen = ls->frames_at_inflight = ls->frames_at_eval = ls->frames_total;
    for (size_t c = 0; c < ls->classes.size(); ++c) {
        rsmp_fir_lockstep::DriftClass& cl = ls->classes[c];
        if (cl.next_pending) {
            for (rsmp::TableRefresher::Table* t : {cl.step_next, cl.run_next})
                if (t && t->state.load(std::memory_order_acquire) != rsmp::TableRefresher::kIdle) {
                    ls->refresher->wait(t);
                    ls->refresher->discard(t);   // (nobody bound the image: it is the next to be filled again)
                }
            cl.next_pending = false;
        }
        const double d = ls->rs[ls->order[cl.rep]]->mirror.drift();
        cl.seen_drift = d;
        cl.rate = 0.0;
        if (std::fabs(d - cl.table_drift) > kLsDriftQuantum || cl.late)
            if (int rc = rebind_class_blocking(ls, c, d)) return rc;
    }
    return RSMP_OK;
}

// A stream's buffered frames alternate between its two history buffers (fir_lockstep.h, LockstepStream):
// the next step (index ls->step) reads `hist` when its index is even.  After any number of steps the handle's
// `cur` is made to name the buffer that holds the frames.
void refresh_history_index(rsmp_fir_lockstep* ls) {
    if (!ls->bound) return;
    for (size_t k = 0; k < ls->rs.size(); ++k) {
        rsmp_fir* r = ls->rs[ls->order[k]];
        const float* live = ls->hist_parity ? ls->streams[k].hist_alt : ls->streams[k].hist;
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
    ls->drift_tolerance = kLsDriftTolerance;
    ls->drift_check_frames = kLsDriftCheckFrames;
    ls->refresher.reset(new rsmp::TableRefresher(ls->device));   // (its thread starts with the batch's first request)
    // Streams that share a polyphase table, a rate pair and a channel count share a class table and
    // a geometry: they become neighbours, then workgroups of `slots` streams.
    // (a stream set to RSMP_FIR_KERNEL_PERIODIC_F32 keeps every product in f32: its own groups)
    // (... and whose f64 drifts lie together: DriftClass)
    typedef std::tuple<const void*, uint32_t, uint32_t, size_t, size_t, bool, long long> Key;
    auto exact_of = [](const rsmp_fir* r) { return r->kernel_mode != RSMP_FIR_KERNEL_AUTO; };
    auto key_of = [&](const rsmp_fir* r) {
        return Key(static_cast<const void*>(r->table.get()), r->in_hz, r->out_hz, r->channels, r->taps, exact_of(r),
                   std::llround(r->mirror.drift() / kLsDriftClass));
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
        rsmp_fir_lockstep::DriftClass cl;
        cl.rep = static_cast<uint32_t>(k);
        cl.first = k;
        cl.count = e - k;
        cl.r0 = r0;
        cl.table_drift = quantized_drift(r0->mirror.drift());
        cl.has_step = geo.periodic;
        if (geo.periodic) {
            cl.step_geo = rsmp::lockstep_class_geometry(geo);
            if (rsmp::class_table_for(ls->device, *r0->table, cl.step_geo, cl.table_drift, &ct) != RSMP_OK)
                return nullptr;
            cl.step_table = ct;
            cl.step_next = ls->refresher->add_table(cl.step_geo, r0->table);
            if (!cl.step_next) {
                rsmp::fail(RSMP_ERR_HIP, "lock-step batch: cannot create an event");
                return nullptr;
            }
        }
        cl.seen_drift = cl.table_drift;
        const uint32_t class_index = static_cast<uint32_t>(ls->classes.size());
        ls->classes.push_back(std::move(cl));
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
            g.pad0 = class_index;   // (host side only: which DriftClass the group's tables belong to)
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
        ls->d_drift_reps.reserve(ls->classes.size() * sizeof(uint32_t)) != hipSuccess ||
        ls->h_drift.reserve(ls->classes.size() * sizeof(double)) != hipSuccess ||
        hipEventCreateWithFlags(&ls->drift_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ls->stage_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ls->probe_ev, hipEventDisableTiming) != hipSuccess ||
        ls->d_probe.reserve(sizeof(uint32_t)) != hipSuccess ||
        ls->h_probe.reserve(sizeof(uint32_t)) != hipSuccess ||
        hipMemset(ls->d_probe.get(), 0, sizeof(uint32_t)) != hipSuccess ||
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
        [&] {
            std::vector<uint32_t> reps;
            for (const auto& cl : ls->classes) reps.push_back(cl.rep);
            return hipMemcpy(ls->d_drift_reps.get(), reps.data(), reps.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        }() != hipSuccess ||
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

static void lockstep_destroy(rsmp_fir_lockstep* ls, bool write_back);
extern "C" void rsmp_fir_lockstep_free(rsmp_fir_lockstep* ls) { lockstep_destroy(ls, true); }
extern "C" void rsmp_fir_lockstep_discard(rsmp_fir_lockstep* ls) { lockstep_destroy(ls, false); }
static void lockstep_destroy(rsmp_fir_lockstep* ls, bool write_back) {
    if (!ls) return;
    DeviceGuard guard(ls->device);
    if (write_back) (void)rsmp_fir_lockstep_sync(ls);
    else if (ls->last_stream) (void)hipStreamSynchronize(ls->last_stream);
    if (ls->drift_ev) (void)hipEventDestroy(ls->drift_ev);
    if (ls->stage_ev) (void)hipEventDestroy(ls->stage_ev);
    if (ls->probe_ev) {
        (void)hipEventSynchronize(ls->probe_ev);   // (a probe's kernels name d_probe / h_probe)
        (void)hipEventDestroy(ls->probe_ev);
    }
    for (hipStream_t& q : ls->plan_candidates) {
        if (!q) continue;
        (void)hipStreamSynchronize(q);
        (void)hipStreamDestroy(q);
        q = nullptr;
    }
    if (ls->plan_stream) {
        ls->plan_stream = nullptr;
    }
    for (hipEvent_t e : {ls->ev_ready, ls->plan_done, ls->ev_commit, ls->slot[0].compute_done, ls->slot[1].compute_done})
        if (e) (void)hipEventDestroy(e);
    if (ls->own_stream) {
        rsmp::split_release_stream(ls->device, ls->own_stream);
        (void)hipStreamDestroy(ls->own_stream);
    }
    for (hipEvent_t e : ls->prof_start) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ls->prof_stop) if (e) (void)hipEventDestroy(e);
    ls->refresher.reset();   // (joins its thread, frees the images: every kernel that read them has been waited for above)
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
    if (int rc = drop_plan_ahead(ls, nullptr)) return rc;
    ls->prev.valid = false;
    for (auto& cl : ls->classes) cl.holds.clear();   // (nothing enqueued or planned ahead names a replaced cache table any more)
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
        s.hist = r->d_hist[ls->hist_parity ? r->cur ^ 1 : r->cur];       // the next step reads the handle's current buffer
        s.hist_alt = r->d_hist[ls->hist_parity ? r->cur : r->cur ^ 1];
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

extern "C" int rsmp_fir_lockstep_rebind_buffers(rsmp_fir_lockstep* ls, const float* const* d_in, float* const* d_out, void* stream) {
    if (!ls || !d_in || !d_out) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_rebind_buffers: null argument");
    if (!ls->bound) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_rebind_buffers: the batch has not been bound yet");
    DeviceGuard guard(ls->device);
    const size_t n = ls->rs.size();
    bool aligned8 = true;
    for (size_t i = 0; i < n; ++i)
        if (reinterpret_cast<uintptr_t>(d_in[i]) % 8 != 0) aligned8 = false;
    if (aligned8 != ls->in_aligned8) {   // (another build of the step kernel's loads: a bind proper, with the capacities as they are)
        std::vector<size_t> caps(n);
        for (size_t k = 0; k < n; ++k) caps[ls->order[k]] = static_cast<size_t>(ls->streams[k].out_cap_frames) * ls->rs[ls->order[k]]->channels;
        return rsmp_fir_lockstep_bind(ls, d_in, d_out, caps.data());
    }
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ls->own_stream;
    if (ls->last_stream && ls->last_stream != s) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    // A run planned ahead survives if it starts at the front of `out` (nothing it computed depends on the buffers but the two pointers
    // in its descriptors); one that appends behind what the old buffers hold does not.
    const bool keep = ls->ahead_inflight && ls->ahead.valid && ls->ahead.append == 0;
    if (!keep) {
        if (int rc = drop_plan_ahead(ls, s)) return rc;
    }
    for (size_t k = 0; k < n; ++k) {
        const uint32_t i = ls->order[k];
        ls->streams[k].in = d_in[i];
        ls->streams[k].out = d_out[i];
    }
    // (in stream order: steps and runs enqueued before this read the old table, those behind it the new one.  A planner already running
    // on the plan stream may see either -- its descriptors are patched when the run is taken over.)
    RSMP_HIP_CHECK(hipMemcpyAsync(ls->d_streams.get(), ls->streams.data(), n * sizeof(LockstepStream), hipMemcpyHostToDevice, s));
    RSMP_HIP_CHECK(hipMemsetAsync(ls->d_cursor.get(), 0, n * sizeof(uint64_t), s));   // nothing has been appended to the new buffers
    ls->rebased = keep;
    ls->last_stream = s;
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
    if (int rc = drop_plan_ahead(ls, s)) return rc;   // (a run planned ahead read the states this step is about to change)
    ls->prev.valid = false;
    if (int rc = poll_drift(ls, s)) return rc;
    if (int rc = flush_tables(ls, s)) return rc;
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
    a.hist_parity = ls->hist_parity;
    ls->hist_parity ^= 1u;
    ls->run_counts_k = 0;
    if (ls->profiling)
        RSMP_HIP_CHECK(rsmp::event_record(ls->prof_start[ls->prof_count % rsmp_fir_lockstep::kProfRing], s));
    RSMP_HIP_CHECK(rsmp::launch_fir_lockstep(a, static_cast<uint32_t>(ls->groups.size()), ls->max_lds, s));
    if (ls->profiling) {
        RSMP_HIP_CHECK(rsmp::event_record(ls->prof_stop[ls->prof_count % rsmp_fir_lockstep::kProfRing], s));
        ++ls->prof_count;
    }
    ls->last_stream = s;
    return request_drift(ls, s, in_frames);
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

extern "C" int rsmp_fir_lockstep_sync_totals(rsmp_fir_lockstep* ls, size_t* accepted, size_t* produced, uint32_t* status_or) {
    if (!ls) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_sync_totals: null batch");
    const size_t n = ls->rs.size();
    const std::vector<FirMirrorState> before = ls->h_states;   // (internal order: what was last exchanged with the handles)
    if (before.size() != n) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_sync_totals: the batch has no states yet");
    if (int rc = rsmp_fir_lockstep_sync(ls)) return rc;
    for (size_t k = 0; k < n; ++k) {
        const uint32_t i = ls->order[k];
        const size_t ch = ls->rs[i]->channels;
        const FirMirrorState &a = before[k], &b = ls->h_states[k];
        if (accepted) accepted[i] = static_cast<size_t>((b.abs_consumed + b.available) - (a.abs_consumed + a.available)) * ch;
        if (produced) produced[i] = static_cast<size_t>(b.abs_out - a.abs_out) * ch;
    }
    if (status_or) {
        std::vector<uint32_t> st(n);
        RSMP_HIP_CHECK(hipMemcpy(st.data(), ls->d_status.get(), n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint32_t v = 0;
        for (uint32_t f : st) v |= f;
        *status_or = v;
    }
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_in_sync(const rsmp_fir_lockstep* ls, int* in_sync) {
    if (!ls || !in_sync) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_in_sync: null argument");
    const size_t n = ls->rs.size();
    *in_sync = 0;
    if (ls->h_states.size() != n) return RSMP_OK;
    for (size_t k = 0; k < n; ++k) {
        const rsmp_fir* r = ls->rs[ls->order[k]];
        const FirMirrorState now = r->mirror.state();
        if (memcmp(&now, &ls->h_states[k], sizeof now) != 0) return RSMP_OK;
        // (the frames the stream has buffered: in the buffer the batch's next step reads)
        const float* live = ls->hist_parity ? ls->streams[k].hist_alt : ls->streams[k].hist;
        if (ls->bound && live != r->d_hist[r->cur]) return RSMP_OK;
    }
    *in_sync = 1;
    return RSMP_OK;
}

extern "C" int rsmp_fir_batch_distinct_states(rsmp_fir* const* rs, size_t n, size_t* distinct) {
    if (!rs || !distinct) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_batch_distinct_states: null argument");
    std::vector<FirMirrorState> seen;
    for (size_t i = 0; i < n; ++i) {
        if (!rs[i]) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_batch_distinct_states: null stream");
        const FirMirrorState s = rs[i]->mirror.state();
        bool found = false;
        for (const FirMirrorState& t : seen)
            if (memcmp(&s, &t, sizeof s) == 0) { found = true; break; }
        if (!found) seen.push_back(s);
    }
    *distinct = seen.size();
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_reset(rsmp_fir_lockstep* ls) {
    if (!ls) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_reset: null batch");
    DeviceGuard guard(ls->device);
    const size_t n = ls->rs.size();
    if (ls->last_stream) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    if (int rc = drop_plan_ahead(ls, nullptr)) return rc;
    ls->prev.valid = false;
    for (auto& cl : ls->classes) cl.holds.clear();
    for (rsmp_fir* r : ls->rs) r->mirror.reset();   // resampler_fir.rs:638-642
    RSMP_HIP_CHECK(hipMemset(ls->d_cursor.get(), 0, n * sizeof(uint64_t)));
    RSMP_HIP_CHECK(hipMemset(ls->d_status.get(), 0, n * sizeof(uint32_t)));
    ++ls->epoch;   // plans made ahead belong to the old states
    if (int rc = upload_states(ls)) return rc;
    if (int rc = rebind_from_host_states(ls)) return rc;   // (fresh streams: drift 0)
    if (int rc = flush_tables(ls, ls->own_stream)) return rc;
    RSMP_HIP_CHECK(hipStreamSynchronize(ls->own_stream));
    return RSMP_OK;
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

extern "C" int rsmp_fir_lockstep_kernel_ms(rsmp_fir_lockstep* ls, float* ms, size_t cap, size_t* launches) {
    if (!ls || !ms || !launches) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_kernel_ms: null argument");
    DeviceGuard guard(ls->device);
    const size_t ring = rsmp_fir_lockstep::kProfRing;
    const size_t n = std::min(cap, std::min(ls->prof_count, ring));
    *launches = n;
    if (n == 0) return RSMP_OK;
    RSMP_HIP_CHECK(hipEventSynchronize(ls->prof_stop[(ls->prof_count - 1) % ring]));
    for (size_t k = 0; k < n; ++k) {   // oldest first
        const size_t i = (ls->prof_count - n + k) % ring;
        RSMP_HIP_CHECK(hipEventElapsedTime(&ms[k], ls->prof_start[i], ls->prof_stop[i]));
    }
    return RSMP_OK;
}

// ---- rsmp_fir_lockstep_run: k consecutive calls per stream in one go -----------------------------------------
namespace {

// The plan stream of a caller's stream: one that runs BESIDE it.  HIP deals a handful of hardware queues to its streams in
// turn, and two streams on one queue run their kernels one after the other (the bench's torch stream and the batch's plan
// stream met on one: planned ahead, nothing overlapped).  Nothing tells which queue a stream has, so it is tried out, once per
// caller's stream, by a probe the DEVICE decides (launch_fir_lockstep_probe_wait: a wave on the caller's stream waits up to
// 1 ms for a word that a kernel on the candidate stores; behind it on one queue that kernel cannot start in time) and whose
// result the host looks at when its event has passed -- no host clock, no host wait: round 4's version timed the pair with
// the host's clock against 200 us around a hipStreamSynchronize, which a busy host fails for both candidates, and the batch
// then silently ran without plan-ahead.  Two candidates created one after the other sit on different queues, so at most one of
// them shares the caller's; a candidate that fails is tried once more after the other one (a host descheduled for a
// millisecond between the two launches reads like a shared queue).  Until a stream's answer is in, its runs plan on the
// stream itself.  (A stream of higher priority has a queue of its own for certain -- and starves the short kernels between
// the bulk launches: a run of 16 calls took 2.7x as long.)
int pick_plan_stream(rsmp_fir_lockstep* ls, hipStream_t s) {
    rsmp_fir_lockstep::PlanPick& pp = ls->plan_pick[s];
    static const bool verbose = rsmp::knob("RSMP_FIR_VERBOSE") != nullptr;
    if (!pp.decided) {
        if (pp.probing && ls->probe_owner == s) {
            if (hipEventQuery(ls->probe_ev) == hipSuccess) {
                pp.probing = false;
                ls->probe_owner = nullptr;
                if (*ls->h_probe.as<volatile uint32_t>() == 1u) {
                    pp.pick = pp.cand;
                    pp.decided = true;
                } else if (++pp.tries >= 4) {
                    pp.pick = -1;
                    pp.decided = true;
                } else {
                    pp.cand ^= 1;
                }
                if (verbose && pp.decided)
                    fprintf(stderr, "[rsmp] lock-step run: plan stream candidate %d runs beside stream %p (%d probes)\n", pp.pick,
                            static_cast<void*>(s), pp.tries + 1);
            } else {
                (void)hipGetLastError();
            }
        }
        if (!pp.decided && !pp.probing && ls->probe_owner == nullptr) {
            const uint32_t token = ++ls->probe_token ? ls->probe_token : ++ls->probe_token;
            *ls->h_probe.as<volatile uint32_t>() = 0u;
            RSMP_HIP_CHECK(rsmp::launch_fir_lockstep_probe_wait(ls->d_probe.as<uint32_t>(), token, 100000u, ls->h_probe.as<uint32_t>(), s));
            RSMP_HIP_CHECK(rsmp::launch_fir_lockstep_probe_set(ls->d_probe.as<uint32_t>(), token, ls->plan_candidates[pp.cand]));
            RSMP_HIP_CHECK(rsmp::event_record(ls->probe_ev, s));
            pp.probing = true;
            ls->probe_owner = s;
            ++ls->stat_probes;
        }
    }
    ls->plan_stream = pp.decided && pp.pick >= 0 ? ls->plan_candidates[pp.pick] : nullptr;
    return RSMP_OK;
}

// Whatever the plan stream was asked to plan ahead is dropped: the caller did something else than repeat its run.  Its
// kernels only WRITE scratch copies and the other slot, but they READ the batch's states, append positions and bound
// pointers: before those change, the plan stream must have finished (`s`: that stream waits; null: the host does).
int drop_plan_ahead(rsmp_fir_lockstep* ls, hipStream_t s) {
    if (ls->ahead_inflight && ls->plan_done) {
        if (s) RSMP_HIP_CHECK(rsmp::stream_wait_event(s, ls->plan_done));
        else RSMP_HIP_CHECK(hipEventSynchronize(ls->plan_done));
    }
    ls->ahead_inflight = false;
    ls->ahead_waited = false;
    ls->ahead.valid = false;
    return RSMP_OK;
}

// The bulk kernels' side of a batch: one geometry + class table per rate pair, the constant half of every
// stream's descriptor, the planner's waves.  Done once, at the first run.
int prepare_run(rsmp_fir_lockstep* ls) {
    if (ls->run_state != 0) return RSMP_OK;
    const size_t n = ls->rs.size();
    std::vector<rsmp::FirStreamDesc> descs(n);
    std::vector<rsmp::LsRunStream> rstreams(n);
    std::memset(descs.data(), 0, n * sizeof(rsmp::FirStreamDesc));
    ls->run_groups.clear();
    for (size_t c = 0; c < ls->classes.size(); ++c) {   // one group of streams per drift class (key + drift: rsmp_fir_lockstep_new)
        rsmp_fir_lockstep::DriftClass& cl = ls->classes[c];
        const size_t k = cl.first, e = cl.first + cl.count;
        const rsmp_fir* r0 = cl.r0;
        const int mode = r0->kernel_mode;
        rsmp::PeriodicGeometry geo;
        if (mode != RSMP_FIR_KERNEL_GENERIC && r0->mirror.periodic_ok())
            geo = rsmp::periodic_geometry(r0->mirror.num(), r0->mirror.den(), static_cast<uint32_t>(r0->taps),
                                          static_cast<uint32_t>(r0->channels), mode != RSMP_FIR_KERNEL_PERIODIC_VECTOR,
                                          mode != RSMP_FIR_KERNEL_PERIODIC_F32);
        const uint64_t den = r0->mirror.den();
        // (a geometry without the wrap variant in the kernel wants a list of wrapped outputs, which the device
        // planner does not keep -- unless the ratio is exact in f64 and no output ever wraps)
        if (!geo.ok || (!geo.inline_wraps && (den & (den - 1)) != 0)) {
            // runs are loops of steps; their per-call counts are gathered with the streams' caller indices
            for (auto& x : ls->classes) x.has_run = false;
            for (size_t i = 0; i < n; ++i) rstreams[i].caller = ls->order[i];
            if (ls->d_run_rs.reserve(n * sizeof(rsmp::LsRunStream)) != hipSuccess)
                return rsmp::fail(RSMP_ERR_HIP, "lock-step run: cannot allocate device state");
            RSMP_HIP_CHECK(hipMemcpy(ls->d_run_rs.get(), rstreams.data(), n * sizeof(rsmp::LsRunStream), hipMemcpyHostToDevice));
            ls->run_state = -1;
            return RSMP_OK;
        }
        rsmp::ClassTable ct;
        if (rsmp::class_table_for(ls->device, *r0->table, geo, cl.table_drift, &ct) != RSMP_OK) return RSMP_ERR_HIP;
        cl.has_run = true;
        cl.run_geo = geo;
        cl.run_table = ct;
        if (!cl.run_next) cl.run_next = ls->refresher->add_table(geo, r0->table);
        if (!cl.run_next) return rsmp::fail(RSMP_ERR_HIP, "lock-step run: cannot create an event");
        rsmp_fir_lockstep::RunGroup g;
        g.geo = geo;
        g.first = k;
        g.count = e - k;
        g.max_out_step = static_cast<uint32_t>(std::ceil(static_cast<double>(ls->step_frames + 8) / r0->mirror.ratio())) + 2;
        ls->run_groups.push_back(g);
        for (size_t i = k; i < e; ++i) {
            const rsmp_fir* r = ls->rs[ls->order[i]];
            rsmp::FirStreamDesc& d = descs[i];
            d.coeffs = r->d_coeffs;
            d.class_coef = ct.d_coef;
            d.class_wrap_coef = ct.d_wrap_coef;
            d.class_meta = ct.d_meta;
            d.channels = static_cast<uint32_t>(r->channels);
            d.taps = static_cast<uint32_t>(r->taps);
            d.num = static_cast<uint32_t>(r->mirror.num());
            d.den = static_cast<uint32_t>(den);
            d.drift = cl.table_drift;
            rstreams[i].wrap_unit = geo.mfma == 3 ? geo.b : geo.den;
            rstreams[i].den = static_cast<uint32_t>(den);
            rstreams[i].channels = static_cast<uint32_t>(r->channels);
            rstreams[i].caller = ls->order[i];
            rstreams[i].class_coef = ct.d_coef;          // (the planner writes these into the run's descriptor: they follow the drift)
            rstreams[i].class_wrap_coef = ct.d_wrap_coef;
            rstreams[i].class_meta = ct.d_meta;
            rstreams[i].drift = cl.table_drift;
        }
    }
    ls->h_run_rs = rstreams;
    if (ls->slot[0].descs.reserve(n * sizeof(rsmp::FirStreamDesc)) != hipSuccess ||
        ls->slot[1].descs.reserve(n * sizeof(rsmp::FirStreamDesc)) != hipSuccess ||
        ls->sp_states.reserve(n * sizeof(FirMirrorState)) != hipSuccess ||
        ls->sp_cursor.reserve(n * sizeof(uint64_t)) != hipSuccess ||
        ls->sp_last.reserve(2 * n * sizeof(uint64_t)) != hipSuccess ||
        ls->sp_status.reserve(n * sizeof(uint32_t)) != hipSuccess ||
        hipStreamCreateWithFlags(&ls->plan_candidates[0], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ls->plan_candidates[1], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ls->ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ls->plan_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ls->ev_commit, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ls->slot[0].compute_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ls->slot[1].compute_done, hipEventDisableTiming) != hipSuccess ||
        ls->d_run_rs.reserve(n * sizeof(rsmp::LsRunStream)) != hipSuccess ||
        ls->d_run_states0.reserve(n * sizeof(FirMirrorState)) != hipSuccess ||
        ls->d_run_work.reserve(std::max<size_t>(64, ls->run_groups.size()) * sizeof(unsigned long long)) != hipSuccess)
        return rsmp::fail(RSMP_ERR_HIP, "lock-step run: cannot allocate device state");
    for (auto& sl : ls->slot)
        RSMP_HIP_CHECK(hipMemcpy(sl.descs.get(), descs.data(), n * sizeof(rsmp::FirStreamDesc), hipMemcpyHostToDevice));
    RSMP_HIP_CHECK(hipMemcpy(ls->d_run_rs.get(), rstreams.data(), n * sizeof(rsmp::LsRunStream), hipMemcpyHostToDevice));
    // (one work counter per run group: a batch of many drift classes on a non-split geometry has more than 64)
    RSMP_HIP_CHECK(hipMemset(ls->d_run_work.get(), 0, std::max<size_t>(64, ls->run_groups.size()) * sizeof(unsigned long long)));
    ls->run_state = 1;
    return RSMP_OK;
}

}  // namespace

extern "C" int rsmp_fir_lockstep_run(rsmp_fir_lockstep* ls, size_t k_steps, size_t in_frames, size_t in_offset_frames,
                                     int append, void* stream) {
    if (!ls || !ls->bound)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_run: bind buffers first");
    if (k_steps == 0) return RSMP_OK;
    if (in_frames > ls->step_frames)
        return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE,
                          "lock-step batch: %zu frames offered, created for %u per step", in_frames, ls->step_frames);
    DeviceGuard guard(ls->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ls->own_stream;
    if (int rc = prepare_run(ls)) return rc;
    // A run is defined as k steps.  The bulk kernels read a stream's accepted frames as ONE span of its input, which
    // they are as long as every call accepts all it is offered (resampler_fir.rs:524-528: always, unless the step is
    // nearly as long as the reference's 4096-frame input buffer).
    size_t max_taps = 0;
    for (const rsmp_fir* r : ls->rs) max_taps = std::max(max_taps, r->taps);
    const bool whole_accept = in_frames + max_taps + 8 <= rsmp::kMirrorInputCapacity;
    // The outputs of a run's calls follow each other in `out`: behind what earlier steps / runs appended, or -- without
    // `append` -- from the front of the buffer (the append position starts again there).
    // (the planner's counters of a run -- outputs, frames, bitmap bits -- are 32-bit: a run whose inputs or whose outputs of
    // any rate pair could pass 2^27 / 2^31 is a loop of steps; 8 -> 384 kHz makes 48 outputs per frame)
    uint64_t max_out_run = 0;
    for (const auto& g : ls->run_groups) max_out_run = std::max<uint64_t>(max_out_run, k_steps * static_cast<uint64_t>(g.max_out_step));
    const bool loop_of_steps = ls->run_state < 0 || k_steps == 1 || !whole_accept ||
                               k_steps * static_cast<uint64_t>(ls->step_frames) > (1u << 27) || max_out_run >= (1ull << 31);
    const size_t n = ls->rs.size();
    if (k_steps > (1u << 20)) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_run: at most 2^20 calls per run");
    const uint32_t k = static_cast<uint32_t>(k_steps);
    // workspaces that grow with k: the per-call counts and records, the predictions, the bitmaps of wrapped outputs, the
    // non-finite marks.  (A reallocation waits for both streams: kernels planned ahead may still use the old buffers.)
    uint32_t wrap_words = 1, nf_words = 0;
    for (const auto& g : ls->run_groups) {
        const uint64_t n_out_max = static_cast<uint64_t>(k) * g.max_out_step;
        wrap_words = std::max<uint32_t>(wrap_words, static_cast<uint32_t>(n_out_max / g.geo.den / 32 + 2));
        nf_words += 1 + static_cast<uint32_t>((g.count * ((n_out_max >> rsmp::kNfChunkShift) + 1) + 31) / 32);
    }
    if (k > ls->run_k || (!loop_of_steps && wrap_words > ls->run_wrap_words)) {
        RSMP_HIP_CHECK(hipStreamSynchronize(s));
        if (int rc = drop_plan_ahead(ls, nullptr)) return rc;
        const uint32_t kk = std::max(k, ls->run_k), ww = std::max(wrap_words, ls->run_wrap_words);
        for (auto& sl : ls->slot)
            if (sl.counts.reserve(2 * n * static_cast<size_t>(kk) * sizeof(uint32_t)) != hipSuccess ||
                sl.recs.reserve(n * static_cast<size_t>(kk) * 24) != hipSuccess ||
                sl.bits.reserve(n * static_cast<size_t>(ww) * sizeof(uint32_t)) != hipSuccess)
                return rsmp::fail(RSMP_ERR_HIP, "lock-step run: cannot allocate the plan of %u calls", k);
        if (ls->d_run_preds.reserve(n * static_cast<size_t>(kk) * sizeof(rsmp::MirrorPred)) != hipSuccess)
            return rsmp::fail(RSMP_ERR_HIP, "lock-step run: cannot allocate the plan of %u calls", k);
        ls->run_k = kk;
        ls->run_wrap_words = ww;
    }
    if (loop_of_steps) {
        if (ls->last_stream && ls->last_stream != s) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));   // (in front of the memset below)
        if (int rc = drop_plan_ahead(ls, s)) return rc;
        ls->prev.valid = false;
        if (!append)   // (the device planner starts a run at the front itself: no launch for it)
            RSMP_HIP_CHECK(hipMemsetAsync(ls->d_cursor.get(), 0, n * sizeof(uint64_t), s));
        const int sl = ls->next_slot;
        for (uint32_t i = 0; i < k; ++i) {
            if (int rc = rsmp_fir_lockstep_step(ls, in_frames, in_offset_frames + static_cast<size_t>(i) * in_frames, nullptr, 1, stream))
                return rc;
            RSMP_HIP_CHECK(rsmp::launch_fir_lockstep_gather_counts(ls->d_counts.as<uint64_t>(), ls->d_run_rs.as<rsmp::LsRunStream>(),
                                                                   ls->slot[sl].counts.as<uint32_t>() + 2 * n * static_cast<size_t>(i),
                                                                   static_cast<uint32_t>(n), s));
        }
        ls->last_slot = sl;
        ls->run_counts_k = k;
        ls->run_planned = false;
        return RSMP_OK;
    }
    if (ls->last_stream && ls->last_stream != s) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    if (nf_words * sizeof(uint32_t) > ls->d_run_nf.capacity()) {
        RSMP_HIP_CHECK(hipStreamSynchronize(s));
        RSMP_HIP_CHECK(ls->d_run_nf.reserve(nf_words * sizeof(uint32_t)));
        RSMP_HIP_CHECK(hipMemsetAsync(ls->d_run_nf.get(), 0, ls->d_run_nf.capacity(), s));
    }
    // ---- the plan: taken from the plan stream if this very run was planned ahead there, made here otherwise --------
    const int sl = ls->next_slot;
    rsmp_fir_lockstep::RunKey key;
    key.k = k;
    key.in_frames = static_cast<uint32_t>(in_frames);
    key.append = append ? 1u : 0u;
    key.parity = ls->hist_parity;
    key.in_offset = in_offset_frames;
    key.seq = ls->run_seq;
    key.slot = sl;
    key.valid = true;
    auto same_key = [](const rsmp_fir_lockstep::RunKey& x, const rsmp_fir_lockstep::RunKey& y) {
        return x.valid && y.valid && x.k == y.k && x.in_frames == y.in_frames && x.append == y.append && x.parity == y.parity &&
               x.in_offset == y.in_offset && x.seq == y.seq && x.slot == y.slot;
    };
    auto plan_args = [&](const rsmp_fir_lockstep::RunKey& kk, bool scratch) {
        rsmp::LsRunArgs a;
        rsmp_fir_lockstep::RunSlot& t = ls->slot[kk.slot];
        a.streams = ls->d_streams.as<LockstepStream>();
        a.rs = ls->d_run_rs.as<rsmp::LsRunStream>();
        a.states_in = ls->d_states.as<FirMirrorState>();
        a.states_out = scratch ? ls->sp_states.as<FirMirrorState>() : ls->d_states.as<FirMirrorState>();
        a.states_before = ls->d_run_states0.as<FirMirrorState>();
        a.preds = ls->d_run_preds.as<rsmp::MirrorPred>();
        a.call_recs = t.recs.get();
        a.cursor_in = ls->d_cursor.as<uint64_t>();
        a.cursor_out = scratch ? ls->sp_cursor.as<uint64_t>() : ls->d_cursor.as<uint64_t>();
        a.descs = t.descs.as<rsmp::FirStreamDesc>();
        a.wrap_bits = t.bits.as<uint32_t>();
        a.counts = t.counts.as<uint32_t>();
        a.last_counts = scratch ? ls->sp_last.as<uint64_t>() : ls->d_counts.as<uint64_t>();
        a.status = scratch ? ls->sp_status.as<uint32_t>() : ls->d_status.as<uint32_t>();
        a.zero_status = scratch ? ls->sp_status.as<uint32_t>() : nullptr;
        a.in_offset = kk.in_offset;
        a.n_streams = static_cast<uint32_t>(n);
        a.k = kk.k;
        a.in_frames = kk.in_frames;
        a.wrap_words = ls->run_wrap_words;
        a.append = kk.append;
        a.hist_parity = kk.parity;
        return a;
    };
    if (ls->profiling)
        RSMP_HIP_CHECK(rsmp::event_record(ls->prof_start[ls->prof_count % rsmp_fir_lockstep::kProfRing], s));
    bool commit_on_q = false;   // this run's states were committed on the plan stream (below)
    bool plan_taken_over = false;
    bool commit_pending = false;   // ... are still to be committed on the caller's stream: by K1 of the next run, or a launch of its own
    rsmp::LsCommitArgs c{};
    if (same_key(ls->ahead, key)) {
        // planned while the previous run computed: wait for it (an event, no host block) and take its results over
        // (a big batch's caller stream has waited already, behind the previous run's split launch -- see below)
        if (!(ls->ahead_waited && ls->ahead_waited_on == s)) RSMP_HIP_CHECK(rsmp::stream_wait_event(s, ls->plan_done));
        ls->ahead_waited = false;
        ls->ahead_inflight = false;
        plan_taken_over = true;
        ++ls->stat_ahead_hits;
        if (ls->rebased) {   // (planned when the batch was bound to other buffers: the two pointers of every descriptor again)
            RSMP_HIP_CHECK(rsmp::launch_fir_lockstep_rebase(ls->slot[sl].descs.as<rsmp::FirStreamDesc>(), ls->d_streams.as<LockstepStream>(),
                                                            ls->d_run_rs.as<rsmp::LsRunStream>(), key.in_offset, static_cast<uint32_t>(n), s));
            ls->rebased = false;
        }
        // (new tables: from the next plan on; this run was planned with the old ones, whose images nobody overwrites
        // before this run's kernels are through -- TableRefresher's guard event.  Behind the wait: the planner read
        // the stream table this may patch.)
        const uint64_t table_ops0 = ls->stat_table_ops;
        if (int rc = poll_drift(ls, s)) return rc;
        if (int rc = flush_tables(ls, s)) return rc;
        // A small batch's period is its PLANNER's (predict + chain + replay, ~150 us for 128 streams x 256 calls, against
        // ~130 us of bulk kernels), and with the commit on the caller's stream the planner's loop crossed queues twice per
        // run -- plan stream -> caller's stream (commit) -> plan stream (the next plan): two event hand-overs of ~12 us in a
        // 185 us period (profiles/r06/c4_run_timeline_128_ahead1.txt).  So the commit goes to the PLAN stream, right behind
        // the plan it commits, the next plan right behind it, and the caller's stream waits for the commit -- unless this
        // call put something on the caller's stream that the next plan must see (new class tables: rare).
        static const bool commit_knob = [] { const char* e = rsmp::knob("RSMP_LS_COMMIT_ON_PLAN"); return !e || atoi(e) != 0; }();
        static const bool commit_any_n = [] { const char* e = rsmp::knob("RSMP_LS_COMMIT_ON_PLAN"); return e && atoi(e) == 2; }();
        commit_on_q = commit_knob && (n < 256 || commit_any_n) && ls->ahead_q != nullptr && ls->stat_table_ops == table_ops0;
        hipStream_t cs = commit_on_q ? ls->ahead_q : s;
        if (commit_on_q && ls->drift_inflight)   // (a reading of the states on the caller's stream: in front of what changes them)
            RSMP_HIP_CHECK(rsmp::stream_wait_event(cs, ls->drift_ev));
        c.states = ls->d_states.as<FirMirrorState>();
        c.sp_states = ls->sp_states.as<FirMirrorState>();
        c.cursor = ls->d_cursor.as<uint64_t>();
        c.sp_cursor = ls->sp_cursor.as<uint64_t>();
        c.last_counts = ls->d_counts.as<uint64_t>();
        c.sp_last_counts = ls->sp_last.as<uint64_t>();
        c.status = ls->d_status.as<uint32_t>();
        c.sp_status = ls->sp_status.as<uint32_t>();
        c.n_streams = static_cast<uint32_t>(n);
        if (commit_on_q) RSMP_HIP_CHECK(rsmp::launch_fir_lockstep_commit(c, cs));
        else commit_pending = true;
        if (commit_on_q) {
            RSMP_HIP_CHECK(rsmp::event_record(ls->ev_commit, cs));
            RSMP_HIP_CHECK(rsmp::stream_wait_event(s, ls->ev_commit));
            ++ls->stat_commits_on_plan_stream;
        }
    } else {
        ls->ahead_waited = false;
        ls->rebased = false;   // (planned here, behind the new table)
        if (ls->ahead.valid) ++ls->stat_ahead_misses;
        if (int rc = drop_plan_ahead(ls, s)) return rc;   // (whatever the plan stream still does: finished before this stream goes on)
        if (int rc = poll_drift(ls, s)) return rc;
        if (int rc = flush_tables(ls, s)) return rc;
        RSMP_HIP_CHECK(rsmp::launch_fir_lockstep_plan(plan_args(key, false), s));
    }
    ls->ahead.valid = false;
    // Is the NEXT run worth planning ahead?  When this run repeats the previous one's shape (the loop of a caller that
    // feeds run after run): then the run after this one is guessed to repeat it again, its input offset moving on as it
    // did between the last two.  Its planner goes to the plan stream below and runs beside this run's bulk kernels (its
    // kernels use no LDS: the split kernel's ring of images fills a CU's).  What that buys depends on the batch: a small one
    // (128 streams: the bulk kernels leave most of a run's time to the chain's latency) runs 1.5x faster, config 4's 1024
    // streams the same -- the stagers of the split kernel and the chain compete for the same issue slots, and each is slowed
    // by what the other takes.  RSMP_LS_AHEAD=0 (debug): every run plans on the caller's stream.
    static const bool ahead_on = [] { const char* e = rsmp::knob("RSMP_LS_AHEAD"); return !e || atoi(e) != 0; }();
    bool repeat = ahead_on && ls->prev.valid && ls->prev.k == key.k && ls->prev.in_frames == key.in_frames && ls->prev.append == key.append;
    if (repeat) {
        if (int rc = pick_plan_stream(ls, s)) return rc;
        repeat = ls->plan_stream != nullptr;
    }
    // (the commit of a run planned ahead: K1 of the next run does it where that is launched on this stream, in front of the bulk
    // kernels -- a launch of 5 us less between two split launches, RSMP_LS_FUSE_COMMIT=0, debug: always a launch of its own)
    static const bool fuse_commit = [] { const char* e = rsmp::knob("RSMP_LS_FUSE_COMMIT"); return !e || atoi(e) != 0; }();
    if (commit_pending && !(fuse_commit && repeat && n >= 256)) {
        RSMP_HIP_CHECK(rsmp::launch_fir_lockstep_commit(c, s));
        commit_pending = false;
    }
    if (repeat) {   // (before this run's bulk kernels are launched: the planner starts as soon as the states are there)
        rsmp_fir_lockstep::RunKey nx = key;
        nx.in_offset = key.in_offset + (key.in_offset - ls->prev.in_offset);
        nx.parity = ls->hist_parity ^ 1u;
        nx.seq = ls->run_seq + 1;
        nx.slot = sl ^ 1;
        hipStream_t q = ls->plan_stream;
        // K1 of the next run -- eighteen 64-bit divisions per call, a kernel of CODE -- goes in front of this run's bulk
        // kernels on the caller's own stream: alone it takes 10-18 us, beside the split kernel (whose sixteen differently
        // programmed waves fill the instruction cache two CUs share) 72 us for a shard of 128 streams and 0.5 ms for the
        // whole batch, in front of a chain that takes 100-200.  It writes the next run's predictions and bitmaps only; the
        // chain and the replay follow on the plan stream, beside the bulk kernels.
        // (A small batch -- a shard of 128 streams -- is bound by the planner's own chain, not by the bulk kernels: there the
        // chain should not start beside the big single-round launch, where it runs 1.5x slower, and K1 on the plan stream
        // delays it by just about that launch: 0.96 against 1.21 us per step; from 256 streams up K1 in front wins, 1024
        // streams 3.56 -> 3.38 us per step.  profiles/r05/c4_shard_sweep*.txt.)
        const bool k1_in_front = n >= 256;
        // its buffers are free (long since); K1 on the very stream that computed on them is behind that anyway, and a big batch
        // does not even record the event (below) -- every event operation is a packet of its own that costs the queue 4-6 us
        // between two kernels (profiles/r06/c4_between_split_launches.txt).  Whoever needs it after all records it now: on the
        // stream that computed, behind everything enqueued there since.
        auto& nslot = ls->slot[nx.slot];
        if (nslot.used && !(k1_in_front && nslot.compute_stream == s)) {
            if (!nslot.compute_recorded) {
                RSMP_HIP_CHECK(rsmp::event_record(nslot.compute_done, nslot.compute_stream));
                nslot.compute_recorded = true;
            }
            RSMP_HIP_CHECK(rsmp::stream_wait_event(k1_in_front ? s : q, nslot.compute_done));
        }
        // (ev_ready below: completed by the K1 launch itself where that is on a stream of the caller's own -- not the legacy handle,
        // which an event must not carry, common.h)
        static const bool stop_ev = [] { const char* e = rsmp::knob("RSMP_LS_STOP_EVENT"); return !e || atoi(e) != 0; }();
        const bool k1_completes_ready = stop_ev && k1_in_front && s != reinterpret_cast<hipStream_t>(RSMP_STREAM_LEGACY);
        if (k1_in_front)
            RSMP_HIP_CHECK(rsmp::launch_fir_lockstep_plan(plan_args(nx, true), s, 1, commit_pending ? &c : nullptr, k1_completes_ready ? ls->ev_ready : nullptr));
        if (commit_on_q && !k1_in_front) {
            // (the states after this run are in place on the plan stream itself: nothing to wait for, unless the probe has
            // just moved the planner to the other candidate)
            if (q != ls->ahead_q) RSMP_HIP_CHECK(rsmp::stream_wait_event(q, ls->ev_commit));
        } else {
            if (!k1_completes_ready) RSMP_HIP_CHECK(rsmp::event_record(ls->ev_ready, s));   // the states after this run are in place (and the next run's predictions made)
            RSMP_HIP_CHECK(rsmp::stream_wait_event(q, ls->ev_ready));
        }
        RSMP_HIP_CHECK(rsmp::launch_fir_lockstep_plan(plan_args(nx, true), q, k1_in_front ? 2 : 3));
        // The split kernel's item tables of that run (one record per item: which frames, which outputs -- from the descriptors the
        // planner has just written) are built here as well, behind the plan on the plan stream: in front of the run's own kernels
        // the table launch was 6 us between two split launches.  (RSMP_LS_ITEMS_AHEAD=0, debug: built where they are used.)
        static const bool items_ahead = [] { const char* e = rsmp::knob("RSMP_LS_ITEMS_AHEAD"); return !e || atoi(e) != 0; }();
        ls->slot[nx.slot].items_valid = false;
        if (items_ahead) {
            std::vector<rsmp::SplitJob> jobs;
            const rsmp::FirStreamDesc* nd = ls->slot[nx.slot].descs.as<rsmp::FirStreamDesc>();
            for (const auto& g : ls->run_groups) {
                if (g.geo.mfma != 3) continue;
                const uint64_t n_out_max = static_cast<uint64_t>(k) * g.max_out_step;
                jobs.push_back(rsmp::SplitJob{nd + g.first, static_cast<uint32_t>(g.count), &g.geo,
                                              static_cast<uint32_t>((n_out_max / g.geo.b + 1) / g.geo.pw + 1), rsmp::NfArgs{}});
            }
            const size_t bytes = rsmp::fir_split_multi_item_words(jobs.data(), jobs.size()) * sizeof(uint32_t);
            auto& ns = ls->slot[nx.slot];
            bool room = bytes <= ns.items.capacity();
            if (!room && bytes != 0) {   // (once per shape of run: nothing in flight reads a table that is about to be built for the first time)
                RSMP_HIP_CHECK(hipStreamSynchronize(q));
                RSMP_HIP_CHECK(hipStreamSynchronize(s));
                RSMP_HIP_CHECK(ns.items.reserve(bytes + bytes / 2));
                room = true;
            }
            if (room && bytes != 0) {
                RSMP_HIP_CHECK(rsmp::launch_fir_split_multi(jobs.data(), jobs.size(), q, 0, ns.items.as<uint32_t>(), q));
                ns.items_seq = nx.seq;
                ns.items_ops = ls->stat_table_ops;
                ns.items_valid = true;
            }
        }
        RSMP_HIP_CHECK(rsmp::event_record(ls->plan_done, q));
        ls->ahead_q = q;
        ls->ahead = nx;
        ls->ahead_inflight = true;
    }
    const rsmp::FirStreamDesc* d_descs = ls->slot[sl].descs.as<rsmp::FirStreamDesc>();
    // The bulk kernels over the run's descriptors.  The rate pairs the split kernel takes go into launches shared by as
    // many of them as have the same kernel build (launch_fir_split_multi), with one item-table launch in front and one
    // repair launch behind for all of them: six rate pairs one after the other were 18 launches, most of a run of 16 calls.
    // (Measured and dropped before that: the rate pairs' launches side by side on streams of their own, each with its share
    // of the compute units -- a run of 16 calls was then bound by the host's ~40 stream / event / launch calls.)
    const size_t n_groups = ls->run_groups.size();
    uint32_t nf_off = 0, max_tail_values = 0;
    std::vector<rsmp::SplitJob> split_jobs;
    std::vector<rsmp::RepairJob> repair_jobs;
    for (size_t gi = 0; gi < n_groups; ++gi) {
        const auto& g = ls->run_groups[gi];
        const uint64_t n_out_max = static_cast<uint64_t>(k) * g.max_out_step;
        const uint32_t max_blocks = static_cast<uint32_t>((n_out_max / g.geo.b + 1) / g.geo.pw + 1);
        rsmp::NfArgs nf;
        nf.chunks = static_cast<uint32_t>(n_out_max >> rsmp::kNfChunkShift) + 1;
        nf.words = ls->d_run_nf.as<uint32_t>() + nf_off;
        if (++ls->run_nf_tag == 0) ls->run_nf_tag = 1;
        nf.tag = ls->run_nf_tag;
        nf_off += 1 + static_cast<uint32_t>((g.count * static_cast<uint64_t>(nf.chunks) + 31) / 32);
        if (g.geo.mfma == 3) {
            split_jobs.push_back(rsmp::SplitJob{d_descs + g.first, static_cast<uint32_t>(g.count), &g.geo, max_blocks, nf});
        } else {
            RSMP_HIP_CHECK(rsmp::launch_fir_periodic(d_descs + g.first, static_cast<uint32_t>(g.count), g.geo, max_blocks,
                                                     ls->d_run_work.as<unsigned long long>() + gi, nf, s));
        }
        repair_jobs.push_back(rsmp::RepairJob{d_descs + g.first, static_cast<uint32_t>(g.count), nf});
        const rsmp_fir* r0 = ls->rs[ls->order[g.first]];
        max_tail_values = std::max<uint32_t>(max_tail_values, static_cast<uint32_t>((r0->taps + 8) * r0->channels));
    }
    // A workgroup of the split kernel fills its CU (512 vector registers per SIMD lane, 136 KB of LDS): the planner of the
    // NEXT run, on the plan stream, does not run beside it but behind it, CU by CU as the launch drains -- for a small
    // batch, whose period is the planner's, that was the bulk launch's length added to every run (K1 "72 us beside the
    // split kernel, 10 alone").  A batch of fewer than 256 streams planned ahead leaves the planner a wave's worth of CUs
    // per stream (one chain wave per stream, four SIMDs per CU); its bulk kernels lose an eighth of the chip they do not
    // need.  RSMP_LS_RESERVE (debug): the number of CUs, 0 = none.
    uint32_t reserve = 0;
    if (repeat && n < 256) {
        static const int knob = [] { const char* e = rsmp::knob("RSMP_LS_RESERVE"); return e ? atoi(e) : -1; }();
        // (the planner's packed workgroups take lockstep_plan_cus(n) CUs -- 32 for 128 streams --, + 4 for its one-wave kernels; the
        // replay behind the chain a wave per chunk: a bulk launch of 64 streams x 4096 calls has 1024 of them, 64 CUs' worth, and with
        // 20 CUs left to it every fourth launch took 0.78 instead of 0.56 ms -- profiles/r06/bulk_distinct_reserve.txt)
        reserve = knob >= 0 ? static_cast<uint32_t>(knob)
                            : std::min<uint32_t>(64u, std::max(rsmp::lockstep_plan_cus(n), rsmp::lockstep_replay_cus(n, k)) + 4u);
    }
    // (the item tables: built on the plan stream behind this run's plan, if it was planned ahead and no table has changed since)
    const auto& cslot = ls->slot[sl];
    const uint32_t* items_prebuilt = plan_taken_over && cslot.items_valid && cslot.items_seq == key.seq && cslot.items_ops == ls->stat_table_ops
                                         ? cslot.items.as<uint32_t>() : nullptr;
    if (!split_jobs.empty()) RSMP_HIP_CHECK(rsmp::launch_fir_split_multi(split_jobs.data(), split_jobs.size(), s, reserve, items_prebuilt));
    // A big batch's planner is through long before its split launch (profiles/r06/c4_run_timeline_1024_ahead1.txt): the wait for
    // it goes HERE, behind the split launch, where the queue's processor handles the cross-queue dependency while the launch
    // drains -- in front of the next run's first kernel it was 11 us of idle chip between two split launches.
    // (RSMP_LS_EARLY_WAIT=0, debug: the wait where the plan is taken over.)
    static const bool early_wait = [] { const char* e = rsmp::knob("RSMP_LS_EARLY_WAIT"); return !e || atoi(e) != 0; }();
    if (early_wait && repeat && n >= 256 && ls->ahead_inflight) {
        RSMP_HIP_CHECK(rsmp::stream_wait_event(s, ls->plan_done));
        ls->ahead_waited = true;
        ls->ahead_waited_on = s;
    }
    // (the repair launch copies the streams' tails as well: one launch and its gap less per run)
    // (the slot's "computed" event: a small batch's plan stream waits for it before it plans into the slot again; a big batch's K1 runs
    // on this very stream and needs none -- RSMP_LS_LAZY_DONE=0, debug: recorded always)
    static const bool lazy_done = [] { const char* e = rsmp::knob("RSMP_LS_LAZY_DONE"); return !e || atoi(e) != 0; }();
    static const bool stop_ev2 = [] { const char* e = rsmp::knob("RSMP_LS_STOP_EVENT"); return !e || atoi(e) != 0; }();
    const bool record_done = !(lazy_done && n >= 256);
    bool done_attached = false;
    const bool attach = record_done && stop_ev2 && s != reinterpret_cast<hipStream_t>(RSMP_STREAM_LEGACY);
    RSMP_HIP_CHECK(rsmp::launch_fir_repair_multi(repair_jobs.data(), repair_jobs.size(), s, d_descs, static_cast<uint32_t>(n), max_tail_values,
                                                 attach ? ls->slot[sl].compute_done : nullptr, &done_attached));
    if (record_done && !done_attached) RSMP_HIP_CHECK(rsmp::event_record(ls->slot[sl].compute_done, s));
    ls->slot[sl].compute_recorded = record_done;
    ls->slot[sl].compute_stream = s;
    ls->slot[sl].used = true;
    if (ls->profiling) {
        RSMP_HIP_CHECK(rsmp::event_record(ls->prof_stop[ls->prof_count % rsmp_fir_lockstep::kProfRing], s));
        ++ls->prof_count;
    }
    ls->hist_parity ^= 1u;
    ls->step += k;
    ++ls->epoch;   // plans the one-call kernel made ahead belong to the states before the run
    ++ls->run_seq;
    ls->run_counts_k = k;
    ls->run_planned = true;
    ls->last_stream = s;
    ls->last_slot = sl;
    ls->next_slot = sl ^ 1;
    ls->prev = key;
    return request_drift(ls, s, static_cast<uint64_t>(k) * in_frames);
}

extern "C" int rsmp_fir_lockstep_run_counts(rsmp_fir_lockstep* ls, size_t* consumed, size_t* produced, size_t max_steps) {
    if (!ls) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_run_counts: null batch");
    if (ls->run_counts_k == 0)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_run_counts: the last launch was not a run of several calls");
    DeviceGuard guard(ls->device);
    const size_t n = ls->rs.size(), k = std::min(ls->run_counts_k, max_steps);
    if (ls->last_stream) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    ls->h_run_counts.resize(2 * n * k);
    RSMP_HIP_CHECK(hipMemcpy(ls->h_run_counts.data(), ls->slot[ls->last_slot].counts.get(), 2 * n * k * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n * k; ++i) {
        if (consumed) consumed[i] = ls->h_run_counts[2 * i];
        if (produced) produced[i] = ls->h_run_counts[2 * i + 1];
    }
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_table_rebinds(const rsmp_fir_lockstep* ls, size_t* rebinds) {
    if (!ls || !rebinds) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_table_rebinds: null argument");
    *rebinds = ls->table_rebinds;
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_run_slow_calls(rsmp_fir_lockstep* ls, size_t* slow_calls) {
    if (!ls || !slow_calls) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_run_slow_calls: null argument");
    *slow_calls = 0;
    if (ls->run_counts_k <= 1 || ls->run_state <= 0 || !ls->run_planned) return RSMP_OK;   // (a loop of steps)
    DeviceGuard guard(ls->device);
    if (ls->last_stream) RSMP_HIP_CHECK(hipStreamSynchronize(ls->last_stream));
    struct Rec { double pos, drift; uint32_t flags, pad; };
    std::vector<Rec> h(ls->rs.size() * ls->run_counts_k);
    RSMP_HIP_CHECK(hipMemcpy(h.data(), ls->slot[ls->last_slot].recs.get(), h.size() * sizeof(Rec), hipMemcpyDeviceToHost));
    for (const Rec& r : h) *slow_calls += r.flags & 1u;
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_stats(const rsmp_fir_lockstep* ls, uint64_t* out, size_t n) {
    if (!ls || !out) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_stats: null argument");
    const uint64_t v[RSMP_LS_STAT_COUNT] = {ls->table_rebinds, ls->stat_ahead_hits, ls->stat_ahead_misses, ls->stat_late_polls,
                                            ls->stat_table_waits, ls->stat_probes, ls->plan_stream ? 1u : 0u, ls->classes.size()};
    for (size_t i = 0; i < n && i < RSMP_LS_STAT_COUNT; ++i) out[i] = v[i];
    return RSMP_OK;
}

extern "C" int rsmp_fir_lockstep_set_drift_policy(rsmp_fir_lockstep* ls, double tolerance_frames, size_t check_frames) {
    if (!ls || !(tolerance_frames >= 2.0 * kLsDriftQuantum) || tolerance_frames > 1e-6 || check_frames == 0)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_set_drift_policy: tolerance in [2e-8, 1e-6] frames, check_frames > 0");
    ls->drift_tolerance = tolerance_frames;
    ls->drift_check_frames = check_frames;
    return RSMP_OK;
}

// A whole buffer per stream, fed as the reference's driver loop feeds it (resample/src/main.rs:226-254: calls of
// `chunk_frames` frames until the input is used up, the last one shorter): floor(total / chunk) equal calls through the
// device planner (rsmp_fir_lockstep_run) and, where total is no multiple of chunk, one more call of the remaining frames
// (rsmp_fir_lockstep_step), their outputs appended behind the run's.  Whatever states the streams are in, nothing is
// planned on the host and nothing waits: the calls' counts are device data (rsmp_fir_lockstep_run_counts for the equal
// calls, rsmp_fir_lockstep_counts for the last one), the streams' states stay on the device (rsmp_fir_lockstep_sync).
// This is the bulk entry point for BATCHES IN DISTINCT STATES: rsmp_fir_batch_resample_bulk_device
// replays every stream's control flow on the host's planning workers -- 0.7-1.0 ms for 64 streams x 4096 calls on a
// free 256-thread host, 19-26 ms where the workers have to be woken on a busy one.
extern "C" int rsmp_fir_lockstep_run_bulk(rsmp_fir_lockstep* ls, size_t total_frames, size_t chunk_frames, size_t in_offset_frames,
                                          int append, void* stream) {
    if (!ls || chunk_frames == 0) return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_fir_lockstep_run_bulk: null batch or zero chunk");
    if (chunk_frames > ls->step_frames)
        return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE, "lock-step batch: calls of %zu frames, created for %u per step", chunk_frames,
                          ls->step_frames);
    // The driver loop this stands for offers a call's remainder again when the call accepts less than its offer (resample/src/main.rs:
    // 226-254); a run's calls read their input at fixed offsets.  So the calls must be ones every stream accepts whole: a stream buffers at
    // most INPUT_CAPACITY = 4096 frames (resampler_fir.rs:18, :524-528) and keeps up to taps + 1 of them between calls.
    size_t max_taps = 0;
    for (const rsmp_fir* r : ls->rs) max_taps = std::max<size_t>(max_taps, r->taps);
    if (chunk_frames + max_taps + 8 > rsmp::kMirrorInputCapacity)
        return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE,
                          "rsmp_fir_lockstep_run_bulk: calls of %zu frames are not accepted whole by a stream of %zu taps (at most %zu)", chunk_frames,
                          max_taps, static_cast<size_t>(rsmp::kMirrorInputCapacity) - max_taps - 8);
    const size_t k = total_frames / chunk_frames, tail = total_frames - k * chunk_frames;
    if (k > 0)
        if (int rc = rsmp_fir_lockstep_run(ls, k, chunk_frames, in_offset_frames, append, stream)) return rc;
    if (tail > 0) {
        const size_t run_k = ls->run_counts_k;
        if (int rc = rsmp_fir_lockstep_step(ls, tail, in_offset_frames + k * chunk_frames, nullptr, (k > 0 || append) ? 1 : 0, stream)) return rc;
        if (k > 0) ls->run_counts_k = run_k;   // (the run's counts stay readable: the last call's are rsmp_fir_lockstep_counts')
    }
    return RSMP_OK;
}
