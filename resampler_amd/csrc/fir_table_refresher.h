// fir_table_refresher.h -- replacement class tables for a lock-step batch, prepared OFF the launch path.
//
// The class tables of a lock-step batch follow the streams' f64 drift (fir_lockstep_api.cpp, DriftClass; the
// reference's `position += ratio`, src/resampler_fir.rs:589, leaves the exact rational position by ~1e-14 of a frame
// per output).  A replacement used to be made inside rsmp_fir_lockstep_run / _step: a process-wide mutex, a host build
// when the image was not ready, hipMalloc and three synchronous hipMemcpy -- each of which waits for the kernels in
// flight -- so a run that crossed a tolerance held the enqueueing thread for milliseconds with the GPU idle behind it
// (0.8 ms of host time per run of config 4 on the driver's box).
//
// Now every refreshable table owns TWO device images, used alternately, and a worker thread per batch does everything
// slow: the host arithmetic (build_class_table, 0.35-0.7 ms), the first refresh's allocations, the upload on a copy
// stream of its own (behind an event recorded when the image was last unbound: what may still read its old contents),
// the wait for it.  The launch path asks, and later finds an atomic flag set and swaps pointers (+ one event record).  Nothing in
// it allocates, copies synchronously, takes a lock another thread holds for long, or waits.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "fir_periodic.h"

namespace rsmp {

class TableRefresher {
public:
    enum { kIdle = 0, kRequested = 1, kReady = 2, kFailed = 3 };
    struct Table {
        PeriodicGeometry geo;
        std::shared_ptr<const std::vector<float>> coeffs;   // the [1024][taps] polyphase table the rows are mixed from
        char* d_buf[2] = {nullptr, nullptr};                // owned device images (allocated by the worker when first needed)
        size_t coef_bytes = 0, wrap_bytes = 0, meta_bytes = 0;
        int next_buf = 0;                                   // the image the next refresh fills; the other one may be bound
        double want_drift = 0.0;                            // request: written before `state` goes to kRequested
        hipEvent_t guard = nullptr;                         // the work that may still read d_buf[next_buf] (record_guard)
        ClassTable ready;                                   // result: pointers into d_buf[next_buf] (no `hold`: owned here)
        double ready_drift = 0.0;
        std::atomic<int> state{kIdle};
    };

    explicit TableRefresher(int device);
    // Joins the worker and frees the images.  The caller has waited for every kernel that reads them.
    ~TableRefresher();
    TableRefresher(const TableRefresher&) = delete;
    TableRefresher& operator=(const TableRefresher&) = delete;

    // Not on the launch path (batch creation / first run): registers a table; no device work.
    Table* add_table(const PeriodicGeometry& geo, std::shared_ptr<const std::vector<float>> coeffs);
    // Launch path: have `t` rebuilt for `drift`; returns at once.  `t` must be kIdle.
    int request(Table* t, double drift);
    // Launch path: a kReady result is taken over -- `t` is idle again and its other image, the one bound until now, is the
    // next to be filled.  The caller then enqueues whatever was planned with the old image and calls record_guard: the
    // next refresh overwrites the old image behind that point of stream `s` (recorded at the REPLACEMENT, not at the
    // next request: the host runs many launches ahead of the device, and an event recorded at request time was reached
    // 20 ms later -- by when the tables it held back were due).
    ClassTable take(Table* t);
    int record_guard(Table* t, hipStream_t s);
    // A result nobody wants (the states changed meanwhile): `t` is idle again, the same image is the next to be filled.
    void discard(Table* t) { t->state.store(kIdle, std::memory_order_release); }
    // Blocks until `t` has left kRequested (only where a batch has run 3x past a tolerance without its tables: never
    // observed; counted by the caller).
    void wait(Table* t);

private:
    void loop();
    int refresh(Table* t);
    int device_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::deque<Table*> queue_;
    std::vector<std::unique_ptr<Table>> tables_;
    bool stop_ = false;
    std::thread worker_;
    // worker-only state
    hipStream_t copy_stream_ = nullptr;
    void* h_stage_ = nullptr;
    size_t h_stage_cap_ = 0;
};

}  // namespace rsmp
