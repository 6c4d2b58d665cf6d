// fir_table_refresher.cpp -- the worker thread that builds and uploads replacement class tables (fir_table_refresher.h).
#include "fir_table_refresher.h"

#include <chrono>
#include <cstdio>
#include <cstring>

#include "common.h"

namespace rsmp {

TableRefresher::TableRefresher(int device) : device_(device) {}


TableRefresher::~TableRefresher() {
    {
        std::lock_guard<std::mutex> lock(mu_);
        stop_ = true;
    }
    cv_work_.notify_all();
    if (worker_.joinable()) worker_.join();
    for (auto& t : tables_) {
        for (char*& p : t->d_buf)
            if (p) { (void)hipFree(p); p = nullptr; }
        if (t->guard) (void)hipEventDestroy(t->guard);
    }
}

TableRefresher::Table* TableRefresher::add_table(const PeriodicGeometry& geo, std::shared_ptr<const std::vector<float>> coeffs) {
    std::unique_ptr<Table> t(new Table);
    t->geo = geo;
    t->coeffs = std::move(coeffs);
    if (hipEventCreateWithFlags(&t->guard, hipEventDisableTiming) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu_);
    tables_.push_back(std::move(t));
    return tables_.back().get();
}

int TableRefresher::record_guard(Table* t, hipStream_t s) {
    RSMP_HIP_CHECK(rsmp::event_record(t->guard, s));
    return RSMP_OK;
}

int TableRefresher::request(Table* t, double drift) {
    t->want_drift = drift;
    t->state.store(kRequested, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lock(mu_);   // (held by the worker only while it takes an entry off the queue)
        queue_.push_back(t);
        // (the worker exists from the batch's first request on: a batch whose streams never age that far -- most -- has no
        // thread; creating it costs the call that asks first ~0.1 ms, once)
        if (!worker_.joinable()) worker_ = std::thread([this] { loop(); });
    }
    cv_work_.notify_one();
    return RSMP_OK;
}

ClassTable TableRefresher::take(Table* t) {
    ClassTable ct = t->ready;
    t->next_buf ^= 1;
    t->state.store(kIdle, std::memory_order_release);
    return ct;
}

void TableRefresher::wait(Table* t) {
    std::unique_lock<std::mutex> lock(mu_);
    cv_done_.wait(lock, [&] { return t->state.load(std::memory_order_acquire) != kRequested; });
}

int TableRefresher::refresh(Table* t) {
    const HostClassTable host = build_class_table(*t->coeffs, t->geo, t->want_drift);
    const size_t cb = host.coef.size() * sizeof(float), wb = host.wrap_coef.size() * sizeof(float),
                 mb = host.meta.size() * sizeof(TileMeta);
    const size_t total = cb + wb + mb;
    if (!t->d_buf[0] || !t->d_buf[1]) {   // the table's first refresh: both images (the sizes depend on the geometry alone)
        char* a = nullptr;
        char* b = nullptr;   // (committed only as a pair: a table is never left with one image)
        RSMP_HIP_CHECK(hipMalloc(&a, total));
        if (hipMalloc(&b, total) != hipSuccess) {
            (void)hipFree(a);
            return rsmp::fail(RSMP_ERR_HIP, "class table refresher: no device memory for a table's second image");
        }
        for (char* p : t->d_buf) if (p) (void)hipFree(p);
        t->d_buf[0] = a;
        t->d_buf[1] = b;
        t->coef_bytes = cb;
        t->wrap_bytes = wb;
        t->meta_bytes = mb;
    } else if (cb != t->coef_bytes || wb != t->wrap_bytes || mb != t->meta_bytes) {
        return rsmp::fail(RSMP_ERR_HIP, "class table refresher: a table's size changed between refreshes");
    }
    if (total > h_stage_cap_) {
        if (h_stage_) (void)hipHostFree(h_stage_);
        h_stage_ = nullptr;
        h_stage_cap_ = 0;
        RSMP_HIP_CHECK(hipHostMalloc(&h_stage_, total + total / 2, hipHostMallocDefault));
        h_stage_cap_ = total + total / 2;
    }
    char* h = static_cast<char*>(h_stage_);
    std::memcpy(h, host.coef.data(), cb);
    std::memcpy(h + cb, host.wrap_coef.data(), wb);
    std::memcpy(h + cb + wb, host.meta.data(), mb);
    char* d = t->d_buf[t->next_buf];
    // the image was bound until the last replacement: what had been enqueued by then may still read it (record_guard)
    RSMP_HIP_CHECK(rsmp::stream_wait_event(copy_stream_, t->guard));
    RSMP_HIP_CHECK(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, copy_stream_));
    RSMP_HIP_CHECK(hipStreamSynchronize(copy_stream_));
    t->ready.d_coef = reinterpret_cast<const float*>(d);
    t->ready.d_wrap_coef = reinterpret_cast<const float*>(d + cb);
    t->ready.d_meta = reinterpret_cast<const TileMeta*>(d + cb + wb);
    t->ready.hold.reset();
    t->ready_drift = t->want_drift;
    return RSMP_OK;
}

void TableRefresher::loop() {
    bool device_ready = false;
    for (;;) {
        Table* t = nullptr;
        {
            std::unique_lock<std::mutex> lock(mu_);
            cv_work_.wait(lock, [&] { return stop_ || !queue_.empty(); });
            if (stop_) break;
            t = queue_.front();
            queue_.pop_front();
        }
        int rc = RSMP_OK;
        if (!device_ready) {   // (the thread's first request: its device and its copy stream)
            if (hipSetDevice(device_) != hipSuccess ||
                hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking) != hipSuccess)
                rc = RSMP_ERR_HIP;
            else
                device_ready = true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        if (rc == RSMP_OK) rc = refresh(t);
        static const bool verbose = rsmp::knob("RSMP_FIR_VERBOSE") != nullptr;
        if (verbose)
            fprintf(stderr, "[rsmp] table refresher: a=%u b=%u drift %.3g -> image %d (%s) in %.3f ms\n", t->geo.a, t->geo.b, t->want_drift,
                    t->next_buf, rc == RSMP_OK ? "ready" : "failed",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        {
            std::lock_guard<std::mutex> lock(mu_);
            t->state.store(rc == RSMP_OK ? kReady : kFailed, std::memory_order_release);
        }
        cv_done_.notify_all();
    }
    if (copy_stream_) {
        (void)hipStreamSynchronize(copy_stream_);
        (void)hipStreamDestroy(copy_stream_);
    }
    if (h_stage_) (void)hipHostFree(h_stage_);
}

}  // namespace rsmp
