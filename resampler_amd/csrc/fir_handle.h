// fir_handle.h -- the ResamplerFir handle behind the C ABI (shared by fir_api.cpp and
// fir_lockstep_api.cpp; not part of the public interface).
#pragma once

#include <hip/hip_runtime.h>

#include <memory>
#include <vector>

#include "device_util.h"
#include "fir_periodic.h"
#include "fir_plan.h"

struct rsmp_fir {
    int device = 0;
    size_t channels = 0;
    size_t taps = 0;
    int attenuation = 0;
    uint32_t in_hz = 0, out_hz = 0;
    int kernel_mode = RSMP_FIR_KERNEL_AUTO;
    rsmp::FirMirror mirror;
    std::shared_ptr<const std::vector<float>> table;
    float* d_coeffs = nullptr;
    float* d_hist[2] = {nullptr, nullptr};  // kInputCapacity * channels floats each
    int cur = 0;
    hipStream_t stream = nullptr;
    // Launch plans travel through a small ring of pinned buffers so the host can enqueue several
    // launches ahead of the GPU (a slot is reused only after its upload has left host memory).
    static constexpr int kPlanSlots = 4;
    hipEvent_t plan_copied[kPlanSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool plan_pending[kPlanSlots] = {false, false, false, false};
    int plan_slot = 0;
    // launch workspace (descs + runs + tile index), host-pinned and device
    rsmp::PinnedBuffer h_plan[kPlanSlots];
    rsmp::DeviceBuffer d_plan[kPlanSlots];
    std::vector<char> plan_image[kPlanSlots];   // what each slot's HBM buffer currently holds
    std::vector<char> plan_scratch;
    // staging for the host-pointer entry points
    rsmp::DeviceBuffer d_stage_in, d_stage_out;
    rsmp::PinnedBuffer h_stage_in, h_stage_out;   // small calls: the kernels read / write mapped host memory, no copy engine
    rsmp::PeriodicState periodic;
    bool last_periodic = false;   // the handle's last launch went through a periodic kernel
    unsigned long long* d_work_counter = nullptr;   // periodic kernel's item queue (leader only), zero between launches
    rsmp::DeviceBuffer d_nf;                        // non-finite marks of the periodic launches (leader only)
    uint32_t nf_tag = 0;
    hipStream_t last_stream = nullptr;              // the stream of the handle's most recent launch
    bool last_stream_valid = false;
    // optional timing of the main convolution launch(es) (rsmp_fir_set_profiling)
    bool profiling = false;
    // ring of event pairs: launches made while profiling is on are timed without any host sync
    static constexpr int kProfRing = 64;
    hipEvent_t prof_start[kProfRing] = {}, prof_stop[kProfRing] = {};
    size_t prof_count = 0;

    rsmp_fir(uint32_t i, uint32_t o, size_t t) : mirror(i, o, t) {}
};

