// device_util.h -- small RAII helpers for HBM / pinned-host workspaces (grow-only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

namespace rsmp {

// Grow-only device buffer.  reserve() may reallocate (contents are NOT preserved).
class DeviceBuffer {
public:
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { if (ptr_) (void)hipFree(ptr_); }

    hipError_t reserve(size_t bytes) {
        if (bytes <= cap_) return hipSuccess;
        if (ptr_) {
            hipError_t e = hipFree(ptr_);  // synchronises with outstanding work on the buffer
            ptr_ = nullptr;
            cap_ = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = bytes + bytes / 2;
        if (want < 4096) want = 4096;
        hipError_t e = hipMalloc(&ptr_, want);
        if (e != hipSuccess) { ptr_ = nullptr; return e; }
        cap_ = want;
        return hipSuccess;
    }
    void* get() const { return ptr_; }
    template <class T> T* as() const { return static_cast<T*>(ptr_); }
    size_t capacity() const { return cap_; }

private:
    void* ptr_ = nullptr;
    size_t cap_ = 0;
};

// Grow-only pinned host buffer.
class PinnedBuffer {
public:
    PinnedBuffer() = default;
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    ~PinnedBuffer() { if (ptr_) (void)hipHostFree(ptr_); }

    // coherent = false: cached on the device like its own memory, visible to the other side at kernel
    // boundaries (staging of small host-slice calls: the kernel reads every sample many times)
    hipError_t reserve(size_t bytes, bool coherent = true) {
        if (bytes <= cap_) return hipSuccess;
        if (ptr_) {
            hipError_t e = hipHostFree(ptr_);
            ptr_ = nullptr;
            cap_ = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = bytes + bytes / 2;
        if (want < 4096) want = 4096;
        // mapped + coherent: small launch plans are read by the kernels straight from this memory
        hipError_t e = hipHostMalloc(&ptr_, want, hipHostMallocMapped | (coherent ? hipHostMallocCoherent : hipHostMallocNonCoherent));
        if (e != hipSuccess) { ptr_ = nullptr; return e; }
        cap_ = want;
        return hipSuccess;
    }
    void* get() const { return ptr_; }
    template <class T> T* as() const { return static_cast<T*>(ptr_); }
    size_t capacity() const { return cap_; }

private:
    void* ptr_ = nullptr;
    size_t cap_ = 0;
};

// Restores the caller's current device on scope exit.
class DeviceGuard {
public:
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        if (prev_ != device) (void)hipSetDevice(device);
        else prev_ = -1;
    }
    ~DeviceGuard() { if (prev_ >= 0) (void)hipSetDevice(prev_); }

private:
    int prev_ = -1;
};

}  // namespace rsmp
