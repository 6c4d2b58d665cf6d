// cli_kernels.hip -- the reference CLI's helpers around the hot path, on the GPU:
//   * the two comparison interpolators, linear and 4-point 3rd-order Hermite
//     (resample/src/interpolation_resampler.rs:41-126): one lane per output value, position in f64 exactly
//     as the reference computes it (output index / ratio), arithmetic in the reference's order;
//   * WAV sample conversion (resample/src/main.rs:128-156): 16 / 24 / 32-bit little-endian PCM -> f32
//     (`s as f32 / (1 << (bits - 1)) as f32`) with mono duplicated to stereo, so a decoded file goes from
//     its PCM bytes in HBM to the interleaved f32 frames the resamplers take in one pass.
// Both are HBM-bound elementwise kernels: coalesced loads / stores, nothing to stage.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/resampler_amd.h"
#include "common.h"
#include "device_util.h"

namespace {

constexpr int kThreads = 256;

template <bool HERMITE>
__global__ __launch_bounds__(kThreads) void interp_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          uint32_t channels, uint64_t input_frames,
                                                          uint64_t output_frames, double ratio) {
    const uint64_t e = blockIdx.x * static_cast<uint64_t>(kThreads) + threadIdx.x;
    if (e >= output_frames * channels) return;
    const uint64_t o = e / channels;
    const uint32_t ch = static_cast<uint32_t>(e - o * channels);
    const double input_pos = static_cast<double>(o) / ratio;                  // :51, :93
    const uint64_t idx = static_cast<uint64_t>(floor(input_pos));
    const float frac = static_cast<float>(input_pos - static_cast<double>(idx));
    const uint64_t last = input_frames - 1;
    if constexpr (!HERMITE) {
        if (idx >= last) {                                                    // :55-62
            out[e] = in[last * channels + ch];
            return;
        }
        const float s0 = in[idx * channels + ch], s1 = in[(idx + 1) * channels + ch];
        out[e] = s0 * (1.0f - frac) + s1 * frac;                              // :71
    } else {
        const uint64_t i_prev = idx > 0 ? idx - 1 : 0;                        // :99-106
        const uint64_t i_cur = idx < last ? idx : last;
        const uint64_t i_n1 = idx + 1 < last ? idx + 1 : last;
        const uint64_t i_n2 = idx + 2 < last ? idx + 2 : last;
        const float previous = in[i_prev * channels + ch], current = in[i_cur * channels + ch];
        const float next_1 = in[i_n1 * channels + ch], next_2 = in[i_n2 * channels + ch];
        const float c0 = current;                                             // :114-117
        const float c1 = (next_1 - previous) * 0.5f;
        const float c2 = previous - current * 2.5f + next_1 * 2.0f - next_2 * 0.5f;
        const float c3 = (next_2 - previous) * 0.5f + (current - next_1) * 1.5f;
        out[e] = ((c3 * frac + c2) * frac + c1) * frac + c0;                   // :119
    }
}

template <int BITS>
__global__ __launch_bounds__(kThreads) void pcm_kernel(const uint8_t* __restrict__ pcm, float* __restrict__ out,
                                                       uint64_t n_samples, uint32_t mono) {
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(kThreads) + threadIdx.x;
    if (i >= n_samples) return;
    int32_t s;
    if constexpr (BITS == 16) {
        s = reinterpret_cast<const int16_t*>(pcm)[i];
    } else if constexpr (BITS == 24) {
        const uint32_t u = static_cast<uint32_t>(pcm[3 * i]) | (static_cast<uint32_t>(pcm[3 * i + 1]) << 8) |
                           (static_cast<uint32_t>(pcm[3 * i + 2]) << 16);
        s = static_cast<int32_t>(u << 8) >> 8;
    } else {
        s = reinterpret_cast<const int32_t*>(pcm)[i];
    }
    // main.rs:131 `(1 << (bits_per_sample - 1)) as f32`: the literal is an i32, so 32-bit files divide by
    // 1i32 << 31 = i32::MIN = -2^31 -- the reference decodes them with inverted polarity, and so does this.
    const float max_value = BITS == 32 ? -2147483648.0f : static_cast<float>(1 << (BITS - 1));
    const float v = static_cast<float>(s) / max_value;
    if (mono) {                                                               // main.rs:141-146
        reinterpret_cast<float2*>(out)[i] = make_float2(v, v);
    } else {
        out[i] = v;
    }
}

int check_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return rsmp::fail(RSMP_ERR_NO_DEVICE, "no HIP device (this engine has no CPU path)");
    return RSMP_OK;
}

}  // namespace

extern "C" size_t rsmp_interp_output_len(size_t channels, uint32_t in_hz, uint32_t out_hz, size_t in_len) {
    if (channels == 0 || in_hz == 0 || out_hz == 0) return 0;
    const double ratio = static_cast<double>(out_hz) / static_cast<double>(in_hz);
    return static_cast<size_t>(std::ceil(static_cast<double>(in_len / channels) * ratio)) * channels;
}

extern "C" int rsmp_interp_resample_device(int mode, size_t channels, uint32_t in_hz, uint32_t out_hz,
                                           const float* d_in, size_t in_len, float* d_out, size_t out_cap,
                                           size_t* produced, void* stream) {
    if (int rc = check_device()) return rc;
    if ((mode != RSMP_INTERP_LINEAR && mode != RSMP_INTERP_HERMITE) || channels == 0 || in_hz == 0 || out_hz == 0)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_interp_resample: invalid mode / channels / rate");
    if (in_len % channels != 0) return rsmp::fail(RSMP_ERR_INVALID_INPUT_BUFFER_SIZE, "Input buffer size is invalid");
    const size_t need = rsmp_interp_output_len(channels, in_hz, out_hz, in_len);
    if (need > out_cap)
        return rsmp::fail(RSMP_ERR_CAPACITY, "interpolator output needs %zu values, room for %zu", need, out_cap);
    if (produced) *produced = need;
    if (need == 0) return RSMP_OK;
    const double ratio = static_cast<double>(out_hz) / static_cast<double>(in_hz);
    const uint64_t in_frames = in_len / channels, out_frames = need / channels;
    const dim3 grid(static_cast<uint32_t>((need + kThreads - 1) / kThreads));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mode == RSMP_INTERP_LINEAR)
        hipLaunchKernelGGL(interp_kernel<false>, grid, dim3(kThreads), 0, s, d_in, d_out, static_cast<uint32_t>(channels),
                           in_frames, out_frames, ratio);
    else
        hipLaunchKernelGGL(interp_kernel<true>, grid, dim3(kThreads), 0, s, d_in, d_out, static_cast<uint32_t>(channels),
                           in_frames, out_frames, ratio);
    RSMP_HIP_CHECK(hipGetLastError());
    return RSMP_OK;
}

extern "C" int rsmp_interp_resample(int mode, size_t channels, uint32_t in_hz, uint32_t out_hz, const float* in,
                                    size_t in_len, float* out, size_t out_cap, size_t* produced) {
    if (int rc = check_device()) return rc;
    rsmp::DeviceBuffer d_in, d_out;
    RSMP_HIP_CHECK(d_in.reserve((in_len + 4) * sizeof(float)));
    RSMP_HIP_CHECK(d_out.reserve((out_cap + 4) * sizeof(float)));
    if (in_len) RSMP_HIP_CHECK(hipMemcpy(d_in.get(), in, in_len * sizeof(float), hipMemcpyHostToDevice));
    size_t p = 0;
    const int rc = rsmp_interp_resample_device(mode, channels, in_hz, out_hz, d_in.as<float>(), in_len,
                                               d_out.as<float>(), out_cap, &p, nullptr);
    if (rc != RSMP_OK) return rc;
    RSMP_HIP_CHECK(hipDeviceSynchronize());
    if (p) RSMP_HIP_CHECK(hipMemcpy(out, d_out.get(), p * sizeof(float), hipMemcpyDeviceToHost));
    if (produced) *produced = p;
    return RSMP_OK;
}

extern "C" int rsmp_pcm_to_stereo_f32_device(const void* d_pcm, int bits, int channels, size_t n_samples,
                                             float* d_out, void* stream) {
    if (int rc = check_device()) return rc;
    if ((bits != 16 && bits != 24 && bits != 32) || (channels != 1 && channels != 2))
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_pcm_to_stereo_f32: 16 / 24 / 32 bits, 1 or 2 channels");
    if (n_samples == 0) return RSMP_OK;
    const dim3 grid(static_cast<uint32_t>((n_samples + kThreads - 1) / kThreads));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint8_t* p = static_cast<const uint8_t*>(d_pcm);
    const uint32_t mono = channels == 1 ? 1u : 0u;
    if (bits == 16) hipLaunchKernelGGL(pcm_kernel<16>, grid, dim3(kThreads), 0, s, p, d_out, static_cast<uint64_t>(n_samples), mono);
    else if (bits == 24) hipLaunchKernelGGL(pcm_kernel<24>, grid, dim3(kThreads), 0, s, p, d_out, static_cast<uint64_t>(n_samples), mono);
    else hipLaunchKernelGGL(pcm_kernel<32>, grid, dim3(kThreads), 0, s, p, d_out, static_cast<uint64_t>(n_samples), mono);
    RSMP_HIP_CHECK(hipGetLastError());
    return RSMP_OK;
}

// ---- measurement aid: the chip's streaming rate --------------------------------------------------------------
// A plain copy, 16 bytes per lane and instruction, four workgroups per compute unit walking the buffer side by side (the
// whole grid touches one contiguous stretch per iteration): what a kernel that reads as much as it writes can reach at
// all.  tools/copy_probe.hip swept the shape on the pool's MI355X (0.56 GB buffers): this one 5.83 TB/s of read +
// written bytes; eight workgroups per CU 4.9, four loads in flight per lane 4.2-4.4, non-temporal accesses 4.3-5.6 --
// the guide's figure is 6.29 TB/s.  bench.py prices every roofline fraction against the 8 TB/s spec AND reports this
// figure beside it (`roofline.device_copy`).
namespace {
typedef float v4f_copy __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(const v4f_copy* __restrict__ src, v4f_copy* __restrict__ dst, uint64_t n16) {
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace

extern "C" int rsmp_device_stream_copy(const float* d_src, float* d_dst, size_t n_values, void* stream) {
    if (int rc = check_device()) return rc;
    if ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(d_dst)) % 16 != 0 || n_values % 4 != 0)
        return rsmp::fail(RSMP_ERR_INVALID_ARGUMENT, "rsmp_device_stream_copy: 16-byte aligned buffers of whole 16-byte pieces");
    if (n_values == 0) return RSMP_OK;
    int cus = 256;
    int device = 0;
    (void)hipGetDevice(&device);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    hipLaunchKernelGGL(stream_copy_kernel, dim3(static_cast<uint32_t>(cus) * 4u), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const v4f_copy*>(d_src), reinterpret_cast<v4f_copy*>(d_dst), static_cast<uint64_t>(n_values / 4));
    RSMP_HIP_CHECK(hipGetLastError());
    return RSMP_OK;
}
