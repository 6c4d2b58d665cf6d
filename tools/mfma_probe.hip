// mfma_probe.hip -- attainable v_mfma_f32_16x16x4_f32 rate on this box: operands in registers, 4 or 8
// independent accumulators, 1-3 waves per SIMD, and the shader clock the chip holds meanwhile
// (s_memtime ticks per 100 MHz s_memrealtime tick).  The FIR matrix-core kernel is judged against this.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* clk, int iters) {
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int main() {
    float* out;
    unsigned long long* clk;
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMalloc(&clk, 256 * 2 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int nacc : {4, 8})
        for (int wps : {1, 2, 3}) {
            const int iters = 4000;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                if (nacc == 4) hipLaunchKernelGGL(probe<4>, dim3(256), dim3(256 * wps), 0, 0, out, clk, iters);
                else hipLaunchKernelGGL(probe<8>, dim3(256), dim3(256 * wps), 0, 0, out, clk, iters);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            std::vector<unsigned long long> h(512);
            CHECK(hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost));
            double ratio = 0;
            for (int b = 0; b < 256; ++b) ratio += double(h[2 * b]) / double(h[2 * b + 1]);
            const double fma = 256.0 * 4 * wps * iters * 8.0 * nacc * 1024.0;
            printf("acc=%d waves/SIMD=%d: %.3f ms  %.2f TFMA/s  s_memtime/s_memrealtime = %.3f (x100 MHz)\n",
                   nacc, wps, ms, fma / ms * 1e-9, ratio / 256);
        }
    return 0;
}
