#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* clk, int iters) {
    v4f acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    bf16x8 a8, b8; bf16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 1e-3f + i); b8[i] = (__bf16)(1.0f + i * 1e-2f); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a4), __builtin_bit_cast(s16x4, b4), acc[i], 0, 0, 0);
            }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y;
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}
int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 4096); hipMalloc(&clk, 256 * 8);
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, clk, iters);
        else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, clk, iters);
        hipDeviceSynchronize();
        unsigned long long h[256]; hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
        printf("%s: %.2f shader cycles per MFMA (one wave per SIMD)\n", mode ? "16x16x16 bf16_1k" : "16x16x32 bf16", m / 256 / (iters * 32.0));
    }
    return 0;
}
