// f16_mfma_probe: does v_mfma_f32_16x16x32_f16 keep fp16 DENORMAL inputs (needed by a two-plane fp16 split of
// f32 operands), and what does it cost per instruction next to the bf16 form?  Also: which instruction the
// compiler picks for a packed f32 -> f16 conversion.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void denorm(float* out) {
    f16x8 a, b;
    // A[class][k]: every element a denormal 2^-20 (fp16 min normal is 2^-14); B = 1
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1.0f; }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    out[threadIdx.x] = acc.x;   // expect 32 * 2^-20 = 3.0517578125e-05
    // denormal times large: 2^-20 * 2^10
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)1024.0f;
    v4f acc2 = {0.f, 0.f, 0.f, 0.f};
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc2, 0, 0, 0);
    out[64 + threadIdx.x] = acc2.x;   // expect 32 * 2^-10 = 0.03125
}

__global__ void cvt(const float* in, unsigned* out) {
    f32x2 v = {in[2 * threadIdx.x], in[2 * threadIdx.x + 1]};
    f16x2 h = __builtin_convertvector(v, f16x2);
    out[threadIdx.x] = __builtin_bit_cast(unsigned, h);
}

template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, unsigned long long* clk, int iters) {
    v4f acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b; bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(1.0f + i * 1e-2f); ab[i] = (__bf16)(float)a[i]; bb[i] = (__bf16)(float)b[i]; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
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
    hipLaunchKernelGGL(denorm, dim3(1), dim3(64), 0, 0, out);
    float h[128]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("denormal A x 1.0: %.10g (expect 3.0517578125e-05)   denormal A x 1024: %.10g (expect 0.03125)\n", h[0], h[64]);
    float hin[128]; for (int i = 0; i < 128; ++i) hin[i] = 0.1f * i + 1e-5f * i;
    float* din; unsigned* dout; hipMalloc(&din, sizeof hin); hipMalloc(&dout, 64 * 4);
    hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, din, dout);
    unsigned ho[64]; hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    printf("cvt sample: %08x\n", ho[3]);
    for (int mode = 0; mode < 2; ++mode) {
        const int iters = 2000;
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, out, clk, iters);
        else hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, out, clk, iters);
        hipDeviceSynchronize();
        unsigned long long hc[256]; hipMemcpy(hc, clk, sizeof hc, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256; ++i) m += hc[i];
        printf("%s: %.2f shader cycles per MFMA (one wave per SIMD)\n", mode ? "16x16x32 bf16" : "16x16x32 f16", m / 256 / (iters * 32.0));
    }
    return 0;
}
