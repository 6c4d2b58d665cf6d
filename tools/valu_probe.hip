// valu_probe.hip -- microbenchmark: fp32 FMA issue rate on gfx950 for the instruction forms the
// FIR kernel can use (v_fma_f32 with an SGPR operand, v_pk_fma_f32 with an SGPR pair), at several
// occupancies.  Prints achieved TFMA/s per form.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void probe(float* out, const float* coef, int iters) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    float b0 = 8, b1 = 9, b2 = 10, b3 = 11, b4 = 12, b5 = 13, b6 = 14, b7 = 15;
    float x0 = threadIdx.x * 1e-3f, x1 = 0.5f;
    float s0 = coef[0], s1 = coef[1];  // uniform -> SGPRs
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile(
                "v_fma_f32 %0, %16, %18, %0\n v_fma_f32 %1, %17, %18, %1\n"
                "v_fma_f32 %2, %16, %18, %2\n v_fma_f32 %3, %17, %18, %3\n"
                "v_fma_f32 %4, %16, %18, %4\n v_fma_f32 %5, %17, %18, %5\n"
                "v_fma_f32 %6, %16, %18, %6\n v_fma_f32 %7, %17, %18, %7\n"
                "v_fma_f32 %8, %16, %19, %8\n v_fma_f32 %9, %17, %19, %9\n"
                "v_fma_f32 %10, %16, %19, %10\n v_fma_f32 %11, %17, %19, %11\n"
                "v_fma_f32 %12, %16, %19, %12\n v_fma_f32 %13, %17, %19, %13\n"
                "v_fma_f32 %14, %16, %19, %14\n v_fma_f32 %15, %17, %19, %15\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                  "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                : "s"(s0), "s"(s1), "v"(x0), "v"(x1));
        } else {
            // 8 packed FMAs = 16 FMAs, SGPR pair broadcast low half
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 A0 = {a0, a1}, A1 = {a2, a3}, A2 = {a4, a5}, A3 = {a6, a7};
            f2 B0 = {b0, b1}, B1 = {b2, b3}, B2 = {b4, b5}, B3 = {b6, b7};
            f2 X = {x0, x1};
            f2 S = {s0, s1};
            asm volatile(
                "v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[0,1,1]\n"
                "v_pk_fma_f32 %1, %8, %9, %1 op_sel:[1,0,0]\n"
                "v_pk_fma_f32 %2, %8, %9, %2 op_sel_hi:[0,1,1]\n"
                "v_pk_fma_f32 %3, %8, %9, %3 op_sel:[1,0,0]\n"
                "v_pk_fma_f32 %4, %8, %9, %4 op_sel_hi:[0,1,1]\n"
                "v_pk_fma_f32 %5, %8, %9, %5 op_sel:[1,0,0]\n"
                "v_pk_fma_f32 %6, %8, %9, %6 op_sel_hi:[0,1,1]\n"
                "v_pk_fma_f32 %7, %8, %9, %7 op_sel:[1,0,0]\n"
                : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(B0), "+v"(B1), "+v"(B2), "+v"(B3)
                : "s"(S), "v"(X));
            a0 = A0.x; a1 = A0.y; a2 = A1.x; a3 = A1.y; a4 = A2.x; a5 = A2.y; a6 = A3.x; a7 = A3.y;
            b0 = B0.x; b1 = B0.y; b2 = B1.x; b3 = B1.y; b4 = B2.x; b5 = B2.y; b6 = B3.x; b7 = B3.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] =
        a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

int main() {
    float *out, *coef;
    CHECK(hipMalloc(&out, 256 * 2048 * 8 * sizeof(float)));
    CHECK(hipMalloc(&coef, 64));
    float h[2] = {1.0f, 0.999f};
    CHECK(hipMemcpy(coef, h, 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int wpc : {4, 8, 16, 32}) {   // waves per CU
            dim3 grid(256 * (wpc / 4)), block(256);
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(probe<0>, grid, block, 0, 0, out, coef, iters);
                else hipLaunchKernelGGL(probe<1>, grid, block, 0, 0, out, coef, iters);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
            }
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            double fma = double(grid.x) * 256 * iters * 16.0;
            printf("%s waves/CU=%2d : %.3f ms  %.2f TFMA/s (%.1f TFLOP/s)\n",
                   mode == 0 ? "v_fma_f32(sgpr)   " : "v_pk_fma_f32(sgpr)", wpc, ms,
                   fma / ms * 1e-9, 2 * fma / ms * 1e-9);
        }
    }
    return 0;
}
