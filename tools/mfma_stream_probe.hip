// mfma_stream_probe.hip -- which ingredient of the FIR matrix-core stream costs MFMA issue rate?
// One wave per SIMD, 144-MFMA units as in the kernel (36 steps x 2 groups x 2 channels), variants:
//   0: operands in registers          1: + B from LDS (ds_read_b64 ring of 4, 3 steps ahead)
//   2: + A from 36 registers          3: + A registers refilled from global memory each block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(768) void probe(float* out, const float* table, int units, int stride) {
    extern __shared__ float lds[];
    const unsigned lane = threadIdx.x & 63, k = lane >> 4, pi = lane & 15;
    for (unsigned i = threadIdx.x; i < 65u * stride + 128; i += blockDim.x) lds[i] = i * 1e-6f;
    __syncthreads();
    typedef const v4f __attribute__((address_space(1)))* gptr_v4f;
    gptr_v4f src = (gptr_v4f)table + lane;
    v4f a[9];
    for (int j = 0; j < 9; ++j) a[j] = src[j * 64];
    const float* xb[2] = {lds + pi * stride + 2 * k, lds + (16 + pi) * stride + 2 * k};
    v2f x[4][2];
    for (int c = 0; c < 3; ++c)
        for (int g = 0; g < 2; ++g) x[c][g] = *reinterpret_cast<const v2f*>(xb[g] + 8 * c);
    float keep = 0.f;
    for (int u = 0; u < units; ++u) {
        v4f acc[2][2];
#pragma unroll
        for (int c = 0; c < 36; ++c) {
            if (MODE >= 1) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    x[(c + 3) & 3][g] = *reinterpret_cast<const v2f*>(xb[g] + 8 * ((c + 3) % 36));
            }
            const v4f av = MODE >= 2 ? a[c >> 2] : a[0];
            const float af = (c & 3) == 0 ? av.x : (c & 3) == 1 ? av.y : (c & 3) == 2 ? av.z : av.w;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const v4f z = v4f{0.f, 0.f, 0.f, 0.f};
                acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, x[c & 3][g].x, c == 0 ? z : acc[g][0], 0, 0, 0);
                acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, x[c & 3][g].y, c == 0 ? z : acc[g][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 3 && (c & 3) == 3) {
                a[c >> 2] = src[(c >> 2) * 64 + ((u & 7) * 9 * 64)];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        keep += acc[0][0].x + acc[0][1].y + acc[1][0].z + acc[1][1].w;
    }
    if (keep == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}

int main() {
    float *out, *table;
    CHECK(hipMalloc(&out, 1 << 22));
    CHECK(hipMalloc(&table, 1 << 20));
    CHECK(hipMemset(table, 0, 1 << 20));
    const int stride = 294, lds_bytes = (65 * stride + 128) * 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const void* fns[4] = {(const void*)probe<0>, (const void*)probe<1>, (const void*)probe<2>, (const void*)probe<3>};
    for (int m = 0; m < 4; ++m) CHECK(hipFuncSetAttribute(fns[m], hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    for (int mode = 0; mode < 4; ++mode)
        for (int wps : {1, 2}) {
            const int units = 400;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                void* args[] = {&out, &table, (void*)&units, (void*)&stride};
                CHECK(hipLaunchKernel(fns[mode], dim3(256), dim3(256 * wps), args, lds_bytes, 0));
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double fma = 256.0 * 4 * wps * units * 144.0 * 1024.0;
            printf("mode %d waves/SIMD=%d: %.3f ms  %.2f TFMA/s\n", mode, wps, ms, fma / ms * 1e-9);
        }
    return 0;
}
