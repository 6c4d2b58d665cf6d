// chunk_probe.hip -- isolates the FIR inner chunk (4 x s_load_dwordx16 + 64 v_pk_fma_f32 with 32
// distinct SGPR pairs, 8 packed accumulators) without LDS / HBM traffic, to learn the attainable
// FMA rate of that instruction mix at several occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const v2f __attribute__((address_space(4)))* const_v2f_ptr;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void pk_fma8(v2f (&acc)[8], v2f c0, v2f c1, v2f c2, v2f c3, v2f x) {
    asm("v_pk_fma_f32 %0, %8, %12, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %8, %12, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %2, %9, %12, %2 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %3, %9, %12, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %4, %10, %12, %4 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %5, %10, %12, %5 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %6, %11, %12, %6 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %7, %11, %12, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]),
          "+v"(acc[6]), "+v"(acc[7])
        : "s"(c0), "s"(c1), "s"(c2), "s"(c3), "v"(x));
}

template <int MODE>  // 0: coefficients reloaded per chunk (same 256 B: scalar-cache hits); 1: loaded once
__global__ void probe(float* out, const float* table, int chunks) {
    v2f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v2f{(float)threadIdx.x, 1.f};
    const_v2f_ptr gc = (const_v2f_ptr)table;
    v2f x[8];
    for (int u = 0; u < 8; ++u) x[u] = v2f{threadIdx.x * 1e-3f + u, 0.5f};
    v2f c[32];
    if (MODE == 1) for (int i = 0; i < 32; ++i) c[i] = gc[i];
    for (int k = 0; k < chunks; ++k) {
        if (MODE == 0) {
            const_v2f_ptr g = gc + 32 * (k & 15);
#pragma unroll
            for (int i = 0; i < 32; ++i) c[i] = g[i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) pk_fma8(acc, c[4 * u], c[4 * u + 1], c[4 * u + 2], c[4 * u + 3], x[u]);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *out, *table;
    CHECK(hipMalloc(&out, 512 * 1024 * sizeof(float)));
    CHECK(hipMalloc(&table, 65536));
    CHECK(hipMemset(table, 0, 65536));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int chunks = 4000;
    for (int mode = 0; mode < 2; ++mode)
        for (int wpc : {8, 12, 16, 24, 32}) {
            dim3 grid(256 * 2), block(64 * wpc / 2);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(probe<0>, grid, block, 0, 0, out, table, chunks);
                else hipLaunchKernelGGL(probe<1>, grid, block, 0, 0, out, table, chunks);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            double fma = double(grid.x) * block.x * chunks * 128.0;
            printf("%s waves/CU=%2d : %.3f ms  %.2f TFMA/s\n",
                   mode == 0 ? "s_load per chunk" : "coefs resident  ", wpc, ms, fma / ms * 1e-9);
        }
    return 0;
}
