// tile_probe.hip -- the periodic FIR tile loop without HBM traffic or barriers: per chunk NT x
// ds_read_b64 of the lane's row (odd frame stride, as the kernel), NC*NT/2 coefficient pairs through
// the scalar cache from a table larger than the cache (each wave walks its own tile), NC*NT packed
// FMAs.  Answers: what FMA rate does the instruction mix allow at a given occupancy, for
// (NC classes x NT taps) = (8 x 8) and (16 x 4)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const v2f __attribute__((address_space(4)))* const_v2f_ptr;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void pk_fma2(v2f& a0, v2f& a1, v2f c, v2f x) {
    asm("v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
        : "+v"(a0), "+v"(a1)
        : "s"(c), "v"(x));
}

// NC classes per tile, NT taps per chunk; row_len taps per tile
template <int NC, int NT>
__global__ void probe(float* out, const float* table, int table_tiles, int tiles, int row_len,
                      int row_stride, int aligned) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 65 * row_stride; i += blockDim.x) lds[i] = i * 1e-6f;
    __syncthreads();
    const float* row = lds + lane * row_stride + (aligned ? (lane & 1) : 0);
    float keep = 0.f;
    const int n_chunks = row_len / NT;
    for (int t = 0; t < tiles; ++t) {
        const int tile = (blockIdx.x * 7 + wave * 3 + t) % table_tiles;
        const_v2f_ptr gc = (const_v2f_ptr)(table + (size_t)tile * row_len * NC);
        const float* px = row + 2 * ((t * 5) % 11);
        v2f acc[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) acc[i] = v2f{0.f, 0.f};
        for (int c = 0; c < n_chunks; ++c) {
            v2f x[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) x[u] = *reinterpret_cast<const v2f*>(px + 2 * (NT * c + u));
            v2f cf[NC * NT / 2];
#pragma unroll
            for (int i = 0; i < NC * NT / 2; ++i) cf[i] = gc[NC * NT / 2 * c + i];
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int k = 0; k < NC / 2; ++k) pk_fma2(acc[2 * k], acc[2 * k + 1], cf[u * NC / 2 + k], x[u]);
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) keep += acc[i].x + acc[i].y;
    }
    if (keep == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}

// Half-chunk software pipeline: the coefficients and samples of the next 4 taps are requested
// before the packed FMAs of the current 4 taps, so every scalar / LDS load has 32 FMAs of cover
// inside its own wave (SMEM returns out of order: the only usable wait is lgkmcnt(0), so exactly one
// group may be in flight).
__global__ void probe_pipe(float* out, const float* table, int table_tiles, int tiles, int row_len,
                           int row_stride) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 65 * row_stride; i += blockDim.x) lds[i] = i * 1e-6f;
    __syncthreads();
    const float* row = lds + lane * row_stride;
    float keep = 0.f;
    const int n_groups = row_len / 4;
    for (int t = 0; t < tiles; ++t) {
        const int tile = (blockIdx.x * 7 + wave * 3 + t) % table_tiles;
        const_v2f_ptr gc = (const_v2f_ptr)(table + (size_t)tile * row_len * 8);
        const float* px = row + 2 * ((t * 5) % 11);
        v2f acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = v2f{0.f, 0.f};
        v2f ca[16], xa[4], cb[16], xb[4];
#pragma unroll
        for (int i = 0; i < 16; ++i) ca[i] = gc[i];
#pragma unroll
        for (int u = 0; u < 4; ++u) xa[u] = *reinterpret_cast<const v2f*>(px + 2 * u);
        for (int g = 0; g < n_groups; g += 2) {
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): group g has landed
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) cb[i] = gc[16 * (g + 1) + i];
#pragma unroll
            for (int u = 0; u < 4; ++u) xb[u] = *reinterpret_cast<const v2f*>(px + 2 * (4 * (g + 1) + u));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) pk_fma2(acc[2 * k], acc[2 * k + 1], ca[u * 4 + k], xa[u]);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);   // group g + 1 has landed
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < n_groups) {
#pragma unroll
                for (int i = 0; i < 16; ++i) ca[i] = gc[16 * (g + 2) + i];
#pragma unroll
                for (int u = 0; u < 4; ++u) xa[u] = *reinterpret_cast<const v2f*>(px + 2 * (4 * (g + 2) + u));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) pk_fma2(acc[2 * k], acc[2 * k + 1], cb[u * 4 + k], xb[u]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) keep += acc[i].x + acc[i].y;
    }
    if (keep == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}

int main() {
    float *out, *table;
    const int row_len = 136;
    CHECK(hipMalloc(&out, 1 << 22));
    CHECK(hipMalloc(&table, (size_t)20 * row_len * 16 * 4));
    CHECK(hipMemset(table, 0, (size_t)20 * row_len * 16 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int row_stride = 2 * 147;   // 8-byte aligned lanes, 38 mod 64 banks: conflict-free b64
    const int lds_bytes = 65 * row_stride * 4;
    CHECK(hipFuncSetAttribute((const void*)probe<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CHECK(hipFuncSetAttribute((const void*)probe<16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CHECK(hipFuncSetAttribute((const void*)probe<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CHECK(hipFuncSetAttribute((const void*)probe_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    for (int variant = 0; variant < 4; variant += 2)
    for (int shape = 0; shape < 3; shape += 2)
        for (int wgs : {1, 2})
            for (int wpw : {4, 6, 8, 12, 16}) {   // waves per workgroup
                const int aligned = variant & 1;
                const int table_tiles = (variant & 2) ? 2 : 20;
                dim3 grid(256 * wgs), block(64 * wpw);
                const int tiles = shape == 1 ? 60 : 120;   // same FMA count per wave
                float ms = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(e0));
                    if (shape == 0) hipLaunchKernelGGL((probe<8, 8>), grid, block, lds_bytes, 0, out, table, table_tiles, tiles, row_len, row_stride, aligned);
                    else if (shape == 1) hipLaunchKernelGGL((probe<16, 4>), grid, block, lds_bytes, 0, out, table, table_tiles / 2, tiles, row_len, row_stride, aligned);
                    else hipLaunchKernelGGL(probe_pipe, grid, block, lds_bytes, 0, out, table, table_tiles, tiles, row_len, row_stride);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                }
                const double fma = double(grid.x) * block.x * 120 * row_len * 8 * 2;
                printf("aligned=%d table_tiles=%d %s wgs/CU=%d waves/WG=%2d (waves/CU=%2d): %.3f ms  %.2f TFMA/s\n",
                       aligned, table_tiles, shape == 0 ? "8x8 " : shape == 1 ? "16x4" : "pipe", wgs, wpw, wgs * wpw, ms, fma / ms * 1e-9);
            }
    return 0;
}
