// bank_probe.hip -- does v_pk_fma_f32 slow down when its two 64-bit VGPR operands (src1 = samples,
// src2/dst = accumulator) sit in the same register banks (index mod 4)?  Explicit registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// 16 pk_fma per iteration; accumulators v[8:9] ... ; MODE 0: x in v[4:5] (banks 0,1) and all
// accumulators in banks 0,1 (v[8:9], v[12:13], ...): every instruction conflicts.
// MODE 1: x in v[6:7] (banks 2,3), accumulators in banks 0,1: never conflicts.
// MODE 2: accumulators alternate banks (compiler-like), x in v[4:5]: half conflict.
template <int MODE>
__global__ void probe(float* out, int iters) {
    float r = threadIdx.x * 1e-3f;
    if (MODE == 0)
        asm volatile(
            "v_mov_b32 v4, %1\n v_mov_b32 v5, %1\n"
            "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n"
            "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n"
            "s_mov_b32 s20, 0x3f7fbe77\n s_mov_b32 s21, 0x3f7fbe77\n"
            "1:\n"
            "v_pk_fma_f32 v[8:9], s[20:21], v[4:5], v[8:9] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[12:13], s[20:21], v[4:5], v[12:13] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[16:17], s[20:21], v[4:5], v[16:17] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[20:21], s[20:21], v[4:5], v[20:21] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[24:25], s[20:21], v[4:5], v[24:25] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[28:29], s[20:21], v[4:5], v[28:29] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[32:33], s[20:21], v[4:5], v[32:33] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[36:37], s[20:21], v[4:5], v[36:37] op_sel_hi:[0,1,1]\n"
            "s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n"
            "v_add_f32 %1, v8, v36\n"
            : "+s"(iters), "+v"(r) :: "v4","v5","v8","v9","v12","v13","v16","v17","v20","v21","v24","v25","v28","v29","v32","v33","v36","v37","s20","s21","scc");
    else if (MODE == 1)
        asm volatile(
            "v_mov_b32 v6, %1\n v_mov_b32 v7, %1\n"
            "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n"
            "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n"
            "s_mov_b32 s20, 0x3f7fbe77\n s_mov_b32 s21, 0x3f7fbe77\n"
            "1:\n"
            "v_pk_fma_f32 v[8:9], s[20:21], v[6:7], v[8:9] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[12:13], s[20:21], v[6:7], v[12:13] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[16:17], s[20:21], v[6:7], v[16:17] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[20:21], s[20:21], v[6:7], v[20:21] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[24:25], s[20:21], v[6:7], v[24:25] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[28:29], s[20:21], v[6:7], v[28:29] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[32:33], s[20:21], v[6:7], v[32:33] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[36:37], s[20:21], v[6:7], v[36:37] op_sel_hi:[0,1,1]\n"
            "s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n"
            "v_add_f32 %1, v8, v36\n"
            : "+s"(iters), "+v"(r) :: "v6","v7","v8","v9","v12","v13","v16","v17","v20","v21","v24","v25","v28","v29","v32","v33","v36","v37","s20","s21","scc");
    else
        asm volatile(
            "v_mov_b32 v4, %1\n v_mov_b32 v5, %1\n"
            "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n"
            "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
            "s_mov_b32 s20, 0x3f7fbe77\n s_mov_b32 s21, 0x3f7fbe77\n"
            "1:\n"
            "v_pk_fma_f32 v[8:9], s[20:21], v[4:5], v[8:9] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[10:11], s[20:21], v[4:5], v[10:11] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[12:13], s[20:21], v[4:5], v[12:13] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[14:15], s[20:21], v[4:5], v[14:15] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[16:17], s[20:21], v[4:5], v[16:17] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[18:19], s[20:21], v[4:5], v[18:19] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[20:21], s[20:21], v[4:5], v[20:21] op_sel_hi:[0,1,1]\n v_pk_fma_f32 v[22:23], s[20:21], v[4:5], v[22:23] op_sel_hi:[0,1,1]\n"
            "s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n"
            "v_add_f32 %1, v8, v22\n"
            : "+s"(iters), "+v"(r) :: "v4","v5","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","s20","s21","scc");
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 512 * 1024 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 200000;
    const char* names[3] = {"same banks (conflict)  ", "different banks        ", "alternating (half)     "};
    for (int mode = 0; mode < 3; ++mode)
        for (int wpc : {8, 24}) {
            dim3 grid(256 * 2), block(64 * wpc / 2);
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(probe<0>, grid, block, 0, 0, out, iters);
                else if (mode == 1) hipLaunchKernelGGL(probe<1>, grid, block, 0, 0, out, iters);
                else hipLaunchKernelGGL(probe<2>, grid, block, 0, 0, out, iters);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            double fma = double(grid.x) * block.x * double(iters) * 16.0;
            printf("%s waves/CU=%2d : %.3f ms  %.2f TFMA/s\n", names[mode], wpc, ms, fma / ms * 1e-9);
        }
    return 0;
}
