// dma_probe.hip -- how fast can P waves of one workgroup per CU stage a 75 KB LDS image with
// global_load_lds (4 B per lane), all 256 CUs streaming distinct HBM regions at once?  Decides how
// many producer waves a double-buffered persistent FIR workgroup needs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef const float __attribute__((address_space(1)))* gconst_f32_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

__global__ __launch_bounds__(1024) void probe(const float* src, unsigned long long* ticks, float* sink,
                                               int producers, int region, int iters, int busy) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t_issue = 0, t_land = 0;
    float acc = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if ((int)wave < producers) {
            gconst_f32_ptr s = (gconst_f32_ptr)src + ((size_t)(it * gridDim.x + blockIdx.x) * region) + lane;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (int base = wave * 64; base < region; base += producers * 64)
                if (base + (int)lane < region)
                    __builtin_amdgcn_global_load_lds(s + base, (lds_void_ptr)(lds + base), 4, 0, 0);
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
            t_issue += t1 - t0;
            t_land += t2 - t0;
        } else if (busy) {
            for (int k = 0; k < 4000; ++k) acc = fmaf(acc, 1.0001f, 0.5f);
        }
    }
    if (acc == 12345.f) sink[0] = acc + lds[threadIdx.x];
    if (lane == 0 && (int)wave < producers) {
        ticks[(blockIdx.x * 16 + wave) * 2] = t_issue;
        ticks[(blockIdx.x * 16 + wave) * 2 + 1] = t_land;
    }
}

int main() {
    const int region = 65 * 294, iters = 16, grid = 256;
    float *src, *sink;
    unsigned long long* ticks;
    const size_t src_floats = (size_t)grid * iters * region + 64;
    CHECK(hipMalloc(&src, src_floats * 4));
    CHECK(hipMemset(src, 0, src_floats * 4));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMalloc(&ticks, grid * 16 * 2 * 8));
    CHECK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int busy = 0; busy < 2; ++busy)
        for (int producers : {1, 2, 4, 8, 16}) {
            CHECK(hipMemset(ticks, 0, grid * 16 * 2 * 8));
            hipLaunchKernelGGL(probe, dim3(grid), dim3(1024), 150 * 1024, 0, src, ticks, sink, producers, region, iters, busy);
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(grid * 16 * 2);
            CHECK(hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost));
            double issue = 0, land = 0;
            int n = 0;
            for (int b = 0; b < grid; ++b)
                for (int w = 0; w < producers; ++w) {
                    issue += h[(b * 16 + w) * 2];
                    land += h[(b * 16 + w) * 2 + 1];
                    ++n;
                }
            printf("other waves %s, producers=%2d: issue %.2f us, landed %.2f us per 75 KB image (all CUs at once: %.2f TB/s)\n",
                   busy ? "busy" : "idle", producers, issue / n / iters / 100, land / n / iters / 100,
                   grid * region * 4.0 / (land / n / iters / 100 * 1e-6) * 1e-12);
        }
    return 0;
}
