// tools/copy_probe.hip -- what a plain device copy reaches on this box: 16 bytes per lane, grid x unroll x (non-)temporal
// hipcc -O3 --offload-arch=gfx950 tools/copy_probe.hip -o tools/bin/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k(const v4* __restrict__ s, v4* __restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(r[u], d + i + u * stride); else d[i + u * stride] = r[u]; }
    }
    for (; i < n; i += stride) d[i] = s[i];
}
template <int U, bool NT> void run(const v4* s, v4* d, size_t n, int wgs, const char* name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<U, NT><<<wgs, 256>>>(s, d, n);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) k<U, NT><<<wgs, 256>>>(s, d, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%-10s wgs %5d  %.4f ms  %.0f GB/s\n", name, wgs, ms, 2.0 * n * 16 / ms / 1e6);
}
int main() {
    for (size_t bytes : {(size_t)560 << 20, (size_t)2048 << 20}) {
        size_t n = bytes / 16; v4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes);
        printf("buffer %zu MiB\n", bytes >> 20);
        for (int wgs : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
            run<1, false>(s, d, n, wgs, "u1");
            run<4, false>(s, d, n, wgs, "u4");
            run<4, true>(s, d, n, wgs, "u4 nt");
            run<8, false>(s, d, n, wgs, "u8");
        }
        hipFree(s); hipFree(d);
    }
}
