// Streaming-read microbenchmark: what HBM / Infinity-Cache read bandwidth can a plain 16B-per-lane kernel reach on
// this MI355X?  Calibrates the roofline next to the 8 TB/s spec.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using v4 = int __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void rd(const v4* __restrict__ p, size_t n, int* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x * UNROLL;
    v4 acc = {0, 0, 0, 0};
    for (; i + (size_t)(UNROLL - 1) * blockDim.x < n; i += stride) {
        v4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(p + i + (size_t)u * blockDim.x) : p[i + (size_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = 1;
}
template <int UNROLL, bool NT>
double run(const v4* d, size_t bytes, int grid, int iters, int* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    size_t n = bytes / 16;
    rd<UNROLL, NT><<<grid, 256>>>(d, n, out);
    hipEventRecord(e0);
    for (int i = 0; i < iters; i++) rd<UNROLL, NT><<<grid, 256>>>(d, n, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)bytes * iters / (ms * 1e-3) / 1e9;
}
int main() {
    size_t big = (size_t)2 << 30, small = (size_t)64 << 20, tiny = (size_t)16 << 20;
    v4* d; int* out; hipMalloc(&d, big); hipMalloc(&out, 4); hipMemset(d, 1, big);
    for (int grid : {256, 512, 1024, 2048, 4096, 16384}) {
        printf("grid %5d | HBM 2GiB: u4 %7.0f  u8 %7.0f  u8nt %7.0f | MALL 64MiB: u4 %7.0f u8 %7.0f u8nt %7.0f | 16MiB/launch: u8 %7.0f (%.2f us) u8nt %7.0f GB/s\n", grid,
               run<4, false>(d, big, grid, 3, out), run<8, false>(d, big, grid, 3, out), run<8, true>(d, big, grid, 3, out),
               run<4, false>(d, small, grid, 50, out), run<8, false>(d, small, grid, 50, out), run<8, true>(d, small, grid, 50, out),
               run<8, false>(d, tiny, grid, 200, out), 16.777216e6 / run<8, false>(d, tiny, grid, 200, out) / 1e3, run<8, true>(d, tiny, grid, 200, out));
    }
    return 0;
}
