// barrier_lab: cost of an in-kernel grid barrier (256 co-resident workgroups, one per CU) vs the 1.55 us kernel edge.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
// flag barrier: every workgroup publishes its own epoch word (write-through store, distinct addresses => no atomic
// serialisation) and 256 threads poll the 256 words with one coalesced sc1 load per round
__device__ __forceinline__ void flag_barrier(unsigned* flags, unsigned epoch) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 64) {
        int spins = 0;
        for (;;) {
            bool ok = true;
            for (unsigned i = threadIdx.x; i < gridDim.x; i += 64)
                ok = ok && (__hip_atomic_load(flags + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= epoch);
            if (__all(ok) || ++spins > (1 << 20)) break;
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(512) void kf(unsigned* flags, int n) {
    for (int i = 0; i < n; i++) flag_barrier(flags, (unsigned)(i + 1));
}
// VAR 0: barrier only.  VAR 1: barrier + every workgroup writes 16 floats (sc1) before and reads all 4096 (sc1) after.
template <int VAR>
__global__ __launch_bounds__(512) void k(unsigned* ctr, float* buf, int n, float* out) {
    float acc = 0.f;
    for (int i = 0; i < n; i++) {
        if (VAR == 1) {
            float* dst = buf + (i & 1) * 4096;
            if (threadIdx.x < 16) __hip_atomic_store((unsigned*)dst + blockIdx.x * 16 + threadIdx.x, __float_as_uint((float)i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        grid_barrier(ctr, (unsigned)(i + 1) * gridDim.x);
        if (VAR == 1) {
            const float* src = buf + (i & 1) * 4096;
            for (int j = threadIdx.x; j < 4096; j += blockDim.x)
                acc += __uint_as_float(__hip_atomic_load((const unsigned*)src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
    if (acc == 12345.f) out[0] = acc;
}
int main() {
    unsigned* ctr; float *buf, *out;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&buf, 2 * 4096 * 4)); CK(hipMalloc(&out, 4));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const int n = 2000;
    for (int var = 0; var < 2; var++)
        for (int threads : {64, 512}) {
            for (int it = 0; it < 2; it++) {
                CK(hipMemset(ctr, 0, 4));
                CK(hipEventRecord(t0));
                if (var == 0) k<0><<<256, threads>>>(ctr, buf, n, out); else k<1><<<256, threads>>>(ctr, buf, n, out);
                CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
            }
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            unsigned c; CK(hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost));
            printf("var %d threads %3d: %6.2f us per barrier  (counter %u, expect %u)\n", var, threads, ms * 1e3 / n, c, 256u * n);
        }
    unsigned* flags; CK(hipMalloc(&flags, 1024 * 4));
    for (int threads : {64, 512}) {
        for (int it = 0; it < 2; it++) {
            CK(hipMemset(flags, 0, 1024 * 4));
            CK(hipEventRecord(t0));
            kf<<<256, threads>>>(flags, n);
            CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
        }
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        printf("flag barrier threads %3d: %6.2f us per barrier\n", threads, ms * 1e3 / n);
    }
    return 0;
}
