// xcd_lab: what does it cost workgroups to meet INSIDE a kernel when only the 32 workgroups of one XCD (one L2) meet, instead of all
// 256 (tools/barrier_lab.hip: 10-19 us chip-wide)?  One 256-thread workgroup per CU, ITER meetings per launch, three protocols:
//   chip : one counter, 256 arrivals (relaxed agent-scope atomic add, sc1 poll by one lane + s_sleep)
//   xcd  : one counter per XCD (HW_REG_XCC_ID), 32 arrivals each -- no cross-XCD traffic at all
//   xcd+payload : the same, every workgroup first publishes 1 KB with write-through (sc1) stores and reads its left neighbour's 1 KB
//                 on the same XCD with sc1 loads after the meeting (the shape of a fused GEMV -> GEMV hand-off)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ int xcc_id() { return (int)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }   // HW_REG_XCC_ID bits 3:0
template <int MODE>
__global__ __launch_bounds__(256) void meet_kernel(unsigned* counters, float* payload, int iters, int* xcc_of, int* slot_of, unsigned* slot_ctr, long long* cyc, float* sink) {
    __shared__ int sh_slot;
    const int xcc = xcc_id();
    if (threadIdx.x == 0) {
        sh_slot = (int)atomicAdd(&slot_ctr[xcc], 1u);          // this workgroup's rank inside its XCD
        xcc_of[blockIdx.x] = xcc;
        slot_of[blockIdx.x] = sh_slot;
    }
    __syncthreads();
    const int slot = sh_slot;
    unsigned* ctr = MODE == 0 ? counters : counters + 64 * (1 + xcc);   // own cache line per XCD
    const unsigned per = MODE == 0 ? gridDim.x : gridDim.x / 8;
    float acc = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (MODE == 2) {
            float* mine = payload + ((size_t)(xcc * 64 + slot) * 256 + threadIdx.x);
            __hip_atomic_store((unsigned*)mine, __float_as_uint((float)(it + slot)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = per * (unsigned)(it + 1);
            while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (MODE == 2) {
            const int left = (slot + (int)per - 1) % (int)per;
            const float* theirs = payload + ((size_t)(xcc * 64 + left) * 256 + threadIdx.x);
            acc += __uint_as_float(__hip_atomic_load((const unsigned*)theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    if (acc == 12345.f) sink[0] = acc;
}
template <int MODE> void run(const char* name, int grid) {
    unsigned *ctr, *slot_ctr; float *payload, *sink; int *xcc_of, *slot_of; long long* cyc;
    CK(hipMalloc(&ctr, 64 * 9 * 4)); CK(hipMalloc(&slot_ctr, 64)); CK(hipMalloc(&payload, 8 * 64 * 256 * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMalloc(&xcc_of, grid * 4)); CK(hipMalloc(&slot_of, grid * 4)); CK(hipMalloc(&cyc, 8));
    const int iters = 200;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f; long long bc = 0;
    std::vector<int> hx(grid);
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemset(ctr, 0, 64 * 9 * 4)); CK(hipMemset(slot_ctr, 0, 64));
        CK(hipEventRecord(e0));
        meet_kernel<MODE><<<grid, 256>>>(ctr, payload, iters, xcc_of, slot_of, slot_ctr, cyc, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        if (ms < best) { best = ms; bc = c; }
    }
    CK(hipMemcpy(hx.data(), xcc_of, grid * 4, hipMemcpyDeviceToHost));
    int per_x[16] = {0}, rr = 0;
    for (int b = 0; b < grid; b++) { per_x[hx[b]]++; rr += hx[b] == (b % 8); }
    printf("%-14s grid %3d: %6.2f us per meeting (events), %6.0f shader cycles per meeting;  workgroups per XCD %d %d %d %d %d %d %d %d, block b on XCD b%%8: %d / %d\n", name, grid,
           best * 1e3 / iters, (double)bc / iters, per_x[0], per_x[1], per_x[2], per_x[3], per_x[4], per_x[5], per_x[6], per_x[7], rr, grid);
}
int main() {
    run<0>("chip", 256);
    run<1>("xcd", 256);
    run<2>("xcd+payload", 256);
    run<0>("chip", 128);
    run<1>("xcd", 128);
    return 0;
}
