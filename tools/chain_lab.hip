// chain_lab.hip -- what does ONE link of a sequential float chain cost on gfx950, by the form of its uniform operand?
// The reference-order attention (softMax's running sum, the value chains: jh_p16.h) is two chains of n dependent float operations per
// head; round 5 priced a link at ~11 cycles (v_readlane -> SGPR operand), round 6 moved the operand into the DPP field of the
// operation itself.  This lab times 4096 links of every candidate form, one wave per SIMD (4 waves per workgroup, like the
// attention kernel: only the clock of wave 0 is reported), s_memtime around the chain:
//   add  / fma   : plain VGPR operands (the floor of a dependent VALU chain for a wave alone on its SIMD)
//   add_dpp / fma_dpp : operand through row_newbcast (what the kernels use)
//   add_sgpr / fma_sgpr : operand in an SGPR loaded beforehand (s_load-style: no lift inside the chain)
//   fma_readlane : v_readlane one link ahead + fma with the SGPR (round 5's form, without its nops)
//   fma_mov_dpp  : v_mov_b32_dpp broadcast into a VGPR one link ahead + plain fma
// Second part: per-CU fill rate of an L2-resident buffer (1.3 MB, the activation image of a 129-row prompt chunk) at 1/2/4/8 waves
// per CU x 16-byte loads: the bound DESIGN 8.2 asserted for the BF16 prompt GEMM (~30 GB/s per CU) against the guide's L2 figure.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/chain_lab.hip -o tools/chain_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../jlama_amd/csrc/jh_p16.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int LINKS = 4096;
enum { F_ADD, F_FMA, F_ADD_DPP, F_FMA_DPP, F_ADD_SGPR, F_FMA_SGPR, F_FMA_READLANE, F_FMA_MOV_DPP, F_COUNT };
static const char* names[] = {"v_add_f32 (plain)", "v_fmac_f32 (plain)", "v_add_f32_dpp row_newbcast", "v_fmac_f32_dpp row_newbcast", "v_add_f32 sgpr operand",
                              "v_fmac_f32 sgpr operand", "v_readlane ahead + v_fmac sgpr", "v_mov_dpp ahead + v_fmac"};

#define REP16(X) X X X X X X X X X X X X X X X X
template <int FORM>
__global__ __launch_bounds__(256) void chain_kernel(const float* in, float* out, long long* cycles) {
    float acc = in[threadIdx.x & 63], x = in[64 + (threadIdx.x & 63)], v = in[128 + (threadIdx.x & 63)];
    const float sx = __builtin_amdgcn_readfirstlane(x);
    long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < LINKS / 16; i++) {
        if (FORM == F_ADD) asm volatile(REP16("v_add_f32 %0, %1, %0\n\t") : "+v"(acc) : "v"(x));
        if (FORM == F_FMA) asm volatile(REP16("v_fmac_f32 %0, %1, %2\n\t") : "+v"(acc) : "v"(x), "v"(v));
        if (FORM == F_ADD_DPP) asm volatile(REP16("v_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(acc) : "v"(x));
        if (FORM == F_FMA_DPP) asm volatile(REP16("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(acc) : "v"(x), "v"(v));
        if (FORM == F_ADD_SGPR) asm volatile(REP16("v_add_f32 %0, %1, %0\n\t") : "+v"(acc) : "s"(sx));
        if (FORM == F_FMA_SGPR) asm volatile(REP16("v_fmac_f32 %0, %1, %2\n\t") : "+v"(acc) : "s"(sx), "v"(v));
        if (FORM == F_FMA_READLANE) {
            float s0, s1;
            asm volatile("v_readlane_b32 %1, %3, 0\n\t"
                         REP16("v_readlane_b32 %2, %3, 1\n\tv_fmac_f32 %0, %1, %4\n\tv_readlane_b32 %1, %3, 2\n\tv_fmac_f32 %0, %2, %4\n\t")
                         : "+v"(acc), "=&s"(s0), "=&s"(s1) : "v"(x), "v"(v));
            i++;   // 32 links per trip
        }
        if (FORM == F_FMA_MOV_DPP) {
            float b0, b1;
            asm volatile("v_mov_b32_dpp %1, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                         REP16("v_mov_b32_dpp %2, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %0, %1, %4\n\tv_mov_b32_dpp %1, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32 %0, %2, %4\n\t")
                         : "+v"(acc), "=&v"(b0), "=&v"(b1) : "v"(x), "v"(v));
            i++;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(acc) : "memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// the attention kernel's own value chain (jh_p16.h: p16_value_chain_tile) on a synthetic transposed tile in LDS: what a link costs WITH its
// operand traffic (4 ds_read_b128 + 1 ds_read_b32 per 16 links, three register sets), wave 0 of a 256-thread workgroup, the others idle
template <int ACTIVE>
__global__ __launch_bounds__(256) void tile_chain_kernel(const float* in, float* out, long long* cycles, int cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = 512, TPP = TP + 4, DW = 32;
    float* vt = (float*)smem;
    float* w = vt + DW * TPP;
    for (int i = threadIdx.x; i < DW * TPP; i += 256) vt[i] = in[i & 255] * 1e-3f;
    for (int i = threadIdx.x; i < TP + 128; i += 256) w[i] = in[(i * 7) & 255] * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    long long t0 = 0, t1 = 0;
    if (threadIdx.x < ACTIVE) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        jh::p16_value_chain_tile(acc, vt + (size_t)(lane & (DW - 1)) * TPP, w + (lane & 15), cnt, TP);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(acc) : "memory");
        out[blockIdx.x * 64 + lane] = acc;
        if (lane == 0) cycles[blockIdx.x] = t1 - t0;
    }
}
__global__ __launch_bounds__(256) void sum_chain_kernel(const float* in, float* out, long long* cycles, int n16) {
    __shared__ float w[1024 + 128];
    for (int i = threadIdx.x; i < 1024 + 128; i += 256) w[i] = in[i & 255] * 1e-3f;
    __syncthreads();
    if (threadIdx.x < 64) {
        long long t0, t1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        const float s = jh::p16_seq_sum_wave(w, n16, threadIdx.x);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
        out[blockIdx.x * 64 + threadIdx.x] = s;
        if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    }
}

// every workgroup (one per CU) streams the SAME `bytes` buffer `reps` times with 16-byte loads, `waves` waves: L2-resident after the first pass
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void fill_kernel(const f32x4* buf, int n16, int reps, float* out) {
    f32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; r++)
        for (int i = threadIdx.x; i < n16; i += blockDim.x * 8) {
            f32x4 a[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int k = i + u * blockDim.x; a[u] = buf[k < n16 ? k : i]; }
#pragma unroll
            for (int u = 0; u < 8; u++) acc += a[u];
        }
    out[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    float *in, *out; long long* cyc;
    CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    std::vector<float> h(1024, 1.0f);
    for (int i = 0; i < 1024; i++) h[i] = 1.0f + i * 1e-3f;
    CK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
    auto run = [&](int form) {
        for (int it = 0; it < 3; it++) {
            switch (form) {
#define L(F) case F: hipLaunchKernelGGL((chain_kernel<F>), dim3(64), dim3(256), 0, 0, in, out, cyc); break;
                L(F_ADD) L(F_FMA) L(F_ADD_DPP) L(F_FMA_DPP) L(F_ADD_SGPR) L(F_FMA_SGPR) L(F_FMA_READLANE) L(F_FMA_MOV_DPP)
#undef L
            }
        }
        CK(hipDeviceSynchronize());
        std::vector<long long> c(64);
        CK(hipMemcpy(c.data(), cyc, 64 * 8, hipMemcpyDeviceToHost));
        std::sort(c.begin(), c.end());
        printf("  %-34s %6.2f cycles per link (s_memtime ticks = shader cycles; median of 64 workgroups)\n", names[form], (double)c[32] / LINKS);
    };
    printf("dependent float chain, %d links, one wave per SIMD (4 waves per workgroup, 64 workgroups):\n", LINKS);
    for (int f = 0; f < F_COUNT; f++) run(f);

    {
        const size_t lds = (size_t)(32 * 516 + 512 + 128) * 4;
        CK(hipFuncSetAttribute((const void*)tile_chain_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void*)tile_chain_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int cnt : {256, 384, 512}) {
            std::vector<long long> c(64);
            for (int act : {64, 32}) {
                for (int it = 0; it < 3; it++) {
                    if (act == 64) hipLaunchKernelGGL(tile_chain_kernel<64>, dim3(64), dim3(256), lds, 0, in, out, cyc, cnt);
                    else hipLaunchKernelGGL(tile_chain_kernel<32>, dim3(64), dim3(256), lds, 0, in, out, cyc, cnt);
                }
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(c.data(), cyc, 64 * 8, hipMemcpyDeviceToHost));
                std::sort(c.begin(), c.end());
                printf("  p16_value_chain_tile, %d links from an LDS tile, %d lanes active: %lld cycles = %.2f per link\n", cnt, act, c[32], (double)c[32] / cnt);
            }
            for (int it = 0; it < 3; it++) hipLaunchKernelGGL(sum_chain_kernel, dim3(64), dim3(256), 0, 0, in, out, cyc, cnt);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(c.data(), cyc, 64 * 8, hipMemcpyDeviceToHost));
            std::sort(c.begin(), c.end());
            printf("  p16_seq_sum_wave,     %d links from LDS:         %lld cycles = %.2f per link\n", cnt, c[32], (double)c[32] / cnt);
        }
    }
    const size_t bytes = 1331200;   // 129 rows x 4096 columns x 2.5 B ~ the activation image of a prompt chunk (BF16 selectors)
    f32x4* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("per-CU fill rate of an L2-resident %.2f MB buffer (every one of 256 workgroups reads all of it, 16-byte loads, 8 in flight per lane):\n", bytes / 1e6);
    for (int waves : {1, 2, 4, 8}) {
        const int reps = 20;
        hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(waves * 64), 0, 0, buf, (int)(bytes / 16), 2, out);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(waves * 64), 0, 0, buf, (int)(bytes / 16), reps, out);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double per_cu = bytes * (double)reps / (ms * 1e-3) / 1e9;
        printf("  %d wave(s) per CU: %7.1f GB/s per CU = %5.1f B/clk at 2.4 GHz, %6.2f TB/s over 256 CUs\n", waves, per_cu, per_cu / 2.4, per_cu * 256 / 1e3);
    }
    return 0;
}
