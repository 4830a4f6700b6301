// tools/gemm_bf16_lab.hip -- the BF16 prompt GEMM at M = 129 (Mistral-7B prefill shapes) in two forms, for round 5:
//   base  gemm_bf16_lds_kernel (jh_kernels.h): every wave requests its weight fragments AND its share of the A chunk
//   ring  gemm_bf16_ring_lab: a fifth LOADER wave stages A (global -> registers one iteration ahead -> LDS double buffer), the
//         MFMA waves request weights only, PW chunks of 4 fragments in flight; A fragments are read a slice ahead of their MFMAs.
// Vector-memory loads retire in order, so in `base` the wait for the young A loads (L2) is a wait for every older weight load
// (HBM) too; `ring` separates the two streams.  Round 4 measured (profiles/NOTEBOOK_rounds_1-4.md 8.3): K walk 770 -> 456 cycles per k slice, gate|up
// un-split 90 -> 61 us, but with the best K split per shape only 6 % per layer -- and WITHOUT any global load the loop still takes
// 45 us (knock-outs below), i.e. the kernel is within 1.4x of its matrix-pipe time at 160 padded rows.
// Results of ring are compared bit for bit with base (same MFMA order).  knock: 1 no A loads, 2 no W loads, 4 no barriers, 8 A
// fragments read for slice 0 only (argv[1]: 0, 3 or 15 are built).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/gemm_bf16_lab.hip -o tools/gemm_bf16_lab
#include "../jlama_amd/csrc/jh_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
using namespace jh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int MT, int CWB, int PW, int knock>   // knock is a compile-time constant: a run-time predicate on the loads costs the whole gain
__global__ __launch_bounds__((CWB + 1) * 64) void gemm_bf16_ring_lab(MfmaBf16TileParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLC = 4, PIECES = MT * SLC;
    i32x4* ring = (i32x4*)smem;                              // [2][MT][SLC][64] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nks = p.k / 16, nsplit = p.nsplit > 1 ? p.nsplit : 1;
    const int per = nks / nsplit, s0 = blockIdx.y * per;     // per % (PW * SLC) == 0
    const int nchunks = per / SLC;
    constexpr bool bar = !(knock & 4);
    if (wv == CWB) {
        const i32x4* ap = (const i32x4*)p.a + (size_t)s0 * 64 + lane;
        const size_t a_rt = (size_t)nks * 64;
        i32x4 r0[PIECES], r1[PIECES];
        auto request = [&](i32x4 (&r)[PIECES], int c) __attribute__((always_inline)) {
            c = c < nchunks ? c : nchunks - 1;
#pragma unroll
            for (int pc = 0; pc < PIECES; pc++) {
                const int t = pc / SLC, q = pc - t * SLC;
                if constexpr (!(knock & 1)) r[pc] = ap[(size_t)t * a_rt + (size_t)(c * SLC + q) * 64];
            }
        };
        auto file = [&](const i32x4 (&r)[PIECES], int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int pc = 0; pc < PIECES; pc++) ring[((size_t)buf * PIECES + pc) * 64 + lane] = r[pc];
        };
        request(r0, 0);
        request(r1, 1);
        file(r0, 0);
        lds_barrier();
        for (int c = 0; c < nchunks; c += 2) {
            request(r0, c + 2);
            file(r1, 1);
            if (bar) lds_barrier();
            request(r1, c + 3);
            file(r0, 0);
            if (bar) lds_barrier();
        }
        return;
    }
    const int nl = lane & 31, h = lane >> 5;
    const int ct = blockIdx.x * CWB + wv;
    const i32x4* wp = (const i32x4*)p.w + ((size_t)ct * nks + s0) * 64 + lane;
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    const int last = per - 1;
    i32x4 wr[PW][SLC];
    auto load_w = [&](i32x4 (&w)[SLC], int sl0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < SLC; q++) {
            int sl = sl0 + q;
            sl = sl < last ? sl : last;
            if constexpr (!(knock & 2)) w[q] = __builtin_nontemporal_load(wp + (size_t)sl * 64);
        }
    };
    auto read_a = [&](i32x4 (&av)[MT], int buf, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < MT; t++) if (!(knock & 8) || q == 0) av[t] = ring[((size_t)buf * PIECES + t * SLC + q) * 64 + lane];
    };
    auto mma_chunk = [&](int buf, const i32x4 (&w)[SLC]) __attribute__((always_inline)) {
        i32x4 a0[MT], a1[MT];
        read_a(a0, buf, 0);
#pragma unroll
        for (int q = 0; q < SLC; q += 2) {
            read_a(a1, buf, q + 1);
            __builtin_amdgcn_sched_barrier(0);
            {
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, w[q]);
#pragma unroll
                for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[t]), bfrag, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q + 2 < SLC) read_a(a0, buf, q + 2);
            __builtin_amdgcn_sched_barrier(0);
            {
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, w[q + 1]);
#pragma unroll
                for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1[t]), bfrag, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int i = 0; i < PW; i++) load_w(wr[i], i * SLC);
    lds_barrier();
    for (int c = 0; c < nchunks; c += PW) {
#pragma unroll
        for (int i = 0; i < PW; i++) {
            mma_chunk(i & 1, wr[i]);
            load_w(wr[i], (c + i + PW) * SLC);
            if (bar) lds_barrier();
        }
    }
    const int ncol = ct * 32 + nl;
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int mrow = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mrow < p.m) {
                if (nsplit > 1) p.ws[((size_t)blockIdx.y * p.m + mrow) * p.n + ncol] = acc[t][r];
                else p.c[(size_t)p.ldc * mrow + ncol] = acc[t][r];
            }
        }
}

int main(int argc, char** argv) {
    const int knock = argc > 1 ? atoi(argv[1]) : 0;
    constexpr int MT = 5, CWB = 4, M = 129;
    const int shapes[4][2] = {{6144, 4096}, {4096, 4096}, {28672, 4096}, {4096, 14336}};
    std::mt19937 rng(1);
    for (auto& sh : shapes) {
        const int N = sh[0], K = sh[1], nks = K / 16;
        const size_t wbytes = (size_t)N * K * 2, abytes = (size_t)MT * 32 * K * 2;
        const int copies = (int)((600u << 20) / wbytes) + 1;               // rotate through > 256 MiB of weights
        std::vector<uint16_t> hw(wbytes / 2), ha(abytes / 2);
        for (auto& x : hw) x = (uint16_t)(0x3c00 + (rng() & 0x1ff));       // bf16 values near 0.01 .. 0.03: no overflow over K
        for (auto& x : ha) x = (uint16_t)(0x3c00 + (rng() & 0x1ff));
        uint16_t *w, *a; float *c0, *c1, *ws;
        CK(hipMalloc(&w, wbytes * copies)); CK(hipMalloc(&a, abytes)); CK(hipMalloc(&c0, (size_t)M * N * 4)); CK(hipMalloc(&c1, (size_t)M * N * 4));
        CK(hipMalloc(&ws, (size_t)8 * M * N * 4));
        for (int l = 0; l < copies; l++) CK(hipMemcpy((char*)w + l * wbytes, hw.data(), wbytes, hipMemcpyHostToDevice));
        CK(hipMemcpy(a, ha.data(), abytes, hipMemcpyHostToDevice));
        const size_t lds = (size_t)2 * MT * 4 * 1024;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_ring_lab<MT, CWB, 4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_ring_lab<MT, CWB, 4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_ring_lab<MT, CWB, 4, 15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int S : {1, 2, 4}) {
            if ((nks / S) % 16) continue;
            const dim3 grid(N / 32 / CWB, S);
            float ms[2] = {0, 0};
            for (int form = 0; form < 2; form++) {
                for (int it = -1; it < 3; it++) {
                    if (it == 0) CK(hipEventRecord(e0, 0));
                    for (int l = 0; l < copies; l++) {
                        MfmaBf16TileParams g{a, (const uint16_t*)((const char*)w + l * wbytes), form ? c1 : c0, nullptr, M, N, K, N, ws, S};
                        if (form == 0) hipLaunchKernelGGL((gemm_bf16_lds_kernel<MT, CWB>), grid, dim3(CWB * 64), lds, 0, g);
                        else if (knock == 0) hipLaunchKernelGGL((gemm_bf16_ring_lab<MT, CWB, 4, 0>), grid, dim3((CWB + 1) * 64), lds, 0, g);
                        else if (knock == 3) hipLaunchKernelGGL((gemm_bf16_ring_lab<MT, CWB, 4, 3>), grid, dim3((CWB + 1) * 64), lds, 0, g);
                        else hipLaunchKernelGGL((gemm_bf16_ring_lab<MT, CWB, 4, 15>), grid, dim3((CWB + 1) * 64), lds, 0, g);
                    }
                }
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms[form], e0, e1));
                ms[form] /= 3.0f * copies;
            }
            int bad = -1;
            if (S == 1 && knock == 0) {
                std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
                CK(hipMemcpy(h0.data(), c0, h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), c1, h1.size() * 4, hipMemcpyDeviceToHost));
                bad = memcmp(h0.data(), h1.data(), h0.size() * 4) != 0;
            }
            printf("M=%d N=%5d K=%5d S=%d (no reduce pass timed): base %7.1f us  ring %7.1f us (%.2f TB/s of weights)%s\n", M, N, K, S, ms[0] * 1e3, ms[1] * 1e3,
                   wbytes / (ms[1] * 1e-3) / 1e12, bad < 0 ? "" : bad ? "  RESULTS DIFFER" : "  results bit-identical");
        }
        hipFree(w); hipFree(a); hipFree(c0); hipFree(c1); hipFree(ws);
    }
    return 0;
}
