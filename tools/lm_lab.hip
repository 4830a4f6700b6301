// lm_lab.hip -- where do the LM head's 68 us go?  The production F32 x Q4 GEMV (gemv_f32q4_kernel, 128256 x 4096) against the
// I8 x Q4 kernel on the SAME matrix (same loads, ~5x less VALU work per byte) and against launch shapes of both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/lm_lab.hip -o tools/lm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../jlama_amd/csrc/jh_kernels.h"
using namespace jh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int K = 4096, N = 128256, nblk = K / 32, COPIES = 3;   // 3 x 295 MB > the 256 MiB Infinity Cache
    const size_t wb = (size_t)N * nblk * 16, sb = (size_t)N * nblk * 4;
    std::vector<uint8_t> hw(wb);
    std::vector<float> hs(sb / 4), hx(K), hn(K, 1.0f);
    srand(5);
    for (auto& b : hw) b = (uint8_t)(rand() & 0xff);
    for (auto& f : hs) f = (float)((rand() % 2000) - 1000) * 1e-4f;
    for (auto& f : hx) f = (float)((rand() % 2000) - 1000) * 1e-3f;
    uint8_t* dw; float *ds, *dx, *dn, *out, *av; int* ai;
    CK(hipMalloc(&dw, wb * COPIES)); CK(hipMalloc(&ds, sb * COPIES)); CK(hipMalloc(&dx, K * 4)); CK(hipMalloc(&dn, K * 4));
    CK(hipMalloc(&out, (size_t)N * 4)); CK(hipMalloc(&av, 4096 * 4)); CK(hipMalloc(&ai, 4096 * 4));
    for (int c = 0; c < COPIES; c++) { CK(hipMemcpy(dw + c * wb, hw.data(), wb, hipMemcpyHostToDevice)); CK(hipMemcpy((char*)ds + c * sb, hs.data(), sb, hipMemcpyHostToDevice)); }
    CK(hipMemcpy(dx, hx.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, hn.data(), K * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    GemvParams p; memset(&p, 0, sizeof(p));
    p.nrows = N; p.K = K; p.ldb = K / 2; p.ldbf = nblk; p.x = dx; p.nw = dn; p.eps = 1e-5f; p.out = out; p.amax_part = av; p.amax_idx = ai;
    auto timeit = [&](const char* name, auto launch) {
        for (int it = -1; it < 5; it++) {
            if (it == 0) CK(hipEventRecord(e0, st));
            for (int c = 0; c < COPIES; c++) { p.w = dw + c * wb; p.ws = (const float*)((const char*)ds + c * sb); launch(); }
        }
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (5 * COPIES);
        printf("%-70s %7.2f us  %5.2f TB/s\n", name, us, (wb + sb) / 1e6 / us);
    };
    const size_t ldsf = lds_bytes_f32(K), ldsi = lds_bytes_i8(K);
    CK(hipFuncSetAttribute((const void*)gemv_f32q4_kernel<PRO_RMS_F32, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf));
    CK(hipFuncSetAttribute((const void*)gemv_f32q4_kernel<PRO_RMS_F32, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf));
    CK(hipFuncSetAttribute((const void*)gemv_f32q4_kernel<PRO_RMS_F32, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf));
    for (int round = 0; round < 3; round++) {
        printf("-- round %d\n", round);
        p.amax_part = av;
        timeit("F32xQ4 production R=2 NB=2, grid 512 x 8 waves", [&] { hipLaunchKernelGGL((gemv_f32q4_kernel<PRO_RMS_F32, 2, 2>), dim3(512), dim3(512), ldsf, st, p); });
        timeit("F32xQ4 R=2 NB=2, grid 1024 x 4 waves", [&] { hipLaunchKernelGGL((gemv_f32q4_kernel<PRO_RMS_F32, 2, 2>), dim3(1024), dim3(256), ldsf, st, p); });
        timeit("F32xQ4 R=1 NB=2, grid 512 x 8 waves", [&] { hipLaunchKernelGGL((gemv_f32q4_kernel<PRO_RMS_F32, 1, 2>), dim3(512), dim3(512), ldsf, st, p); });
        timeit("F32xQ4 R=1 NB=2, grid 768 x 8 waves", [&] { hipLaunchKernelGGL((gemv_f32q4_kernel<PRO_RMS_F32, 1, 2>), dim3(768), dim3(512), ldsf, st, p); });
        timeit("F32xQ4 R=1 NB=2, grid 1024 x 8 waves", [&] { hipLaunchKernelGGL((gemv_f32q4_kernel<PRO_RMS_F32, 1, 2>), dim3(1024), dim3(512), ldsf, st, p); });
        timeit("F32xQ4 R=4 NB=2, grid 512 x 4 waves", [&] { hipLaunchKernelGGL((gemv_f32q4_kernel<PRO_RMS_F32, 4, 2>), dim3(512), dim3(256), ldsf, st, p); });
        p.amax_part = nullptr;
        timeit("F32xQ4 R=2 NB=2 without the argmax partials, grid 512 x 8", [&] { hipLaunchKernelGGL((gemv_f32q4_kernel<PRO_RMS_F32, 2, 2>), dim3(512), dim3(512), ldsf, st, p); });
        timeit("I8xQ4 kernel (PRO_RMS_Q8, pipelined R=2 NB=2), grid 256 x 8 waves", [&] { hipLaunchKernelGGL((gemv_i8q4_kernel<PRO_RMS_Q8, EPI_STORE, 2, 2, 1>), dim3(256), dim3(512), ldsi, st, p); });
    }
    return 0;
}
