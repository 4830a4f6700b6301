// gemm_t16_lab: timing of the reference-order prompt GEMM on the F16 MFMA (jh_t16.h: gemm_t16_kernel) on the Llama-3-8B shapes,
// synthetic operands (timing only: bit-identity is what tests/test_gpu_parity.py checks through the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../jlama_amd/csrc/jh_t16.h"
using namespace jh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static int g_x = 0;   // argv[1] = x: the 32x32x16 form (gemm_t16x_kernel)
template <int EPI, int MT, int CW>
static void run(const char* tag, int N, int K, int M, int layers) {
    const int nblk = K / 32, ntiles = (EPI == EPI_SILU_MUL) ? N / 8 : N / 16, rowsw = ntiles * 16;
    i32x4* w; f32x4t* ws; i32x4* asel; float *ad, *out, *resid;
    const size_t wb = t16_w_bytes(rowsw, K), sb = t16_s_bytes(rowsw, K);
    CK(hipMalloc(&w, wb * layers)); CK(hipMalloc(&ws, sb * layers));
    CK(hipMemset(w, 0x5a, wb * layers)); CK(hipMemset(ws, 0, sb * layers));
    CK(hipMalloc(&asel, (size_t)M * nblk * 256)); CK(hipMemset(asel, 0, (size_t)M * nblk * 256));
    CK(hipMalloc(&ad, (size_t)nblk * 256 * 4)); CK(hipMemset(ad, 0, (size_t)nblk * 256 * 4));
    CK(hipMalloc(&out, (size_t)M * N * 4)); CK(hipMalloc(&resid, (size_t)M * N * 4)); CK(hipMemset(resid, 0, (size_t)M * N * 4));
    const int nslices = (ntiles + 2 * CW - 1) / (2 * CW), nrt = (M + MT - 1) / MT;
    const size_t lds = lds_bytes_gemm_t16(MT);
    CK(hipFuncSetAttribute((const void*)gemm_t16_kernel<EPI, MT, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)gemm_t16x_kernel<EPI, MT, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = ((nslices + 7) / 8) * 8 * nrt;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 3;
    for (int it = -1; it < reps; it++) {
        if (it == 0) CK(hipEventRecord(e0));
        for (int l = 0; l < layers; l++) {
            GemmT16Params g{(const i32x4*)((const char*)w + wb * l), (const f32x4t*)((const char*)ws + sb * l), ntiles, K, M, asel, ad, 256, out, N, resid, N, nslices, nrt};
            if (g_x) gemm_t16x_kernel<EPI, MT, CW><<<grid, CW * 64, lds>>>(g); else gemm_t16_kernel<EPI, MT, CW><<<grid, CW * 64, lds>>>(g);
        }
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    const double us = ms * 1e3 / (reps * layers);
    const double mfma = (double)((M + MT - 1) / MT * MT) * ntiles * nblk;          // MFMAs per launch
    printf("%-8s M %3d N %5d K %5d  MT %d CW %d grid %5d lds %6zu: %8.1f us   %.1f cycles/MFMA/SIMD at 2.1 GHz\n", tag, M, N, K, MT, CW, grid, lds, us,
           us * 1e-6 * 2.1e9 / (mfma / 1024.0));
    CK(hipFree(w)); CK(hipFree(ws)); CK(hipFree(asel)); CK(hipFree(ad)); CK(hipFree(out)); CK(hipFree(resid));
}
int main(int argc, char** argv) {
    g_x = argc > 1 && argv[1][0] == 'x';
    const int M = 129;
    run<EPI_SILU_MUL, 8, 4>("gate|up", 14336, 4096, M, 8);
    run<EPI_RESID, 8, 4>("down", 4096, 14336, M, 8);
    run<EPI_STORE, 8, 4>("q|k|v", 6144, 4096, M, 16);
    run<EPI_RESID, 8, 4>("o", 4096, 4096, M, 16);
    run<EPI_SILU_MUL, 8, 4>("gate|up", 14336, 4096, 128, 8);
    run<EPI_RESID, 8, 4>("down", 4096, 14336, 128, 8);
    return 0;
}
