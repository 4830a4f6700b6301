// gemv_lab2: cost of each prologue / epilogue of the REAL gemv_i8q4_kernel on a gate/up-sized matrix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../jlama_amd/csrc/jh_kernels.h"
using namespace jh;
template <int PRO, int EPI, int R, int NB, int PIPE>
void run(const char* tag, GemvParams p, const uint8_t* w, const float* ws, int nrows_total, int layers, int waves, int grid_override = 0) {
    const size_t lw = (size_t)nrows_total * p.ldb, ls = (size_t)nrows_total * p.ldbf;
    const int total = (EPI == EPI_SILU_MUL) ? 2 * p.nrows : p.nrows;
    const int ngroups = total / R;
    int grid = (ngroups + waves - 1) / waves;
    if (grid_override) grid = grid_override;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t lds = lds_bytes_i8(p.K);
    for (int it = -1; it < 3; it++) {
        if (it == 0) (void)hipEventRecord(e0);
        for (int l = 0; l < layers; l++) {
            GemvParams q = p;
            q.w = w + l * lw; q.ws = ws + l * ls;
            if (EPI == EPI_SILU_MUL) { q.w2 = q.w + (size_t)p.nrows * p.ldb; q.ws2 = q.ws + (size_t)p.nrows * p.ldbf; }
            gemv_i8q4_kernel<PRO, EPI, R, NB, PIPE><<<grid, waves * 64, lds>>>(q);
        }
    }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / (3 * layers), bytes = (double)nrows_total * p.K * 0.625;
    printf("%-34s R %2d waves %2d grid %4d pipe %d: %7.2f us  %6.0f GB/s  (%s)\n", tag, R, waves, grid, PIPE, us, bytes / us / 1e3, hipGetErrorString(hipGetLastError()));
}
int main() {
    const int nrows = 28672, K = 4096, layers = 16, H = 14336;
    uint8_t* w; float *ws, *out, *x, *nw, *ad; int8_t* aq;
    (void)hipMalloc(&w, (size_t)layers * nrows * K / 2); (void)hipMalloc(&ws, (size_t)layers * nrows * K / 32 * 4);
    (void)hipMalloc(&out, nrows * 4); (void)hipMalloc(&x, K * 4); (void)hipMalloc(&nw, K * 4); (void)hipMalloc(&ad, K / 32 * 4); (void)hipMalloc(&aq, K);
    (void)hipMemset(w, 0x37, (size_t)layers * nrows * K / 2); (void)hipMemset(ws, 0, (size_t)layers * nrows * K / 32 * 4);
    (void)hipMemset(x, 0, K * 4); (void)hipMemset(nw, 0, K * 4); (void)hipMemset(ad, 0, K / 32 * 4); (void)hipMemset(aq, 1, K);
    GemvParams p; memset(&p, 0, sizeof(p));
    p.K = K; p.ldb = K / 2; p.ldbf = K / 32; p.x = x; p.nw = nw; p.eps = 1e-5f; p.aq = aq; p.ad = ad; p.out = out; p.resid = x;
    p.nrows = nrows;
    run<PRO_Q8, EPI_STORE, 4, 2, 1>("Q8, store, pipe", p, w, ws, nrows, layers, 8, 256);
    run<PRO_RMS_Q8, EPI_STORE, 4, 2, 1>("rms, store, pipe", p, w, ws, nrows, layers, 8, 256);
    run<PRO_RMS_Q8, EPI_STORE, 4, 2, 1>("rms, store, pipe", p, w, ws, nrows, layers, 8, 512);
    run<PRO_RMS_Q8, EPI_STORE, 4, 2, 1>("rms, store, pipe", p, w, ws, nrows, layers, 4, 512);
    run<PRO_RMS_Q8, EPI_STORE, 4, 2, 1>("rms, store, pipe", p, w, ws, nrows, layers, 4, 1024);
    run<PRO_RMS_Q8, EPI_STORE, 2, 2, 1>("rms, store, pipe", p, w, ws, nrows, layers, 8, 256);
    run<PRO_RMS_Q8, EPI_STORE, 2, 2, 1>("rms, store, pipe", p, w, ws, nrows, layers, 8, 512);
    p.nrows = H;
    run<PRO_RMS_Q8, EPI_SILU_MUL, 4, 2, 1>("rms, silu*mul, pipe", p, w, ws, nrows, layers, 8, 256);
    run<PRO_RMS_Q8, EPI_SILU_MUL, 4, 2, 1>("rms, silu*mul, pipe", p, w, ws, nrows, layers, 8, 512);
    run<PRO_RMS_Q8, EPI_SILU_MUL, 2, 2, 1>("rms, silu*mul, pipe", p, w, ws, nrows, layers, 8, 512);
    return 0;
}
