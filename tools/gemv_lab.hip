// gemv_lab: which stage of the Q4 GEMV limits bandwidth?  Variants: 0 loads only, 1 + integer dot, 2 + reduce/store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../jlama_amd/csrc/jh_kernels.h"
using namespace jh;
template <int VAR, int R, int NB, bool SCALES>
__global__ __launch_bounds__(1024) void lab(const uint8_t* w, const float* ws, float* out, int nrows, int ldb, int ldbf) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int g = blockIdx.x * nw + wave;
    if (g * R >= nrows) return;
    i32x4 wv[R][NB]; float sv[R][NB];
    const uint8_t* wb = w + (size_t)g * R * ldb; const float* sb = ws + (size_t)g * R * ldbf;
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int i = 0; i < NB; i++) {
            wv[r][i] = __builtin_nontemporal_load((const i32x4*)(wb + (size_t)r * ldb) + lane + 64 * i);
            sv[r][i] = SCALES ? __builtin_nontemporal_load(sb + (size_t)r * ldbf + lane + 64 * i) : 1.0f;
        }
    float acc[R];
    if (VAR == 0) {
#pragma unroll
        for (int r = 0; r < R; r++) { int x = 0;
#pragma unroll
            for (int i = 0; i < NB; i++) x ^= wv[r][i].x ^ wv[r][i].y ^ wv[r][i].z ^ wv[r][i].w ^ __float_as_int(sv[r][i]);
            acc[r] = __int_as_float(x); }
        int t = 0;
#pragma unroll
        for (int r = 0; r < R; r++) t ^= __float_as_int(acc[r]);
        if (t == 0x12345) out[g] = 1.0f;
        return;
    }
    const i32x4 al = {0x01020304 + lane, 0x05060708, 0x01010101, 0x02020202}, ah = {0x03030303, 0x7f7f7f7f, 0x01020304, lane};
#pragma unroll
    for (int r = 0; r < R; r++) { acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NB; i++) { const int isum = q4_block_dot(wv[r][i], al, ah) - 8 * 77; acc[r] = fmaf(0.5f * sv[r][i], (float)isum, acc[r]); } }
    if (VAR == 1) { float t = 0; 
#pragma unroll
        for (int r = 0; r < R; r++) t += acc[r];
        if (t == 12345.f) out[g] = t; return; }
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]);
    float v = 0;
#pragma unroll
    for (int r = 0; r < R; r++) if (lane == r) v = acc[r];
    if (lane < R) out[g * R + lane] = v;
}
template <int VAR, int R, int NB, bool SC>
void run(const uint8_t* w, const float* ws, float* out, int nrows, int K, int layers, int waves) {
    const int ldb = K / 2, ldbf = K / 32;
    const size_t lw = (size_t)nrows * ldb, ls = (size_t)nrows * ldbf;
    const int ngroups = nrows / R, grid = (ngroups + waves - 1) / waves;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = -1; it < 3; it++) {
        if (it == 0) (void)hipEventRecord(e0);
        for (int l = 0; l < layers; l++) lab<VAR, R, NB, SC><<<grid, waves * 64>>>(w + l * lw, ws + l * ls, out, nrows, ldb, ldbf);
    }
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / (3 * layers), bytes = (double)nrows * K * (SC ? 0.625 : 0.5);
    printf("var %d R %2d NB %d scales %d waves %2d grid %5d: %7.2f us  %6.0f GB/s\n", VAR, R, NB, (int)SC, waves, grid, us, bytes / us / 1e3);
}
int main(int argc, char** argv) {
    const int nrows = 28672, K = 4096, layers = 16;
    uint8_t* w; float* ws; float* out;
    (void)hipMalloc(&w, (size_t)layers * nrows * K / 2); (void)hipMalloc(&ws, (size_t)layers * nrows * K / 32 * 4); (void)hipMalloc(&out, nrows * 4);
    (void)hipMemset(w, 0x37, (size_t)layers * nrows * K / 2); (void)hipMemset(ws, 0, (size_t)layers * nrows * K / 32 * 4);
    for (int waves : {4, 8, 16}) {
        run<0, 4, 2, true>(w, ws, out, nrows, K, layers, waves);
        run<0, 8, 2, true>(w, ws, out, nrows, K, layers, waves);
        run<0, 8, 2, false>(w, ws, out, nrows, K, layers, waves);
        run<1, 8, 2, true>(w, ws, out, nrows, K, layers, waves);
        run<2, 8, 2, true>(w, ws, out, nrows, K, layers, waves);
        run<2, 4, 2, true>(w, ws, out, nrows, K, layers, waves);
        run<2, 2, 2, true>(w, ws, out, nrows, K, layers, waves);
    }
    run<0, 14, 2, true>(w, ws, out, nrows, K, layers, 8);
    run<2, 14, 2, true>(w, ws, out, nrows, K, layers, 8);
    return 0;
}
