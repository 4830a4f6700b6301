// prefetch_lab: can a concurrent "touch" kernel on a second stream pull the NEXT GEMV's weights into the 256 MiB
// Infinity Cache while the current (latency-bound) kernels run, and how fast does the GEMV stream from the cache?
//   mode 0: [gap][gemv W_l] on one stream, weights from HBM (baseline)
//   mode 1: same, but stream B touches W_l during gap(l) (gemv(l) waits for the touch)
//   mode 2: same as 1 but the touch of W_{l+1} also overlaps gemv(l)  (continuous prefetch, one layer ahead)
//   mode 3: gemv on ONE resident layer (pure cache-hit timing, upper bound)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../jlama_amd/csrc/jh_kernels.h"
using namespace jh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int R, int NB>
__global__ __launch_bounds__(512) void gemv(const uint8_t* w, const float* ws, float* out, int nrows, int ldb, int ldbf) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const i32x4 al = {0x01020304 + lane, 0x05060708, 0x01010101, 0x02020202}, ah = {0x03030303, 0x7f7f7f7f, 0x01020304, lane};
    for (int g = blockIdx.x * nw + wave; g * R < nrows; g += gridDim.x * nw) {
        i32x4 wv[R][NB]; float sv[R][NB];
        const uint8_t* wb = w + (size_t)g * R * ldb; const float* sb = ws + (size_t)g * R * ldbf;
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int i = 0; i < NB; i++) {
                wv[r][i] = __builtin_nontemporal_load((const i32x4*)(wb + (size_t)r * ldb) + lane + 64 * i);
                sv[r][i] = __builtin_nontemporal_load(sb + (size_t)r * ldbf + lane + 64 * i);
            }
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) { acc[r] = 0.f;
#pragma unroll
            for (int i = 0; i < NB; i++) { const int isum = q4_block_dot(wv[r][i], al, ah) - 8 * 77; acc[r] = fmaf(0.5f * sv[r][i], (float)isum, acc[r]); } }
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]);
        float v = 0;
#pragma unroll
        for (int r = 0; r < R; r++) if (lane == r) v = acc[r];
        if (lane < R) out[g * R + lane] = v;
    }
}
// touch: read every 16 bytes (NT = 0 plain loads, 1 non-temporal), keep nothing
template <int NT>
__global__ __launch_bounds__(256) void touch(const i32x4* p, size_t n16, int* sink) {
    int x = 0;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        i32x4 a, b, c, d;
        if (NT) { a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(p + i + stride); c = __builtin_nontemporal_load(p + i + 2 * stride); d = __builtin_nontemporal_load(p + i + 3 * stride); }
        else { a = p[i]; b = p[i + stride]; c = p[i + 2 * stride]; d = p[i + 3 * stride]; }
        x ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    for (; i < n16; i += stride) x ^= p[i].x;
    if (x == 0x12345678) *sink = x;
}
__global__ void gap(int us) {   // a latency-bound phase: 8 workgroups idle for `us` microseconds, no memory traffic
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)us * 100) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv) {
    const int nrows = 28672, K = 4096, layers = 24;
    const int gap_us = argc > 1 ? atoi(argv[1]) : 12;
    const int ldb = K / 2, ldbf = K / 32;
    const size_t lw = (size_t)nrows * ldb, ls = (size_t)nrows * ldbf * 4, lbytes = lw + ls;   // one layer: nibbles | scales, contiguous
    uint8_t* base; float* out; int* sink;
    CK(hipMalloc(&base, lbytes * layers)); CK(hipMalloc(&out, nrows * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(base, 0x37, lbytes * layers));
    hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    constexpr int R = 8, NB = 2;
    const int waves = 8, grid = nrows / R / waves;
    auto W = [&](int l) { return base + (size_t)l * lbytes; };
    auto S = [&](int l) { return (const float*)(base + (size_t)l * lbytes + lw); };
    std::vector<hipEvent_t> ev(4 * layers + 8);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    for (int nt = 0; nt < 2; nt++)
    for (int tgrid : {256, 1024})
    for (int mode = 0; mode < 4; mode++) {
        if ((mode == 0 || mode == 3) && (nt || tgrid != 256)) continue;
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(A, hipStreamCaptureModeGlobal));
        int e = 0;
        auto fork = [&]() { CK(hipEventRecord(ev[e], A)); CK(hipStreamWaitEvent(B, ev[e], 0)); e++; };
        auto join = [&]() { CK(hipEventRecord(ev[e], B)); CK(hipStreamWaitEvent(A, ev[e], 0)); e++; };
        auto do_touch = [&](int l) { if (nt) touch<1><<<tgrid, 256, 0, B>>>((const i32x4*)W(l), lbytes / 16, sink); else touch<0><<<tgrid, 256, 0, B>>>((const i32x4*)W(l), lbytes / 16, sink); };
        if (mode == 2) { fork(); do_touch(0); }
        for (int l = 0; l < layers; l++) {
            if (mode == 1) { fork(); do_touch(l); }
            gap<<<8, 64, 0, A>>>(gap_us);
            if (mode == 1) join();
            if (mode == 2) { join(); if (l + 1 < layers) { fork(); do_touch(l + 1); } }   // touch(l+1) overlaps gemv(l) and gap(l+1)
            const int wl = mode == 3 ? 0 : l;
            gemv<R, NB><<<grid, waves * 64, 0, A>>>(W(wl), S(wl), out, nrows, ldb, ldbf);
        }
        CK(hipStreamEndCapture(A, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        CK(hipGraphLaunch(exec, A)); CK(hipStreamSynchronize(A));
        CK(hipEventRecord(t0, A));
        for (int it = 0; it < 3; it++) CK(hipGraphLaunch(exec, A));
        CK(hipEventRecord(t1, A)); CK(hipStreamSynchronize(A));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        const double us = ms * 1e3 / (3 * layers);
        printf("mode %d nt %d touch-grid %4d gap %2d us: %7.2f us/layer  (gemv+sync part %6.2f us => %5.0f GB/s effective)\n", mode, nt, tgrid, gap_us, us,
               us - gap_us, lbytes / (us - gap_us) / 1e3);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    return 0;
}
