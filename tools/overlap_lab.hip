// tools/overlap_lab.hip -- VERDICT r4 item 2: can the per-launch fixed cost of the decode GEMVs (launch edge + ramp + activation round
// trip + prologue, ~3.5 us x 160 launches of a 1.7 ms token) be hidden by making kernel n+1 CO-RESIDENT with kernel n?
//
//   chain    per layer three dependent I8 x Q4 GEMVs of the 8B shapes: o-proj 4096x4096 -> gate|up 2x14336x4096 (out = g*u) -> down
//            4096x14336, L layers of distinct weights (> 256 MiB: streamed from HBM), activations F32, Q8 prologue per workgroup
//   base     one captured stream: kernel k+1 starts when k has finished (what the product's token graph does)
//   overlap  two captured streams, even / odd kernels: the graph edge is k -> k+2, so k+1 is launched while k runs (1 workgroup per
//            CU each, both fit), requests its first weight groups at once (they do not depend on k), then waits for k's 256
//            per-workgroup flag words (write-through stores, cache-bypassing loads), re-reads the activation row with sc1 loads and
//            goes on.  Every wait is bounded (JH_LAB_TIMEOUT_US of the 100 MHz wall clock): a wedged meeting sets `fail`, never hangs.
//   knock    overlap without the wait (results wrong): what the overlap could buy at most.
// Kill criterion (VERDICT): ship only if a layer drops >= 10 %.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/overlap_lab.hip -o tools/overlap_lab
#include "../jlama_amd/csrc/jh_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace jh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct LabParams {
    const uint8_t* w; const float* ws;   // Q4 nibbles [nrows, K/2], scales [nrows, K/32]
    const float* x; float* out;          // activation row in (K), out (nout)
    int nrows, K, pair;                  // pair = 1: out[j] = dot(row 2j) * dot(row 2j+1)
    const unsigned* wait_flags;          // [256] flags of the predecessor (nullptr: none)
    unsigned* my_flags;                  // [256]
    const unsigned* seq;                 // replays so far
    int mode;                            // 0 base, 1 overlap (every workgroup polls the 256 flags), 2 overlap without the wait,
                                         // 3 overlap, ONE watcher wave polls the flags and raises a `go` word the other workgroups poll
    unsigned* go;                        // [1] per kernel (mode 3)
    int* fail;
    long long timeout_ticks;
};
__device__ __forceinline__ unsigned ld_sc1_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1_f32(float* p, float v) { __hip_atomic_store((unsigned*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NB, int R, int UM>
__global__ __launch_bounds__(512, 4) void lab_gemv(LabParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, nblk = K / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    const int rows_wg = p.nrows / gridDim.x, row0 = blockIdx.x * rows_wg;
    const int groups = rows_wg / R;            // row groups of this workgroup, dealt to the waves round-robin
    const unsigned want = *p.seq + 1u;
    // ---- the first weight group of every wave is requested BEFORE anything that depends on the predecessor
    i32x4 wq[R][NB]; float sq[R][NB];
    auto issue = [&](int g) __attribute__((always_inline)) {
        g = g < groups ? g : groups - 1;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const size_t row = (size_t)row0 + (size_t)g * R + r;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                wq[r][i] = __builtin_nontemporal_load((const i32x4*)(p.w + row * (K / 2)) + lane + 64 * i);
                sq[r][i] = __builtin_nontemporal_load(p.ws + row * nblk + lane + 64 * i);
            }
        }
    };
    if (p.mode != 0) issue(wave);
    // ---- wait for the predecessor's flags (wave 0 polls 256 words with one 16-byte sc1 load per lane)
    if (p.mode == 1 && p.wait_flags) {
        if (wave == 0) {
            const long long t0 = wall_clock64();
            for (;;) {
                const unsigned a = ld_sc1_u32(p.wait_flags + lane * 4), b = ld_sc1_u32(p.wait_flags + lane * 4 + 1);
                const unsigned c = ld_sc1_u32(p.wait_flags + lane * 4 + 2), d = ld_sc1_u32(p.wait_flags + lane * 4 + 3);
                const bool ok = a == want && b == want && c == want && d == want;
                if (__all(ok)) break;
                if (wall_clock64() - t0 > p.timeout_ticks) { if (lane == 0) *p.fail = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
    if (p.mode == 3 && p.wait_flags) {
        if (blockIdx.x == 0 && wave == 1) {          // the watcher: alone on the 8 flag lines
            const long long t0 = wall_clock64();
            for (;;) {
                const unsigned a = ld_sc1_u32(p.wait_flags + lane * 4), b = ld_sc1_u32(p.wait_flags + lane * 4 + 1);
                const unsigned c = ld_sc1_u32(p.wait_flags + lane * 4 + 2), d = ld_sc1_u32(p.wait_flags + lane * 4 + 3);
                if (__all(a == want && b == want && c == want && d == want)) break;
                if (wall_clock64() - t0 > p.timeout_ticks) { if (lane == 0) *p.fail = 1; break; }
            }
            if (lane == 0) __hip_atomic_store(p.go, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wave == 0) {                              // everybody: one word, one request per round
            const long long t0 = wall_clock64();
            while (ld_sc1_u32(p.go) != want) {
                if (wall_clock64() - t0 > p.timeout_ticks) { if (lane == 0) *p.fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    // ---- prologue as in the product (quad_quantize_store: 8 elements per thread, 4 lanes per Q8 block, Panama rule); UM units per
    // thread held in registers, one round trip
    {
        const ActI8 a = carve_i8(smem, nblk);
        const int units = K / 8;
        float xv[UM][8];
#pragma unroll
        for (int u = 0; u < UM; u++) {
            int unit = tid + u * 512;
            unit = unit < units ? unit : units - 1;
            f32x4 xa, xb;
            if (p.mode != 0) {
                asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1" : "=&v"(xa), "=&v"(xb) : "v"(p.x + unit * 8) : "memory");
            } else { xa = *(const f32x4*)(p.x + unit * 8); xb = *(const f32x4*)(p.x + unit * 8 + 4); }
            xv[u][0] = xa.x; xv[u][1] = xa.y; xv[u][2] = xa.z; xv[u][3] = xa.w; xv[u][4] = xb.x; xv[u][5] = xb.y; xv[u][6] = xb.z; xv[u][7] = xb.w;
        }
        if (p.mode == 0) issue(wave);                  // product order: activation row first, then the weights (vmcnt retires oldest-first)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < UM; u++) {
            const int unit = tid + u * 512;
            if (unit < units) quad_quantize_store(xv[u], unit, a);
        }
    }
    lds_barrier();
    i32x4* alo = (i32x4*)smem; i32x4* ahi = alo + nblk; float* ad = (float*)(ahi + nblk); int* asum = (int*)(ad + nblk);
    // ---- stream: the wave's next group is requested before the current one is reduced (two register sets, as the product's PIPE form)
    for (int g = wave; g < groups; g += nw) {
        i32x4 cw[R][NB]; float cs[R][NB];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int i = 0; i < NB; i++) { cw[r][i] = wq[r][i]; cs[r][i] = sq[r][i]; }
        if (g + nw < groups) issue(g + nw);
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const i32x4 al = alo[lane + 64 * i], ah = ahi[lane + 64 * i];
            const float da = ad[lane + 64 * i];
            const int sa = asum[lane + 64 * i];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int isum = q4_block_dot(cw[r][i], al, ah) - 8 * sa;
                acc[r] = fmaf(da * cs[r][i], (float)isum, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]);
        if (lane == 0) {
            if (p.pair) {
#pragma unroll
                for (int r = 0; r < R; r += 2) {
                    const float v = acc[r] * acc[r + 1];
                    float* dst = p.out + (row0 + g * R + r) / 2;
                    if (p.mode != 0) st_sc1_f32(dst, v); else *dst = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    float* dst = p.out + row0 + g * R + r;
                    if (p.mode != 0) st_sc1_f32(dst, acc[r]); else *dst = acc[r];
                }
            }
        }
    }
    if (p.mode != 0) {   // write-through stores performed, then this workgroup's flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.my_flags + blockIdx.x, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void bump_seq(unsigned* seq) { *seq = *seq + 1u; }

struct Layer { uint8_t *wo, *wg, *wd; float *so, *sg, *sd; };

int main(int argc, char** argv) {
    const int E = 4096, H = 14336, L = argc > 1 ? atoi(argv[1]) : 8, REPS = 20;
    const long long timeout_us = getenv("JH_LAB_TIMEOUT_US") ? atoll(getenv("JH_LAB_TIMEOUT_US")) : 20000;
    std::vector<Layer> ly(L);
    const size_t nbig = (size_t)2 * H * E / 2;
    std::vector<uint8_t> hw(nbig);
    unsigned lcg = 12345u;
    for (auto& b : hw) { lcg = lcg * 1664525u + 1013904223u; b = (uint8_t)(lcg >> 24); }
    auto upload = [&](uint8_t** w, float** sc, size_t rows, size_t K, float scale) {
        CK(hipMalloc(w, rows * K / 2)); CK(hipMemcpy(*w, hw.data(), rows * K / 2, hipMemcpyHostToDevice));
        std::vector<float> hs(rows * (K / 32), scale);
        CK(hipMalloc(sc, hs.size() * 4)); CK(hipMemcpy(*sc, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    };
    for (auto& l : ly) {   // the same random image in every layer, at distinct addresses (streamed from HBM all the same)
        upload(&l.wo, &l.so, E, E, 0.004f); upload(&l.wg, &l.sg, (size_t)2 * H, E, 0.008f); upload(&l.wd, &l.sd, E, H, 0.002f);
    }
    float *xa, *xb, *xh; unsigned *flags, *seq, *go; int* fail;
    CK(hipMalloc(&xa, E * 4)); CK(hipMalloc(&xb, E * 4)); CK(hipMalloc(&xh, H * 4));
    std::vector<float> hx(E);
    for (int i = 0; i < E; i++) hx[i] = 0.01f * (float)((i * 37) % 101 - 50);
    const int NK = 3 * L;
    CK(hipMalloc(&flags, (size_t)NK * 256 * 4)); CK(hipMalloc(&seq, 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&go, (size_t)NK * 256));
    CK(hipMemset(go, 0, (size_t)NK * 256));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t fork, join, t0, t1; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const size_t lds_e = lds_bytes_i8(E), lds_h = lds_bytes_i8(H);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_gemv<7, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
    std::vector<float> ref(E);
    for (int mode = 0; mode < 4; mode++) {
        CK(hipMemset(flags, 0, (size_t)NK * 256 * 4)); CK(hipMemset(seq, 0, 4)); CK(hipMemset(fail, 0, 4));
        CK(hipMemcpy(xa, hx.data(), E * 4, hipMemcpyHostToDevice));
        hipGraph_t graph; hipGraphExec_t exec;
        const bool nograph = getenv("JH_LAB_NOGRAPH") != nullptr;
        int k = 0;
        auto enqueue_all = [&]() {
        k = 0;
        if (mode != 0) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
        auto launch = [&](const uint8_t* w, const float* ws, const float* x, float* out, int nrows, int K, int pair) {
            LabParams p{w, ws, x, out, nrows, K, pair, k > 0 ? flags + (size_t)(k - 1) * 256 : nullptr, flags + (size_t)k * 256, seq, mode, go + (size_t)k * 64, fail, timeout_us * 100};
            hipStream_t st = (mode != 0 && (k & 1)) ? s1 : s0;
            if (K == E) hipLaunchKernelGGL((lab_gemv<2, 2, 1>), dim3(256), dim3(512), lds_e, st, p);
            else hipLaunchKernelGGL((lab_gemv<7, 1, 4>), dim3(256), dim3(512), lds_h, st, p);
            k++;
        };
        for (int l = 0; l < L; l++) {
            launch(ly[l].wo, ly[l].so, xa, xb, E, E, 0);          // "o-proj"   x  -> x'
            launch(ly[l].wg, ly[l].sg, xb, xh, 2 * H, E, 1);      // "gate|up"  x' -> h     (a buffer is rewritten two kernels after its last reader)
            launch(ly[l].wd, ly[l].sd, xh, xa, E, H, 0);          // "down"     h  -> x
        }
        if (mode != 0) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
        hipLaunchKernelGGL(bump_seq, dim3(1), dim3(1), 0, s0, seq);
        };
        if (!nograph) {
            CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
            enqueue_all();
            CK(hipStreamEndCapture(s0, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            CK(hipGraphLaunch(exec, s0));
        } else enqueue_all();
        CK(hipStreamSynchronize(s0));
        std::vector<float> got(E);
        CK(hipMemcpy(got.data(), xa, E * 4, hipMemcpyDeviceToHost));
        if (mode == 0) ref = got;
        const bool same = memcmp(ref.data(), got.data(), E * 4) == 0;
        CK(hipMemcpy(xa, hx.data(), E * 4, hipMemcpyHostToDevice));
        CK(hipEventRecord(t0, s0));
        for (int it = 0; it < REPS; it++) { if (nograph) enqueue_all(); else CK(hipGraphLaunch(exec, s0)); }
        CK(hipEventRecord(t1, s0)); CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, t0, t1));
        int hfail = 0; CK(hipMemcpy(&hfail, fail, 4, hipMemcpyDeviceToHost));
        const char* names[] = {"base (one stream, k -> k+1)", "overlap (two streams, flags)", "overlap, wait knocked out", "overlap, watcher + go word"};
        printf("%-32s %7.2f us per layer (3 GEMVs, %d layers)  first-run results %s%s\n", names[mode], ms * 1e3 / (REPS * L), L,
               mode == 0 ? "= reference" : same ? "bit-identical to base" : "DIFFER from base", hfail ? "  [a bounded wait TIMED OUT]" : "");
        if (!nograph) { CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); }
    }
    return 0;
}
