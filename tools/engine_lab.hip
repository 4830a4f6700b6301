// engine_lab.hip -- does keeping the weight stream running ACROSS a phase boundary pay on this chip with these kernels?
// Two dependent I8 x Q4 GEMVs (o-projection-sized 4096x4096 -> gate/up-sized 28672x4096, the second consuming the Q8-quantized
// output of the first: an all-to-all hand-off) as
//   (A) two launches of the production register-staged kernel (fused Q8 prologue each), and
//   (B) ONE persistent launch: per workgroup one LDS-DMA loader wave streams its share of W1 and then W2 through a ring without
//       stopping; consumer waves compute phase 1, publish their outputs as 8-byte {value, epoch} granules (one sc1 store each),
//       one consumer wave per workgroup gathers all 4096 granules (sc1 loads, re-polling the missing ones), quantizes the row
//       into LDS and releases the others; by then the ring holds the first slots of W2.
// Results are bit-identical by construction (same lane->block map, fma order, wave reduction, quantizer).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/engine_lab.hip -o tools/engine_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../jlama_amd/csrc/jh_kernels.h"
using namespace jh;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct EngParams {
    const uint8_t* w1; const float* ws1; int n1;
    const uint8_t* w2; const float* ws2; int n2;
    const float* x;                 // [K] F32 input of phase 1
    unsigned long long* gran;       // [n1] {float bits, epoch}
    float* y2;                      // [n2]
    int K, epoch, slots, depth, mode;   // mode 1: no hand-off (phase 2 reuses the phase-1 activation) = the pure streaming time
    // round 6 knobs (MI355X_MICROARCH.md price list): `thin` -- gather-pass: 0 the loader refills as slots free up (its burst queues in
    // front of the gather's loads on this CU's memory path), 1 at most ONE fill outstanding while the workgroup gathers, 2 no new fill
    // at all while it gathers; `one_gatherer` -- the sweep by ONE consumer wave (the others wait) instead of a chunk per consumer
    int thin, one_gatherer;
    int* fail;                      // set when a bounded spin gave up
    long long* dbg;                 // optional [grid][8] wall-clock stamps (100 MHz)
};

template <int NT>
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, NT ? 2 : 0);
}
__device__ __forceinline__ unsigned lds_off(const volatile void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const volatile void*)p; }
__device__ __forceinline__ int flag_read(const volatile int* p) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_off(p)) : "memory");
    return v;
}
__device__ __forceinline__ void flag_write(volatile int* p, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(lds_off(p)), "v"(v) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_rt(int n) {
    switch (n) {
#define W(k) case k: wait_vm<k>(); break;
        W(0) W(20) W(40) W(60)
#undef W
        default: wait_vm<0>();
    }
}
constexpr int SPIN_MAX = 1 << 22;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// four 16-byte loads served by L2 (sc1), 1 KiB apart, one wait: each 16 bytes = two self-validating {value, epoch} granules
__device__ __forceinline__ void ld4_sc1_x4(const u32x4* p, u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                 "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:3072 sc1"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
}

// K = 4096 (NB = 2), 8 rows per slot = 16 KiB of nibbles + 4 KiB of scales = 20 one-KiB DMA loads
template <int NCONS>
__global__ __launch_bounds__((NCONS + 1) * 64) void engine2_kernel(EngParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int NB = 2, nblk = 128, RPS = 8, LOADS = 20, SLOT = LOADS * 1024, SLOT_W = 16 * 1024;
    constexpr int ACT_BYTES = nblk * 40 + 16 + 32 * 8 + 256 * 4;   // = lds_bytes_i8(4096)
    char* ring = smem;
    volatile int* ready = (volatile int*)(smem + (size_t)p.slots * SLOT);
    volatile int* freed = ready + 16;
    volatile int* act2_ready = freed + 16;
    volatile int* gathering = act2_ready + 1;               // 1 while this workgroup's consumers sweep the granules
    char* act1m = (char*)(act2_ready + 16);                 // ActI8 of the phase-1 input
    char* act2m = act1m + ACT_BYTES;               // ActI8 of the phase-2 input (= quantized phase-1 output)
    float* y1 = (float*)(act2m + ACT_BYTES);       // [4096] gathered phase-1 outputs
    const ActI8 a1 = carve_i8(act1m, nblk), a2 = carve_i8(act2m, nblk);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S1 = p.n1 / RPS / gridDim.x, S2 = p.n2 / RPS / gridDim.x, nslots = S1 + S2;   // host: exact divisions
    if (tid < 48) ((volatile int*)ready)[tid] = 0;
    // phase-1 input: Q8 quantize the F32 row (maybeQuantize), all waves: 512 units of 8
    for (int unit = tid; unit < p.K / 8; unit += blockDim.x) {
        const float4 xa = *(const float4*)(p.x + unit * 8), xb = *(const float4*)(p.x + unit * 8 + 4);
        const float y[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
        quad_quantize_store(y, unit, a1);
    }
    __syncthreads();
    if (wave == 0) {
        // ---------------- loader: W1 slots then W2 slots, `depth` in flight, never stops at the phase boundary
        int published = 0;
        for (int s = 0; s < nslots + p.depth; s++) {
            if (s < nslots) {
                const int slot = s % p.slots;
                if (s >= p.slots) {
                    int spin = 0;
                    while (flag_read(&freed[slot]) != s - p.slots + 1) { __builtin_amdgcn_s_sleep(1); if (++spin > SPIN_MAX) { *p.fail = 1; break; } }
                }
                if (p.thin && s >= S1) {                        // gather-pass knob: stay out of the gather's way
                    int spin = 0;
                    if (p.thin == 2 && flag_read(gathering) == 1) {
                        wait_vm<0>();                           // everything issued so far has landed: publish it before going quiet
                        for (; published < s; published++) if (lane == 0) flag_write(&ready[published % p.slots], published + 1);
                        while (flag_read(gathering) == 1) { __builtin_amdgcn_s_sleep(2); if (++spin > SPIN_MAX) { *p.fail = 5; break; } }
                    }
                    else if (flag_read(gathering) == 1) wait_vm<0>();   // one outstanding fill: the previous one has landed before the next is issued
                }
                char* dst = ring + (size_t)slot * SLOT;
                const bool ph1 = s < S1;
                const size_t r0 = ph1 ? ((size_t)blockIdx.x * S1 + s) * RPS : ((size_t)blockIdx.x * S2 + (s - S1)) * RPS;
                const char* gw = (const char*)(ph1 ? p.w1 : p.w2) + r0 * (nblk * 16) + lane * 16;
                const char* gs = (const char*)(ph1 ? p.ws1 : p.ws2) + r0 * (nblk * 4) + lane * 16;
#pragma unroll
                for (int i = 0; i < 16; i++) glds16<1>(gw + (size_t)i * 1024, dst + i * 1024);
#pragma unroll
                for (int i = 0; i < 4; i++) glds16<1>(gs + (size_t)i * 1024, dst + SLOT_W + i * 1024);
                if (p.dbg && lane == 0 && (s == S1 - 1 || s == nslots - 1)) p.dbg[blockIdx.x * 8 + (s == S1 - 1 ? 6 : 7)] = wall_clock64();
            }
            const int done = s - p.depth;
            if (done >= published && done < nslots) {
                published = done + 1;
                const int newer = (s < nslots ? s : nslots - 1) - done;
                wait_vm_rt(newer * LOADS);
                if (lane == 0) flag_write(&ready[done % p.slots], done + 1);
            }
        }
    } else {
        const int c = wave - 1;
        i32x4 rlo[NB], rhi[NB];
        float rd[NB];
        int rs8[NB];
        auto load_act = [&](const ActI8& a) {
#pragma unroll
            for (int i = 0; i < NB; i++) {
                const int b = lane + 64 * i;
                rlo[i] = a.lo[b]; rhi[i] = a.hi[b]; rd[i] = a.d[b]; rs8[i] = 8 * a.asum[b];
            }
        };
        load_act(a1);
        bool phase2 = false;
        const bool stamp = p.dbg && c == 0 && lane == 0;
        long long* dbg = p.dbg + blockIdx.x * 8;
        if (stamp) dbg[0] = wall_clock64();
        int polls = 0;
        for (int s = c; s < nslots; s += NCONS) {
            if (s >= S1 && !phase2) {
                // ---- the hand-off: the phase-1 row is gathered in 1024-granule chunks (8 x 16-byte sc1 loads per lane, one round
                //      trip when everything has landed), chunk j by consumer j % NCONS, each quantizing the units it gathered.
                if (stamp) dbg[1] = wall_clock64();
                if (p.mode == 0) {
                    if (lane == 0) flag_write(gathering, 1);
                    const int jstep = p.one_gatherer ? 1 : NCONS;
                    for (int j = p.one_gatherer ? (c == 0 ? 0 : p.n1 / 1024) : c; j < p.n1 / 1024; j += jstep) {
                        const u32x4* gp = (const u32x4*)(p.gran + j * 1024) + lane;
                        u32x4 g[8];
                        int spin = 0;
                        bool all;
                        do {
                            ld4_sc1_x4(gp, g[0], g[1], g[2], g[3]);
                            ld4_sc1_x4(gp + 256, g[4], g[5], g[6], g[7]);
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7])::"memory");
                            all = true;
#pragma unroll
                            for (int u = 0; u < 8; u++) all = all && ((int)g[u].y == p.epoch) && ((int)g[u].w == p.epoch);
                            all = __all(all);
                            if (stamp && polls++ == 0) dbg[2] = wall_clock64();
                            if (!all && ++spin > SPIN_MAX / 64) { *p.fail = 2; break; }
                        } while (!all);
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            y1[j * 1024 + u * 128 + lane * 2] = __uint_as_float(g[u].x);
                            y1[j * 1024 + u * 128 + lane * 2 + 1] = __uint_as_float(g[u].z);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        for (int unit = j * 128 + lane; unit < (j + 1) * 128; unit += 64) {   // LlamaModel.maybeQuantize
                            float y[8];
#pragma unroll
                            for (int i = 0; i < 8; i++) y[i] = y1[unit * 8 + i];
                            quad_quantize_store(y, unit, a2);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_fetch_add((int*)act2_ready, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    int spin = 0;
                    while (flag_read(act2_ready) != p.n1 / 1024) { __builtin_amdgcn_s_sleep(1); if (++spin > SPIN_MAX) { *p.fail = 3; break; } }
                    if (lane == 0) flag_write(gathering, 2);
                }
                if (stamp) { dbg[3] = wall_clock64(); dbg[5] = polls; }
                load_act(p.mode == 0 ? a2 : a1);
                phase2 = true;
            }
            const int slot = s % p.slots;
            {
                int spin = 0;
                while (flag_read(&ready[slot]) != s + 1) { __builtin_amdgcn_s_sleep(1); if (++spin > SPIN_MAX) { *p.fail = 4; break; } }
            }
            const char* base = ring + (size_t)slot * SLOT;
            const i32x4* wv = (const i32x4*)base;
            const float* sv = (const float*)(base + SLOT_W);
            float res = 0.0f;
#pragma unroll
            for (int r = 0; r < RPS; r++) {
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < NB; i++) {
                    const int b = r * nblk + lane + 64 * i;
                    const int isum = q4_block_dot(wv[b], rlo[i], rhi[i]) - rs8[i];
                    acc = fmaf(rd[i] * sv[b], (float)isum, acc);
                }
                acc = wave_sum(acc);
                if (lane == r) res = acc;
            }
            if (lane == 0) flag_write(&freed[slot], s + 1);
            if (s < S1) {
                const size_t row = ((size_t)blockIdx.x * S1 + s) * RPS + lane;
                if (lane < RPS)
                    __hip_atomic_store(p.gran + row, ((unsigned long long)(unsigned)p.epoch << 32) | __float_as_uint(res), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (lane < RPS) {
                p.y2[((size_t)blockIdx.x * S2 + (s - S1)) * RPS + lane] = res;
            }
        }
        if (stamp) dbg[4] = wall_clock64();
    }
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int K = 4096, N1 = 4096, N2 = 28672, nblk = K / 32, COPIES = 16;
    const size_t w1b = (size_t)N1 * nblk * 16, s1b = (size_t)N1 * nblk * 4, w2b = (size_t)N2 * nblk * 16, s2b = (size_t)N2 * nblk * 4;
    const size_t w1s = w1b + 4096, s1s = s1b + 4096, w2s = w2b + 4096, s2s = s2b + 4096;
    std::vector<uint8_t> hw1(w1s * COPIES), hw2(w2s * COPIES);
    std::vector<float> hs1(s1s * COPIES / 4), hs2(s2s * COPIES / 4), hx(K);
    srand(11);
    for (auto& b : hw1) b = (uint8_t)(rand() & 0xff);
    for (auto& b : hw2) b = (uint8_t)(rand() & 0xff);
    for (auto& f : hs1) f = (float)((rand() % 2000) - 1000) * 1e-4f;
    for (auto& f : hs2) f = (float)((rand() % 2000) - 1000) * 1e-4f;
    for (auto& f : hx) f = (float)((rand() % 2000) - 1000) * 1e-3f;
    uint8_t *dw1, *dw2; float *ds1, *ds2, *dx, *y1, *y2a, *y2b; unsigned long long* gran; int* fail;
    CK(hipMalloc(&dw1, hw1.size())); CK(hipMalloc(&dw2, hw2.size())); CK(hipMalloc(&ds1, hs1.size() * 4)); CK(hipMalloc(&ds2, hs2.size() * 4));
    CK(hipMalloc(&dx, K * 4)); CK(hipMalloc(&y1, N1 * 4)); CK(hipMalloc(&y2a, N2 * 4)); CK(hipMalloc(&y2b, N2 * 4));
    CK(hipMalloc(&gran, N1 * 8)); CK(hipMemset(gran, 0, N1 * 8)); CK(hipMalloc(&fail, 4)); CK(hipMemset(fail, 0, 4));
    CK(hipMemcpy(dw1, hw1.data(), hw1.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dw2, hw2.data(), hw2.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(ds1, hs1.data(), hs1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ds2, hs2.data(), hs2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, hx.data(), K * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // ---------------- (A) two launches of the production kernel with the shapes launch_gemv_i8q4 plans for them
    //                  (4096 rows: single shot R=1, 16 waves; 28672 rows: pipelined R=2, 8 waves, one workgroup per CU)
    auto launch_pair = [&](int c) {
        GemvParams g; memset(&g, 0, sizeof(g));
        g.K = K; g.ldb = K / 2; g.ldbf = nblk;
        g.w = dw1 + c * w1s; g.ws = (const float*)((const char*)ds1 + c * s1s); g.nrows = N1; g.x = dx; g.out = y1;
        hipLaunchKernelGGL((gemv_i8q4_kernel<PRO_QUANT_Q8, EPI_STORE, 1, 2, 0>), dim3(256), dim3(1024), lds_bytes_i8(K), st, g);
        g.w = dw2 + c * w2s; g.ws = (const float*)((const char*)ds2 + c * s2s); g.nrows = N2; g.x = y1; g.out = y2a;
        hipLaunchKernelGGL((gemv_i8q4_kernel<PRO_QUANT_Q8, EPI_STORE, 2, 2, 1>), dim3(256), dim3(512), lds_bytes_i8(K), st, g);
    };
    for (int it = -1; it < 4; it++) {
        if (it == 0) CK(hipEventRecord(e0, st));
        for (int c = 0; c < COPIES; c++) launch_pair(c);
    }
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float msA; CK(hipEventElapsedTime(&msA, e0, e1));
    launch_pair(0);
    CK(hipStreamSynchronize(st));
    std::vector<float> ra(N2), rb(N2);
    CK(hipMemcpy(ra.data(), y2a, N2 * 4, hipMemcpyDeviceToHost));
    printf("(A) two launches, register-staged, fused Q8 prologues: %.2f us per pair  (%.1f MB => %.2f TB/s)\n", msA * 1e3 / (4 * COPIES),
           (w1b + s1b + w2b + s2b) / 1e6, (w1b + s1b + w2b + s2b) / 1e6 / (msA * 1e3 / (4 * COPIES)));
    // ---------------- (B) one persistent launch
    int epoch = 1;
    for (int mode : {0, 1})
    for (int ncons : {3, 7})
        for (int slots : {6})
            for (int depth : {1})
            for (int thin : {0, 1, 2})
            for (int oneg : {0, 1}) {
                if (mode == 1 && (thin || oneg)) continue;
                EngParams e; memset(&e, 0, sizeof(e));
                e.n1 = N1; e.n2 = N2; e.x = dx; e.gran = gran; e.y2 = y2b; e.K = K; e.slots = slots; e.depth = depth; e.fail = fail; e.mode = mode;
                e.thin = thin; e.one_gatherer = oneg;
                const size_t lds = (size_t)slots * 20 * 1024 + 48 * 4 + 2 * lds_bytes_i8(K) + (size_t)N1 * 4;
                auto launch = [&](int c) {
                    e.w1 = dw1 + c * w1s; e.ws1 = (const float*)((const char*)ds1 + c * s1s); e.w2 = dw2 + c * w2s; e.ws2 = (const float*)((const char*)ds2 + c * s2s);
                    e.epoch = epoch++;
                    if (ncons == 3) hipLaunchKernelGGL((engine2_kernel<3>), dim3(256), dim3(256), lds, st, e);
                    else hipLaunchKernelGGL((engine2_kernel<7>), dim3(256), dim3(512), lds, st, e);
                };
                CK(hipFuncSetAttribute((const void*)engine2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                CK(hipFuncSetAttribute((const void*)engine2_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                CK(hipMemset(y2b, 0xff, N2 * 4));
                CK(hipMemset(fail, 0, 4));
                launch(0);
                CK(hipStreamSynchronize(st)); CK(hipGetLastError());
                CK(hipMemcpy(rb.data(), y2b, N2 * 4, hipMemcpyDeviceToHost));
                int bad = 0, hf = 0;
                for (int i = 0; i < N2; i++) bad += memcmp(&ra[i], &rb[i], 4) != 0;
                CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
                for (int it = -1; it < 4; it++) {
                    if (it == 0) CK(hipEventRecord(e0, st));
                    for (int c = 0; c < COPIES; c++) launch(c);
                }
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float msB; CK(hipEventElapsedTime(&msB, e0, e1));
                printf("(B) one persistent launch%s, %d consumers, ring %d slots, depth %d, loader %s, %s: %.2f us per pair  mismatches=%d fail=%d  (x%.2f of A)\n", mode ? " WITHOUT the hand-off (streaming floor; results differ by design)" : "", ncons, slots, depth,
                       thin == 0 ? "unthrottled" : thin == 1 ? "one fill outstanding during the gather" : "paused during the gather", oneg ? "ONE gatherer wave" : "a chunk per consumer",
                       msB * 1e3 / (4 * COPIES), bad, hf, msB / msA);
            }
    // ---------------- where the time goes: wall-clock stamps of consumer 0 / the loader of every workgroup, one launch each
    long long* dbg;
    CK(hipMalloc(&dbg, 256 * 8 * 8));
    for (int mode : {0, 2, 1}) {
        EngParams e; memset(&e, 0, sizeof(e));
        e.n1 = N1; e.n2 = N2; e.x = dx; e.gran = gran; e.y2 = y2b; e.K = K; e.slots = 6; e.depth = 1; e.fail = fail; e.mode = mode == 1; e.dbg = dbg;
        if (mode == 2) { e.thin = 2; e.one_gatherer = 1; printf("  (next three: loader paused during the gather, one gatherer wave)\n"); }
        const size_t lds = (size_t)6 * 20 * 1024 + 48 * 4 + 2 * lds_bytes_i8(K) + (size_t)N1 * 4;
        for (int rep = 0; rep < 3; rep++) {
            const int c = 3 + rep;
            e.w1 = dw1 + c * w1s; e.ws1 = (const float*)((const char*)ds1 + c * s1s); e.w2 = dw2 + c * w2s; e.ws2 = (const float*)((const char*)ds2 + c * s2s);
            e.epoch = epoch++;
            CK(hipMemset(dbg, 0, 256 * 8 * 8));
            hipLaunchKernelGGL((engine2_kernel<7>), dim3(256), dim3(512), lds, st, e);
            CK(hipStreamSynchronize(st));
            std::vector<long long> h(256 * 8);
            CK(hipMemcpy(h.data(), dbg, 256 * 8 * 8, hipMemcpyDeviceToHost));
            long long t0 = h[0];
            for (int b = 0; b < 256; b++) t0 = std::min(t0, h[b * 8]);
            auto stat = [&](int k, const char* name) {
                double mn = 1e9, mx = 0, sum = 0;
                for (int b = 0; b < 256; b++) { const double v = (h[b * 8 + k] - t0) * 0.01; mn = std::min(mn, v); mx = std::max(mx, v); sum += v; }
                printf("    %-44s min %6.2f  mean %6.2f  max %6.2f us\n", name, mn, sum / 256, mx);
            };
            double pl = 0; for (int b = 0; b < 256; b++) pl += h[b * 8 + 5];
            printf("  mode %d launch %d (7 consumers, ring 6, depth 1), time since the first workgroup's start:\n", mode == 1, rep);
            stat(0, "consumers start (after the Q8 prologue)");
            stat(6, "loader issued the last W1 slot");
            stat(1, "consumer 0 reaches the hand-off");
            if (mode != 1) stat(2, "first gather poll returned");
            stat(3, "phase-2 activation ready");
            stat(7, "loader issued the last W2 slot");
            stat(4, "consumer 0 done");
            if (mode != 1) printf("    mean polls of chunk(s) by consumer 0: %.2f\n", pl / 256);
        }
    }
    return 0;
}
