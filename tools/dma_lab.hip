// dma_lab.hip -- loader/consumer I8 x Q4 GEMV with the weight stream going HBM -> LDS by LDS-DMA (global_load_lds), the
// structure MI355X_MICROARCH.md prices as "ldsdma-fill" (one loader wave per CU, consumer waves reading the ring).
// Measures it against the register-staged gemv_i8q4_kernel (jh_kernels.h) on the decode shapes, bit-exactly (same
// lane->block map, same fma order, same wave reduction), with enough weight copies to stream from HBM.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/dma_lab.hip -o tools/dma_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../jlama_amd/csrc/jh_kernels.h"
using namespace jh;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct DmaParams {
    const uint8_t* w; const float* ws; float* out;
    const int8_t* aq; const float* ad;
    int nrows, K;
    int rows_per_slot, wloads, sloads;     // slot = rows_per_slot rows: wloads x 1 KiB of nibbles + sloads x 1 KiB of scales
    int slots;                             // ring depth
    int depth;                             // slots the loader keeps in flight before it waits for the oldest
    int nt;
    int nload;                             // loader waves per workgroup (loader w fills slots s = w mod nload)
    int interleave;                        // 1: workgroup b takes global slots b, b+grid, ... (the chip sweeps memory together)
};

template <int NT>
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    // LDS destination = wave-uniform base + lane*16 (cdna_hip_programming.md: "wave-uniform base + lane x size")
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, NT ? 2 : 0);
}
// LDS flag words through inline asm: hipcc treats an LDS-DMA in flight as a pending LDS write and puts `s_waitcnt vmcnt(0)`
// in front of every ds_read it can see (the loader would drain its whole queue before each poll); an asm ds_read is
// invisible to that pass.  The flags never alias DMA destinations.
__device__ __forceinline__ unsigned lds_off(const volatile void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const volatile void*)p; }
__device__ __forceinline__ int flag_read(const volatile int* p) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_off(p)) : "memory");
    return v;
}
__device__ __forceinline__ void flag_write(volatile int* p, int v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(lds_off(p)), "v"(v) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_rt(int n) {   // counted wait with a runtime count (<= 63): a small switch ladder
    switch (n) {
#define W(k) case k: wait_vm<k>(); break;
        W(0) W(1) W(2) W(3) W(4) W(5) W(6) W(7) W(8) W(9) W(10) W(11) W(12) W(13) W(14) W(15) W(16) W(17) W(18) W(19) W(20) W(21) W(22) W(23)
        W(24) W(25) W(26) W(27) W(28) W(29) W(30) W(31) W(32) W(33) W(34) W(35) W(36) W(37) W(38) W(39) W(40) W(41) W(42) W(43) W(44) W(45)
        W(46) W(47) W(48) W(49) W(50) W(51) W(52) W(53) W(54) W(55) W(56) W(57) W(58) W(59) W(60) W(61) W(62) W(63)
#undef W
        default: wait_vm<0>();
    }
}

// NB = Q blocks per lane per row (K = NB*64*32); NCONS consumer waves + 1 loader wave per workgroup
template <int NB, int NCONS, int NT>
__global__ __launch_bounds__(512) void gemv_dma_kernel(DmaParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int nblk = NB * 64;
    const int slot_w = p.wloads * 1024, slot_s = p.sloads * 1024, slot_bytes = slot_w + slot_s;
    char* ring = smem;                                            // [slots][slot_bytes]
    volatile int* ready = (volatile int*)(smem + (size_t)p.slots * slot_bytes);   // [slots] sequence number + 1 of the slot's content
    volatile int* freed = ready + 16;                             // [slots] sequence number + 1 the consumers are done with
    i32x4* alo = (i32x4*)(freed + 16);                            // activation row: [nblk] lo, [nblk] hi, [nblk] d, [nblk] asum
    i32x4* ahi = alo + nblk;
    float* adv = (float*)(ahi + nblk);
    int* asum = (int*)(adv + nblk);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // rows of this workgroup: contiguous, a whole number of slots
    const int slots_total = p.nrows / p.rows_per_slot;
    const int per = (slots_total + gridDim.x - 1) / gridDim.x;
    int s0 = blockIdx.x * per, sstep = 1;
    int s1 = s0 + per;
    if (s1 > slots_total) s1 = slots_total;
    int nslots = s1 > s0 ? s1 - s0 : 0;
    if (p.interleave) {   // global slot of local slot s = blockIdx + s*grid
        s0 = blockIdx.x; sstep = gridDim.x;
        nslots = ((int)blockIdx.x < slots_total) ? (slots_total - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    }
    if (tid < 16) { ready[tid] = 0; freed[tid] = 0; }
    // activation (pre-quantized) -> LDS, by everyone
    for (int b = tid; b < nblk; b += blockDim.x) {
        const i32x4* src = (const i32x4*)(p.aq + (size_t)b * QB);
        const i32x4 l = src[0], h = src[1];
        alo[b] = l; ahi[b] = h; adv[b] = p.ad[b];
        int s = 0;
        s = sdot4(l.x, 0x01010101, s); s = sdot4(l.y, 0x01010101, s); s = sdot4(l.z, 0x01010101, s); s = sdot4(l.w, 0x01010101, s);
        s = sdot4(h.x, 0x01010101, s); s = sdot4(h.y, 0x01010101, s); s = sdot4(h.z, 0x01010101, s); s = sdot4(h.w, 0x01010101, s);
        asum[b] = s;
    }
    __syncthreads();
    if (wave < p.nload) {
        // ---------------- loaders: loader w owns local slots w, w+nload, ...; each keeps `depth` of ITS slots in flight
        const size_t row_bytes = (size_t)nblk * 16, srow_bytes = (size_t)nblk * 4;
        const int loads_per_slot = p.wloads + p.sloads;
        const int mine = (nslots - wave + p.nload - 1) / p.nload;      // number of slots of this loader
        for (int k = 0; k < mine + p.depth; k++) {
            const int s = wave + k * p.nload;
            if (k < mine) {
                const int slot = s % p.slots;
                if (s >= p.slots) {   // wait until the consumers released this slot's previous content
                    while (flag_read(&freed[slot]) != s - p.slots + 1) __builtin_amdgcn_s_sleep(1);
                }
                char* dst = ring + (size_t)slot * slot_bytes;
                const size_t r0 = (size_t)(s0 + (size_t)s * sstep) * p.rows_per_slot;
                const char* gw = (const char*)p.w + r0 * row_bytes + lane * 16;
                const char* gs = (const char*)p.ws + r0 * srow_bytes + lane * 16;
                for (int i = 0; i < p.wloads; i++) glds16<NT>(gw + (size_t)i * 1024, dst + i * 1024);
                for (int i = 0; i < p.sloads; i++) glds16<NT>(gs + (size_t)i * 1024, dst + slot_w + i * 1024);
            }
            const int kd = k - p.depth;     // my slot issued `depth` iterations ago has landed once at most depth*loads are pending
            if (kd >= 0 && kd < mine) {
                const int newer = (k < mine ? k : mine - 1) - kd;   // my slots issued after it
                wait_vm_rt(newer * loads_per_slot);
                const int done = wave + kd * p.nload;
                if (lane == 0) flag_write(&ready[done % p.slots], done + 1);
            }
        }
    } else {
        // ---------------- consumers: slot s -> consumer (s % NCONS)
        const int c = wave - p.nload;
        i32x4 rlo[NB], rhi[NB];
        float rd[NB];
        int rs8[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int b = lane + 64 * i;
            rlo[i] = alo[b]; rhi[i] = ahi[b]; rd[i] = adv[b]; rs8[i] = 8 * asum[b];
        }
        for (int s = c; s < nslots; s += NCONS) {
            const int slot = s % p.slots;
            while (flag_read(&ready[slot]) != s + 1) __builtin_amdgcn_s_sleep(1);
            const char* base = ring + (size_t)slot * slot_bytes;
            const i32x4* wv = (const i32x4*)base;
            const float* sv = (const float*)(base + slot_w);
            float res = 0.0f;
            for (int r = 0; r < p.rows_per_slot; r++) {
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
            if (lane == 0) flag_write(&freed[slot], s + 1);      // every lane's ds_reads of this slot were consumed above
            if (lane < p.rows_per_slot) p.out[(size_t)(s0 + (size_t)s * sstep) * p.rows_per_slot + lane] = res;
        }
    }
}

template <int NB, int NCONS, int NT>
float run_dma(DmaParams p, int grid, int copies, size_t wstride, size_t sstride, int iters, hipStream_t st, bool check_only = false) {
    const int nblk = NB * 64;
    const size_t lds = (size_t)p.slots * (p.wloads + p.sloads) * 1024 + 128 + (size_t)nblk * 40;
    CK(hipFuncSetAttribute((const void*)gemv_dma_kernel<NB, NCONS, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint8_t* w0 = p.w; const float* s0 = p.ws;
    for (int it = -1; it < iters; it++) {
        if (it == 0) CK(hipEventRecord(e0, st));
        for (int c = 0; c < copies; c++) {
            p.w = w0 + c * wstride; p.ws = (const float*)((const char*)s0 + c * sstride);
            hipLaunchKernelGGL((gemv_dma_kernel<NB, NCONS, NT>), dim3(grid), dim3((NCONS + p.nload) * 64), lds, st, p);
            if (check_only) break;
        }
        if (check_only) break;
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    float ms = 0;
    if (!check_only) CK(hipEventElapsedTime(&ms, e0, e1));
    return check_only ? 0.f : ms * 1e3f / (iters * copies);
}

template <int R, int NB, int PIPE>
float run_ref(GemvParams p, int grid, int threads, int copies, size_t wstride, size_t sstride, int iters, hipStream_t st, bool once = false) {
    const size_t lds = lds_bytes_i8(p.K);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint8_t* w0 = p.w; const float* s0 = p.ws;
    for (int it = -1; it < iters; it++) {
        if (it == 0) CK(hipEventRecord(e0, st));
        for (int c = 0; c < copies; c++) {
            p.w = w0 + c * wstride; p.ws = (const float*)((const char*)s0 + c * sstride);
            hipLaunchKernelGGL((gemv_i8q4_kernel<PRO_Q8, EPI_STORE, R, NB, PIPE>), dim3(grid), dim3(threads), lds, st, p);
            if (once) break;
        }
        if (once) break;
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    float ms = 0;
    if (!once) CK(hipEventElapsedTime(&ms, e0, e1));
    return once ? 0.f : ms * 1e3f / (iters * copies);
}

template <int NB>
void shape(const char* name, int nrows, int K, int copies, hipStream_t st) {
    const int nblk = K / 32;
    const size_t wbytes = (size_t)nrows * nblk * 16, sbytes = (size_t)nrows * nblk * 4;
    const size_t wstride = wbytes + 4096, sstride = sbytes + 4096;     // + pad: the last slot's scale loads may over-read
    std::vector<uint8_t> hw(wstride * copies);
    std::vector<float> hs(sstride * copies / 4);
    srand(7);
    for (auto& b : hw) b = (uint8_t)(rand() & 0xff);
    for (auto& f : hs) f = (float)((rand() % 2000) - 1000) * 1e-4f;
    std::vector<int8_t> haq(K);
    std::vector<float> had(nblk);
    for (auto& a : haq) a = (int8_t)((rand() % 255) - 127);
    for (auto& d : had) d = (float)(rand() % 1000) * 1e-3f;
    uint8_t* dw; float* ds; int8_t* daq; float* dad; float *o_ref, *o_dma;
    CK(hipMalloc(&dw, hw.size())); CK(hipMalloc(&ds, hs.size() * 4)); CK(hipMalloc(&daq, K)); CK(hipMalloc(&dad, nblk * 4));
    CK(hipMalloc(&o_ref, nrows * 4)); CK(hipMalloc(&o_dma, nrows * 4));
    CK(hipMemcpy(dw, hw.data(), hw.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(daq, haq.data(), K, hipMemcpyHostToDevice)); CK(hipMemcpy(dad, had.data(), nblk * 4, hipMemcpyHostToDevice));
    GemvParams g; memset(&g, 0, sizeof(g));
    g.w = dw; g.ws = ds; g.out = o_ref; g.nrows = nrows; g.K = K; g.ldb = K / 2; g.ldbf = nblk; g.aq = daq; g.ad = dad;
    const double mb = (double)(wbytes + sbytes) / 1e6;
    // register-staged baseline: the production plans (PIPE=1 R=2 for gate/up-sized, single shot otherwise)
    float t_ref;
    if (NB == 2 && nrows >= 16384) t_ref = run_ref<2, NB, 1>(g, 256, 512, copies, wstride, sstride, 4, st);
    else if (NB == 2) t_ref = run_ref<2, NB, 0>(g, 256, ((nrows / 2 + 255) / 256) * 64, copies, wstride, sstride, 4, st);
    else t_ref = run_ref<1, NB, 0>(g, nrows / 8, 512, copies, wstride, sstride, 4, st);   // production plan of the down projection: 8 single-row waves per workgroup
    g.w = dw; g.ws = ds;
    if (NB == 2 && nrows >= 16384) run_ref<2, NB, 1>(g, 256, 512, 1, wstride, sstride, 1, st, true);
    else if (NB == 2) run_ref<2, NB, 0>(g, 256, ((nrows / 2 + 255) / 256) * 64, 1, wstride, sstride, 1, st, true);
    else run_ref<1, NB, 0>(g, nrows / 8, 512, 1, wstride, sstride, 1, st, true);
    std::vector<float> href(nrows), hdma(nrows);
    CK(hipMemcpy(href.data(), o_ref, nrows * 4, hipMemcpyDeviceToHost));
    printf("%-8s N=%d K=%d  %.1f MB   register-staged: %.2f us  %.2f TB/s\n", name, nrows, K, mb, t_ref, mb / t_ref);
    DmaParams d; memset(&d, 0, sizeof(d));
    d.w = dw; d.ws = ds; d.out = o_dma; d.aq = daq; d.ad = dad; d.nrows = nrows; d.K = K;
    d.rows_per_slot = NB == 2 ? 8 : 2;
    d.wloads = d.rows_per_slot * nblk * 16 / 1024;
    d.sloads = (d.rows_per_slot * nblk * 4 + 1023) / 1024;
    struct V { int grid, slots, depth, nload, inter; };
    const V vs[] = {{256, 6, 2, 1, 0}, {256, 6, 2, 1, 1}, {256, 6, 1, 2, 0}, {256, 6, 1, 2, 1}, {256, 6, 2, 2, 1}, {512, 3, 1, 1, 0}, {512, 3, 1, 1, 1},
                    {256, 7, 3, 1, 1}, {256, 4, 1, 2, 1}};
    for (const V& v : vs) {
        d.slots = v.slots; d.depth = v.depth; d.nt = 1; d.nload = v.nload; d.interleave = v.inter;
        if ((d.wloads + d.sloads) * v.depth > 63) continue;
        if ((size_t)v.slots * (d.wloads + d.sloads) * 1024 * (v.grid / 256) + (size_t)nblk * 40 * (v.grid / 256) > 158 * 1024) continue;
        CK(hipMemset(o_dma, 0xff, nrows * 4));
        d.w = dw; d.ws = ds;
        run_dma<NB, 3, 1>(d, v.grid, 1, wstride, sstride, 1, st, true);
        CK(hipMemcpy(hdma.data(), o_dma, nrows * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < nrows; i++) bad += memcmp(&href[i], &hdma[i], 4) != 0;
        const float t3 = run_dma<NB, 3, 1>(d, v.grid, copies, wstride, sstride, 4, st);
        const float t6 = run_dma<NB, 6, 1>(d, v.grid, copies, wstride, sstride, 4, st);
        printf("   dma grid=%d slots=%d depth=%d loaders=%d interleave=%d  mismatches=%d   3 consumers: %.2f us %.2f TB/s   6 consumers: %.2f us %.2f TB/s\n",
               v.grid, v.slots, v.depth, v.nload, v.inter, bad, t3, mb / t3, t6, mb / t6);
    }
    hipFree(dw); hipFree(ds); hipFree(daq); hipFree(dad); hipFree(o_ref); hipFree(o_dma);
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    shape<2>("gate/up", 28672, 4096, 24, st);
    shape<2>("gate/up x4 rows (steady state)", 114688, 4096, 8, st);
    shape<7>("down", 4096, 14336, 24, st);
    return 0;
}
