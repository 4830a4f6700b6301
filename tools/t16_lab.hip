// t16_lab: the reference-order I8 x Q4 GEMV with the pair sums on the integer MFMA ("one-hot selector" operand).
//
// Reference order (PTO:807-850): lane t of 16: acc_t = fma(da*sb, (float)(lo_t*a[t] + hi_t*a[t+16]), acc_t), blocks ascending.
// v_mfma_i32_16x16x32_i8 with K = one Q block: B column j = the block of weight row j (32 int8 = 16*(nib-8)),
// A row t = one-hot selector of the activation block (a[t] at the k slot of element t, a[t+16] at that of element t+16) =>
// D[t][j] = 16 * (lo_t*a[t] + hi_t*a[t+16]) of row j, exactly.  Lane (j = l&15, g = l>>4) holds D rows t = 4g..4g+3: four
// chains per lane, cvt + fma per (row, block, t).
// Questions: is it bit-exact, what does a launch cost on the decode shapes (solo: one wave owns a 16-row tile).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <cstdint>
#include "../jlama_amd/csrc/jh_kernels.h"
using namespace jh;
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------------------------------------- layout probe
__global__ void probe_kernel(const long* a, const long* b, int* d) {
    i32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_i32_16x16x32_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int i = 0; i < 4; i++) d[threadIdx.x * 4 + i] = c[i];
}
static int probe_layout() {
    // assumed: A lane l = row l&15, k = 8*(l>>4)+byte; B lane l = col l&15, k = 8*(l>>4)+byte; D lane l reg r = row 4*(l>>4)+r, col l&15
    int8_t A[16][32], B[32][16];
    srand(7);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 32; k++) A[i][k] = (int8_t)(rand() % 255 - 127);
    for (int k = 0; k < 32; k++) for (int j = 0; j < 16; j++) B[k][j] = (int8_t)(rand() % 255 - 127);
    long ha[64], hb[64];
    for (int l = 0; l < 64; l++) {
        uint64_t x = 0, y = 0;
        for (int q = 0; q < 8; q++) {
            x |= (uint64_t)(uint8_t)A[l & 15][8 * (l >> 4) + q] << (8 * q);
            y |= (uint64_t)(uint8_t)B[8 * (l >> 4) + q][l & 15] << (8 * q);
        }
        ha[l] = (long)x; hb[l] = (long)y;
    }
    long *da, *db; int* dd;
    CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dd, 1024));
    CK(hipMemcpy(da, ha, 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, 512, hipMemcpyHostToDevice));
    probe_kernel<<<1, 64>>>(da, db, dd);
    int hd[256];
    CK(hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            const int i = 4 * (l >> 4) + r, j = l & 15;
            int ref = 0;
            for (int k = 0; k < 32; k++) ref += (int)A[i][k] * (int)B[k][j];
            if (ref != hd[l * 4 + r]) bad++;
        }
    printf("layout probe v_mfma_i32_16x16x32_i8: %s (%d mismatches)\n", bad ? "ASSUMPTION WRONG" : "as assumed", bad);
    return bad;
}

// ---------------------------------------------------------------------------------------------- the kernel
// T16 weight layout: tile = 16 weight rows.  nibbles [tile][q = blk/4][lane = 16g + j][4 dwords]: dword d = dword g of block 4q+d
// of row j (a 4x4 dword transpose inside every 64-byte group of a row, done once at load time) -> one 16-byte load per lane and 4
// blocks, 1 KiB contiguous per wave instruction, and the registers ARE the MFMA B operands of blocks 4q..4q+3 (no shuffles).
// scales [tile][q][j][4]: the 4 block scales of row j (lanes j, j+16, j+32, j+48 read the same 16 bytes).
struct T16Params {
    const i32x4* w;
    const f32x4v* s;
    const int8_t* aq;    // pre-quantized activation row [K]
    const float* ad;     // block scales [K/32]
    float* out;          // [ntiles*16]
    int ntiles, K;
};
constexpr int SEL_STRIDE = 136;   // bytes per block in the selector table: 16 x 8 B (active lanes) + 8 zero bytes (everyone else)

// selector table in LDS: block b, t: {a[32b+t] << 8*(t&3), a[32b+16+t] << 8*(t&3)}; + d16[nblk]
__device__ __forceinline__ void fill_selector(char* smem, const T16Params& p, int nblk) {
    float* d16 = (float*)(smem + (size_t)nblk * SEL_STRIDE);
    // 16 consecutive codes per thread (one half of a block): one round trip, all loads independent
    for (int h = threadIdx.x; h < nblk * 2; h += blockDim.x) {
        const i32x4 v = *(const i32x4*)(p.aq + (size_t)h * 16);
        const int b = h >> 1, half = h & 1;
        int* dst = (int*)(smem + (size_t)b * SEL_STRIDE) + half;
#pragma unroll
        for (int t = 0; t < 16; t++) dst[2 * t] = v[t >> 2] & (0xFF << (8 * (t & 3)));
    }
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
        *(i32x2*)(smem + (size_t)b * SEL_STRIDE + 128) = i32x2{0, 0};
        d16[b] = p.ad[b] * 0.0625f;
    }
}

__device__ __forceinline__ float xlane_add32(float v) {   // v + v[lane ^ 32]
    return v + __shfl_xor(v, 32);
}
__device__ __forceinline__ float xlane_add16(float v) {
    return v + __shfl_xor(v, 16);
}

typedef float f32x2v __attribute__((ext_vector_type(2)));
constexpr int MAGIC_I = 0x4B400000;          // as float: 12582912.0 = 1.5 * 2^23; + n (|n| < 2^22) is exact in the low mantissa bits
constexpr float MAGIC_F = 12582912.0f;

// PIPE = 1: software pipeline over the q steps (selector reads one step ahead, chain one step behind the MFMAs)
__device__ __forceinline__ float cvt1(int a) { return (float)a; }   // in C: hipcc must see the MFMA -> VALU read hazard (asm hides it)
__device__ __forceinline__ float sub1(float a, float b) { float r; asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int NW, int D, int PIPE, int CVT>
__global__ __launch_bounds__(NW * 64) void gemv_t16_kernel(T16Params p, int tiles_per_wave, int aw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nblk = p.K / QB, nq = nblk >> 2;                 // host: nq % D == 0
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const float* d16 = (const float*)(smem + (size_t)nblk * SEL_STRIDE);
    const bool active = (j >> 2) == g;                         // the 16 lanes (t = j, g == t/4) read their entry, the others the zero word
    const char* sel = smem + (active ? j * 8 : 128);
    int tile0 = wave < aw ? (blockIdx.x * aw + wave) * tiles_per_wave : p.ntiles;   // waves [aw, NW) only help with the prologue
    if (tile0 > p.ntiles) tile0 = p.ntiles;
    int tile1 = tile0 + tiles_per_wave;
    if (tile1 > p.ntiles) tile1 = p.ntiles;
    const int items = (tile1 - tile0) * nq;                    // the wave's stream is contiguous: [tile][q] x 1 KiB (+ 256 B of scales)

    i32x4 wq[D];
    f32x4v sq[D];
    const i32x4* wp = p.w + (size_t)tile0 * nq * 64 + lane;
    const f32x4v* sp_ = p.s + (size_t)tile0 * nq * 16 + j;
    int li = 0;
    auto issue = [&](i32x4& w, f32x4v& s) __attribute__((always_inline)) {
        const int i = li < items ? li : items - 1;
        w = __builtin_nontemporal_load(wp + (size_t)i * 64);
        s = __builtin_nontemporal_load(sp_ + (size_t)i * 16);
        ++li;
    };
    fill_selector(smem, p, nblk);       // (lab stand-in for the quantizing prologue: its global loads would drain the ring)
    lds_barrier();
    if (items == 0) return;
#pragma unroll
    for (int d = 0; d < D; d++) {
        issue(wq[d], sq[d]);
        __builtin_amdgcn_sched_barrier(0);
    }

    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    float park = 0.f;
    int ct = tile0;
    struct Sel { long a[4]; f32x4v da; };
    auto read_sel = [&](int q, Sel& x) __attribute__((always_inline)) {
        const char* sp = sel + (size_t)(4 * q) * SEL_STRIDE;
#pragma unroll
        for (int d = 0; d < 4; d++) x.a[d] = *(const long*)(sp + d * SEL_STRIDE);
        x.da = *(const f32x4v*)(d16 + 4 * q);
    };
    auto mfmas = [&](const i32x4& w, const f32x4v& sc, const Sel& x, i32x4 (&dd)[4], f32x4v& s) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int lo = ((w[d] << 4) & (int)0xF0F0F0F0) ^ (int)0x80808080;
            const int hi = (w[d] & (int)0xF0F0F0F0) ^ (int)0x80808080;
            const long b = (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
            const int zi = CVT ? 0 : MAGIC_I;
            const i32x4 z = {zi, zi, zi, zi};
            dd[d] = __builtin_amdgcn_mfma_i32_16x16x32_i8(x.a[d], b, z, 0, 0, 0);
        }
        s[0] = mul1(x.da[0], sc[0]); s[1] = mul1(x.da[1], sc[1]); s[2] = mul1(x.da[2], sc[2]); s[3] = mul1(x.da[3], sc[3]);
    };
    // scalar f32 ops written as asm: hipcc's SLP vectoriser otherwise packs them into v_pk_fma_f32 / v_pk_add_f32, which cost
    // several issue slots each beside MFMAs on this chip (MI355X_MICROARCH.md "price of one filler")
    auto chain = [&](const i32x4 (&dd)[4], const f32x4v& s) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            float f0, f1, f2, f3;
            if constexpr (CVT) {
                f0 = cvt1(dd[d][0]); f1 = cvt1(dd[d][1]); f2 = cvt1(dd[d][2]); f3 = cvt1(dd[d][3]);
            } else {
                f0 = sub1(__int_as_float(dd[d][0]), MAGIC_F); f1 = sub1(__int_as_float(dd[d][1]), MAGIC_F);
                f2 = sub1(__int_as_float(dd[d][2]), MAGIC_F); f3 = sub1(__int_as_float(dd[d][3]), MAGIC_F);
            }
            acc0 = fma1(s[d], f0, acc0); acc1 = fma1(s[d], f1, acc1); acc2 = fma1(s[d], f2, acc2); acc3 = fma1(s[d], f3, acc3);
        }
    };
    auto tile_end = [&]() __attribute__((always_inline)) {
        // halving tree over t = 4g + i: (t, t+8) = lanes l, l+32; (t, t+4) = lanes l, l+16; then i, i+2; then 0, 1
        float a0 = xlane_add32(acc0), a1 = xlane_add32(acc1), a2 = xlane_add32(acc2), a3 = xlane_add32(acc3);
        a0 = xlane_add16(a0); a1 = xlane_add16(a1); a2 = xlane_add16(a2); a3 = xlane_add16(a3);
        const float r = (a0 + a2) + (a1 + a3);
        if (g == ((ct - tile0) & 3)) park = r;              // every lane group holds the 16 row sums; group n & 3 keeps tile n's
        if (((ct - tile0) & 3) == 3 || ct + 1 == tile1) {
            const int tt = ct - ((ct - tile0) & 3) + g;
            if (tt <= ct) p.out[(size_t)tt * 16 + j] = park;
        }
        acc0 = acc1 = acc2 = acc3 = 0.f;
        ++ct;
    };
    if constexpr (PIPE == 0) {
        int cq = 0;
        for (int it = 0; it < items; it += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                Sel x; i32x4 dd[4]; f32x4v s;
                read_sel(cq + d, x);
                mfmas(wq[d], sq[d], x, dd, s);
                __builtin_amdgcn_sched_barrier(0);
                issue(wq[d], sq[d]);
                __builtin_amdgcn_sched_barrier(0);
                chain(dd, s);
                __builtin_amdgcn_sched_barrier(0);
            }
            cq += D;
            if (cq == nq) { cq = 0; tile_end(); }
        }
    } else {
        // D is even: two register sets alternate (x[d & 1] = selector of this step, read one step earlier; dd[d & 1] = this step's MFMA
        // results, chained one step later), so nothing is copied; the ring slot of step `it + d` is d.  The chain behind the very
        // first step is a no-op (f = 0, s = 0: fma(0, 0, acc) = acc).
        static_assert(D % 2 == 0, "pipelined form alternates two register sets");
        Sel x[2];
        i32x4 dd[2][4];
        f32x4v sv[2];
#pragma unroll
        for (int k = 0; k < 4; k++) dd[1][k] = CVT ? i32x4{0, 0, 0, 0} : i32x4{MAGIC_I, MAGIC_I, MAGIC_I, MAGIC_I};
        sv[1] = f32x4v{0.f, 0.f, 0.f, 0.f};
        read_sel(0, x[0]);
        int cq = 0;             // q (within its tile) of the step whose MFMAs are issued next
        for (int it = 0; it < items; it += D) {
            if (it > 0 && cq == 0) {                        // the previous body closed a tile: its last chain first
                chain(dd[1], sv[1]);
                tile_end();
#pragma unroll
                for (int k = 0; k < 4; k++) dd[1][k] = CVT ? i32x4{0, 0, 0, 0} : i32x4{MAGIC_I, MAGIC_I, MAGIC_I, MAGIC_I};
                sv[1] = f32x4v{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int d = 0; d < D; d++) {
                constexpr int dummy = 0; (void)dummy;
                const int c = d & 1, o = c ^ 1;
                int qn = cq + d + 1;
                qn = qn == nq ? 0 : qn;
                read_sel(qn, x[o]);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(wq[d], sq[d], x[c], dd[c], sv[c]);
                __builtin_amdgcn_sched_barrier(0);
                chain(dd[o], sv[o]);                        // the previous step's sums while this step's MFMAs run
                __builtin_amdgcn_sched_barrier(0);
                issue(wq[d], sq[d]);
                __builtin_amdgcn_sched_barrier(0);
            }
            cq += D;
            if (cq == nq) cq = 0;
        }
        chain(dd[1], sv[1]);
        tile_end();
    }
}

// ---------------------------------------------------------------------------------------------- team kernel
// Few-row matrices (o-proj, down, q|k|v: 1-2 tiles per CU): one wave per tile leaves the CU's other SIMDs idle and is
// instruction-issue bound (~400 cycles per 4 blocks).  Only the fma chain is sequential; everything in front of it (unpack, MFMA,
// convert, scale product) is order-free.  H helper waves take the q steps of the workgroup's tiles round-robin and leave
// (float)isum and da*sb in LDS; 4 owner waves (owner r = D register r = chains t = 4g + r) run the chains.
// Hand-off format: 16*isum is a multiple of 16 below 2^15 -> EXACT in f16 (11 significant bits), so the sums travel as packed
// halves (half the LDS traffic, which bounds this kernel) and the owner's chain step is ONE v_fma_mix_f32 (f16 operand widened
// exactly, f32 fma, one rounding -- bit for bit fma(s, (float)isum16, acc)).  The scale products travel compactly: lane (j, g)
// multiplies only block 4q+g's.  One barrier per round of H * SPR steps, double-buffered.
//   scale layout for this kernel: [tile][q][g][j] (one dword per lane).
__device__ __forceinline__ float fma_mix_lo(float s, int h2, float acc) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(s), "v"(h2));
    return acc;
}
__device__ __forceinline__ float fma_mix_hi(float s, int h2, float acc) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(s), "v"(h2));
    return acc;
}
constexpr int TEAM_SLOT = 2048 + 256;   // bytes per q step in the hand-off buffer: F [2 reg pairs][64 lanes][16 B] + S [16 rows][4 blocks]
template <int H, int D, int SPR, bool DBG = false>
__global__ __launch_bounds__((H + 4) * 64) void gemv_t16_team_kernel(T16Params p, int tiles_per_wg, long long* dbg = nullptr, int knock = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RS = H * SPR;                                 // steps per round
    const int nblk = p.K / QB, nq = nblk >> 2;                 // host: nq % RS == 0
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const float* d16 = (const float*)(smem + (size_t)nblk * SEL_STRIDE);
    char* buf = smem + (((size_t)nblk * SEL_STRIDE + nblk * 4 + 15) & ~(size_t)15);   // [2][RS][TEAM_SLOT]
    float* red = (float*)(buf + 2 * RS * TEAM_SLOT);           // [4][16] owners' partial trees
    const bool active = (j >> 2) == g;
    const char* sel = smem + (active ? j * 8 : 128);
    int tile0 = blockIdx.x * tiles_per_wg;
    int tile1 = tile0 + tiles_per_wg;
    if (tile1 > p.ntiles) tile1 = p.ntiles;
    const int steps = (tile1 - tile0) * nq, rounds = steps / RS;
    fill_selector(smem, p, nblk);
    lds_barrier();
    if (wave < H) {
        // ---------------- helper: steps wave, wave + H, ...  (its SPR steps of a round are wave + H*k)
        i32x4 wq[D];
        float sq[D];
        const i32x4* wp = p.w + ((size_t)tile0 * nq + wave) * 64 + lane;
        const float* sp_ = (const float*)p.s + ((size_t)tile0 * nq + wave) * 64 + lane;
        const int nmine = rounds * SPR;
        int li = 0;
        auto issue = [&](i32x4& w, float& s) __attribute__((always_inline)) {
            const int i = li < nmine ? li : nmine - 1;
            w = __builtin_nontemporal_load(wp + (size_t)i * (64 * H));
            s = __builtin_nontemporal_load(sp_ + (size_t)i * (64 * H));
            ++li;
        };
#pragma unroll
        for (int d = 0; d < D; d++) {
            issue(wq[d], sq[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
        int q = wave;                                           // q of this helper's next step within its tile
        auto work = [&](const i32x4& w, float sc, char* slot) __attribute__((always_inline)) {
            const char* sp = sel + (size_t)(4 * q) * SEL_STRIDE;
            const float da = d16[4 * q + g];
            i32x4 dd[4];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const long a = *(const long*)(sp + d * SEL_STRIDE);
                const int lo = ((w[d] << 4) & (int)0xF0F0F0F0) ^ (int)0x80808080;
                const int hi = (w[d] & (int)0xF0F0F0F0) ^ (int)0x80808080;
                const long b = (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
                const i32x4 z = {0, 0, 0, 0};
                if (knock & 8) dd[d] = i32x4{(int)a, (int)b, (int)(a >> 32), (int)(b >> 32)};
                else dd[d] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, z, 0, 0, 0);
            }
            if (!(knock & 2)) ((float*)(slot + 2048))[j * 4 + g] = mul1(da, sc);
            typedef __fp16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int rp = 0; rp < 2; rp++) {
                i32x4 f;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const int r = 2 * rp + k;
                    const h2 p0 = __builtin_amdgcn_cvt_pkrtz((float)dd[0][r], (float)dd[1][r]);
                    const h2 p1 = __builtin_amdgcn_cvt_pkrtz((float)dd[2][r], (float)dd[3][r]);
                    f[2 * k] = __builtin_bit_cast(int, p0);
                    f[2 * k + 1] = __builtin_bit_cast(int, p1);
                }
                if (!(knock & 1)) ((i32x4*)slot)[rp * 64 + lane] = f;
            }
            q += H;
            if (q >= nq) q -= nq;
        };
        static_assert(D % SPR == 0, "a round's steps are whole ring slots");
        for (int i = 0; i < nmine; i += D) {                    // host: nmine % D == 0
#pragma unroll
            for (int d = 0; d < D; d++) {
                const i32x4 w = wq[d];
                const float sc = sq[d];
                __builtin_amdgcn_sched_barrier(0);
                issue(wq[d], sq[d]);
                __builtin_amdgcn_sched_barrier(0);
                const int st = i + d, round = st / SPR, k = st % SPR;
                long long t0 = 0, t1 = 0;
                if (DBG) t0 = clock64();
                work(w, sc, buf + (size_t)((round & 1) * RS + k * H + wave) * TEAM_SLOT);
                if (DBG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t1 = clock64(); }
                if (k == SPR - 1 && !(knock & 16)) lds_barrier();
                if (DBG && blockIdx.x == 7 && lane == 0 && k == SPR - 1) {
                    long long* o = dbg + ((size_t)wave * 64 + round) * 4;
                    o[0] = t0; o[1] = t1; o[2] = t1; o[3] = clock64();
                }
            }
        }
        lds_barrier();                                          // the owners run one round behind ...
        lds_barrier();                                          // ... and read one round ahead of their chain
    } else {
        // ---------------- owner r: chains t = 4g + r of the 16 rows.  The LDS reads of round i are issued before the chain of round
        // i-1 runs (two register sets), so a round costs max(read issue, chain) instead of their sum; owners outrank helpers at issue.
        const int r = wave - H;
        if (p.ntiles < 0) __builtin_amdgcn_s_setprio(3);   // (tried: owners at priority 3 -- slower)
        float acc = 0.f;
        int done = 0, ct = tile0;                               // steps chained so far in the current tile
        i32x2 f[2][RS] = {};
        f32x4v sv[2][RS] = {};
        auto read_round = [&](int i, i32x2 (&ff)[RS], f32x4v (&ss)[RS]) __attribute__((always_inline)) {
            if (knock & 4) return;
            const char* rb = buf + (size_t)((i & 1) * RS) * TEAM_SLOT;
#pragma unroll
            for (int h = 0; h < RS; h++) {
                ff[h] = *(const i32x2*)(rb + (size_t)h * TEAM_SLOT + ((r >> 1) * 64 + lane) * 16 + (r & 1) * 8);
                ss[h] = *(const f32x4v*)(rb + (size_t)h * TEAM_SLOT + 2048 + j * 16);
            }
        };
        auto chain_round = [&](const i32x2 (&ff)[RS], const f32x4v (&ss)[RS]) __attribute__((always_inline)) {
#pragma unroll
            for (int h = 0; h < RS; h++) {
                if (knock & 4) continue;
                acc = fma_mix_lo(ss[h][0], ff[h][0], acc); acc = fma_mix_hi(ss[h][1], ff[h][0], acc);
                acc = fma_mix_lo(ss[h][2], ff[h][1], acc); acc = fma_mix_hi(ss[h][3], ff[h][1], acc);
            }
            done += RS;
            if (done == nq) {
                // halving tree: (t, t+8) = lanes l, l^32; (t, t+4) = l, l^16; then owners (r, r+2), then (0, 1) -- through LDS
                float a = xlane_add32(acc);
                a = xlane_add16(a);
                if (lane < 16) red[r * 16 + lane] = a;
                acc = 0.f; done = 0;
            }
        };
        auto finish_tile = [&]() __attribute__((always_inline)) {   // after the barrier that follows a closed tile
            if (r == 0 && lane < 16) {
                const float b0 = red[lane], b1 = red[16 + lane], b2 = red[32 + lane], b3 = red[48 + lane];
                p.out[(size_t)ct * 16 + lane] = (b0 + b2) + (b1 + b3);
            }
            ++ct;
        };
        lds_barrier();                                          // round 0 written
        // rounds is even (host): iteration pairs keep the register sets static
        read_round(0, f[0], sv[0]);
        lds_barrier();                                          // round 1 written, round 0 in registers (lds_barrier waits for the reads)
        for (int i = 1; i < rounds; i += 2) {
            read_round(i, f[1], sv[1]);
            chain_round(f[0], sv[0]);
            const bool closed0 = done == 0;
            if (!(knock & 16)) lds_barrier();
            if (closed0) finish_tile();
            if (i + 1 < rounds) read_round(i + 1, f[0], sv[0]);
            chain_round(f[1], sv[1]);
            const bool closed1 = done == 0;
            if (!(knock & 16)) lds_barrier();
            if (closed1) finish_tile();
        }
    }
}

template <int H, int D, int SPR>
static void run_team(const char* tag, int nrows, int K, int layers, const std::vector<uint8_t>& nib, const std::vector<float>& sc,
                     const std::vector<int8_t>& aq, const std::vector<float>& ad, const std::vector<float>& ref, int grid_cus) {
    const int nblk = K / 32, nq = nblk / 4, ntiles = nrows / 16;
    constexpr int RS = H * SPR;
    if (nq % RS) { printf("%s: nq %% (H*SPR) != 0, skipped\n", tag); return; }
    int grid = grid_cus < ntiles ? grid_cus : ntiles;
    const int t_wg = (ntiles + grid - 1) / grid;
    grid = (ntiles + t_wg - 1) / t_wg;
    if ((t_wg * nq / H) % D) { printf("%s: steps per helper %% D != 0 (H %d D %d), skipped\n", tag, H, D); return; }
    if ((t_wg * nq / RS) % 2) { printf("%s: odd number of rounds (H %d SPR %d), skipped\n", tag, H, SPR); return; }
    std::vector<int> tw((size_t)ntiles * nq * 64 * 4);
    std::vector<float> ts((size_t)ntiles * nq * 64);
    for (int u = 0; u < ntiles; u++)
        for (int q = 0; q < nq; q++)
            for (int l = 0; l < 64; l++) {
                const int jj = l & 15, gg = l >> 4, row = u * 16 + jj;
                for (int d = 0; d < 4; d++) {
                    int v;
                    memcpy(&v, &nib[(size_t)row * (K / 2) + (size_t)(4 * q + d) * 16 + gg * 4], 4);
                    tw[(((size_t)u * nq + q) * 64 + l) * 4 + d] = v;
                }
                ts[((size_t)u * nq + q) * 64 + l] = sc[(size_t)row * nblk + 4 * q + gg];
            }
    i32x4* dw; f32x4v* ds; int8_t* daq; float *dad, *dout;
    const size_t wb = tw.size() * 4, sb = ts.size() * 4;
    CK(hipMalloc(&dw, wb * layers)); CK(hipMalloc(&ds, sb * layers));
    for (int l = 0; l < layers; l++) {
        CK(hipMemcpy((char*)dw + wb * l, tw.data(), wb, hipMemcpyHostToDevice));
        CK(hipMemcpy((char*)ds + sb * l, ts.data(), sb, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&daq, K)); CK(hipMalloc(&dad, nblk * 4)); CK(hipMalloc(&dout, nrows * 4));
    CK(hipMemcpy(daq, aq.data(), K, hipMemcpyHostToDevice)); CK(hipMemcpy(dad, ad.data(), nblk * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, nrows * 4));
    T16Params p{dw, ds, daq, dad, dout, ntiles, K};
    const size_t lds = (((size_t)nblk * SEL_STRIDE + nblk * 4 + 15) & ~(size_t)15) + (size_t)2 * RS * TEAM_SLOT + 64 * 4;
    CK(hipFuncSetAttribute((const void*)gemv_t16_team_kernel<H, D, SPR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 5;
    for (int it = -1; it < reps; it++) {
        if (it == 0) CK(hipEventRecord(e0));
        for (int l = 0; l < layers; l++) {
            T16Params q = p;
            q.w = (const i32x4*)((const char*)dw + wb * l); q.s = (const f32x4v*)((const char*)ds + sb * l);
            gemv_t16_team_kernel<H, D, SPR><<<grid, (H + 4) * 64, lds>>>(q, t_wg);
        }
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (getenv("T16_KNOCK")) {
        const int ks[] = {1, 2, 3, 4, 8, 11, 15, 16, 31};
        for (int kn : ks) {
            for (int it = -1; it < reps; it++) {
                if (it == 0) CK(hipEventRecord(e0));
                for (int l = 0; l < layers; l++) {
                    T16Params q = p;
                    q.w = (const i32x4*)((const char*)dw + wb * l); q.s = (const f32x4v*)((const char*)ds + sb * l);
                    gemv_t16_team_kernel<H, D, SPR><<<grid, (H + 4) * 64, lds>>>(q, t_wg, nullptr, kn);
                }
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float m2; CK(hipEventElapsedTime(&m2, e0, e1));
            printf("    knock %2d (1 no F writes, 2 no S write, 4 owner idle, 8 no MFMA, 16 no round barriers): %7.2f us\n", kn, m2 * 1e3 / (reps * layers));
        }
        for (int it = 0; it < 2; it++) for (int l = 0; l < layers; l++) {   // restore a clean output
            T16Params q = p;
            q.w = (const i32x4*)((const char*)dw + wb * l); q.s = (const f32x4v*)((const char*)ds + sb * l);
            gemv_t16_team_kernel<H, D, SPR><<<grid, (H + 4) * 64, lds>>>(q, t_wg);
        }
    }
    CK(hipGetLastError());
    std::vector<float> out(nrows);
    CK(hipMemcpy(out.data(), dout, nrows * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; double maxd = 0;
    for (int i = 0; i < nrows; i++)
        if (memcmp(&out[i], &ref[i], 4)) { bad++; const double dd = fabs((double)out[i] - ref[i]); if (dd > maxd) maxd = dd; }
    const double us = ms * 1e3 / (reps * layers), bytes = (double)nrows * K * 0.625;
    if (getenv("T16_DBG")) {
        long long* ddbg; CK(hipMalloc(&ddbg, 16 * 64 * 4 * 8)); CK(hipMemset(ddbg, 0, 16 * 64 * 4 * 8));
        CK(hipFuncSetAttribute((const void*)gemv_t16_team_kernel<H, D, SPR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        gemv_t16_team_kernel<H, D, SPR, true><<<grid, (H + 4) * 64, lds>>>(p, t_wg, ddbg);
        CK(hipDeviceSynchronize());
        std::vector<long long> hd(16 * 64 * 4);
        CK(hipMemcpy(hd.data(), ddbg, hd.size() * 8, hipMemcpyDeviceToHost));
        const int rounds = t_wg * nq / RS;
        const long long base = hd[0];
        printf("  stamps of workgroup 7 (cycles): helper h: last step of the round starts, LDS written, -, past barrier | owner: start, reads landed, chain done, past barrier\n");
        for (int i = 0; i < rounds && i < 16; i++) {
            printf("  round %2d h0 %6lld %6lld %6lld  h%d %6lld %6lld %6lld  o0 %6lld %6lld %6lld %6lld\n", i,
                   hd[(0 * 64 + i) * 4 + 0] - base, hd[(0 * 64 + i) * 4 + 1] - base, hd[(0 * 64 + i) * 4 + 3] - base, H - 1,
                   hd[((H - 1) * 64 + i) * 4 + 0] - base, hd[((H - 1) * 64 + i) * 4 + 1] - base, hd[((H - 1) * 64 + i) * 4 + 3] - base,
                   hd[(H * 64 + i) * 4 + 0] - base, hd[(H * 64 + i) * 4 + 1] - base, hd[(H * 64 + i) * 4 + 2] - base, hd[(H * 64 + i) * 4 + 3] - base);
        }
        CK(hipFree(ddbg));
    }
    printf("%-10s %6dx%-6d TEAM H %d D %d SPR %d grid %4d tiles/wg %d lds %zu: %7.2f us  %6.0f GB/s   mismatching rows %zu (max |d| %.3g)\n", tag, nrows, K, H, D, SPR,
           grid, t_wg, lds, us, bytes / us / 1e3, bad, maxd);
    CK(hipFree(dw)); CK(hipFree(ds)); CK(hipFree(daq)); CK(hipFree(dad)); CK(hipFree(dout));
}

// ---------------------------------------------------------------------------------------------- host
static float ref_dot(const int8_t* a, const float* ad, const uint8_t* nib, const float* sc, int nblk) {
    float acc[16];
    for (int t = 0; t < 16; t++) acc[t] = 0.f;
    for (int b = 0; b < nblk; b++) {
        const float scale = ad[b] * sc[b];
        for (int t = 0; t < 16; t++) {
            const int lo = (nib[b * 16 + t] & 15) - 8, hi = (nib[b * 16 + t] >> 4) - 8;
            const int isum = lo * a[b * 32 + t] + hi * a[b * 32 + 16 + t];
            acc[t] = fmaf(scale, (float)isum, acc[t]);
        }
    }
    for (int h = 8; h >= 1; h >>= 1)
        for (int t = 0; t < h; t++) acc[t] = acc[t] + acc[t + h];
    return acc[0];
}

template <int NW, int D, int PIPE, int CVT>
static void run(const char* tag, int nrows, int K, int layers, const std::vector<uint8_t>& nib, const std::vector<float>& sc,
                const std::vector<int8_t>& aq, const std::vector<float>& ad, const std::vector<float>& ref, int grid_cus) {
    const int nblk = K / 32, nq = nblk / 4, ntiles = nrows / 16;
    if (nq % D) { printf("%s: nq %% D != 0, skipped\n", tag); return; }
    // repack one layer, replicate
    std::vector<int> tw((size_t)ntiles * nq * 64 * 4);
    std::vector<float> ts((size_t)ntiles * nq * 16 * 4);
    for (int u = 0; u < ntiles; u++)
        for (int q = 0; q < nq; q++) {
            for (int l = 0; l < 64; l++) {
                const int jj = l & 15, gg = l >> 4, row = u * 16 + jj;
                for (int d = 0; d < 4; d++) {
                    int v;
                    memcpy(&v, &nib[(size_t)row * (K / 2) + (size_t)(4 * q + d) * 16 + gg * 4], 4);
                    tw[(((size_t)u * nq + q) * 64 + l) * 4 + d] = v;
                }
            }
            for (int jj = 0; jj < 16; jj++)
                for (int d = 0; d < 4; d++) ts[(((size_t)u * nq + q) * 16 + jj) * 4 + d] = sc[(size_t)(u * 16 + jj) * nblk + 4 * q + d];
        }
    i32x4* dw; f32x4v* ds; int8_t* daq; float *dad, *dout;
    const size_t wb = tw.size() * 4, sb = ts.size() * 4;
    CK(hipMalloc(&dw, wb * layers)); CK(hipMalloc(&ds, sb * layers));
    for (int l = 0; l < layers; l++) {
        CK(hipMemcpy((char*)dw + wb * l, tw.data(), wb, hipMemcpyHostToDevice));
        CK(hipMemcpy((char*)ds + sb * l, ts.data(), sb, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&daq, K)); CK(hipMalloc(&dad, nblk * 4)); CK(hipMalloc(&dout, nrows * 4));
    CK(hipMemcpy(daq, aq.data(), K, hipMemcpyHostToDevice)); CK(hipMemcpy(dad, ad.data(), nblk * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, nrows * 4));
    T16Params p{dw, ds, daq, dad, dout, ntiles, K};
    // every CU the same number of tiles where possible: grid = CUs, the tiles of a workgroup spread over its NW waves
    int grid = grid_cus < ntiles ? grid_cus : ntiles;
    const int t_wg = (ntiles + grid - 1) / grid;
    const int tpw = (t_wg + NW - 1) / NW;
    const int aw = (t_wg + tpw - 1) / tpw;
    grid = (ntiles + aw * tpw - 1) / (aw * tpw);
    const size_t lds = (size_t)nblk * SEL_STRIDE + nblk * 4 + 16;
    CK(hipFuncSetAttribute((const void*)gemv_t16_kernel<NW, D, PIPE, CVT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 5;
    for (int it = -1; it < reps; it++) {
        if (it == 0) CK(hipEventRecord(e0));
        for (int l = 0; l < layers; l++) {
            T16Params q = p;
            q.w = (const i32x4*)((const char*)dw + wb * l); q.s = (const f32x4v*)((const char*)ds + sb * l);
            gemv_t16_kernel<NW, D, PIPE, CVT><<<grid, NW * 64, lds>>>(q, tpw, aw);
        }
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    std::vector<float> out(nrows);
    CK(hipMemcpy(out.data(), dout, nrows * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; double maxd = 0;
    for (int i = 0; i < nrows; i++) {
        if (memcmp(&out[i], &ref[i], 4)) { bad++; const double dd = fabs((double)out[i] - ref[i]); if (dd > maxd) maxd = dd; }
    }
    const double us = ms * 1e3 / (reps * layers), bytes = (double)nrows * K * 0.625;
    printf("%-10s %6dx%-6d pipe %d cvt %d NW %2d (active %d) D %2d grid %4d tiles/wave %d: %7.2f us  %6.0f GB/s   mismatching rows %zu (max |d| %.3g)\n", tag, nrows, K, PIPE, CVT, NW, aw, D, grid, tpw,
           us, bytes / us / 1e3, bad, maxd);
    CK(hipFree(dw)); CK(hipFree(ds)); CK(hipFree(daq)); CK(hipFree(dad)); CK(hipFree(dout));
}

int main(int argc, char** argv) {
    if (probe_layout()) return 1;
    struct Shape { const char* name; int nrows, K, layers; };
    const Shape shapes[] = {{"gate|up", 28672, 4096, 12}, {"q|k|v", 6144, 4096, 32}, {"o", 4096, 4096, 32}, {"down", 4096, 14336, 12}};
    for (const Shape& sh : shapes) {
        if (argc > 1 && strcmp(argv[1], sh.name)) continue;
        const int nrows = sh.nrows, K = sh.K, nblk = K / 32;
        std::vector<uint8_t> nib((size_t)nrows * K / 2);
        std::vector<float> sc((size_t)nrows * nblk), ad(nblk), ref(nrows);
        std::vector<int8_t> aq(K);
        srand(1234);
        for (auto& v : nib) v = (uint8_t)(rand() & 255);
        for (auto& v : sc) v = (float)((rand() % 2000) + 1) * 1.37e-4f;
        for (auto& v : ad) v = (float)((rand() % 2000) + 1) * 3.1e-3f;
        for (auto& v : aq) v = (int8_t)(rand() % 255 - 127);
#pragma omp parallel for
        for (int i = 0; i < nrows; i++) ref[i] = ref_dot(aq.data(), ad.data(), &nib[(size_t)i * K / 2], &sc[(size_t)i * nblk], nblk);
        if (nrows >= 16384) {
            run<7, 4, 0, 1>(sh.name, nrows, K, sh.layers, nib, sc, aq, ad, ref, 256);
            run<7, 4, 1, 1>(sh.name, nrows, K, sh.layers, nib, sc, aq, ad, ref, 256);
            run<7, 4, 1, 0>(sh.name, nrows, K, sh.layers, nib, sc, aq, ad, ref, 256);
            run<7, 8, 1, 1>(sh.name, nrows, K, sh.layers, nib, sc, aq, ad, ref, 256);
            run<7, 2, 1, 1>(sh.name, nrows, K, sh.layers, nib, sc, aq, ad, ref, 256);
        } else {
            run<8, 8, 1, 1>(sh.name, nrows, K, sh.layers, nib, sc, aq, ad, ref, 256);
            run_team<8, 4, 1>(sh.name, nrows, K, sh.layers, nib, sc, aq, ad, ref, 256);
            run_team<8, 7, 1>(sh.name, nrows, K, sh.layers, nib, sc, aq, ad, ref, 256);
        }
    }
    return 0;
}
