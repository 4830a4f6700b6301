// jh_t16.h -- reference-order I8 x Q4 GEMV with the block pair sums on the integer MFMA pipe ("t16": 16-row tiles).
//
// What the reference fixes (GemmerI8Q4_512, PTO:807-850; oracle jo_dot_i8q4_scalar): per output row 16 float lanes, lane t walks
// the Q blocks in ascending K with  acc_t = fma(da*sb, (float)(lo_t*a[t] + hi_t*a[t+16]), acc_t), then the halving tree.  Only
// that fma chain is ordered; the integer pair sum in front of it is exact whichever way it is computed.  jh_p16.h computes it on
// the VALU (byte transpose + v_perm + v_dot4 + cvt per lane and block: 2.0 wave-instructions per (row, block), VALU-bound while
// the MFMA pipe idles).  Here ONE v_mfma_i32_16x16x32_i8 delivers the 16 pair sums of 16 weight rows for a whole block:
//   B (k x 16 columns)  = the block of 16 weight rows as int8 16*(nib-8)              -- column j = weight row j
//   A (16 rows x k)     = a one-hot "selector" of the activation block: row t holds a[t] at the k slot of element t and
//                         a[t+16] at the k slot of element t+16, zero elsewhere
//   D[t][j]             = 16 * (lo_t*a[t] + hi_t*a[t+16]) of weight row j, an exact int32.
// Lane (j = l & 15, g = l >> 4) receives D rows t = 4g..4g+3 of column j: four chains per lane, and what is left on the VALU is
// cvt + fma per (row, block, t) plus 3 bit ops per weight dword: 0.9 wave-instructions per (row, block).  The k index of an MFMA
// is ours to assign as long as A and B agree: lane group g supplies dword g of the block (elements 4g..4g+3 in the low nibbles,
// 16+4g..16+4g+3 in the high ones), so the selector lane (t, g) is non-zero only for g == t/4.
//
// Weight layout "T16" (a second resident copy, made once per weight by t16_pack_kernel -- this part has the HBM for it):
//   nibbles [tile][q = blk/4][lane = 16g + j][4 dwords]: dword d = dword g of block 4q+d of the tile's row j  (a 4x4 dword
//           transpose inside every 64-byte group of a row): one 16-byte load per lane and 4 blocks, 1 KiB contiguous per wave
//           instruction, and the loaded registers ARE the B operands of blocks 4q..4q+3 -- no shuffles;
//   scales  [tile][q][j][4 floats]: the four block scales of row j (lanes j, j+16, j+32, j+48 read the same 16 bytes).
// A tile's 16 rows are chosen by the packer: for gate|up, tile u = gate rows 8u..8u+7 then up rows 8u..8u+7, so SiLU(gate)*up
// meets inside a 16-lane row and 14336 hidden units split evenly over 256 CUs (7 tiles each).
// One wave owns a tile ("solo"): right for matrices with >= 4 tiles per CU (gate|up), where the launch is then HBM-bound
// (tools/t16_lab.hip: 13.8 us for the 8B gate|up against 20.0 us of the p16 kernel, same bits).
// Compiled with -ffp-contract=off like the rest: every FMA is explicit.
#pragma once
#include "jh_p16.h"

namespace jh {

typedef float f32x4t __attribute__((ext_vector_type(4)));
constexpr int T16_SEL_STRIDE = 136;   // bytes per block in the selector table: 16 x 8 B (the active lanes) + 8 zero bytes (everyone else)

// ------------------------------------------------------------------------------------------------ packer
// mode 0: tile u = rows 16u..16u+15 of w;  mode 1 (gate|up): tile u = rows 8u..8u+7 of w (gate) then 8u..8u+7 of w2 (up).
// One thread per (tile, q, j): 64 contiguous bytes (4 blocks) of its row in, 4 x 16 bytes out (one per lane group).
__global__ __launch_bounds__(256) void t16_pack_kernel(const i32x4* __restrict__ w, const float* __restrict__ ws, const i32x4* __restrict__ w2,
                                                       const float* __restrict__ ws2, int nblk, int ntiles, int mode, i32x4* __restrict__ tw,
                                                       f32x4t* __restrict__ ts) {
    const int nq = nblk >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)ntiles * nq * 16) return;
    const int j = (int)(idx & 15);
    const long long uq = idx >> 4;
    const int q = (int)(uq % nq), u = (int)(uq / nq);
    const i32x4* src;
    const float* ssrc;
    if (mode == 1) {
        const int row = 8 * u + (j & 7);
        src = ((j < 8) ? w : w2) + (size_t)row * nblk;
        ssrc = ((j < 8) ? ws : ws2) + (size_t)row * nblk;
    } else {
        const int row = 16 * u + j;
        src = w + (size_t)row * nblk;
        ssrc = ws + (size_t)row * nblk;
    }
    const i32x4 b0 = src[4 * q], b1 = src[4 * q + 1], b2 = src[4 * q + 2], b3 = src[4 * q + 3];
    i32x4* dst = tw + (size_t)uq * 64 + j;
    dst[0] = i32x4{b0.x, b1.x, b2.x, b3.x};
    dst[16] = i32x4{b0.y, b1.y, b2.y, b3.y};
    dst[32] = i32x4{b0.z, b1.z, b2.z, b3.z};
    dst[48] = i32x4{b0.w, b1.w, b2.w, b3.w};
    ts[(size_t)uq * 16 + j] = *(const f32x4t*)(ssrc + 4 * q);
}
static inline size_t t16_w_bytes(int rows, int K) { return (size_t)rows * (K / QB) * 16; }
static inline size_t t16_s_bytes(int rows, int K) { return (size_t)rows * (K / QB) * 4; }

// ------------------------------------------------------------------------------------------------ activation row in LDS
struct ActT16 {
    char* sel;     // [nblk][136 B]: entry t (8 B) = {a[32b+t] << 8*(t&3), a[32b+16+t] << 8*(t&3)} as bytes of the two dwords; +128: zeros
    float* d16;    // [nblk] activation block scale / 16
    double* red;   // [32]
};
__device__ __forceinline__ ActT16 carve_t16(char* smem, int nblk) {
    ActT16 a;
    a.sel = smem;
    a.d16 = (float*)(smem + (size_t)nblk * T16_SEL_STRIDE);
    a.red = (double*)(a.d16 + ((nblk + 1) & ~1));
    return a;
}
static inline size_t lds_bytes_t16(int K) {
    const size_t nblk = (size_t)K / QB;
    return nblk * T16_SEL_STRIDE + ((nblk + 1) & ~(size_t)1) * 4 + 32 * 8;
}
// Panama quantizeQ8_512 (PTO:1684-1723) of 8 consecutive values per lane exactly as quad_quantize_store_p16; the codes are filed
// as selector entries: the lanes of a quad hold elements 0-7, 8-15 (codes of a[t]) and 16-23, 24-31 (codes of a[t+16]) of one
// block; sub and sub^2 exchange their packed codes, the lower lane files t = 8*(sub&1)+0..3, the upper one t = ...+4..7.
__device__ __forceinline__ void quad_quantize_store_t16(const float (&y)[8], int unit, const ActT16& a) {
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) amax = fmaxf(amax, fabsf(y[i]));
    amax = fmaxf(amax, dpp_f<0xB1>(amax));
    amax = fmaxf(amax, dpp_f<0x4E>(amax));
    const float d = amax / 127.0f;
    const float id = (amax != 0.0f) ? 127.0f / amax : 0.0f;
    int q[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float v = y[i] * id;
        v = v + 0.5f;
        q[i] = f2b(v);
    }
    const int px = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
    const int py = q[4] | (q[5] << 8) | (q[6] << 16) | (q[7] << 24);
    const int ox = __builtin_amdgcn_mov_dpp(px, 0x4E, 0xf, 0xf, true);   // the lane two over (sub ^ 2)
    const int oy = __builtin_amdgcn_mov_dpp(py, 0x4E, 0xf, 0xf, true);
    const int blk = unit >> 2, sub = unit & 3;
    const bool lower = sub < 2;
    const int lo_dw = lower ? px : oy, hi_dw = lower ? ox : py;   // codes of a[t0..t0+3] / a[16+t0..16+t0+3]
    const int t0 = (sub & 1) * 8 + (lower ? 0 : 4);               // a multiple of 4: entry t0+i keeps byte i in place
    i32x2* dst = (i32x2*)(a.sel + (size_t)blk * T16_SEL_STRIDE) + t0;
    dst[0] = i32x2{lo_dw & 0x000000FF, hi_dw & 0x000000FF};
    dst[1] = i32x2{lo_dw & 0x0000FF00, hi_dw & 0x0000FF00};
    dst[2] = i32x2{lo_dw & 0x00FF0000, hi_dw & 0x00FF0000};
    dst[3] = i32x2{lo_dw & (int)0xFF000000, hi_dw & (int)0xFF000000};
    if (sub == 0) {
        a.d16[blk] = d * 0.0625f;
        *(i32x2*)(a.sel + (size_t)blk * T16_SEL_STRIDE + 128) = i32x2{0, 0};
    }
}
template <int PRO, int UM, int NT>
__device__ __forceinline__ void stage_finish_t16(const GemvParams& p, const ActT16& a, ActRegsP16<UM>& r) {
    static_assert(PRO == PRO_RMS_Q8 || PRO == PRO_QUANT_Q8, "t16 prologues: RMSNorm+Q8 or plain Q8");
    const int units = p.K / 8;
    float fs = 1.0f;
    if (PRO == PRO_RMS_Q8) fs = rms_factor_p16<UM, NT>(p, r, a.red);
#pragma unroll
    for (int u = 0; u < UM; u++) {
        const int unit = threadIdx.x + u * NT;
        if (unit < units) {
            float y[8];
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = (PRO == PRO_RMS_Q8) ? r.wv[u][i] * (fs * r.xv[u][i]) : r.xv[u][i];   // (0 + w) * ((float)ss * x)
            quad_quantize_store_t16(y, unit, a);
        }
    }
    lds_barrier();
}

// ------------------------------------------------------------------------------------------------ the GEMV
// Wave `w < aw` of a workgroup owns tiles [tile0, tile0 + tiles_per_wave) -- a contiguous stream of 1 KiB (+ 256 B of scales) per
// q step (4 blocks) -- through a ring of D prefetched steps; waves [aw, NT/64) only help with the activation prologue.
// The q steps are software-pipelined (a lone wave issues one instruction per ~4-5 cycles and nothing else hides its latencies):
// the selector of step n+1 is read, and the chain of step n-1 runs, beside the MFMAs of step n; two register sets alternate, so
// D is even and the host picks it as a divisor of K/128.
// EPI_SILU_MUL: tiles packed in mode 1, nrows = hidden units, out[8u + j] = SiLU(gate_j) * up_j (MLPBlock.java:132-142).
// EPI_STORE / EPI_RESID: tiles packed in mode 0, out[16u + j] (+ resid).
template <int PRO, int EPI, int D, int UM, int NT>
__global__ __launch_bounds__(NT) void gemv_t16_kernel(GemvParams p, int tiles_per_wave, int aw) {
    static_assert(D % 2 == 0, "the pipelined form alternates two register sets");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nblk = p.K / QB, nq = nblk >> 2;                 // host: nq % D == 0
    const ActT16 a = carve_t16(smem, nblk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int ntiles = (EPI == EPI_SILU_MUL) ? p.nrows >> 3 : p.nrows >> 4;
    int tile0 = wave < aw ? (blockIdx.x * aw + wave) * tiles_per_wave : ntiles;
    if (tile0 > ntiles) tile0 = ntiles;
    int tile1 = tile0 + tiles_per_wave;
    if (tile1 > ntiles) tile1 = ntiles;
    const int items = (tile1 - tile0) * nq;

    ActRegsP16<UM> ar;
    if (items == 0) {
        // helper wave: its own copy of the prologue (same barriers).  The two paths must not join (see gemv_i8q4_p16_kernel)
        stage_issue_p16<PRO, UM, NT>(p, ar);
        stage_finish_t16<PRO, UM, NT>(p, a, ar);
        return;
    }
    stage_issue_p16<PRO, UM, NT>(p, ar);                       // activation loads first: vmcnt retires oldest-first
    i32x4 wq[D];
    f32x4t sq[D];
    const i32x4* wp = (const i32x4*)p.w + (size_t)tile0 * nq * 64 + lane;
    const f32x4t* sp_ = (const f32x4t*)p.ws + (size_t)tile0 * nq * 16 + j;
    int li = 0;
    auto issue = [&](i32x4& w, f32x4t& s) __attribute__((always_inline)) {
        const int i = li < items ? li : items - 1;              // branch-free: past the end the last step is requested again (unused)
        w = __builtin_nontemporal_load(wp + (size_t)i * 64);
        s = __builtin_nontemporal_load(sp_ + (size_t)i * 16);
        ++li;
    };
#pragma unroll
    for (int d = 0; d < D; d++) {                              // the ring is in flight across the prologue
        issue(wq[d], sq[d]);
        __builtin_amdgcn_sched_barrier(0);
    }
    stage_finish_t16<PRO, UM, NT>(p, a, ar);

    // this lane's selector address: the 16 lanes (t = j, g == t/4) read their entry, the others the zero word
    const char* sel = a.sel + (((j >> 2) == g) ? j * 8 : 128);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, park = 0.f;
    int ct = tile0;
    struct Sel { long a[4]; f32x4t da; };
    auto read_sel = [&](int q, Sel& x) __attribute__((always_inline)) {
        const char* sp = sel + (size_t)(4 * q) * T16_SEL_STRIDE;
#pragma unroll
        for (int d = 0; d < 4; d++) x.a[d] = *(const long*)(sp + d * T16_SEL_STRIDE);
        x.da = *(const f32x4t*)(a.d16 + 4 * q);
    };
    auto mfmas = [&](const i32x4& w, const f32x4t& sc, const Sel& x, i32x4 (&dd)[4], f32x4t& s) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int lo = nib_lo16(w[d]), hi = nib_hi16(w[d]);
            const long b = (long)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
            const i32x4 z = {0, 0, 0, 0};
            dd[d] = __builtin_amdgcn_mfma_i32_16x16x32_i8(x.a[d], b, z, 0, 0, 0);
        }
        // (da/16) * sb: the 1/16 undoes the 16*(nib-8) unpack (a power of two: every rounding unchanged)
        s[0] = mul1(x.da[0], sc[0]); s[1] = mul1(x.da[1], sc[1]); s[2] = mul1(x.da[2], sc[2]); s[3] = mul1(x.da[3], sc[3]);
    };
    // the converts stay in C (hipcc must see the MFMA -> VALU read hazard; asm hides it), the fmas are asm: hipcc's SLP vectoriser
    // otherwise packs them into v_pk_fma_f32, which costs several issue slots beside MFMAs on this chip
    auto chain = [&](const i32x4 (&dd)[4], const f32x4t& s) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const float f0 = (float)dd[d][0], f1 = (float)dd[d][1], f2 = (float)dd[d][2], f3 = (float)dd[d][3];
            acc0 = fma1(s[d], f0, acc0); acc1 = fma1(s[d], f1, acc1); acc2 = fma1(s[d], f2, acc2); acc3 = fma1(s[d], f3, acc3);
        }
    };
    auto tile_end = [&]() __attribute__((always_inline)) {
        // reduceLanes(ADD) as the halving tree over t = 4g + i: (t, t+8) = lanes l, l^32; (t, t+4) = lanes l, l^16; then registers
        // (i, i+2), then (0, 1) -- jo_reduce16's association; every lane group ends with the 16 row sums
        float a0 = acc0 + __shfl_xor(acc0, 32), a1 = acc1 + __shfl_xor(acc1, 32), a2 = acc2 + __shfl_xor(acc2, 32), a3 = acc3 + __shfl_xor(acc3, 32);
        a0 = a0 + __shfl_xor(a0, 16); a1 = a1 + __shfl_xor(a1, 16); a2 = a2 + __shfl_xor(a2, 16); a3 = a3 + __shfl_xor(a3, 16);
        float r = (a0 + a2) + (a1 + a3);
        if (EPI == EPI_SILU_MUL) {
            const float up = dpp_f<0x128>(r);                   // row_ror:8: lane j < 8 (gate row) receives the up row of its unit
            r = silu_ref(r) * up;                               // MLPBlock.java:132-142; SiLU in double, once per tile
        }
        // results are parked: lane group n & 3 keeps tile n's, one store per 4 tiles (normally once, after the loop)
        const int n = (ct - tile0) & 3;
        if (g == n) park = r;
        if (n == 3 || ct + 1 == tile1) {
            const int tt = ct - n + g;
            if (tt <= ct) {
                if (EPI == EPI_SILU_MUL) {
                    if (j < 8) p.out[(size_t)tt * 8 + j] = park;
                } else {
                    const int row = tt * 16 + j;
                    float v = park;
                    if (EPI == EPI_RESID) v = v + p.resid[row];  // accumulate(...) TransformerBlock.java:185,203
                    p.out[row] = v;
                }
            }
        }
        acc0 = acc1 = acc2 = acc3 = 0.f;
        ++ct;
    };
    Sel x[2];
    i32x4 dd[2][4];
    f32x4t sv[2];
    // the chain behind the very first step is a no-op: fma(0, 0, acc) = acc
#pragma unroll
    for (int k = 0; k < 4; k++) dd[1][k] = i32x4{0, 0, 0, 0};
    sv[1] = f32x4t{0.f, 0.f, 0.f, 0.f};
    read_sel(0, x[0]);
    int cq = 0;                                                 // q (within its tile) of the step whose MFMAs are issued next
    for (int it = 0; it < items; it += D) {
        if (it > 0 && cq == 0) {                                // the previous ring block closed a tile: its last chain first
            chain(dd[1], sv[1]);
            tile_end();
#pragma unroll
            for (int k = 0; k < 4; k++) dd[1][k] = i32x4{0, 0, 0, 0};
            sv[1] = f32x4t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int d = 0; d < D; d++) {
            const int c = d & 1, o = c ^ 1;
            int qn = cq + d + 1;
            qn = qn == nq ? 0 : qn;
            read_sel(qn, x[o]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(wq[d], sq[d], x[c], dd[c], sv[c]);
            __builtin_amdgcn_sched_barrier(0);
            chain(dd[o], sv[o]);                                // the previous step's sums while this step's MFMAs run
            __builtin_amdgcn_sched_barrier(0);
            issue(wq[d], sq[d]);                                // the slot's registers were used up by mfmas(): refill in place
            __builtin_amdgcn_sched_barrier(0);
        }
        cq += D;
        if (cq == nq) cq = 0;
    }
    chain(dd[1], sv[1]);
    tile_end();
}

}  // namespace jh
