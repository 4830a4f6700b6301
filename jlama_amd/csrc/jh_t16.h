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
static __global__ __launch_bounds__(256) void t16_pack_kernel(const i32x4* __restrict__ w, const float* __restrict__ ws, const i32x4* __restrict__ w2,
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
#define JH_TSTAMP(kk) do { if (p.dbg && lane == 0) p.dbg[((size_t)blockIdx.x * (NT / 64) + wave) * 8 + (kk)] = wall_clock64(); } while (0)
    JH_TSTAMP(0);
    if (items == 0) {
        // helper wave: its own copy of the prologue (same barriers).  The two paths must not join (see gemv_i8q4_p16_kernel)
        stage_issue_p16<PRO, UM, NT>(p, ar);
        stage_finish_t16<PRO, UM, NT>(p, a, ar);
        JH_TSTAMP(2);
        return;
    }
    stage_issue_p16<PRO, UM, NT>(p, ar);                       // activation loads first: vmcnt retires oldest-first
    JH_TSTAMP(6);
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
    // (no barrier / pause between the row's requests and the ring's, unlike gemv_i8q4_p16_kernel: this launch is HBM-bound from its
    // first request on -- same-box A/B on the 8B gate|up: as is 14.1 us / 697.6 tok/s, a workgroup barrier here 14.4 / 693.9,
    // s_sleep 8 in front 14.5, s_sleep 2 between the requests 14.3)
#pragma unroll
    for (int d = 0; d < D; d++) {                              // the ring is in flight across the prologue
        issue(wq[d], sq[d]);
        __builtin_amdgcn_sched_barrier(0);
    }
    JH_TSTAMP(1);
    stage_finish_t16<PRO, UM, NT>(p, a, ar);
    JH_TSTAMP(2);

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
    JH_TSTAMP(3);
    JH_TSTAMP(4);
    chain(dd[1], sv[1]);
    tile_end();
    JH_TSTAMP(5);
#undef JH_TSTAMP
}

// ================================================================================================ prompt rows (M > 1)
// AbstractModel.batchForward in reference order on the matrix pipe.  The reference's batch GEMM (GemmerI8Q4_512 tiles,
// PTO:852-1043) keeps ONE 16-lane accumulator per output whatever the tile, so a prompt row's result does not depend on how many
// rows are processed together: every (prompt row, weight row) pair is the GEMV's chain.  Per (prompt row m, 16-row weight tile,
// block) the only ordered work is 16 x 16 fmas; the pair sums come from ONE MFMA.  With M rows the converts of the integer form
// would equal the fmas in VALU work, so this kernel uses the F16 MFMA with F32 output instead: int8 codes (|a| <= 127) and
// nib - 8 are exact in f16, their products and the two-term sums exact in f32 whatever the matrix pipe's internal order --
// D[t][j] IS (float)(lo_t*a[t] + hi_t*a[t+16]), and the chain step is the reference's own  acc = fma(da*sb, D, acc)  (no 1/16
// trick here: nib - 8 is unpacked exactly, 0x6400|nib = 1024+nib then a packed f16 add of -1032).
// Operands: weights in T16 order (the GEMV's copy; a weight tile is unpacked once per block and serves MT prompt rows, and is
// fetched from HBM once: the row tiles of a weight slice are neighbours in the launch order on ONE XCD, so all but the first
// hit in its L2); activations as ready-made one-hot selector operands [row][block][16 entries x 16 B] (rows_act_t16_kernel),
// staged through LDS in chunks of KC blocks (double-buffered) for the waves of the workgroup.
//   workgroup = CW waves x 2 tiles (CW * 32 weight rows) x MT = 8 prompt rows; 256 threads (up to 256 VGPRs: 64 accumulators).
struct RowsT16Params {
    const float* x; int ldx;          // [rows][ldx] F32
    const float* nw; float eps;       // PRO_RMS_Q8
    int K;
    i32x4* asel;                      // out: [rows][nblk][16] one-hot selector entries, 8 halves each (k slots 0-3: a[t] at slot t&3; 4-7: a[t+16])
    float* ad;                        // out: [nblk][ad_stride] block scales (d = max/127), row-minor so that a row tile's scales are contiguous
    int ad_stride;
};
template <int PRO, int UM>
__global__ __launch_bounds__(P16_THREADS) void rows_act_t16_kernel(RowsT16Params rp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nblk = rp.K / QB;
    const ActT16 a = carve_t16(smem, nblk);
    GemvParams p{};
    p.x = rp.x + (size_t)blockIdx.x * rp.ldx; p.nw = rp.nw; p.eps = rp.eps; p.K = rp.K;
    ActRegsP16<UM> ar;
    stage_issue_p16<PRO, UM, P16_THREADS>(p, ar);
    stage_finish_t16<PRO, UM, P16_THREADS>(p, a, ar);          // the GEMV's own prologue: int8 selector table + d/16 in LDS
    i32x4* dst = rp.asel + (size_t)blockIdx.x * nblk * 16;
    for (int i = threadIdx.x; i < nblk * 16; i += P16_THREADS) {
        const int b = i >> 4, t = i & 15, y = t & 3;
        const i32x2 e = *(const i32x2*)(a.sel + (size_t)b * T16_SEL_STRIDE + t * 8);
        const int lo = (int)(int8_t)(e.x >> (8 * y)), hi = (int)(int8_t)(e.y >> (8 * y));
        const _Float16 hl = (_Float16)lo, hh = (_Float16)hi;   // exact: |code| <= 127
        const unsigned ul = (unsigned)__builtin_bit_cast(unsigned short, hl) << (16 * (y & 1));
        const unsigned uh = (unsigned)__builtin_bit_cast(unsigned short, hh) << (16 * (y & 1));
        i32x4 v = {0, 0, 0, 0};
        v[y >> 1] = (int)ul;
        v[2 + (y >> 1)] = (int)uh;
        dst[i] = v;
    }
    for (int b = threadIdx.x; b < nblk; b += P16_THREADS) rp.ad[(size_t)b * rp.ad_stride + blockIdx.x] = a.d16[b] * 16.0f;   // d (x16: exact)
}

struct GemmT16Params {
    const i32x4* w; const f32x4t* ws;      // T16 copy of the weight (mode 1 for EPI_SILU_MUL)
    int ntiles, K, M;
    const i32x4* asel; const float* ad; int ad_stride;   // rows_act_t16_kernel's output
    float* out; int ldc;                   // [M][ldc]
    const float* resid; int ldr;           // EPI_RESID
    int nslices, nrt;                      // weight slices of 2*CW tiles, row tiles of MT rows
};
constexpr int GT16_KC = 16;                // blocks per staged activation chunk
constexpr int GT16_ENTRY = 272;            // bytes per (row, block) in LDS: 16 entries x 16 B + 16 zero bytes (the inactive lanes' operand)
constexpr int GT16_SKEW = 2;               // the MFMAs run this many prompt rows ahead of the chains that read them
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
static inline size_t lds_bytes_gemm_t16(int MT) { return (size_t)2 * MT * GT16_KC * GT16_ENTRY + (size_t)2 * MT * GT16_KC * 4; }

// 8 nibbles of one dword -> the MFMA B operand of its lane: halves 0-3 = lo nibbles - 8 (elements 4g..4g+3), 4-7 = hi nibbles - 8.
// 11 VALU per dword: two masks + one shift for the whole dword, then one v_perm per half pair that also drops in the 0x64 exponent
// bytes (0x6400 | nib = 1024 + nib, ulp 1 there), and the packed f16 add of -1032 (1024 + nib - 1032 = nib - 8, exact).
__device__ __forceinline__ f16x8 nib8_to_f16(int w) {
    const unsigned u = (unsigned)w, lo = u & 0x0F0F0F0Fu, hi = (u >> 4) & 0x0F0F0F0Fu;
    const unsigned k = 0x64646464u;                            // perm selector bytes 4-7 take from src0 = k
    const unsigned l01 = __builtin_amdgcn_perm(k, lo, 0x04010400u), l23 = __builtin_amdgcn_perm(k, lo, 0x04030402u);
    const unsigned h01 = __builtin_amdgcn_perm(k, hi, 0x04010400u), h23 = __builtin_amdgcn_perm(k, hi, 0x04030402u);
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 c = {(_Float16)-1032.0f, (_Float16)-1032.0f};
    const h2 a0 = __builtin_bit_cast(h2, l01) + c, a1 = __builtin_bit_cast(h2, l23) + c, a2 = __builtin_bit_cast(h2, h01) + c, a3 = __builtin_bit_cast(h2, h23) + c;
    f16x8 r;
    r[0] = a0[0]; r[1] = a0[1]; r[2] = a1[0]; r[3] = a1[1]; r[4] = a2[0]; r[5] = a2[1]; r[6] = a3[0]; r[7] = a3[1];
    return r;
}
// acc = fma(s, d, acc) as a plain C fma (hipcc must see the MFMA -> VALU hazard on d), fenced so that the SLP vectoriser cannot
// re-pair neighbouring chains behind our back
__device__ __forceinline__ void fma_c(float& acc, float s, float d) {
    acc = __builtin_fmaf(s, d, acc);
    asm volatile("" : "+v"(acc));
}
// two chains per instruction: v_pk_fma_f32 with the scale broadcast by op_sel (hipcc folds the splat), the two D registers and the
// two accumulators as even-aligned pairs -- each half is the same IEEE fma as v_fmac_f32.  Measured and NOT used (GT16_PK = 0):
// on this chip v_pk_fma_f32 costs 1.75x a v_fmac_f32 (tools/pkfma_lab.hip: 1.23 vs 1.40 ns per F32 FMA wave-op at 4 waves per SIMD),
// beside MFMAs it is slower than the scalar pair (hipcc itself un-packs the ones it schedules behind an MFMA), gate|up 550 -> 609 us
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fma_pk(f32x2& acc, f32x2 s, f32x2 d) {
    acc = __builtin_elementwise_fma(s, d, acc);
    asm volatile("" : "+v"(acc));
}
#ifndef GT16_PK
#define GT16_PK 0
#endif

template <int EPI, int MT, int CW>
__global__ __launch_bounds__(CW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_t16_kernel(GemmT16Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KC = GT16_KC, NT = CW * 64;
    const int nblk = p.K / QB, nq = nblk >> 2, nchunks = nblk / KC;   // host: nblk % KC == 0
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    // launch order: block x runs on XCD x & 7 (observed placement; only speed depends on it).  The row tiles of one weight slice are
    // consecutive on one XCD: the first fetches the slice from HBM, the others find it in that XCD's L2.
    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;
    const int rt = kx % p.nrt, slice = (kx / p.nrt) * 8 + xcd;
    if (slice >= p.nslices) return;
    const int m0 = rt * MT;
    int tA = slice * (2 * CW) + 2 * wave, tB = tA + 1;
    const bool okA = tA < p.ntiles, okB = tB < p.ntiles;
    tA = okA ? tA : p.ntiles - 1; tB = okB ? tB : p.ntiles - 1;
    char* selbuf = smem;                                        // [2][MT][KC][272]
    float* dabuf = (float*)(smem + (size_t)2 * MT * KC * GT16_ENTRY);   // [2][KC][MT]
    // zero words (the operand of the 48 lanes that are not (t, t/4)): written once, never overwritten by the staging
    for (int i = threadIdx.x; i < 2 * MT * KC; i += NT) *(i32x4*)(selbuf + (size_t)i * GT16_ENTRY + 256) = i32x4{0, 0, 0, 0};
    auto stage = [&](int c, int buf) __attribute__((always_inline)) {   // activation chunk c -> LDS buffer buf
        for (int i = threadIdx.x; i < MT * KC * 16; i += NT) {
            const int e = i & 15, mb = i >> 4, kb = mb % KC, m = mb / KC;
            int mm = m0 + m;
            mm = mm < p.M ? mm : p.M - 1;                        // rows past M replicate the last row (never stored)
            const i32x4 v = p.asel[((size_t)mm * nblk + (size_t)c * KC + kb) * 16 + e];
            *(i32x4*)(selbuf + ((size_t)(buf * MT + m) * KC + kb) * GT16_ENTRY + e * 16) = v;
        }
        for (int i = threadIdx.x; i < MT * KC; i += NT) {
            const int m = i % MT, kb = i / MT;
            int mm = m0 + m;
            mm = mm < p.M ? mm : p.M - 1;
            dabuf[(buf * KC + kb) * MT + m] = p.ad[(size_t)(c * KC + kb) * p.ad_stride + mm];
        }
    };
    const char* asel_lane = selbuf + (((j >> 2) == g) ? j * 16 : 256);
    const i32x4* wA = p.w + (size_t)tA * nq * 64 + lane;
    const i32x4* wB = p.w + (size_t)tB * nq * 64 + lane;
    const f32x4t* sA = p.ws + (size_t)tA * nq * 16 + j;
    const f32x4t* sB = p.ws + (size_t)tB * nq * 16 + j;
    f32x2 acc[MT][2][2];                                        // chains t = 4g + {0,1} and 4g + {2,3} of (prompt row, tile)
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int i = 0; i < 2; i++) acc[m][u][i] = f32x2{0.f, 0.f};
    stage(0, 0);
    i32x4 wa = wA[0], wb = wB[0];                               // (default cache policy: the sibling row tiles re-read these lines from L2)
    f32x4t sa = sA[0], sb = sB[0];
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        const int buf = c & 1;
        if (c + 1 < nchunks) stage(c + 1, buf ^ 1);             // lands while this chunk is consumed; the barrier below publishes it
        // the selector operands (and block scales) of block kb+1 are read from LDS while block kb's MFMAs and chains run: an LDS
        // round trip per (row, block) in front of its MFMA would otherwise be the critical path (two waves per SIMD do not hide it)
        f16x8 av[2][MT];
        f32x4t dav[2][MT / 4];
        auto read_block = [&](int kb, f16x8 (&a)[MT], f32x4t (&dv)[MT / 4]) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < MT; m++) a[m] = *(const f16x8*)(asel_lane + ((size_t)(buf * MT + m) * KC + kb) * GT16_ENTRY);
#pragma unroll
            for (int h = 0; h < MT / 4; h++) dv[h] = *(const f32x4t*)(dabuf + (buf * KC + kb) * MT + 4 * h);
        };
        read_block(0, av[0], dav[0]);
#pragma unroll
        for (int qq = 0; qq < KC / 4; qq++) {
            const int q = c * (KC / 4) + qq;
            const i32x4 wa_c = wa, wb_c = wb;
            const f32x4t sa_c = sa, sb_c = sb;
            const int qn = q + 1 < nq ? q + 1 : q;              // branch-free prefetch of the next q step (the last one twice)
            wa = wA[(size_t)qn * 64]; wb = wB[(size_t)qn * 64];
            sa = sA[(size_t)qn * 16]; sb = sB[(size_t)qn * 16];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int kb = 4 * qq + d, cb = kb & 1;
                if (kb + 1 < KC) read_block(kb + 1, av[cb ^ 1], dav[cb ^ 1]);
                const f16x8 bA = nib8_to_f16(wa_c[d]), bB = nib8_to_f16(wb_c[d]);
                // the MFMAs of prompt rows m+1, m+2 are issued before the chains of row m read their results (SK + 1 result sets): the
                // matrix pipe runs beside the VALU instead of in front of it
                constexpr int SK = GT16_SKEW;
                f32x4t dA[SK + 1], dB[SK + 1];
                auto pair = [&](int m, f32x4t& ra, f32x4t& rb) __attribute__((always_inline)) {
                    const f32x4t z = {0.f, 0.f, 0.f, 0.f};
                    ra = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[cb][m], bA, z, 0, 0, 0);
                    rb = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[cb][m], bB, z, 0, 0, 0);
                };
#pragma unroll
                for (int m = 0; m < SK; m++) pair(m, dA[m], dB[m]);
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    const int cur = m % (SK + 1);
                    if (m + SK < MT) pair(m + SK, dA[(m + SK) % (SK + 1)], dB[(m + SK) % (SK + 1)]);
                    const float da = dav[cb][m >> 2][m & 3];
                    if constexpr (GT16_PK) {
                        const f32x2 s = f32x2{da, da} * f32x2{sa_c[d], sb_c[d]};   // afr[blk] * bfr[blk] (PTO:819), both tiles in one v_pk_mul_f32
                        fma_pk(acc[m][0][0], s.xx, dA[cur].xy); fma_pk(acc[m][0][1], s.xx, dA[cur].zw);
                        fma_pk(acc[m][1][0], s.yy, dB[cur].xy); fma_pk(acc[m][1][1], s.yy, dB[cur].zw);
                    } else {
                        const float s0 = mul1(da, sa_c[d]), s1 = mul1(da, sb_c[d]);
                        float a;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            a = acc[m][0][i >> 1][i & 1]; fma_c(a, s0, dA[cur][i]); acc[m][0][i >> 1][i & 1] = a;
                            a = acc[m][1][i >> 1][i & 1]; fma_c(a, s1, dB[cur][i]); acc[m][1][i >> 1][i & 1] = a;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();
    }
    // halving tree per (prompt row, tile): (t, t+8) = lanes l, l^32; (t, t+4) = l, l^16; registers (i, i+2); (0, 1).  Every lane group
    // ends with the 16 row sums; group g stores prompt rows m = g and g + 4.
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int tile = u ? tB : tA;
        const bool ok = u ? okB : okA;
        float mine0 = 0.f, mine1 = 0.f;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            float a0 = acc[m][u][0].x, a1 = acc[m][u][0].y, a2 = acc[m][u][1].x, a3 = acc[m][u][1].y;
            a0 = a0 + __shfl_xor(a0, 32); a1 = a1 + __shfl_xor(a1, 32); a2 = a2 + __shfl_xor(a2, 32); a3 = a3 + __shfl_xor(a3, 32);
            a0 = a0 + __shfl_xor(a0, 16); a1 = a1 + __shfl_xor(a1, 16); a2 = a2 + __shfl_xor(a2, 16); a3 = a3 + __shfl_xor(a3, 16);
            const float r = (a0 + a2) + (a1 + a3);
            if ((m & 3) == g) { if (m < 4) mine0 = r; else mine1 = r; }
        }
#pragma unroll
        for (int h = 0; h < MT / 4; h++) {
            const int mrow = m0 + g + 4 * h;
            float v = h ? mine1 : mine0;
            if (EPI == EPI_SILU_MUL) {
                const float up = dpp_f<0x128>(v);               // row_ror:8: the up row of the lane's hidden unit (tile = 8 gate + 8 up rows)
                v = silu_ref(v) * up;                           // MLPBlock.java:132-142
                if (ok && mrow < p.M && j < 8) p.out[(size_t)mrow * p.ldc + (size_t)tile * 8 + j] = v;
            } else {
                const int row = tile * 16 + j;
                if (EPI == EPI_RESID && ok && mrow < p.M) v = v + p.resid[(size_t)mrow * p.ldr + row];   // TransformerBlock.java:185,203
                if (ok && mrow < p.M) p.out[(size_t)mrow * p.ldc + row] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ the same GEMM on v_mfma_f32_32x32x16_f16
// Why: beside the chains' v_fmac the MFMA does not hide on this chip -- tools/pkfma_lab.hip: one 16x16x32 MFMA + 4 v_fmac (one prompt
// row x one tile x one block) takes 12.9 ns per SIMD at two waves, the sum of its parts (7.0 + 4 x 1.4) -- so the matrix pipe's time
// per (row, tile, block) counts in full, and the one-hot operand wastes most of it.  The 32x32x16 shape lets FOUR prompt rows share
// an MFMA: 8 pipe cycles per (row, tile, block) instead of 16.  Same operands in HBM and LDS, other register roles:
//   A (32 rows x 16 k)  row 8 * mq + t' = chain t' (MFMA 1) or 8 + t' (MFMA 2) of prompt row 4 * quad + mq
//   B (16 k x 32 cols)  cols 0-15 = weight rows of tile A, 16-31 = tile B
//   K = 16 holds half a block's 32 elements:
//     MFMA 1: lane group h = lane >> 5 supplies dword h     of the block (elements 4h..4h+3 low nibbles, 16+4h.. high): chains t < 8
//     MFMA 2: lane group h             supplies dword 2 + h                                                          : chains t >= 8
//   chain t' (and 8 + t') is non-zero only in lane group t' >> 2: a lane reads its two 16-byte selector entries there, the zero word
//   elsewhere -- two LDS reads per (row quad, block), a quarter of the 16x16 form's traffic per row.
//   D register i of lane (col j' = lane & 31, h): prompt row 4 * quad + (i >> 2), chain t = 4h + (i & 3) (MFMA 1) or 8 + 4h + (i & 3).
// Per (prompt row, block, tile PAIR): half an MFMA issue (was two), ONE scale product (the lane owns one weight row; was two), 8 v_fmac.
template <int EPI, int MT, int CW>
__global__ __launch_bounds__(CW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_t16x_kernel(GemmT16Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KC = GT16_KC, NT = CW * 64, NQ = MT / 4;
    static_assert(MT == 8, "gemm_t16x_kernel: two row quads per block (the MFMA / chain schedule below is written out for them)");
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int nblk = p.K / QB, nq = nblk >> 2, nchunks = nblk / KC;   // host: nblk % KC == 0
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int jp = lane & 31, h = lane >> 5, u = jp >> 4, j = jp & 15;
    const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3;             // launch order as gemm_t16_kernel: row tiles of a slice are neighbours on one XCD
    const int rt = kx % p.nrt, slice = (kx / p.nrt) * 8 + xcd;
    if (slice >= p.nslices) return;
    const int m0 = rt * MT;
    int tile = slice * (2 * CW) + 2 * wave + u;                       // the tile of this lane's weight row
    const bool ok = tile < p.ntiles;
    tile = ok ? tile : p.ntiles - 1;
    char* selbuf = smem;                                              // [2][MT][KC][272]
    float* dabuf = (float*)(smem + (size_t)2 * MT * KC * GT16_ENTRY); // [2][KC][MT]
    for (int i = threadIdx.x; i < 2 * MT * KC; i += NT) *(i32x4*)(selbuf + (size_t)i * GT16_ENTRY + 256) = i32x4{0, 0, 0, 0};
    auto stage = [&](int c, int buf) __attribute__((always_inline)) {
        for (int i = threadIdx.x; i < MT * KC * 16; i += NT) {
            const int e = i & 15, mb = i >> 4, kb = mb % KC, m = mb / KC;
            int mm = m0 + m;
            mm = mm < p.M ? mm : p.M - 1;
            const i32x4 v = p.asel[((size_t)mm * nblk + (size_t)c * KC + kb) * 16 + e];
            *(i32x4*)(selbuf + ((size_t)(buf * MT + m) * KC + kb) * GT16_ENTRY + e * 16) = v;
        }
        for (int i = threadIdx.x; i < MT * KC; i += NT) {
            const int m = i % MT, kb = i / MT;
            int mm = m0 + m;
            mm = mm < p.M ? mm : p.M - 1;
            dabuf[(buf * KC + kb) * MT + m] = p.ad[(size_t)(c * KC + kb) * p.ad_stride + mm];
        }
    };
    // as an A supplier the lane is (prompt row mq of its quad, chain t' or 8 + t', k group h)
    const int mq = jp >> 3, tq = jp & 7;
    const bool mine = (tq >> 2) == h;
    const char* a1_lane = selbuf + (size_t)mq * KC * GT16_ENTRY + (mine ? tq * 16 : 256);
    const char* a2_lane = selbuf + (size_t)mq * KC * GT16_ENTRY + (mine ? (8 + tq) * 16 : 256);
    const i32x4* w1 = p.w + (size_t)tile * nq * 64 + 16 * h + j;       // dword h of blocks 4q..4q+3 of the lane's weight row
    const i32x4* w2 = w1 + 32;                                        // dword 2 + h
    const f32x4t* sw = p.ws + (size_t)tile * nq * 16 + j;
    float acc[MT][8];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc[m][i] = 0.f;
    stage(0, 0);
    i32x4 wa = w1[0], wb = w2[0];
    f32x4t sc = sw[0];
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        const int buf = c & 1;
        if (c + 1 < nchunks) stage(c + 1, buf ^ 1);
        f16x8 av[2][NQ][2];                                           // [set][row quad][MFMA 1 / 2]
        f32x4t dav[2][MT / 4];
        auto read_block = [&](int kb, f16x8 (&a)[NQ][2], f32x4t (&dv)[MT / 4]) __attribute__((always_inline)) {
#pragma unroll
            for (int qd = 0; qd < NQ; qd++) {
                a[qd][0] = *(const f16x8*)(a1_lane + ((size_t)(buf * MT + 4 * qd) * KC + kb) * GT16_ENTRY);
                a[qd][1] = *(const f16x8*)(a2_lane + ((size_t)(buf * MT + 4 * qd) * KC + kb) * GT16_ENTRY);
            }
#pragma unroll
            for (int q4 = 0; q4 < MT / 4; q4++) dv[q4] = *(const f32x4t*)(dabuf + (buf * KC + kb) * MT + 4 * q4);
        };
        read_block(0, av[0], dav[0]);
#pragma unroll
        for (int qq = 0; qq < KC / 4; qq++) {
            const int q = c * (KC / 4) + qq;
            const i32x4 wa_c = wa, wb_c = wb;
            const f32x4t sc_c = sc;
            const int qn = q + 1 < nq ? q + 1 : q;
            wa = w1[(size_t)qn * 64]; wb = w2[(size_t)qn * 64];
            sc = sw[(size_t)qn * 16];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int kb = 4 * qq + d, cb = kb & 1;
                if (kb + 1 < KC) read_block(kb + 1, av[cb ^ 1], dav[cb ^ 1]);
                const f16x8 b1 = nib8_to_f16(wa_c[d]), b2 = nib8_to_f16(wb_c[d]);
                const float swd = sc_c[d];
                f32x16 D1[NQ], D2[NQ];
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                auto mf = [&](int qd) __attribute__((always_inline)) {
                    D1[qd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[cb][qd][0], b1, z, 0, 0, 0);
                    D2[qd] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[cb][qd][1], b2, z, 0, 0, 0);
                };
                auto chains = [&](int qd) __attribute__((always_inline)) {
#pragma unroll
                    for (int mm = 0; mm < 4; mm++) {
                        const int m = 4 * qd + mm;
                        const float s = dav[cb][qd][mm] * swd;                 // afr[blk] * bfr[blk] (PTO:819): the reference's own product (-ffp-contract=off)
#pragma unroll
                        for (int r = 0; r < 4; r++) fma_c(acc[m][r], s, D1[qd][4 * mm + r]);
#pragma unroll
                        for (int r = 0; r < 4; r++) fma_c(acc[m][4 + r], s, D2[qd][4 * mm + r]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                mf(0); mf(1);                                                  // the second quad's MFMAs run beside the first quad's chains
                __builtin_amdgcn_sched_barrier(0);
                chains(0);
                chains(1);
            }
        }
        __syncthreads();
    }
    // halving tree per (prompt row, weight row): (t, t+8) = registers i, i+4; (t, t+4) = lanes l, l^32; registers (i, i+2); (0, 1)
    float res[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        float a0 = acc[m][0] + acc[m][4], a1 = acc[m][1] + acc[m][5], a2 = acc[m][2] + acc[m][6], a3 = acc[m][3] + acc[m][7];
        a0 = a0 + __shfl_xor(a0, 32); a1 = a1 + __shfl_xor(a1, 32); a2 = a2 + __shfl_xor(a2, 32); a3 = a3 + __shfl_xor(a3, 32);
        res[m] = (a0 + a2) + (a1 + a3);
    }
#pragma unroll
    for (int m = 0; m < MT; m++) {                                    // both lane groups hold every sum: group h stores the rows m with m % 2 == h
        const int mrow = m0 + m;
        float v = res[m];
        if (EPI == EPI_SILU_MUL) {
            const float up = dpp_f<0x128>(v);                         // row_ror:8 (tile = 8 gate + 8 up rows)
            v = silu_ref(v) * up;                                     // MLPBlock.java:132-142
            if (ok && mrow < p.M && j < 8 && (m & 1) == h) p.out[(size_t)mrow * p.ldc + (size_t)tile * 8 + j] = v;
        } else {
            const int row = tile * 16 + j;
            if (EPI == EPI_RESID && ok && mrow < p.M && (m & 1) == h) v = v + p.resid[(size_t)mrow * p.ldr + row];   // TransformerBlock.java:185,203
            if (ok && mrow < p.M && (m & 1) == h) p.out[(size_t)mrow * p.ldc + row] = v;
        }
    }
}

}  // namespace jh
