// jh_p16.h -- the decode path with every float accumulation in EXACTLY the order of the reference's Panama AVX-512 provider
// ("reference order"), built to stream at the same rate as the order-free kernels of jh_kernels.h.  Results are bit-identical
// to a plain-C restatement of that provider (the parity tests' checker), not merely inside the Q8 noise floor.
//
// Panama-512 order (FloatVector.SPECIES_512 = 16 float lanes), what has to be reproduced:
//   I8 x Q4  (GemmerI8Q4_512, PTO:807-850):  lane t:  acc_t = fma(da*sb, (float)(short)(lo_t*a[t] + hi_t*a[t+16]), acc_t)
//            over Q blocks in ascending K, then reduceLanes(ADD) = the halving tree (v[i]+v[i+8], +4, +2, +1).
//   F32 x Q4 (GemmerF32Q4_512, PTO:336-374): acc_t = fma(a[t], (float)(lo_t-8)*s, acc_t); acc_t = fma(a[t+16], (float)(hi_t-8)*s, acc_t)
//   F32 x F32 (GemmerF32, PTO:1086-1102):    acc_t = fma(a[l+t], b[l+t], acc_t) for l = 0,16,...; same tree.
//   softMax (VectorMath.java:69-90): float sum of exp in index order;  saxpy over V (PTO:2593-2611, 2648-2698): one fma chain
//            per output element over positions in ascending order.
//
// MI355X mapping ("p16": a 16-lane DPP row plays the 16 Panama lanes, a wave64 serves 4 weight rows):
//   * the chain of lane t over the Q blocks is sequential by definition, so ONE GPU lane owns (row, t) and walks K -- but in the
//     checkpoint's layout the bytes arrive the other way round: a 16-byte load is one whole Q4 block = byte t of 16 DIFFERENT
//     chains.  Round 3 transposed the 16x16 byte matrix of a group of 16 blocks in registers (28 DPP / v_perm ops per group, a
//     fifth of the kernel's VALU work).  Now the weights these kernels read are a resident copy in "P16T" order, made once per
//     weight by p16t_pack_kernel: inside every group of 16 blocks (256 bytes of a row) byte 16t + c = byte t of block c, so the
//     16-byte load of lane t IS byte t of the group's 16 blocks, no shuffles; the four rows of a quad are interleaved per group so
//     that a wave instruction reads 1 KiB contiguous; rows are padded to whole groups (zeros);
//   * nibbles become int8 16*(nib-8) = ((nib << 4) ^ 0x80) with two bit ops per dword (4 blocks), so the pair sum
//     lo*a[t] + hi*a[t+16] is ONE v_dot4_i32_i8 against a (a[t], a[t+16], 0, 0) activation word (byte-selected by v_perm), exact;
//     the 1/16 goes into the activation block scale (a power of two: every rounding is unchanged);
//   * the per-block scale product da*sb lives in the lane that loaded the block and reaches the 16 chains through the DPP
//     operand of v_fmac_f32 (row_newbcast): acc = fma(bcast(da*sb), (float)isum, acc) is ONE instruction per step;
//   * per step: perm + dot4 + cvt + fmac  (+ amortised 0.75 unpack) = 4.75 VALU ops for 2 weights (matrices with many rows per
//     CU take the MFMA form of jh_t16.h instead: there the pair sums cost no VALU work at all);
//   * a prefetch ring of D groups (16 blocks each) per lane keeps D KiB per wave in flight; the activation row is quantized once
//     per workgroup into LDS (same fused RMSNorm + Q8 prologue, Panama rule) as pair words, 4 blocks per 8-byte read.
// Compiled with -ffp-contract=off like the rest: every FMA is explicit.
#pragma once
#include <type_traits>
#include "jh_kernels.h"

namespace jh {

// reduceLanes(ADD) of a 16-lane row as the halving tree: after the rotate-by-8 step the row's values have period 8, so
// rotating by 4 / 2 / 1 pairs lane i with the partner the tree prescribes (float addition commutes): every lane of the
// row ends with ((v0+v8)+(v4+v12)) + ((v2+v10)+(v6+v14)) + ... in exactly jo_reduce16's association.
__device__ __forceinline__ float row16_tree_sum(float v) {
    v = v + dpp_f<0x128>(v);   // row_ror:8
    v = v + dpp_f<0x124>(v);   // row_ror:4
    v = v + dpp_f<0x122>(v);   // row_ror:2
    v = v + dpp_f<0x121>(v);   // row_ror:1
    return v;
}

__device__ __forceinline__ int perm_b(int hi_src, int lo_src, int sel) {   // selector byte 0-3: lo_src, 4-7: hi_src, 0x0C: zero
    return (int)__builtin_amdgcn_perm((unsigned)hi_src, (unsigned)lo_src, (unsigned)sel);
}

// P16T copy of a Q4 weight: row stride G * 256 bytes (G = groups of 16 blocks, the last one zero-padded); inside a group the 16-byte
// chunk t holds nibble pair t (elements t and t+16) of the group's 16 blocks, four blocks per dword in the nibble order below.
// One thread per output 16-byte chunk.
static __global__ __launch_bounds__(256) void p16t_pack_kernel(const uint8_t* __restrict__ w, int nrows, int nblk, int ldb, uint8_t* __restrict__ out) {
    const int G = (nblk + 15) >> 4;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)nrows * G * 16) return;
    const int t = (int)(idx & 15);
    const long long rg = idx >> 4;
    const int g = (int)(rg % G);
    const long long row = rg / G;
    const uint8_t* src = w + (size_t)row * ldb + (size_t)g * 256 + t;
    // where the chunk goes: the four rows of a row quad are INTERLEAVED per group -- [quad][group][row in quad][chunk t] -- so that the
    // wave instruction of a quad (lane = 16 r + t) reads 1 KiB contiguous, like the T16 copies (round 5: as four 256-byte pieces 2 KB
    // apart the LM head streamed at 4.4 TB/s against gate|up's 5.3, profiles/r05e_*)
    const long long oidx = ((row >> 2) * G + g) * 64 + (row & 3) * 16 + t;
    // dword d of the chunk = nibble pair t of blocks 4d..4d+3 (b0..b3), nibble positions (from bit 0):
    //   [lo_b1, lo_b0, hi_b1, hi_b0, lo_b3, lo_b2, hi_b3, hi_b2]
    // so that  x & 0xF0F0F0F0         = bytes [lo_b0, hi_b0, lo_b2, hi_b2] * 16  (one op)
    //          (x << 4) & 0xF0F0F0F0  = bytes [lo_b1, hi_b1, lo_b3, hi_b3] * 16  (two ops)
    // are the v_dot4 operands of the four blocks as they stand (against pair words placed in the low / high half): no byte shuffle
    i32x4 v = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 16; c++) {
        const unsigned byte = (16 * g + c < nblk) ? (unsigned)src[c * 16] : 0u;
        const unsigned lo = byte & 15u, hi = byte >> 4;
        const int b = c & 3;
        const int plo = (b == 0) ? 4 : (b == 1) ? 0 : (b == 2) ? 20 : 16;   // bit position of the low nibble; the high one sits 8 bits above
        v[c >> 2] |= (int)((lo << plo) | (hi << (plo + 8)));
    }
    ((i32x4*)out)[oidx] = v;
}
static inline size_t p16t_row_bytes(int K) { return (size_t)((K / QB + 15) / 16) * 256; }   // bytes per row; a quad's four rows share 4 of them, interleaved
// 32-bit LDS byte address of a shared-memory pointer (operand of the asm ds_read forms below)
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
// this lane's window into a P16T / BF16T copy: row `row` of its quad, then chunk (64 * group + t) in 16-byte units
__device__ __forceinline__ const uint8_t* p16t_row_ptr(const uint8_t* base, int row, int ldb) {
    return base + (size_t)(row >> 2) * ((size_t)ldb * 4) + (size_t)(row & 3) * 256;
}

// acc = fma(p[lane N of the row], f, acc) / m = p[lane N of the row] * f: the DPP operand of a VOP2 instruction does the broadcast
template <int N> __device__ __forceinline__ void fmac_bcast(float& acc, float p, float f);
template <int N> __device__ __forceinline__ float mul_bcast(float p, float f);
#define JH_P16_BCAST(N)                                                                                                          \
    template <> __device__ __forceinline__ void fmac_bcast<N>(float& acc, float p, float f) {                                    \
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(p), "v"(f));   \
    }                                                                                                                            \
    template <> __device__ __forceinline__ float mul_bcast<N>(float p, float f) { /* hipcc folds the move into v_mul_f32_dpp */ \
        return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(p), 0x150 + N, 0xf, 0xf, true)) * f;               \
    }
JH_P16_BCAST(0) JH_P16_BCAST(1) JH_P16_BCAST(2) JH_P16_BCAST(3) JH_P16_BCAST(4) JH_P16_BCAST(5) JH_P16_BCAST(6) JH_P16_BCAST(7)
JH_P16_BCAST(8) JH_P16_BCAST(9) JH_P16_BCAST(10) JH_P16_BCAST(11) JH_P16_BCAST(12) JH_P16_BCAST(13) JH_P16_BCAST(14) JH_P16_BCAST(15)
#undef JH_P16_BCAST

// (float)(int8) of byte N of x in one instruction (SDWA source select + sign extension)
template <int N> __device__ __forceinline__ float cvt_sbyte(int x) {
    float f;
    if constexpr (N == 0) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(f) : "v"(x));
    else if constexpr (N == 1) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(f) : "v"(x));
    else if constexpr (N == 2) asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(f) : "v"(x));
    else asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(f) : "v"(x));
    return f;
}
// 8 nibbles of a dword -> int8 16*(nib-8): the nibbles at even positions (needs the shift) / at odd positions.
// T16 dwords (jh_t16.h): even = low nibbles (elements 4g..), odd = high nibbles (elements 16+4g..).
// P16T dwords: odd = [lo_b0, hi_b0, lo_b2, hi_b2] ("E"), even = [lo_b1, hi_b1, lo_b3, hi_b3] ("O") of blocks b0..b3.
__device__ __forceinline__ int nib_lo16(int x) { return ((x << 4) & (int)0xF0F0F0F0) ^ (int)0x80808080; }
__device__ __forceinline__ int nib_hi16(int x) { return (x & (int)0xF0F0F0F0) ^ (int)0x80808080; }

// ------------------------------------------------------------------------------------------------ activation row in LDS
struct ActP16 {
    i32x4* pt;     // [ceil(nblk/4)][16 lanes]: the pair words pw = (a[b*32+t] & 0xff) | (a[b*32+16+t] & 0xff) << 8 of blocks 4j..4j+3, one
                   // per dword: blocks 4j, 4j+1 in the LOW half, 4j+2, 4j+3 in the HIGH half (the other half zero) -- the v_dot4
                   // partners of the P16T weight bytes
    float* d16;    // [nblk] activation block scale / 16
    double* red;   // [32] reduction scratch
};
__device__ __forceinline__ ActP16 carve_p16(char* smem, int nblk) {
    ActP16 a;
    a.pt = (i32x4*)smem;
    a.d16 = (float*)(a.pt + ((nblk + 15) >> 4) * 64);   // whole groups: the reads of a short last group stay inside the table
    a.red = (double*)(a.d16 + ((nblk + 1) & ~1));
    return a;
}
static inline size_t lds_bytes_p16(int K) {
    const size_t nblk = (size_t)K / QB;
    return ((nblk + 15) / 16) * 1024 + ((nblk + 1) & ~(size_t)1) * 4 + 32 * 8;
}

// Quantize 8 consecutive values held by one lane (Panama quantizeQ8_512, PTO:1684-1723, exactly as quad_quantize_store) and file
// them as pair words: the lanes of a quad hold elements 0-7, 8-15 (codes of a[t]) and 16-23, 24-31 (codes of a[t+16]) of one
// block; sub and sub^2 exchange their packed codes, the lower lane pairs t = 8*(sub&1)+0..3, the upper one t = ...+4..7.
__device__ __forceinline__ void quad_quantize_store_p16(const float (&y)[8], int unit, const ActP16& a) {
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
    const int lo_dw = lower ? px : oy, hi_dw = lower ? ox : py;
    const int w0 = perm_b(hi_dw, lo_dw, 0x05010400);   // [lo0, hi0, lo1, hi1]
    const int w1 = perm_b(hi_dw, lo_dw, 0x07030602);   // [lo2, hi2, lo3, hi3]
    const int t0 = (sub & 1) * 8 + (lower ? 0 : 4);
    unsigned* dst = (unsigned*)a.pt + ((size_t)((blk >> 2) * 16 + t0)) * 4 + (blk & 3);   // dword (blk & 3) of entries t0..t0+3
    const int sh = (blk & 2) ? 16 : 0;                                                     // blocks 4j+2, 4j+3: the high half
    dst[0] = ((unsigned)w0 & 0xffffu) << sh;
    dst[4] = ((unsigned)w0 >> 16) << sh;
    dst[8] = ((unsigned)w1 & 0xffffu) << sh;
    dst[12] = ((unsigned)w1 >> 16) << sh;
    if (sub == 0) a.d16[blk] = d * 0.0625f;
}

// Activation prologue of the p16 GEMVs in two halves (the weight ring is requested between them: vmcnt retires oldest-first, the
// row must be asked for before the weights).  The whole row lives in registers: UM 8-element units per thread of the 512-thread
// workgroup (the host picks UM >= K / 4096) -- no second, serialised round trip and no load loop whose waits would drain the ring.
constexpr int P16_THREADS = 512;
template <int UM>
struct ActRegsP16 {
    float xv[UM][8];
    float wv[UM][8];
};
template <int PRO, int UM, int NT = P16_THREADS>
__device__ __forceinline__ void stage_issue_p16(const GemvParams& p, ActRegsP16<UM>& r, int tid = threadIdx.x) {
    const int units = p.K / 8;
#pragma unroll
    for (int u = 0; u < UM; u++) {
        int unit = tid + u * NT;
        unit = unit < units ? unit : units - 1;             // branch-free (clamped): a guarded load is waited for at the end of its block
        const float4 xa = *(const float4*)(p.x + unit * 8), xb = *(const float4*)(p.x + unit * 8 + 4);
        r.xv[u][0] = xa.x; r.xv[u][1] = xa.y; r.xv[u][2] = xa.z; r.xv[u][3] = xa.w;
        r.xv[u][4] = xb.x; r.xv[u][5] = xb.y; r.xv[u][6] = xb.z; r.xv[u][7] = xb.w;
        if (PRO == PRO_RMS_Q8 || PRO == PRO_RMS_F32) load8_norm(p.nw, unit * 8, r.wv[u]);
    }
}
// RMSNorm scale factor of the row held in r (RMSNorm.java:41-49: float squares, double sum, /E, +eps, 1/sqrt in double)
template <int UM, int NT = P16_THREADS>
__device__ __forceinline__ float rms_factor_p16(const GemvParams& p, const ActRegsP16<UM>& r, double* red) {
    const int units = p.K / 8;
    double ss = 0.0;
#pragma unroll
    for (int u = 0; u < UM; u++)
        if ((int)threadIdx.x + u * NT < units)
#pragma unroll
            for (int i = 0; i < 8; i++) ss += (double)(r.xv[u][i] * r.xv[u][i]);
    ss = block_sum_d(ss, red);
    ss /= (double)p.K;
    ss += (double)p.eps;
    ss = 1.0 / sqrt(ss);
    return (float)ss;
}
template <int PRO, int UM, int NT = P16_THREADS>
__device__ __forceinline__ void stage_finish_p16(const GemvParams& p, const ActP16& a, ActRegsP16<UM>& r, int tid = threadIdx.x) {
    static_assert(PRO == PRO_RMS_Q8 || PRO == PRO_QUANT_Q8, "p16 prologues: RMSNorm+Q8 or plain Q8");
    const int units = p.K / 8;
    float fs = 1.0f;
    if (PRO == PRO_RMS_Q8) fs = rms_factor_p16<UM, NT>(p, r, a.red);   // (block reduction: every thread of the workgroup, tid = threadIdx.x)
#pragma unroll
    for (int u = 0; u < UM; u++) {
        const int unit = tid + u * NT;
        if (unit < units) {
            float y[8];
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = (PRO == PRO_RMS_Q8) ? r.wv[u][i] * (fs * r.xv[u][i]) : r.xv[u][i];   // (0 + w) * ((float)ss * x)
            quad_quantize_store_p16(y, unit, a);
        }
    }
    lds_barrier();
}

// ------------------------------------------------------------------------------------------------ I8 x Q4 GEMV, reference order
// one group of 16 blocks: 4 dwords x (nibble pair t of 4 consecutive blocks)
// Four steps' pair sums at once: 4 x v_dot4_i32_i8 (the non-accumulating VOP3P form: hipcc only emits v_dot4c + a zeroing move) followed
// by their 4 converts.  Inside one asm block on purpose: a DOT result needs 3 wait states before a VALU read, which hipcc cannot
// know about asm -- here every convert sits 4 instructions behind its dot.  e / o: the weight bytes of blocks (b0, b2) / (b1, b3);
// pp: their pair words (b0, b1 in the low half, b2, b3 in the high half): the zero half of a pair word blanks the other block.
__device__ __forceinline__ void dot4x4_cvt(const i32x4 pp, int e, int o, float& f0, float& f1, float& f2, float& f3) {
    int i0, i1, i2, i3;
    asm("v_dot4_i32_i8 %4, %8, %12, 0\n\t"
        "v_dot4_i32_i8 %5, %9, %13, 0\n\t"
        "v_dot4_i32_i8 %6, %10, %12, 0\n\t"
        "v_dot4_i32_i8 %7, %11, %13, 0\n\t"
        "v_cvt_f32_i32_e32 %0, %4\n\t"
        "v_cvt_f32_i32_e32 %1, %5\n\t"
        "v_cvt_f32_i32_e32 %2, %6\n\t"
        "v_cvt_f32_i32_e32 %3, %7"
        : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3), "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3)
        : "v"(pp.x), "v"(pp.y), "v"(pp.z), "v"(pp.w), "v"(e), "v"(o));
}
struct Quad4 { float f0, f1, f2, f3; };
__device__ __forceinline__ Quad4 p16_quad_sums(int x, const i32x4 pp) {   // (float) pair sums of 4 consecutive blocks (x16)
    Quad4 q;
    dot4x4_cvt(pp, nib_hi16(x), nib_lo16(x), q.f0, q.f1, q.f2, q.f3);   // 3 bit ops, no byte shuffle: the P16T nibble order is the operand order
    return q;
}
template <int K4, bool FULL>
__device__ __forceinline__ void p16_quad_chain(const Quad4& q, float pscale, int nb, float& acc) {
    if (FULL || 4 * K4 + 0 < nb) fmac_bcast<4 * K4 + 0>(acc, pscale, q.f0);
    if (FULL || 4 * K4 + 1 < nb) fmac_bcast<4 * K4 + 1>(acc, pscale, q.f1);
    if (FULL || 4 * K4 + 2 < nb) fmac_bcast<4 * K4 + 2>(acc, pscale, q.f2);
    if (FULL || 4 * K4 + 3 < nb) fmac_bcast<4 * K4 + 3>(acc, pscale, q.f3);
}
// One group: the pair sums of quad k+1 are computed beside the 4 chained fmacs of quad k (independent work for the chain's
// latency); a scheduling fence per quad, because the chain is serial and hipcc otherwise hoists every independent unpack /
// convert of the group in front of it (hundreds of live registers, spills).  Blocks past a short last group hold clamped
// copies: their sums are computed and dropped.
template <bool FULL>
__device__ __forceinline__ void p16_group_i8(const i32x4& x, const i32x4* pt, float pscale, int nb, float& acc) {
    const Quad4 q0 = p16_quad_sums(x.x, pt[0]);
    __builtin_amdgcn_sched_barrier(0);
    const Quad4 q1 = p16_quad_sums(x.y, pt[16]);
    p16_quad_chain<0, FULL>(q0, pscale, nb, acc);
    __builtin_amdgcn_sched_barrier(0);
    const Quad4 q2 = p16_quad_sums(x.z, pt[32]);
    p16_quad_chain<1, FULL>(q1, pscale, nb, acc);
    __builtin_amdgcn_sched_barrier(0);
    const Quad4 q3 = p16_quad_sums(x.w, pt[48]);
    p16_quad_chain<2, FULL>(q2, pscale, nb, acc);
    __builtin_amdgcn_sched_barrier(0);
    p16_quad_chain<3, FULL>(q3, pscale, nb, acc);
}

// The pair sums of the NEXT quad and the 4 chained fmas of THIS quad in ONE asm statement, interleaved: dot4 x 4, then fma / convert
// alternating -- every convert sits >= 5 instructions behind its dot (a DOT result needs 3 wait states before a VALU read), every
// dependent fma has an independent convert in its latency, and nothing pads the chain: as 4 separate statements hipcc put an
// `s_nop 0` behind each fma (its fixed pad for an asm output), 12 of the ~96 instructions of a group in kernels whose working waves
// sit alone on their SIMD and issue one instruction per ~4.5 cycles (round 6: the few-row GEMVs are ISSUE-bound, not HBM-bound).
template <int K4> __device__ __forceinline__ void p16_sums_chain(const i32x4 pp, int e, int o, Quad4& nq, const Quad4& q, float pscale, float& acc);
#define JH_P16_SUMS_CHAIN(K4, B0, B1, B2, B3)                                                                                                   \
    template <> __device__ __forceinline__ void p16_sums_chain<K4>(const i32x4 pp, int e, int o, Quad4& nq, const Quad4& q, float pscale, float& acc) { \
        int i0, i1, i2, i3;                                                                                                                     \
        asm volatile("v_dot4_i32_i8 %5, %9, %13, 0\n\t"                                                                                          \
                     "v_dot4_i32_i8 %6, %10, %14, 0\n\t"                                                                                         \
                     "v_dot4_i32_i8 %7, %11, %13, 0\n\t"                                                                                         \
                     "v_dot4_i32_i8 %8, %12, %14, 0\n\t"                                                                                         \
                     "v_fmac_f32_dpp %4, %15, %16 row_newbcast:" #B0 " row_mask:0xf bank_mask:0xf\n\t"                                           \
                     "v_cvt_f32_i32_e32 %0, %5\n\t"                                                                                              \
                     "v_fmac_f32_dpp %4, %15, %17 row_newbcast:" #B1 " row_mask:0xf bank_mask:0xf\n\t"                                           \
                     "v_cvt_f32_i32_e32 %1, %6\n\t"                                                                                              \
                     "v_fmac_f32_dpp %4, %15, %18 row_newbcast:" #B2 " row_mask:0xf bank_mask:0xf\n\t"                                           \
                     "v_cvt_f32_i32_e32 %2, %7\n\t"                                                                                              \
                     "v_fmac_f32_dpp %4, %15, %19 row_newbcast:" #B3 " row_mask:0xf bank_mask:0xf\n\t"                                           \
                     "v_cvt_f32_i32_e32 %3, %8"                                                                                                  \
                     : "=&v"(nq.f0), "=&v"(nq.f1), "=&v"(nq.f2), "=&v"(nq.f3), "+v"(acc), "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3)              \
                     : "v"(pp.x), "v"(pp.y), "v"(pp.z), "v"(pp.w), "v"(e), "v"(o), "v"(pscale), "v"(q.f0), "v"(q.f1), "v"(q.f2), "v"(q.f3));     \
    }
JH_P16_SUMS_CHAIN(0, 0, 1, 2, 3) JH_P16_SUMS_CHAIN(1, 4, 5, 6, 7) JH_P16_SUMS_CHAIN(2, 8, 9, 10, 11)
#undef JH_P16_SUMS_CHAIN
// the group's last quad: 4 chained fmas (blocks 12..15), one statement
__device__ __forceinline__ void p16_chain_last(const Quad4& q, float pscale, float& acc) {
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %3 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %4 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %0, %1, %5 row_newbcast:15 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(pscale), "v"(q.f0), "v"(q.f1), "v"(q.f2), "v"(q.f3));
}

// The same group with its activation operands held in REGISTERS and refilled in place for the NEXT group: hipcc closes every LDS
// read it emits with s_waitcnt lgkmcnt(0) a few instructions later, i.e. a working wave (alone on its SIMD in the few-row GEMVs)
// sat out the LDS round trip four times per group (gemv_timeline: 550-720 cycles per group for ~95 instructions).  Here the reads
// are asm (no automatic wait), issued one group ahead in consumption order -- d16, pair words of quads 0..3 -- and every use waits
// with lgkmcnt(4): LDS operations retire in order, so at most the four younger requests are still in flight.
struct PairRegsP16 { i32x4 w0, w1, w2, w3; float d; };
// (the group's distance from the ring block's first group is a compile-time constant: it travels in the instruction's 16-bit offset
// field, so a block needs ONE address register per table instead of an add per request -- every instruction of these loops is an
// issue slot of a wave that sits alone on its SIMD)
template <int OFF = 0> __device__ __forceinline__ void p16_req_pairs(i32x4& w, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(w) : "v"(addr), "i"(OFF) : "memory");
}
template <int OFF = 0> __device__ __forceinline__ void p16_req_d(float& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "i"(OFF) : "memory");
}
template <class F, int... I> __device__ __forceinline__ void p16_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void p16_static_for(F&& f) { p16_static_for_impl(f, std::make_integer_sequence<int, N>{}); }
__device__ __forceinline__ void p16_tie4(i32x4& w) { asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w)::"memory"); }
__device__ __forceinline__ void p16_tie4(float& d) { asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(d)::"memory"); }
// LAST: the wave's final group requests nothing -- a request still in flight when its registers are dead would land in whatever
// hipcc has put there since (it cannot see the asm reads) -- and therefore waits for everything at once.
template <bool FULL, bool LAST, int NOFF = 0>
__device__ __forceinline__ void p16_group_i8_regs(const i32x4& x, PairRegsP16& r, unsigned pt_blk, float pscale, int nb, float& acc) {
    if constexpr (!LAST) p16_tie4(r.w0);                    // (LAST: the caller has waited for all four before it branched)
    const Quad4 q0 = p16_quad_sums(x.x, r.w0);
    if constexpr (!LAST) p16_req_pairs<NOFF>(r.w0, pt_blk);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FULL) {                                   // whole group (every real shape): sums of quad k+1 fused with the chain of quad k
        Quad4 q1, q2, q3;
        if constexpr (!LAST) p16_tie4(r.w1);
        p16_sums_chain<0>(r.w1, nib_hi16(x.y), nib_lo16(x.y), q1, q0, pscale, acc);
        if constexpr (!LAST) p16_req_pairs<NOFF + 256>(r.w1, pt_blk);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LAST) p16_tie4(r.w2);
        p16_sums_chain<1>(r.w2, nib_hi16(x.z), nib_lo16(x.z), q2, q1, pscale, acc);
        if constexpr (!LAST) p16_req_pairs<NOFF + 512>(r.w2, pt_blk);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LAST) p16_tie4(r.w3);
        p16_sums_chain<2>(r.w3, nib_hi16(x.w), nib_lo16(x.w), q3, q2, pscale, acc);
        if constexpr (!LAST) p16_req_pairs<NOFF + 768>(r.w3, pt_blk);
        __builtin_amdgcn_sched_barrier(0);
        p16_chain_last(q3, pscale, acc);
        return;
    }
    if constexpr (!LAST) p16_tie4(r.w1);
    const Quad4 q1 = p16_quad_sums(x.y, r.w1);
    if constexpr (!LAST) p16_req_pairs<NOFF + 256>(r.w1, pt_blk);
    p16_quad_chain<0, FULL>(q0, pscale, nb, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LAST) p16_tie4(r.w2);
    const Quad4 q2 = p16_quad_sums(x.z, r.w2);
    if constexpr (!LAST) p16_req_pairs<NOFF + 512>(r.w2, pt_blk);
    p16_quad_chain<1, FULL>(q1, pscale, nb, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!LAST) p16_tie4(r.w3);
    const Quad4 q3 = p16_quad_sums(x.w, r.w3);
    if constexpr (!LAST) p16_req_pairs<NOFF + 768>(r.w3, pt_blk);
    p16_quad_chain<2, FULL>(q2, pscale, nb, acc);
    __builtin_amdgcn_sched_barrier(0);
    p16_quad_chain<3, FULL>(q3, pscale, nb, acc);
}

// (da/16) * sb of the block this lane loaded, later read through the DPP operand of v_fmac_f32.  Written by asm with two idle
// states behind it: a DPP read of a VGPR needs 2 wait states after the VALU write, and hipcc's hazard recognizer does not see
// inside the asm blocks that do the reading.
__device__ __forceinline__ float p16_scale_product(float da16, float sb) {
    float r;
    asm volatile("v_mul_f32_e32 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(da16), "v"(sb));
    return r;
}

// Work of a wave: row quads [q0, q1) (4 consecutive weight rows, one per 16-lane row of the wave), NP passes each (gate then up
// for EPI_SILU_MUL), G groups of 16 blocks per pass -- a flat list of items streamed through a ring of D prefetched groups,
// consumed in blocks of D (the host picks D | G, so a pass ends exactly at a block end and the ring needs no bounds checks).
// `per` = row quads per task wave, `tw` = task waves per workgroup (the host sizes grid x tw x per so that every CU gets the same
// number of row quads).
template <int PRO, int EPI, int D, int UM, int NT = P16_THREADS>
__global__ __launch_bounds__(NT) void gemv_i8q4_p16_kernel(GemvParams p, int per, int tw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nblk = p.K / QB, G = (nblk + 15) >> 4;        // host: G % D == 0
    const ActP16 a = carve_p16(smem, nblk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane >> 4, t = lane & 15;
    constexpr int NP = (EPI == EPI_SILU_MUL) ? 2 : 1;
    const int nq = (p.nrows + 3) >> 2;
    // waves [0, tw) of a workgroup own row quads, the others only help with the activation prologue (a 16-lane row per chain
    // caps the useful waves at rows/4: the o- and down-projections have 4 per CU)
    int q0 = wave < tw ? (blockIdx.x * tw + wave) * per : nq;
    if (q0 > nq) q0 = nq;
    int q1 = q0 + per;
    if (q1 > nq) q1 = nq;
    const int items = (q1 - q0) * NP * G;                   // a multiple of D

    i32x4 wq[D];
    float sq[D];
    int lq = q0, lpass = 0, lg = 0;                         // load cursor (wave-uniform)
    // The weight / scale stream is addressed as  table base + group distance (lg * 1024 / lg * 64: scalar)  +  this lane's row offset
    // (ONE 32-bit VGPR per table, set per row): global_load ... saddr, two scalar 64-bit adds and two loads per group where per-lane
    // 64-bit addressing took 11 instructions (two v_lshl_or, two sign extensions, two 64-bit adds, a clamp ...).  No clamp: the
    // surplus lanes of a short last group read the next row's first scales (unused), behind the last row the 64 bytes of padding
    // every scale allocation carries (jh_model_set_weight).  Round 6 same-box A/B (Llama-3-8B, K = 256): 655 -> 664 tok/s; the
    // buffer-descriptor form of the same idea (4 instructions per group) cost the o-projection 0.2 us of latency: 661.
    // Row switch: a quad's four rows differ by lane-constant offsets (row r of the quad, chunk t), the quad itself by a UNIFORM one,
    // so wv / sv are written once and the switch is scalar arithmetic.  (New VGPRs behind 2 * D loads in flight are a trap: in one
    // build of round 6 hipcc gave a next-row offset the register of a pending ring load, the s_waitcnt vmcnt(0) in front of it
    // drained the ring instead of overlapping it, and the o-projection ran 12 % slower with byte-identical streaming code --
    // tools/isa_ring_drain.py.)  Whole quads only: the host refuses rows % 4 != 0 (every projection of the path is a multiple of 32).
    const int wv = r * 256 + t * 16, sv = (r * p.ldbf + t) * 4;   // this lane's byte offsets inside its quad's weight / scale rows
    size_t qw = 0, qs = 0;                                  // the quad's byte offsets (wave-uniform)
    auto set_row = [&]() __attribute__((always_inline)) {
        const int lqc = lq < nq ? lq : nq - 1;              // (the cursor runs one quad past the wave's share: stay inside the table)
        qw = (size_t)lqc * ((size_t)p.ldb * 4);
        qs = (size_t)lqc * ((size_t)p.ldbf * 16);
    };
    set_row();
    auto load_group = [&](i32x4& w, float& s) __attribute__((always_inline)) {
        const char* bw = (const char*)((NP == 2 && lpass) ? p.w2 : p.w) + qw + ((size_t)lg << 10);
        const char* bs = (const char*)((NP == 2 && lpass) ? p.ws2 : p.ws) + qs + ((size_t)lg << 6);
        w = __builtin_nontemporal_load((const i32x4*)(bw + (unsigned)wv));   // chunk t of group lg = byte t of its 16 blocks (rows padded to whole groups)
        s = __builtin_nontemporal_load((const float*)(bs + (unsigned)sv));
    };
    auto next_row = [&]() __attribute__((always_inline)) {
        lg = 0;
        if (++lpass == NP) { lpass = 0; ++lq; }
        set_row();
    };
    auto issue = [&](i32x4& w, float& s) __attribute__((always_inline)) {
        load_group(w, s);
        if (++lg == G) next_row();
    };
    // The same request where the load cursor cannot leave its row: slot d of a ring block asks for group (cg + d + D) mod G, and with
    // D | G that is the row's last group only for d == D - 1.  Slots 0 .. D-2 therefore stay free of branches (see the note on
    // pending LDS writes below).
    auto issue_in_row = [&](i32x4& w, float& s) __attribute__((always_inline)) {
        load_group(w, s);
        ++lg;
    };
    // Results are parked: the 16 lanes of a row all hold a finished row sum, lane t keeps the one of the wave's task n == t, and the
    // wave stores once per 16 tasks (normally once, after the loop): no store or residual load inside the streaming loop, whose
    // vmcnt waits then count ring loads only.
    auto resid_of_batch = [&](int qb) __attribute__((always_inline)) {   // residual of (task qb + t, row r), requested a batch ahead
        int row = 4 * (qb + t) + r;
        row = row < p.nrows ? row : p.nrows - 1;
        return p.resid[row];
    };
    ActRegsP16<UM> ar;
#define JH_GSTAMP(kk) do { if (p.dbg && lane == 0) p.dbg[((size_t)blockIdx.x * (NT / 64) + wave) * 8 + (kk)] = wall_clock64(); } while (0)
    JH_GSTAMP(0);
    if (items == 0) {
        // helper wave: its own copy of the prologue (same barriers).  The two paths must not join: hipcc computes ONE vmcnt per wait
        // and would size the working waves' waits for the path without ring loads, i.e. drain the ring inside the prologue.
        stage_issue_p16<PRO, UM, NT>(p, ar);
        __builtin_amdgcn_s_barrier();                       // (the working waves' "row requested" barrier, below)
        stage_finish_p16<PRO, UM, NT>(p, a, ar);
        JH_GSTAMP(2);
        if constexpr (EPI == EPI_TP) tp_signal(p);       // every wave of the workgroup passes one tp_signal (its barrier)
        return;
    }
    stage_issue_p16<PRO, UM, NT>(p, ar);                        // activation loads first: vmcnt retires oldest-first
    JH_GSTAMP(6);                                           // activation row requested
    float rv = 0.0f;
    if (EPI == EPI_RESID) rv = resid_of_batch(q0);
    // the ring is in flight across the prologue.  Its first D groups are groups 0 .. D-1 of the wave's first row (D | G), so only
    // the last of them can end the row: the others are requested without the row-switch branch and with constant group distances
    // (as D generic issue() calls this was ~20 scalar instructions + a branch per group in front of the first weight byte: 0.3 us
    // of a wave that issues one instruction per 4-8 cycles -- tools/gemv_timeline.py "entry -> ring requested")
    // ... and they wait until EVERY wave of the workgroup has requested its share of the activation row (a barrier, ~100 cycles).
    // Requested back to back without it (16 loads per wave in ~40 instructions), the rings of the first waves fill the CU's request
    // path in front of the later waves' activation loads; the quantized row -- what every wave waits for at the next barrier --
    // arrives later and the o-projection runs 4.8 -> 5.5 us (664 tok/s against 684).  The generic issue() form had paced itself by
    // accident (~20 scalar instructions per group).  Same-box A/B, 8B, K = 256 (tools/build_variant.py): generic 683.7-684.4;
    // compact, unpaced 664.7; s_sleep (x 64 cycles) in front / between the requests 0/2 682.3, 0/4 685.5-689.2, 0/8 670.7, 16/0 686.2,
    // 8/2 689.8-691.5; **barrier, then the ring at once 698.6-701.0** (barrier + s_sleep 1 / 2 between: 694.5-699.6 / 696.3-696.9).
    __builtin_amdgcn_s_barrier();
    p16_static_for<D - 1>([&](auto dc) __attribute__((always_inline)) {
        issue_in_row(wq[decltype(dc)::value], sq[decltype(dc)::value]);
        __builtin_amdgcn_sched_barrier(0);                  // request order = consumption order (the loop's vmcnt waits count on it)
    });
    issue(wq[D - 1], sq[D - 1]);
    __builtin_amdgcn_sched_barrier(0);
    JH_GSTAMP(1);                                           // activation row + ring requested
    stage_finish_p16<PRO, UM, NT>(p, a, ar);
    JH_GSTAMP(2);                                           // activation operands in LDS (workgroup barrier passed)

    float acc = 0.0f, gres = 0.0f, gsel = 0.0f, usel = 0.0f;
    int cq = q0, cpass = 0, cg = 0;                         // compute cursor
    // one group in two halves with the slot's refill between them: prep() takes the scale product out of the loaded registers
    // activation operands of the group being consumed (registers), requested one group ahead: see p16_group_i8_regs
    PairRegsP16 pr;
    const unsigned pt_base = lds_addr(a.pt) + (unsigned)t * 16u, d_base = lds_addr(a.d16) + (unsigned)t * 4u;
    // LDS addresses of this lane's pair words / block scale of the ring block's FIRST group (cg); the other groups of the block sit
    // d * 1024 / d * 64 bytes further = the offset field of the request.  (A short last group reads up to 15 floats past the scale
    // table: inside the reduction scratch behind it, never used -- the chain's guarded links stop at nb.)
    unsigned blk_pt = pt_base, blk_d = d_base;
    // A register with a pending LDS write must not cross a basic-block boundary: hipcc does not know about the write, and at a
    // control-flow merge it is free to COPY the register (it did: `v_mov_b64 v[38:39], v[54:55]` in front of the wait, i.e. a copy of
    // words that need not have landed -- correct only by timing).  So the pipeline lives inside ONE straight-line region, a ring block
    // of D groups: the block's first group is requested at its top, slots 0 .. D-2 request the next group, the last slot requests
    // nothing and waits for everything before its first use; `issue`'s row switch, a short last group, pass_end() and the loop's
    // back-edge all sit behind that slot.  tools/isa_pending_lds.py checks the compiled ISA for exactly this (tests/test_tools.py).
    auto request_first = [&]() __attribute__((always_inline)) {
        p16_req_d<0>(pr.d, blk_d);
        p16_req_pairs<0>(pr.w0, blk_pt); p16_req_pairs<256>(pr.w1, blk_pt); p16_req_pairs<512>(pr.w2, blk_pt); p16_req_pairs<768>(pr.w3, blk_pt);
    };
    // slot d < D-1 of a ring block: scale product out of the loaded registers, the group's chains with the NEXT group's operands
    // requested as its registers free up; REFILL: the slot's registers ask for the group D ahead (main loop) or for nothing (last block)
    auto slot = [&](auto dc, auto refill) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        p16_tie4(pr.d);
        const float ps = p16_scale_product(pr.d, sq[d]);    // lane t carries the scale product of block 16 * (cg + d) + t
        p16_req_d<(d + 1) * 64>(pr.d, blk_d);
        __builtin_amdgcn_sched_barrier(0);                  // keep the slots in program order (hipcc otherwise hoists all D unpacks
        p16_group_i8_regs<true, false, (d + 1) * 1024>(wq[d], pr, blk_pt, ps, 16, acc);   // to the top of the block, which then waits for every load in flight)
        __builtin_amdgcn_sched_barrier(0);
        // The slot is refilled AFTER its group is consumed, into the same registers.  Requested before the compute (round 4), the
        // new words needed fresh registers, and hipcc closed the loop with a copy of the whole ring behind s_waitcnt vmcnt(13..1):
        // every block of D groups ended by waiting for the load requested one group earlier -- a full memory round trip per block
        // (tools/gemv_timeline.py: the main loop of the down-projection was 6.8 of its 11.8 us, and ring depth made no difference).
        if constexpr (decltype(refill)::value) {
            issue_in_row(wq[d], sq[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the block's last slot: nothing left to request from LDS; a short last group (K / 32 not a multiple of 16) takes the guarded chain
    auto last_slot = [&](auto refill) __attribute__((always_inline)) {
        p16_tie4(pr.d);
        const float ps = p16_scale_product(pr.d, sq[D - 1]);
        __builtin_amdgcn_sched_barrier(0);
        const int nb = nblk - 16 * (cg + D - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pr.w0), "+v"(pr.w1), "+v"(pr.w2), "+v"(pr.w3)::"memory");   // before the branch below
        if (nb < 16) p16_group_i8_regs<false, true>(wq[D - 1], pr, blk_pt, ps, nb, acc);
        else p16_group_i8_regs<true, true>(wq[D - 1], pr, blk_pt, ps, 16, acc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (decltype(refill)::value) {
            issue(wq[D - 1], sq[D - 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto pass_end = [&]() __attribute__((always_inline)) {
        const float res = row16_tree_sum(acc);
        acc = 0.0f;
        if (EPI == EPI_SILU_MUL && cpass == 0) {
            gres = res;                                     // gate pass done, the up pass of the same rows follows
            cpass = 1;
            return;
        }
        cpass = 0;
        const int n = (cq - q0) & 15;
        if (t == n) { gsel = gres; usel = res; }
        if (n == 15 || cq + 1 == q1) {
            const int row = 4 * (cq - n + t) + r;
            if (t <= n && row < p.nrows) {
                float v = usel;
                if (EPI == EPI_SILU_MUL) v = silu_ref(gsel) * usel;   // MLPBlock.java:132-142; SiLU in double (~500 SIMD cycles) once per 16 tasks
                if (EPI == EPI_RESID) v = v + rv;                      // accumulate(...) TransformerBlock.java:185,203
                if (EPI == EPI_TP) tp_store(p, row, v);                // (a tensor-parallel shard stores into every shard's slot)
                else p.out[row] = v;
            }
            if (EPI == EPI_RESID && cq + 1 < q1) rv = resid_of_batch(cq + 1);
        }
        ++cq;
    };
    for (int it = 0; it + D < items; it += D) {
        request_first();
        p16_static_for<D - 1>([&](auto dc) __attribute__((always_inline)) { slot(dc, std::true_type{}); });
        last_slot(std::true_type{});
        cg += D; blk_pt += (unsigned)D * 1024u; blk_d += (unsigned)D * 64u;
        if (cg == G) { cg = 0; blk_pt = pt_base; blk_d = d_base; pass_end(); }
    }
    JH_GSTAMP(3);                                           // every group but the last ring block consumed
    request_first();
    p16_static_for<D - 1>([&](auto dc) __attribute__((always_inline)) { slot(dc, std::false_type{}); });   // last block: nothing left to request
    last_slot(std::false_type{});
    JH_GSTAMP(4);
    pass_end();
    JH_GSTAMP(5);                                           // row sums stored
#undef JH_GSTAMP
    if constexpr (EPI == EPI_TP) tp_signal(p);
}

// ------------------------------------------------------------------------------------------------ F32 x Q4 GEMV (LM head), reference order
// acc_t = fma(a[t], (float)(lo_t-8)*s, acc_t); acc_t = fma(a[t+16], (float)(hi_t-8)*s, acc_t) per block (PTO:336-374) with the final
// RMSNorm prologue and the per-workgroup argmax partials of gemv_f32q4_kernel (strict >, lowest index first: AbstractModel.java:455-469).
// (float)(nib-8)*s == (float)(16*(nib-8)) * (s/16) exactly, and the left factor is one SDWA convert of the unpacked int8.
struct ActF32P16 {
    float2* af;    // [nblk][16]: (y[b*32+t], y[b*32+16+t])
    double* red;   // [32]
    float* bestv;  // [16]
    int* besti;    // [16]
};
__device__ __forceinline__ ActF32P16 carve_f32_p16(char* smem, int nblk) {
    ActF32P16 a;
    a.af = (float2*)smem;
    a.red = (double*)(a.af + (size_t)((nblk + 15) & ~15) * 16);
    a.bestv = (float*)(a.red + 32);
    a.besti = (int*)(a.bestv + 16);
    return a;
}
static inline size_t lds_bytes_f32_p16(int K) { return (size_t)(((K / QB) + 15) & ~15) * 128 + 32 * 8 + 16 * 4 + 16 * 4; }

// acc = fma(a, b, acc), pinned where it is written: the chain's only reader is the end of the row, and hipcc otherwise sinks every
// fma of a ring block down there, keeping all their operands alive (256 VGPRs + spills)
__device__ __forceinline__ void fmac_pinned(float& acc, float a, float b) {
    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}
struct QuadW { float wl0, wl1, wl2, wl3, wh0, wh1, wh2, wh3; float2 a0, a1, a2, a3; };
// dequantized weights (float)(nib-8)*s of 4 consecutive blocks for this lane's element pair, and the matching activation pairs
template <int K4>
__device__ __forceinline__ QuadW p16_quad_deq(int x, const float2* af, float s16) {
    const int e = nib_hi16(x), o = nib_lo16(x);   // P16T: e = [lo_b0, hi_b0, lo_b2, hi_b2], o = [lo_b1, hi_b1, lo_b3, hi_b3] (x16)
    QuadW q;
    q.a0 = af[(4 * K4 + 0) * 16]; q.a1 = af[(4 * K4 + 1) * 16]; q.a2 = af[(4 * K4 + 2) * 16]; q.a3 = af[(4 * K4 + 3) * 16];
    const float l0 = cvt_sbyte<0>(e), h0 = cvt_sbyte<1>(e), l2 = cvt_sbyte<2>(e), h2 = cvt_sbyte<3>(e);
    const float l1 = cvt_sbyte<0>(o), h1 = cvt_sbyte<1>(o), l3 = cvt_sbyte<2>(o), h3 = cvt_sbyte<3>(o);
    q.wl0 = mul_bcast<4 * K4 + 0>(s16, l0); q.wh0 = mul_bcast<4 * K4 + 0>(s16, h0);
    q.wl1 = mul_bcast<4 * K4 + 1>(s16, l1); q.wh1 = mul_bcast<4 * K4 + 1>(s16, h1);
    q.wl2 = mul_bcast<4 * K4 + 2>(s16, l2); q.wh2 = mul_bcast<4 * K4 + 2>(s16, h2);
    q.wl3 = mul_bcast<4 * K4 + 3>(s16, l3); q.wh3 = mul_bcast<4 * K4 + 3>(s16, h3);
    return q;
}
template <int K4, bool FULL>
__device__ __forceinline__ void p16_quad_chain_f32(const QuadW& q, int nb, float& acc) {
    if (FULL || 4 * K4 + 0 < nb) { fmac_pinned(acc, q.a0.x, q.wl0); fmac_pinned(acc, q.a0.y, q.wh0); }
    if (FULL || 4 * K4 + 1 < nb) { fmac_pinned(acc, q.a1.x, q.wl1); fmac_pinned(acc, q.a1.y, q.wh1); }
    if (FULL || 4 * K4 + 2 < nb) { fmac_pinned(acc, q.a2.x, q.wl2); fmac_pinned(acc, q.a2.y, q.wh2); }
    if (FULL || 4 * K4 + 3 < nb) { fmac_pinned(acc, q.a3.x, q.wl3); fmac_pinned(acc, q.a3.y, q.wh3); }
}
// the dequantization of quad k+1 runs beside the 8 chained fmas of quad k (see p16_group_i8)
template <bool FULL>
__device__ __forceinline__ void p16_group_f32(const i32x4& x, const float2* af, float s16, int nb, float& acc) {
    const QuadW q0 = p16_quad_deq<0>(x.x, af, s16);
    __builtin_amdgcn_sched_barrier(0);
    const QuadW q1 = p16_quad_deq<1>(x.y, af, s16);
    p16_quad_chain_f32<0, FULL>(q0, nb, acc);
    __builtin_amdgcn_sched_barrier(0);
    const QuadW q2 = p16_quad_deq<2>(x.z, af, s16);
    p16_quad_chain_f32<1, FULL>(q1, nb, acc);
    __builtin_amdgcn_sched_barrier(0);
    const QuadW q3 = p16_quad_deq<3>(x.w, af, s16);
    p16_quad_chain_f32<2, FULL>(q2, nb, acc);
    __builtin_amdgcn_sched_barrier(0);
    p16_quad_chain_f32<3, FULL>(q3, nb, acc);
}

template <int PRO, int D, int UM>
__global__ __launch_bounds__(P16_THREADS) void gemv_f32q4_p16_kernel(GemvParams p, int per, int tw) {
    static_assert(PRO == PRO_RMS_F32 || PRO == PRO_F32, "LM head prologues");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, nblk = K / QB, G = (nblk + 15) >> 4;   // host: G % D == 0, K <= UM * 4096
    const ActF32P16 a = carve_f32_p16(smem, nblk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int r = lane >> 4, t = lane & 15;
    const int nq = (p.nrows + 3) >> 2;
    int q0 = wave < tw ? (blockIdx.x * tw + wave) * per : nq;
    if (q0 > nq) q0 = nq;
    int q1 = q0 + per;
    if (q1 > nq) q1 = nq;
    const int items = (q1 - q0) * G;

    ActRegsP16<UM> ar;
    auto fill_row = [&]() __attribute__((always_inline)) {  // second half of the prologue: final RMSNorm, the row as (y[t], y[t+16]) pairs
        float fs = 1.0f;
        if (PRO == PRO_RMS_F32) fs = rms_factor_p16<UM>(p, ar, a.red);
        const int units = K / 8;
#pragma unroll
        for (int u = 0; u < UM; u++) {
            const int unit = threadIdx.x + u * P16_THREADS;
            if (unit < units) {
                // elements e0..e0+7 of block b: half = (e0 & 16) != 0, lanes t = (e0 & 15) .. +7
                const int e0 = unit * 8, b = e0 >> 5, half = (e0 >> 4) & 1, t0 = e0 & 15;
                float* dst = (float*)(a.af + (size_t)b * 16 + t0) + half;
#pragma unroll
                for (int i = 0; i < 8; i++) dst[2 * i] = (PRO == PRO_RMS_F32) ? ar.wv[u][i] * (fs * ar.xv[u][i]) : ar.xv[u][i];
            }
        }
        lds_barrier();
    };
    i32x4 wq[D];
    float sq[D];
    int lq = q0, lg = 0;
    const uint8_t* wrow;
    const float* srow;
    auto set_row = [&]() __attribute__((always_inline)) {
        int row = 4 * lq + r;
        row = row < p.nrows ? row : p.nrows - 1;
        wrow = p16t_row_ptr(p.w, row, p.ldb);
        srow = p.ws + (size_t)row * p.ldbf;
    };
    set_row();
    auto issue = [&](i32x4& w, float& s) __attribute__((always_inline)) {
        int b = 16 * lg + t;
        w = __builtin_nontemporal_load((const i32x4*)wrow + (64 * lg + t));   // P16T copy
        b = b < nblk ? b : nblk - 1;
        s = __builtin_nontemporal_load(srow + b);
        if (++lg == G) { lg = 0; ++lq; set_row(); }
    };
    float bestv = -INFINITY;
    int besti = 0x7fffffff;
    if (items == 0) {
        // helper wave: own copy of the prologue, never joins the streaming path before the argmax merge (see gemv_i8q4_p16_kernel)
        stage_issue_p16<PRO, UM>(p, ar);
        fill_row();
    } else {
    stage_issue_p16<PRO, UM>(p, ar);                        // activation row (+ norm weights) requested before the weight ring
#pragma unroll
    for (int d = 0; d < D; d++) {
        issue(wq[d], sq[d]);
        __builtin_amdgcn_sched_barrier(0);                  // request order = consumption order
    }
    fill_row();

    float acc = 0.0f;
    int cq = q0, cg = 0;
    auto prep = [&](i32x4& x, float s) __attribute__((always_inline)) {
        (void)x;
        return p16_scale_product(0.0625f, s);               // s/16 (exact), pinned here so that the slot's registers are free for the refill
    };
    auto compute = [&](const i32x4& x, float s16, int g, bool can_be_short) __attribute__((always_inline)) {
        const float2* af = a.af + (size_t)(16 * g) * 16 + t;
        const int nb = nblk - 16 * g;
        if (can_be_short && nb < 16) p16_group_f32<false>(x, af, s16, nb, acc);
        else p16_group_f32<true>(x, af, s16, 16, acc);
    };
    float park = 0.0f;
    auto row_end = [&]() __attribute__((always_inline)) {
        const float res = row16_tree_sum(acc);
        acc = 0.0f;
        const int row = 4 * cq + r;
        if (row < p.nrows && res > bestv) { bestv = res; besti = row; }   // rows ascend within a lane: strict > keeps the first
        const int n = (cq - q0) & 15;                       // logits parked in lane t == n, stored once per 16 tasks (no store in the loop)
        if (t == n) park = res;
        if (n == 15 || cq + 1 == q1) {
            const int orow = 4 * (cq - n + t) + r;
            if (t <= n && orow < p.nrows) p.out[orow] = park;
        }
        ++cq;
    };
    {
        for (int it = 0; it + D < items; it += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                const float s16 = prep(wq[d], sq[d]);       // refill after the group, in place: see gemv_i8q4_p16_kernel
                __builtin_amdgcn_sched_barrier(0);
                compute(wq[d], s16, cg + d, d == D - 1);
                __builtin_amdgcn_sched_barrier(0);
                issue(wq[d], sq[d]);
                __builtin_amdgcn_sched_barrier(0);
            }
            cg += D;
            if (cg == G) { cg = 0; row_end(); }
        }
#pragma unroll
        for (int d = 0; d < D; d++) {
            i32x4 x = wq[d];
            const float s16 = prep(x, sq[d]);
            compute(x, s16, cg + d, d == D - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        row_end();
    }
    }   // streaming path
    if (p.amax_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {   // per-lane bests -> wave best (value desc, index asc)
            const float ov = __shfl_xor(bestv, o);
            const int oi = __shfl_xor(besti, o);
            if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
        }
        if (lane == 0) { a.bestv[wave] = bestv; a.besti[wave] = besti; }
        lds_barrier();
        if (threadIdx.x == 0) {
            for (int w = 1; w < nwaves; w++)
                if (a.bestv[w] > bestv || (a.bestv[w] == bestv && a.besti[w] < besti)) { bestv = a.bestv[w]; besti = a.besti[w]; }
            p.amax_part[blockIdx.x] = bestv;
            p.amax_idx[blockIdx.x] = besti;
        }
    }
}

// 16 links of a sequential float chain in 16 instructions: the link's uniform operand is lane k of a register every 16-lane row
// holds a copy of (lane t = element 16 j + t), taken through the DPP operand of the add / fma itself (row_newbcast:k) -- no
// v_readlane, no SGPR, no wait state between the lift and the use.  Round 5 lifted the uniform operand into SGPRs (readlane + the two
// wait states a VALU needs behind it: 2-3 issue slots per link, and a wave alone on its SIMD issues one instruction per ~4.5 cycles).
__device__ __forceinline__ void add16_bcast(float& sum, float x) {      // sum = fl(sum + x[k]), k = 0..15 in order
    asm volatile(
        "s_nop 1\n\t"   // a DPP read needs 2 wait states behind a VALU write of its source (x may have just been moved); hipcc cannot see inside
        "v_add_f32_dpp %0, %1, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(sum) : "v"(x));
}
// acc = fma(w[k], v_k, acc), k = 0..15 in order: w = one register (lane t of every row holds weight 16 j + t), v = this lane's 16 values.
// ONE asm statement: between separate statements hipcc pads the producer -> consumer edge of `acc` it cannot see into (round 6 timeline:
// 23 cycles per link as 16 statements, against 7-8 for the 16 adds of add16_bcast in one).
__device__ __forceinline__ void fma16_bcast(float& acc, float w, const f32x4 (&v)[4]) {
    asm volatile(
        "s_nop 1\n\t"   // (DPP source w: 2 wait states behind a VALU write)
        "v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(w), "v"(v[0].x), "v"(v[0].y), "v"(v[0].z), "v"(v[0].w), "v"(v[1].x), "v"(v[1].y), "v"(v[1].z), "v"(v[1].w), "v"(v[2].x), "v"(v[2].y),
          "v"(v[2].z), "v"(v[2].w), "v"(v[3].x), "v"(v[3].y), "v"(v[3].z), "v"(v[3].w));
}
// the same for a chain that ends inside the group: links k < cnt only (cnt wave-uniform)
__device__ __forceinline__ void fma16_bcast_tail(float& acc, float w, const f32x4 (&v)[4], int cnt) {
#define JH_FT(K, E) if (K < cnt) fmac_bcast<K>(acc, w, E);
    JH_FT(0, v[0].x) JH_FT(1, v[0].y) JH_FT(2, v[0].z) JH_FT(3, v[0].w) JH_FT(4, v[1].x) JH_FT(5, v[1].y) JH_FT(6, v[1].z) JH_FT(7, v[1].w)
    JH_FT(8, v[2].x) JH_FT(9, v[2].y) JH_FT(10, v[2].z) JH_FT(11, v[2].w) JH_FT(12, v[3].x) JH_FT(13, v[3].y) JH_FT(14, v[3].z) JH_FT(15, v[3].w)
#undef JH_FT
}

// softMax's running sum (VectorMath.java:80-85) by ONE wave: w[0, n16) = the exponentials, zero-padded to a whole group of 16 (and
// readable 64 floats further); register j holds w[16 j + t] in lane t of every 16-lane row, link k of a group is one v_add_f32
// whose DPP operand broadcasts lane k.  Every lane returns the same sum.
__device__ __forceinline__ float p16_seq_sum_wave(const float* w, int n16, int lane) {
    float sum = 0.0f;
    const int ng = n16 >> 4, t16 = lane & 15;
    float x0 = w[t16], x1 = w[16 + t16], x2 = w[32 + t16], x3 = w[48 + t16];
    int g = 0;
    for (; g + 4 <= ng; g += 4) {                           // 64 links per trip, the next 64 values in flight
        const float* nx = w + (g + 4) * 16 + t16;
        const float y0 = nx[0], y1 = nx[16], y2 = nx[32], y3 = nx[48];
        add16_bcast(sum, x0); add16_bcast(sum, x1); add16_bcast(sum, x2); add16_bcast(sum, x3);
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
    }
    if (g < ng) add16_bcast(sum, x0);
    if (g + 1 < ng) add16_bcast(sum, x1);
    if (g + 2 < ng) add16_bcast(sum, x2);
    return sum;
}
// value[d] += fma chain over the cnt positions of one V tile (saxpy per position, PTO:2648-2698), ONE wave: lanes 0..31 own one
// column each (lanes 32..63 run the same chain on column lane - 32).  vrow = this lane's column in the TRANSPOSED tile (positions
// contiguous: one ds_read_b128 = 4 links), wt = the tile's normalised weights (+ lane & 15): register j holds weight 16 j + t in lane
// t of every row and reaches the chain through the DPP operand of the fma (fma16_bcast) -- one instruction per link, plain C
// around it (every wait is hipcc's own), the column's values one 16-link piece ahead.  Reads past cnt stay inside the tile / the
// padded weight row and are never used, except by guarded links.
__device__ __forceinline__ void p16_value_chain_tile(float& acc, const float* vrow, const float* wt, int cnt, int TP) {
    // Pieces of 16 links through THREE register sets: piece k+2 is requested while piece k runs.  One piece ahead was not enough:
    // 16 dependent fmas take ~77 cycles (tools/chain_lab.hip: 4.8 cycles per v_fmac_f32_dpp link for a wave alone on its SIMD), the
    // four ds_read_b128 behind them ~130 -- the chain waited for LDS at every piece (round 6 timeline: 11 cycles per link).
    (void)TP;
    const int npf = cnt >> 4, rem = cnt & 15;               // full pieces, links of the partial one
    f32x4 b0[4], b1[4], b2[4];
    float w0, w1, w2;
    auto ldp = [&](f32x4 (&v)[4], float& w, int piece) __attribute__((always_inline)) {   // (past the tile's used part: inside the padded
        const f32x4* q = (const f32x4*)(vrow + 16 * piece);                               //  rows / weight row, never used unguarded)
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
        w = wt[16 * piece];
    };
    ldp(b0, w0, 0);
    ldp(b1, w1, 1);
    int pc = 0;
    for (; pc + 3 <= npf; pc += 3) {
        ldp(b2, w2, pc + 2);
        __builtin_amdgcn_sched_barrier(0);
        fma16_bcast(acc, w0, b0);
        __builtin_amdgcn_sched_barrier(0);
        ldp(b0, w0, pc + 3);
        __builtin_amdgcn_sched_barrier(0);
        fma16_bcast(acc, w1, b1);
        __builtin_amdgcn_sched_barrier(0);
        ldp(b1, w1, pc + 4);
        __builtin_amdgcn_sched_barrier(0);
        fma16_bcast(acc, w2, b2);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int left = npf - pc;                              // 0, 1 or 2 full pieces, then the partial one (the values past n in the tile
    if (left == 0) {                                        // may be anything, 0 * NaN: guarded links)
        if (rem) fma16_bcast_tail(acc, w0, b0, rem);
    } else if (left == 1) {
        fma16_bcast(acc, w0, b0);
        if (rem) fma16_bcast_tail(acc, w1, b1, rem);
    } else {
        ldp(b2, w2, pc + 2);
        fma16_bcast(acc, w0, b0);
        fma16_bcast(acc, w1, b1);
        if (rem) fma16_bcast_tail(acc, w2, b2, rem);
    }
}

// ------------------------------------------------------------------------------------------------ decode attention, reference order
// CausalSelfAttention.java:199-357 in two launches, because the reference's order makes two parts of it sequential over the whole
// context -- the float sum of the exponentials and the fma chain of every output element over the positions -- while its traffic
// (K and V of the context) must be spread over many CUs to arrive in time (one CU ingests ~25 GB/s):
//   attn_p16_scores_kernel  grid (position slices, kv heads): KV row write + RoPE (q of the group's heads, k of the new row), then
//        scores[t] = GemmerF32 16-lane dot (fma over 16-element steps, halving tree) * attentionScale for its slice, for the GROUP
//        query heads sharing the kv head (K is read once per group) -> scaled scores in global memory;
//   attn_p16_av_kernel      grid (32-column slices of the head, query heads): softMax over the whole score row (max, (float)exp in
//        double, FLOAT sum in index order by one lane, division), then value[d] = one fma chain per element over positions 0..pos
//        (saxpy per position, PTO:2648-2698) for its 32 columns; V tiles go through LDS with 16-byte loads.
constexpr int P16_ATT_THREADS = 256;
// The one-launch kernel (attn_p16_fused_kernel) runs EIGHT waves: its phases in front of the two sequential chains -- the score
// passes (a wave issues one instruction per ~5-8 cycles; 128 positions per pass instead of 64), the exponentials (7 waves instead of
// 3 beside the chain owner), the V tile's transposition -- are instruction-issue bound and split over the waves, and only half of
// the chip's CUs host an attention workgroup at all.  Same-box A/B (8B, K = 256): 256 threads 697.4-698.0 tok/s, probe 7.24 us;
// 512 threads 707.5-707.8, 6.86 us.
#ifndef P16_FUSED_THREADS_N
#define P16_FUSED_THREADS_N 512
#endif
constexpr int P16_FUSED_THREADS = P16_FUSED_THREADS_N;
template <int HS, int GROUP>
__global__ __launch_bounds__(P16_ATT_THREADS) void attn_p16_scores_kernel(AttnParams p, float* scores, int sc_stride) {
    __shared__ float qs[GROUP * HS];
    __shared__ float knew[HS];
    constexpr int NT = P16_ATT_THREADS, half = HS / 2, NC = HS / 16, RP = NT / 16;   // RP positions per pass
    const int pos = p.st->pos, n = pos + 1;
    const float* qkv_row = p.qkv;
    const int kvh = blockIdx.y, split = blockIdx.x, S = gridDim.x;
    const int chunk = (((n + S - 1) / S) + RP - 1) / RP * RP;   // whole passes per slice
    const int t0 = split * chunk;
    if (t0 >= n) return;
    const int t1 = t0 + chunk < n ? t0 + chunk : n;
    const int KV = p.n_kv_heads * HS, A = p.n_heads * HS;
    const int tid = threadIdx.x, l = tid & 15, prow = tid >> 4;
    // ---- the K rows of the first two passes are requested before q / rope: their addresses depend on the position only, so the
    // kernel pays one memory round trip for both (a 16-lane row per position: lane l reads k[l], k[16+l], ... -- GemmerF32's lanes)
    constexpr int PB = 2;
    float kv[PB][NC];
    auto load_k = [&](int tb) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PB; u++) {
            int tt = tb + u * RP;
            tt = tt < n ? tt : n - 1;
            const float* krow = kv_row(p, 0, tt, KV) + (size_t)kvh * HS + l;
#pragma unroll
            for (int c = 0; c < NC; c++) kv[u][c] = krow[16 * c];
        }
    };
    load_k(t0 + prow);
    // ---- RoPE of the group's q heads and of the new k row (table row pos + 2*kvHead: CausalSelfAttention.java:247-286); every
    // slice rotates them locally, bit-identically; the slice that owns `pos` writes the KV page rows
    const float* rf = p.rope + ((size_t)pos * half + (size_t)(kvh + p.kv_head_offset) * HS) * 2;
    const bool owner = pos >= t0 && pos < t1;
    for (int i = tid; i < (GROUP + 1) * half; i += NT) {
        const int gi = i / half, d = i - gi * half;
        const float c = rf[2 * d], s = rf[2 * d + 1];
        if (gi < GROUP) {
            const float* qh = qkv_row + (size_t)(kvh * GROUP + gi) * HS;
            const float q0 = qh[d], q1 = qh[d + half];
            const float r0 = q0 * c - q1 * s, r1 = q0 * s + q1 * c;   // contraction off: mul, mul, sub / add as in Java
            qs[gi * HS + d] = r0; qs[gi * HS + d + half] = r1;
            if (p.tap_q && split == 0) {
                p.tap_q[(size_t)(kvh * GROUP + gi) * HS + d] = r0;
                p.tap_q[(size_t)(kvh * GROUP + gi) * HS + d + half] = r1;
            }
        } else {
            const float* kh = qkv_row + A + (size_t)kvh * HS;
            const float k0 = kh[d], k1 = kh[d + half];
            const float r0 = k0 * c - k1 * s, r1 = k0 * s + k1 * c;
            knew[d] = r0; knew[d + half] = r1;
            if (owner) {   // K is stored post-RoPE (:273-286 rotates the page row in place)
                float* kdst = (float*)kv_row(p, 0, pos, KV) + (size_t)kvh * HS;
                kdst[d] = r0; kdst[d + half] = r1;
            }
        }
    }
    if (owner)
        for (int d = tid; d < HS; d += NT) ((float*)kv_row(p, 1, pos, KV) + (size_t)kvh * HS)[d] = qkv_row[A + KV + (size_t)kvh * HS + d];
    __syncthreads();
    float q[GROUP][NC];
#pragma unroll
    for (int gi = 0; gi < GROUP; gi++)
#pragma unroll
        for (int c = 0; c < NC; c++) q[gi][c] = qs[gi * HS + 16 * c + l];
    // ---- scores
    for (int tb = t0 + prow; tb < t1; tb += RP * PB) {
        if (tb != t0 + prow) load_k(tb);                    // long slices: further passes, two at a time
#pragma unroll
        for (int u = 0; u < PB; u++) {
            const int tt = tb + u * RP;
            if (tt >= t1) break;   // uniform per 16-lane row; the DPP tree below stays inside the row
            if (tt == pos) {
#pragma unroll
                for (int c = 0; c < NC; c++) kv[u][c] = knew[16 * c + l];   // the page row is being written by this workgroup just now
            }
#pragma unroll
            for (int gi = 0; gi < GROUP; gi++) {
                float acc = 0.0f;
#pragma unroll
                for (int c = 0; c < NC; c++) acc = fmaf(q[gi][c], kv[u][c], acc);   // GemmerF32 (PTO:1086-1102): lane l, steps of 16
                acc = row16_tree_sum(acc);
                if (l == 0) scores[(size_t)(kvh * GROUP + gi) * sc_stride + tt] = acc * p.scale;   // ops.scale after the dot (:332)
            }
        }
    }
}

template <int HS, int RU>
__device__ __forceinline__ void p16_av_load_tile(const AttnParams& p, f32x4 (&vreg)[RU], int tile, int n, int KV, int kvh, int d0, int vr, int vc) {
    // 32-row groups that lie wholly behind the context are neither requested nor filed (wave-uniform: a session sized for 512
    // positions otherwise moves 16 groups for a 130-position context); inside the last group rows past n are clamped copies
#pragma unroll
    for (int u = 0; u < RU; u++) {
        if (tile * (32 * RU) + 32 * u >= n) continue;
        int tt = tile * (32 * RU) + vr + 32 * u;
        tt = tt < n ? tt : n - 1;
        const float* vrow = kv_row(p, 1, tt, KV) + (size_t)kvh * HS + d0;
        vreg[u] = ((const f32x4*)vrow)[vc];
    }
}
template <int RU>
__device__ __forceinline__ void p16_av_store_tile(float* vt, const f32x4 (&vreg)[RU], int vr, int vc, int tile, int n) {
    // TRANSPOSED tile [32 columns][TP + 4]: a column's positions are contiguous (one ds_read_b128 = 4 links of its chain)
    constexpr int TPP = 32 * RU + 4;
#pragma unroll
    for (int u = 0; u < RU; u++)
        if (tile * (32 * RU) + 32 * u < n) {
            float* dst = vt + (size_t)(4 * vc) * TPP + vr + 32 * u;
            dst[0] = vreg[u].x; dst[TPP] = vreg[u].y; dst[2 * TPP] = vreg[u].z; dst[3 * TPP] = vreg[u].w;
        }
}
// Second launch of the two-launch form (contexts beyond what one workgroup ingests: JH_P16_ATT_FUSED, 1 k positions): softMax of one
// head's score row, then the value chains of 32 columns.  Two sequential chains of n links each bound it (tools/chain_lab.hip: 6.4 /
// 9.3 cycles per link with their LDS operand traffic), so everything else is taken off wave 0 and, at length, off the sum itself:
//   * the running sum of n >= 6144 exponentials is evaluated EXACTLY by all 256 threads (jh_seqsum.h / seq_pass: inside a binade
//     fl(s + x) depends on s only through the parity of its significand; maps compose; one scan per binade crossing) instead of n
//     dependent adds -- the sampler's method, bit-identical to the index-order loop by construction (tests/test_seqsum.py);
//   * V tiles of TP = 32 * RU <= 256 positions are double-buffered: waves 1..3 request and file tile k+1 while wave 0 runs the
//     chains of tile k (round 5: one tile buffer, every tile's memory round trip in front of its chain).
constexpr int P16_AV_SEQ_MIN = 6144;     // positions from which the parallel exact sum beats the chain (round 6, 8B: 1 k 456 vs 518 tok/s, 2 k 383 vs 415,
                                         // 4 k 301 vs 309, 8.1 k 212 vs 203: a scan + two barriers per binade crossing against 6.4 cycles per link)
constexpr int P16_AV_SEQ_E = 16;         // values per thread and window of seq_pass<256, 16>
template <int HS, int RU>
__global__ __launch_bounds__(P16_ATT_THREADS) void attn_p16_av_kernel(AttnParams p, const float* scores, int sc_stride, int w_cap, int seq_min) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ SeqShared sh;
    constexpr int NT = P16_ATT_THREADS, DW = 32, TP = 32 * RU, TPP = TP + 4;   // columns per workgroup, positions per tile
    constexpr int LT = NT - 64, RUV = (TP * 8 + LT - 1) / LT;                   // loader threads (waves 1..3), 16-byte pieces per loader thread and tile
    const int h = blockIdx.y, group = p.n_heads / p.n_kv_heads, kvh = h / group, d0 = blockIdx.x * DW;
    const int pos = p.st->pos, n = pos + 1;
    const int KV = p.n_kv_heads * HS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* vt0 = (float*)smem;                 // [2][DW][TPP] V tiles, transposed (a column's positions contiguous: one ds_read_b128 = 4 links)
    float* redf = vt0 + (size_t)2 * DW * TPP;  // [16]
    float* w = redf + 16;                      // [w_cap] scores -> exponentials -> softmax weights (zero-padded to a group of 16, readable 64 further)
    float* xt = w + w_cap;                     // [256 * 16] seq_pass's window
    const int lt = tid - 64, lvr = lt >> 3, lvc = lt & 7;   // loader thread: rows lvr, lvr + 24, ... of a tile, 16-byte piece lvc of the 32 columns
    auto load_tile = [&](f32x4 (&v)[RUV], int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < RUV; u++) {
            const int row = lvr + (LT / 8) * u;
            if (row >= TP || tile * TP + (row & ~31) >= n) continue;   // 32-row groups wholly behind the context are neither requested nor filed
            int tt = tile * TP + row;
            tt = tt < n ? tt : n - 1;                                   // inside the last group rows past n are clamped copies
            v[u] = ((const f32x4*)(kv_row(p, 1, tt, KV) + (size_t)kvh * HS + d0))[lvc];
        }
    };
    auto file_tile = [&](const f32x4 (&v)[RUV], int tile) __attribute__((always_inline)) {
        float* vt = vt0 + (size_t)(tile & 1) * DW * TPP;
#pragma unroll
        for (int u = 0; u < RUV; u++) {
            const int row = lvr + (LT / 8) * u;
            if (row >= TP || tile * TP + (row & ~31) >= n) continue;
            float* dst = vt + (size_t)(4 * lvc) * TPP + row;
            dst[0] = v[u].x; dst[TPP] = v[u].y; dst[2 * TPP] = v[u].z; dst[3 * TPP] = v[u].w;
        }
    };
    f32x4 vreg[RUV];
    if (wave != 0) load_tile(vreg, 0);         // in flight across the softmax
    float m = -INFINITY;
    {
        const float* srow = scores + (size_t)h * sc_stride;
        for (int tt = tid; tt < n; tt += NT) {
            const float s = srow[tt];
            w[tt] = s;
            m = fmaxf(m, s);
        }
    }
    m = wave_max(m);
    if (lane == 0) redf[wave] = m;
    __syncthreads();
    m = redf[0];
    for (int i = 1; i < NT / 64; i++) m = fmaxf(m, redf[i]);
    const int n16 = (n + 15) & ~15;            // zero-padded to a whole group of 16: fl(s + 0) = s
    for (int tt = tid; tt < n16; tt += NT) w[tt] = tt < n ? (float)exp((double)(w[tt] - m)) : 0.0f;   // (float)FastMath.exp(x - max)
    if (wave != 0) file_tile(vreg, 0);
    __syncthreads();
    float sum;
    if (n16 >= seq_min) {                      // VectorMath.java:80-85, one float accumulator in index order -- evaluated exactly by every thread
        int pick;
        seq_pass<NT, P16_AV_SEQ_E>(w, n16, INFINITY, sh, xt, sum, pick);
    } else {
        if (wave == 0) {
            const float s0 = p16_seq_sum_wave(w, n16, lane);
            if (lane == 0) redf[8] = s0;
        }
        __syncthreads();
        sum = redf[8];
    }
    for (int tt = tid; tt < n; tt += NT) w[tt] = w[tt] / sum;
    // ---- value[d] = fma chain over positions: lanes 0..31 of wave 0 own one column each; waves 1..3 stream the next tile meanwhile
    float acc = 0.0f;
    const int ntiles = (n + TP - 1) / TP;
    for (int tile = 0; tile < ntiles; tile++) {
        __syncthreads();                       // tile `tile` filed (and, first time round, the normalised weights visible); the other buffer is free
        if (wave == 0) {
            const int tbase = tile * TP, cnt = n - tbase < TP ? n - tbase : TP;
            p16_value_chain_tile(acc, vt0 + (size_t)(tile & 1) * DW * TPP + (size_t)(lane & (DW - 1)) * TPP, w + tbase + (lane & 15), cnt, TP);
        } else if (tile + 1 < ntiles) {
            f32x4 vnext[RUV];
            load_tile(vnext, tile + 1);
            file_tile(vnext, tile + 1);
        }
    }
    if (tid < DW) p.outf[(size_t)h * HS + d0 + tid] = acc;
}
// ------------------------------------------------------------------------------------------------ the two launches above in ONE
// (round 5).  Contexts of up to ~1 k positions do not need the scores spread over the chip: the K rows of a kv head are 512 bytes
// per position, and a workgroup that asks for them with 16-byte loads, two passes ahead, ingests a 400-position context in ~2 us --
// while the second launch of the split form costs a kernel boundary + ramp + the score round trip through memory (~4 us of the
// pair's 11.6).  Grid = n_kv_heads x group x HS/32 workgroups, id % n_kv_heads = kv head, so that the 16 workgroups that read one
// kv head's K / V run on one XCD (id % 8) and all but the first find the rows in its L2.  A workgroup = one query head x 32 value
// columns: it rotates q and the new k itself (bit-identical in every workgroup), takes ALL scores of its head (redundantly with the
// other three column quarters: 0.1 us of fmas), runs the softmax, then the value chains of its 32 columns exactly as
// attn_p16_av_kernel.  The workgroup (query head 0 of the group, columns 0..31) files the new K / V rows into the KV page.
//   score of position tt: GemmerF32's 16 lanes = the 4 lanes of a quad x 4 chains each (lane qd owns chains t = 4 qd + j: elements
//   16 c + 4 qd + j are ONE 16-byte piece of q and of the K row); the halving tree runs (t, t+8) -> quad_perm xor 2, (t, t+4) ->
//   quad_perm xor 1, then (0,2)/(1,3) and the last add inside the lane: jo_reduce16's association.
template <int HS, int RU, bool POW2>   // POW2: ctxPerPage is a power of two (every geometry but the 70B ones): the page split is a shift and a mask, no branch
__global__ __launch_bounds__(P16_FUSED_THREADS) void attn_p16_fused_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = P16_FUSED_THREADS, DW = 32, TP = 32 * RU, TPP = TP + 4, half = HS / 2, NC = HS / 16, PPB = NT / 4;   // PPB positions per pass
    constexpr int VR = NT / 8, RUV = TP / VR;                   // V rows requested per step by the workgroup, steps per tile
    static_assert(RUV >= 1 && RUV * VR == TP, "a V tile is a whole number of request steps");
    const int group = p.n_heads / p.n_kv_heads;
    const int id = blockIdx.x, kvh = id % p.n_kv_heads, rest = id / p.n_kv_heads, gi = rest % group, colq = rest / group;
    const int h = kvh * group + gi, d0 = colq * DW;
    const int pos = p.st->pos, n = pos + 1;
    const int KV = p.n_kv_heads * HS, A = p.n_heads * HS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, qd = tid & 3, prow = tid >> 2;
    float* vt = (float*)smem;              // [DW][TPP] V tile, TRANSPOSED: a column's positions are contiguous (one ds_read_b128 = 4 links of
                                           // its chain; rows 4 banks apart: conflict-free)
    float* redf = vt + (size_t)DW * TPP;   // [16]
    float* qs = redf + 16;                 // [HS] rotated q of this head
    float* knew = qs + HS;                 // [HS] rotated k of the new row
    float* w = knew + HS;                  // [n, padded to 64] scores -> softmax weights
    const float* qkv_row = p.qkv;
#define JH_FSTAMP(kk) do { if (p.dbg && tid == 0 && (id >> 3) < 16) p.dbg[(id >> 3) * 16 + (kk)] = wall_clock64(); } while (0)   // workgroups of kv head (id & 7)
    JH_FSTAMP(0);
    // ---- K rows of the first two passes: their addresses depend on the position only (one round trip together with q / rope)
    f32x4 ka[NC], kb[NC], kc[NC], kd[NC];
    // Row addresses.  A thread's K rows (prow + 64 * pass) and V rows (vr + 32 * u) are arithmetic progressions, so the page split
    // (position / ctxPerPage, position % ctxPerPage) is taken ONCE per progression and stepped -- a shift and a mask per step for
    // power-of-two pages, a subtract loop otherwise -- and a row's place is a 32-bit BYTE offset behind kv_base (the host launches
    // this kernel only while a layer page group stays below 4 GiB: contexts of <= ~1k positions are a few hundred MB), i.e. one
    // scalar base + one VGPR per load.  Round 6: kv_row() per row (a 64-bit multiply-add chain, and the division path's code at
    // every one of ~50 call sites) was most of the 2.4 us between kernel entry and the last request: ~2100 instructions in front of
    // the first barrier for a wave that issues one per ~4.5 cycles.
    const unsigned cpp = (unsigned)p.ctx_per_page, pe4 = (unsigned)p.page_elems * 4u, kvl4 = (unsigned)KV * 4u;
    const unsigned lrow_k = (unsigned)(p.rel_layer_in_page * 2) * cpp, lrow_v = lrow_k + cpp;
    struct RowWalk { unsigned cp, rc; };
    auto walk_at = [&](int tt) __attribute__((always_inline)) {
        RowWalk q;
        if constexpr (POW2) { q.cp = (unsigned)tt >> p.cpp_shift; q.rc = (unsigned)tt & (cpp - 1u); }
        else { q.cp = (unsigned)tt / cpp; q.rc = (unsigned)tt - q.cp * cpp; }
        return q;
    };
    auto walk_step = [&](RowWalk& q, unsigned d) __attribute__((always_inline)) {
        q.rc += d;
        if constexpr (POW2) { q.cp += q.rc >> p.cpp_shift; q.rc &= cpp - 1u; }
        else while (q.rc >= cpp) { q.rc -= cpp; ++q.cp; }
    };
    auto walk_off = [&](const RowWalk& q, unsigned lrow) __attribute__((always_inline)) { return q.cp * pe4 + (lrow + q.rc) * kvl4; };
    const char* kvb = (const char*)p.kv_base;
    const unsigned off_new_k = walk_off(walk_at(pos), lrow_k);      // the row being filed right now: what rows past n are clamped to
    RowWalk kw = walk_at(prow);                                  // K row of the next pass to be requested (requests go in pass order)
    int kw_t = prow;
    const unsigned k_lane = (unsigned)(kvh * HS + 4 * qd) * 4u;
    auto load_k = [&](f32x4 (&k)[NC], int pass) __attribute__((always_inline)) {
        (void)pass;
        unsigned off = walk_off(kw, lrow_k);
        off = kw_t < n ? off : off_new_k;
        const char* krow = kvb + (off + k_lane);
#pragma unroll
        for (int c = 0; c < NC; c++) k[c] = *(const f32x4*)(krow + 64 * c);
        walk_step(kw, (unsigned)PPB);
        kw_t += PPB;
    };
    const int npass = (n + PPB - 1) / PPB;
    // the RoPE operands (q / new k of the q|k|v row the previous launch just wrote: L2-warm; the table row) are requested BEFORE the
    // K rows (written tokens ago: HBM): loads retire in order, so the rotation and its barrier run while the K rows are still coming
    static_assert(2 * half <= NT, "one RoPE element pair per thread");
    const float* rf = p.rope + ((size_t)pos * half + (size_t)(kvh + p.kv_head_offset) * HS) * 2;
    const bool rope_lane = tid < 2 * half;
    const int rwhich = tid / half, rd = tid - rwhich * half;
    float rc = 0.f, rsn = 0.f, rx0 = 0.f, rx1 = 0.f;
    if (rope_lane) {
        const float* src = rwhich == 0 ? qkv_row + (size_t)h * HS : qkv_row + A + (size_t)kvh * HS;
        rc = rf[2 * rd]; rsn = rf[2 * rd + 1];
        rx0 = src[rd]; rx1 = src[rd + half];
    }
    load_k(ka, 0);
    if (npass > 1) load_k(kb, 1);
    // (passes 2, 3 and the V tile are requested BEHIND the rotation's barrier: a wave alone on its SIMD issues one instruction per ~4.5
    // cycles, and their ~400 instructions in front of the barrier delayed the first scores; K pass 0 is in flight either way)
    // ---- V tile 0 of this workgroup's 32 columns (position `pos` is being filed by another workgroup right now: it comes from the
    // q|k|v row instead; rows past n are clamped copies of it)
    const int vr = tid >> 3, vc = tid & 7;
    const unsigned v_lane = (unsigned)(kvh * HS + d0 + 4 * vc) * 4u;
    const f32x4* v_new = (const f32x4*)(qkv_row + A + KV + (size_t)kvh * HS + d0) + vc;   // the new row: not in the page yet
    auto load_v = [&](f32x4 (&vreg)[RUV], int tile) __attribute__((always_inline)) {
        RowWalk vw = walk_at(tile * TP + vr);
#pragma unroll
        for (int u = 0; u < RUV; u++) {
            if (tile * TP + VR * u < n) {                      // (wave-uniform)
                const int tt = tile * TP + vr + VR * u;
                const f32x4* vrow = tt >= pos ? v_new : (const f32x4*)(kvb + (walk_off(vw, lrow_v) + v_lane));   // rows past n: clamped copies of the new one
                vreg[u] = *vrow;
            }
            walk_step(vw, (unsigned)VR);
        }
    };
    // ---- RoPE of this head's q and of the new k row (table row pos + 2*kvHead: CausalSelfAttention.java:247-286)
    const bool owner = gi == 0 && colq == 0;
    if (rope_lane) {
        const int d = rd;
        const float r0 = rx0 * rc - rx1 * rsn, r1 = rx0 * rsn + rx1 * rc;   // contraction off: mul, mul, sub / add as in Java
        if (rwhich == 0) {
            qs[d] = r0; qs[d + half] = r1;
            if (p.tap_q && colq == 0) { p.tap_q[(size_t)h * HS + d] = r0; p.tap_q[(size_t)h * HS + d + half] = r1; }
        } else {
            knew[d] = r0; knew[d + half] = r1;
            if (owner) {   // K is stored post-RoPE (:273-286 rotates the page row in place)
                float* kdst = (float*)(kvb + off_new_k) + (size_t)kvh * HS;
                kdst[d] = r0; kdst[d + half] = r1;
            }
        }
    }
    if (owner)
        for (int d = tid; d < HS; d += NT) ((float*)(kvb + walk_off(walk_at(pos), lrow_v)) + (size_t)kvh * HS)[d] = qkv_row[A + KV + (size_t)kvh * HS + d];
    auto file_v = [&](const f32x4 (&vreg)[RUV], int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < RUV; u++)
            if (tile * TP + VR * u < n) {
                float* dst = vt + (size_t)(4 * vc) * TPP + vr + VR * u;
                dst[0] = vreg[u].x; dst[TPP] = vreg[u].y; dst[2 * TPP] = vreg[u].z; dst[3 * TPP] = vreg[u].w;
            }
    };
    f32x4 vreg[RUV];
    JH_FSTAMP(1);
    __syncthreads();
    JH_FSTAMP(2);
    f32x4 q4[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) q4[c] = *(const f32x4*)(qs + 16 * c + 4 * qd);
    if (npass > 2) load_k(kc, 2);
    if (npass > 3) load_k(kd, 3);
    load_v(vreg, 0);                                        // in flight across the scores and the softmax
    // ---- scores: PPB positions per pass, two passes in flight
    float m = -INFINITY;
    auto score_pass = [&](f32x4 (&k)[NC], int pass) __attribute__((always_inline)) {
        const int tt = pass * PPB + prow;
        if (tt == pos) {                                    // the page row is being written just now: the rotated row from LDS
#pragma unroll
            for (int c = 0; c < NC; c++) k[c] = *(const f32x4*)(knew + 16 * c + 4 * qd);
        }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; c++) {                      // GemmerF32 (PTO:1086-1102): chain t = 4 qd + j, steps of 16
            s0 = fmaf(q4[c].x, k[c].x, s0); s1 = fmaf(q4[c].y, k[c].y, s1); s2 = fmaf(q4[c].z, k[c].z, s2); s3 = fmaf(q4[c].w, k[c].w, s3);
        }
        s0 = s0 + dpp_f<0x4E>(s0); s1 = s1 + dpp_f<0x4E>(s1); s2 = s2 + dpp_f<0x4E>(s2); s3 = s3 + dpp_f<0x4E>(s3);   // v[i] + v[i + 8]
        s0 = s0 + dpp_f<0xB1>(s0); s1 = s1 + dpp_f<0xB1>(s1); s2 = s2 + dpp_f<0xB1>(s2); s3 = s3 + dpp_f<0xB1>(s3);   // a[i] + a[i + 4]
        const float sc = ((s0 + s2) + (s1 + s3)) * p.scale;                                                           // (b0+b2) + (b1+b3); ops.scale after the dot (:332)
        if (tt < n) {
            if (qd == 0) w[tt] = sc;
            m = fmaxf(m, sc);
        }
    };
    for (int pass = 0; pass < npass; pass += 4) {          // four passes (256 positions) in flight
        score_pass(ka, pass);
        if (pass + 4 < npass) load_k(ka, pass + 4);
        if (pass + 1 < npass) { score_pass(kb, pass + 1); if (pass + 5 < npass) load_k(kb, pass + 5); }
        if (pass + 2 < npass) { score_pass(kc, pass + 2); if (pass + 6 < npass) load_k(kc, pass + 6); }
        if (pass + 3 < npass) { score_pass(kd, pass + 3); if (pass + 7 < npass) load_k(kd, pass + 7); }
    }
    // ---- softMax (VectorMath.java:69-90): max, (float)exp in double, FLOAT sum in index order by one lane, division
    JH_FSTAMP(3);
    m = wave_max(m);
    if (lane == 0) redf[wave] = m;
    __syncthreads();
    m = redf[0];
    for (int i = 1; i < NT / 64; i++) m = fmaxf(m, redf[i]);
    const int n16 = (n + 15) & ~15;                         // the row is zero-padded to a whole group of 16: fl(s + 0) = s for the sum chain
    // Wave 0 owns the two sequential chains, so everything else is kept off it: it files ITS share of V tile 0 while waves 1..3 take
    // the exponentials (double exp: ~0.3 us per round), and they file theirs while it sums.  (Round 6 stamps: filed behind the sum,
    // wave 0's 64 ds_write_b32 per lane sat on the critical path for 0.56 us.)
    if (wave == 0) file_v(vreg, 0);
    else for (int tt = tid - 64; tt < n16; tt += NT - 64) w[tt] = tt < n ? (float)exp((double)(w[tt] - m)) : 0.0f;
    JH_FSTAMP(4);
    __syncthreads();                                        // exponentials visible
    JH_FSTAMP(5);
    if (wave != 0) file_v(vreg, 0);                         // waves 1..3 file their share of tile 0 WHILE wave 0 sums
    if (wave == 0) {
        const float sum = p16_seq_sum_wave(w, n16, lane);   // one float accumulator in index order (VectorMath.java:80-85)
        if (lane == 0) redf[8] = sum;
        JH_FSTAMP(6);
    }
    JH_FSTAMP(8);
    __syncthreads();
    JH_FSTAMP(9);
    const float sum = redf[8];
    for (int tt = tid; tt < n; tt += NT) w[tt] = w[tt] / sum;
    JH_FSTAMP(10);
    // ---- value[d] = fma chain over positions (saxpy per position, PTO:2648-2698): lanes 0..31 of wave 0 own one column each
    float acc = 0.0f;
    const int ntiles = (n + TP - 1) / TP;
    for (int tile = 0; tile < ntiles; tile++) {
        if (tile > 0) {
            __syncthreads();
            f32x4 vnext[RUV];
            load_v(vnext, tile);
            file_v(vnext, tile);
        }
        __syncthreads();
        if (tile == 0) JH_FSTAMP(11);
        if (wave == 0) {
            const int tbase = tile * TP, cnt = n - tbase < TP ? n - tbase : TP;
            p16_value_chain_tile(acc, vt + (size_t)(lane & (DW - 1)) * TPP, w + tbase + (lane & 15), cnt, TP);
        }
    }
    JH_FSTAMP(12);
    if (tid < DW) p.outf[(size_t)h * HS + d0 + tid] = acc;
    JH_FSTAMP(7);
#undef JH_FSTAMP
}
static inline size_t lds_bytes_attn_p16_fused(int max_ctx, int hs) {
    const int want = ((max_ctx + 63) & ~63) / 32;
    const int ru = want <= 2 ? 2 : want <= 4 ? 4 : want <= 8 ? 8 : 16;
    return ((size_t)32 * (ru * 32 + 4) + 16 + 2 * (size_t)hs + (size_t)((max_ctx + 63) & ~63) + 2 * 128) * 4;
}
static inline int p16_av_rows(int max_ctx) {   // RU: the whole context in one tile when it fits
    const int want = ((max_ctx + 63) & ~63) / 32;
    return want <= 2 ? 2 : want <= 4 ? 4 : want <= 8 ? 8 : 16;
}
// the two-launch form's second kernel: tiles of at most 256 positions, two of them; weight row + 256 readable floats; seq_pass's window
static inline int p16_av2_rows(int max_ctx) {
    const int want = ((max_ctx + 63) & ~63) / 32;
    return want <= 2 ? 2 : want <= 4 ? 4 : 8;
}
static inline int p16_av2_wcap(int max_ctx) { return ((max_ctx + 63) & ~63) + 256; }
static inline size_t lds_bytes_attn_p16(int max_ctx) {
    return ((size_t)2 * 32 * (p16_av2_rows(max_ctx) * 32 + 4) + 16 + (size_t)p16_av2_wcap(max_ctx) + (size_t)P16_ATT_THREADS * P16_AV_SEQ_E) * 4;
}

// ------------------------------------------------------------------------------------------------ prompt rows in reference order
// AbstractModel.batchForward (AbstractModel.java:295-312): the projections of a prompt chunk run on the F16 MFMA (jh_t16.h:
// rows_act_t16_kernel + gemm_t16_kernel, bit-identical to the GEMVs here by construction); this file keeps the attention side.
// Every (row, head) of CausalSelfAttention.java:314-356 is the computation the two decode kernels above do for one row -- the
// same 16-lane dot per score, the same float sum of the exponentials in index order, the same fma chain per output element --
// but laid out so that the rows of a chunk share what they read:
//   chain          = (row r of an 8-row tile, query head gi of the kv head's group): NCH = 8 * GROUP chains per (tile, kv head)
//   sT             = [tile][kv head][position j][chain] floats: scores, then exponentials (one 4*NCH-byte line per position)
//   rows_rope_kv   K (post-RoPE) and V rows of the whole chunk into the KV pages, q rotated in place, before any score is taken
//   rows_scores    a K row in registers serves all NCH chains (the decode kernel reads it once per row)
//   rows_max / rows_exp / rows_sum   softMax (VectorMath.java:69-90): max, (float)exp(double), then ONE float accumulator per chain
//                  in index order -- a lane per chain, 4*NCH bytes per step coalesced
//   rows_av        a V tile in LDS serves the 8 rows x GROUP heads of the tile (8 waves, a row each; lane = column pair): every
//                  lane runs GROUP * HS/64 fma chains over the positions, weights = e / sum broadcast from LDS
constexpr int P16_ROWS_TILE = 8;      // prompt rows per tile
constexpr int P16_ROWS_TJ = 64;       // positions per LDS tile / per scores workgroup

template <int HS>
__global__ __launch_bounds__(128) void rows_rope_kv_p16_kernel(AttnParams p) {
    constexpr int half = HS / 2;
    const int z = blockIdx.x, kvh = blockIdx.y, pos = p.batch_pos0 + z, group = p.n_heads / p.n_kv_heads;
    const int KV = p.n_kv_heads * HS, A = p.n_heads * HS;
    float* qkv_row = const_cast<float*>(p.qkv) + (size_t)z * p.ldqkv;
    const float* rf = p.rope + ((size_t)pos * half + (size_t)(kvh + p.kv_head_offset) * HS) * 2;
    const float* kh = qkv_row + A + (size_t)kvh * HS;
    float* kdst = (float*)kv_row(p, 0, pos, KV) + (size_t)kvh * HS;
    for (int d = threadIdx.x; d < half; d += 128) {
        const float c = rf[2 * d], s = rf[2 * d + 1];
        const float k0 = kh[d], k1 = kh[d + half];
        const float r0 = k0 * c - k1 * s, r1 = k0 * s + k1 * c;   // as attn_p16_scores_kernel (CausalSelfAttention.java:273-286)
        kdst[d] = r0; kdst[d + half] = r1;
        for (int gi = 0; gi < group; gi++) {                       // the group's query heads use the same table row (:247-266)
            float* qh = qkv_row + (size_t)(kvh * group + gi) * HS;
            const float q0 = qh[d], q1 = qh[d + half];
            qh[d] = q0 * c - q1 * s; qh[d + half] = q0 * s + q1 * c;
        }
    }
    float* vdst = (float*)kv_row(p, 1, pos, KV) + (size_t)kvh * HS;
    for (int d = threadIdx.x; d < HS; d += 128) vdst[d] = qkv_row[A + KV + (size_t)kvh * HS + d];
}

// grid (position slices of 64, kv heads, row tiles).  A 16-lane row owns positions j0 + prow + 16u (u = 0..3), K rows in
// registers; the chains' q vectors come from LDS in GemmerF32's lane order.  Pairs (row, position) beyond the row's context
// are computed like the others and never read.
template <int HS, int GROUP>
__global__ __launch_bounds__(256) void rows_scores_p16_kernel(AttnParams p, int rows, float* sT, int sc_stride) {
    constexpr int NCH = P16_ROWS_TILE * GROUP, NC = HS / 16, NR = (NCH + 15) / 16, PU = P16_ROWS_TJ / 16;
    __shared__ __attribute__((aligned(16))) float qs[NCH * 16 * NC];   // [chain][lane l][step c] = q[16c + l]
    const int ztile = blockIdx.z, kvh = blockIdx.y, j0 = blockIdx.x * P16_ROWS_TJ, z0 = ztile * P16_ROWS_TILE;
    const int zlast = z0 + P16_ROWS_TILE - 1 < rows - 1 ? z0 + P16_ROWS_TILE - 1 : rows - 1;
    const int nmax = p.batch_pos0 + zlast + 1;
    if (j0 >= nmax) return;
    const int KV = p.n_kv_heads * HS;
    const int tid = threadIdx.x, l = tid & 15, prow = tid >> 4;
    float kv[PU][NC];
#pragma unroll
    for (int u = 0; u < PU; u++) {
        int tt = j0 + prow + 16 * u;
        tt = tt < nmax ? tt : nmax - 1;
        const float* krow = kv_row(p, 0, tt, KV) + (size_t)kvh * HS + l;
#pragma unroll
        for (int c = 0; c < NC; c++) kv[u][c] = krow[16 * c];
    }
    for (int i = tid; i < NCH * HS; i += 256) {
        const int chain = i / HS, d = i - chain * HS, r = chain / GROUP, gi = chain - r * GROUP, z = z0 + r;
        const float v = z < rows ? p.qkv[(size_t)z * p.ldqkv + (size_t)(kvh * GROUP + gi) * HS + d] : 0.0f;
        qs[(chain * 16 + (d & 15)) * NC + (d >> 4)] = v;
    }
    __syncthreads();
    float res[PU][NR];
#pragma unroll
    for (int u = 0; u < PU; u++)
#pragma unroll
        for (int cb = 0; cb < NR; cb++) res[u][cb] = 0.0f;
#pragma unroll
    for (int cb = 0; cb < NR; cb++) {
#pragma unroll 2
        for (int ci = 0; ci < (NCH - cb * 16 < 16 ? NCH - cb * 16 : 16); ci++) {   // (not unrolled further: 8 q registers per chain in flight)
            const int chain = cb * 16 + ci;
            float q[NC];
#pragma unroll
            for (int c4 = 0; c4 < NC / 4; c4++) {
                const f32x4 t = *(const f32x4*)(qs + (chain * 16 + l) * NC + 4 * c4);
                q[4 * c4] = t.x; q[4 * c4 + 1] = t.y; q[4 * c4 + 2] = t.z; q[4 * c4 + 3] = t.w;
            }
#pragma unroll
            for (int u = 0; u < PU; u++) {
                float acc = 0.0f;
#pragma unroll
                for (int c = 0; c < NC; c++) acc = fmaf(q[c], kv[u][c], acc);   // GemmerF32 (PTO:1086-1102): lane l, steps of 16
                acc = row16_tree_sum(acc);                                      // every lane of the row holds the sum
                if (l == ci) res[u][cb] = acc;
            }
        }
    }
    float* out = sT + ((size_t)(ztile * p.n_kv_heads + kvh) * sc_stride) * NCH;
#pragma unroll
    for (int u = 0; u < PU; u++) {
        const int tt = j0 + prow + 16 * u;
        if (tt >= nmax) continue;
#pragma unroll
        for (int cb = 0; cb < NR; cb++)
            if (cb * 16 + l < NCH) out[(size_t)tt * NCH + cb * 16 + l] = res[u][cb] * p.scale;   // ops.scale after the dot (:332)
    }
}

// grid (row tiles * kv heads): maximum of every chain's score row
template <int GROUP>
__global__ __launch_bounds__(256) void rows_max_p16_kernel(int rows, int pos0, int n_kv, const float* sT, int sc_stride, float* mx) {
    constexpr int NCH = P16_ROWS_TILE * GROUP, JL = 256 / NCH;
    __shared__ float red[256];
    const int unit = blockIdx.x, ztile = unit / n_kv;
    const int tid = threadIdx.x, chain = tid % NCH, jl = tid / NCH, z = ztile * P16_ROWS_TILE + chain / GROUP;
    const int n = z < rows ? pos0 + z + 1 : 0;
    const float* src = sT + (size_t)unit * sc_stride * NCH + chain;
    float m = -INFINITY;
    for (int j = jl; j < n; j += JL) m = fmaxf(m, src[(size_t)j * NCH]);
    red[tid] = m;
    __syncthreads();
    if (tid < NCH) {
        for (int i = 1; i < JL; i++) m = fmaxf(m, red[i * NCH + tid]);
        mx[(size_t)unit * NCH + tid] = m;
    }
}
// grid (1024-element slices of a unit's lines, row tiles * kv heads): e = (float)FastMath.exp(x - max) in place (VectorMath.java:76-79)
template <int GROUP>
__global__ __launch_bounds__(256) void rows_exp_p16_kernel(int rows, int pos0, int n_kv, float* sT, int sc_stride, const float* mx) {
    constexpr int NCH = P16_ROWS_TILE * GROUP;
    const int unit = blockIdx.y, ztile = unit / n_kv, z0 = ztile * P16_ROWS_TILE;
    const int zlast = z0 + P16_ROWS_TILE - 1 < rows - 1 ? z0 + P16_ROWS_TILE - 1 : rows - 1;
    const int nmax = pos0 + zlast + 1;
    const int e0 = (blockIdx.x * 256 + threadIdx.x) * 4;   // element of the unit: position e0 / NCH, chains e0 % NCH .. + 3
    const int j = e0 / NCH, c0 = e0 - j * NCH;
    if (j >= nmax) return;
    f32x4* ptr = (f32x4*)(sT + (size_t)unit * sc_stride * NCH + e0);
    const f32x4 s = *ptr, m = *(const f32x4*)(mx + (size_t)unit * NCH + c0);
    f32x4 e;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int z = z0 + (c0 + i) / GROUP;
        e[i] = (z < rows && j <= pos0 + z) ? (float)exp((double)(s[i] - m[i])) : 0.0f;
    }
    *ptr = e;
}
// one wave per (row tile, kv head): lane = chain, one float accumulator in index order (VectorMath.java:80-85)
template <int GROUP>
__global__ __launch_bounds__(64) void rows_sum_p16_kernel(int rows, int pos0, int n_kv, const float* sT, int sc_stride, float* sums) {
    constexpr int NCH = P16_ROWS_TILE * GROUP;
    const int unit = blockIdx.x, ztile = unit / n_kv, lane = threadIdx.x;
    if (lane >= NCH) return;
    const int z = ztile * P16_ROWS_TILE + lane / GROUP;
    const int n = z < rows ? pos0 + z + 1 : 0;
    const float* src = sT + (size_t)unit * sc_stride * NCH + lane;
    float sum = 0.0f;
    int j = 0;
    for (; j + 8 <= n; j += 8) {
        float e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = src[(size_t)(j + i) * NCH];
#pragma unroll
        for (int i = 0; i < 8; i++) sum += e[i];
    }
    for (; j < n; j++) sum += src[(size_t)j * NCH];
    sums[(size_t)unit * NCH + lane] = sum;
}

// grid (row tiles * kv heads), 8 waves: wave w = row w of the tile.  LDS: two V tiles [TJ][HS] and two weight tiles [TJ][NCH];
// the next tiles are requested before the current ones are consumed and filed after, one barrier per tile.
constexpr int P16_ROWS_AV_THREADS = 64 * P16_ROWS_TILE;
static inline size_t lds_bytes_rows_av_p16(int hs, int group) { return (size_t)2 * P16_ROWS_TJ * (hs + P16_ROWS_TILE * group) * 4; }
template <int HS, int GROUP>
__global__ __launch_bounds__(P16_ROWS_AV_THREADS) void rows_av_p16_kernel(AttnParams p, int rows, const float* eT, int sc_stride, const float* sums) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = P16_ROWS_AV_THREADS, TJ = P16_ROWS_TJ, NCH = P16_ROWS_TILE * GROUP, DPL = HS / 64;
    constexpr int VU = TJ * HS / 4 / NT;                    // 16-byte V loads per thread and tile
    constexpr int PN4 = TJ * NCH / 4, PU = (PN4 + NT - 1) / NT;   // 16-byte weight loads per tile / per thread
    const int unit = blockIdx.x, ztile = unit / p.n_kv_heads, kvh = unit - ztile * p.n_kv_heads, z0 = ztile * P16_ROWS_TILE;
    const int zlast = z0 + P16_ROWS_TILE - 1 < rows - 1 ? z0 + P16_ROWS_TILE - 1 : rows - 1;
    const int nmax = p.batch_pos0 + zlast + 1;
    const int KV = p.n_kv_heads * HS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, z = z0 + w;
    const int n = z < rows ? p.batch_pos0 + z + 1 : 0;
    float* vt = (float*)smem;                               // [2][TJ][HS]
    float* pt = vt + 2 * TJ * HS;                           // [2][TJ][NCH]
    const float* esrc = eT + (size_t)unit * sc_stride * NCH;
    const f32x4 sum4 = *(const f32x4*)(sums + (size_t)unit * NCH + (tid * 4) % NCH);   // NT * 4 is a multiple of NCH: the same chains every tile
    f32x4 vreg[VU], preg[PU];
    auto request = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < VU; u++) {
            const int idx = tid + NT * u, pr = idx / (HS / 4), c4 = idx - pr * (HS / 4);
            int tt = tile * TJ + pr;
            tt = tt < nmax ? tt : nmax - 1;
            vreg[u] = *(const f32x4*)(kv_row(p, 1, tt, KV) + (size_t)kvh * HS + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const int idx = tid + NT * u;
            if (idx < PN4) preg[u] = *(const f32x4*)(esrc + (size_t)tile * TJ * NCH + 4 * idx);   // lines up to sc_stride exist
        }
    };
    auto file = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < VU; u++) *(f32x4*)(vt + (size_t)buf * TJ * HS + 4 * (tid + NT * u)) = vreg[u];
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const int idx = tid + NT * u;
            if (idx < PN4) {
                f32x4 wv;
                wv.x = preg[u].x / sum4.x; wv.y = preg[u].y / sum4.y; wv.z = preg[u].z / sum4.z; wv.w = preg[u].w / sum4.w;   // VectorMath.java:86-89
                *(f32x4*)(pt + (size_t)buf * TJ * NCH + 4 * idx) = wv;
            }
        }
    };
    float acc[GROUP][DPL];
#pragma unroll
    for (int gi = 0; gi < GROUP; gi++)
#pragma unroll
        for (int k = 0; k < DPL; k++) acc[gi][k] = 0.0f;
    const int ntiles = (nmax + TJ - 1) / TJ;
    request(0);
    file(0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; tile++) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) request(tile + 1);
        const int cnt = n - tile * TJ < TJ ? n - tile * TJ : TJ;   // this row's positions in the tile (<= 0: none)
        const float* vb = vt + (size_t)buf * TJ * HS + lane * DPL;
        const float* pb = pt + (size_t)buf * TJ * NCH + w * GROUP;
        auto step = [&](int jj) __attribute__((always_inline)) {
            float v[DPL], pw[GROUP];
#pragma unroll
            for (int k = 0; k < DPL; k++) v[k] = vb[jj * HS + k];
#pragma unroll
            for (int gi = 0; gi < GROUP; gi++) pw[gi] = pb[jj * NCH + gi];
#pragma unroll
            for (int gi = 0; gi < GROUP; gi++)
#pragma unroll
                for (int k = 0; k < DPL; k++) acc[gi][k] = fmaf(v[k], pw[gi], acc[gi][k]);   // saxpy per position (PTO:2648-2698)
        };
        int jj = 0;
        for (; jj + 8 <= cnt; jj += 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) step(jj + i);
        }
        for (; jj < cnt; jj++) step(jj);
        if (tile + 1 < ntiles) file(buf ^ 1);
        __syncthreads();
    }
    if (z < rows) {
#pragma unroll
        for (int gi = 0; gi < GROUP; gi++)
#pragma unroll
            for (int k = 0; k < DPL; k++) p.outf[(size_t)z * p.ldo + (size_t)(kvh * GROUP + gi) * HS + lane * DPL + k] = acc[gi][k];
    }
}


// ------------------------------------------------------------------------------------------------ Tier 1 in reference order
// batchDotProduct of the provider API (jh_gemm_*) with every float accumulation in the Panama-512 order, for EVERY offset / stride /
// window combination the reference's C entry points accept (nc/simd/vector_simd.h:22-38) and any M: one 16-lane DPP row per output
// element C[i, j] (four outputs per wave), lane t walks chain t over K in ascending order, then the halving tree.  Operands are read
// in the reference's own layouts (no operand copies): this is the Tier-1 path, whose cost is the PCIe round trip of the call.
//   I8 x Q4   PTO:807-850     acc_t = fma(da*sb, (float)(short)(lo_t*a[t] + hi_t*a[t+16]), acc_t) per block
//   F32 x Q4  PTO:336-374     acc_t = fma(a[t], (float)(lo_t-8)*s, acc_t); acc_t = fma(a[t+16], (float)(hi_t-8)*s, acc_t) per block
//   F32 x F32 PTO:1086-1102   acc_t = fma(a[l+t], b[l+t], acc_t), l = 0, 16, ...
//   BF16 x BF16 / F32 x BF16  PTO:1279-1311 / 1511-1538: per 32-element step elements t, then 16 + t = the same chain as l = 0, 16, ...
template <int KIND>
__global__ __launch_bounds__(256) void gemm_reford_kernel(GemmParams p) {
    const int lane = threadIdx.x & 63, r = lane >> 4, t = lane & 15;
    const long long total = (long long)p.m * p.n;
    const long long o = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + r;
    const bool live = o < total;
    const long long oo = live ? o : total - 1;              // surplus rows recompute the last output (every lane takes part in the DPP tree)
    const int i = (int)(oo / p.n), j = p.n0 + (int)(oo % p.n);
    float acc = 0.0f;
    if (KIND == G_Q8Q4) {
        const int8_t* ar = (const int8_t*)p.a + (size_t)p.lda * i + p.aoffset;
        const float* afr = p.af + (size_t)p.ldaf * i + p.aoffset / QB;
        const uint8_t* br = (const uint8_t*)p.b + (size_t)p.ldb * j + p.boffset;
        const float* bfr = p.bf + (size_t)p.ldbf * j + (p.boffset * 2) / QB;
        const int nblk = p.k / QB;
#pragma unroll 4
        for (int blk = 0; blk < nblk; blk++) {
            const int bb = br[blk * 16 + t];
            const int lo = (bb & 0x0F) - 8, hi = ((bb >> 4) & 0x0F) - 8;
            const int isum = lo * (int)ar[blk * 32 + t] + hi * (int)ar[blk * 32 + 16 + t];   // |.| <= 2032: the int16 of the reference, exact
            const float scale = afr[blk] * bfr[blk];
            acc = fmaf(scale, (float)isum, acc);
        }
    } else if (KIND == G_F32Q4) {
        const float* ar = (const float*)p.a + (size_t)p.lda * i + p.aoffset;
        const uint8_t* br = (const uint8_t*)p.b + (size_t)p.ldb * j + p.boffset;
        const float* bfr = p.bf + (size_t)p.ldbf * j + (p.boffset * 2) / QB;
        const int nblk = p.k / QB;
#pragma unroll 4
        for (int blk = 0; blk < nblk; blk++) {
            const int bb = br[blk * 16 + t];
            const float s = bfr[blk];
            const float wl = (float)((bb & 0x0F) - 8) * s, wh = (float)(((bb >> 4) & 0x0F) - 8) * s;
            acc = fmaf(ar[blk * 32 + t], wl, acc);
            acc = fmaf(ar[blk * 32 + 16 + t], wh, acc);
        }
    } else if (KIND == G_F32) {
        const float* ar = (const float*)p.a + (size_t)p.lda * i + p.aoffset;
        const float* br = (const float*)p.b + (size_t)p.ldb * j + p.boffset;
#pragma unroll 8
        for (int l = 0; l < p.k; l += 16) acc = fmaf(ar[l + t], br[l + t], acc);
    } else if (KIND == G_BF16) {
        const uint16_t* ar = (const uint16_t*)p.a + (size_t)p.lda * i + p.aoffset;
        const uint16_t* br = (const uint16_t*)p.b + (size_t)p.ldb * j + p.boffset;
#pragma unroll 8
        for (int l = 0; l < p.k; l += 16) acc = fmaf(bf16_to_f32(ar[l + t]), bf16_to_f32(br[l + t]), acc);
    } else {
        const float* ar = (const float*)p.a + (size_t)p.lda * i + p.aoffset;
        const uint16_t* br = (const uint16_t*)p.b + (size_t)p.ldb * j + p.boffset;
#pragma unroll 8
        for (int l = 0; l < p.k; l += 16) acc = fmaf(ar[l + t], bf16_to_f32(br[l + t]), acc);
    }
    acc = row16_tree_sum(acc);
    if (live && t == 0) p.r[(size_t)p.ldc * i + j - p.roffset] = acc;
}

}  // namespace jh
