// jh_kernels.h -- hand-written CDNA4 (gfx950, wave64) kernels for the Jlama decode hot path.
//
// What each kernel computes is fixed by the reference (file:line cited per kernel; abbreviations as in
// core/ = jlama-core/src/main/java/com/github/tjake/jlama/, PTO =
// core/tensor/operations/PanamaTensorOperations.java).  HOW it is computed is MI355X-first:
//   * Q4 weights are streamed exactly once per token with 16-byte-per-lane coalesced loads (one Q4 block of 32
//     weights per lane, 1 KiB per wave instruction), never staged through LDS (GEMV: LDS round trip is pure
//     overhead, cdna_hip_programming.md "glds vs register staging" table);
//   * the I8 activation row is quantized ONCE per workgroup into LDS (fused RMSNorm+Q8 prologue) and then
//     held in registers by the lane that owns the matching K blocks;
//   * block sums are exact integers (v_dot4_i32_i8), the -8 nibble bias is folded in as -8*sum(a_block);
//   * the K reduction is a wave64 DPP reduction (no LDS permutes), epilogues (residual add, SiLU*up+Q8) are fused.
// Compiled with -ffp-contract=off: every FMA is explicit (fmaf) where the reference calls FloatVector.fma().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "jh_seqsum.h"

namespace jh {

constexpr int QB = 32;  // Q4/Q8 block size (Q4ByteBufferTensor.java:36, Q8ByteBufferTensor.java:39)
using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x2 = int __attribute__((ext_vector_type(2)));
using f32x4 = float __attribute__((ext_vector_type(4)));   // native vector (arrays of HIP's float4 class type end up in scratch)

// ------------------------------------------------------------------------------------------------ helpers
// 64-lane reductions entirely on the VALU's DPP path: quad_perm / row_half_mirror / row_mirror inside a 16-lane row,
// then row_bcast15 (rows 1,3 += last lane of rows 0,2) and row_bcast31 (rows 2,3 += lane 31): lane 63 holds the total,
// v_readlane hands it to every lane as a scalar.  No LDS permute (ds_bpermute costs a ~100-cycle round trip per step).
// Same value as the xor butterfly: ((r3+r2)+(r1+r0)) vs ((r0+r1)+(r2+r3)), float addition commutes.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_fm(float v, float old) {   // rows outside ROWMASK receive `old`
    return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(old), __float_as_uint(v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]  (xor 1)
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]  (xor 2)
    v += dpp_f<0x141>(v);   // row_half_mirror      (pairs lanes across 4)
    v += dpp_f<0x140>(v);   // row_mirror           (pairs lanes across 8)
    v += dpp_fm<0x142, 0xA>(v, 0.0f);   // row_bcast15
    v += dpp_fm<0x143, 0xC>(v, 0.0f);   // row_bcast31
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    v = fmaxf(v, dpp_fm<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_fm<0x143, 0xC>(v, v));
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}
// sum over groups of LPR = 16 or 32 consecutive lanes; the result is valid in the LAST lane of every group
template <int LPR>
__device__ __forceinline__ float group_sum_last(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    if (LPR == 32) v += dpp_fm<0x142, 0xA>(v, 0.0f);
    return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_update_dpp(0, (unsigned)u, CTRL, 0xf, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0, (unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_dm(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_update_dpp(0, (unsigned)u, CTRL, ROWMASK, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0, (unsigned)(u >> 32), CTRL, ROWMASK, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    v += dpp_dm<0x142, 0xA>(v);   // row_bcast15 (other rows receive +0.0)
    v += dpp_dm<0x143, 0xC>(v);   // row_bcast31
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, 63), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// Workgroup barrier for LDS hand-offs ONLY.  __syncthreads() carries a workgroup-scope fence, for which hipcc emits
// s_waitcnt vmcnt(0): every global load in flight (our whole prefetched weight stream) would have to land before the
// barrier.  This one drains just the LDS queue (lgkmcnt), so weights keep streaming across prologue barriers
// (cdna_hip_programming.md §5 "Pipelining across barriers").  Never use it to order GLOBAL memory between waves.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// FloatConversions.float32ToBFloat16 (core/math/FloatConversions.java:35-60): RNE incl. carry into exponent
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t nbits = __float_as_uint(f);
    uint32_t s = (nbits >> 16) & 0x8000u, e = (nbits >> 16) & 0x7f80u, m = nbits & 0x7fffffu;
    if (e != 0x7f80u) {
        uint32_t mshift = m >> 16, masked = m & 0xffffu, m1;
        if (masked > 0x8000u) m1 = mshift + 1;
        else if (masked < 0x8000u) m1 = mshift;
        else m1 = (mshift & 1u) ? mshift + 1 : mshift;
        return (uint16_t)(s | (e + m1));
    }
    return m != 0 ? (uint16_t)0x7fc0 : (uint16_t)(nbits >> 16);
}

// Java (byte)(float) on the quantizer's value range: truncate toward zero, keep low 8 bits (PTO:1705-1710 F2B)
__device__ __forceinline__ int f2b(float v) { return ((int)v) & 0xff; }

// SiLU exactly as ActivationFunction.java:31: (float)(x * (1.0f / (1.0f + exp(-x)))) evaluated in double
__device__ __forceinline__ float silu_ref(float x) {
    double dx = (double)x;
    return (float)(dx * (1.0 / (1.0 + exp(-dx))));
}

__device__ __forceinline__ int sdot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

// exact integer dot of one Q4 block (16 bytes: low nibble = elem j, high = elem j+16,
// Q4ByteBufferTensor.java:88-106) with 32 int8 activations held as alo (elems 0..15) / ahi (16..31).
// Returns sum_t a[t]*nib[t]  (bias -8*sum(a) is applied by the caller).
__device__ __forceinline__ int q4_block_dot(const i32x4& w, const i32x4& alo, const i32x4& ahi) {
    int s = 0;
    s = sdot4(alo.x, w.x & 0x0F0F0F0F, s);
    s = sdot4(alo.y, w.y & 0x0F0F0F0F, s);
    s = sdot4(alo.z, w.z & 0x0F0F0F0F, s);
    s = sdot4(alo.w, w.w & 0x0F0F0F0F, s);
    s = sdot4(ahi.x, (w.x >> 4) & 0x0F0F0F0F, s);
    s = sdot4(ahi.y, (w.y >> 4) & 0x0F0F0F0F, s);
    s = sdot4(ahi.z, (w.z >> 4) & 0x0F0F0F0F, s);
    s = sdot4(ahi.w, (w.w >> 4) & 0x0F0F0F0F, s);
    return s;
}

__device__ __forceinline__ i32x4 ldg_nt(const i32x4* p) {
    // streamed-once weights: non-temporal hint (MI355X_MICROARCH.md row "nt-weights")
    return __builtin_nontemporal_load(p);
}

// system-scope accesses: memory that kernels of several devices meet in (tensor-parallel slots, flags)
__device__ __forceinline__ void st_sys(float* p, float v) {
    __hip_atomic_store((unsigned*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float ld_sys(const float* p) {
    return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

// ------------------------------------------------------------------------------------------------ decode state
struct DecodeState {
    int pos;     // position of the row being forwarded
    int token;   // its token id
    int step;    // index into out_tokens
    int done;    // a stop token was sampled (jh_session_set_eos): later replays of the decode graph change nothing
};

// ------------------------------------------------------------------------------------------------ GEMV params
// Prologues: how the activation row reaches LDS.  Epilogues: what happens to the dot products.
enum { PRO_Q8 = 0,       // already I8 + scales in global memory (Tier-1 jh_gemm_q8_q4)
       PRO_RMS_Q8 = 1,   // RMSNorm (core/model/RMSNorm.java:33-56) then Q8 quantize (PTO:1684-1723)
       PRO_QUANT_Q8 = 2, // Q8 quantize an F32 row (LlamaModel.maybeQuantize, core/model/llama/LlamaModel.java:176-184)
       PRO_F32 = 3,      // F32 row as is (F32xQ4)
       PRO_RMS_F32 = 4,  // RMSNorm, keep F32 (LM head: AbstractModel.java:443-449)
       PRO_ATTN_Q8 = 5 };// combine the attention slices (softmax-weighted sum) then Q8 quantize: o-projection input
enum { EPI_STORE = 0,    // out[j] = dot
       EPI_RESID = 1,    // out[j] = dot + resid[j]            (TransformerBlock.java:185,203)
       EPI_SILU_MUL = 2, // out[j] = silu(dot_gate[j]) * dot_up[j] (MLPBlock.java:132-142)
       EPI_TP = 3 };     // tensor-parallel shard: dot -> this shard's slot on every shard + workgroup flags (GemvParams::tp_*)

struct GemvParams {
    // No arrays in here on purpose: a kernarg array indexed by a runtime value makes hipcc spill the whole struct to
    // scratch (or fetch the pointer with a dependent vector load) before the first weight load can issue.
    const uint8_t* w;      // Q4 nibbles [nrows, K/2]  (q|k|v are stored stacked in one allocation)
    const float* ws;       // F32 block scales [nrows, K/32]
    const uint8_t* w2;     // EPI_SILU_MUL: the up-projection (w = gate)
    const float* ws2;
    float* out;            // F32 output [nrows]
    int nrows;
    int K;                 // columns (multiple of 32)
    int ldb;               // bytes per nibble row
    int ldbf;              // floats per scale row
    const float* x;        // F32 activation row (PRO_RMS_*, PRO_QUANT_Q8, PRO_F32)
    const float* nw;       // norm weights, F32 (BF16 on disk is widened at upload)
    float eps;
    const int8_t* aq;      // PRO_Q8: pre-quantized activation
    const float* ad;
    const float* resid;    // EPI_RESID
    float* amax_part;      // LM head: per-workgroup (max logit, index) partials
    int* amax_idx;
    // PRO_ATTN_Q8: attention slices published by attn_decode_kernel in "direct" mode (context <= direct_max)
    const float* part_o;   // [n_heads][part_stride][head_size]
    const float* part_ml;  // [n_heads][part_stride][2] = (max, sum) of each slice's local softmax
    const DecodeState* st;
    int direct_max, direct_chunk, part_stride, head_size, n_heads;
    float* tap_att;        // optional: combined attention output [A] ("after_attention" tap), written by workgroup 0
    // EPI_TP (tensor-parallel shard inside a replayed token graph): the partial row goes straight into this shard's slot on
    // EVERY shard (system-scope stores: peer memory over xGMI) instead of `out`, and each workgroup then raises its flag word
    // on every shard to the sequence number of this (token, layer) -- no separate scatter launch.
    float* const* tp_dst;        // [tp_n] slot of this shard on shard j
    unsigned* const* tp_flags;   // [tp_n] flag row of this shard on shard j, one word per workgroup of THIS launch
    const unsigned* tp_seq;      // tokens replayed so far (device word)
    int tp_n, tp_li, tp_L;
    long long* dbg;              // optional phase timestamps (wall_clock64, 100 MHz) of the reference-order few-row GEMVs: [workgroup][wave][8]
};
// EPI_TP's store: the shard's slot on every shard
__device__ __forceinline__ void tp_store(const GemvParams& p, int row, float v) {
    for (int j = 0; j < p.tp_n; j++) st_sys(p.tp_dst[j] + row, v);
}
// last statement of an EPI_TP kernel, passed once by EVERY wave of the workgroup (it holds a barrier).
// No fence: the slot stores are system-scope write-through stores, performed once vmcnt reaches 0; a __threadfence_system() here
// would be an L2 write-back + invalidate PER WAVE (4096 of them per o-projection launch: measured 5x slower than the separate
// scatter kernel).  Every wave drains its stores, the barrier collects the waves, one lane raises the flags (write-through too).
__device__ __forceinline__ void tp_signal(const GemvParams& p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned seq = *p.tp_seq * (unsigned)p.tp_L + (unsigned)p.tp_li + 1u;
        for (int j = 0; j < p.tp_n; j++) __hip_atomic_store(p.tp_flags[j] + blockIdx.x, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// LDS carve for an I8 activation row of nblk blocks
struct ActI8 {
    i32x4* lo;     // [nblk] elements 0..15 of each block
    i32x4* hi;     // [nblk] elements 16..31
    float* d;      // [nblk] block scales
    int* asum;     // [nblk] sum of the block's int8 values
    double* red;   // [32] reduction scratch
    float* wts;    // [256] PRO_ATTN_Q8: softmax-combine weight of (head, slice)
};
__device__ __forceinline__ ActI8 carve_i8(char* smem, int nblk) {
    ActI8 a;
    a.lo = (i32x4*)smem;
    a.hi = a.lo + nblk;
    a.d = (float*)(a.hi + nblk);
    a.asum = (int*)(a.d + nblk);
    a.red = (double*)(a.asum + nblk);   // 40*nblk bytes in: 8-byte aligned (no integer cast: keeps the LDS address space)
    a.wts = (float*)(a.red + 32);
    return a;
}
static inline size_t lds_bytes_i8(int K) { return (size_t)(K / QB) * (16 + 16 + 4 + 4) + 16 + 32 * 8 + 256 * 4; }

// block-wide double sum, result broadcast to every thread.  red: >= 32 doubles of LDS.
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum_d(v);
    if (lane == 0) red[wave] = v;
    lds_barrier();
    double t = 0.0;
    for (int i = 0; i < nw; i++) t += red[i];  // fixed order => deterministic
    lds_barrier();
    return t;
}

// RMSNorm scale factor (core/model/RMSNorm.java:41-49): float squares, double sum, /E, +eps, 1/sqrt in double.
__device__ __forceinline__ float rms_factor(const float* __restrict__ x, int K, float eps, double* red) {
    double ss = 0.0;
    const float4* x4 = (const float4*)x;
    for (int j = threadIdx.x; j < K / 4; j += blockDim.x) {
        float4 v = x4[j];
        ss += (double)(v.x * v.x);
        ss += (double)(v.y * v.y);
        ss += (double)(v.z * v.z);
        ss += (double)(v.w * v.w);
    }
    ss = block_sum_d(ss, red);
    ss /= (double)K;
    ss += (double)eps;
    ss = 1.0 / sqrt(ss);
    return (float)ss;
}

// norm weights are widened to F32 once at upload (bf16 -> f32 is exact), so the prologue's loads are branch-free
__device__ __forceinline__ void load8_norm(const float* nw, int e0, float (&w)[8]) {
    float4 a = *(const float4*)(nw + e0), b = *(const float4*)(nw + e0 + 4);
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}

// Quantize 8 consecutive values held by one lane; the 4 lanes of a quad cover one block of 32.
// Panama quantizeQ8_512 (PTO:1684-1723): d = max/127, id = 127/max (0 if max==0), q = (byte)(x*id + 0.5f).
__device__ __forceinline__ void quad_quantize_store(const float (&y)[8], int unit, const ActI8& a) {
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) amax = fmaxf(amax, fabsf(y[i]));
    amax = fmaxf(amax, dpp_f<0xB1>(amax));   // quad butterfly on the DPP path (no LDS round trip)
    amax = fmaxf(amax, dpp_f<0x4E>(amax));
    const float d = amax / 127.0f;
    const float id = (amax != 0.0f) ? 127.0f / amax : 0.0f;
    int q[8];
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float v = y[i] * id;
        v = v + 0.5f;
        q[i] = f2b(v);
        s += (int)(int8_t)q[i];
    }
    s += __builtin_amdgcn_update_dpp(0, s, 0xB1, 0xf, 0xf, false);
    s += __builtin_amdgcn_update_dpp(0, s, 0x4E, 0xf, 0xf, false);
    const int blk = unit >> 2, sub = unit & 3;
    i32x2 packed;
    packed.x = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
    packed.y = q[4] | (q[5] << 8) | (q[6] << 16) | (q[7] << 24);
    i32x2* dst = (i32x2*)((sub < 2) ? (a.lo + blk) : (a.hi + blk)) + (sub & 1);
    *dst = packed;
    if (sub == 0) {
        a.d[blk] = d;
        a.asum[blk] = s;
    }
}

// Prologue: build the I8 activation row in LDS (once per workgroup), in two halves so the caller can put its
// weight loads BETWEEN them:  stage_issue() only issues the loads of x / norm weights (one round trip, up to UMAX
// 8-element units per thread kept in registers);  stage_finish() reduces, quantizes and fills LDS.
// Order matters: s_waitcnt vmcnt retires loads oldest-first, so x must be requested BEFORE the (much larger) weight
// stream or the prologue would wait for every weight byte.
// register-resident units per thread: 2 covers K <= 8192 at 512 threads; the plain-quantize prologue (o-proj, down:
// K up to 14336 = 1792 units) keeps 4 so that no unit needs a second, serialised global round trip
template <int PRO> struct UMaxFor { static constexpr int v = 2; };
template <> struct UMaxFor<PRO_QUANT_Q8> { static constexpr int v = 4; };
template <int UMAX>
struct ActRegsT {
    float xv[UMAX][8];
    float wv[UMAX][8];
    float po[UMAX][4][8];   // PRO_ATTN_Q8: this thread's 8 output elements of up to 4 slices
    float m, l;             // PRO_ATTN_Q8: (max, sum) of slice (tid&3) of head (tid>>2)
    int n;                  // context length pos+1
    i32x4 ql, qh;           // PRO_Q8: block min(tid, nblk-1) of the pre-quantized row, requested BEFORE the weight stream
    float qd;
};
template <int PRO>
__device__ __forceinline__ void stage_issue(const GemvParams& p, ActRegsT<UMaxFor<PRO>::v>& r) {
    constexpr int UMAX = UMaxFor<PRO>::v;
    if (PRO == PRO_Q8) {
        // vmcnt retires oldest-first: the codes must be requested before the weights or this prologue waits for every weight byte
        const int nblk = p.K / QB;
        const int b = (int)threadIdx.x < nblk ? (int)threadIdx.x : nblk - 1;
        const i32x4* src = (const i32x4*)(p.aq + (size_t)b * QB);
        r.ql = src[0];
        r.qh = src[1];
        r.qd = p.ad[b];
        return;
    }
    const int units = p.K / 8, T = blockDim.x;
    // branch-free (clamped) addresses: a guarded load would make hipcc wait for it at the end of its basic block,
    // serialising this round trip with the weight stream that is issued next
    if (PRO == PRO_ATTN_Q8) {
        r.n = p.st->pos + 1;
        int t = threadIdx.x;
        t = t < p.n_heads * 4 ? t : p.n_heads * 4 - 1;
        const float* mlp = p.part_ml + ((size_t)(t >> 2) * p.part_stride + (t & 3)) * 2;
        r.m = mlp[0];
        r.l = mlp[1];
    }
#pragma unroll
    for (int u = 0; u < UMAX; u++) {
        int unit = threadIdx.x + u * T;
        unit = unit < units ? unit : units - 1;
        const float4 xa = *(const float4*)(p.x + unit * 8), xb = *(const float4*)(p.x + unit * 8 + 4);
        r.xv[u][0] = xa.x; r.xv[u][1] = xa.y; r.xv[u][2] = xa.z; r.xv[u][3] = xa.w;
        r.xv[u][4] = xb.x; r.xv[u][5] = xb.y; r.xv[u][6] = xb.z; r.xv[u][7] = xb.w;
        if (PRO == PRO_RMS_Q8) load8_norm(p.nw, unit * 8, r.wv[u]);
        if (PRO == PRO_ATTN_Q8) {
            const int e0 = unit * 8, h = e0 / p.head_size, d0 = e0 - h * p.head_size;
#pragma unroll
            for (int sl = 0; sl < 4; sl++) {
                const float* po = p.part_o + ((size_t)h * p.part_stride + sl) * p.head_size + d0;
                const float4 a = *(const float4*)po, b = *(const float4*)(po + 4);
                r.po[u][sl][0] = a.x; r.po[u][sl][1] = a.y; r.po[u][sl][2] = a.z; r.po[u][sl][3] = a.w;
                r.po[u][sl][4] = b.x; r.po[u][sl][5] = b.y; r.po[u][sl][6] = b.z; r.po[u][sl][7] = b.w;
            }
        }
    }
}
template <int PRO>
__device__ __forceinline__ void stage_finish(const GemvParams& p, const ActI8& a, ActRegsT<UMaxFor<PRO>::v>& r) {
    constexpr int UMAX = UMaxFor<PRO>::v;
    const int K = p.K, nblk = K / QB;
    if (PRO == PRO_Q8) {
        auto put = [&](int blk, const i32x4& l, const i32x4& h, float d) {
            a.lo[blk] = l;
            a.hi[blk] = h;
            a.d[blk] = d;
            int s = 0;
            s = sdot4(l.x, 0x01010101, s); s = sdot4(l.y, 0x01010101, s);
            s = sdot4(l.z, 0x01010101, s); s = sdot4(l.w, 0x01010101, s);
            s = sdot4(h.x, 0x01010101, s); s = sdot4(h.y, 0x01010101, s);
            s = sdot4(h.z, 0x01010101, s); s = sdot4(h.w, 0x01010101, s);
            a.asum[blk] = s;
        };
        if ((int)threadIdx.x < nblk) put(threadIdx.x, r.ql, r.qh, r.qd);          // fetched by stage_issue, ahead of the weights
        for (int blk = threadIdx.x + blockDim.x; blk < nblk; blk += blockDim.x) {   // rows longer than the workgroup
            const i32x4* src = (const i32x4*)(p.aq + (size_t)blk * QB);
            put(blk, src[0], src[1], p.ad[blk]);
        }
    } else {
        const int units = K / 8, T = blockDim.x;
        float fs = 1.0f;
        bool direct = false;
        int S = 1;
        if (PRO == PRO_ATTN_Q8) {
            // "direct" mode of attn_decode_kernel: <= 4 slices per head were published as (o_s, m_s, l_s); combine them
            // here, under this GEMV's weight prefetch:  w_s = l_s*exp(m_s - M) / sum_s(l_s*exp(m_s - M)),  o = sum_s w_s*o_s
            direct = r.n <= p.direct_max;
            S = (r.n + p.direct_chunk - 1) / p.direct_chunk;
            if (direct) {
                const bool valid = (int)(threadIdx.x & 3) < S && (int)threadIdx.x < p.n_heads * 4;
                const float m = valid ? r.m : -INFINITY;
                float M = fmaxf(m, dpp_f<0xB1>(m));
                M = fmaxf(M, dpp_f<0x4E>(M));
                float e = valid ? r.l * (float)exp((double)(m - M)) : 0.0f;
                float L = e + dpp_f<0xB1>(e);
                L = L + dpp_f<0x4E>(L);
                if ((int)threadIdx.x < p.n_heads * 4) a.wts[threadIdx.x] = e / L;
            }
            lds_barrier();
        }
        if (PRO == PRO_RMS_Q8) {
            // RMSNorm (core/model/RMSNorm.java:41-49): float squares, double sum, /E, +eps, 1/sqrt in double
            double ss = 0.0;
#pragma unroll
            for (int u = 0; u < UMAX; u++)
                if (threadIdx.x + u * T < units)
#pragma unroll
                    for (int i = 0; i < 8; i++) ss += (double)(r.xv[u][i] * r.xv[u][i]);
            for (int unit = threadIdx.x + UMAX * T; unit < units; unit += T) {   // rows longer than UMAX*T*8
                const float4 xa = *(const float4*)(p.x + unit * 8), xb = *(const float4*)(p.x + unit * 8 + 4);
                ss += (double)(xa.x * xa.x); ss += (double)(xa.y * xa.y); ss += (double)(xa.z * xa.z); ss += (double)(xa.w * xa.w);
                ss += (double)(xb.x * xb.x); ss += (double)(xb.y * xb.y); ss += (double)(xb.z * xb.z); ss += (double)(xb.w * xb.w);
            }
            ss = block_sum_d(ss, a.red);
            ss /= (double)K;
            ss += (double)p.eps;
            ss = 1.0 / sqrt(ss);
            fs = (float)ss;
        }
#pragma unroll
        for (int u = 0; u < UMAX; u++) {
            const int unit = threadIdx.x + u * T;
            if (unit < units) {
                float y[8];
#pragma unroll
                for (int i = 0; i < 8; i++) y[i] = (PRO == PRO_RMS_Q8) ? r.wv[u][i] * (fs * r.xv[u][i]) : r.xv[u][i];  // (0 + w) * ((float)ss * x)
                if (PRO == PRO_ATTN_Q8 && direct) {
                    const int h = (unit * 8) / p.head_size;
#pragma unroll
                    for (int i = 0; i < 8; i++) y[i] = 0.0f;
#pragma unroll
                    for (int sl = 0; sl < 4; sl++)
                        if (sl < S) {
                            const float w = a.wts[h * 4 + sl];
#pragma unroll
                            for (int i = 0; i < 8; i++) y[i] = fmaf(r.po[u][sl][i], w, y[i]);
                        }
                    if (p.tap_att && blockIdx.x == 0) {
#pragma unroll
                        for (int i = 0; i < 8; i++) p.tap_att[unit * 8 + i] = y[i];
                    }
                }
                quad_quantize_store(y, unit, a);
            }
        }
        for (int unit = threadIdx.x + UMAX * T; unit < units; unit += T) {
            const float4 xa = *(const float4*)(p.x + unit * 8), xb = *(const float4*)(p.x + unit * 8 + 4);
            float y[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            if (PRO == PRO_RMS_Q8) {
                float w[8];
                load8_norm(p.nw, unit * 8, w);
#pragma unroll
                for (int i = 0; i < 8; i++) y[i] = w[i] * (fs * y[i]);
            }
            quad_quantize_store(y, unit, a);
        }
    }
    lds_barrier();
}

// ------------------------------------------------------------------------------------------------ K1: streaming GEMV I8 x Q4
// batchDotProduct I8xQ4 at M=1 (GemmerI8Q4_512, PTO:768-1044; C twin nc/simd/vector_simd.c:261-437):
//   C[j] = sum_blk (da[blk]*sb[j,blk]) * (float) sum_t a[blk,t]*(nib[j,blk,t]-8)
// Shape of the kernel = the streaming-read microbenchmark that reaches 7.1 TB/s on this chip (tools/membw.hip):
// one 256-thread workgroup per CU, every lane keeps >= 8 independent 16-byte non-temporal loads in flight.
// A wave owns a contiguous range of rows and walks it in groups of R rows; lane l owns K blocks l, l+64, ...
// (NB per lane).  The next group's weights are loaded into a second register set BEFORE the current group is
// reduced, and the first group is requested before the activation prologue runs, so the HBM stream never drains.
template <int R, int NB>
struct WBuf {
    i32x4 w[R][NB];
    float s[R][NB];
};

template <int EPI, int R, int NB>
__device__ __forceinline__ void load_group(const GemvParams& p, int g, int lane, WBuf<R, NB>& b) {
    if (EPI == EPI_SILU_MUL) {
        // group g = hidden units j0..j0+R/2-1: rows [0,R/2) are gate rows, [R/2,R) the matching up rows
        const int j0 = g * (R / 2);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int j = j0 + r % (R / 2);
            const i32x4* wp = (const i32x4*)((r < R / 2 ? p.w : p.w2) + (size_t)j * p.ldb);
            const float* sp = (r < R / 2 ? p.ws : p.ws2) + (size_t)j * p.ldbf;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                b.w[r][i] = __builtin_nontemporal_load(wp + lane + 64 * i);
                b.s[r][i] = __builtin_nontemporal_load(sp + lane + 64 * i);
            }
        }
    } else {
        const uint8_t* wbase = p.w + (size_t)g * R * p.ldb;
        const float* sbase = p.ws + (size_t)g * R * p.ldbf;
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int i = 0; i < NB; i++) {
                b.w[r][i] = __builtin_nontemporal_load((const i32x4*)(wbase + (size_t)r * p.ldb) + lane + 64 * i);
                b.s[r][i] = __builtin_nontemporal_load(sbase + (size_t)r * p.ldbf + lane + 64 * i);
            }
    }
}

template <int EPI, int R>
__device__ __forceinline__ void store_group(const GemvParams& p, int g, int lane, const float (&acc)[R]) {
    // acc[] holds full sums in every lane (butterfly).  Lane r stores row r of the group: one coalesced store.
    if (EPI == EPI_SILU_MUL) {
        float gsel = 0.0f, usel = 0.0f;   // select first, then ONE double-precision SiLU per lane (not R/2 serial ones)
#pragma unroll
        for (int r = 0; r < R / 2; r++)
            if (lane == r) { gsel = acc[r]; usel = acc[r + R / 2]; }
        const float h = silu_ref(gsel) * usel;
        if (lane < R / 2) p.out[g * (R / 2) + lane] = h;
    } else {
        float v = 0.0f;
#pragma unroll
        for (int r = 0; r < R; r++)
            if (lane == r) v = acc[r];
        if (lane < R) {
            if (EPI == EPI_RESID) v = v + p.resid[g * R + lane];   // accumulate(...) TransformerBlock.java:185,203
            if (EPI == EPI_TP) tp_store(p, g * R + lane, v);
            else p.out[g * R + lane] = v;
        }
    }
}

// PIPE = 0: one group per wave, every weight load issued before the activation prologue ("single shot": the whole
//           GEMV is requested from HBM at t=0 and the prologue hides under the first-byte latency);
// PIPE = 1: a wave walks several groups, loading group g+1 into a second register set while it reduces group g.
template <int PRO, int EPI, int R, int NB, int PIPE>
__global__ __launch_bounds__((R * NB > 4) ? 512 : 1024) void gemv_i8q4_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nblk = p.K / QB;
    const ActI8 a = carve_i8(smem, nblk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    // rows -> groups -> contiguous group range per wave
    const int total = (EPI == EPI_SILU_MUL) ? 2 * p.nrows : p.nrows;
    const int ngroups = total / R;
    const int tw = gridDim.x * nwaves, gw = blockIdx.x * nwaves + wave;
    const int per = (ngroups + tw - 1) / tw;
    const int g0 = gw * per;
    int g1 = g0 + per;
    if (g1 > ngroups) g1 = ngroups;

    if constexpr (NB > 0 && PIPE == 0) {
        WBuf<R, NB> cur;
        ActRegsT<UMaxFor<PRO>::v> ar;
        const int g = gw;   // host launches >= ngroups waves
        stage_issue<PRO>(p, ar);                                    // activation loads first (retire first) ...
        load_group<EPI, R, NB>(p, g < ngroups ? g : ngroups - 1, lane, cur);   // ... then this wave's whole weight stream
        stage_finish<PRO>(p, a, ar);                                // prologue runs under the weights' flight time
        if (g < ngroups) {
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = 0.0f;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                const int blk = lane + 64 * i;
                const i32x4 alo = a.lo[blk], ahi = a.hi[blk];
                const float adv = a.d[blk];
                const int as8 = 8 * a.asum[blk];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int isum = q4_block_dot(cur.w[r][i], alo, ahi) - as8;
                    acc[r] = fmaf(adv * cur.s[r][i], (float)isum, acc[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]);
            store_group<EPI, R>(p, g, lane, acc);
        }
    } else if constexpr (NB > 0) {
        WBuf<R, NB> cur, nxt;
        ActRegsT<UMaxFor<PRO>::v> ar;
        stage_issue<PRO>(p, ar);
        load_group<EPI, R, NB>(p, g0 < ngroups ? g0 : ngroups - 1, lane, cur);   // first group in flight across the prologue
        stage_finish<PRO>(p, a, ar);
        i32x4 alo[NB], ahi[NB];
        float adv[NB];
        int as8[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int blk = lane + 64 * i;
            alo[i] = a.lo[blk]; ahi[i] = a.hi[blk]; adv[i] = a.d[blk]; as8[i] = 8 * a.asum[blk];
        }
        // EPI_SILU_MUL: the double-precision SiLU costs ~500 SIMD cycles per evaluation, so the wave parks each group's
        // (gate, up) sums in lane `nbuf + r` and evaluates SiLU*up ONCE for up to 64 hidden units (one coalesced store)
        float gsel = 0.0f, usel = 0.0f;
        int nbuf = 0, obase = g0 * (R / 2);
        for (int g = g0; g < g1; g++) {
            if (g + 1 < g1) load_group<EPI, R, NB>(p, g + 1, lane, nxt);
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = 0.0f;
#pragma unroll
            for (int i = 0; i < NB; i++)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int isum = q4_block_dot(cur.w[r][i], alo[i], ahi[i]) - as8[i];
                    acc[r] = fmaf(adv[i] * cur.s[r][i], (float)isum, acc[r]);
                }
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]);
            if constexpr (EPI == EPI_SILU_MUL) {
#pragma unroll
                for (int r = 0; r < R / 2; r++)
                    if (lane == nbuf + r) { gsel = acc[r]; usel = acc[r + R / 2]; }
                nbuf += R / 2;
                if (nbuf + R / 2 > 64 || g + 1 == g1) {
                    const float h = silu_ref(gsel) * usel;
                    if (lane < nbuf) p.out[obase + lane] = h;
                    obase += nbuf;
                    nbuf = 0;
                }
            } else {
                store_group<EPI, R>(p, g, lane, acc);
            }
            cur = nxt;
        }
    } else {
        // generic K (any multiple of 32): activation re-read from LDS per block, no register double buffering
        ActRegsT<UMaxFor<PRO>::v> ar;
        stage_issue<PRO>(p, ar);
        stage_finish<PRO>(p, a, ar);
        for (int g = g0; g < g1; g++) {
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = 0.0f;
            for (int blk = lane; blk < nblk; blk += 64) {
                const i32x4 l = a.lo[blk], h = a.hi[blk];
                const float da = a.d[blk];
                const int as8 = 8 * a.asum[blk];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint8_t* wr;
                    const float* sr;
                    if (EPI == EPI_SILU_MUL) {
                        const int j = g * (R / 2) + r % (R / 2);
                        wr = (r < R / 2 ? p.w : p.w2) + (size_t)j * p.ldb;
                        sr = (r < R / 2 ? p.ws : p.ws2) + (size_t)j * p.ldbf;
                    } else {
                        wr = p.w + (size_t)(g * R + r) * p.ldb;
                        sr = p.ws + (size_t)(g * R + r) * p.ldbf;
                    }
                    const i32x4 wv = __builtin_nontemporal_load((const i32x4*)wr + blk);
                    const int isum = q4_block_dot(wv, l, h) - as8;
                    acc[r] = fmaf(da * sr[blk], (float)isum, acc[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]);
            store_group<EPI, R>(p, g, lane, acc);
        }
    }
    if constexpr (EPI == EPI_TP) tp_signal(p);
}

// ------------------------------------------------------------------------------------------------ K1c: streaming GEMV F32 x Q4
// batchDotProduct F32xQ4 at M=1 (GemmerF32Q4_512 PTO:336-374; C twin vector_simd.c:880-965): dequantize first
// w = float(nib-8)*scale, then acc = fma(a, w, acc).  fma(scale, float(nib), -8*scale) == round(scale*(nib-8))
// exactly, so the per-weight cost is cvt + fma + fma.  Used by the LM head (AbstractModel.java:443-449): the
// normed hidden row is NOT re-quantized.  The epilogue keeps a per-workgroup running argmax
// (strict >, lowest index wins: AbstractModel.java:455-469).
struct ActF32 {
    float4* y;    // [8][nblk] chunk-major: chunk c (4 floats) of block blk at y[c*nblk + blk]
    double* red;  // [32]
    float* bestv; // [16]
    int* besti;   // [16]
};
__device__ __forceinline__ ActF32 carve_f32(char* smem, int nblk) {
    ActF32 a;
    a.y = (float4*)smem;
    a.red = (double*)(a.y + 8 * (size_t)nblk);
    a.bestv = (float*)(a.red + 32);
    a.besti = (int*)(a.bestv + 16);
    return a;
}
static inline size_t lds_bytes_f32(int K) { return (size_t)K * 4 + 32 * 8 + 16 * 4 + 16 * 4; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
// Per weight: v_cvt_f32_ubyteN + fma (dequantize) + fma (accumulate) -- at 525 M weights the LM head was VALU-bound (69 us
// against 54 us of HBM time).  Two weights per v_pk_fma_f32: even elements accumulate in .x, odd ones in .y (each an
// ascending-k fma chain with the reference's per-element operations); the halves are added once per row.
// byte N of a dword as float in ONE instruction.  Written as asm because hipcc folds `(x & 0x0F0F0F0F) >> 8 & 0xff` back into
// per-nibble shift + and + cvt_ubyte0 (3 VALU ops per weight instead of 1 + 3/8).
template <int N>
__device__ __forceinline__ float cvt_ubyte(int x) {
    float f;
    if constexpr (N == 0) asm("v_cvt_f32_ubyte0_e32 %0, %1" : "=v"(f) : "v"(x));
    else if constexpr (N == 1) asm("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(f) : "v"(x));
    else if constexpr (N == 2) asm("v_cvt_f32_ubyte2_e32 %0, %1" : "=v"(f) : "v"(x));
    else asm("v_cvt_f32_ubyte3_e32 %0, %1" : "=v"(f) : "v"(x));
    return f;
}
__device__ __forceinline__ f32x2 q4_block_dot_f32(const i32x4& w, float scale, const float4 (&y)[8], f32x2 acc) {
    const f32x2 sc = {scale, scale};
    const float m8s = -8.0f * scale;
    const f32x2 m8 = {m8s, m8s};
    const int wl[4] = {w.x & 0x0F0F0F0F, w.y & 0x0F0F0F0F, w.z & 0x0F0F0F0F, w.w & 0x0F0F0F0F};
    const int wh[4] = {(w.x >> 4) & 0x0F0F0F0F, (w.y >> 4) & 0x0F0F0F0F, (w.z >> 4) & 0x0F0F0F0F,
                       (w.w >> 4) & 0x0F0F0F0F};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const f32x2 f01 = {cvt_ubyte<0>(wl[c]), cvt_ubyte<1>(wl[c])};
        const f32x2 f23 = {cvt_ubyte<2>(wl[c]), cvt_ubyte<3>(wl[c])};
        acc = __builtin_elementwise_fma(f32x2{y[c].x, y[c].y}, __builtin_elementwise_fma(sc, f01, m8), acc);
        acc = __builtin_elementwise_fma(f32x2{y[c].z, y[c].w}, __builtin_elementwise_fma(sc, f23, m8), acc);
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const f32x2 f01 = {cvt_ubyte<0>(wh[c]), cvt_ubyte<1>(wh[c])};
        const f32x2 f23 = {cvt_ubyte<2>(wh[c]), cvt_ubyte<3>(wh[c])};
        acc = __builtin_elementwise_fma(f32x2{y[4 + c].x, y[4 + c].y}, __builtin_elementwise_fma(sc, f01, m8), acc);
        acc = __builtin_elementwise_fma(f32x2{y[4 + c].z, y[4 + c].w}, __builtin_elementwise_fma(sc, f23, m8), acc);
    }
    return acc;
}

template <int PRO, int R, int NB>
__global__ __launch_bounds__(512) void gemv_f32q4_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, nblk = K / QB;
    const ActF32 a = carve_f32(smem, nblk);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int ngroups = p.nrows / R;
    const int tw = gridDim.x * nwaves, gw = blockIdx.x * nwaves + wave;
    const int per = (ngroups + tw - 1) / tw;
    const int g0 = gw * per;
    int g1 = g0 + per;
    if (g1 > ngroups) g1 = ngroups;

    WBuf<R, (NB > 0 ? NB : 1)> cur, nxt;
    if constexpr (NB > 0) {
        if (g0 < g1) load_group<EPI_STORE, R, NB>(p, g0, lane, cur);
    }

    float fs = 1.0f;
    if (PRO == PRO_RMS_F32) fs = rms_factor(p.x, K, p.eps, a.red);
    for (int unit = threadIdx.x; unit < K / 8; unit += blockDim.x) {
        const int e0 = unit * 8;
        float4 xa = *(const float4*)(p.x + e0), xb = *(const float4*)(p.x + e0 + 4);
        if (PRO == PRO_RMS_F32) {
            float w[8];
            load8_norm(p.nw, e0, w);
            xa.x = w[0] * (fs * xa.x); xa.y = w[1] * (fs * xa.y); xa.z = w[2] * (fs * xa.z); xa.w = w[3] * (fs * xa.w);
            xb.x = w[4] * (fs * xb.x); xb.y = w[5] * (fs * xb.y); xb.z = w[6] * (fs * xb.z); xb.w = w[7] * (fs * xb.w);
        }
        const int blk = unit >> 2, c = (unit & 3) * 2;
        a.y[(size_t)c * nblk + blk] = xa;
        a.y[(size_t)(c + 1) * nblk + blk] = xb;
    }
    lds_barrier();

    float bestv = -INFINITY;
    int besti = 0x7fffffff;
    if constexpr (NB > 0) {
        float4 yr[NB][8];
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int c = 0; c < 8; c++) yr[i][c] = a.y[(size_t)c * nblk + lane + 64 * i];
        for (int g = g0; g < g1; g++) {
            if (g + 1 < g1) load_group<EPI_STORE, R, NB>(p, g + 1, lane, nxt);
            f32x2 acc2[R];
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; r++) acc2[r] = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < NB; i++)
#pragma unroll
                for (int r = 0; r < R; r++) acc2[r] = q4_block_dot_f32(cur.w[r][i], cur.s[r][i], yr[i], acc2[r]);
#pragma unroll
            for (int r = 0; r < R; r++) {
                acc[r] = wave_sum(acc2[r].x + acc2[r].y);
                if (acc[r] > bestv) { bestv = acc[r]; besti = g * R + r; }  // rows ascend: strict > keeps the first
            }
            store_group<EPI_STORE, R>(p, g, lane, acc);
            cur = nxt;
        }
    } else {
        for (int g = g0; g < g1; g++) {
            const uint8_t* wbase = p.w + (size_t)g * R * p.ldb;
            const float* sbase = p.ws + (size_t)g * R * p.ldbf;
            f32x2 acc2[R];
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; r++) acc2[r] = f32x2{0.0f, 0.0f};
            for (int blk = lane; blk < nblk; blk += 64) {
                float4 yb[8];
#pragma unroll
                for (int c = 0; c < 8; c++) yb[c] = a.y[(size_t)c * nblk + blk];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const i32x4 wv = __builtin_nontemporal_load((const i32x4*)(wbase + (size_t)r * p.ldb) + blk);
                    acc2[r] = q4_block_dot_f32(wv, sbase[(size_t)r * p.ldbf + blk], yb, acc2[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                acc[r] = wave_sum(acc2[r].x + acc2[r].y);
                if (acc[r] > bestv) { bestv = acc[r]; besti = g * R + r; }
            }
            store_group<EPI_STORE, R>(p, g, lane, acc);
        }
    }
    if (p.amax_part) {
        if (lane == 0) { a.bestv[wave] = bestv; a.besti[wave] = besti; }
        lds_barrier();
        if (threadIdx.x == 0) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int w = 0; w < nwaves; w++) {
                const float v = a.bestv[w];
                const int i = a.besti[w];
                if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
            }
            p.amax_part[blockIdx.x] = bv;
            p.amax_idx[blockIdx.x] = bi;
        }
    }
}

// ------------------------------------------------------------------------------------------------ decode state
// ------------------------------------------------------------------------------------------------ K3: GEMV BF16 weights
// Dense BF16 model (Mistral-7B, config 4) at M=1.  Reference arithmetic: activations RNE-rounded to BF16
// (PTO:1624-1628 -> FloatConversions.float32ToBFloat16), both operands widened by <<16, F32 fma accumulate
// (GemmerBF16 PTO:1279-1311); the LM head multiplies the UN-quantized F32 normed row (GemmerF32BF16 PTO:1511-1538,
// AbstractModel.java:443-449).  Decode is HBM-bound (2 B/weight): one 16-byte non-temporal load = 8 weights per
// lane; lane l owns 8-element chunks l, l+64, ...; the activation row sits in LDS as F32 in two float4 planes
// (conflict-free ds_read_b128).  MFMA is for the batched prefill GEMM, not for this kernel.
enum { PROB_RMS_BF16 = 0, PROB_QUANT_BF16 = 1, PROB_RMS_F32 = 2, PROB_F32 = 3 };
struct ActBF {
    float4* lo;    // [K/8] elements 0..3 of each 8-element chunk
    float4* hi;    // [K/8] elements 4..7
    double* red;   // [32]
    float* bestv;  // [16]
    int* besti;    // [16]
};
__device__ __forceinline__ ActBF carve_bf(char* smem, int K) {
    ActBF a;
    a.lo = (float4*)smem;
    a.hi = a.lo + K / 8;
    a.red = (double*)(a.hi + K / 8);
    a.bestv = (float*)(a.red + 32);
    a.besti = (int*)(a.bestv + 16);
    return a;
}
static inline size_t lds_bytes_bf(int K) { return (size_t)K * 4 + 32 * 8 + 16 * 4 + 16 * 4; }

__device__ __forceinline__ float chunk_dot_bf16(const i32x4& w, const float4& alo, const float4& ahi, float acc) {
    acc = fmaf(alo.x, __int_as_float(w.x << 16), acc);
    acc = fmaf(alo.y, __int_as_float(w.x & 0xffff0000), acc);
    acc = fmaf(alo.z, __int_as_float(w.y << 16), acc);
    acc = fmaf(alo.w, __int_as_float(w.y & 0xffff0000), acc);
    acc = fmaf(ahi.x, __int_as_float(w.z << 16), acc);
    acc = fmaf(ahi.y, __int_as_float(w.z & 0xffff0000), acc);
    acc = fmaf(ahi.z, __int_as_float(w.w << 16), acc);
    acc = fmaf(ahi.w, __int_as_float(w.w & 0xffff0000), acc);
    return acc;
}

template <int PRO, int EPI, int R, bool ARGMAX>
__global__ __launch_bounds__(512) void gemv_bf16_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, nch = K / 8;
    const ActBF a = carve_bf(smem, K);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int T = blockDim.x;
    // ---- prologue: x (+ norm weights) in one round trip, branch-free
    constexpr int UM = 2;
    float xv[UM][8], wv[UM][8];
#pragma unroll
    for (int u = 0; u < UM; u++) {
        int unit = threadIdx.x + u * T;
        unit = unit < nch ? unit : nch - 1;
        const float4 xa = *(const float4*)(p.x + unit * 8), xb = *(const float4*)(p.x + unit * 8 + 4);
        xv[u][0] = xa.x; xv[u][1] = xa.y; xv[u][2] = xa.z; xv[u][3] = xa.w;
        xv[u][4] = xb.x; xv[u][5] = xb.y; xv[u][6] = xb.z; xv[u][7] = xb.w;
        if (PRO == PROB_RMS_BF16 || PRO == PROB_RMS_F32) load8_norm(p.nw, unit * 8, wv[u]);
    }
    float fs = 1.0f;
    if (PRO == PROB_RMS_BF16 || PRO == PROB_RMS_F32) {
        double ss = 0.0;
#pragma unroll
        for (int u = 0; u < UM; u++)
            if (threadIdx.x + u * T < nch)
#pragma unroll
                for (int i = 0; i < 8; i++) ss += (double)(xv[u][i] * xv[u][i]);
        for (int unit = threadIdx.x + UM * T; unit < nch; unit += T) {
            const float4 xa = *(const float4*)(p.x + unit * 8), xb = *(const float4*)(p.x + unit * 8 + 4);
            ss += (double)(xa.x * xa.x); ss += (double)(xa.y * xa.y); ss += (double)(xa.z * xa.z); ss += (double)(xa.w * xa.w);
            ss += (double)(xb.x * xb.x); ss += (double)(xb.y * xb.y); ss += (double)(xb.z * xb.z); ss += (double)(xb.w * xb.w);
        }
        ss = block_sum_d(ss, a.red);
        ss /= (double)K;
        ss += (double)p.eps;
        ss = 1.0 / sqrt(ss);
        fs = (float)ss;
    }
    auto finish_unit = [&](int unit, float (&y)[8]) {
        if (PRO == PROB_RMS_BF16 || PRO == PROB_QUANT_BF16) {
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = bf16_to_f32(f32_to_bf16(y[i]));   // quantizeBF16: RNE, then widened again
        }
        a.lo[unit] = make_float4(y[0], y[1], y[2], y[3]);
        a.hi[unit] = make_float4(y[4], y[5], y[6], y[7]);
    };
#pragma unroll
    for (int u = 0; u < UM; u++) {
        const int unit = threadIdx.x + u * T;
        if (unit < nch) {
            float y[8];
#pragma unroll
            for (int i = 0; i < 8; i++)
                y[i] = (PRO == PROB_RMS_BF16 || PRO == PROB_RMS_F32) ? wv[u][i] * (fs * xv[u][i]) : xv[u][i];
            finish_unit(unit, y);
        }
    }
    for (int unit = threadIdx.x + UM * T; unit < nch; unit += T) {
        const float4 xa = *(const float4*)(p.x + unit * 8), xb = *(const float4*)(p.x + unit * 8 + 4);
        float y[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
        if (PRO == PROB_RMS_BF16 || PRO == PROB_RMS_F32) {
            float w[8];
            load8_norm(p.nw, unit * 8, w);
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = w[i] * (fs * y[i]);
        }
        finish_unit(unit, y);
    }
    lds_barrier();

    const int total = (EPI == EPI_SILU_MUL) ? 2 * p.nrows : p.nrows;
    const int ngroups = total / R;
    float bestv = -INFINITY;
    int besti = 0x7fffffff;
    for (int g = blockIdx.x * nwaves + wave; g < ngroups; g += gridDim.x * nwaves) {
        const uint8_t* rowp[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            if (EPI == EPI_SILU_MUL) rowp[r] = (r < R / 2 ? p.w : p.w2) + (size_t)(g * (R / 2) + r % (R / 2)) * p.ldb;
            else rowp[r] = p.w + (size_t)(g * R + r) * p.ldb;
        }
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = 0.0f;
#pragma unroll 4
        for (int c = lane; c < nch; c += 64) {
            const float4 alo = a.lo[c], ahi = a.hi[c];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const i32x4 wv4 = __builtin_nontemporal_load((const i32x4*)rowp[r] + c);
                acc[r] = chunk_dot_bf16(wv4, alo, ahi, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            acc[r] = wave_sum(acc[r]);
            if (ARGMAX && acc[r] > bestv) { bestv = acc[r]; besti = g * R + r; }
        }
        store_group<EPI, R>(p, g, lane, acc);
    }
    if (ARGMAX && p.amax_part) {
        if (lane == 0) { a.bestv[wave] = bestv; a.besti[wave] = besti; }
        lds_barrier();
        if (threadIdx.x == 0) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int w = 0; w < nwaves; w++) {
                const float v = a.bestv[w];
                const int i = a.besti[w];
                if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
            }
            p.amax_part[blockIdx.x] = bv;
            p.amax_idx[blockIdx.x] = bi;
        }
    }
}

// ------------------------------------------------------------------------------------------------ K3m: batched BF16 GEMM on MFMA
// batchDotProduct BF16 x BF16 -> F32 for M > 1 (prefill of a dense BF16 model, GemmerBF16 PTO:1233-1311):
//   C[i, j] = sum_k A[i,k] * W[j,k],  A = BF16 activations [M, lda], W = BF16 weights [N, ldb] (row = output).
// This is the one place on the path where the matrix cores apply (M=129 => ~129 flop/B).
// v_mfma_f32_32x32x16_bf16: lane l holds A[m = l&31][k = (l>>5)*8 .. +7] and W[n = l&31][same k]  (8 bf16 = 16 B each);
// D: col n = l&31, row m = (r&3) + 8*(r>>2) + 4*(l>>5) for accumulator register r (cdna_hip_programming.md §3).
// Tiling: a wave owns 32 weight rows (one MFMA column tile) and ALL of M (MT <= 8 row tiles => every weight byte is
// read from HBM exactly once, straight into registers: per 64-wide K slice lane (n, h) loads the 4 chunks
// k = s*16 + h*8 (s = 0..3), i.e. lanes h=0/1 of a row cover its 128 contiguous bytes).  The A slice [M, 64] is staged
// through LDS once per workgroup (row stride 144 B: conflict-free ds_read_b128 for the 16-lane groups), double
// buffered; products are exact in F32, accumulation order differs from Panama's 16-lane order (tolerance, not bits).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct MfmaGemmParams {
    const uint16_t* a; const uint16_t* w; float* c;
    int m, n0, n, k, lda, ldb, ldc, roffset;   // same meaning as gemm_bf16 (nc/simd/vector_simd.h:34), offsets pre-applied
    const float* resid;                        // optional: C += resid (same layout as C), the residual stream in prefill
    float* ws; int nsplit;                     // split-K: workgroup row blockIdx.y handles K slices [y*k/nsplit, ...) and
                                               // writes its partial [m][n] (ld = n) to ws + y*m*n; splitk_reduce_kernel sums
};
constexpr int MG_KS = 64;            // K slice per stage
constexpr int MG_ASTRIDE = 144;      // bytes per A row in LDS (128 + 16 pad)

template <int MT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemm_bf16_mfma_kernel(MfmaGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, h = lane >> 5;
    const int ntile = blockIdx.x * WAVES + wave;          // 32-column tile of this wave
    const int ncol0 = p.n0 + ntile * 32;
    const bool active = ntile * 32 < p.n;
    constexpr int MROWS = MT * 32;
    constexpr int ABYTES = MROWS * MG_ASTRIDE;   // one A stage; stage b lives at smem + b*ABYTES
    const int nsplit = p.nsplit > 1 ? p.nsplit : 1;
    const int nslices = p.k / MG_KS / nsplit, slice0 = blockIdx.y * nslices;

    // cooperative A slice load: chunk = (row, c8) with 8 x 16-byte chunks per row
    constexpr int CHUNKS = MROWS * 8, NT = WAVES * 64, CPT = (CHUNKS + NT - 1) / NT;
    i32x4 areg[CPT];
    auto load_a = [&](int slice) {
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            int ch = tid + i * NT;
            ch = ch < CHUNKS ? ch : CHUNKS - 1;
            int row = ch >> 3;
            const int c8 = ch & 7;
            row = row < p.m ? row : p.m - 1;                 // rows beyond M replicate the last row (never stored)
            areg[i] = *(const i32x4*)(p.a + (size_t)row * p.lda + (size_t)(slice0 + slice) * MG_KS + c8 * 8);
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int ch = tid + i * NT;
            if (ch < CHUNKS) *(i32x4*)(smem + buf * ABYTES + (ch >> 3) * MG_ASTRIDE + (ch & 7) * 16) = areg[i];
        }
    };
    const uint16_t* wrow = p.w + (size_t)(active ? (ncol0 + nl) : p.n0) * p.ldb;
    i32x4 wreg[2][4];
    auto load_w = [&](int slice, int buf) {
#pragma unroll
        for (int s = 0; s < 4; s++)
            wreg[buf][s] = __builtin_nontemporal_load((const i32x4*)(wrow + (size_t)(slice0 + slice) * MG_KS + s * 16 + h * 8));
    };

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;

    load_a(0);
    load_w(0, 0);
    store_a(0);
    __syncthreads();
    for (int sl = 0; sl < nslices; sl++) {
        const int cur = sl & 1;
        if (sl + 1 < nslices) {
            load_a(sl + 1);
            load_w(sl + 1, cur ^ 1);
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, wreg[cur][s]);
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const i32x4 av = *(const i32x4*)(smem + cur * ABYTES + (t * 32 + nl) * MG_ASTRIDE + s * 32 + h * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), bfrag, acc[t], 0, 0, 0);
            }
        }
        if (sl + 1 < nslices) store_a(cur ^ 1);
        __syncthreads();
    }
    if (!active) return;
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int mrow = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mrow < p.m) {
                if (nsplit > 1) {
                    p.ws[((size_t)blockIdx.y * p.m + mrow) * p.n + (ncol0 - p.n0 + nl)] = acc[t][r];
                } else {
                    const size_t idx = (size_t)p.ldc * mrow + (ncol0 + nl) - p.roffset;
                    p.c[idx] = p.resid ? acc[t][r] + p.resid[idx] : acc[t][r];
                }
            }
        }
}
// second pass of a split-K GEMM: C[i][j] = sum_s ws[s][i][j] (ascending K ranges => deterministic) (+ resid)
static __global__ void splitk_reduce_kernel(const float* ws, int nsplit, int m, int n, int n0, float* c, int ldc, int roffset, const float* resid) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)m * n) return;
    const int row = (int)(i / n), col = (int)(i % n);
    float v = 0.0f;
    for (int s = 0; s < nsplit; s++) v += ws[(size_t)s * m * n + i];
    const size_t idx = (size_t)ldc * row + (n0 + col) - roffset;
    c[idx] = resid ? v + resid[idx] : v;
}
// the same sums, four columns per thread (n, ldc, n0 - roffset multiples of 4; 16-byte aligned buffers): grid (n/4 / 256, m)
static __global__ __launch_bounds__(256) void splitk_reduce4_kernel(const f32x4* ws, int nsplit, int m, int n4, f32x4* c, int ldc4, int coff4, const f32x4* resid) {
    const int col = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (col >= n4) return;
    const size_t i = (size_t)row * n4 + col, stride = (size_t)m * n4;
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s = 0; s < nsplit; s++) v = v + ws[s * stride + i];
    const size_t idx = (size_t)ldc4 * row + coff4 + col;
    c[idx] = resid ? v + resid[idx] : v;
}

// ---- K3t: BF16 GEMM on MFMA-ordered operands (the prefill path of a BF16 model).
// Both operands are tiled so that one wave-load is one MFMA operand, 1 KB contiguous:
//   A  [row tile][k slice of 16][h][m][8 bf16]     written by rows_bf16_kernel (ldq < 0)
//   W  [col tile][k slice of 16][h][n][8 bf16]     re-tiled copy (retile16_kernel)
// A wave owns one column tile and ALL row tiles (MT accumulator tiles): every weight byte is fetched once, straight into
// registers; per k slice it loads one W fragment and MT A fragments (no LDS, no barrier).  The CWB waves of a
// workgroup take adjacent column tiles and walk K in step, so an A fragment missed by one wave is an L1 hit for the
// others.  K can be split over workgroup rows (partials to the workspace, splitk_reduce_kernel).
struct MfmaBf16TileParams {
    const uint16_t* a; const uint16_t* w; float* c; const float* resid;
    int m, n, k, ldc;
    float* ws; int nsplit;
};
template <int MT, int CWB>
__global__ __launch_bounds__(CWB * 64) void gemm_bf16_tile_kernel(MfmaBf16TileParams p) {
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, h = lane >> 5;
    const int ct = blockIdx.x * CWB + wv;
    if (ct * 32 >= p.n) return;
    const int nks = p.k / 16, nsplit = p.nsplit > 1 ? p.nsplit : 1;
    const int per = nks / nsplit, s0 = blockIdx.y * per;
    const i32x4* wp = (const i32x4*)p.w + ((size_t)ct * nks + s0) * 64 + lane;
    const i32x4* ap = (const i32x4*)p.a + (size_t)s0 * 64 + lane;          // + (rt*nks + s)*64
    const size_t a_rt = (size_t)nks * 64;
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    // two register sets of one k slice each (W fragment + MT A fragments), ping-pong: the next slice's 1+MT loads are
    // issued before the current slice's MT MFMAs
    i32x4 w0, w1, a0[MT], a1[MT];
    auto load_slice = [&](int s, i32x4& wv, i32x4 (&av)[MT]) __attribute__((always_inline)) {
        s = s < per ? s : per - 1;                                           // clamped: the last prefetch reloads
        wv = __builtin_nontemporal_load(wp + (size_t)s * 64);
#pragma unroll
        for (int t = 0; t < MT; t++) av[t] = ap[(size_t)t * a_rt + (size_t)s * 64];
    };
    auto mma_slice = [&](const i32x4& wv, const i32x4 (&av)[MT]) __attribute__((always_inline)) {
        const bf16x8 bfrag = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[t]), bfrag, acc[t], 0, 0, 0);
    };
    load_slice(0, w0, a0);
    for (int s = 0; s < per; s += 2) {   // host guarantees per % 2 == 0
        load_slice(s + 1, w1, a1);
        mma_slice(w0, a0);
        load_slice(s + 2, w0, a0);
        mma_slice(w1, a1);
    }
    const int ncol = ct * 32 + nl;
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int mrow = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mrow < p.m) {
                if (nsplit > 1) {
                    p.ws[((size_t)blockIdx.y * p.m + mrow) * p.n + ncol] = acc[t][r];
                } else {
                    const size_t idx = (size_t)p.ldc * mrow + ncol;
                    p.c[idx] = p.resid ? acc[t][r] + p.resid[idx] : acc[t][r];
                }
            }
        }
}
// The same GEMM with the A fragments through LDS.  In gemm_bf16_tile_kernel every wave fetches its own MT A fragments per k
// slice ((1 + MT) KB through the CU's texture path per wave and slice, L1 hits for all but the first wave): 5.5 MB per CU for the
// gate|up GEMM at M = 129 = 39 us at 64 B/clk, the bound of that kernel.  Here the CWB waves of a workgroup fetch a chunk of
// SLC k slices of A once (MT*SLC 1-KB pieces, spread over the waves, registers -> LDS, double buffered, one barrier per chunk)
// and read the fragments back with ds_read_b128; only the weight fragment (1 KB per wave and slice, used MT times) still comes
// through the texture path.  Same MFMA order, same results.
template <int MT, int CWB>
__global__ __launch_bounds__(CWB * 64) void gemm_bf16_lds_kernel(MfmaBf16TileParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLC = 4;                                   // k slices per chunk
    constexpr int PIECES = MT * SLC, APW = (PIECES + CWB - 1) / CWB;
    i32x4* ring = (i32x4*)smem;                              // [2][MT][SLC][64] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, h = lane >> 5;
    const int ct = blockIdx.x * CWB + wv;                    // host: n % (32*CWB) == 0
    const int nks = p.k / 16, nsplit = p.nsplit > 1 ? p.nsplit : 1;
    const int per = nks / nsplit, s0 = blockIdx.y * per;     // host: per % (2*SLC) == 0
    const i32x4* wp = (const i32x4*)p.w + ((size_t)ct * nks + s0) * 64 + lane;
    const i32x4* ap = (const i32x4*)p.a + (size_t)s0 * 64 + lane;          // + (rt*nks + s)*64
    const size_t a_rt = (size_t)nks * 64;
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    i32x4 ast[APW], w0[SLC], w1[SLC];
    const int last = per - 1;
    auto load_a = [&](int sl0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < APW; i++) {
            int pc = wv + i * CWB;
            pc = pc < PIECES ? pc : PIECES - 1;
            const int t = pc / SLC, q = pc - t * SLC;
            int sl = sl0 + q;
            sl = sl < last ? sl : last;
            ast[i] = ap[(size_t)t * a_rt + (size_t)sl * 64];
        }
    };
    auto store_a = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < APW; i++) {
            const int pc = wv + i * CWB;
            if (pc < PIECES) ring[((size_t)buf * PIECES + pc) * 64 + lane] = ast[i];
        }
    };
    auto load_w = [&](i32x4 (&wr)[SLC], int sl0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < SLC; q++) {
            int sl = sl0 + q;
            sl = sl < last ? sl : last;
            wr[q] = __builtin_nontemporal_load(wp + (size_t)sl * 64);
        }
    };
    auto mma_chunk = [&](int buf, const i32x4 (&wr)[SLC]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < SLC; q++) {
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, wr[q]);
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const i32x4 av = ring[((size_t)buf * PIECES + t * SLC + q) * 64 + lane];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), bfrag, acc[t], 0, 0, 0);
            }
        }
    };
    load_a(0);
    load_w(w0, 0);
    store_a(0);
    lds_barrier();
    for (int s = 0; s < per; s += 2 * SLC) {
        load_a(s + SLC);
        load_w(w1, s + SLC);
        mma_chunk(0, w0);
        store_a(1);
        lds_barrier();
        load_a(s + 2 * SLC);
        load_w(w0, s + 2 * SLC);
        mma_chunk(1, w1);
        store_a(0);
        lds_barrier();
    }
    const int ncol = ct * 32 + nl;
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int mrow = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mrow < p.m) {
                if (nsplit > 1) {
                    p.ws[((size_t)blockIdx.y * p.m + mrow) * p.n + ncol] = acc[t][r];
                } else {
                    const size_t idx = (size_t)p.ldc * mrow + ncol;
                    p.c[idx] = p.resid ? acc[t][r] + p.resid[idx] : acc[t][r];
                }
            }
        }
}
// Round 5: the same GEMM with TWO column tiles per MFMA wave and a LOADER wave.
// What bounded gemm_bf16_lds_kernel at M = 129 (tools/gemm_bf16_lab.hip, profiles/r04h_*): every MFMA read its A fragment from LDS
// (1 KiB per 32-cycle instruction and SIMD = the LDS's 128 B/clk, so the matrix pipe idled at 58 cycles per MFMA even with no
// global load at all), and the MFMA waves staged A themselves, so their waits for the young A loads (L2) were waits for every older
// weight load (HBM) -- vector loads retire in order.  Here
//   * an MFMA wave owns 2 adjacent column tiles: an A fragment read from LDS feeds 2 MFMAs (2 x MT accumulator tiles; 64 B/clk);
//   * a third wave does nothing but stage A (global -> registers one chunk ahead -> LDS double buffer, one barrier per chunk), so the
//     MFMA waves request WEIGHTS only, PW chunks of 4 k slices ahead (PW x 8 KiB per wave in flight);
//   * 2 MFMA waves per workgroup = 128 columns: the un-split gate|up GEMM of an 8B-class model is 224 workgroups -- a workgroup per
//     CU with NO split-K reduce pass; the smaller GEMMs split K until the chip is covered (partials + splitk_reduce4_kernel).
// MFMA order inside an output element is the k-slice order, as before: results equal gemm_bf16_lds_kernel's bit for bit per split.
constexpr int BF16_CW2_WAVES = 3;                            // 2 MFMA waves + the loader
template <int MT, int PW, int KNOCK = 0>                     // KNOCK (measurement only, wrong results): 1 = the loader re-files chunk 0 (no A traffic), 2 = weights loaded once
__global__ __launch_bounds__(BF16_CW2_WAVES * 64) void gemm_bf16_cw2_kernel(MfmaBf16TileParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLC = 4, PIECES = MT * SLC, NW = 2;
    i32x4* ring = (i32x4*)smem;                              // [2][MT][SLC][64] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nks = p.k / 16, nsplit = p.nsplit > 1 ? p.nsplit : 1;
    const int per = nks / nsplit, s0 = blockIdx.y * per;     // host: per % (PW * SLC) == 0
    const int nchunks = per / SLC;
    if (wv == NW) {
        // ---- loader: A chunk c+2 requested while chunk c+1 is filed and chunk c is multiplied
        const i32x4* ap = (const i32x4*)p.a + (size_t)s0 * 64 + lane;
        const size_t a_rt = (size_t)nks * 64;
        i32x4 r0[PIECES], r1[PIECES];
        auto request = [&](i32x4 (&r)[PIECES], int c) __attribute__((always_inline)) {
            c = c < nchunks ? c : nchunks - 1;
            if (KNOCK == 1 && c > 1) return;
#pragma unroll
            for (int pc = 0; pc < PIECES; pc++) {
                const int t = pc / SLC, q = pc - t * SLC;
                r[pc] = ap[(size_t)t * a_rt + (size_t)(c * SLC + q) * 64];
            }
        };
        auto file = [&](const i32x4 (&r)[PIECES], int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int pc = 0; pc < PIECES; pc++) ring[((size_t)buf * PIECES + pc) * 64 + lane] = r[pc];
        };
        request(r0, 0);
        request(r1, 1);
        file(r0, 0);
        lds_barrier();
        for (int c = 0; c < nchunks; c += 2) {
            request(r0, c + 2);
            file(r1, 1);
            lds_barrier();
            request(r1, c + 3);
            file(r0, 0);
            lds_barrier();
        }
        return;
    }
    const int nl = lane & 31, h = lane >> 5;
    const int ct0 = (blockIdx.x * NW + wv) * 2;              // host: n % 128 == 0
    const i32x4* wp0 = (const i32x4*)p.w + ((size_t)ct0 * nks + s0) * 64 + lane;
    const i32x4* wp1 = wp0 + (size_t)nks * 64;
    f32x16 acc0[MT], acc1[MT];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[t][r] = 0.0f; acc1[t][r] = 0.0f; }
    const int last = per - 1;
    i32x4 wr0[PW][SLC], wr1[PW][SLC];
    auto load_w = [&](i32x4 (&w0)[SLC], i32x4 (&w1)[SLC], int sl0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < SLC; q++) {
            int sl = sl0 + q;
            sl = sl < last ? sl : last;
            if (KNOCK == 2 && sl0 >= PW * SLC) continue;
            w0[q] = __builtin_nontemporal_load(wp0 + (size_t)sl * 64);
            w1[q] = __builtin_nontemporal_load(wp1 + (size_t)sl * 64);
        }
    };
    auto read_a = [&](i32x4 (&av)[MT], int buf, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < MT; t++) av[t] = ring[((size_t)buf * PIECES + t * SLC + q) * 64 + lane];
    };
    auto mma_slice = [&](const i32x4 (&av)[MT], const i32x4& w0, const i32x4& w1) __attribute__((always_inline)) {
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, w0), b1 = __builtin_bit_cast(bf16x8, w1);
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, av[t]);
            acc0[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc0[t], 0, 0, 0);
            acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc1[t], 0, 0, 0);
        }
    };
    auto mma_chunk = [&](int buf, const i32x4 (&w0)[SLC], const i32x4 (&w1)[SLC]) __attribute__((always_inline)) {
        i32x4 a0[MT], a1[MT];                                // A fragments are read a slice ahead of their MFMAs
        read_a(a0, buf, 0);
#pragma unroll
        for (int q = 0; q < SLC; q += 2) {
            read_a(a1, buf, q + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_slice(a0, w0[q], w1[q]);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 2 < SLC) read_a(a0, buf, q + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma_slice(a1, w0[q + 1], w1[q + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int i = 0; i < PW; i++) load_w(wr0[i], wr1[i], i * SLC);
    lds_barrier();
    for (int c = 0; c < nchunks; c += PW) {
#pragma unroll
        for (int i = 0; i < PW; i++) {
            mma_chunk(i & 1, wr0[i], wr1[i]);                // PW is even: chunk c + i sits in buffer i & 1
            load_w(wr0[i], wr1[i], (c + i + PW) * SLC);
            lds_barrier();
        }
    }
#pragma unroll
    for (int cc = 0; cc < 2; cc++) {
        const int ncol = (ct0 + cc) * 32 + nl;
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int mrow = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = cc ? acc1[t][r] : acc0[t][r];
                if (mrow < p.m) {
                    if (nsplit > 1) {
                        p.ws[((size_t)blockIdx.y * p.m + mrow) * p.n + ncol] = v;
                    } else {
                        const size_t idx = (size_t)p.ldc * mrow + ncol;
                        p.c[idx] = p.resid ? v + p.resid[idx] : v;
                    }
                }
            }
    }
}
// Round 6: the same GEMM with EIGHT MFMA waves per workgroup.  What bounded gemm_bf16_cw2_kernel at M = 129
// was not the memory system (tools/chain_lab.hip: a CU fills 85-105 GB/s from an L2-resident buffer with 4-8 waves, not the ~30 GB/s
// DESIGN assumed for the activation re-reads): its two MFMA waves sat on two of the CU's four SIMDs and issued a 32x32x16 every
// 56-64 cycles each -- an issue-rate ceiling of ~4 TB/s of weights, reached at half that.  Here
//   * 8 MFMA waves = 4 column tiles (128 columns) x 2 halves of the workgroup's K range: two waves per SIMD, every matrix pipe
//     busy, 8 x PW x 4 KiB of weights in flight per CU; each wave keeps MT accumulator tiles (all prompt rows) of ITS half;
//   * a ninth wave stages the A chunks of BOTH halves with LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write; 1 KiB per
//     instruction straight into the double-buffered ring), one barrier per chunk -- the MFMA waves request weights only;
//   * the halves meet in LDS at the end (the ring's space: 16 KiB per row tile), half 0 + half 1 in that order, then the store
//     (or the split-K partial when the grid also splits K: few-column GEMMs need it to cover the chip).
constexpr int BF16_W8_WAVES = 10;                            // 8 MFMA waves + one LDS-DMA loader per K half
constexpr int BF16_W8_NBUF = 3;                              // A ring: chunk c is multiplied while c+1 has landed and c+2 is in flight
template <int MT>
__global__ __launch_bounds__(BF16_W8_WAVES * 64) void gemm_bf16_w8_kernel(MfmaBf16TileParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int SLC = 4, PIECES = MT * SLC, NC = 4, PW = 2, NBUF = BF16_W8_NBUF;
    i32x4* ring = (i32x4*)smem;                              // [NBUF][2 halves][MT][SLC][64] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nks = p.k / 16, nsplit = p.nsplit > 1 ? p.nsplit : 1;
    const int per = nks / nsplit, s0 = blockIdx.y * per, half = per / 2;   // host: half % (SLC * PW) == 0
    const int nchunks = half / SLC;
    if (wv >= 2 * NC) {
        // ---- loader of K half lh: a lone wave is latency-bound (~25 GB/s from L2 with one fill outstanding, tools/chain_lab.hip), so it
        // keeps TWO fills in flight -- chunk c+2 is requested while chunk c+1 lands and chunk c is multiplied
        const int lh = wv - 2 * NC;
        const i32x4* ap = (const i32x4*)p.a + (size_t)(s0 + lh * half) * 64 + lane;
        const size_t a_rt = (size_t)nks * 64;
        auto fill = [&](int c) __attribute__((always_inline)) {
            const int buf = c % NBUF;
#pragma unroll
            for (int pc = 0; pc < PIECES; pc++) {
                const int t = pc / SLC, q = pc - t * SLC;
                const i32x4* g = ap + (size_t)t * a_rt + (size_t)(c * SLC + q) * 64;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(ring + ((size_t)(buf * 2 + lh) * PIECES + pc) * 64), 16, 0, 0);
            }
        };
        fill(0);
        if (nchunks > 1) { fill(1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // chunk 0 staged
        for (int c = 0; c < nchunks; c++) {
            if (c + 2 < nchunks) { fill(c + 2); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory"); }   // chunk c+1 has landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();                        // the halves' meeting
        return;
    }
    const int ct = wv & (NC - 1), kh = wv >> 2;
    const int nl = lane & 31, h = lane >> 5;
    const int ctg = blockIdx.x * NC + ct;                    // host: n % 128 == 0
    const i32x4* wp = (const i32x4*)p.w + ((size_t)ctg * nks + s0 + (size_t)kh * half) * 64 + lane;
    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    const int last = half - 1;
    i32x4 wr[PW][SLC];
    auto load_w = [&](i32x4 (&w)[SLC], int sl0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < SLC; q++) {
            int sl = sl0 + q;
            sl = sl < last ? sl : last;                      // (the prefetch past the end reloads the last slice)
            w[q] = __builtin_nontemporal_load(wp + (size_t)sl * 64);
        }
    };
    auto mma_chunk = [&](int buf, const i32x4 (&w)[SLC]) __attribute__((always_inline)) {
        const i32x4* base = ring + ((size_t)(buf * 2 + kh) * PIECES) * 64 + lane;
        i32x4 a0[MT], a1[MT];                                // A fragments are read a slice ahead of their MFMAs
#pragma unroll
        for (int t = 0; t < MT; t++) a0[t] = base[(size_t)(t * SLC) * 64];
#pragma unroll
        for (int q = 0; q < SLC; q += 2) {
#pragma unroll
            for (int t = 0; t < MT; t++) a1[t] = base[(size_t)(t * SLC + q + 1) * 64];
            __builtin_amdgcn_sched_barrier(0);
            {
                const bf16x8 b = __builtin_bit_cast(bf16x8, w[q]);
#pragma unroll
                for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[t]), b, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q + 2 < SLC) {
#pragma unroll
                for (int t = 0; t < MT; t++) a0[t] = base[(size_t)(t * SLC + q + 2) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const bf16x8 b = __builtin_bit_cast(bf16x8, w[q + 1]);
#pragma unroll
                for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1[t]), b, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int i = 0; i < PW; i++) load_w(wr[i], i * SLC);
    lds_barrier();                                           // chunk 0 staged
    for (int c = 0; c < nchunks; c += PW) {
#pragma unroll
        for (int i = 0; i < PW; i++) {
            mma_chunk((c + i) % NBUF, wr[i]);
            load_w(wr[i], (c + i + PW) * SLC);
            lds_barrier();
        }
    }
    // ---- half 1 hands its accumulators to half 0 through LDS (the ring is free: the last barrier closed the last chunk)
    f32x4* red = (f32x4*)smem;                               // [NC][MT][4][64] x 16 B = MT x 16 KiB
    if (kh == 1) {
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++)
                red[((size_t)(ct * MT + t) * 4 + r4) * 64 + lane] = f32x4{acc[t][4 * r4], acc[t][4 * r4 + 1], acc[t][4 * r4 + 2], acc[t][4 * r4 + 3]};
    }
    lds_barrier();
    if (kh == 1) return;
    const int ncol = ctg * 32 + nl;
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r4 = 0; r4 < 4; r4++) {
            const f32x4 o = red[((size_t)(ct * MT + t) * 4 + r4) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = 4 * r4 + i;
                const int mrow = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = acc[t][r] + o[i];            // half 0 + half 1
                if (mrow < p.m) {
                    if (nsplit > 1) {
                        p.ws[((size_t)blockIdx.y * p.m + mrow) * p.n + ncol] = v;
                    } else {
                        const size_t idx = (size_t)p.ldc * mrow + ncol;
                        p.c[idx] = p.resid ? v + p.resid[idx] : v;
                    }
                }
            }
        }
}
// ------------------------------------------------------------------------------------------------ K2: batched I8 x Q4 GEMM on MFMA
// batchDotProduct I8 x Q4 -> F32 for M > 1 (prefill of a JQ4 model; GemmerI8Q4_512 2x2 tile PTO:958-1043, C twin
// nc/simd/vector_simd.c:261-437):  C[i,j] = sum_blk (da[i,blk]*sb[j,blk]) * sum_t a[i,blk,t]*(nib[j,blk,t]-8).
// v_mfma_i32_32x32x32_i8 has K = 32 = exactly one Q block, so the block sums stay EXACT integers:
//   A operand  lane (m, h): a[m][blk*32 + h*16 .. +15]           (16 int8)
//   B operand  lane (n, h): nibbles of W block (n, blk): h = 0 -> low nibbles (elements 0..15), h = 1 -> high nibbles
//              (elements 16..31)  -- Q4ByteBufferTensor.java:88-106 -- kept unsigned 0..15; the -8 bias is removed as
//              isum = mfma - 8*sum(a[m,blk]) (exact).
// then acc = fma(da*sb, (float)isum, acc) per block in ascending K order, like the reference.  A wave owns 32 weight
// rows and all of M (weights read once from HBM); the A slice [M,128] (4 blocks) goes through LDS with its block
// scales and 8*sum(a) per (row, block).
typedef int i32x16 __attribute__((ext_vector_type(16)));
struct MfmaQ4Params {
    const int8_t* a; const float* af; const uint8_t* w; const float* ws; float* c;
    const float* resid;    // optional: C += resid (same layout as C), e.g. the residual stream in prefill
    int m, n0, n, k, lda, ldaf, ldb, ldbf, ldc, roffset;   // gemm_q8_q4 meanings (vector_simd.h:22), offsets pre-applied
};
constexpr int MQ_ASTRIDE = 48;   // bytes per A row in LDS: 32 int8 + 16 pad (conflict-free ds_read_b128)

// Workgroup = one 32-column tile x ALL of M; its KSPLIT waves each own a contiguous range of Q blocks (split-K) and run
// independent, barrier-free pipelines: one stage = ONE Q block (A slice [M,32] + its scales + 8*sum(a) staged in the
// wave's private LDS region, LDS operations of a wave are ordered), weights straight from HBM to registers.  The small
// per-stage body also keeps hipcc from hoisting a whole K slice of MFMAs / scale reads and spilling.  Partial
// accumulators meet in LDS at the end.
template <int MT, int KSPLIT>
__global__ __launch_bounds__(KSPLIT * 64) void gemm_q8q4_mfma_kernel(MfmaQ4Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, ks = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, h = lane >> 5;
    const int ncol0 = p.n0 + blockIdx.x * 32;
    constexpr int MROWS = MT * 32;
    constexpr int ABYTES = MROWS * MQ_ASTRIDE;
    constexpr int SBYTES = MROWS * 4;
    constexpr int STAGE = ABYTES + 2 * SBYTES;            // A | dA | s8
    char* const my = smem + ks * 2 * STAGE;               // this wave's two stages
    const int nblk = p.k / QB;
    const int b0 = (int)((long long)nblk * ks / KSPLIT), b1 = (int)((long long)nblk * (ks + 1) / KSPLIT);
    constexpr int CHUNKS = MROWS * 2, CPT = CHUNKS / 64;  // 2 x 16-byte chunks per row, 64 lanes
    i32x4 areg[CPT];
    float dreg[CPT];
    auto load_a = [&](int blk) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int ch = lane + i * 64;
            int row = ch >> 1;
            row = row < p.m ? row : p.m - 1;               // rows beyond M replicate the last row (never stored)
            areg[i] = *(const i32x4*)(p.a + (size_t)row * p.lda + blk * QB + (ch & 1) * 16);
            dreg[i] = p.af[(size_t)row * p.ldaf + blk];
        }
    };
    auto store_a = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int ch = lane + i * 64;
            int s = 0;
            s = sdot4(areg[i].x, 0x01010101, s); s = sdot4(areg[i].y, 0x01010101, s);
            s = sdot4(areg[i].z, 0x01010101, s); s = sdot4(areg[i].w, 0x01010101, s);
            s += __builtin_amdgcn_update_dpp(0, s, 0xB1, 0xf, 0xf, false);   // + the other half block (adjacent lane)
            *(i32x4*)(my + buf * STAGE + (ch >> 1) * MQ_ASTRIDE + (ch & 1) * 16) = areg[i];
            if ((ch & 1) == 0) {
                ((float*)(my + buf * STAGE + ABYTES))[ch >> 1] = dreg[i];
                ((int*)(my + buf * STAGE + ABYTES + SBYTES))[ch >> 1] = 8 * s;
            }
        }
    };
    const uint8_t* wrow = p.w + (size_t)(ncol0 + nl) * p.ldb;
    const float* srow = p.ws + (size_t)(ncol0 + nl) * p.ldbf;

    float acc[MT][16];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    int zk = 0;

    if (b0 < b1) {
        load_a(b0);
        i32x4 wcur = __builtin_nontemporal_load((const i32x4*)(wrow + (size_t)b0 * 16));
        float scur = srow[b0];
        store_a(0);
        for (int blk = b0; blk < b1; blk++) {
            const int cur = (blk - b0) & 1;
            i32x4 wnext = wcur;
            float snext = scur;
            if (blk + 1 < b1) {
                load_a(blk + 1);
                wnext = __builtin_nontemporal_load((const i32x4*)(wrow + (size_t)(blk + 1) * 16));
                snext = srow[blk + 1];
            }
            i32x4 bw = wcur;
            if (h) bw = (bw >> 4);
            bw = bw & 0x0F0F0F0F;
            const char* stg = my + cur * STAGE;
            const float* dAp = (const float*)(stg + ABYTES);
            const int* s8p = (const int*)(stg + ABYTES + SBYTES);
#pragma unroll
            for (int t = 0; t < MT; t++) {
                const i32x4 av = *(const i32x4*)(stg + (t * 32 + nl) * MQ_ASTRIDE + h * 16);
                i32x16 z;
#pragma unroll
                for (int r = 0; r < 16; r++) z[r] = zk;
                const i32x16 d = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bw, z, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; j++) {   // rows t*32 + 4h + 8j + (0..3)
                    const float4 da4 = *(const float4*)(dAp + t * 32 + 4 * h + 8 * j);
                    const i32x4 s84 = *(const i32x4*)(s8p + t * 32 + 4 * h + 8 * j);
                    acc[t][4 * j + 0] = fmaf(da4.x * scur, (float)(d[4 * j + 0] - s84.x), acc[t][4 * j + 0]);
                    acc[t][4 * j + 1] = fmaf(da4.y * scur, (float)(d[4 * j + 1] - s84.y), acc[t][4 * j + 1]);
                    acc[t][4 * j + 2] = fmaf(da4.z * scur, (float)(d[4 * j + 2] - s84.z), acc[t][4 * j + 2]);
                    acc[t][4 * j + 3] = fmaf(da4.w * scur, (float)(d[4 * j + 3] - s84.w), acc[t][4 * j + 3]);
                }
            }
            asm volatile("" : "+v"(zk));   // opaque zero: the next block's MFMAs cannot be pulled above this point
            if (blk + 1 < b1) store_a(cur ^ 1);
            wcur = wnext;
            scur = snext;
        }
    }
    // ---- split-K reduction through LDS: red[ks][t][r][lane]
    __syncthreads();
    float* red = (float*)smem;
    if (KSPLIT > 1) {
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) red[((ks * MT + t) * 16 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < MT; t++) {
        if (KSPLIT > 1 && (t % KSPLIT) != ks) continue;   // wave ks finishes tiles t = ks, ks+KSPLIT, ...
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = acc[t][r];
            if (KSPLIT > 1) {
                v = 0.0f;
#pragma unroll
                for (int q = 0; q < KSPLIT; q++) v += red[((q * MT + t) * 16 + r) * 64 + lane];   // ascending K ranges
            }
            const int mrow = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mrow < p.m) {
                const size_t idx = (size_t)p.ldc * mrow + (ncol0 + nl) - p.roffset;
                p.c[idx] = p.resid ? v + p.resid[idx] : v;
            }
        }
    }
}

// ---- K2b: the same GEMM with one 32x32 output tile per workgroup (the default for K % 128 == 0).
// gemm_q8q4_mfma_kernel above keeps ALL of M in one wave's accumulators: few, fat, serial waves (<= 1 per SIMD).  The
// per-block F32 scaling that the reference arithmetic demands -- acc = fma(da*sb, (float)isum, acc) for every output and
// every Q block -- is 3 VALU ops per output per block against one 32-cycle MFMA, so the kernel is VALU-bound and
// wants many independent waves whose VALU work overlaps the others' MFMAs and loads.  Here:
//   workgroup = output tile (32 rows of M) x (32 weight rows), its S waves split K (S*4 | K/32) and meet in LDS;
//   operands go global -> registers directly (A is L2-resident, W streams from HBM once per XCD: blockIdx is mapped so
//   that all row tiles of a weight tile run on the SAME XCD and share its L2);  no barrier in the main loop;
//   nibbles are unpacked to int8 16*(nib-8) = ((nib << 4) ^ 0x80), so the MFMA sums 16*isum exactly and the -8 bias
//   needs no correction term; 1/16 is folded into the weight scale (a power of two: every rounding is unchanged);
//   the wave's activation block scales sit transposed in its private LDS slice ([blk][32 rows]) so the 16 rows a lane
//   owns are 4 broadcast ds_read_b128 per block.
// TILED: both operands are stored in MFMA order (prefill path: the activation rows are written that way by
// rows_quant_kernel, the weights are a re-tiled resident copy made at first use):
//   A  [row tile][blk][h][m][16 B]   -> lane (m, h) reads ONE contiguous 1 KB per block per wave
//   W  [col tile][blk][n][16 B], scales [col tile][blk][n]   -> 512 B / 128 B contiguous per block per wave
// Row-major operands (Tier-1 callers) touch 32 cache lines per load and use 16-32 B of each: ~8x the L2->L1 traffic.
// CW: column tiles per workgroup.  The CW*S waves of a workgroup share the row tile: the A tile of a block is fetched
// once into the CU's L1 and hit by the other CW-1 waves (the kernel is bound by the per-CU L1 miss path, A is 2/3 of
// its traffic), and the activation-scale slices in LDS are shared too.  The host guarantees n % (32*CW) == 0.
// single-instruction f32 ops hipcc cannot re-pack into v_pk_*_f32
__device__ __forceinline__ float mul1(float a, float b) { float r; asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fma1(float a, float b, float c) { asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(c) : "v"(a), "v"(b)); return c; }

template <int S, bool TILED, int CW>
__global__ __launch_bounds__(S * CW * 64) __attribute__((amdgpu_waves_per_eu(3))) void gemm_q8q4_tile_kernel(MfmaQ4Params p, int mtiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wv_id % S, cw = wv_id / S;
    const int nl = lane & 31, h = lane >> 5;
    // blockIdx.x = g_lo + 8*(rt + mtiles*g_hi), column group cg = g_hi*8 + g_lo (CW tiles): consecutive workgroups go
    // to consecutive XCDs, so XCD = g_lo for every row tile of a column group (they share the weights in its L2)
    const int g_lo = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int rt = rest % mtiles, ct = ((rest / mtiles) * 8 + g_lo) * CW + cw;
    if (ct * 32 >= p.n) return;                           // whole workgroups only (n % (32*CW) == 0)
    const int ncol = p.n0 + ct * 32 + nl;
    int arow = rt * 32 + nl;
    arow = arow < p.m ? arow : p.m - 1;                   // rows beyond M replicate the last row (never stored)
    const int nblk = p.k / QB, nbr = nblk / S, b0 = ks * nbr;
    float* dA = (float*)smem + (size_t)ks * nbr * 32;     // the K range's slice: [nbr][32 rows], shared by the CW waves
    if (TILED) {   // block scales arrive as [row tile][blk][32 rows] (rows_quant_kernel, tiled): a straight, coalesced copy
        const float4* src = (const float4*)(p.af + ((size_t)rt * nblk + b0) * 32);
        for (int i4 = lane + 64 * cw; i4 < nbr * 8; i4 += 64 * CW) ((float4*)dA)[i4] = src[i4];
    } else {       // row-major da[row][blk]: transpose while staging; lane (row, h) of wave cw covers blocks h + 2*cw, step 2*CW
        const float* src = p.af + (size_t)arow * p.ldaf + b0;
        for (int i = h + 2 * cw; i < nbr; i += 2 * CW) dA[i * 32 + nl] = src[i];
    }
    if (CW > 1) __syncthreads();
    // per-block strides: row-major A/W advance 32 / 16 bytes and 1 scale per block; tiled operands 1 KB / 512 B / 32 scales
    const int8_t* ap = TILED ? p.a + (((size_t)rt * nblk + b0) * 64 + lane) * 16 : p.a + (size_t)arow * p.lda + (size_t)b0 * QB + h * 16;
    const uint8_t* wp = TILED ? p.w + (((size_t)(p.n0 / 32 + ct) * nblk + b0) * 32 + nl) * 16 : p.w + (size_t)ncol * p.ldb + (size_t)b0 * 16;
    const float* sp = TILED ? p.ws + ((size_t)(p.n0 / 32 + ct) * nblk + b0) * 32 + nl : p.ws + (size_t)ncol * p.ldbf + b0;
    constexpr int ASTEP = TILED ? 1024 : QB, WSTEP = TILED ? 512 : 16, SSTEP = TILED ? 32 : 1;
    // two register sets of 4 blocks each (A 16 B, W 16 B, scale per block), ping-pong: the loads of chunk c+1 are issued
    // before chunk c is computed, so they have ~250 instructions of cover; the loop is unrolled over both sets (no copies)
    i32x4 a0[4], w0[4], a1[4], w1[4];
    float s0[4], s1[4];
    const int last = nbr - 1;
    auto load_chunk = [&](i32x4 (&av)[4], i32x4 (&wv)[4], float (&sv)[4], int blk0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int blk = blk0 + q;
            blk = blk < last ? blk : last;                    // branch-free: blocks past the range reload the last one
            av[q] = *(const i32x4*)(ap + (size_t)blk * ASTEP);
            wv[q] = __builtin_nontemporal_load((const i32x4*)(wp + (size_t)blk * WSTEP));
            sv[q] = sp[(size_t)blk * SSTEP];
        }
    };
    const int sh = h ? 0 : 4;
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    auto compute_chunk = [&](const i32x4 (&av)[4], const i32x4 (&wv)[4], const float (&sv)[4], int blk0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            i32x4 bw = wv[q] << sh;
            bw = (bw & (int)0xF0F0F0F0) ^ (int)0x80808080;     // int8 16*(nib-8): low nibbles for h=0, high for h=1
            const i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // inline-constant C operand
            const i32x16 d = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[q], bw, z, 0, 0, 0);
            const float s16 = sv[q] * 0.0625f;
            const float* dr = dA + (blk0 + q) * 32 + 4 * h;
#pragma unroll
            for (int j = 0; j < 4; j++) {   // rows 4h + 8j + (0..3)
                const float4 da4 = *(const float4*)(dr + 8 * j);
                // single f32 instructions (mul1 / fma1): the packed forms cost more than two scalar ones beside MFMAs on this chip
                acc[4 * j + 0] = fma1(mul1(da4.x, s16), (float)d[4 * j + 0], acc[4 * j + 0]);
                acc[4 * j + 1] = fma1(mul1(da4.y, s16), (float)d[4 * j + 1], acc[4 * j + 1]);
                acc[4 * j + 2] = fma1(mul1(da4.z, s16), (float)d[4 * j + 2], acc[4 * j + 2]);
                acc[4 * j + 3] = fma1(mul1(da4.w, s16), (float)d[4 * j + 3], acc[4 * j + 3]);
            }
            // pin this block's scaling here (the optimizer otherwise sinks all four blocks' VALU work below the fourth
            // MFMA and keeps four result tiles live): the accumulators pass through an opaque asm
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                         "+v"(acc[8]), "+v"(acc[9]), "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]));
        }
    };
    load_chunk(a0, w0, s0, 0);
    for (int c = 0; c < nbr; c += 8) {   // host guarantees nbr % 8 == 0
        load_chunk(a1, w1, s1, c + 4);
        compute_chunk(a0, w0, s0, c);
        load_chunk(a0, w0, s0, c + 8);
        compute_chunk(a1, w1, s1, c + 4);
    }
    // ---- split-K partials meet in LDS (after every wave is done with its scale slice); wave ks finishes registers
    // r = ks*(16/S) ..., summing the K ranges in ascending order
    float* red = (float*)smem + (size_t)cw * S * 16 * 64;   // one reduction region per column tile of the workgroup
    if (S > 1) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; r++) red[(ks * 16 + r) * 64 + lane] = acc[r];
        __syncthreads();
    }
    constexpr int RP = 16 / S;
#pragma unroll
    for (int i = 0; i < RP; i++) {
        const int r = S > 1 ? ks * RP + i : i;
        float v;
        if (S > 1) {
            v = 0.0f;
#pragma unroll
            for (int q = 0; q < S; q++) v += red[(q * 16 + r) * 64 + lane];
        } else {
            v = acc[i];
        }
        const int mrow = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (mrow < p.m) {
            const size_t idx = (size_t)p.ldc * mrow + ncol - p.roffset;
            p.c[idx] = p.resid ? v + p.resid[idx] : v;
        }
    }
}

// ---- I8 x Q4 GEMM, MFMA-ordered operands, A through LDS ------------------------------------------------------------------
// gemm_q8q4_tile_kernel fetches the activation tile once per WAVE (1 KB per Q block and 32x32 output tile, L1 hits for the
// other waves of the workgroup): with the weights that is 1.5 KB through the CU's texture path per MFMA, and the knock-out runs
// show the path, not the VALU scaling, as the bound.  Here a workgroup = ONE row tile x (CW waves x CT column tiles each):
// the A blocks of a 4-block chunk are fetched once per workgroup (each wave one block, registers -> LDS, double buffered, one
// barrier per chunk) and read back by every wave with one ds_read_b128 per block that serves its CT MFMAs; per MFMA the
// texture path now carries 512 B of weights + 128 B of scales + 1 KB / (CW*CT) of A.
// VALU per (block, tile): 8 (nibble unpack) + 16 v_cvt_f32_i32 + 16 v_mul_f32 + 16 v_fma_f32 + 1, all SINGLE f32 instructions
// written as asm helpers: on this chip a wave64 f32 VALU op issues in 2 cycles while the packed forms (v_pk_mul/fma_f32,
// which hipcc's SLP vectoriser produces by itself from adjacent scalar ops) cost more than two scalar ones beside MFMAs
// (MI355X_MICROARCH.md, "price of one filler beside MFMAs").  Arithmetic per output and block: acc = fma(da*sb, (float)isum, acc).
// A wave issues at most one instruction per 4 cycles while a SIMD retires a VALU op in 2, so the chip only fills with >= 4
// resident waves per SIMD: S waves of a workgroup split the K range of each column tile (own A ring per slice, partial tiles
// meet in LDS, ascending order) -- at M = 129 one wave per output tile is 2.2 waves per SIMD for the whole launch.
// grid.y = further K slices across workgroups (small N): partials go to a workspace [slice][m][n], splitk_reduce_kernel adds them.
template <int CW, int CT, int S, bool PK = false>
__global__ __launch_bounds__(CW * S * 64) void gemm_q8q4_lds_kernel(MfmaQ4Params p, int mtiles, int nbz, float* part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wv % CW, ks = wv / CW;                    // column-tile slot and K slice of this wave
    const int nl = lane & 31, h = lane >> 5;
    // blockIdx.x = g_lo + 8*(rt + mtiles*g_hi): the row tiles of a column group run on ONE XCD and share its L2 copy of the weights
    const int g_lo = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int rt = rest % mtiles, cg = (rest / mtiles) * 8 + g_lo;
    if (cg * CW * CT * 32 >= p.n) return;                    // whole workgroups only (n % (32*CW*CT) == 0)
    const int ct0 = (cg * CW + cw) * CT;
    const int nblk = p.k / QB, z = blockIdx.y, nbs = nbz / S, b0 = z * nbz + ks * nbs;   // this wave's blocks: [b0, b0 + nbs)
    float* dA = (float*)smem;                                // [nbz][32 rows] activation block scales of the workgroup's K range
    i32x4* ring = (i32x4*)(smem + (size_t)nbz * 128) + (size_t)ks * 2 * 4 * 64;   // per K slice: [2 chunks][4 blocks][64 lanes] x 16 B
    {
        const float4* src = (const float4*)(p.af + ((size_t)rt * nblk + z * nbz) * 32);
        for (int i4 = tid; i4 < nbz * 8; i4 += CW * S * 64) ((float4*)dA)[i4] = src[i4];
    }
    const float* dAs = dA + (size_t)ks * nbs * 32;
    const i32x4* ap = (const i32x4*)p.a + ((size_t)rt * nblk + b0) * 64 + lane;            // + blk * 64
    const i32x4* wp[CT];
    const float* sp[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) {
        wp[t] = (const i32x4*)p.w + ((size_t)(p.n0 / 32 + ct0 + t) * nblk + b0) * 32 + nl;   // + blk * 32
        sp[t] = p.ws + ((size_t)(p.n0 / 32 + ct0 + t) * nblk + b0) * 32 + nl;                // + blk * 32
    }
    constexpr int APW = (4 + CW - 1) / CW;                   // A blocks of a chunk fetched by one wave of the K slice
    const int last = nbs - 1;
    i32x4 ast[APW], w0[4][CT], w1[4][CT];
    float s0[4][CT], s1[4][CT];
    auto load_a = [&](int blk0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < APW; i++) {
            int blk = blk0 + cw + i * CW;
            blk = blk < last ? blk : last;
            ast[i] = ap[(size_t)blk * 64];
        }
    };
    auto store_a = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < APW; i++)
            if (cw + i * CW < 4) ring[(buf * 4 + cw + i * CW) * 64 + lane] = ast[i];
    };
    auto load_w = [&](i32x4 (&wv_)[4][CT], float (&sv)[4][CT], int blk0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int blk = blk0 + q;
            blk = blk < last ? blk : last;
#pragma unroll
            for (int t = 0; t < CT; t++) {
                wv_[q][t] = __builtin_nontemporal_load(wp[t] + (size_t)blk * 32);
                sv[q][t] = sp[t][(size_t)blk * 32];
            }
        }
    };
    const int sh = h ? 0 : 4;
    float acc[CT][16];
#pragma unroll
    for (int t = 0; t < CT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.0f;
    auto compute_chunk = [&](int buf, const i32x4 (&wv_)[4][CT], const float (&sv)[4][CT], int blk0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const i32x4 av = ring[(buf * 4 + q) * 64 + lane];
            const float* dr = dAs + (blk0 + q) * 32 + 4 * h;
            float da[16];
#pragma unroll
            for (int j = 0; j < 4; j++) {   // rows 4h + 8j + (0..3)
                const float4 v = *(const float4*)(dr + 8 * j);
                da[4 * j + 0] = v.x; da[4 * j + 1] = v.y; da[4 * j + 2] = v.z; da[4 * j + 3] = v.w;
            }
#pragma unroll
            for (int t = 0; t < CT; t++) {
                i32x4 bw = wv_[q][t] << sh;
                bw = (bw & (int)0xF0F0F0F0) ^ (int)0x80808080;                   // int8 16*(nib-8): low nibbles for h=0, high for h=1
                const i32x16 z16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const i32x16 d = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bw, z16, 0, 0, 0);
                const float s16 = sv[q][t] * 0.0625f;
                if constexpr (PK) {   // experiment: the packed forms
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 pp = f32x2{da[r], da[r + 1]} * s16;
                        const f32x2 dd = f32x2{(float)d[r], (float)d[r + 1]};
                        const f32x2 aa = __builtin_elementwise_fma(pp, dd, f32x2{acc[t][r], acc[t][r + 1]});
                        acc[t][r] = aa.x; acc[t][r + 1] = aa.y;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[t][r] = fma1(mul1(da[r], s16), (float)d[r], acc[t][r]);   // the conversion stays visible to hipcc: it owns the MFMA -> VALU hazard
                }
                // pin this tile's scaling here (the optimizer otherwise sinks all the VALU work below the chunk's last MFMA and
                // keeps eight result tiles live): the accumulators pass through an opaque asm
                asm volatile("" : "+v"(acc[t][0]), "+v"(acc[t][1]), "+v"(acc[t][2]), "+v"(acc[t][3]), "+v"(acc[t][4]), "+v"(acc[t][5]),
                             "+v"(acc[t][6]), "+v"(acc[t][7]), "+v"(acc[t][8]), "+v"(acc[t][9]), "+v"(acc[t][10]), "+v"(acc[t][11]),
                             "+v"(acc[t][12]), "+v"(acc[t][13]), "+v"(acc[t][14]), "+v"(acc[t][15]));
            }
        }
    };
    load_a(0);
    load_w(w0, s0, 0);
    store_a(0);
    lds_barrier();
    for (int c = 0; c < nbs; c += 8) {   // host guarantees nbs % 8 == 0
        load_a(c + 4);
        load_w(w1, s1, c + 4);
        compute_chunk(0, w0, s0, c);
        store_a(1);
        lds_barrier();
        load_a(c + 8);
        load_w(w0, s0, c + 8);
        compute_chunk(1, w1, s1, c + 4);
        store_a(0);
        lds_barrier();
    }
    // ---- K slices of the workgroup meet in LDS (the scale / ring regions are dead now); wave ks finishes registers
    // r = ks*(16/S) ..., summing the slices in ascending K order
    float* red = (float*)smem + (size_t)cw * S * CT * 16 * 64;
    if (S > 1) {
#pragma unroll
        for (int t = 0; t < CT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) red[((ks * CT + t) * 16 + r) * 64 + lane] = acc[t][r];
        lds_barrier();
    }
    constexpr int RP = 16 / S;
#pragma unroll
    for (int t = 0; t < CT; t++) {
        const int col = (ct0 + t) * 32 + nl;
#pragma unroll
        for (int i = 0; i < RP; i++) {
            const int r = S > 1 ? ks * RP + i : i;
            float v;
            if (S > 1) {
                v = 0.0f;
#pragma unroll
                for (int q = 0; q < S; q++) v += red[((q * CT + t) * 16 + r) * 64 + lane];
            } else {
                v = acc[t][i];
            }
            const int mrow = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (mrow < p.m) {
                if (part) {
                    part[((size_t)z * p.m + mrow) * p.n + col] = v;
                } else {
                    const size_t idx = (size_t)p.ldc * mrow + p.n0 + col - p.roffset;
                    p.c[idx] = p.resid ? v + p.resid[idx] : v;
                }
            }
        }
    }
}

static __global__ void add_rows_kernel(const float* a, const float* b, float* out, int n) {   // out = a + b (residual add)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}
static __global__ void set_int_kernel(int* p, int v) { *p = v; }
// tensor-parallel one-shot reduction: a shard's partial [E] goes straight into its slot of EVERY shard's
// slot buffer (peer stores over xGMI when the destination lives on another device) ...
static __global__ void tp_scatter_kernel(const float* part, float* const* dst, int n_dst, int E) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    const float v = part[i];
    for (int j = 0; j < n_dst; j++) dst[j][i] = v;
}
// ... and every shard sums the N slots locally, in shard order 0..N-1 (the lock-step order: results never depend on which
// peer arrived first)
static __global__ void tp_sum_kernel(const float* slots, int n, int E, float* out, size_t stride) {   // stride: floats between two shards' slots
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    float v = slots[i];
    for (int k = 1; k < n; k++) v = v + slots[k * stride + i];
    out[i] = v;
}
// ---- the same reduction without the host in it: every shard replays ONE captured graph per token, the shards meet in kernels.
// A shard's scatter workgroup b stores its 256 elements into its slot on every shard (system-scope stores: peer memory over
// xGMI), fences, then raises its flag word b on every shard to the sequence number of this (token, layer); the consumer's
// workgroup b waits for the N producers' flag b -- the element ranges coincide, so no grid-wide meeting is needed -- and sums the
// N slots in shard order 0..N-1 (+ residual).  Sequence numbers only grow, slots are reused two rounds later (by then every
// shard has consumed them: it could not have produced its next partial otherwise).
struct TPMail { int token, pos; unsigned seq; int pad; };   // shard 0 -> the others: the row to process next
// Every wait is bounded (50 ms of the 100 MHz wall clock): a peer that never arrives -- e.g. two shards' streams serialised onto
// one hardware queue -- must end in an error code on the host (word seqp[1]), never in a hung GPU.  Once raised the flag makes
// every later wait of the launch fall through at once.
// the bound in ticks of the 100 MHz wall clock: the word behind the error word (seq buffer: {tokens replayed, error, bound});
// 50 ms when all shards share a process (their graphs are launched back to back), seconds when every shard is its own process
// (a rank may still be instantiating its graph when the others already wait)
__device__ __forceinline__ long long tp_wait_bound(const unsigned* err) {
    const unsigned b = err[1];
    return b ? (long long)b : 5000000LL;
}
__device__ __forceinline__ bool tp_wait_ge(const unsigned* f, unsigned want, unsigned* err) {
    const long long t0 = wall_clock64();
    // relaxed system-scope polls (an acquire here is a cache invalidate per poll): what the flag guards is read with ld_sys,
    // cache-bypassing loads issued after this loop has ended
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if (wall_clock64() - t0 > tp_wait_bound(err)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}
static __global__ __launch_bounds__(256) void tp_scatter_flag_kernel(const float* part, float* const* dst, unsigned* const* fdst, int n_dst, int E,
                                                              const unsigned* seqp, int li, int L) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < E) {
        const float v = part[i];
        for (int j = 0; j < n_dst; j++) st_sys(dst[j] + i, v);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned seq = *seqp * (unsigned)L + (unsigned)li + 1u;
        for (int j = 0; j < n_dst; j++) __hip_atomic_store(fdst[j] + blockIdx.x, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static __global__ __launch_bounds__(256) void tp_sum_wait_kernel(const float* slots, const unsigned* flags, int n, int E, int nwg, unsigned* seqp,
                                                          int li, int L, const float* resid, float* out) {
    const unsigned seq = *seqp * (unsigned)L + (unsigned)li + 1u;
    if ((int)threadIdx.x < n) tp_wait_ge(flags + (size_t)threadIdx.x * nwg + blockIdx.x, seq, seqp + 1);
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= E) return;
    float v = ld_sys(slots + i);
    for (int k = 1; k < n; k++) v = v + ld_sys(slots + (size_t)k * E + i);   // shard order: the lock-step order
    out[i] = resid[i] + v;                                                      // TransformerBlock.java:185 / :203
}
// the same meeting when the producers were the o-proj / down GEMVs themselves (GemvParams::tp_*): their workgroups own row ranges
// that depend on the launch plan, so every consumer workgroup waits for ALL nflags workgroup flags of the N producers (a few
// hundred words polled by 16 workgroups -- not the 256 x 256 of a grid barrier)
static __global__ __launch_bounds__(256) void tp_sum_wait_all_kernel(const float* slots, const unsigned* flags, int n, int E, int nflags, int stride,
                                                              unsigned* seqp, int li, int L, const float* resid, float* out) {
    const unsigned seq = *seqp * (unsigned)L + (unsigned)li + 1u;
    {   // a thread's flags are polled TOGETHER (independent loads, one memory round trip per sweep), bounded like tp_wait_ge
        const long long t0 = wall_clock64();
        unsigned* err = seqp + 1;
        for (;;) {
            bool all = true;
            for (int i = threadIdx.x; i < n * nflags; i += 256)
                all &= __hip_atomic_load(flags + (size_t)(i / nflags) * stride + (i % nflags), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= seq;
            if (all) break;
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (wall_clock64() - t0 > tp_wait_bound(err)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= E) return;
    float v = ld_sys(slots + i);
    for (int k = 1; k < n; k++) v = v + ld_sys(slots + (size_t)k * E + i);   // shard order: the lock-step order
    out[i] = resid[i] + v;                                                      // TransformerBlock.java:185 / :203
}
// shard 0, after finish_token_kernel: the next row (token, position) to every other shard's mailbox
static __global__ void tp_publish_token_kernel(const DecodeState* st, TPMail* const* mails, int n, const unsigned* seqp) {
    if (threadIdx.x >= (unsigned)n) return;
    TPMail* m = mails[threadIdx.x];
    __hip_atomic_store(&m->token, st->token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&m->pos, st->pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __hip_atomic_store(&m->seq, *seqp + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the other shards, first node of their token graph: wait for the row of this replay, take it over
static __global__ void tp_wait_token_kernel(const TPMail* mail, unsigned* seqp, DecodeState* st) {
    tp_wait_ge(&mail->seq, *seqp, seqp + 1);
    st->token = __hip_atomic_load(&mail->token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    st->pos = __hip_atomic_load(&mail->pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
static __global__ void tp_bump_seq_kernel(unsigned* seqp) { *seqp = *seqp + 1u; }
static __global__ void set_pos_kernel(DecodeState* st, int pos) { st->pos = pos; }   // pipeline stages: the token / step words stay
// one pipeline stage per process: the token id arrives in device memory (shipped by the last stage), never through the host
static __global__ void set_state_dev_kernel(DecodeState* st, int pos, const int* token_dev) {
    st->pos = pos; st->step = 0; st->done = 0;
    if (token_dev) st->token = *token_dev;
}
static __global__ void store_token_kernel(const DecodeState* st, int* out) { *out = st->token; }
static __global__ void set_state_kernel(DecodeState* st, int pos, int token, int step) {
    st->pos = pos; st->token = token; st->step = step; st->done = 0;
}

// EmbedInput for a Q4 / BF16 / F32 table (core/model/llama/LlamaModel.java:67-98): x[j] = (nib-8)*scale
__device__ __forceinline__ void embed_row(const void* table, const float* scales, int dtype, int token, int E,
                                          float* x) {
    if (dtype == 3) {
        const uint8_t* nr = (const uint8_t*)table + (size_t)token * (E / 2);
        const float* sr = scales + (size_t)token * (E / QB);
        for (int j = threadIdx.x; j < E / 2; j += blockDim.x) {
            const int blk = j / 16, in = j % 16;
            const uint8_t b = nr[j];
            const float s = sr[blk];
            x[blk * 32 + in] = (float)((int)(b & 0x0F) - 8) * s;
            x[blk * 32 + in + 16] = (float)((int)((b >> 4) & 0x0F) - 8) * s;
        }
    } else if (dtype == 1) {
        const uint16_t* r = (const uint16_t*)table + (size_t)token * E;
        for (int j = threadIdx.x; j < E; j += blockDim.x) x[j] = bf16_to_f32(r[j]);
    } else {
        const float* r = (const float*)table + (size_t)token * E;
        for (int j = threadIdx.x; j < E; j += blockDim.x) x[j] = r[j];
    }
}

static __global__ void embed_kernel(const void* table, const float* scales, int dtype, const DecodeState* st, int E,
                             float* x) {
    embed_row(table, scales, dtype, st->token, E, x);
}

// argmax over the LM head's per-workgroup partials -> next token; advance the decode state and look up the
// next embedding row, so a greedy decode step needs no host round trip (AbstractModel.java:590-599).
// Stop tokens (Config.eosTokens, AbstractModel.java:600-603): the step that samples one is the last; st->done freezes
// the state, so the remaining replays of an already-queued decode loop emit nothing.
// ---- temperature sampling inside the device loop (AbstractModel.sample, core/model/AbstractModel.java:471-489):
//   v_i = (float)exp((logit_i - max) / T) in double;  sum = float running sum over i in INDEX ORDER;  acc += v_i / sum until
//   acc >= u -> token i (V-1 if never).  The exponentials are independent (sample_exp_kernel, whole chip); the two float
//   accumulations are defined by their index order -- sample_pick_kernel below reproduces them bit for bit -- and a T > 0
//   generation stays at one graph replay per token, no host round trip.
static __global__ __launch_bounds__(256) void sample_exp_kernel(const float* logits, int V, const float* partv, int nparts, float temperature, float* prob) {
    __shared__ float red[4];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, partv[i]);   // the LM head's per-workgroup maxima
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const double maxv = (double)fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < V) prob[i] = (float)exp(((double)logits[i] - maxv) / (double)temperature);
}
// The two accumulations, exactly, by 1024 lanes (jh_seqsum.h).  The values are taken in windows of 1024 x SAMPLE_E: a lane owns
// SAMPLE_E consecutive values (coalesced 16-byte loads, transposed through LDS), adds them to the two reference values of the
// running sum's binade (2 float adds per value), a scan composes the lanes' parity -> increment maps, and the first lane whose
// partial sum leaves the binade (or reaches u) walks its own values with plain adds; the lanes behind it redo their maps in the
// new binade from registers.  V / (1024 E) windows + one scan per binade crossing instead of 128256 dependent adds.
//   sample_exp_kernel (chip) -> sample_sum_kernel (1 workgroup) -> sample_norm_kernel (chip: v / sum) -> sample_pick_kernel (1 workgroup)
constexpr int SAMPLE_T_DEFAULT = 1024, SAMPLE_E_DEFAULT = 16, SAMPLE_T_MAX = 1024;
struct SeqShared {
    SeqStep tot[SAMPLE_T_MAX / 64];
    int first[SAMPLE_T_MAX / 64];
    int m_end;
    float s;
    int pick;
};
static inline size_t lds_bytes_sample(int t, int e) { return (size_t)t * e * 4; }
// inclusive scan of the lanes' maps over a wave: row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast:15 / row_bcast:31
// carry the row totals over (lanes without a source receive the identity map {0, 0})
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ SeqStep seq_dpp(SeqStep h) {
    SeqStep o;
    o.d0 = __builtin_amdgcn_update_dpp(0, h.d0, CTRL, ROW_MASK, 0xf, false);
    o.d1 = __builtin_amdgcn_update_dpp(0, h.d1, CTRL, ROW_MASK, 0xf, false);
    return o;
}
__device__ __forceinline__ SeqStep seq_wave_scan(SeqStep h) {
    h = seq_compose(seq_dpp<0x111, 0xf>(h), h);
    h = seq_compose(seq_dpp<0x112, 0xf>(h), h);
    h = seq_compose(seq_dpp<0x114, 0xf>(h), h);
    h = seq_compose(seq_dpp<0x118, 0xf>(h), h);
    h = seq_compose(seq_dpp<0x142, 0xa>(h), h);
    h = seq_compose(seq_dpp<0x143, 0xc>(h), h);
    return h;
}
// s = sum of val[0..V) in index order; pick = first i with s >= u (then s is the sum up to i), else -1.  xt: SAMPLE_T * SAMPLE_E floats of LDS.
// (tools/seqsum_lab.hip: 1024 x 16 and 512 x 16 are the fastest shapes; more values per lane lengthen every walk,
// fewer lanes lengthen every window)
template <int SAMPLE_T, int SAMPLE_E>
__device__ __forceinline__ void seq_pass(const float* __restrict__ val, int V, float u, SeqShared& sh, float* xt, float& s_out, int& pick_out) {
    constexpr int G = SAMPLE_E / 4;                             // 16-byte groups per lane and window
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 g4[G];
    auto request = [&](int wpos) __attribute__((always_inline)) {   // window [wpos, wpos + T*E): group j*T + tid of the window
#pragma unroll
        for (int j = 0; j < G; j++) {
            const int e = wpos + 4 * (j * SAMPLE_T + tid);
            if (e + 4 <= V) g4[j] = *(const f32x4*)(val + e);
            else {
                g4[j].x = e < V ? val[e] : 0.0f; g4[j].y = e + 1 < V ? val[e + 1] : 0.0f;       // + 0: the sum stays as it is
                g4[j].z = e + 2 < V ? val[e + 2] : 0.0f; g4[j].w = 0.0f;
            }
        }
    };
    // LDS slot of window group g: g ^ ((g >> 3) & 3) -- the lanes' own groups G*tid + i then spread over all banks
    auto slot = [](int g) __attribute__((always_inline)) { return g ^ ((g >> 3) & 3); };
    float s = 0.0f;
    int pick = -1;
    request(0);
    for (int wpos = 0; wpos < V && pick < 0; wpos += SAMPLE_T * SAMPLE_E) {
        float x[SAMPLE_E];
#pragma unroll
        for (int j = 0; j < G; j++) *(f32x4*)(xt + 4 * slot(j * SAMPLE_T + tid)) = g4[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < G; j++) {
            const f32x4 t = *(const f32x4*)(xt + 4 * slot(G * tid + j));
            x[4 * j] = t.x; x[4 * j + 1] = t.y; x[4 * j + 2] = t.z; x[4 * j + 3] = t.w;
        }
        if (wpos + SAMPLE_T * SAMPLE_E < V) request(wpos + SAMPLE_T * SAMPLE_E);   // in flight across this window's scans
        auto lane_map = [&](int k, bool live) __attribute__((always_inline)) {
            SeqStep h{0, 0};
            if (live) {
                SeqRef ref = seq_ref_begin(k);
#pragma unroll
                for (int i = 0; i < SAMPLE_E; i++) seq_ref_add(ref, x[i]);
                h = seq_ref_end(ref, k);
            }
            return h;
        };
        // the reference's own loop over this lane's values, from the exact sum w in front of them
        auto walk = [&](float& w, int& pk) __attribute__((always_inline)) {
            const int e0 = wpos + tid * SAMPLE_E;
#pragma unroll
            for (int i = 0; i < SAMPLE_E; i++) {
                if (pk < 0) {
                    w += x[i];
                    if (w >= u && e0 + i < V) pk = e0 + i;
                }
            }
        };
        int lo = 0;                                             // lanes below lo are behind the running sum already
        while (true) {
            const int k = seq_k(s), m_in = seq_m(s), th = seq_threshold(u, k);
            const SeqStep h = seq_wave_scan(lane_map(k, tid >= lo));
            if (lane == 63) sh.tot[wave] = h;
            __syncthreads();                                    // (also: every lane has read its window values from xt)
            SeqStep t = lane < SAMPLE_T / 64 ? sh.tot[lane & (SAMPLE_T / 64 - 1)] : SeqStep{0, 0};
            t = seq_wave_scan(t);                               // lanes 0..15: the waves' totals up to and including wave `lane`
            SeqStep pre{0, 0};
            if (wave > 0) { pre.d0 = __builtin_amdgcn_readlane(t.d0, wave - 1); pre.d1 = __builtin_amdgcn_readlane(t.d1, wave - 1); }
            const int m_wave = seq_apply(m_in, pre);            // the sum in front of this wave
            const int m = seq_apply(m_wave, h);
            const unsigned long long hit = __ballot(tid >= lo && m >= th);
            if (lane == 0) sh.first[wave] = hit ? wave : 0x7fffffff;
            if (lane == 63 && wave == SAMPLE_T / 64 - 1) sh.m_end = m;
            __syncthreads();
            int we = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < SAMPLE_T / 64; w++) we = we < sh.first[w] ? we : sh.first[w];
            if (we == 0x7fffffff) {
                s = seq_value(sh.m_end, k);
                break;
            }
            if (wave == we) {
                // this wave holds the first lane that leaves the binade (or reaches u): it settles ALL its lanes among itself --
                // walk that lane, redo the maps of the lanes behind it in the new binade, scan, next such lane, ... -- no barrier
                int kk = k, mw = m_wave, wlo = lo > wave * 64 ? lo - wave * 64 : 0, pk = -1;
                unsigned long long hh = hit;
                SeqStep hs = h;
                float sw = 0.0f;
                while (true) {
                    const int te = __builtin_ctzll(hh);         // uniform
                    int m_before = seq_apply(mw, hs);           // behind this lane ...
                    m_before = __shfl_up(m_before, 1);          // ... so in front of the next one
                    if (lane == wlo) m_before = mw;             // (lanes below wlo hold identity maps: lane wlo starts at mw)
                    float w = seq_value(m_before, kk);
                    int p1 = -1;
                    if (lane == te) walk(w, p1);
                    sw = __shfl(w, te);
                    pk = __shfl(p1, te);
                    if (pk >= 0 || te == 63) break;
                    wlo = te + 1;
                    kk = seq_k(sw);
                    mw = seq_m(sw);
                    const int th2 = seq_threshold(u, kk);
                    hs = seq_wave_scan(lane_map(kk, lane >= wlo));
                    hh = __ballot(lane >= wlo && seq_apply(mw, hs) >= th2);
                    if (!hh) {
                        sw = seq_value(__builtin_amdgcn_readlane(seq_apply(mw, hs), 63), kk);
                        break;
                    }
                }
                if (lane == 0) { sh.s = sw; sh.pick = pk; }
            }
            __syncthreads();
            s = sh.s;
            pick = sh.pick;
            if (pick >= 0) break;
            lo = (we + 1) * 64;
            if (lo >= SAMPLE_T) break;
        }
    }
    s_out = s;
    pick_out = pick;
}
template <int SAMPLE_T, int SAMPLE_E>
__global__ __launch_bounds__(SAMPLE_T) void sample_sum_kernel(const float* prob, int V, const DecodeState* st, float* sum_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ SeqShared sh;
    if (st->done) return;
    float sum;
    int pick;
    seq_pass<SAMPLE_T, SAMPLE_E>(prob, V, INFINITY, sh, (float*)smem, sum, pick);   // AbstractModel.java:475-480
    if (threadIdx.x == 0) *sum_out = sum;
}
static __global__ __launch_bounds__(256) void sample_norm_kernel(float* prob, int V, const DecodeState* st, const float* sum) {
    if (st->done) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < V) prob[i] = prob[i] / *sum;                              // :481-483
}
template <int SAMPLE_T, int SAMPLE_E>
__global__ __launch_bounds__(SAMPLE_T) void sample_pick_kernel(const float* prob, int V, const float* u, const DecodeState* st, int* tok_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ SeqShared sh;
    if (st->done) return;
    float acc;
    int pick;
    seq_pass<SAMPLE_T, SAMPLE_E>(prob, V, u[st->step], sh, (float*)smem, acc, pick);   // :484-489
    if (threadIdx.x == 0) *tok_out = pick >= 0 ? pick : V - 1;
}

static __global__ void finish_token_kernel(const float* partv, const int* parti, int nparts, DecodeState* st,
                                    int* out_tokens, const void* table, const float* scales, int dtype, int E,
                                    float* x, int do_embed, const int* eos,   // eos: [count, id0, id1, ...] (jh_session_set_eos)
                                    const int* forced) {                      // non-null: the sampled id (sample_pick_kernel) replaces the argmax
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ int tok;
    if (st->done) return;   // uniform: every thread reads the same word
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
        const float v = partv[i];
        const int id = parti[i];
        if (v > bv || (v == bv && id < bi)) { bv = v; bi = id; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < nw; w++)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        if (forced) bi = *forced;
        tok = bi;
        out_tokens[st->step] = bi;
        st->token = bi;
        st->pos = st->pos + 1;
        st->step = st->step + 1;
        int stop = 0;
        const int n_eos = eos[0];
        for (int i = 0; i < n_eos; i++) stop |= (eos[1 + i] == bi);
        st->done = stop;
    }
    __syncthreads();
    if (do_embed) embed_row(table, scales, dtype, tok, E, x);
}

// ------------------------------------------------------------------------------------------------ K4: decode attention
// CausalSelfAttention.forward for ONE new position (core/model/CausalSelfAttention.java:199-357), fused:
//   copy K,V row into the KV page (:226-241) -> RoPE on q (all heads) and on the stored k row (:247-286, incl. the
//   per-kv-head table offset g = kvHead*headSize+i => effective position pos+2*kvHead) -> per head:
//   scores = q.K^T (:324-330) * attentionScale (:332) -> softMax (VectorMath.java:69-90) -> saxpy over V (:349-354).
// Grid (max_splits, n_kv_heads): a workgroup serves the `group` q-heads of one kv head for one slice of the
// context, so K/V are read once per group (GQA).  Every global load a thread will need (q, rope entry, its K rows,
// its V rows) is issued up front so the kernel pays ONE memory round trip, not five.  Slices publish (o, m, l)
// with write-through (sc1) stores + a relaxed agent-scope ticket; the last-arriving workgroup of the kv head
// combines them reading with sc1 loads -- no release/acquire fences (cdna_hip_programming.md Guideline 16, R1).
struct AttnParams {
    const float* qkv;      // [A + 2*KV]: q | k | v of the new row (F32, pre-RoPE)
    const float* rope;     // [ctx*hs/2][2]
    float* kv_base;        // context page 0 of this layer page; page cp lives at kv_base + cp*page_elems,
    long long page_elems;  //   each page = [layersPerPage, 2, ctxPerPage, KV] F32 (KvBufferCache.java:99-112)
    int rel_layer_in_page, ctx_per_page, cpp_shift;   // cpp_shift = log2(ctx_per_page) or -1
    int n_heads, n_kv_heads, head_size;
    int kv_head_offset;    // tensor-parallel shard: global index of local kv head 0 (RoPE rows are indexed globally)
    const DecodeState* st;
    float scale;
    float* part_o;         // [n_heads][part_stride][hs]   slice outputs
    float* part_ml;        // [n_heads][part_stride][2]    slice (max, sum)
    unsigned* counters;    // [n_kv_heads], zero between launches
    int max_splits;        // slices in "ticket" mode (long contexts)
    int mid_splits, mid_max;   // contexts of <= mid_max rows use at most mid_splits slices (0 = no such tier)
    int part_stride;       // >= max(max_splits, 4)
    int direct_max;        // contexts up to this length use "direct" mode: <= 4 slices of direct_chunk rows, combined by
    int direct_chunk;      //   the o-projection's prologue (PRO_ATTN_Q8); 0 disables
    float* outf;           // [A] attention output, ticket mode only (the o-projection then quantizes it)
    float* tap_q;          // roped q [A] (tap), may be null
    long long* dbg;        // optional phase timestamps (wall_clock64, 100 MHz) of workgroups with kvh == 0: [split][16]
    // reference-order prompt chunks (jh_p16.h, rows_*_p16_kernel): prompt row z sits at position batch_pos0 + z, its q|k|v row at
    // qkv + z*ldqkv, its output row at outf + z*ldo
    int batch_pos0, ldqkv, ldo;
};
#define JH_ATT_STAMP(k) do { if (p.dbg && threadIdx.x == 0 && blockIdx.y == 0) p.dbg[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)

template <class P>
__device__ __forceinline__ const float* kv_row(const P& p, int which, int t, int kvlen) {
    // page / row-in-page of position t.  ctx_per_page is a power of two for the usual geometries (32, 128): shift
    // instead of a ~40-instruction integer division per row on the kernel's address-generation critical path.
    int cp, rc;
    if (p.cpp_shift >= 0) { cp = t >> p.cpp_shift; rc = t & (p.ctx_per_page - 1); }
    else { cp = t / p.ctx_per_page; rc = t - cp * p.ctx_per_page; }
    return p.kv_base + (size_t)cp * p.page_elems + ((size_t)(p.rel_layer_in_page * 2 + which) * p.ctx_per_page + rc) * kvlen;
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store((unsigned*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

constexpr int ATT_THREADS = 512;

// PRE = row steps prefetched into registers before anything else (branch-free, clamped): PRE*RPS rows per slice
// (HS=128: 16 rows per step).  Contexts of <= max_splits*32 rows have slices of <= 32 rows, for which PRE=2 issues a
// quarter of the load instructions of PRE=8 (the texture path needs ~16 cycles per 16-byte wave load); the host picks
// the variant per token from the position it already knows.  Any PRE is correct for any context (tail loops).
template <int HS, int GROUP, int PRE>
__global__ __launch_bounds__(ATT_THREADS) void attn_decode_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = ATT_THREADS, NW = NT / 64;
    constexpr int LPR = HS / 4;          // lanes per K/V row (float4 each): 32 for HS=128, 16 for HS=64
    constexpr int RPS = NT / LPR;        // rows per workgroup step (scores and PV use the same row->thread map)
    constexpr int half = HS / 2;
    constexpr int NQ = (GROUP * half + NT - 1) / NT;   // q-rotation pairs per thread
    constexpr int CS = 16;               // slices combined per batch of in-flight loads (ticket mode); a second batch for the long-context tier
    JH_ATT_STAMP(0);
    const int pos = p.st->pos;
    const int kvh = blockIdx.y, split = blockIdx.x;
    const int n = pos + 1;
    const bool direct = n <= p.direct_max;
    int S;
    if (direct) S = (n + p.direct_chunk - 1) / p.direct_chunk;
    else {
        S = (n + 31) / 32;
        int cap = p.max_splits;
        if (p.mid_max && n <= p.mid_max && p.mid_splits < cap) cap = p.mid_splits;
        if (S > cap) S = cap;
    }
    if (split >= S) return;
    const int chunk = direct ? p.direct_chunk : (n + S - 1) / S;
    const int t0 = split * chunk;
    int t1 = t0 + chunk;
    if (t1 > n) t1 = n;
    const int cnt = t1 - t0;
    const int KV = p.n_kv_heads * HS, A = p.n_heads * HS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rsub = tid / LPR, c4 = tid % LPR;   // thread -> (row within step, float4 column)

    float* qs = (float*)smem;                // GROUP*HS roped q, later the slice output
    float* knew = qs + GROUP * HS;           // HS
    float* vnew = knew + HS;                 // HS
    float* red = vnew + HS;                  // RPS*GROUP*HS floats
    float* ml = red + RPS * GROUP * HS;      // 2*GROUP (m, l)
    int* flag = (int*)(ml + 2 * GROUP);      // 4 ints
    float* sc = (float*)(flag + 4);          // GROUP*sc_cap: scores / weights; combine scratch

    // ---- ONE round trip: every global load of the main phase is issued here, branch-free (clamped addresses), in
    // the order the results are consumed (vmcnt retires oldest-first): q + rope, new k/v, then K rows, then V rows.
    const float* rf = p.rope + ((size_t)pos * half + (size_t)(kvh + p.kv_head_offset) * HS) * 2;  // rf[poffset + g], g = kvHead*HS + i
    float q0[NQ], q1[NQ], qc[NQ], qsn[NQ];
#pragma unroll
    for (int u = 0; u < NQ; u++) {
        int i = tid + u * NT;
        i = i < GROUP * half ? i : GROUP * half - 1;
        const int gi = i / half, d = i - gi * half;
        const float* qh = p.qkv + (size_t)(kvh * GROUP + gi) * HS;
        q0[u] = qh[d]; q1[u] = qh[d + half];
        qc[u] = rf[2 * d]; qsn[u] = rf[2 * d + 1];
    }
    const int dk = tid < half ? tid : half - 1;
    const float k0n = p.qkv[A + (size_t)kvh * HS + dk], k1n = p.qkv[A + (size_t)kvh * HS + dk + half];
    const float kc = rf[2 * dk], ksn = rf[2 * dk + 1];
    const float vn = p.qkv[A + KV + (size_t)kvh * HS + (tid < HS ? tid : HS - 1)];
    float4 kreg[PRE], vreg[PRE];
#pragma unroll
    for (int k = 0; k < PRE; k++) {
        int t = t0 + rsub + k * RPS;
        t = t < 0 ? 0 : (t > pos ? pos : t);   // any valid row; out-of-slice rows are never used
        kreg[k] = ((const float4*)(kv_row(p, 0, t, KV) + (size_t)kvh * HS))[c4];
    }
#pragma unroll
    for (int k = 0; k < PRE; k++) {
        int t = t0 + rsub + k * RPS;
        t = t < 0 ? 0 : (t > pos ? pos : t);
        vreg[k] = ((const float4*)(kv_row(p, 1, t, KV) + (size_t)kvh * HS))[c4];
    }

    JH_ATT_STAMP(1);   // all loads issued
    // ---- RoPE (q for the group's heads, k for this kv head) ------------------------------------------------
#pragma unroll
    for (int u = 0; u < NQ; u++) {
        const int i = tid + u * NT;
        if (i < GROUP * half) {
            const int gi = i / half, d = i - gi * half;
            const float r0 = q0[u] * qc[u] - q1[u] * qsn[u];   // contraction is off: mul, mul, sub as in Java
            const float r1 = q0[u] * qsn[u] + q1[u] * qc[u];
            qs[gi * HS + d] = r0;
            qs[gi * HS + d + half] = r1;
            if (p.tap_q && split == 0) {
                p.tap_q[(size_t)(kvh * GROUP + gi) * HS + d] = r0;
                p.tap_q[(size_t)(kvh * GROUP + gi) * HS + d + half] = r1;
            }
        }
    }
    const bool owns_new = (pos >= t0 && pos < t1);
    if (owns_new) {
        if (tid < half) {
            const float r0 = k0n * kc - k1n * ksn;
            const float r1 = k0n * ksn + k1n * kc;
            knew[tid] = r0; knew[tid + half] = r1;
            float* kdst = (float*)kv_row(p, 0, pos, KV) + (size_t)kvh * HS;
            kdst[tid] = r0; kdst[tid + half] = r1;   // K is stored post-RoPE (:273-286 rotates the page row in place)
        }
        if (tid < HS) {
            vnew[tid] = vn;
            ((float*)kv_row(p, 1, pos, KV) + (size_t)kvh * HS)[tid] = vn;
        }
    }
    lds_barrier();
    JH_ATT_STAMP(2);   // q/rope arrived, rope done

    // ---- scores: LPR lanes x float4 cover one K row ---------------------------------------------------------
    float4 qv[GROUP];
#pragma unroll
    for (int gi = 0; gi < GROUP; gi++) qv[gi] = ((const float4*)(qs + gi * HS))[c4];
    const float4 knew4 = ((const float4*)knew)[c4];   // read once: keeps the LDS address space (no flat loads)
#pragma unroll
    for (int k = 0; k < PRE; k++) {
        const int tt = rsub + k * RPS;
        if (tt < cnt) {
            const bool isnew = (t0 + tt == pos);
            float4 kv4;
            kv4.x = isnew ? knew4.x : kreg[k].x; kv4.y = isnew ? knew4.y : kreg[k].y;
            kv4.z = isnew ? knew4.z : kreg[k].z; kv4.w = isnew ? knew4.w : kreg[k].w;
#pragma unroll
            for (int gi = 0; gi < GROUP; gi++) {
                float s = qv[gi].x * kv4.x;
                s = fmaf(qv[gi].y, kv4.y, s);
                s = fmaf(qv[gi].z, kv4.z, s);
                s = fmaf(qv[gi].w, kv4.w, s);
                s = group_sum_last<LPR>(s);
                if (c4 == LPR - 1) sc[gi * chunk + tt] = s * p.scale;   // ops.scale after the dot (:332): separate rounding
            }
        }
    }
    // slices longer than PRE*RPS rows (long contexts): further rounds of TR row steps, every load of a round issued
    // (branch-free, clamped) before the first one is consumed -- a step-at-a-time loop pays the full memory latency per step
    constexpr int TR = 8;
    for (int kb = PRE; kb * RPS < cnt; kb += TR) {   // cnt is uniform across the workgroup: no divergence on the round count
        float4 kt[TR];
#pragma unroll
        for (int u = 0; u < TR; u++) {
            int t = t0 + rsub + (kb + u) * RPS;
            t = t > pos ? pos : t;
            kt[u] = ((const float4*)(kv_row(p, 0, t, KV) + (size_t)kvh * HS))[c4];
        }
#pragma unroll
        for (int u = 0; u < TR; u++) {
            const int tt = rsub + (kb + u) * RPS, t = t0 + tt;
            if (tt < cnt) {
                float4 kv4 = kt[u];
                if (t == pos) kv4 = knew4;
#pragma unroll
                for (int gi = 0; gi < GROUP; gi++) {
                    float s = qv[gi].x * kv4.x;
                    s = fmaf(qv[gi].y, kv4.y, s);
                    s = fmaf(qv[gi].z, kv4.z, s);
                    s = fmaf(qv[gi].w, kv4.w, s);
                    s = group_sum_last<LPR>(s);
                    if (c4 == LPR - 1) sc[gi * chunk + tt] = s * p.scale;
                }
            }
        }
    }
    lds_barrier();
    JH_ATT_STAMP(3);   // K arrived, scores done

    // ---- local softmax per head (one wave per head, round-robin) ------------------------------------------
    for (int gi = wave; gi < GROUP; gi += NW) {
        float m = -INFINITY;
        for (int tt = lane; tt < cnt; tt += 64) m = fmaxf(m, sc[gi * chunk + tt]);
        m = wave_max(m);
        float l = 0.0f;
        for (int tt = lane; tt < cnt; tt += 64) {
            const float e = (float)exp((double)(sc[gi * chunk + tt] - m));  // (float)FastMath.exp(x - max)
            sc[gi * chunk + tt] = e;
            l += e;
        }
        l = wave_sum(l);
        for (int tt = lane; tt < cnt; tt += 64) sc[gi * chunk + tt] = sc[gi * chunk + tt] / l;  // normalise by division
        if (lane == 0) { ml[2 * gi] = m; ml[2 * gi + 1] = l; }
    }
    lds_barrier();
    JH_ATT_STAMP(4);   // softmax done

    // ---- o = sum_t w[t] * V[t]: thread = (row-in-step, float4 column); fma chain per element ----------------
    {
        float4 acc[GROUP];
#pragma unroll
        for (int gi = 0; gi < GROUP; gi++) acc[gi] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vnew4 = ((const float4*)vnew)[c4];
#pragma unroll
        for (int k = 0; k < PRE; k++) {
            const int tt = rsub + k * RPS;
            if (tt < cnt) {
                const bool isnew = (t0 + tt == pos);
                float4 v4;
                v4.x = isnew ? vnew4.x : vreg[k].x; v4.y = isnew ? vnew4.y : vreg[k].y;
                v4.z = isnew ? vnew4.z : vreg[k].z; v4.w = isnew ? vnew4.w : vreg[k].w;
#pragma unroll
                for (int gi = 0; gi < GROUP; gi++) {
                    const float w = sc[gi * chunk + tt];
                    acc[gi].x = fmaf(v4.x, w, acc[gi].x);
                    acc[gi].y = fmaf(v4.y, w, acc[gi].y);
                    acc[gi].z = fmaf(v4.z, w, acc[gi].z);
                    acc[gi].w = fmaf(v4.w, w, acc[gi].w);
                }
            }
        }
        for (int kb = PRE; kb * RPS < cnt; kb += TR) {
            float4 vt[TR];
#pragma unroll
            for (int u = 0; u < TR; u++) {
                int t = t0 + rsub + (kb + u) * RPS;
                t = t > pos ? pos : t;
                vt[u] = ((const float4*)(kv_row(p, 1, t, KV) + (size_t)kvh * HS))[c4];
            }
#pragma unroll
            for (int u = 0; u < TR; u++) {
                const int tt = rsub + (kb + u) * RPS, t = t0 + tt;
                if (tt < cnt) {
                    float4 v4 = vt[u];
                    if (t == pos) v4 = vnew4;
#pragma unroll
                    for (int gi = 0; gi < GROUP; gi++) {
                        const float w = sc[gi * chunk + tt];
                        acc[gi].x = fmaf(v4.x, w, acc[gi].x);
                        acc[gi].y = fmaf(v4.y, w, acc[gi].y);
                        acc[gi].z = fmaf(v4.z, w, acc[gi].z);
                        acc[gi].w = fmaf(v4.w, w, acc[gi].w);
                    }
                }
            }
        }
#pragma unroll
        for (int gi = 0; gi < GROUP; gi++) ((float4*)(red + ((size_t)rsub * GROUP + gi) * HS))[c4] = acc[gi];
    }
    lds_barrier();
    JH_ATT_STAMP(5);   // PV done
    float* oloc = qs;  // reuse q storage for the slice's output [GROUP][HS]
    for (int i = tid; i < GROUP * HS; i += NT) {
        float s = 0.0f;
#pragma unroll
        for (int rg = 0; rg < RPS; rg++) s += red[(size_t)rg * GROUP * HS + i];
        oloc[i] = s;
    }
    lds_barrier();
    JH_ATT_STAMP(6);   // slice reduced

    float* my_o = p.part_o + ((size_t)(kvh * GROUP) * p.part_stride + split) * HS;   // + gi*part_stride*HS + d
    if (direct) {
        // publish the slice and finish: the o-projection's prologue (direct mode) or attn_combine_kernel merges the
        // slices after the kernel boundary (= visibility, no write-through stores, no ticket)
        for (int i = tid; i < GROUP * HS; i += NT) {
            const int gi = i / HS, d = i - gi * HS;
            my_o[(size_t)gi * p.part_stride * HS + d] = oloc[i];
        }
        if (tid < GROUP) {
            float* pr = p.part_ml + ((size_t)(kvh * GROUP + tid) * p.part_stride + split) * 2;
            pr[0] = ml[2 * tid];
            pr[1] = ml[2 * tid + 1];
        }
        JH_ATT_STAMP(7);
        return;
    }
    if (S > 1) {
        // ticket mode: publish this slice's (o, m, l) write-through; the last arriver of the kv head combines
        for (int i = tid; i < GROUP * HS; i += NT) {
            const int gi = i / HS, d = i - gi * HS;
            st_sc1(my_o + (size_t)gi * p.part_stride * HS + d, oloc[i]);
        }
        if (tid < GROUP) {
            float* pr = p.part_ml + ((size_t)(kvh * GROUP + tid) * p.part_stride + split) * 2;
            st_sc1(pr, ml[2 * tid]);
            st_sc1(pr + 1, ml[2 * tid + 1]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
        __syncthreads();
        JH_ATT_STAMP(7);   // published + drained
        if (tid == 0) {
            const unsigned tk = __hip_atomic_fetch_add(&p.counters[kvh], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (tk == (unsigned)(S - 1));
            if (last) __hip_atomic_store(&p.counters[kvh], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag[0] = last;
        }
        __syncthreads();
        JH_ATT_STAMP(8);   // ticket drawn
        if (!flag[0]) return;
        // combine: w_s = l_s * exp(m_s - M) / sum_s(l_s * exp(m_s - M)); o = sum_s w_s * o_s.
        float* cm = sc;               // [GROUP][S] m_s, then w_s
        float* cl = sc + GROUP * S;   // [GROUP][S] l_s
        constexpr int EPT = (GROUP * HS + NT - 1) / NT;   // output elements per thread
        for (int i = tid; i < GROUP * S; i += NT) {
            const int gi = i / S, s = i - gi * S;
            const float* pr = p.part_ml + ((size_t)(kvh * GROUP + gi) * p.part_stride + s) * 2;
            cm[i] = ld_sc1(pr);
            cl[i] = ld_sc1(pr + 1);
        }
        float pv[EPT][CS];
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            int i = tid + e * NT;
            i = i < GROUP * HS ? i : GROUP * HS - 1;
            const int gi = i / HS, d = i - gi * HS;
            const float* pb = p.part_o + (size_t)(kvh * GROUP + gi) * p.part_stride * HS + d;
#pragma unroll
            for (int s = 0; s < CS; s++) pv[e][s] = ld_sc1(pb + (size_t)(s < S ? s : S - 1) * HS);
        }
        __syncthreads();
        for (int gi = wave; gi < GROUP; gi += NW) {
            float m = -INFINITY;
            for (int s = lane; s < S; s += 64) m = fmaxf(m, cm[gi * S + s]);
            m = wave_max(m);
            float L = 0.0f;
            for (int s = lane; s < S; s += 64) {
                const float w = cl[gi * S + s] * (float)exp((double)(cm[gi * S + s] - m));
                cm[gi * S + s] = w;
                L += w;
            }
            L = wave_sum(L);
            for (int s = lane; s < S; s += 64) cm[gi * S + s] = cm[gi * S + s] / L;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int i = tid + e * NT;
            if (i >= GROUP * HS) break;
            const int gi = i / HS, d = i - gi * HS;
            const float* pb = p.part_o + (size_t)(kvh * GROUP + gi) * p.part_stride * HS + d;
            float o = 0.0f;
#pragma unroll
            for (int s = 0; s < CS; s++)
                if (s < S) o = fmaf(pv[e][s], cm[gi * S + s], o);
            for (int s = CS; s < S; s++) o = fmaf(ld_sc1(pb + (size_t)s * HS), cm[gi * S + s], o);
            oloc[i] = o;
        }
        __syncthreads();
    }
    JH_ATT_STAMP(9);   // combined
    for (int i = tid; i < GROUP * HS; i += NT) p.outf[(size_t)kvh * GROUP * HS + i] = oloc[i];
}

// ------------------------------------------------------------------------------------------------ batched prefill
// AbstractModel.batchForward (core/model/AbstractModel.java:295-312) for a chunk of B <= 256 prompt rows: the
// projections run as MFMA GEMMs over all rows (gemm_q8q4_mfma_kernel), the per-row work between them is below.
// Per-row arithmetic is the decode path's (same RMSNorm / Q8 / RoPE / softmax / SiLU expressions).
struct RowsParams {
    const float* x; int ldx;        // input rows (F32)
    const float* x2; int ldx2;      // ROWS_SILU_MUL: the `up` rows
    const float* nw; float eps;     // ROWS_RMS: norm weights (F32)
    int K, rows;
    int8_t* q; int ldq;             // Q8 codes, natural element order; ldq < 0: MFMA-tiled [row tile][blk][h][m][16 B]
    float* d; int ldd;              // block scales
    float* keep;                    // optional: the F32 value that was quantized (row-major, ld = K), taps
};
enum { ROWS_QUANT = 0, ROWS_RMS = 1, ROWS_SILU_MUL = 2 };

// One workgroup per row; a quad of lanes owns one Q8 block of 32 (Panama quantizeQ8_512, PTO:1684-1723).
template <int MODE>
__global__ __launch_bounds__(256) void rows_quant_kernel(RowsParams p) {
    __shared__ double red[8];
    const int row = blockIdx.x, T = blockDim.x;
    const float* x = p.x + (size_t)row * p.ldx;
    const int units = p.K / 8;
    float fs = 1.0f;
    if (MODE == ROWS_RMS) fs = rms_factor(x, p.K, p.eps, red);
    // gridDim.y workgroups share a row (blocks are independent; ROWS_RMS needs the whole row and is launched with y = 1)
    for (int unit = blockIdx.y * T + threadIdx.x; unit < units; unit += T * gridDim.y) {   // units, T multiples of 4: quads stay together
        const float4 xa = *(const float4*)(x + unit * 8), xb = *(const float4*)(x + unit * 8 + 4);
        float y[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
        if (MODE == ROWS_RMS) {
            float w[8];
            load8_norm(p.nw, unit * 8, w);
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = w[i] * (fs * y[i]);
        }
        if (MODE == ROWS_SILU_MUL) {
            const float* u = p.x2 + (size_t)row * p.ldx2 + unit * 8;
            const float4 ua = *(const float4*)u, ub = *(const float4*)(u + 4);
            const float uu[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = silu_ref(y[i]) * uu[i];   // MLPBlock.java:131-142
        }
        if (p.keep) {
#pragma unroll
            for (int i = 0; i < 8; i++) p.keep[(size_t)row * p.K + unit * 8 + i] = y[i];
        }
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
        i32x2 packed;
        packed.x = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
        packed.y = q[4] | (q[5] << 8) | (q[6] << 16) | (q[7] << 24);
        if (p.ldq < 0) {   // element e = unit*8 of block blk = unit/4: half h = (unit&3)>>1, 8-byte slot (unit&1)
            const size_t nblk = p.K / QB;
            const size_t off = ((((size_t)(row >> 5) * nblk + (unit >> 2)) * 2 + ((unit & 3) >> 1)) * 32 + (row & 31)) * 16 + (unit & 1) * 8;
            *(i32x2*)(p.q + off) = packed;
        } else {
            *(i32x2*)(p.q + (size_t)row * p.ldq + unit * 8) = packed;
        }
        if ((unit & 3) == 0) {
            if (p.ldq < 0) p.d[(((size_t)(row >> 5) * (p.K / QB)) + (unit >> 2)) * 32 + (row & 31)] = d;   // [row tile][blk][32 rows]
            else p.d[(size_t)row * p.ldd + (unit >> 2)] = d;
        }
    }
}

// BF16 models (config 4): the activation rows are RNE-rounded to BF16 instead (FloatConversions.java:35-60) -- same
// modes, output [rows][K] bf16 in p.q (ldq in elements).
template <int MODE>
__global__ __launch_bounds__(256) void rows_bf16_kernel(RowsParams p) {
    __shared__ double red[8];
    const int row = blockIdx.x, T = blockDim.x;
    const float* x = p.x + (size_t)row * p.ldx;
    const int units = p.K / 8;
    float fs = 1.0f;
    if (MODE == ROWS_RMS) fs = rms_factor(x, p.K, p.eps, red);
    uint16_t* out = (uint16_t*)p.q + (size_t)row * p.ldq;
    for (int unit = blockIdx.y * T + threadIdx.x; unit < units; unit += T * gridDim.y) {
        const float4 xa = *(const float4*)(x + unit * 8), xb = *(const float4*)(x + unit * 8 + 4);
        float y[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
        if (MODE == ROWS_RMS) {
            float w[8];
            load8_norm(p.nw, unit * 8, w);
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = w[i] * (fs * y[i]);
        }
        if (MODE == ROWS_SILU_MUL) {
            const float* u = p.x2 + (size_t)row * p.ldx2 + unit * 8;
            const float4 ua = *(const float4*)u, ub = *(const float4*)(u + 4);
            const float uu[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = silu_ref(y[i]) * uu[i];
        }
        i32x4 packed;
        packed.x = (int)f32_to_bf16(y[0]) | ((int)f32_to_bf16(y[1]) << 16);
        packed.y = (int)f32_to_bf16(y[2]) | ((int)f32_to_bf16(y[3]) << 16);
        packed.z = (int)f32_to_bf16(y[4]) | ((int)f32_to_bf16(y[5]) << 16);
        packed.w = (int)f32_to_bf16(y[6]) | ((int)f32_to_bf16(y[7]) << 16);
        if (p.ldq < 0) {   // MFMA order: unit = 8 consecutive k = one lane's operand: k slice unit/2, half unit&1
            const size_t nks = p.K / 16;
            ((i32x4*)p.q)[(((size_t)(row >> 5) * nks + (unit >> 1)) * 2 + (unit & 1)) * 32 + (row & 31)] = packed;
        } else {
            *(i32x4*)(out + unit * 8) = packed;
        }
    }
}

// Re-tile a row-major weight into MFMA order for the prefill GEMMs:
//   Q4   [N][K/2] (+ scales [N][K/32])  ->  wt [N/32][K/32][32 rows][16 B],  st [N/32][K/32][32]
//   BF16 [N][K]                         ->  wt [N/32][K/16][h][32 rows][8 values]
// At streaming rate, for both weight types (it also runs in front of every prefill GEMM when the MFMA-ordered copy is NOT kept
// resident, JH_TILED_COPY=transient):  a weight row is `nch` 16-byte chunks (Q4: one block of 32 nibbles; BF16: 8 values --
// the [k slice][h] pair is chunk index c8 = 2*ksl + h, so both layouts are dst = ((panel*nch + chunk)*32 + row)*16 B), the
// destination is [N/32][nch][32 rows][16 B] (+ Q4 scales [N/32][nch][32]).  One workgroup moves a 32-row x 64-chunk tile
// through LDS: reads are 1 KiB contiguous per wave (a row's 64 chunks), writes 1 KiB contiguous per wave (2 chunks x 32 rows).
// LDS rows are padded by one chunk so that the transposed ds_read_b128 (32 lanes = 32 rows of one chunk) is conflict-free.
static __global__ __launch_bounds__(256) void retile16_kernel(const i32x4* __restrict__ w, const float* __restrict__ ws, int N, int nch,
                                                       i32x4* __restrict__ wt, float* __restrict__ st) {
    __shared__ i32x4 tile[32 * 65];
    __shared__ float stile[32 * 65];
    const int panel = blockIdx.y, c0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cin = c0 + lane;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int r = wave + 4 * i;
        if (cin < nch) {
            const size_t src = (size_t)(panel * 32 + r) * nch + cin;
            tile[r * 65 + lane] = __builtin_nontemporal_load(w + src);
            if (ws) stile[r * 65 + lane] = __builtin_nontemporal_load(ws + src);
        }
    }
    __syncthreads();
    const int row = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int cl = 2 * (wave + 4 * i) + hh;        // chunk within the tile
        if (c0 + cl < nch) {
            const size_t dst = ((size_t)panel * nch + c0 + cl) * 32 + row;
            wt[dst] = tile[row * 65 + cl];
            if (ws) st[dst] = stile[row * 65 + cl];
        }
    }
}

static __global__ void embed_rows_kernel(const void* table, const float* scales, int dtype, const int* tokens, int E, float* x) {
    embed_row(table, scales, dtype, tokens[blockIdx.x], E, x + (size_t)blockIdx.x * E);
}

struct PrefillAttnParams {
    float* qkv; int ldqkv;        // [rows][A + 2*KV] F32: q | k | v per row (q is rotated in place)
    const float* rope;
    float* kv_base; long long page_elems;
    int rel_layer_in_page, ctx_per_page, cpp_shift;
    int n_heads, n_kv_heads, head_size, kv_head_offset;
    const int* start_pos;  // device word: position of the chunk's first row (read at run time => one captured graph per
    int rows;              //   chunk shape serves every chunk position)
    float scale;
    float* out; int ldo;          // [rows][A]
};

// RoPE of q (in place) and k, and the KV page writes, for every row of the chunk (CausalSelfAttention.java:199-286;
// the table offset is position*half + kvHead*headSize for q and k alike -- SURVEY.md 8a "RoPE quirk").
static __global__ __launch_bounds__(256) void rows_rope_kv_kernel(PrefillAttnParams p) {
    const int row = blockIdx.x, pos = p.start_pos[0] + row;
    const int HS = p.head_size, half = HS / 2, A = p.n_heads * HS, KV = p.n_kv_heads * HS, group = p.n_heads / p.n_kv_heads;
    float* r = p.qkv + (size_t)row * p.ldqkv;
    float* krow = (float*)kv_row(p, 0, pos, KV);
    float* vrow = (float*)kv_row(p, 1, pos, KV);
    const int npairs = (p.n_heads + p.n_kv_heads) * half;
    const int t0 = blockIdx.y * blockDim.x + threadIdx.x, tstep = blockDim.x * gridDim.y;   // gridDim.y workgroups per row
    for (int i = t0; i < npairs; i += tstep) {
        const int hh = i / half, d = i - hh * half;
        const bool isq = hh < p.n_heads;
        const int kvh = isq ? hh / group : hh - p.n_heads;
        const float* rf = p.rope + ((size_t)pos * half + (size_t)(kvh + p.kv_head_offset) * HS) * 2;
        float* v = isq ? r + (size_t)hh * HS : r + A + (size_t)kvh * HS;
        const float a = v[d], b = v[d + half], c = rf[2 * d], s = rf[2 * d + 1];
        const float r0 = a * c - b * s;
        const float r1 = a * s + b * c;
        if (isq) { v[d] = r0; v[d + half] = r1; }
        else { krow[(size_t)kvh * HS + d] = r0; krow[(size_t)kvh * HS + d + half] = r1; }
    }
    for (int i = t0; i < KV; i += tstep) vrow[i] = r[A + KV + i];
}

// Causal attention of one chunk row against positions [0, start_pos+row]: workgroup = (kv head, row), the GROUP query
// heads of the kv head share every K/V load.  Scores -> softmax ((float)exp in double, division) -> weighted V sum, as
// CausalSelfAttention.java:322-362 / attn_decode_kernel.  LDS: scores [GROUP][n].
constexpr int PF_THREADS = 256;
template <int HS, int GROUP>
__global__ __launch_bounds__(PF_THREADS) void attn_prefill_kernel(PrefillAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = PF_THREADS, NW = NT / 64, LPR = HS / 4, RPS = NT / LPR;
    const int kvh = blockIdx.x, row = blockIdx.y;
    const int pos = p.start_pos[0] + row, n = pos + 1;
    const int KV = p.n_kv_heads * HS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rsub = tid / LPR, c4 = tid % LPR;
    float* red = (float*)smem;                 // RPS*GROUP*HS
    float* sc = red + RPS * GROUP * HS;        // GROUP*n
    const float* qrow = p.qkv + (size_t)row * p.ldqkv + (size_t)kvh * GROUP * HS;
    float4 qv[GROUP];
#pragma unroll
    for (int gi = 0; gi < GROUP; gi++) qv[gi] = ((const float4*)(qrow + gi * HS))[c4];
    constexpr int UN = 4;   // K rows in flight per thread (branch-free clamped loads: one round trip per UN steps)
    for (int t0 = rsub; t0 < n; t0 += RPS * UN) {
        float4 kreg[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            int t = t0 + u * RPS;
            t = t < n ? t : n - 1;
            kreg[u] = ((const float4*)(kv_row(p, 0, t, KV) + (size_t)kvh * HS))[c4];
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int t = t0 + u * RPS;
#pragma unroll
            for (int gi = 0; gi < GROUP; gi++) {
                float s = qv[gi].x * kreg[u].x;
                s = fmaf(qv[gi].y, kreg[u].y, s);
                s = fmaf(qv[gi].z, kreg[u].z, s);
                s = fmaf(qv[gi].w, kreg[u].w, s);
                s = group_sum_last<LPR>(s);
                if (c4 == LPR - 1 && t < n) sc[gi * n + t] = s * p.scale;
            }
        }
    }
    __syncthreads();
    for (int gi = wave; gi < GROUP; gi += NW) {
        float m = -INFINITY;
        for (int t = lane; t < n; t += 64) m = fmaxf(m, sc[gi * n + t]);
        m = wave_max(m);
        float l = 0.0f;
        for (int t = lane; t < n; t += 64) {
            const float e = (float)exp((double)(sc[gi * n + t] - m));
            sc[gi * n + t] = e;
            l += e;
        }
        l = wave_sum(l);
        for (int t = lane; t < n; t += 64) sc[gi * n + t] = sc[gi * n + t] / l;
    }
    __syncthreads();
    float4 acc[GROUP];
#pragma unroll
    for (int gi = 0; gi < GROUP; gi++) acc[gi] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = rsub; t0 < n; t0 += RPS * UN) {
        float4 vreg[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            int t = t0 + u * RPS;
            t = t < n ? t : n - 1;
            vreg[u] = ((const float4*)(kv_row(p, 1, t, KV) + (size_t)kvh * HS))[c4];
        }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const int t = t0 + u * RPS;
            if (t < n) {
#pragma unroll
                for (int gi = 0; gi < GROUP; gi++) {
                    const float w = sc[gi * n + t];
                    acc[gi].x = fmaf(vreg[u].x, w, acc[gi].x);
                    acc[gi].y = fmaf(vreg[u].y, w, acc[gi].y);
                    acc[gi].z = fmaf(vreg[u].z, w, acc[gi].z);
                    acc[gi].w = fmaf(vreg[u].w, w, acc[gi].w);
                }
            }
        }
    }
#pragma unroll
    for (int gi = 0; gi < GROUP; gi++) ((float4*)(red + ((size_t)rsub * GROUP + gi) * HS))[c4] = acc[gi];
    __syncthreads();
    float* orow = p.out + (size_t)row * p.ldo + (size_t)kvh * GROUP * HS;
    for (int i = tid; i < GROUP * HS; i += NT) {
        float s = 0.0f;
#pragma unroll
        for (int rg = 0; rg < RPS; rg++) s += red[(size_t)rg * GROUP * HS + i];
        orow[i] = s;
    }
}

// ---- blockwise causal prefill attention on the matrix cores (reference loop CausalSelfAttention.java:199-357).
// attn_prefill_kernel above re-reads every K/V row once per query row (O(n^2) traffic: 1.5 s of an 8100-row prompt).  Here a
// workgroup owns a tile of 32 query rows of ONE kv head -- its GROUP waves are the GROUP query heads sharing that kv head, so a
// K/V tile staged in LDS is read from HBM/L2 once per 32 x GROUP query rows -- and walks the causal key range in tiles of 32:
//   S = Q K^T   v_mfma_f32_32x32x2_f32 (F32 in, F32 accumulate: exact products, a different summation order only), HS/2 per tile;
//   online softmax per row in the reference's float/double recipe ((float)exp((double)(s - max)), running max / sum, rescale);
//   O += P V    the same MFMA, HS/32 output tiles x 16 per key tile.
// MFMA operand layout (32x32x2): A lane l = A[m = l&31][k = l>>5], B lane l = B[k = l>>5][n = l&31]; D reg r of lane l =
// D[m = (r&3) + 8*(r>>2) + 4*(l>>5)][n = l&31].  The k index of an instruction is OURS to assign as long as A and B agree:
// instruction j of QK^T multiplies dims (j, HS/2 + j) so that a lane's Q operands are HS/2 CONTIGUOUS floats, instruction j of
// PV multiplies keys (j, 16 + j).  P goes from the D layout (lane = key column) to the A layout (lane = query row) through a
// wave-private LDS transpose.  Long contexts: `nsplit` workgroups share the key range of a query tile (each a contiguous run of
// key tiles), partial (O, max, sum) go to a workspace and attn_prefill_combine_kernel merges them in split order.
struct PrefillMfmaExtra {
    int nsplit;          // key-range splits per query tile
    float* ws_o;         // [rows][n_heads][nsplit][HS] unnormalised partial outputs (nsplit > 1)
    float* ws_ml;        // [rows][n_heads][nsplit][2]  (running max, running sum)
};
typedef float f32x16v __attribute__((ext_vector_type(16)));

template <int HS, int GROUP>
__global__ __launch_bounds__(GROUP * 64) void attn_prefill_mfma_kernel(PrefillAttnParams p, PrefillMfmaExtra e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = HS + 4;            // K tile row stride (floats): 16 consecutive lanes' ds_read_b128 hit distinct banks
    constexpr int SS = 33;                // per-wave transpose buffer row stride
    constexpr int NT = GROUP * 64;
    float* Kt = (float*)smem;             // [32][KS]
    float* Vt = Kt + 32 * KS;             // [32][HS]
    float* Sx = Vt + 32 * HS;             // [GROUP][32][SS]
    float* Cb = Sx + GROUP * 32 * SS;     // [GROUP][32]  per-row factors handed from the A layout to the D layout
    const int tid = threadIdx.x, lane = tid & 63, g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 31, kh = lane >> 5;
    const int qt = blockIdx.x / e.nsplit, split = blockIdx.x - qt * e.nsplit;
    const int kvh = blockIdx.y, head = kvh * GROUP + g;
    const int start = p.start_pos[0];
    const int KV = p.n_kv_heads * HS;
    const int row0 = qt * 32;
    int qrow = row0 + ml;
    qrow = qrow < p.rows ? qrow : p.rows - 1;          // rows beyond the chunk replicate the last row (never stored)
    const int qpos = start + qrow;
    // key tiles this query tile needs: keys 0 .. start + min(row0 + 31, rows - 1)
    const int last_row = (row0 + 31 < p.rows ? row0 + 31 : p.rows - 1);
    const int ntiles = (start + last_row) / 32 + 1;
    const int t_begin = (int)((long long)ntiles * split / e.nsplit), t_end = (int)((long long)ntiles * (split + 1) / e.nsplit);
    // Q operands: lane (m, kh) holds Q[row m][head][kh*HS/2 + j], j < HS/2 (roped in place by rows_rope_kv_kernel)
    float qa[HS / 2];
    {
        const float4* qp = (const float4*)(p.qkv + (size_t)qrow * p.ldqkv + (size_t)head * HS + kh * (HS / 2));
#pragma unroll
        for (int j = 0; j < HS / 8; j++) {
            const float4 v = qp[j];
            qa[4 * j] = v.x; qa[4 * j + 1] = v.y; qa[4 * j + 2] = v.z; qa[4 * j + 3] = v.w;
        }
    }
    f32x16v oacc[HS / 32];
#pragma unroll
    for (int dt = 0; dt < HS / 32; dt++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[dt][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    float* sx = Sx + g * 32 * SS;
    float* cb = Cb + g * 32;
    for (int kt = t_begin; kt < t_end; kt++) {
        __syncthreads();                                 // every wave is done reading the previous tile
        // ---- stage K and V rows kt*32 .. kt*32+31 of this kv head (positions beyond the newest row are clamped: masked below)
#pragma unroll 4
        for (int i = tid; i < 32 * (HS / 4); i += NT) {
            const int k = i / (HS / 4), d4 = i - k * (HS / 4);
            int t = kt * 32 + k;
            const int tmax = start + p.rows - 1;
            t = t < tmax ? t : tmax;
            const float4 kv4 = ((const float4*)(kv_row(p, 0, t, KV) + (size_t)kvh * HS))[d4];
            const float4 vv4 = ((const float4*)(kv_row(p, 1, t, KV) + (size_t)kvh * HS))[d4];
            *(float4*)(Kt + k * KS + 4 * d4) = kv4;
            *(float4*)(Vt + k * HS + 4 * d4) = vv4;
        }
        __syncthreads();
        // ---- S = Q K^T over this key tile: lane (n = key, kh) feeds K[key n][kh*HS/2 + j]
        f32x16v sacc;
#pragma unroll
        for (int r = 0; r < 16; r++) sacc[r] = 0.0f;
        {
            const float* kr = Kt + ml * KS + kh * (HS / 2);
#pragma unroll
            for (int j4 = 0; j4 < HS / 8; j4++) {
                const float4 b = *(const float4*)(kr + 4 * j4);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[4 * j4 + 0], b.x, sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[4 * j4 + 1], b.y, sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[4 * j4 + 2], b.z, sacc, 0, 0, 0);
                sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[4 * j4 + 3], b.w, sacc, 0, 0, 0);
            }
        }
        // ---- D layout (lane = key column ml, regs = rows) -> wave-private LDS, scaled (ops.scale after the dot, :332)
#pragma unroll
        for (int r = 0; r < 16; r++) sx[((r & 3) + 8 * (r >> 2) + 4 * kh) * SS + ml] = sacc[r] * p.scale;
        // ---- A layout: lane (m = ml, kh) owns row m, keys kh*16 .. kh*16+15 of the tile; causal mask; online softmax
        float sv[16];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int key = kt * 32 + kh * 16 + j;
            float v = sx[ml * SS + kh * 16 + j];
            v = key <= qpos ? v : -INFINITY;
            sv[j] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));                // the other half of the row's keys
        const float m_new = fmaxf(m_run, mx);
        float ls = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float pj = sv[j] == -INFINITY ? 0.0f : (float)exp((double)(sv[j] - m_new));
            sv[j] = pj;
            ls += pj;
        }
        ls += __shfl_xor(ls, 32);
        const float corr = m_run == -INFINITY ? 0.0f : (float)exp((double)(m_run - m_new));
        l_run = l_run * corr + ls;
        m_run = m_new;
        if (kh == 0) cb[ml] = corr;
        // ---- rescale O (D layout: reg r of lane (n, kh) is row (r&3) + 8*(r>>2) + 4*kh), then O += P V
        float cr[16];
#pragma unroll
        for (int r = 0; r < 16; r++) cr[r] = cb[(r & 3) + 8 * (r >> 2) + 4 * kh];
#pragma unroll
        for (int dt = 0; dt < HS / 32; dt++) {
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[dt][r] *= cr[r];
            const float* vr = Vt + (kh * 16) * HS + dt * 32 + ml;     // V[key kh*16 + j][dim dt*32 + ml]
#pragma unroll
            for (int j = 0; j < 16; j++) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[j], vr[j * HS], oacc[dt], 0, 0, 0);
        }
    }
    // ---- epilogue: rows of the D layout need their (max, sum): hand them over through the per-wave buffer
    __syncthreads();
    if (kh == 0) { cb[ml] = l_run; sx[ml] = m_run; }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int row = row0 + m;
        if (row >= p.rows) continue;
        const float l = cb[m];
        if (e.nsplit == 1) {
            float* orow = p.out + (size_t)row * p.ldo + (size_t)head * HS;
#pragma unroll
            for (int dt = 0; dt < HS / 32; dt++) orow[dt * 32 + ml] = oacc[dt][r] / l;
        } else {
            float* orow = e.ws_o + (((size_t)row * p.n_heads + head) * e.nsplit + split) * HS;
#pragma unroll
            for (int dt = 0; dt < HS / 32; dt++) orow[dt * 32 + ml] = oacc[dt][r];
            if (ml == 0) {
                float* mlp = e.ws_ml + (((size_t)row * p.n_heads + head) * e.nsplit + split) * 2;
                mlp[0] = sx[m];
                mlp[1] = l;
            }
        }
    }
}
static inline size_t prefill_mfma_lds(int hs, int group) { return ((size_t)32 * (hs + 4) + 32 * hs + (size_t)group * 32 * 33 + group * 32) * 4; }

// merge the key-range splits of attn_prefill_mfma_kernel: w_s = exp(m_s - M), out = sum_s w_s O_s / sum_s w_s l_s (split order)
static __global__ __launch_bounds__(128) void attn_prefill_combine_kernel(PrefillAttnParams p, PrefillMfmaExtra e) {
    const int row = blockIdx.x, head = blockIdx.y, HS = p.head_size;
    const float* ml = e.ws_ml + ((size_t)row * p.n_heads + head) * e.nsplit * 2;
    float M = -INFINITY;
    for (int s = 0; s < e.nsplit; s++) M = fmaxf(M, ml[2 * s]);
    float L = 0.0f;
    for (int s = 0; s < e.nsplit; s++) {
        const float w = ml[2 * s] == -INFINITY ? 0.0f : (float)exp((double)(ml[2 * s] - M));
        L += ml[2 * s + 1] * w;
    }
    const float* ob = e.ws_o + ((size_t)row * p.n_heads + head) * e.nsplit * HS;
    for (int d = threadIdx.x; d < HS; d += blockDim.x) {
        float o = 0.0f;
        for (int s = 0; s < e.nsplit; s++) {
            const float w = ml[2 * s] == -INFINITY ? 0.0f : (float)exp((double)(ml[2 * s] - M));
            o = fmaf(ob[(size_t)s * HS + d], w, o);
        }
        p.out[(size_t)row * p.ldo + (size_t)head * HS + d] = o / L;
    }
}

// ------------------------------------------------------------------------------------------------ Tier-1 generic kernels
// One wave per output element C[i, j]; lanes stride over K.  Correct for every offset/stride combination the
// reference's C entry points accept (nc/simd/vector_simd.h:22-38); used for M>1, windows, F32/BF16 operands.
enum { G_Q8Q4 = 0, G_F32Q4 = 1, G_F32 = 2, G_BF16 = 3, G_F32BF16 = 4 };
struct GemmParams {
    const void* a; const float* af; const void* b; const float* bf; float* r;
    int aoffset, boffset, roffset, m, n0, n, k, lda, ldaf, ldb, ldbf, ldc;
};
template <int KIND>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmParams p) {
    const int lane = threadIdx.x & 63;
    const long long widx = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (widx >= (long long)p.m * p.n) return;
    const int i = (int)(widx / p.n), j = p.n0 + (int)(widx % p.n);
    float acc = 0.0f;
    if (KIND == G_Q8Q4) {
        const int8_t* ar = (const int8_t*)p.a + (size_t)p.lda * i + p.aoffset;
        const float* afr = p.af + (size_t)p.ldaf * i + p.aoffset / QB;
        const uint8_t* br = (const uint8_t*)p.b + (size_t)p.ldb * j + p.boffset;
        const float* bfr = p.bf + (size_t)p.ldbf * j + (p.boffset * 2) / QB;
        for (int blk = lane; blk < p.k / QB; blk += 64) {
            int isum = 0;
            for (int t = 0; t < 16; t++) {
                const int bb = br[blk * 16 + t];
                isum += (int)ar[blk * 32 + t] * ((bb & 0x0F) - 8) + (int)ar[blk * 32 + t + 16] * (((bb >> 4) & 0x0F) - 8);
            }
            acc = fmaf(afr[blk] * bfr[blk], (float)isum, acc);
        }
    } else if (KIND == G_F32Q4) {
        const float* ar = (const float*)p.a + (size_t)p.lda * i + p.aoffset;
        const uint8_t* br = (const uint8_t*)p.b + (size_t)p.ldb * j + p.boffset;
        const float* bfr = p.bf + (size_t)p.ldbf * j + (p.boffset * 2) / QB;
        for (int blk = lane; blk < p.k / QB; blk += 64) {
            const float s = bfr[blk];
            for (int t = 0; t < 16; t++) {
                const int bb = br[blk * 16 + t];
                acc = fmaf(ar[blk * 32 + t], (float)((bb & 0x0F) - 8) * s, acc);
            }
            for (int t = 0; t < 16; t++) {
                const int bb = br[blk * 16 + t];
                acc = fmaf(ar[blk * 32 + t + 16], (float)(((bb >> 4) & 0x0F) - 8) * s, acc);
            }
        }
    } else if (KIND == G_F32) {
        const float* ar = (const float*)p.a + (size_t)p.lda * i + p.aoffset;
        const float* br = (const float*)p.b + (size_t)p.ldb * j + p.boffset;
        for (int kk = lane; kk < p.k; kk += 64) acc = fmaf(ar[kk], br[kk], acc);
    } else if (KIND == G_BF16) {
        const uint16_t* ar = (const uint16_t*)p.a + (size_t)p.lda * i + p.aoffset;
        const uint16_t* br = (const uint16_t*)p.b + (size_t)p.ldb * j + p.boffset;
        for (int kk = lane; kk < p.k; kk += 64) acc = fmaf(bf16_to_f32(ar[kk]), bf16_to_f32(br[kk]), acc);
    } else {
        const float* ar = (const float*)p.a + (size_t)p.lda * i + p.aoffset;
        const uint16_t* br = (const uint16_t*)p.b + (size_t)p.ldb * j + p.boffset;
        for (int kk = lane; kk < p.k; kk += 64) acc = fmaf(ar[kk], bf16_to_f32(br[kk]), acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) p.r[(size_t)p.ldc * i + j - p.roffset] = acc;
}

// element-wise (PTO:2281-2295, 2297-2325, 2083-2097, 2499-2514, 2593-2611)
enum { EW_ACC = 0, EW_MACC = 1, EW_SCALE = 2, EW_SAXPY = 3, EW_SILU_MUL = 4 };
template <int OP>
__global__ void ew_kernel(float* a, const float* b, float f, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (OP == EW_ACC) a[i] = a[i] + b[i];
    else if (OP == EW_MACC) a[i] = a[i] * b[i];
    else if (OP == EW_SCALE) a[i] = a[i] * f;
    else if (OP == EW_SAXPY) a[i] = fmaf(b[i], f, a[i]);
    else a[i] = silu_ref(a[i]) * b[i];
}
static __global__ void acc_q4_kernel(float* a, const uint8_t* nib, const float* sc, int offset, int n) {
    const int i = offset + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= offset + n) return;
    const int blk = i / 32, in = i % 32;
    const uint8_t b = nib[blk * 16 + (in & 15)];
    const int x = (in < 16) ? (b & 0x0F) - 8 : ((b >> 4) & 0x0F) - 8;
    a[i] = a[i] + (float)x * sc[blk];
}
// batched saxpy: thread per element, fma chain over rows in ascending order (PTO:2648-2698)
static __global__ void saxpy_batch_kernel(const float* alpha, const float* x, int ldx, float* y, int limit, int rows) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= limit) return;
    float acc = y[t];
    for (int n = 0; n < rows; n++) acc = fmaf(x[(size_t)n * ldx + t], alpha[n], acc);
    y[t] = acc;
}
// quantize F32 -> I8 (PTO:1684-1723): one 32-lane half-wave per block
static __global__ void quantize_q8_kernel(const float* x, int rows, int ldx, int offset, int length, int8_t* q, int ldq,
                                   float* d, int ldd) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int hw = gid >> 5, l = gid & 31;
    const int bpr = length / QB;
    if (hw >= rows * bpr) return;
    const int r = hw / bpr, blk = hw % bpr;
    const int e = offset + blk * QB + l;
    const float y = x[(size_t)r * ldx + e];
    float amax = fabsf(y);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float dd = amax / 127.0f;
    const float id = (amax != 0.0f) ? 127.0f / amax : 0.0f;
    float v = y * id;
    v = v + 0.5f;
    q[(size_t)r * ldq + e] = (int8_t)f2b(v);
    if (l == 0) d[(size_t)r * ldd + e / QB] = dd;
}
static __global__ void widen_bf16_kernel(const uint16_t* in, long long n, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bf16_to_f32(in[i]);
}
static __global__ void quantize_bf16_kernel(const float* x, long long n, uint16_t* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f32_to_bf16(x[i]);
}
// BF16 result tensor of a Tier-1 GEMM: out[i*ld + j] = bf16(in[i*ld + j]) for j < n, one grid row per matrix row
static __global__ void store_bf16_2d_kernel(const float* in, uint16_t* out, int n, int ld) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[(size_t)blockIdx.y * ld + j] = f32_to_bf16(in[(size_t)blockIdx.y * ld + j]);
}
// RMSNorm.forward (core/model/RMSNorm.java:33-56), single workgroup per row
static __global__ __launch_bounds__(1024) void rmsnorm_kernel(const float* x, const float* w, float adj, int n, float eps,
                                                       float* out) {
    __shared__ double red[32];
    double ss = 0.0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) { const float v = x[j]; ss += (double)(v * v); }
    ss = block_sum_d(ss, red);
    ss /= (double)n;
    ss += (double)eps;
    ss = 1.0 / sqrt(ss);
    const float fs = (float)ss;
    for (int j = threadIdx.x; j < n; j += blockDim.x) out[j] = (adj + w[j]) * (fs * x[j]);
}
// VectorMath.softMax (core/math/VectorMath.java:69-90), single workgroup
// LayerNorm.forward (core/model/LayerNorm.java:41-67), GPT-2's norm: FLOAT sums accumulated in index order (one lane
// walks the row so the running sums round exactly as the Java loop's), var = sumSq/E - mean^2,
// 1/(float)sqrt(var+eps), then ((x-mean)*inv)*w + b with no fused multiply-add.  One wave per row.
static __global__ __launch_bounds__(64) void layernorm_kernel(const float* x, const float* w, const float* b, int ld, int offset, int length,
                                                       int divisor, float eps, float* out) {
    const float* row = x + (size_t)blockIdx.x * ld;
    float* orow = out + (size_t)blockIdx.x * ld;
    __shared__ float stats[2];
    if (threadIdx.x == 0) {
        float sum = 0.0f, sumsq = 0.0f;
        for (int i = offset; i < offset + length; i++) {
            const float v = row[i];
            sum += v;
            sumsq += v * v;
        }
        const float mean = sum / (float)divisor;
        const float variance = sumsq / (float)divisor - mean * mean;
        stats[0] = mean;
        stats[1] = 1.0f / (float)sqrt((double)(variance + eps));
    }
    __syncthreads();
    const float mean = stats[0], inv = stats[1];
    for (int i = offset + threadIdx.x; i < offset + length; i += blockDim.x) orow[i] = (row[i] - mean) * inv * w[i] + b[i];
}
// ActivationFunction.eval GELU (core/math/ActivationFunction.java:32-34): tanh approximation evaluated in double
static __global__ void gelu_kernel(float* x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = (double)x[i];
    x[i] = (float)(0.5 * v * (1.0 + tanh(sqrt(2.0 / 3.14159265358979323846) * (v + 0.044715 * pow(v, 3.0)))));
}
static __global__ __launch_bounds__(1024) void softmax_kernel(float* x, int offset, int length) {
    __shared__ float redf[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float m = -INFINITY;
    for (int i = offset + threadIdx.x; i < offset + length; i += blockDim.x) m = fmaxf(m, x[i]);
    m = wave_max(m);
    if (lane == 0) redf[wave] = m;
    __syncthreads();
    m = redf[0];
    for (int i = 1; i < nw; i++) m = fmaxf(m, redf[i]);
    __syncthreads();
    float s = 0.0f;
    for (int i = offset + threadIdx.x; i < offset + length; i += blockDim.x) {
        const float e = (float)exp((double)(x[i] - m));
        x[i] = e;
        s += e;
    }
    s = wave_sum(s);
    if (lane == 0) redf[wave] = s;
    __syncthreads();
    s = 0.0f;
    for (int i = 0; i < nw; i++) s += redf[i];
    for (int i = threadIdx.x; i < offset + length; i += blockDim.x) x[i] = x[i] / s;  // reference divides from index 0
}
// RoPE rotation, GQA branch (core/model/CausalSelfAttention.java:247-286)
static __global__ void rope_kernel(float* q, float* k, const float* rope, int position, int n_heads, int n_kv_heads, int hs) {
    const int half = hs / 2, group = n_heads / n_kv_heads;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int poffset = position * half;
    if (i < n_heads * half) {
        const int h = i / half, d = i % half;
        const int g = (h / group) * hs + d;
        const float fcr = rope[(size_t)(poffset + g) * 2], fci = rope[(size_t)(poffset + g) * 2 + 1];
        const float q0 = q[h * hs + d], q1 = q[h * hs + d + half];
        q[h * hs + d] = q0 * fcr - q1 * fci;
        q[h * hs + d + half] = q0 * fci + q1 * fcr;
    } else if (i < (n_heads + n_kv_heads) * half) {
        const int ii = i - n_heads * half;
        const int h = ii / half, d = ii % half;
        const int g = h * hs + d;
        const float fcr = rope[(size_t)(poffset + g) * 2], fci = rope[(size_t)(poffset + g) * 2 + 1];
        const float k0 = k[h * hs + d], k1 = k[h * hs + d + half];
        k[h * hs + d] = k0 * fcr - k1 * fci;
        k[h * hs + d + half] = k0 * fci + k1 * fcr;
    }
}

}  // namespace jh
