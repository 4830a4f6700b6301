// jh_strict.h -- the FIRST implementation of the reference-order kernels, kept as the cross-check of jh_p16.h (JH_STRICT_LEGACY=1):
// byte-granular weight loads, one workgroup per query head -- simple enough to audit line by line against the oracle, 5x slower.
// "strict order" kernels: the decode path with every float accumulation performed in EXACTLY the
// order of the reference's Panama AVX-512 provider, so that results are bit-identical to a plain-C restatement of that
// provider (what the parity tests compare against) instead of merely within the Q8 noise floor.  Selected per session
// (jh_session_set_strict / JH_STRICT_ORDER=1).  Purpose: prove that the only difference between the fast kernels
// (jh_kernels.h) and the reference is float summation order -- a verification mode, ~2-4x slower than the fast path.
//
// Panama-512 order (FloatVector.SPECIES_512 = 16 float lanes):
//   I8 x Q4  (GemmerI8Q4_512, PTO:807-850):  lane t:  acc_t = fma(da*sb, (float)(short)(lo_t*a[t] + hi_t*a[t+16]), acc_t)
//            over Q blocks in ascending K, then reduceLanes(ADD) = the halving tree (v[i]+v[i+8], +4, +2, +1).
//   F32 x Q4 (GemmerF32Q4_512, PTO:336-374): acc_t = fma(a[t], (float)(lo_t-8)*s, acc_t); acc_t = fma(a[t+16], (float)(hi_t-8)*s, acc_t)
//   F32 x F32 (GemmerF32, PTO:1086-1102):    acc_t = fma(a[l+t], b[l+t], acc_t) for l = 0,16,...; same tree.
//   softMax (VectorMath.java:69-90): float sum of exp in index order;  saxpy over V (PTO:2593-2611, 2648-2698):
//            one fma chain per output element over positions in ascending order.
// GPU mapping: a 16-lane DPP row plays the 16 SIMD lanes; a wave64 therefore serves 4 output rows at a time.
#pragma once
#include "jh_p16.h"   // row16_tree_sum

namespace jh {

// ---- I8 x Q4 GEMV, Panama order.  Same prologues / epilogues as gemv_i8q4_kernel.  Lane (r, t) of a wave: output row
// group r (0..3), SIMD lane t (0..15); it reads byte t of every 16-byte Q4 block of its row (low nibble = element t, high
// nibble = element t+16) and the matching activation codes from LDS.
template <int PRO, int EPI>
__global__ __launch_bounds__(256) void gemv_i8q4_strict_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nblk = p.K / QB;
    const ActI8 a = carve_i8(smem, nblk);
    ActRegsT<UMaxFor<PRO>::v> ar;
    stage_issue<PRO>(p, ar);
    stage_finish<PRO>(p, a, ar);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane >> 4, t = lane & 15;
    const int8_t* alo = (const int8_t*)a.lo + t;   // a[blk*32 + t]      at alo[blk*16]
    const int8_t* ahi = (const int8_t*)a.hi + t;   // a[blk*32 + 16 + t] at ahi[blk*16]
    const int nunits = p.nrows;                    // EPI_SILU_MUL: hidden units (gate row j and up row j)
    for (int j0 = (blockIdx.x * nwaves + wave) * 4; j0 < nunits; j0 += gridDim.x * nwaves * 4) {
        const int j = j0 + r;
        const int jc = j < nunits ? j : nunits - 1;   // clamped: idle row groups recompute the last row, never store
        float res[2] = {0.0f, 0.0f};
#pragma unroll
        for (int which = 0; which < (EPI == EPI_SILU_MUL ? 2 : 1); which++) {
            const uint8_t* wrow = (which ? p.w2 : p.w) + (size_t)jc * p.ldb + t;
            const float* srow = (which ? p.ws2 : p.ws) + (size_t)jc * p.ldbf;
            float acc = 0.0f;
#pragma unroll 8
            for (int blk = 0; blk < nblk; blk++) {
                const int b = wrow[(size_t)blk * 16];
                const int lo = (b & 0x0F) - 8, hi = ((b >> 4) & 0x0F) - 8;
                const int isum = lo * (int)alo[blk * 16] + hi * (int)ahi[blk * 16];   // |.| <= 2032: the (short) cast is exact
                const float scale = a.d[blk] * srow[blk];                             // af * bf  (PTO:819)
                acc = fmaf(scale, (float)isum, acc);
            }
            res[which] = row16_tree_sum(acc);
        }
        if (t == 0 && j < nunits) {
            if (EPI == EPI_SILU_MUL) p.out[j] = silu_ref(res[0]) * res[1];
            else if (EPI == EPI_RESID) p.out[j] = res[0] + p.resid[j];
            else p.out[j] = res[0];
        }
    }
}

// ---- F32 x Q4 GEMV (LM head), Panama order, with the final RMSNorm prologue and per-workgroup argmax partials of
// gemv_f32q4_kernel (strict >, lowest index first: AbstractModel.java:455-469).
template <int PRO>
__global__ __launch_bounds__(256) void gemv_f32q4_strict_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, nblk = K / QB;
    float* y = (float*)smem;                 // [K] the (normed) activation row
    double* red = (double*)(y + K);          // [32]
    float* bestv = (float*)(red + 32);       // [16]
    int* besti = (int*)(bestv + 16);         // [16]
    float fs = 1.0f;
    if (PRO == PRO_RMS_F32) fs = rms_factor(p.x, K, p.eps, red);
    for (int e = threadIdx.x; e < K; e += blockDim.x) y[e] = (PRO == PRO_RMS_F32) ? p.nw[e] * (fs * p.x[e]) : p.x[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int r = lane >> 4, t = lane & 15;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int j0 = (blockIdx.x * nwaves + wave) * 4; j0 < p.nrows; j0 += gridDim.x * nwaves * 4) {
        const int j = j0 + r;
        const int jc = j < p.nrows ? j : p.nrows - 1;
        const uint8_t* wrow = p.w + (size_t)jc * p.ldb + t;
        const float* srow = p.ws + (size_t)jc * p.ldbf;
        float acc = 0.0f;
#pragma unroll 8
        for (int blk = 0; blk < nblk; blk++) {
            const int b = wrow[(size_t)blk * 16];
            const float scale = srow[blk];
            const float low = (float)((b & 0x0F) - 8) * scale;          // dequantize first (PTO:350-358)
            const float high = (float)(((b >> 4) & 0x0F) - 8) * scale;
            acc = fmaf(y[blk * 32 + t], low, acc);
            acc = fmaf(y[blk * 32 + 16 + t], high, acc);
        }
        acc = row16_tree_sum(acc);
        if (j < p.nrows) {
            if (t == 0) p.out[j] = acc;
            if (acc > bv) { bv = acc; bi = j; }   // rows ascend within a lane: strict > keeps the first
        }
    }
    if (p.amax_part) {
        // per-lane bests -> workgroup best (value desc, index asc)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { bestv[wave] = bv; besti[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < nwaves; w++)
                if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) { bv = bestv[w]; bi = besti[w]; }
            p.amax_part[blockIdx.x] = bv;
            p.amax_idx[blockIdx.x] = bi;
        }
    }
}
static inline size_t lds_bytes_f32_strict(int K) { return (size_t)K * 4 + 32 * 8 + 16 * 4 + 16 * 4; }

// ---- decode attention, reference order (CausalSelfAttention.java:199-357).  One workgroup per QUERY head:
//   KV row write + RoPE (q head, and the kv head's new k row: written to the page by the group's first head only; every
//   head of the group rotates it locally, bit-identically, so nobody reads a row another workgroup is writing);
//   scores[t] = GemmerF32 16-lane dot (fma over 16-element steps, halving tree) * attentionScale;
//   softMax: max, (float)exp((double)(x-max)), FLOAT sum in index order, division;
//   value[d] = fma chain over positions 0..pos (saxpy per position, PTO:2648-2698).
__global__ __launch_bounds__(256) void attn_strict_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int HS = p.head_size, half = HS / 2, group = p.n_heads / p.n_kv_heads;
    const int h = blockIdx.x, kvh = h / group;
    const int pos = p.st->pos, n = pos + 1;
    const int KV = p.n_kv_heads * HS, A = p.n_heads * HS;
    const int tid = threadIdx.x, NT = blockDim.x;
    float* q = (float*)smem;      // [HS] roped q of this head
    float* knew = q + HS;         // [HS] roped k of the new position (this kv head)
    float* vnew = knew + HS;      // [HS]
    float* redf = vnew + HS;      // [16]
    float* sc = redf + 16;        // [n]
    const float* rf = p.rope + ((size_t)pos * half + (size_t)(kvh + p.kv_head_offset) * HS) * 2;
    for (int d = tid; d < half; d += NT) {
        const float c = rf[2 * d], s = rf[2 * d + 1];
        const float* qh = p.qkv + (size_t)h * HS;
        const float q0 = qh[d], q1 = qh[d + half];
        const float r0 = q0 * c - q1 * s, r1 = q0 * s + q1 * c;   // contraction off: mul, mul, sub / add as in Java
        q[d] = r0; q[d + half] = r1;
        if (p.tap_q) { p.tap_q[(size_t)h * HS + d] = r0; p.tap_q[(size_t)h * HS + d + half] = r1; }
        const float* kh = p.qkv + A + (size_t)kvh * HS;
        const float k0 = kh[d], k1 = kh[d + half];
        const float s0 = k0 * c - k1 * s, s1 = k0 * s + k1 * c;
        knew[d] = s0; knew[d + half] = s1;
        if (h % group == 0) {
            float* kdst = (float*)kv_row(p, 0, pos, KV) + (size_t)kvh * HS;
            kdst[d] = s0; kdst[d + half] = s1;
        }
    }
    for (int d = tid; d < HS; d += NT) {
        const float v = p.qkv[A + KV + (size_t)kvh * HS + d];
        vnew[d] = v;
        if (h % group == 0) ((float*)kv_row(p, 1, pos, KV) + (size_t)kvh * HS)[d] = v;
    }
    __syncthreads();
    // scores: a 16-lane row per position
    const int l = tid & 15;
    for (int t0 = 0; t0 < n; t0 += NT / 16) {
        const int t = t0 + (tid >> 4);
        const int tc = t < n ? t : n - 1;
        const float* krow = kv_row(p, 0, tc, KV) + (size_t)kvh * HS;
        float acc = 0.0f;
        for (int c = 0; c < HS; c += 16) {
            const float kv = (tc == pos) ? knew[c + l] : krow[c + l];
            acc = fmaf(q[c + l], kv, acc);
        }
        acc = row16_tree_sum(acc);
        if (l == 0 && t < n) sc[t] = acc * p.scale;   // ops.scale after the dot (:332)
    }
    __syncthreads();
    // max (order-free), exp, sequential float sum, division
    float m = -INFINITY;
    for (int t = tid; t < n; t += NT) m = fmaxf(m, sc[t]);
    m = wave_max(m);
    if ((tid & 63) == 0) redf[tid >> 6] = m;
    __syncthreads();
    m = redf[0];
    for (int w = 1; w < NT / 64; w++) m = fmaxf(m, redf[w]);
    __syncthreads();
    for (int t = tid; t < n; t += NT) sc[t] = (float)exp((double)(sc[t] - m));
    __syncthreads();
    if (tid == 0) {
        float sum = 0.0f;
        for (int t = 0; t < n; t++) sum += sc[t];
        redf[8] = sum;
    }
    __syncthreads();
    const float sum = redf[8];
    for (int t = tid; t < n; t += NT) sc[t] = sc[t] / sum;
    __syncthreads();
    // value = sum_t w[t] * V[t]: one fma chain per element, positions ascending
    for (int d = tid; d < HS; d += NT) {
        float acc = 0.0f;
#pragma unroll 4
        for (int t = 0; t < pos; t++) acc = fmaf(kv_row(p, 1, t, KV)[(size_t)kvh * HS + d], sc[t], acc);
        acc = fmaf(vnew[d], sc[pos], acc);
        p.outf[(size_t)h * HS + d] = acc;
    }
}
static inline size_t lds_bytes_attn_strict(int head_size, int max_ctx) { return ((size_t)3 * head_size + 16 + (size_t)max_ctx) * 4; }

}  // namespace jh
