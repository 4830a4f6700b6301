// jh_bf16r.h -- dense BF16 weights (Mistral-7B, BASELINE configs[3]) with every float accumulation in the order of the
// reference's Panama AVX-512 provider ("reference order"), so that a BF16 session can be bit-identical to the restated
// provider like the JQ4 sessions of jh_p16.h / jh_t16.h.
//
// What has to be reproduced (FloatVector.SPECIES_512 = 16 float lanes, ShortVector.SPECIES_512 = 32 shorts per step):
//   BF16 x BF16 (GemmerBF16 1x1, PTO:1279-1311):  per 32-element step l:  acc_t = fma(a[l+t], b[l+t], acc_t);
//            acc_t = fma(a[l+16+t], b[l+16+t], acc_t)   (convertShape part 0 / part 1, both operands widened by << 16),
//            then reduceLanes(ADD) = the halving tree (row16_tree_sum).  The activation row was rounded to BF16 before
//            (RNE, PTO:1624-1628 -> FloatConversions.float32ToBFloat16).
//   F32 x BF16 (GemmerF32BF16 1x1, PTO:1511-1538): the same chains with the UN-rounded F32 row (LM head,
//            AbstractModel.java:443-449).
//   The batch form (GemmerBF16 with M rows, AbstractModel.batchForward) keeps one 16-lane accumulator per (prompt row,
//   weight row): every pair is the GEMV's chain, batching only shares the weight side.
//
// MI355X mapping (same "p16" idea as jh_p16.h: a 16-lane DPP row plays the 16 Panama lanes, a wave64 serves 4 weight rows):
//   * lane (r, t) owns chain t of weight row 4q + r.  In the checkpoint's row-major layout a 16-byte load is 8 consecutive
//     elements = 8 DIFFERENT chains, so the kernels read a resident copy in "BF16T" order made once per weight
//     (bf16t_pack_kernel): inside every group of 128 elements (256 bytes of a row) the 16-byte chunk t holds
//     dword i = (e[32i + t], e[32i + 16 + t]) for the group's four 32-element steps i = 0..3 -- one load per lane IS the
//     next 8 links of its chain, in order; a wave instruction still moves 4 x 256 contiguous bytes;
//   * widening is one shift / one mask per element, the chain one v_fmac_f32 per element, pinned in program order;
//   * the activation row lives in LDS as F32 (already BF16-rounded and widened for BF16 x BF16) in the same pair order:
//     float4 entry (p, t) = (y[64p + t], y[64p + 16 + t], y[64p + 32 + t], y[64p + 48 + t]) -- two conflict-free
//     ds_read_b128 per group, broadcast to the wave's four 16-lane rows;
//   * decode is HBM-bound (2 bytes per weight): the VALU work (2 ops per weight) is ~1/5 of the stream time, so the
//     reference order costs nothing against the order-free kernel (gemv_bf16_kernel);
//   * the M-row form (gemm_bf16r_kernel) keeps 4 row quads x 8 prompt rows of accumulators per lane, the activations of
//     an 8-row tile staged through LDS in K chunks; row tiles of one weight slice are neighbours on one XCD (L2 reuse).
// Compiled with -ffp-contract=off like the rest: every FMA is explicit.
#pragma once
#include "jh_p16.h"

namespace jh {

constexpr int BF16R_GROUP = 128;   // elements of a row per 16-lane x 16-byte load
static inline size_t bf16t_row_bytes(int K) { return (size_t)((K + BF16R_GROUP - 1) / BF16R_GROUP) * 256; }

// BF16T copy of a row-major BF16 weight [nrows, ldw elements]: one thread per output 16-byte chunk; steps past K hold zeros
// (never multiplied: the kernels stop at the row's last step).
static __global__ __launch_bounds__(256) void bf16t_pack_kernel(const uint16_t* __restrict__ w, int nrows, int K, int ldw, uint8_t* __restrict__ out) {
    const int G = (K + BF16R_GROUP - 1) / BF16R_GROUP;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)nrows * G * 16) return;
    const int t = (int)(idx & 15);
    const long long rg = idx >> 4;
    const int g = (int)(rg % G);
    const long long row = rg / G;
    const uint16_t* src = w + (size_t)row * ldw + (size_t)g * BF16R_GROUP;
    i32x4 v = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int e = g * BF16R_GROUP + 32 * i;
        if (e < K) v[i] = (int)((unsigned)src[32 * i + t] | ((unsigned)src[32 * i + 16 + t] << 16));
    }
    ((i32x4*)out)[((row >> 2) * G + g) * 64 + (row & 3) * 16 + t] = v;   // the rows of a quad interleaved per group, as in the P16T copies (jh_p16.h)
}

// ------------------------------------------------------------------------------------------------ activation row in LDS
struct ActBFR {
    f32x4* a4;     // [2 * G][16]: entry (p, t) = y[64p + t], y[64p + 16 + t], y[64p + 32 + t], y[64p + 48 + t]
    double* red;   // [32]
    float* bestv;  // [16]
    int* besti;    // [16]
};
__device__ __forceinline__ ActBFR carve_bfr(char* smem, int K) {
    const int G = (K + BF16R_GROUP - 1) / BF16R_GROUP;
    ActBFR a;
    a.a4 = (f32x4*)smem;
    a.red = (double*)(a.a4 + (size_t)G * 32);
    a.bestv = (float*)(a.red + 32);
    a.besti = (int*)(a.bestv + 16);
    return a;
}
static inline size_t lds_bytes_bfr(int K) { return (size_t)((K + BF16R_GROUP - 1) / BF16R_GROUP) * 512 + 32 * 8 + 16 * 4 + 16 * 4; }
// float index of element e inside the pair-ordered row
__device__ __forceinline__ int bfr_slot(int e) {
    const int b = e >> 5, half = (e >> 4) & 1, t = e & 15;
    return (((b >> 1) * 16 + t) << 2) + ((b & 1) << 1) + half;
}

// the 8 links one 16-byte chunk adds to a chain (NS = valid 32-element steps of the group, 4 except in a short last group)
template <int NS>
__device__ __forceinline__ void bfr_group(const i32x4& x, const f32x4* af, float& acc) {
    const f32x4 a0 = af[0];
    if (NS >= 1) { fmac_pinned(acc, a0.x, __int_as_float(x.x << 16)); fmac_pinned(acc, a0.y, __int_as_float(x.x & (int)0xffff0000)); }
    if (NS >= 2) { fmac_pinned(acc, a0.z, __int_as_float(x.y << 16)); fmac_pinned(acc, a0.w, __int_as_float(x.y & (int)0xffff0000)); }
    if (NS >= 3) {
        const f32x4 a1 = af[16];
        fmac_pinned(acc, a1.x, __int_as_float(x.z << 16)); fmac_pinned(acc, a1.y, __int_as_float(x.z & (int)0xffff0000));
        if (NS >= 4) { fmac_pinned(acc, a1.z, __int_as_float(x.w << 16)); fmac_pinned(acc, a1.w, __int_as_float(x.w & (int)0xffff0000)); }
    }
}

// ------------------------------------------------------------------------------------------------ GEMV (decode), reference order
// PRO: PROB_RMS_BF16 (RMSNorm, round to BF16), PROB_QUANT_BF16 (round to BF16), PROB_RMS_F32 (RMSNorm, keep F32: LM head), PROB_F32.
// Work split as gemv_i8q4_p16_kernel: waves [0, tw) own `per` row quads each, NP passes per quad (gate then up for EPI_SILU_MUL),
// G groups per pass streamed through a ring of D prefetched chunks (the host picks D | G).
template <int PRO, int EPI, bool ARGMAX, int D, int UM>
__global__ __launch_bounds__(P16_THREADS) void gemv_bf16r_kernel(GemvParams p, int per, int tw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool RMS = (PRO == PROB_RMS_BF16 || PRO == PROB_RMS_F32), ROUND = (PRO == PROB_RMS_BF16 || PRO == PROB_QUANT_BF16);
    const int K = p.K, G = (K + BF16R_GROUP - 1) / BF16R_GROUP;   // host: G % D == 0, K % 32 == 0, K <= UM * 4096
    const ActBFR a = carve_bfr(smem, K);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int r = lane >> 4, t = lane & 15;
    constexpr int NP = (EPI == EPI_SILU_MUL) ? 2 : 1;
    const int nq = (p.nrows + 3) >> 2;
    int q0 = wave < tw ? (blockIdx.x * tw + wave) * per : nq;
    if (q0 > nq) q0 = nq;
    int q1 = q0 + per;
    if (q1 > nq) q1 = nq;
    const int items = (q1 - q0) * NP * G;

    float xv[UM][8], wv[UM][8];
    auto stage_issue = [&]() __attribute__((always_inline)) {   // activation row (+ norm weights): one round trip, branch-free
        const int units = K / 8;
#pragma unroll
        for (int u = 0; u < UM; u++) {
            int unit = threadIdx.x + u * P16_THREADS;
            unit = unit < units ? unit : units - 1;
            const float4 xa = *(const float4*)(p.x + unit * 8), xb = *(const float4*)(p.x + unit * 8 + 4);
            xv[u][0] = xa.x; xv[u][1] = xa.y; xv[u][2] = xa.z; xv[u][3] = xa.w;
            xv[u][4] = xb.x; xv[u][5] = xb.y; xv[u][6] = xb.z; xv[u][7] = xb.w;
            if (RMS) load8_norm(p.nw, unit * 8, wv[u]);
        }
    };
    auto stage_finish = [&]() __attribute__((always_inline)) {
        const int units = K / 8;
        float fs = 1.0f;
        if (RMS) {   // RMSNorm.java:41-49: float squares, double sum, /E, +eps, 1/sqrt in double
            double ss = 0.0;
#pragma unroll
            for (int u = 0; u < UM; u++)
                if ((int)threadIdx.x + u * P16_THREADS < units)
#pragma unroll
                    for (int i = 0; i < 8; i++) ss += (double)(xv[u][i] * xv[u][i]);
            ss = block_sum_d(ss, a.red);
            ss /= (double)K;
            ss += (double)p.eps;
            ss = 1.0 / sqrt(ss);
            fs = (float)ss;
        }
        float* af = (float*)a.a4;
#pragma unroll
        for (int u = 0; u < UM; u++) {
            const int unit = threadIdx.x + u * P16_THREADS;
            if (unit < units) {
                const int s0 = bfr_slot(unit * 8);          // 8 consecutive elements: t = t0 .. t0 + 7 of one half-step, 16 bytes apart
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float y = RMS ? wv[u][i] * (fs * xv[u][i]) : xv[u][i];
                    if (ROUND) y = bf16_to_f32(f32_to_bf16(y));   // quantizeBF16 (RNE), widened again
                    af[s0 + 4 * i] = y;
                }
            }
        }
        lds_barrier();
    };

    i32x4 wq[D];
    int lq = q0, lpass = 0, lg = 0;
    const uint8_t* wrow;
    auto set_row = [&]() __attribute__((always_inline)) {
        int row = 4 * lq + r;
        row = row < p.nrows ? row : p.nrows - 1;
        wrow = p16t_row_ptr((NP == 2 && lpass) ? p.w2 : p.w, row, p.ldb);
    };
    set_row();
    auto issue = [&](i32x4& w) __attribute__((always_inline)) {
        w = __builtin_nontemporal_load((const i32x4*)wrow + 64 * lg + t);   // BF16T copy: chunk t of group lg (1 KiB contiguous per wave instruction)
        if (++lg == G) {
            lg = 0;
            if (++lpass == NP) { lpass = 0; ++lq; }
            set_row();
        }
    };
    float bestv = -INFINITY;
    int besti = 0x7fffffff;
    if (items == 0) {
        // helper wave: its own copy of the prologue; must not join the streaming path before the argmax merge (see gemv_i8q4_p16_kernel)
        stage_issue();
        stage_finish();
    } else {
        stage_issue();                                          // activation loads first: vmcnt retires oldest-first
        auto resid_of_batch = [&](int qb) __attribute__((always_inline)) {
            int row = 4 * (qb + t) + r;
            row = row < p.nrows ? row : p.nrows - 1;
            return p.resid[row];
        };
        float rv = 0.0f;
        if (EPI == EPI_RESID) rv = resid_of_batch(q0);
#pragma unroll
        for (int d = 0; d < D; d++) {
            issue(wq[d]);
            __builtin_amdgcn_sched_barrier(0);                  // request order = consumption order
        }
        stage_finish();

        float acc = 0.0f, gres = 0.0f, gsel = 0.0f, usel = 0.0f;
        int cq = q0, cpass = 0, cg = 0;
        const int last_ns = (K - (G - 1) * BF16R_GROUP) / 32;   // steps of the row's last group (1..4)
        auto compute = [&](const i32x4& x, int g, bool can_be_short) __attribute__((always_inline)) {
            const f32x4* af = a.a4 + (size_t)(2 * g) * 16 + t;
            if (can_be_short && g == G - 1 && last_ns < 4) {
                if (last_ns == 1) bfr_group<1>(x, af, acc);
                else if (last_ns == 2) bfr_group<2>(x, af, acc);
                else bfr_group<3>(x, af, acc);
            } else {
                bfr_group<4>(x, af, acc);
            }
        };
        auto pass_end = [&]() __attribute__((always_inline)) {
            const float res = row16_tree_sum(acc);
            acc = 0.0f;
            if (EPI == EPI_SILU_MUL && cpass == 0) {
                gres = res;
                cpass = 1;
                return;
            }
            cpass = 0;
            const int row0 = 4 * cq + r;
            if (ARGMAX && row0 < p.nrows && res > bestv) { bestv = res; besti = row0; }   // rows ascend within a lane: strict > keeps the first
            const int n = (cq - q0) & 15;
            if (t == n) { gsel = gres; usel = res; }
            if (n == 15 || cq + 1 == q1) {                      // results parked in lane t == n, stored once per 16 tasks
                const int row = 4 * (cq - n + t) + r;
                if (t <= n && row < p.nrows) {
                    float v = usel;
                    if (EPI == EPI_SILU_MUL) v = silu_ref(gsel) * usel;   // MLPBlock.java:132-142
                    if (EPI == EPI_RESID) v = v + rv;                      // TransformerBlock.java:185,203
                    p.out[row] = v;
                }
                if (EPI == EPI_RESID && cq + 1 < q1) rv = resid_of_batch(cq + 1);
            }
            ++cq;
        };
        for (int it = 0; it + D < items; it += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                compute(wq[d], cg + d, d == D - 1);              // refilled in place AFTER the group (see gemv_i8q4_p16_kernel: requested
                __builtin_amdgcn_sched_barrier(0);              // before it, the ring is copied -- behind vmcnt waits -- at the loop end)
                issue(wq[d]);
                __builtin_amdgcn_sched_barrier(0);
            }
            cg += D;
            if (cg == G) { cg = 0; pass_end(); }
        }
#pragma unroll
        for (int d = 0; d < D; d++) {
            compute(wq[d], cg + d, d == D - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        pass_end();
    }
    if (ARGMAX && p.amax_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
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

// ------------------------------------------------------------------------------------------------ prompt rows (M > 1), reference order
// GemmerBF16 with M rows (AbstractModel.batchForward, AbstractModel.java:295-312): every (prompt row, weight row) pair is the
// GEMV's 16-lane chain; only the weight side is shared.  What sizes the tile (first version, 8 prompt rows x 4 row quads per wave,
// ran at 1/6 of its fma issue rate: profiles/r05d_*): every fma needs an activation value and a weight value, and the two paths
// they come by are narrow -- LDS delivers 128 B/clk per CU, i.e. 32 x QW fmas per clock when an activation word read from LDS is
// reused for QW row quads held in registers; the CU's L1 fill path delivers ~10 B/clk (~25 GB/s, the rate every streaming kernel of
// this library sees), i.e. ~5 x MR fmas per clock when a weight fetched into the CU is reused for MR prompt rows.  With
// QW x MR accumulators per lane bounded by the register file, QW = 4 and MR = 24 balance the two (128 vs 125 fmas per clock, against
// 228 the packed fma could issue); 129 rows = 6 tiles of 24 (144 padded rows).
// The activations of a chunk are written ONCE per projection input as the LDS image the GEMM reads (rows_act_bf16r_kernel):
//   image[tile][p][j][rq][t][rr] (floats)   p = 64-element pair of steps, j = 0..3: element 64p + 16j + t,
//                                           rq = row quad of the tile (0 .. MR/4 - 1), t = chain, rr = row inside the quad
// so that one ds_read_b128 hands lane t the values of 4 prompt rows for one link of its chain (16 lanes = 256 contiguous bytes,
// conflict-free; the wave's four 16-lane rows read the same addresses: broadcast), and a K chunk of a tile is contiguous.
constexpr int BFR_MR = 24;           // prompt rows per tile
constexpr int BFR_RQ = BFR_MR / 4;   // row quads per tile
constexpr int BFR_QW = 4;            // weight row quads (4 weight rows each) per wave
constexpr int BFR_WAVES = 4;         // waves per workgroup: 64 weight rows
constexpr int BFR_CG = 2;            // groups of 128 elements per LDS chunk (256 elements x 24 rows x 4 B = 24 KiB; two buffers)
constexpr int BFR_GROUP_F4 = 8 * BFR_RQ * 16;   // f32x4 words of one group's image: [p(2)][j(4)][rq][t(16)]
static inline size_t bfr_image_floats(int rows_cap, int K) { return (size_t)((rows_cap + BFR_MR - 1) / BFR_MR) * ((K + BF16R_GROUP - 1) / BF16R_GROUP) * BFR_GROUP_F4 * 4; }
static inline size_t lds_bytes_gemm_bf16r() { return (size_t)2 * BFR_CG * BFR_GROUP_F4 * 16; }

enum { PROB_SILU_BF16 = 4 };         // y = silu(gate) * up (MLPBlock.java:132-142), rounded to BF16: the down projection's input
struct RowsBfrParams {
    const float* x; int ldx;         // input rows (F32)
    const float* x2; int ldx2;       // PROB_SILU_BF16: the `up` rows
    const float* nw; float eps;      // PROB_RMS_BF16: norm weights (F32)
    int K;                           // multiple of 128
    float* image;                    // bfr image of the chunk's rows
};
// one workgroup per prompt row
template <int PRO>
__global__ __launch_bounds__(256) void rows_act_bf16r_kernel(RowsBfrParams p) {
    __shared__ double red[32];
    const int row = blockIdx.x;
    const float* x = p.x + (size_t)row * p.ldx;
    float fs = 1.0f;
    if (PRO == PROB_RMS_BF16) fs = rms_factor(x, p.K, p.eps, red);
    const int NP = p.K / 64, rin = row % BFR_MR;
    float* img = p.image + (size_t)(row / BFR_MR) * NP * (BFR_GROUP_F4 * 2) + (rin >> 2) * 64 + (rin & 3);   // (BFR_GROUP_F4 * 4 floats per group = 2 p)
    for (int unit = threadIdx.x; unit < p.K / 8; unit += blockDim.x) {
        const int e0 = unit * 8;
        const float4 xa = *(const float4*)(x + e0), xb = *(const float4*)(x + e0 + 4);
        float y[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
        if (PRO == PROB_RMS_BF16) {
            float w[8];
            load8_norm(p.nw, e0, w);
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = w[i] * (fs * y[i]);
        }
        if (PRO == PROB_SILU_BF16) {
            const float* u = p.x2 + (size_t)row * p.ldx2 + e0;
            const float4 ua = *(const float4*)u, ub = *(const float4*)(u + 4);
            const float uu[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = silu_ref(y[i]) * uu[i];
        }
        const int pp = e0 >> 6, j = (e0 >> 4) & 3, t0 = e0 & 15;
        float* dst = img + ((size_t)pp * 4 + j) * (BFR_RQ * 64) + t0 * 4;
#pragma unroll
        for (int i = 0; i < 8; i++) dst[4 * i] = bf16_to_f32(f32_to_bf16(y[i]));   // quantizeBF16 (RNE), widened again
    }
}

struct GemmBfrParams {
    const uint8_t* w;                // BF16T copy, row stride ldb bytes
    int ldb, nrows, K, rows;         // weight rows, K (multiple of 256), prompt rows
    const float* image;              // activations (rows_act_bf16r_kernel)
    float* out; int ldc;             // out[row * ldc + weight row]
    const float* resid; int ldr;     // EPI_RESID
    int nslices, nrt;                // 64-row weight slices, MR-row tiles
};
// grid: ((nslices + 7) / 8) * 8 * nrt workgroups; the row tiles of one weight slice are consecutive in launch order on ONE XCD
// (blockIdx & 7), so all but the first find the slice in that XCD's L2.
template <int EPI>
__global__ __launch_bounds__(BFR_WAVES * 64, 2) void gemm_bf16r_kernel(GemmBfrParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = (f32x4*)smem;                                  // [2][BFR_CG * BFR_GROUP_F4] f32x4
    constexpr int NT = BFR_WAVES * 64, CHUNK4 = BFR_CG * BFR_GROUP_F4, LU = (CHUNK4 + NT - 1) / NT;   // 16-byte units per chunk / per thread
    const int id = blockIdx.x, xcd = id & 7, k = id >> 3, rt = k % p.nrt, slice = (k / p.nrt) * 8 + xcd;
    if (slice >= p.nslices) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane >> 4, t = lane & 15;
    const int G = p.K / BF16R_GROUP, nchunks = G / BFR_CG;      // host: G % 2 == 0
    // uniform base pointers + 32-bit lane offsets (a weight copy is < 4 GiB, an activation image < 64 MiB): per-lane 64-bit pointers
    // of four row quads and of the staging loads cost the registers the accumulators need
    const char* img = (const char*)p.image + (size_t)rt * G * BFR_GROUP_F4 * 16;
    unsigned woff[BFR_QW];
#pragma unroll
    for (int q = 0; q < BFR_QW; q++) {
        int row = slice * 64 + wave * 16 + q * 4 + r;
        row = row < p.nrows ? row : p.nrows - 1;
        woff[q] = (unsigned)(row >> 2) * ((unsigned)p.ldb * 4u) + (unsigned)(row & 3) * 256u + (unsigned)t * 16u;
    }
    // accumulators as register PAIRS updated in place by v_pk_fma_f32 (two prompt rows per instruction; each half is the fused fma of
    // the chain).  Written as asm: left to hipcc, the packed fmas get fresh destination registers, the loop carries copies of every
    // accumulator, and under the 256-register budget of two workgroups per CU the prefetched loads are spilled as they arrive.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc[BFR_QW][BFR_MR / 2];
#pragma unroll
    for (int q = 0; q < BFR_QW; q++)
#pragma unroll
        for (int m = 0; m < BFR_MR / 2; m++) acc[q][m] = f32x2{0.0f, 0.0f};
    f32x4 sreg[LU];
    auto stage_load = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < LU; u++) {
            int idx = tid + u * NT;
            idx = idx < CHUNK4 ? idx : CHUNK4 - 1;
            sreg[u] = *(const f32x4*)(img + (unsigned)(c * CHUNK4 + idx) * 16u);
        }
    };
    auto stage_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < LU; u++)
            if (tid + u * NT < CHUNK4) lds[buf * CHUNK4 + tid + u * NT] = sreg[u];
    };
    i32x4 wa[BFR_QW], wb[BFR_QW];
    auto wload = [&](i32x4 (&w)[BFR_QW], int g) __attribute__((always_inline)) {
        const unsigned gg = (unsigned)(g < G ? g : G - 1) * 1024u;
#pragma unroll
        for (int q = 0; q < BFR_QW; q++) w[q] = *(const i32x4*)(p.w + (woff[q] + gg));
    };
    // The activation words of a link: BFR_RQ ds_read_b128, 256 bytes apart, written as asm and NOT waited for -- hipcc otherwise
    // hoists every LDS read of a group to its top (the reads of 8 links live at once: hundreds of registers, spills) and closes each
    // with lgkmcnt(0).  bfr_tie: at most N younger LDS reads stay in flight, the words become readable (jh_p16.h: lds_tie).
    auto read_link = [&](f32x4 (&av)[BFR_RQ], unsigned abase, int i) __attribute__((always_inline)) {
        const unsigned ad = abase + (unsigned)(i * BFR_RQ) * 256u;
        static_assert(BFR_RQ == 4 || BFR_RQ == 5 || BFR_RQ == 6, "asm below");
        if constexpr (BFR_RQ == 4)
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:256\n\tds_read_b128 %2, %4 offset:512\n\tds_read_b128 %3, %4 offset:768"
                         : "=&v"(av[0]), "=&v"(av[1]), "=&v"(av[2]), "=&v"(av[3]) : "v"(ad) : "memory");
        else if constexpr (BFR_RQ == 5)
            asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:256\n\tds_read_b128 %2, %5 offset:512\n\tds_read_b128 %3, %5 offset:768\n\t"
                         "ds_read_b128 %4, %5 offset:1024"
                         : "=&v"(av[0]), "=&v"(av[1]), "=&v"(av[2]), "=&v"(av[3]), "=&v"(av[4]) : "v"(ad) : "memory");
        else
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:256\n\tds_read_b128 %2, %6 offset:512\n\tds_read_b128 %3, %6 offset:768\n\t"
                         "ds_read_b128 %4, %6 offset:1024\n\tds_read_b128 %5, %6 offset:1280"
                         : "=&v"(av[0]), "=&v"(av[1]), "=&v"(av[2]), "=&v"(av[3]), "=&v"(av[4]), "=&v"(av[5]) : "v"(ad) : "memory");
    };
    auto tie_link = [&](f32x4 (&av)[BFR_RQ]) __attribute__((always_inline)) {   // the OLDER link's words have landed (the younger link's BFR_RQ reads may still fly)
        if constexpr (BFR_RQ == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3])::"memory");
        else if constexpr (BFR_RQ == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5])::"memory");
    };
    auto tie_last = [&](f32x4 (&av)[BFR_RQ]) __attribute__((always_inline)) {
        if constexpr (BFR_RQ == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3])::"memory");
        else if constexpr (BFR_RQ == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5])::"memory");
    };
    auto fma_link = [&](const f32x4 (&av)[BFR_RQ], const i32x4 (&w)[BFR_QW], int i) __attribute__((always_inline)) {
        f32x2 wv[BFR_QW];
#pragma unroll
        for (int q = 0; q < BFR_QW; q++) {
            const int d = w[q][i >> 1];
            const float f = __int_as_float((i & 1) ? (d & (int)0xffff0000) : (d << 16));
            wv[q] = f32x2{f, f};
        }
#pragma unroll
        for (int q = 0; q < BFR_QW; q++)
#pragma unroll
            for (int rq = 0; rq < BFR_RQ; rq++) {
                const f32x2 alo = __builtin_shufflevector(av[rq], av[rq], 0, 1), ahi = __builtin_shufflevector(av[rq], av[rq], 2, 3);
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[q][2 * rq]) : "v"(alo), "v"(wv[q]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[q][2 * rq + 1]) : "v"(ahi), "v"(wv[q]));
            }
    };
    // one group = 8 links; the words of link i+1 are requested before the fmas of link i
    auto compute = [&](const i32x4 (&w)[BFR_QW], const f32x4* a) __attribute__((always_inline)) {
        const unsigned abase = lds_addr(a + t);
        f32x4 av0[BFR_RQ], av1[BFR_RQ];
        read_link(av0, abase, 0);
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            read_link(av1, abase, i + 1);
            tie_link(av0);
            fma_link(av0, w, i);
            if (i + 2 < 8) { read_link(av0, abase, i + 2); tie_link(av1); } else tie_last(av1);
            fma_link(av1, w, i + 1);
        }
    };
    stage_load(0);
    wload(wa, 0);
    stage_store(0);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        const int buf = c & 1;
        if (c + 1 < nchunks) stage_load(c + 1);
        const f32x4* a = lds + buf * CHUNK4;
        const int g0 = c * BFR_CG;
#pragma unroll
        for (int gi = 0; gi < BFR_CG; gi += 2) {
            wload(wb, g0 + gi + 1);
            compute(wa, a + (size_t)gi * BFR_GROUP_F4);
            wload(wa, g0 + gi + 2);
            compute(wb, a + (size_t)(gi + 1) * BFR_GROUP_F4);
        }
        if (c + 1 < nchunks) stage_store(buf ^ 1);
        __syncthreads();
    }
    // ---- epilogue: the 16 lanes of a row all hold the finished sums; result (q, m) is stored by lane (q * MR + m) mod 16
#pragma unroll
    for (int q = 0; q < BFR_QW; q++)
#pragma unroll
        for (int m = 0; m < BFR_MR; m++) {
            const float res = row16_tree_sum((m & 1) ? acc[q][m >> 1].y : acc[q][m >> 1].x);
            if (t == ((q * BFR_MR + m) & 15)) {
                const int wr = slice * 64 + wave * 16 + q * 4 + r, prow = rt * BFR_MR + m;
                if (wr < p.nrows && prow < p.rows) {
                    float v = res;
                    if (EPI == EPI_RESID) v = v + p.resid[(size_t)prow * p.ldr + wr];
                    p.out[(size_t)prow * p.ldc + wr] = v;
                }
            }
            __builtin_amdgcn_sched_barrier(0);                  // one result at a time (96 of them: hipcc otherwise keeps them all live)
        }
}

}  // namespace jh
