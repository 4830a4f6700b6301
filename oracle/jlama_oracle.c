/*
 * jlama_oracle.c -- CPU restatement of the Jlama hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity ORACLE for jlama-hip.  It is NOT part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (libjlamahip.so) never links, loads or calls anything in here.
 *
 * Every function cites the reference file:line it restates.  Path abbreviations:
 *   core/  = jlama-core/src/main/java/com/github/tjake/jlama/
 *   nc/    = jlama-native/src/main/c/
 *   PTO    = core/tensor/operations/PanamaTensorOperations.java
 *
 * Which Panama variant: the AVX-512 one (vectorType == AVX_512, FloatVector.SPECIES_512 = 16
 * float lanes), because that is what core/util/MachineSpec.java:45-64 selects on any
 * AVX-512 host.  `reduceLanes(ADD)` order is unspecified by the Vector API; we use the
 * halving tree HotSpot emits on x86 (512->256->128->64->32), see jo_reduce16().
 *
 * Third-party arithmetic that is NOT in /root/reference: net.jafama:jafama:2.3.2
 * (FastMath.exp/sqrt/pow/cos/sin).  All call sites evaluate in double and cast to float,
 * so libm's correctly-rounded-in-practice double functions reproduce the float result
 * except for rare double-rounding cases.  Pinned only by the RoPE KAT
 * (jlama-tests/.../model/TestCorrectness.java:92-115); exp/sqrt are "parity unpinned"
 * beyond the logit tolerance.
 *
 * Build: see oracle/Makefile (-O3 -march=x86-64-v3 -ffp-contract=off: Java never contracts a*b+c; every
 * FMA below is an explicit fmaf() where the reference calls FloatVector.fma()).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
/* Optional x86 SIMD bodies for the two GEMV inner loops that dominate a full-size (8B) oracle run.  They execute the
 * SAME per-lane operations in the SAME order as the scalar statements next to them (16 float lanes = two 8-lane
 * registers, one fmadd per lane per step), so results are bit-identical; tests/test_oracle.py asserts that against the
 * always-scalar jo_gemm_*_scalar entry points.  Without AVX2+FMA the scalar text is what runs. */
#if defined(__AVX2__) && defined(__FMA__) && !defined(JO_NO_SIMD)
#include <immintrin.h>
#define JO_SIMD 1
#else
#define JO_SIMD 0
#endif

#define JO_BLOCK 32
#define JO_HALF 16

int jo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
static inline int jo_thread_id(void) {
#ifdef _OPENMP
    return omp_get_thread_num();
#else
    return 0;
#endif
}

/* dtype tags shared with include/jlama_hip.h */
enum { JO_DT_F32 = 0, JO_DT_BF16 = 1, JO_DT_I8 = 2, JO_DT_Q4 = 3 };

/* ------------------------------------------------------------------------------------
 * scalar helpers
 * ---------------------------------------------------------------------------------- */

/* Java (byte)(float): float->int saturating (NaN->0), then keep the low 8 bits (JLS 5.1.3). */
static inline int8_t jo_f2b(float f) {
    int32_t i;
    if (f != f) i = 0;
    else if (f >= 2147483648.0f) i = INT32_MAX;
    else if (f <= -2147483648.0f) i = INT32_MIN;
    else i = (int32_t)f; /* truncation toward zero */
    return (int8_t)(uint8_t)(i & 0xff);
}

/* core/math/FloatConversions.java:31-33 */
float jo_bf16_to_f32(uint16_t raw) {
    uint32_t u = ((uint32_t)raw) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* core/math/FloatConversions.java:35-60 (+ round() :71-90): round-to-nearest-even on the
 * 16 dropped mantissa bits; the carry may ripple into the exponent field. */
uint16_t jo_f32_to_bf16(float n) {
    uint32_t nbits;
    memcpy(&nbits, &n, 4);
    int s = (nbits >> 16) & 0x8000;
    int e = (nbits >> 16) & 0x7f80;
    int m = (nbits & 0x7fffff);
    if (e != 0x7f80) {
        int shifts = 16, mid = 1 << (shifts - 1), mask = (1 << shifts) - 1;
        int mshift = m >> shifts, masked = m & mask, cmp = masked - mid, m1;
        if (cmp > 0) m1 = mshift + 1;
        else if (cmp < 0) m1 = mshift;
        else m1 = (mshift & 1) ? mshift + 1 : mshift;
        return (uint16_t)(s | (e + m1));
    }
    return m != 0 ? (uint16_t)0x7fc0 : (uint16_t)(nbits >> 16);
}

/* HotSpot x86 lowering of FloatVector.reduceLanes(ADD) for a 512-bit species:
 * add upper 256 to lower 256, upper 128 to lower 128, then 64, then 32.  (Order is
 * unspecified by the API: last-bit differences vs. a real JVM are inherent.) */
static inline float jo_reduce16(const float* v) {
    float a[8], b[4], c[2];
    for (int i = 0; i < 8; i++) a[i] = v[i] + v[i + 8];
    for (int i = 0; i < 4; i++) b[i] = a[i] + a[i + 4];
    for (int i = 0; i < 2; i++) c[i] = b[i] + b[i + 2];
    return c[0] + c[1];
}

/* ------------------------------------------------------------------------------------
 * a1. Q4 weight format -- core/tensor/Q4ByteBufferTensor.java
 * ---------------------------------------------------------------------------------- */

/* processBlock, Q4ByteBufferTensor.java:66-106.  x: [rows, cols] F32 row-major.
 * nib: rows*cols/2 bytes, byte j of a block: low nibble = elem j, high = elem j+16.
 * scales: [rows, cols/32] F32 ("blockF", :41,140). */
void jo_q4_quantize(const float* x, int64_t rows, int cols, uint8_t* nib, float* scales) {
    int nb = cols / JO_BLOCK;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++) {
        for (int b = 0; b < nb; b++) {
            const float* xb = x + r * (int64_t)cols + (int64_t)b * JO_BLOCK;
            float max = 1.4e-45f /* Float.MIN_VALUE */, amax = 1.4e-45f;
            for (int i = 0; i < JO_BLOCK; i++) {
                float v = xb[i];
                float absv = v < 0 ? -v : v;
                if (absv > amax) { max = v; amax = absv; }
            }
            float scale = max / -8.0f;
            float iscale = scale != 0.0f ? 1.0f / scale : 0.0f;
            scales[r * nb + b] = scale;
            uint8_t* ob = nib + (r * (int64_t)cols + (int64_t)b * JO_BLOCK) / 2;
            for (int j = 0; j < JO_HALF; j++) {
                float f0 = xb[j] * iscale;
                float f1 = xb[j + JO_HALF] * iscale;
                int8_t t0 = jo_f2b(f0 + 8.5f), t1 = jo_f2b(f1 + 8.5f);
                int8_t fb0 = t0 < 15 ? t0 : 15; /* (byte) Math.min(15, (byte)(..)) */
                int8_t fb1 = t1 < 15 ? t1 : 15;
                ob[j] = (uint8_t)((fb0) | ((fb1) << 4)); /* (byte)((fb0) | ((fb1) << 4)) */
            }
        }
    }
}

/* get(), Q4ByteBufferTensor.java:179-197: value = (nibble - 8) * scale. */
static inline float jo_q4_get(const uint8_t* nib_row, const float* scale_row, int c) {
    int blk = c / JO_BLOCK, in = c % JO_BLOCK;
    uint8_t b0 = in < JO_HALF ? nib_row[blk * JO_HALF + in] : nib_row[blk * JO_HALF + in - JO_HALF];
    int x = in < JO_HALF ? (b0 & 0x0F) - 8 : ((b0 >> 4) & 0x0F) - 8;
    return (float)x * scale_row[blk];
}

void jo_q4_dequantize(const uint8_t* nib, const float* scales, int64_t rows, int cols, float* out) {
    int nb = cols / JO_BLOCK;
    for (int64_t r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++)
            out[r * cols + c] = jo_q4_get(nib + r * (cols / 2), scales + r * nb, c);
}

/* ------------------------------------------------------------------------------------
 * a2. dynamic activation quantizer F32 -> I8, Panama AVX-512
 *     PTO:1684-1723 (quantizeQ8_512).  NOT the scalar ctor (Q8ByteBufferTensor.java:67-88).
 * ---------------------------------------------------------------------------------- */
void jo_q8_quantize(const float* x, int rows, int ldx, int offset, int length, int8_t* q, int ldq,
                    float* d, int ldd) {
    for (int b = 0; b < rows; b++) {
        for (int i = offset; i < offset + length; i += JO_BLOCK) {
            const float* xb = x + (int64_t)b * ldx + i;
            float maxScalar = 0.0f; /* abs() lanes, reduceLanes(MAX) */
            for (int t = 0; t < JO_BLOCK; t++) {
                float a = fabsf(xb[t]);
                if (a > maxScalar) maxScalar = a;
            }
            float dd = maxScalar / 127.0f;
            float id = (maxScalar != 0.0f) ? 127.0f / maxScalar : 0.0f;
            for (int t = 0; t < JO_BLOCK; t++) {
                float v = xb[t] * id;  /* fv.mul(vid) */
                v = v + 0.5f;          /* .add(F32_ROUND_UP_512) */
                q[(int64_t)b * ldq + i + t] = jo_f2b(v); /* F2B: truncation, NOT round-to-nearest */
            }
            d[(int64_t)b * ldd + i / JO_BLOCK] = dd;
        }
    }
}

/* F32 -> BF16 activation quantizer: PTO:1624-1628 returns new BFloat16BufferTensor(ft)
 * = element-wise FloatConversions.float32ToBFloat16 (RNE). */
void jo_bf16_quantize(const float* x, int64_t n, uint16_t* out) {
    for (int64_t i = 0; i < n; i++) out[i] = jo_f32_to_bf16(x[i]);
}

/* ------------------------------------------------------------------------------------
 * a3-a6. batchDotProduct.  Semantics (core/tensor/operations/TensorOperations.java:62-72,
 * NaiveTensorOperations.java:79-100, PTO:95-146):
 *   result[i, j + rRowOff'] = sum_k a[i, aColOff+k] * b[j, bColOff+k],  i in [0,M),
 *   j in [bRowOff, bRowOff+N).  Panama writes column (j + rOffset), rOffset = rRowOffset
 *   (PTO:145,227); the model only uses bRowOff==0 or rRowOff==0 where Naive agrees.
 * All variants below take dense row-major operands + leading dimensions.
 * ---------------------------------------------------------------------------------- */

/* I8 x Q4 -> F32, GemmerI8Q4_512 1x1 tile PTO:807-850 (1x4 :852-955 and 2x2 :958-1043 do
 * the same per-element arithmetic).  Lane t of 16 accumulates over blocks in ascending K:
 *   acc_t = fma(sa*sb, (float)(short)(lo[t]*a[t] + hi[t]*a[t+16]), acc_t). */
static inline float jo_dot_i8q4_scalar(const int8_t* ap, const float* afr, const uint8_t* bp, const float* bfr, int nblk) {
    float acc[16];
    for (int t = 0; t < 16; t++) acc[t] = 0.0f;
    for (int blk = 0; blk < nblk; blk++, ap += JO_BLOCK, bp += JO_HALF) {
        float scale = afr[blk] * bfr[blk];
        for (int t = 0; t < 16; t++) {
            int16_t lo = (int16_t)((bp[t] & 0x0F) - 8);
            int16_t hi = (int16_t)(((bp[t] >> 4) & 0x0F) - 8);
            int16_t isum = (int16_t)(lo * ap[t] + hi * ap[t + 16]); /* |.| <= 2032: exact */
            acc[t] = fmaf(scale, (float)isum, acc[t]);
        }
    }
    return jo_reduce16(acc);
}
#if JO_SIMD
/* same lanes, same order: lanes 0..7 in acc0, 8..15 in acc1 */
static inline float jo_dot_i8q4_simd(const int8_t* ap, const float* afr, const uint8_t* bp, const float* bfr, int nblk) {
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
    const __m128i m4 = _mm_set1_epi8(0x0F), e8 = _mm_set1_epi8(8);
    for (int blk = 0; blk < nblk; blk++, ap += JO_BLOCK, bp += JO_HALF) {
        const __m256 sc = _mm256_set1_ps(afr[blk] * bfr[blk]);
        const __m128i bb = _mm_loadu_si128((const __m128i*)bp);
        const __m128i lo = _mm_sub_epi8(_mm_and_si128(bb, m4), e8);
        const __m128i hi = _mm_sub_epi8(_mm_and_si128(_mm_srli_epi16(bb, 4), m4), e8);
        const __m256i a0 = _mm256_cvtepi8_epi16(_mm_loadu_si128((const __m128i*)ap));
        const __m256i a1 = _mm256_cvtepi8_epi16(_mm_loadu_si128((const __m128i*)(ap + 16)));
        const __m256i is = _mm256_add_epi16(_mm256_mullo_epi16(_mm256_cvtepi8_epi16(lo), a0),
                                            _mm256_mullo_epi16(_mm256_cvtepi8_epi16(hi), a1));
        const __m256 f0 = _mm256_cvtepi32_ps(_mm256_cvtepi16_epi32(_mm256_castsi256_si128(is)));
        const __m256 f1 = _mm256_cvtepi32_ps(_mm256_cvtepi16_epi32(_mm256_extracti128_si256(is, 1)));
        acc0 = _mm256_fmadd_ps(sc, f0, acc0);
        acc1 = _mm256_fmadd_ps(sc, f1, acc1);
    }
    float acc[16];
    _mm256_storeu_ps(acc, acc0);
    _mm256_storeu_ps(acc + 8, acc1);
    return jo_reduce16(acc);
}
#endif
static void jo_gemm_i8q4_impl(int simd, const int8_t* a, const float* af, int lda, int ldaf, const uint8_t* b,
                              const float* bf, int ldb_bytes, int ldbf, float* r, int ldc, int M, int aColOff,
                              int bColOff, int K, int rRowOff, int bRowOff, int N) {
    (void)simd;
#pragma omp parallel for schedule(static) if (N >= 512)
    for (int j = bRowOff; j < bRowOff + N; j++) {
        for (int i = 0; i < M; i++) {
            const int8_t* ap = a + (int64_t)i * lda + aColOff;
            const float* afr = af + (int64_t)i * ldaf + aColOff / JO_BLOCK;
            const uint8_t* bp = b + (int64_t)j * ldb_bytes + bColOff / 2;
            const float* bfr = bf + (int64_t)j * ldbf + bColOff / JO_BLOCK;
#if JO_SIMD
            if (simd) { r[(int64_t)i * ldc + j + rRowOff] = jo_dot_i8q4_simd(ap, afr, bp, bfr, K / JO_BLOCK); continue; }
#endif
            r[(int64_t)i * ldc + j + rRowOff] = jo_dot_i8q4_scalar(ap, afr, bp, bfr, K / JO_BLOCK);
        }
    }
}
void jo_gemm_i8q4(const int8_t* a, const float* af, int lda, int ldaf, const uint8_t* b,
                  const float* bf, int ldb_bytes, int ldbf, float* r, int ldc, int M, int aColOff,
                  int bColOff, int K, int rRowOff, int bRowOff, int N) {
    jo_gemm_i8q4_impl(1, a, af, lda, ldaf, b, bf, ldb_bytes, ldbf, r, ldc, M, aColOff, bColOff, K, rRowOff, bRowOff, N);
}
/* the scalar text only (what the SIMD body is checked against) */
void jo_gemm_i8q4_scalar(const int8_t* a, const float* af, int lda, int ldaf, const uint8_t* b,
                         const float* bf, int ldb_bytes, int ldbf, float* r, int ldc, int M, int aColOff,
                         int bColOff, int K, int rRowOff, int bRowOff, int N) {
    jo_gemm_i8q4_impl(0, a, af, lda, ldaf, b, bf, ldb_bytes, ldbf, r, ldc, M, aColOff, bColOff, K, rRowOff, bRowOff, N);
}
int jo_simd_enabled(void) { return JO_SIMD; }

/* F32 x Q4 -> F32, GemmerF32Q4_512 1x1 PTO:336-374: dequantize first w = float(nib-8)*scale,
 * then acc = fma(a_lo, w_lo, acc); acc = fma(a_hi, w_hi, acc) per block, 16 lanes. */
static inline float jo_dot_f32q4_scalar(const float* ap, const uint8_t* bp, const float* bfr, int nblk) {
    float acc[16];
    for (int t = 0; t < 16; t++) acc[t] = 0.0f;
    for (int blk = 0; blk < nblk; blk++, ap += JO_BLOCK, bp += JO_HALF) {
        float scale = bfr[blk];
        for (int t = 0; t < 16; t++) {
            float low = (float)((bp[t] & 0x0F) - 8) * scale;
            acc[t] = fmaf(ap[t], low, acc[t]);
        }
        for (int t = 0; t < 16; t++) {
            float high = (float)(((bp[t] >> 4) & 0x0F) - 8) * scale;
            acc[t] = fmaf(ap[t + 16], high, acc[t]);
        }
    }
    return jo_reduce16(acc);
}
#if JO_SIMD
static inline float jo_dot_f32q4_simd(const float* ap, const uint8_t* bp, const float* bfr, int nblk) {
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
    const __m128i m4 = _mm_set1_epi8(0x0F), e8 = _mm_set1_epi8(8);
    for (int blk = 0; blk < nblk; blk++, ap += JO_BLOCK, bp += JO_HALF) {
        const __m256 sc = _mm256_set1_ps(bfr[blk]);
        const __m128i bb = _mm_loadu_si128((const __m128i*)bp);
        const __m128i lo = _mm_sub_epi8(_mm_and_si128(bb, m4), e8);
        const __m128i hi = _mm_sub_epi8(_mm_and_si128(_mm_srli_epi16(bb, 4), m4), e8);
        const __m256 l0 = _mm256_mul_ps(_mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(lo)), sc);
        const __m256 l1 = _mm256_mul_ps(_mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(_mm_srli_si128(lo, 8))), sc);
        const __m256 h0 = _mm256_mul_ps(_mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(hi)), sc);
        const __m256 h1 = _mm256_mul_ps(_mm256_cvtepi32_ps(_mm256_cvtepi8_epi32(_mm_srli_si128(hi, 8))), sc);
        acc0 = _mm256_fmadd_ps(_mm256_loadu_ps(ap), l0, acc0);
        acc1 = _mm256_fmadd_ps(_mm256_loadu_ps(ap + 8), l1, acc1);
        acc0 = _mm256_fmadd_ps(_mm256_loadu_ps(ap + 16), h0, acc0);
        acc1 = _mm256_fmadd_ps(_mm256_loadu_ps(ap + 24), h1, acc1);
    }
    float acc[16];
    _mm256_storeu_ps(acc, acc0);
    _mm256_storeu_ps(acc + 8, acc1);
    return jo_reduce16(acc);
}
#endif
static void jo_gemm_f32q4_impl(int simd, const float* a, int lda, const uint8_t* b, const float* bf, int ldb_bytes,
                               int ldbf, float* r, int ldc, int M, int aColOff, int bColOff, int K, int rRowOff,
                               int bRowOff, int N) {
    (void)simd;
#pragma omp parallel for schedule(static) if (N >= 512)
    for (int j = bRowOff; j < bRowOff + N; j++) {
        for (int i = 0; i < M; i++) {
            const float* ap = a + (int64_t)i * lda + aColOff;
            const uint8_t* bp = b + (int64_t)j * ldb_bytes + bColOff / 2;
            const float* bfr = bf + (int64_t)j * ldbf + bColOff / JO_BLOCK;
#if JO_SIMD
            if (simd) { r[(int64_t)i * ldc + j + rRowOff] = jo_dot_f32q4_simd(ap, bp, bfr, K / JO_BLOCK); continue; }
#endif
            r[(int64_t)i * ldc + j + rRowOff] = jo_dot_f32q4_scalar(ap, bp, bfr, K / JO_BLOCK);
        }
    }
}
void jo_gemm_f32q4(const float* a, int lda, const uint8_t* b, const float* bf, int ldb_bytes,
                   int ldbf, float* r, int ldc, int M, int aColOff, int bColOff, int K, int rRowOff,
                   int bRowOff, int N) {
    jo_gemm_f32q4_impl(1, a, lda, b, bf, ldb_bytes, ldbf, r, ldc, M, aColOff, bColOff, K, rRowOff, bRowOff, N);
}
void jo_gemm_f32q4_scalar(const float* a, int lda, const uint8_t* b, const float* bf, int ldb_bytes,
                          int ldbf, float* r, int ldc, int M, int aColOff, int bColOff, int K, int rRowOff,
                          int bRowOff, int N) {
    jo_gemm_f32q4_impl(0, a, lda, b, bf, ldb_bytes, ldbf, r, ldc, M, aColOff, bColOff, K, rRowOff, bRowOff, N);
}

/* F32 x F32 -> F32, GemmerF32 1x1 PTO:1086-1102: 16 lanes, one fma per 16-element step. */
void jo_gemm_f32(const float* a, int lda, const float* b, int ldb, float* r, int ldc, int M,
                 int aColOff, int bColOff, int K, int rRowOff, int bRowOff, int N) {
#pragma omp parallel for schedule(static) if (N >= 512)
    for (int j = bRowOff; j < bRowOff + N; j++) {
        for (int i = 0; i < M; i++) {
            float acc[16];
            for (int t = 0; t < 16; t++) acc[t] = 0.0f;
            const float* ap = a + (int64_t)i * lda + aColOff;
            const float* bp = b + (int64_t)j * ldb + bColOff;
            for (int l = 0; l < K; l += 16)
                for (int t = 0; t < 16; t++) acc[t] = fmaf(ap[l + t], bp[l + t], acc[t]);
            r[(int64_t)i * ldc + j + rRowOff] = jo_reduce16(acc);
        }
    }
}

/* BF16 x BF16 -> F32, GemmerBF16 1x1 PTO:1279-1311: per 32-element step lanes t<16 take
 * elements t then t+16 (convertShape part 0 / part 1), both operands widened by <<16. */
static inline float jo_dot_bf16_scalar(const uint16_t* ap, const uint16_t* bp, int K) {
    float acc[16];
    for (int t = 0; t < 16; t++) acc[t] = 0.0f;
    for (int l = 0; l < K; l += 32) {
        for (int t = 0; t < 16; t++)
            acc[t] = fmaf(jo_bf16_to_f32(ap[l + t]), jo_bf16_to_f32(bp[l + t]), acc[t]);
        for (int t = 0; t < 16; t++)
            acc[t] = fmaf(jo_bf16_to_f32(ap[l + 16 + t]), jo_bf16_to_f32(bp[l + 16 + t]), acc[t]);
    }
    return jo_reduce16(acc);
}
/* F32 x BF16 -> F32, GemmerF32BF16 1x1 PTO:1511-1538. */
static inline float jo_dot_f32bf16_scalar(const float* ap, const uint16_t* bp, int K) {
    float acc[16];
    for (int t = 0; t < 16; t++) acc[t] = 0.0f;
    for (int l = 0; l < K; l += 32) {
        for (int t = 0; t < 16; t++) acc[t] = fmaf(ap[l + t], jo_bf16_to_f32(bp[l + t]), acc[t]);
        for (int t = 0; t < 16; t++) acc[t] = fmaf(ap[l + 16 + t], jo_bf16_to_f32(bp[l + 16 + t]), acc[t]);
    }
    return jo_reduce16(acc);
}
#if JO_SIMD
/* same lanes, same order: lanes 0..7 in acc0, 8..15 in acc1; widening = zero-extend to 32 bits, shift left 16 */
static inline __m256 jo_widen8_bf16(const uint16_t* p) {
    return _mm256_castsi256_ps(_mm256_slli_epi32(_mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i*)p)), 16));
}
static inline float jo_dot_bf16_simd(const uint16_t* ap, const uint16_t* bp, int K) {
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
    for (int l = 0; l < K; l += 32) {
        acc0 = _mm256_fmadd_ps(jo_widen8_bf16(ap + l), jo_widen8_bf16(bp + l), acc0);
        acc1 = _mm256_fmadd_ps(jo_widen8_bf16(ap + l + 8), jo_widen8_bf16(bp + l + 8), acc1);
        acc0 = _mm256_fmadd_ps(jo_widen8_bf16(ap + l + 16), jo_widen8_bf16(bp + l + 16), acc0);
        acc1 = _mm256_fmadd_ps(jo_widen8_bf16(ap + l + 24), jo_widen8_bf16(bp + l + 24), acc1);
    }
    float acc[16];
    _mm256_storeu_ps(acc, acc0);
    _mm256_storeu_ps(acc + 8, acc1);
    return jo_reduce16(acc);
}
static inline float jo_dot_f32bf16_simd(const float* ap, const uint16_t* bp, int K) {
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
    for (int l = 0; l < K; l += 32) {
        acc0 = _mm256_fmadd_ps(_mm256_loadu_ps(ap + l), jo_widen8_bf16(bp + l), acc0);
        acc1 = _mm256_fmadd_ps(_mm256_loadu_ps(ap + l + 8), jo_widen8_bf16(bp + l + 8), acc1);
        acc0 = _mm256_fmadd_ps(_mm256_loadu_ps(ap + l + 16), jo_widen8_bf16(bp + l + 16), acc0);
        acc1 = _mm256_fmadd_ps(_mm256_loadu_ps(ap + l + 24), jo_widen8_bf16(bp + l + 24), acc1);
    }
    float acc[16];
    _mm256_storeu_ps(acc, acc0);
    _mm256_storeu_ps(acc + 8, acc1);
    return jo_reduce16(acc);
}
#endif
#if JO_SIMD
/* four weight rows at a time: the activation widening is shared and eight independent fma chains keep the FMA ports busy;
 * every row still is exactly jo_dot_*_simd (same lanes, same order) */
#define JO_DOT4_BODY(LOADA)                                                                                   \
    __m256 c0[4], c1[4];                                                                                      \
    for (int q = 0; q < 4; q++) { c0[q] = _mm256_setzero_ps(); c1[q] = _mm256_setzero_ps(); }                 \
    for (int l = 0; l < K; l += 32) {                                                                         \
        const __m256 a0 = LOADA(ap + l), a1 = LOADA(ap + l + 8), a2 = LOADA(ap + l + 16), a3 = LOADA(ap + l + 24); \
        for (int q = 0; q < 4; q++) {                                                                         \
            const uint16_t* bp = b4[q] + l;                                                                   \
            c0[q] = _mm256_fmadd_ps(a0, jo_widen8_bf16(bp), c0[q]);                                           \
            c1[q] = _mm256_fmadd_ps(a1, jo_widen8_bf16(bp + 8), c1[q]);                                       \
            c0[q] = _mm256_fmadd_ps(a2, jo_widen8_bf16(bp + 16), c0[q]);                                      \
            c1[q] = _mm256_fmadd_ps(a3, jo_widen8_bf16(bp + 24), c1[q]);                                      \
        }                                                                                                     \
    }                                                                                                         \
    for (int q = 0; q < 4; q++) {                                                                             \
        float acc[16];                                                                                        \
        _mm256_storeu_ps(acc, c0[q]);                                                                         \
        _mm256_storeu_ps(acc + 8, c1[q]);                                                                     \
        out[q] = jo_reduce16(acc);                                                                            \
    }
static inline void jo_dot4_bf16_simd(const uint16_t* ap, const uint16_t* const* b4, int K, float* out) { JO_DOT4_BODY(jo_widen8_bf16) }
static inline void jo_dot4_f32bf16_simd(const float* ap, const uint16_t* const* b4, int K, float* out) { JO_DOT4_BODY(_mm256_loadu_ps) }
#endif
static void jo_gemm_bf16_impl(int simd, const uint16_t* a, int lda, const uint16_t* b, int ldb, float* r, int ldc, int M,
                              int aColOff, int bColOff, int K, int rRowOff, int bRowOff, int N) {
    (void)simd;
#if JO_SIMD
    if (simd && N >= 4) {
        const int N4 = N & ~3;
#pragma omp parallel for schedule(static) if (N >= 512)
        for (int j = bRowOff; j < bRowOff + N4; j += 4) {
            const uint16_t* b4[4];
            for (int q = 0; q < 4; q++) b4[q] = b + (int64_t)(j + q) * ldb + bColOff;
            for (int i = 0; i < M; i++) {
                float o4[4];
                jo_dot4_bf16_simd(a + (int64_t)i * lda + aColOff, b4, K, o4);
                for (int q = 0; q < 4; q++) r[(int64_t)i * ldc + j + q + rRowOff] = o4[q];
            }
        }
        bRowOff += N4; N -= N4;
    }
#endif
#pragma omp parallel for schedule(static) if (N >= 512)
    for (int j = bRowOff; j < bRowOff + N; j++) {
        for (int i = 0; i < M; i++) {
            const uint16_t* ap = a + (int64_t)i * lda + aColOff;
            const uint16_t* bp = b + (int64_t)j * ldb + bColOff;
#if JO_SIMD
            if (simd) { r[(int64_t)i * ldc + j + rRowOff] = jo_dot_bf16_simd(ap, bp, K); continue; }
#endif
            r[(int64_t)i * ldc + j + rRowOff] = jo_dot_bf16_scalar(ap, bp, K);
        }
    }
}
void jo_gemm_bf16(const uint16_t* a, int lda, const uint16_t* b, int ldb, float* r, int ldc, int M,
                  int aColOff, int bColOff, int K, int rRowOff, int bRowOff, int N) {
    jo_gemm_bf16_impl(1, a, lda, b, ldb, r, ldc, M, aColOff, bColOff, K, rRowOff, bRowOff, N);
}
void jo_gemm_bf16_scalar(const uint16_t* a, int lda, const uint16_t* b, int ldb, float* r, int ldc, int M,
                         int aColOff, int bColOff, int K, int rRowOff, int bRowOff, int N) {
    jo_gemm_bf16_impl(0, a, lda, b, ldb, r, ldc, M, aColOff, bColOff, K, rRowOff, bRowOff, N);
}
static void jo_gemm_f32bf16_impl(int simd, const float* a, int lda, const uint16_t* b, int ldb, float* r, int ldc, int M,
                                 int aColOff, int bColOff, int K, int rRowOff, int bRowOff, int N) {
    (void)simd;
#if JO_SIMD
    if (simd && N >= 4) {
        const int N4 = N & ~3;
#pragma omp parallel for schedule(static) if (N >= 512)
        for (int j = bRowOff; j < bRowOff + N4; j += 4) {
            const uint16_t* b4[4];
            for (int q = 0; q < 4; q++) b4[q] = b + (int64_t)(j + q) * ldb + bColOff;
            for (int i = 0; i < M; i++) {
                float o4[4];
                jo_dot4_f32bf16_simd(a + (int64_t)i * lda + aColOff, b4, K, o4);
                for (int q = 0; q < 4; q++) r[(int64_t)i * ldc + j + q + rRowOff] = o4[q];
            }
        }
        bRowOff += N4; N -= N4;
    }
#endif
#pragma omp parallel for schedule(static) if (N >= 512)
    for (int j = bRowOff; j < bRowOff + N; j++) {
        for (int i = 0; i < M; i++) {
            const float* ap = a + (int64_t)i * lda + aColOff;
            const uint16_t* bp = b + (int64_t)j * ldb + bColOff;
#if JO_SIMD
            if (simd) { r[(int64_t)i * ldc + j + rRowOff] = jo_dot_f32bf16_simd(ap, bp, K); continue; }
#endif
            r[(int64_t)i * ldc + j + rRowOff] = jo_dot_f32bf16_scalar(ap, bp, K);
        }
    }
}
void jo_gemm_f32bf16(const float* a, int lda, const uint16_t* b, int ldb, float* r, int ldc, int M,
                     int aColOff, int bColOff, int K, int rRowOff, int bRowOff, int N) {
    jo_gemm_f32bf16_impl(1, a, lda, b, ldb, r, ldc, M, aColOff, bColOff, K, rRowOff, bRowOff, N);
}
void jo_gemm_f32bf16_scalar(const float* a, int lda, const uint16_t* b, int ldb, float* r, int ldc, int M,
                            int aColOff, int bColOff, int K, int rRowOff, int bRowOff, int N) {
    jo_gemm_f32bf16_impl(0, a, lda, b, ldb, r, ldc, M, aColOff, bColOff, K, rRowOff, bRowOff, N);
}

/* The control implementation, NaiveTensorOperations.java:61-100: float s += a.get()*b.get()
 * sequentially over k with every operand dequantized through get().  adt/bdt in JO_DT_*. */
static inline float jo_get_a(int adt, const void* a, const float* af, int lda, int ldaf, int i, int c) {
    switch (adt) {
        case JO_DT_F32: return ((const float*)a)[(int64_t)i * lda + c];
        case JO_DT_BF16: return jo_bf16_to_f32(((const uint16_t*)a)[(int64_t)i * lda + c]);
        case JO_DT_I8: /* Q8ByteBufferTensor.get :140-146: b * d */
            return (float)((const int8_t*)a)[(int64_t)i * lda + c] * af[(int64_t)i * ldaf + c / JO_BLOCK];
        default: return jo_q4_get((const uint8_t*)a + (int64_t)i * (lda / 2), af + (int64_t)i * ldaf, c);
    }
}
void jo_gemm_naive(int adt, const void* a, const float* af, int lda, int ldaf, int bdt, const void* b,
                   const float* bf, int ldb, int ldbf, float* r, int ldc, int M, int aColOff, int bColOff,
                   int K, int rRowOff, int bRowOff, int N) {
    for (int i = 0; i < M; i++)
        for (int j = bRowOff, rr = rRowOff; j < bRowOff + N; j++, rr++) {
            float s = 0;
            for (int k = 0; k < K; k++)
                s += jo_get_a(adt, a, af, lda, ldaf, i, aColOff + k) * jo_get_a(bdt, b, bf, ldb, ldbf, j, bColOff + k);
            r[(int64_t)i * ldc + rr] = s;
        }
}

/* ------------------------------------------------------------------------------------
 * element-wise ops behind TensorOperations
 * ---------------------------------------------------------------------------------- */
/* accumulateF32 PTO:2281-2295 */
void jo_accumulate_f32(float* a, const float* b, int offset, int length) {
    for (int i = offset; i < offset + length; i++) a[i] = a[i] + b[i];
}
/* accumulateF32Q4_256 PTO:2297-2325: a[t] += (float)(nib-8) * scale (mul then add). */
void jo_accumulate_f32q4(float* a, const uint8_t* nib_row, const float* scale_row, int offset, int length) {
    for (int i = offset; i < offset + length; i++) a[i] = a[i] + jo_q4_get(nib_row, scale_row, i);
}
/* maccumulateF32 PTO:2083-2097 */
void jo_maccumulate_f32(float* a, const float* b, int offset, int length) {
    for (int i = offset; i < offset + length; i++) a[i] = a[i] * b[i];
}
/* scaleF32 PTO:2499-2514 */
void jo_scale_f32(float factor, float* a, int offset, int length) {
    for (int i = offset; i < offset + length; i++) a[i] = a[i] * factor;
}
/* saxpyF32 PTO:2593-2611: vector body is x.fma(alpha, y) (limit is a multiple of 16 on the path);
 * scalar tail :2607-2610 is y + alpha*x (two roundings) for limit % 16 leftovers. */
void jo_saxpy_f32(float alpha, const float* x, float* y, int xoffset, int yoffset, int limit) {
    int ub = limit - (limit % 16);
    int t = 0;
    for (; t < ub; t++) y[yoffset + t] = fmaf(x[xoffset + t], alpha, y[yoffset + t]);
    for (; t < limit; t++) y[yoffset + t] = y[yoffset + t] + (alpha * x[xoffset + t]);
}
/* batched saxpy PTO:2648-2698: 4 rows at a time chained fma, then singles; per element this
 * is one fma chain over rows in ascending order. x: [rows, ldx]. */
void jo_saxpy_batch_f32(const float* alpha, const float* x, int ldx, float* y, int xoffset, int yoffset,
                        int limit, int aOffset, int xRowOffset, int batchSize) {
    for (int n = 0; n < batchSize; n++)
        jo_saxpy_f32(alpha[aOffset + n], x + (int64_t)(xRowOffset + n) * ldx, y, xoffset, yoffset, limit);
}

/* ------------------------------------------------------------------------------------
 * a10. RMSNorm.forward -- core/model/RMSNorm.java:33-56
 * ---------------------------------------------------------------------------------- */
void jo_rmsnorm(const float* x, const float* w, float weight_adj, int E, float eps, float* out) {
    double ss = 0.0;
    for (int j = 0; j < E; j++) {
        float v = x[j];
        ss += v * v; /* float product, double accumulate */
    }
    ss /= E;
    ss += eps; /* float eps widened */
    ss = 1.0 / sqrt(ss); /* FastMath.sqrt */
    for (int j = 0; j < E; j++) out[j] = (weight_adj + w[j]) * ((float)ss * x[j]);
}

/* ------------------------------------------------------------------------------------
 * a8. RoPE table -- core/math/VectorMath.java:148-165 (precomputeFreqsCis)
 * out: [end * dim/2][2] = (cos, sin)
 * ---------------------------------------------------------------------------------- */
/* LayerNorm.forward core/model/LayerNorm.java:41-67 (GPT-2): float running sums in index order */
void jo_layernorm(const float* x, const float* w, const float* b, int offset, int length, int divisor, float eps, float* out) {
    float sum = 0.0f, sumSq = 0.0f;
    for (int i = offset; i < offset + length; i++) {
        float v = x[i];
        sum += v;
        sumSq += v * v;
    }
    float mean = sum / (float)divisor;
    float variance = sumSq / (float)divisor - mean * mean;
    float invStddev = 1.0f / (float)sqrt((double)(variance + eps));
    for (int i = offset; i < offset + length; i++) out[i] = (x[i] - mean) * invStddev * w[i] + b[i];
}
/* ActivationFunction.eval(GELU) core/math/ActivationFunction.java:32-34 */
float jo_gelu(float x) {
    double v = (double)x;
    return (float)(0.5 * v * (1.0 + tanh(sqrt(2.0 / 3.14159265358979323846) * (v + 0.044715 * pow(v, 3.0)))));
}

void jo_rope_table(int dim, int end, double theta, double scaling, float* out) {
    int half = dim / 2;
    float* freqs = (float*)malloc(sizeof(float) * half);
    float step = 0.0f;
    for (int i = 0; i < half; i++, step = (float)(step + 2.0))
        freqs[i] = (float)((1.0 / pow(theta, (double)(step / dim))) / scaling);
    for (int p = 0; p < end; p++) {
        float t = (float)p;
        for (int i = 0; i < half; i++) {
            float ang = t * freqs[i]; /* outerProduct: float product */
            out[((int64_t)p * half + i) * 2 + 0] = (float)cos((double)ang);
            out[((int64_t)p * half + i) * 2 + 1] = (float)sin((double)ang);
        }
    }
    free(freqs);
}

/* softMax -- core/math/VectorMath.java:69-90 */
void jo_softmax(float* x, int offset, int length) {
    int size = offset + length;
    float max_val = x[offset];
    for (int i = offset + 1; i < size; i++)
        if (x[i] > max_val) max_val = x[i];
    float sum = 0.0f;
    for (int i = offset; i < size; i++) {
        x[i] = (float)exp((double)(x[i] - max_val));
        sum += x[i];
    }
    for (int i = 0; i < size; i++) x[i] = x[i] / sum;
}

/* SiLU -- core/math/ActivationFunction.java:31: (float)(x * (1.0f / (1.0f + exp(-x)))), all in double */
float jo_silu(float x) { return (float)((double)x * (1.0 / (1.0 + exp((double)(-x))))); }

/* KV page geometry -- core/tensor/KvBufferCache.java:224-280 (computePageSize).
 * s = 2 * sizeof(working dtype) * kvSegmentLength; returns layersPerPage, ctxPerPage. */
void jo_kv_page_geometry(int64_t maxPageBytes, int nLayers, int contextLength, int kvLength, int dtypeSize,
                         int* layersPerPage, int* ctxPerPage) {
    int64_t s = 2LL * dtypeSize * kvLength;
    int optL = 1, optC = 1;
    int64_t maxProduct = 0;
    for (int x = nLayers; x >= 1; x--) {
        int64_t y = maxPageBytes / (x * s);
        if (y >= 1 && y <= contextLength) {
            int64_t product = x * y;
            if (product > maxProduct) { optL = x; optC = (int)y; maxProduct = product; }
            if (product < maxProduct) break;
        }
    }
    *layersPerPage = optL;
    *ctxPerPage = optC;
}

/* ------------------------------------------------------------------------------------
 * a13/a14. whole-model forward in the reference's op order.
 *   AbstractModel.forward core/model/AbstractModel.java:314-329
 *   TransformerBlock.forward core/model/TransformerBlock.java:158-215
 *   CausalSelfAttention.forward core/model/CausalSelfAttention.java:145-385 (GQA branch)
 *   MLPBlock.forward core/model/MLPBlock.java:105-166
 *   LlamaModel core/model/llama/LlamaModel.java:67-184 (embedding, maybeQuantize, lm head)
 *   AbstractModel.sample core/model/AbstractModel.java:443-491
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t embedding_length, hidden_length, n_heads, n_kv_heads, head_size;
    int32_t n_layers, vocab_size, context_length;
    int32_t weight_dtype;     /* JO_DT_Q4 (JQ4: I8 activations) | JO_DT_BF16 (BF16 activations) | JO_DT_F32 */
    int32_t layer_start, layer_end; /* DistributedContext.layerStart/End (core/model/DistributedContext.java:75-77) */
    float rms_eps;
    float rope_theta;
    float rope_scaling;
} jo_config;

/* slots 0..11 mirror include/jlama_hip.h (Llama family); 12.. are the extra tensors of the GPT-2 family
 * (core/model/gpt2/GPT2Model.java:53-129): attention / MLP biases, LayerNorm biases, learned position embeddings.
 * For GPT-2, GATE = mlp.c_fc (no UP projection), DOWN = mlp.c_proj, EMBED = wte (also the LM head, :112-127). */
enum { JO_W_Q = 0, JO_W_K, JO_W_V, JO_W_O, JO_W_GATE, JO_W_UP, JO_W_DOWN, JO_W_NORM1, JO_W_NORM2,
       JO_W_EMBED, JO_W_LMHEAD, JO_W_FINALNORM,
       JO_W_QB, JO_W_KB, JO_W_VB, JO_W_OB, JO_W_GATEB, JO_W_DOWNB, JO_W_NORM1B, JO_W_NORM2B, JO_W_WPE, JO_W_FINALNORMB,
       JO_W_COUNT };
enum { JO_ARCH_LLAMA = 0, JO_ARCH_GPT2 = 1 };

typedef struct { int dtype; const void* data; const float* scales; int rows, cols; } jo_weight;

/* optional reference GEMM entry points (oracle/_ref/libjlama_ref.so = nc/simd/vector_simd.c
 * compiled as-is; signatures nc/simd/vector_simd.h:22,30) */
typedef void (*jo_ref_q8q4_fn)(int, const float*, const char*, int, const float*, const char*, int, float*, int,
                               int, int, int, int, int, int, int, int, int);
typedef void (*jo_ref_f32q4_fn)(int, const float*, int, const float*, const char*, int, float*, int, int, int,
                                int, int, int, int, int, int);

typedef struct jo_model {
    jo_config c;
    jo_weight* layer_w; /* [n_layers][JO_W_COUNT] */
    jo_weight global_w[JO_W_COUNT];
    float* rope;        /* [context_length * head_size/2][2] */
    float attention_scale;
    jo_ref_q8q4_fn ref_q8q4;
    jo_ref_f32q4_fn ref_f32q4;
    int ref_flags;
    int nthreads;       /* pchunk split for ref GEMMs (core/math/VectorMath.java:38-67) */
    int kv_head_offset; /* tensor-parallel shard: global index of local kv head 0 (DistributedContext.groupHeadStart) */
    int arch;           /* JO_ARCH_LLAMA: RMSNorm, RoPE, SiLU(gate)*up;  JO_ARCH_GPT2: LayerNorm+bias, wte+wpe, biases, GELU, no RoPE */
} jo_model;

typedef struct jo_session {
    jo_model* m;
    int layers_per_page, ctx_per_page, n_layer_pages, n_ctx_pages;
    float** pages; /* [n_layer_pages * n_ctx_pages] lazily allocated, each [layersPerPage,2,ctxPerPage,kvLength] */
    /* taps (DebugSupport names, core/model/TransformerBlock.java:165-205) */
    int tap_layer;
    float* taps[16];
    int tap_len[16];
} jo_session;

jo_model* jo_model_create(const jo_config* cfg) {
    jo_model* m = (jo_model*)calloc(1, sizeof(jo_model));
    m->c = *cfg;
    m->layer_w = (jo_weight*)calloc((size_t)cfg->n_layers * JO_W_COUNT, sizeof(jo_weight));
    int half = cfg->head_size / 2;
    m->rope = (float*)malloc(sizeof(float) * 2 * (size_t)cfg->context_length * half);
    /* Config ctor core/safetensors/Config.java:270-274 */
    jo_rope_table(cfg->head_size, cfg->context_length, (double)cfg->rope_theta, (double)cfg->rope_scaling, m->rope);
    /* CausalSelfAttention.java:134 */
    m->attention_scale = (float)(1.0 / sqrt((double)cfg->head_size));
    m->nthreads = 1;
    return m;
}
void jo_model_destroy(jo_model* m) {
    if (!m) return;
    free(m->layer_w);
    free(m->rope);
    free(m);
}
/* pointers are borrowed (caller keeps the arrays alive) */
int jo_model_set_weight(jo_model* m, int layer, int which, int dtype, const void* data, const float* scales,
                        int rows, int cols) {
    if (which < 0 || which >= JO_W_COUNT) return -1;
    jo_weight* w = layer < 0 ? &m->global_w[which] : &m->layer_w[(size_t)layer * JO_W_COUNT + which];
    w->dtype = dtype; w->data = data; w->scales = scales; w->rows = rows; w->cols = cols;
    return 0;
}
void jo_model_set_kv_head_offset(jo_model* m, int off) { m->kv_head_offset = off; }
void jo_model_set_arch(jo_model* m, int arch) { m->arch = arch; }
void jo_model_set_ref_gemm(jo_model* m, void* q8q4, void* f32q4, int flags, int nthreads) {
    m->ref_q8q4 = (jo_ref_q8q4_fn)q8q4;
    m->ref_f32q4 = (jo_ref_f32q4_fn)f32q4;
    m->ref_flags = flags;
    m->nthreads = nthreads > 0 ? nthreads : 1;
#ifdef _OPENMP
    /* one team size for every parallel region (GEMM chunks, heads): libgomp tears its thread pool down and rebuilds it
     * whenever consecutive regions ask for different team sizes, which cost milliseconds per layer once the heads loop
     * went parallel (position >= 64) */
    omp_set_num_threads(m->nthreads);
#endif
}

jo_session* jo_session_create(jo_model* m, int64_t max_page_bytes) {
    jo_session* s = (jo_session*)calloc(1, sizeof(jo_session));
    s->m = m;
    int kvLength = m->c.n_kv_heads * m->c.head_size;
    int nl = m->c.layer_end - m->c.layer_start;
    jo_kv_page_geometry(max_page_bytes > 0 ? max_page_bytes : (1 << 23), nl, m->c.context_length, kvLength, 4,
                        &s->layers_per_page, &s->ctx_per_page);
    s->n_layer_pages = (nl + s->layers_per_page - 1) / s->layers_per_page;
    s->n_ctx_pages = (m->c.context_length + s->ctx_per_page - 1) / s->ctx_per_page;
    s->pages = (float**)calloc((size_t)s->n_layer_pages * s->n_ctx_pages, sizeof(float*));
    s->tap_layer = -1;
    return s;
}
void jo_session_destroy(jo_session* s) {
    if (!s) return;
    for (int i = 0; i < s->n_layer_pages * s->n_ctx_pages; i++) free(s->pages[i]);
    free(s->pages);
    for (int i = 0; i < 16; i++) free(s->taps[i]);
    free(s);
}
void jo_session_page_info(jo_session* s, int* out4) {
    out4[0] = s->layers_per_page; out4[1] = s->ctx_per_page; out4[2] = s->n_layer_pages; out4[3] = s->n_ctx_pages;
}

/* KvBuffer.getTensorForPosition core/tensor/KvBufferCache.java:307-321: row (layer, idx, pos) */
static float* jo_kv_row(jo_session* s, int rel_layer, int idx, int pos) {
    int kvLength = s->m->c.n_kv_heads * s->m->c.head_size;
    int lp = rel_layer / s->layers_per_page, cp = pos / s->ctx_per_page;
    int rl = rel_layer % s->layers_per_page, rc = pos % s->ctx_per_page;
    float** slot = &s->pages[(size_t)lp * s->n_ctx_pages + cp];
    if (!*slot) *slot = (float*)calloc((size_t)s->layers_per_page * 2 * s->ctx_per_page * kvLength, sizeof(float));
    return *slot + (((size_t)rl * 2 + idx) * s->ctx_per_page + rc) * kvLength;
}

enum { JO_TAP_INPUT_EMB = 0, JO_TAP_LN_EMB, JO_TAP_QUERY, JO_TAP_KEY, JO_TAP_VALUE, JO_TAP_QUERY_ROPE,
       JO_TAP_KEY_ROPE, JO_TAP_AFTER_ATTENTION, JO_TAP_POST_ATTN, JO_TAP_PRE_FF_NORM, JO_TAP_POST_FF,
       JO_TAP_POST_FF_RES, JO_TAP_COUNT };

static void jo_tap(jo_session* s, int layer, int which, const float* v, int n) {
    if (layer != s->tap_layer) return;
    s->taps[which] = (float*)realloc(s->taps[which], sizeof(float) * (size_t)n);
    memcpy(s->taps[which], v, sizeof(float) * (size_t)n);
    s->tap_len[which] = n;
}
void jo_session_set_tap_layer(jo_session* s, int layer) { s->tap_layer = layer; }
int jo_session_get_tap(jo_session* s, int which, float* out, int n) {
    if (which < 0 || which >= JO_TAP_COUNT || !s->taps[which]) return -1;
    int m = s->tap_len[which] < n ? s->tap_len[which] : n;
    memcpy(out, s->taps[which], sizeof(float) * (size_t)m);
    return m;
}

/* working buffer of a quantized activation batch */
typedef struct { int8_t* q; float* d; uint16_t* h; const float* f; int K; } jo_act;

/* LlamaModel.maybeQuantize core/model/llama/LlamaModel.java:176-184 + dtype policy
 * core/model/AbstractModel.java:119-169: Q4 model => I8, BF16 model => BF16, F32 => copy. */
static void jo_maybe_quantize(jo_model* m, const float* x, int B, int K, jo_act* out) {
    out->K = K; out->f = x; out->q = NULL; out->d = NULL; out->h = NULL;
    if (m->c.weight_dtype == JO_DT_Q4) {
        out->q = (int8_t*)malloc((size_t)B * K);
        out->d = (float*)malloc(sizeof(float) * (size_t)B * (K / JO_BLOCK));
        jo_q8_quantize(x, B, K, 0, K, out->q, K, out->d, K / JO_BLOCK);
    } else if (m->c.weight_dtype == JO_DT_BF16) {
        out->h = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)B * K);
        jo_bf16_quantize(x, (int64_t)B * K, out->h);
    }
}
static void jo_act_free(jo_act* a) { free(a->q); free(a->d); free(a->h); }

/* dotProductChunk over the full N range of weight w, A columns [aColOff, aColOff+K) */
static void jo_weight_gemm(jo_model* m, const jo_act* a, int B, const jo_weight* w, int colOff, int K,
                           float* r, int ldc) {
    int N = w->rows;
    if (w->dtype == JO_DT_Q4 && a->q) {
        if (m->ref_q8q4) {
            /* CPU-baseline leg: reference C SIMD GEMM, N split into nthreads chunks like
             * VectorMath.pchunk (core/math/VectorMath.java:38-67); marshaling as
             * jlama-native/.../NativeSimdTensorOperations.java:96-107,204-223 */
            int T = m->nthreads, chunk = N / T, rem = N % T;
#pragma omp parallel for schedule(static) num_threads(T)
            for (int t = 0; t < T; t++) {
                int n0 = t * chunk, n = (t == T - 1) ? chunk + rem : chunk;
                m->ref_q8q4(m->ref_flags, a->d, (const char*)a->q, colOff, w->scales, (const char*)w->data,
                            colOff / 2, r, 0, B, n0, n, K, a->K, a->K / JO_BLOCK, w->cols / 2,
                            w->cols / JO_BLOCK, ldc);
            }
        } else {
            jo_gemm_i8q4(a->q, a->d, a->K, a->K / JO_BLOCK, (const uint8_t*)w->data, w->scales, w->cols / 2,
                         w->cols / JO_BLOCK, r, ldc, B, colOff, colOff, K, 0, 0, N);
        }
    } else if (w->dtype == JO_DT_Q4) {
        if (m->ref_f32q4) {
            int T = m->nthreads, chunk = N / T, rem = N % T;
#pragma omp parallel for schedule(static) num_threads(T)
            for (int t = 0; t < T; t++) {
                int n0 = t * chunk, n = (t == T - 1) ? chunk + rem : chunk;
                m->ref_f32q4(m->ref_flags, a->f, colOff, w->scales, (const char*)w->data, colOff / 2, r, 0, B,
                             n0, n, K, a->K, w->cols / 2, w->cols / JO_BLOCK, ldc);
            }
        } else {
            jo_gemm_f32q4(a->f, a->K, (const uint8_t*)w->data, w->scales, w->cols / 2, w->cols / JO_BLOCK, r,
                          ldc, B, colOff, colOff, K, 0, 0, N);
        }
    } else if (w->dtype == JO_DT_BF16 && a->h) {
        jo_gemm_bf16(a->h, a->K, (const uint16_t*)w->data, w->cols, r, ldc, B, colOff, colOff, K, 0, 0, N);
    } else if (w->dtype == JO_DT_BF16) {
        jo_gemm_f32bf16(a->f, a->K, (const uint16_t*)w->data, w->cols, r, ldc, B, colOff, colOff, K, 0, 0, N);
    } else {
        jo_gemm_f32(a->f, a->K, (const float*)w->data, w->cols, r, ldc, B, colOff, colOff, K, 0, 0, N);
    }
}

/* read a 1-D norm weight (BF16 or F32; never quantized: AbstractTensor.java:284) */
static void jo_load_norm(const jo_weight* w, int E, float* out) {
    for (int j = 0; j < E; j++)
        out[j] = w->dtype == JO_DT_BF16 ? jo_bf16_to_f32(((const uint16_t*)w->data)[j]) : ((const float*)w->data)[j];
}

/* preAttentionNorm / preFFNorm / output norm of one row: RMSNorm (Llama family, core/model/RMSNorm.java:33-56) or
 * LayerNorm with bias (GPT-2, core/model/LayerNorm.java:41-67) */
static void jo_norm_row(jo_model* m, const float* x, const jo_weight* w, const jo_weight* b, float* nw, float* nb, float* out) {
    int E = m->c.embedding_length;
    jo_load_norm(w, E, nw);
    if (m->arch == JO_ARCH_GPT2) {
        jo_load_norm(b, E, nb);
        jo_layernorm(x, nw, nb, 0, E, E, m->c.rms_eps, out);
    } else {
        jo_rmsnorm(x, nw, 0.0f, E, m->c.rms_eps, out);
    }
}
/* accumulate(result, bias, 0, n) per batch row (TensorOperations.accumulate with a one-row bias, PTO:2150-2218) */
static void jo_add_bias(const jo_weight* b, float* r, int B, int n, float* tmp) {
    if (!b->data) return;
    jo_load_norm(b, n, tmp);
    for (int i = 0; i < B; i++) jo_accumulate_f32(r + (size_t)i * n, tmp, 0, n);
}

/* EmbedInput: LlamaModel.loadInputWeights core/model/llama/LlamaModel.java:67-98.  For a Q4
 * table the row stays Q4 and every consumer reads (nib-8)*scale; BF16 widens by <<16. */
static void jo_embed(jo_model* m, int token, int position, float* out) {
    const jo_weight* w = &m->global_w[JO_W_EMBED];
    int E = m->c.embedding_length;
    if (m->arch == JO_ARCH_GPT2) {
        /* GPT2Model.loadInputWeights core/model/gpt2/GPT2Model.java:53-68: v = wte.get(token, i) + wpe.get(position, i) */
        const float* wte = (const float*)w->data + (size_t)token * E;
        const float* wpe = (const float*)m->global_w[JO_W_WPE].data + (size_t)position * E;
        for (int j = 0; j < E; j++) out[j] = wte[j] + wpe[j];
        return;
    }
    if (w->dtype == JO_DT_Q4) {
        const uint8_t* nr = (const uint8_t*)w->data + (size_t)token * (E / 2);
        const float* sr = w->scales + (size_t)token * (E / JO_BLOCK);
        for (int j = 0; j < E; j++) out[j] = jo_q4_get(nr, sr, j);
    } else if (w->dtype == JO_DT_BF16) {
        for (int j = 0; j < E; j++) out[j] = jo_bf16_to_f32(((const uint16_t*)w->data)[(size_t)token * E + j]);
    } else {
        memcpy(out, (const float*)w->data + (size_t)token * E, sizeof(float) * E);
    }
}

/* ---- one TransformerBlock in two halves, split where the reference's tensor-parallel shards synchronise
 * (tensorReducer: CausalSelfAttention.java:378, MLPBlock.java:160).  A shard model carries its LOCAL heads / kv heads /
 * hidden rows (DistributedContext.java:79-98); the partial [B,E] results are summed over shards BEFORE the residual. */

/* attention half: preAttentionNorm -> q,k,v -> KV write + RoPE -> per-head attention -> o-projection over this
 * shard's attention segment.  x: [B,E] in;  att_out: [B,E] out (no residual). */
static void jo_layer_attn(jo_session* s, int li, const float* x, int B, int start_pos, float* att_out) {
    jo_model* m = s->m;
    const jo_config* c = &m->c;
    int E = c->embedding_length, hs = c->head_size;
    int A = c->n_heads * hs, KV = c->n_kv_heads * hs, half = hs / 2;
    int group = c->n_heads / c->n_kv_heads;
    int rel = li - c->layer_start;
    const jo_weight* W = &m->layer_w[(size_t)li * JO_W_COUNT];
    float* ln = (float*)malloc(sizeof(float) * (size_t)B * E);
    float* q = (float*)malloc(sizeof(float) * (size_t)B * A);
    float* k = (float*)malloc(sizeof(float) * (size_t)B * KV);
    float* v = (float*)malloc(sizeof(float) * (size_t)B * KV);
    float* val = (float*)malloc(sizeof(float) * (size_t)B * A);
    float* nw = (float*)malloc(sizeof(float) * (size_t)(A > E ? A : E));
    float* nb = (float*)malloc(sizeof(float) * (size_t)E);
    int max_ctx_alloc = ((start_pos + B) / s->ctx_per_page + 1) * s->ctx_per_page;
    float* attn_all = (float*)malloc(sizeof(float) * (size_t)max_ctx_alloc * (size_t)jo_num_threads());

    jo_tap(s, li, JO_TAP_INPUT_EMB, x, B * E);
    /* preAttentionNorm TransformerBlock.java:167 */
    for (int b = 0; b < B; b++) jo_norm_row(m, x + (size_t)b * E, &W[JO_W_NORM1], &W[JO_W_NORM1B], nw, nb, ln + (size_t)b * E);
    jo_tap(s, li, JO_TAP_LN_EMB, ln, B * E);
    jo_act qa;
    jo_maybe_quantize(m, ln, B, E, &qa); /* :172 */
    /* Q,K,V GEMMs CausalSelfAttention.java:161-171 */
    jo_weight_gemm(m, &qa, B, &W[JO_W_Q], 0, E, q, A);
    jo_weight_gemm(m, &qa, B, &W[JO_W_K], 0, E, k, KV);
    jo_weight_gemm(m, &qa, B, &W[JO_W_V], 0, E, v, KV);
    jo_act_free(&qa);
    /* queryAttnBias / keyAttnBias / valueAttnBias (CausalSelfAttention.java:183-191; present for GPT-2) */
    jo_add_bias(&W[JO_W_QB], q, B, A, nw);
    jo_add_bias(&W[JO_W_KB], k, B, KV, nw);
    jo_add_bias(&W[JO_W_VB], v, B, KV, nw);
    jo_tap(s, li, JO_TAP_QUERY, q, B * A);
    jo_tap(s, li, JO_TAP_KEY, k, B * KV);
    jo_tap(s, li, JO_TAP_VALUE, v, B * KV);
    memset(val, 0, sizeof(float) * (size_t)B * A); /* TensorCache buffers are zeroed on release (TensorCache.java:104-111) */

    for (int bi = 0, position = start_pos; bi < B; bi++, position++) {
        float* key = jo_kv_row(s, rel, 0, position);
        float* vrow = jo_kv_row(s, rel, 1, position);
        memcpy(key, k + (size_t)bi * KV, sizeof(float) * KV);  /* :226-241 copyFrom (KV dtype == working dtype F32) */
        memcpy(vrow, v + (size_t)bi * KV, sizeof(float) * KV);
        float* query = q + (size_t)bi * A;
        float* value = val + (size_t)bi * A;
        /* RoPE :247-286 (GQA branch). table index poffset + g, g over kvHead*hs + [0,half) with the GLOBAL kv head
         * index (dctx.groupHeadStart.. :275) => effective position pos + 2*kvHead (quirk kept). */
        int poffset = position * half + m->kv_head_offset * hs;
        /* c.ropeFreqs.ifPresent(...) :247: GPT-2 has no RoPE (learned position embeddings) */
        for (int h = 0; m->arch != JO_ARCH_GPT2 && h < c->n_heads; h++) {
            int offset = h * hs, goffset = (h / group) * hs; /* Config.maybeMapToGroupHead */
            for (int i = offset, gg = goffset; i < offset + half; i++, gg++) {
                float q0 = query[i], q1 = query[i + half];
                float fcr = m->rope[(size_t)(poffset + gg) * 2], fci = m->rope[(size_t)(poffset + gg) * 2 + 1];
                query[i] = q0 * fcr - q1 * fci;
                query[i + half] = q0 * fci + q1 * fcr;
            }
        }
        for (int h = 0; m->arch != JO_ARCH_GPT2 && h < c->n_kv_heads; h++) {
            int offset = h * hs;
            for (int i = offset; i < offset + half; i++) {
                float k0 = key[i], k1 = key[i + half];
                float fcr = m->rope[(size_t)(poffset + i) * 2], fci = m->rope[(size_t)(poffset + i) * 2 + 1];
                key[i] = k0 * fcr - k1 * fci;
                key[i + half] = k0 * fci + k1 * fcr;
            }
        }
        if (bi == B - 1) {
            jo_tap(s, li, JO_TAP_QUERY_ROPE, query, A);
            jo_tap(s, li, JO_TAP_KEY_ROPE, key, KV);
        }
        /* attention per head :314-356 */
        int npages = position / s->ctx_per_page + 1;
        for (int pg = 0; pg < npages; pg++) { (void)jo_kv_row(s, rel, 0, pg * s->ctx_per_page); } /* materialise pages before the parallel loop */
        /* VectorMath.pfor(headStart, headEnd, ...) core/math/VectorMath.java:34-36 */
#pragma omp parallel for schedule(static) if (c->n_heads >= 8 && position >= 64)
        for (int h = 0; h < c->n_heads; h++) {
            float* attn = attn_all + (size_t)max_ctx_alloc * (size_t)jo_thread_id();
            int xoffset = (h / group) * hs, yoffset = h * hs;
            for (int pg = 0; pg < npages; pg++) {
                int len = s->ctx_per_page, off = pg * len;
                int size = pg == npages - 1 ? (position + 1) - off : len;
                float* kpage = jo_kv_row(s, rel, 0, off);
                /* batchDotProduct(attn, query, kvp[i], yoffset, xoffset, headSize, offset, 0, size) :328-329 */
                jo_gemm_f32(query, A, kpage, KV, attn, 0, 1, yoffset, xoffset, hs, off, 0, size);
            }
            jo_scale_f32(m->attention_scale, attn, 0, position + 1); /* :332 */
            jo_softmax(attn, 0, position + 1);                      /* :345 */
            for (int pg = 0; pg < npages; pg++) {
                int len = s->ctx_per_page, off = pg * len;
                int size = pg == npages - 1 ? (position + 1) - off : len;
                float* vpage = jo_kv_row(s, rel, 1, off);
                /* saxpy(attn, vvp[i], value, xoffset, yoffset, headSize, offset, 0, size) :349-354 */
                jo_saxpy_batch_f32(attn, vpage, KV, value, xoffset, yoffset, hs, off, 0, size);
            }
        }
    }
    jo_tap(s, li, JO_TAP_AFTER_ATTENTION, val, B * A);
    /* O projection :363-376 (over this shard's attention segment) */
    jo_act va;
    jo_maybe_quantize(m, val, B, A, &va);
    jo_weight_gemm(m, &va, B, &W[JO_W_O], 0, A, att_out, E);
    jo_act_free(&va);
    jo_add_bias(&W[JO_W_OB], att_out, B, E, nw);   /* outputProjectionBias :378-380 (after the reducer) */
    jo_tap(s, li, JO_TAP_POST_ATTN, att_out, B * E);
    free(ln); free(q); free(k); free(v); free(val); free(nw); free(nb); free(attn_all);
}

/* feed-forward half: preFFNorm -> gate, up -> SiLU*up -> down over this shard's hidden segment.
 * att_res: [B,E] in (= attention output + residual);  ff: [B,E] out (no residual). */
static void jo_layer_ffn(jo_session* s, int li, const float* att_res, int B, float* ff) {
    jo_model* m = s->m;
    const jo_config* c = &m->c;
    int E = c->embedding_length, H = c->hidden_length;
    const jo_weight* W = &m->layer_w[(size_t)li * JO_W_COUNT];
    float* ln = (float*)malloc(sizeof(float) * (size_t)B * E);
    float* g = (float*)malloc(sizeof(float) * (size_t)B * H);
    float* u = (float*)malloc(sizeof(float) * (size_t)B * H);
    float* nw = (float*)malloc(sizeof(float) * (size_t)(H > E ? H : E));
    float* nb = (float*)malloc(sizeof(float) * (size_t)E);
    /* preFFNorm :187 */
    for (int b = 0; b < B; b++) jo_norm_row(m, att_res + (size_t)b * E, &W[JO_W_NORM2], &W[JO_W_NORM2B], nw, nb, ln + (size_t)b * E);
    jo_tap(s, li, JO_TAP_PRE_FF_NORM, ln, B * E);
    jo_act fa;
    jo_maybe_quantize(m, ln, B, E, &fa); /* :192 */
    /* MLPBlock.forward MLPBlock.java:117-142 */
    jo_weight_gemm(m, &fa, B, &W[JO_W_GATE], 0, E, g, H);
    if (W[JO_W_UP].data) jo_weight_gemm(m, &fa, B, &W[JO_W_UP], 0, E, u, H);   /* upProjectionWeights != null :119-126 */
    jo_act_free(&fa);
    jo_add_bias(&W[JO_W_GATEB], g, B, H, nw);                                  /* fullyConnectedBias :128-130 */
    if (m->arch == JO_ARCH_GPT2) { for (size_t t = 0; t < (size_t)B * H; t++) g[t] = jo_gelu(g[t]); }   /* c.activationFunction = GELU (GPT2Config.java:49) */
    else { for (size_t t = 0; t < (size_t)B * H; t++) g[t] = jo_silu(g[t]); }
    if (W[JO_W_UP].data) jo_maccumulate_f32(g, u, 0, B * H);
    jo_act ha;
    jo_maybe_quantize(m, g, B, H, &ha); /* :144 */
    jo_weight_gemm(m, &ha, B, &W[JO_W_DOWN], 0, H, ff, E);
    jo_act_free(&ha);
    jo_add_bias(&W[JO_W_DOWNB], ff, B, E, nw);                                 /* projectionBias :163 */
    jo_tap(s, li, JO_TAP_POST_FF, ff, B * E);
    free(ln); free(g); free(u); free(nw); free(nb);
}

/* forward a batch of B rows through layers [layer_start, layer_end).  x: [B,E] in/out.
 * If tokens != NULL the rows are first filled from the embedding table. */
int jo_forward(jo_session* s, const int32_t* tokens, float* x, int B, int start_pos) {
    jo_model* m = s->m;
    const jo_config* c = &m->c;
    int E = c->embedding_length;
    if (tokens)
        for (int b = 0; b < B; b++) jo_embed(m, tokens[b], start_pos + b, x + (size_t)b * E);
    float* att_out = (float*)malloc(sizeof(float) * (size_t)B * E);
    float* ff = (float*)malloc(sizeof(float) * (size_t)B * E);
    for (int li = c->layer_start; li < c->layer_end; li++) {
        jo_layer_attn(s, li, x, B, start_pos, att_out);
        /* residual TransformerBlock.java:185: lnattn += embedding */
        for (int b = 0; b < B; b++) jo_accumulate_f32(att_out + (size_t)b * E, x + (size_t)b * E, 0, E);
        jo_layer_ffn(s, li, att_out, B, ff);
        /* residual TransformerBlock.java:203: lnpostFF += lnattn */
        for (int b = 0; b < B; b++) jo_accumulate_f32(ff + (size_t)b * E, att_out + (size_t)b * E, 0, E);
        jo_tap(s, li, JO_TAP_POST_FF_RES, ff, B * E);
        memcpy(x, ff, sizeof(float) * (size_t)B * E);
    }
    free(att_out); free(ff);
    return 0;
}

/* The two halves on their own, for a tensor-parallel engine that reduces the partial results between them (one shard
 * per process; the sum over shards is the caller's all-reduce). */
int jo_tp_attn(jo_session* s, int layer, const float* x, int B, int start_pos, float* partial) {
    if (layer < s->m->c.layer_start || layer >= s->m->c.layer_end) return -1;
    jo_layer_attn(s, layer, x, B, start_pos, partial);
    return 0;
}
int jo_tp_ffn(jo_session* s, int layer, const float* att_res, int B, float* partial) {
    if (layer < s->m->c.layer_start || layer >= s->m->c.layer_end) return -1;
    jo_layer_ffn(s, layer, att_res, B, partial);
    return 0;
}
void jo_embed_rows(jo_model* m, const int32_t* tokens, int B, float* x) {
    for (int b = 0; b < B; b++) jo_embed(m, tokens[b], b, x + (size_t)b * m->c.embedding_length);
}

/* All model shards in one process, in lock step (head split, DistributedContext.java:79-98): shard r holds heads
 * [r*heads/N, ...), kv heads [r*kvHeads/N, ...), hidden rows [r*H/N, ...) and the matching K columns of the o / down
 * projections; partial results are summed in shard order 0..N-1, then the residual is added
 * (CausalSelfAttention.java:378, MLPBlock.java:160, TransformerBlock.java:185,203). */
int jo_forward_tp(jo_session** shards, int nshards, const int32_t* tokens, float* x, int B, int start_pos) {
    jo_model* m0 = shards[0]->m;
    const jo_config* c = &m0->c;
    int E = c->embedding_length;
    if (tokens) jo_embed_rows(m0, tokens, B, x);
    float* part = (float*)malloc(sizeof(float) * (size_t)B * E);
    float* att_out = (float*)malloc(sizeof(float) * (size_t)B * E);
    float* ff = (float*)malloc(sizeof(float) * (size_t)B * E);
    for (int li = c->layer_start; li < c->layer_end; li++) {
        for (int r = 0; r < nshards; r++) {
            jo_layer_attn(shards[r], li, x, B, start_pos, r == 0 ? att_out : part);
            if (r > 0) jo_accumulate_f32(att_out, part, 0, B * E);
        }
        for (int b = 0; b < B; b++) jo_accumulate_f32(att_out + (size_t)b * E, x + (size_t)b * E, 0, E);
        for (int r = 0; r < nshards; r++) {
            jo_layer_ffn(shards[r], li, att_out, B, r == 0 ? ff : part);
            if (r > 0) jo_accumulate_f32(ff, part, 0, B * E);
        }
        for (int b = 0; b < B; b++) jo_accumulate_f32(ff + (size_t)b * E, att_out + (size_t)b * E, 0, E);
        memcpy(x, ff, sizeof(float) * (size_t)B * E);
    }
    free(part); free(att_out); free(ff);
    return 0;
}

/* AbstractModel.sample core/model/AbstractModel.java:443-491.  last_row: [E].  logits: [V] out. */
int jo_sample(jo_model* m, const float* last_row, float temperature, float uniform, float* logits) {
    const jo_config* c = &m->c;
    int E = c->embedding_length, V = c->vocab_size;
    float* nw = (float*)malloc(sizeof(float) * E);
    float* nb = (float*)malloc(sizeof(float) * E);
    float* emb = (float*)malloc(sizeof(float) * E);
    jo_norm_row(m, last_row, &m->global_w[JO_W_FINALNORM], &m->global_w[JO_W_FINALNORMB], nw, nb, emb);   /* getOutputLayerNorm */
    /* LM head: the F32 normed row is NOT re-quantized (:443-449) => F32xQ4 / F32xBF16 / F32xF32 */
    jo_act a = { NULL, NULL, NULL, emb, E };
    const jo_weight* w = m->global_w[JO_W_LMHEAD].data ? &m->global_w[JO_W_LMHEAD] : &m->global_w[JO_W_EMBED];
    jo_weight_gemm(m, &a, 1, w, 0, E, logits, V);
    free(nw); free(nb); free(emb);
    int maxi = INT32_MIN;
    double maxv = -INFINITY;
    for (int i = 0; i < V; i++) {
        float v = logits[i];
        if (v > maxv) { maxi = i; maxv = v; }
    }
    if (temperature == 0.0f) return maxi;
    float sum = 0;
    for (int i = 0; i < V; i++) {
        float v = (float)exp(((double)logits[i] - maxv) / (double)temperature);
        sum += v;
        logits[i] = v;
    }
    float acc = 0;
    for (int i = 0; i < V; i++) {
        float v = logits[i] / sum;
        acc += v;
        if (acc >= uniform) return i;
    }
    return V - 1;
}

/* AbstractModel.generate core/model/AbstractModel.java:515-646 at the token-id level:
 * batchForward(prompt) in chunks of max_batch (jlama.max_batch_size=256, :57,:304) -> sample ->
 * decode loop.  Writes n_gen token ids; returns number generated.  times_ms[0]=prompt,
 * times_ms[1]=decode (clock starts after the first sampled token, :589). */
int jo_generate(jo_session* s, const int32_t* prompt, int n_prompt, int n_gen, float temperature,
                int32_t* out_tokens, float* logits_last, double* times_ms) {
    jo_model* m = s->m;
    int E = m->c.embedding_length, V = m->c.vocab_size;
    const int MAXB = 256;
    float* x = (float*)malloc(sizeof(float) * (size_t)MAXB * E);
    float* logits = (float*)malloc(sizeof(float) * (size_t)V);
    double t0 = 0, t1 = 0, t2 = 0;
#ifdef _OPENMP
    t0 = omp_get_wtime();
#endif
    int lastB = 0;
    for (int i = 0; i < n_prompt; i += MAXB) {
        int B = n_prompt - i < MAXB ? n_prompt - i : MAXB;
        jo_forward(s, prompt + i, x, B, i);
        lastB = B;
    }
    int next = jo_sample(m, x + (size_t)(lastB - 1) * E, temperature, 0.5f, logits);
#ifdef _OPENMP
    t1 = omp_get_wtime();
#endif
    int n = 0;
    out_tokens[n++] = next;
    for (int pos = n_prompt; n < n_gen; pos++) {
        int32_t tok = next;
        jo_forward(s, &tok, x, 1, pos);
        next = jo_sample(m, x, temperature, 0.5f, logits);
        out_tokens[n++] = next;
    }
#ifdef _OPENMP
    t2 = omp_get_wtime();
#endif
    if (logits_last) memcpy(logits_last, logits, sizeof(float) * (size_t)V);
    if (times_ms) { times_ms[0] = (t1 - t0) * 1e3; times_ms[1] = (t2 - t1) * 1e3; }
    free(x); free(logits);
    return n;
}

