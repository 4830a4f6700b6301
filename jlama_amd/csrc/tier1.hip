// tier1.hip -- part of libjlamahip.so (C ABI: include/jlama_hip.h).  Tier 1: the TensorOperations provider entry points on host buffers / registered tensors.
#include "jh_host.h"
#include "jh_launch.h"

namespace {
// Shared Tier-1 GEMM driver.  a_es/b_es: element size in bytes of A / B storage rows (Q4: ldb already in bytes).
int tier1_gemm(int kind, int64_t b_id, int64_t bf_id, const void* a, const float* af, int aoffset, const void* b,
               const float* bf, int boffset, float* r, int roffset, int m, int n0, int n, int k, int lda, int ldaf,
               int ldb, int ldbf, int ldc, uint16_t* cr = nullptr) {
    if (m < 0 || n < 0 || k < 0 || (!r && !cr) || !a) return set_err(JH_ERR_INVALID, "gemm: bad argument");
    const bool q4 = (kind == G_Q8Q4 || kind == G_F32Q4);
    if (q4 && (k % QB)) return set_err(JH_ERR_INVALID, "gemm: K must be a multiple of 32 for Q4/Q8 blocks");
    if (m == 0 || n == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    const size_t a_es = (kind == G_Q8Q4) ? 1 : (kind == G_BF16 ? 2 : 4);
    const size_t b_es = q4 ? 1 : ((kind == G_BF16 || kind == G_F32BF16) ? 2 : 4);
    // ---- A (+ scales)
    const size_t a_elems = (size_t)lda * (m - 1) + aoffset + k;
    void *dA = nullptr, *dAf = nullptr;
    JHCHK(dev_buf(0, a_elems * a_es, &dA));
    HIPCHK(hipMemcpyAsync(dA, a, a_elems * a_es, hipMemcpyHostToDevice, st));
    if (kind == G_Q8Q4) {
        if (!af) return set_err(JH_ERR_INVALID, "gemm_q8_q4: af is null");
        const size_t af_elems = (size_t)ldaf * (m - 1) + aoffset / QB + k / QB;
        JHCHK(dev_buf(1, af_elems * 4, &dAf));
        HIPCHK(hipMemcpyAsync(dAf, af, af_elems * 4, hipMemcpyHostToDevice, st));
    }
    // ---- B (+ scales): registered (whole tensor resident) or copied rows [n0, n0+n)
    const uint8_t* dB = nullptr;
    const float* dBf = nullptr;
    if (b_id >= 0) {
        dB = (const uint8_t*)reg_ptr(b_id);
        if (!dB) return set_err(JH_ERR_INVALID, "gemm: unknown b_id");
    } else {
        if (!b) return set_err(JH_ERR_INVALID, "gemm: b is null and not registered");
        const size_t row0 = (size_t)ldb * n0 * b_es;
        const size_t bytes = ((size_t)ldb * (n - 1) + boffset + (q4 ? k / 2 : k)) * b_es;
        void* t = nullptr;
        JHCHK(dev_buf(2, bytes, &t));
        HIPCHK(hipMemcpyAsync(t, (const uint8_t*)b + row0, bytes, hipMemcpyHostToDevice, st));
        dB = (const uint8_t*)t - row0;  // so that kernel-side ldb*j indexing lands in the copied window
    }
    if (q4) {
        if (bf_id >= 0) {
            dBf = (const float*)reg_ptr(bf_id);
            if (!dBf) return set_err(JH_ERR_INVALID, "gemm: unknown bf_id");
        } else {
            if (!bf) return set_err(JH_ERR_INVALID, "gemm: bf is null and not registered");
            const size_t row0 = (size_t)ldbf * n0;
            const size_t elems = (size_t)ldbf * (n - 1) + (boffset * 2) / QB + k / QB;
            void* t = nullptr;
            JHCHK(dev_buf(3, elems * 4, &t));
            HIPCHK(hipMemcpyAsync(t, bf + row0, elems * 4, hipMemcpyHostToDevice, st));
            dBf = (const float*)t - row0;
        }
    }
    // ---- R
    const long long cmin = (long long)n0 - roffset;
    if (cmin < 0) return set_err(JH_ERR_INVALID, "gemm: n0 - roffset < 0");
    const size_t r_elems = (size_t)ldc * (m - 1) + (size_t)cmin + n;
    void* dR = nullptr;
    JHCHK(dev_buf(4, r_elems * 4, &dR));

    // JH_STRICT_ORDER=1: every float accumulation in the Panama-512 order (gemm_reford_kernel, jh_p16.h) -- any M, any window;
    // results are bit-identical to the reference's compiled C GEMM where that keeps Panama's order (F32xQ4, F32xF32, BF16xBF16,
    // F32xBF16) and to the restated Panama provider for I8xQ4 (the C twin groups its floats differently)
    const bool strict = opt_int("JH_STRICT_ORDER", 0) != 0;
    if (strict && !q4 && (k % 16))
        return set_err(JH_ERR_UNSUPPORTED, "gemm (reference order): K must be a multiple of the 16-lane species (Panama reads whole vectors)");
    if (strict && q4 && ((aoffset % QB) || (boffset % 16)))
        return set_err(JH_ERR_UNSUPPORTED, "gemm (reference order): column offsets must be whole Q blocks");
    bool fast = !strict && (m == 1) && q4 && (aoffset % QB == 0) && (boffset % 16 == 0) && (ldb % 16 == 0) &&
                !opt_int("JH_TIER1_GENERIC", 0);
    if (strict) {
        GemmParams g{dA, (const float*)dAf, dB, dBf, (float*)dR, aoffset, boffset, roffset, m, n0, n, k,
                     lda, ldaf, ldb, ldbf, ldc};
        const long long outs = (long long)m * n;
        const unsigned grid = (unsigned)((outs + 15) / 16);   // 4 waves x 4 outputs per 256-thread workgroup
        switch (kind) {
            case G_Q8Q4: hipLaunchKernelGGL((gemm_reford_kernel<G_Q8Q4>), dim3(grid), dim3(256), 0, st, g); break;
            case G_F32Q4: hipLaunchKernelGGL((gemm_reford_kernel<G_F32Q4>), dim3(grid), dim3(256), 0, st, g); break;
            case G_F32: hipLaunchKernelGGL((gemm_reford_kernel<G_F32>), dim3(grid), dim3(256), 0, st, g); break;
            case G_BF16: hipLaunchKernelGGL((gemm_reford_kernel<G_BF16>), dim3(grid), dim3(256), 0, st, g); break;
            default: hipLaunchKernelGGL((gemm_reford_kernel<G_F32BF16>), dim3(grid), dim3(256), 0, st, g); break;
        }
        HIPCHK(hipGetLastError());
        fast = true;
    } else if (fast) {
        GemvParams p;
        memset(&p, 0, sizeof(p));
        p.nrows = n;
        p.K = k;
        p.ldb = ldb;
        p.ldbf = ldbf;
        p.w = dB + (size_t)ldb * n0 + boffset;
        p.ws = dBf + (size_t)ldbf * n0 + (boffset * 2) / QB;
        p.out = (float*)dR + cmin;
        LaunchCfg cfg{opt_int("JH_GEMV_R", 0), opt_int("JH_GEMV_WAVES", 0), 0, opt_int("JH_GEMV_PIPE", -1)};
        LaunchCfg cfgf{opt_int("JH_GEMV_R", 0), 8, g_cu_count * 2, 1};
        if (kind == G_Q8Q4) {
            p.aq = (const int8_t*)dA + aoffset;
            p.ad = (const float*)dAf + aoffset / QB;
            JHCHK((launch_gemv_i8q4<PRO_Q8, EPI_STORE>(p, cfg, st)));
        } else {
            p.x = (const float*)dA + aoffset;
            if (aoffset % 4) fast = false;
            else JHCHK((launch_gemv_f32q4<PRO_F32>(p, cfgf, nullptr, st)));
        }
    }
    if (!fast && kind == G_Q8Q4 && m >= 2 && m <= 256 && (n % 32) == 0 && (aoffset % QB) == 0 && (boffset % 16) == 0 && (lda % 16) == 0 &&
        (ldb % 16) == 0 && !opt_int("JH_TIER1_GENERIC", 0)) {
        // batched I8 x Q4 GEMM on the matrix cores (prefill shape), exact integer block sums
        MfmaQ4Params g;
        g.a = (const int8_t*)dA + aoffset; g.af = (const float*)dAf + aoffset / QB;
        g.w = dB + boffset; g.ws = dBf + (boffset * 2) / QB; g.c = (float*)dR; g.resid = nullptr;
        g.m = m; g.n0 = n0; g.n = n; g.k = k; g.lda = lda; g.ldaf = ldaf; g.ldb = ldb; g.ldbf = ldbf; g.ldc = ldc; g.roffset = roffset;
        const int rcm = launch_gemm_q8q4_mfma(g, st);
        if (rcm == JH_OK) fast = true;
        else if (rcm != JH_ERR_UNSUPPORTED) return rcm;
    }
    if (!fast && kind == G_BF16 && m >= 2 && m <= 256 && (k % MG_KS) == 0 && (n % 32) == 0 && (aoffset % 8) == 0 && (boffset % 8) == 0 &&
        (lda % 8) == 0 && (ldb % 8) == 0 && !opt_int("JH_TIER1_GENERIC", 0)) {
        // batched BF16 GEMM on the matrix cores (prefill shape)
        MfmaGemmParams g;
        g.a = (const uint16_t*)dA + aoffset; g.w = (const uint16_t*)dB + boffset; g.c = (float*)dR;
        g.m = m; g.n0 = n0; g.n = n; g.k = k; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.roffset = roffset; g.resid = nullptr;
        g.ws = nullptr; g.nsplit = 1;
        { void* wsp = nullptr; JHCHK(dev_buf(7, BF16_SPLITK_WS_BYTES, &wsp)); g.ws = (float*)wsp; }
        JHCHK(launch_gemm_bf16_mfma(g, st));
        fast = true;
    }
    if (!fast) {
        GemmParams g{dA, (const float*)dAf, dB, dBf, (float*)dR, aoffset, boffset, roffset, m, n0, n, k,
                     lda, ldaf, ldb, ldbf, ldc};
        if (!q4) { g.ldb = ldb; }
        const long long waves = (long long)m * n;
        const int grid = (int)((waves + 3) / 4);
        switch (kind) {
            case G_Q8Q4: hipLaunchKernelGGL((gemm_generic_kernel<G_Q8Q4>), dim3(grid), dim3(256), 0, st, g); break;
            case G_F32Q4: hipLaunchKernelGGL((gemm_generic_kernel<G_F32Q4>), dim3(grid), dim3(256), 0, st, g); break;
            case G_F32: hipLaunchKernelGGL((gemm_generic_kernel<G_F32>), dim3(grid), dim3(256), 0, st, g); break;
            case G_BF16: hipLaunchKernelGGL((gemm_generic_kernel<G_BF16>), dim3(grid), dim3(256), 0, st, g); break;
            default: hipLaunchKernelGGL((gemm_generic_kernel<G_F32BF16>), dim3(grid), dim3(256), 0, st, g); break;
        }
        HIPCHK(hipGetLastError());
    }
    if (cr) {   // BF16 result tensor (vector_simd.c:1060-1064): round on the device, ship 2 bytes per element
        void* dC = nullptr;
        JHCHK(dev_buf(5, r_elems * 2, &dC));
        hipLaunchKernelGGL(store_bf16_2d_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)m), dim3(256), 0, st, (const float*)dR + cmin,
                           (uint16_t*)dC + cmin, n, ldc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy2DAsync(cr + cmin, (size_t)ldc * 2, (uint16_t*)dC + cmin, (size_t)ldc * 2, (size_t)n * 2, m, hipMemcpyDeviceToHost, st));
    } else {
        HIPCHK(hipMemcpy2DAsync(r + cmin, (size_t)ldc * 4, (float*)dR + cmin, (size_t)ldc * 4, (size_t)n * 4, m,
                                hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}

// element-wise Tier-1 helper: a (in/out) and b windows copied, op applied, a copied back
template <int OP>
int tier1_ew(float* a, const float* b, float f, int offset, int length) {
    if (length < 0 || !a) return set_err(JH_ERR_INVALID, "elementwise: bad argument");
    if (length == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dA = nullptr, *dB = nullptr;
    JHCHK(dev_buf(0, (size_t)length * 4, &dA));
    HIPCHK(hipMemcpyAsync(dA, a + offset, (size_t)length * 4, hipMemcpyHostToDevice, st));
    if (b) {
        JHCHK(dev_buf(1, (size_t)length * 4, &dB));
        HIPCHK(hipMemcpyAsync(dB, b + offset, (size_t)length * 4, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL((ew_kernel<OP>), dim3((length + 255) / 256), dim3(256), 0, st, (float*)dA, (const float*)dB, f, length);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(a + offset, dA, (size_t)length * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
}  // namespace

extern "C" {

int jh_gemm_q8_q4(int64_t b_id, int64_t bf_id, const float* af, const int8_t* a, int aoffset, const float* bf,
                  const uint8_t* b, int boffset, float* r, int roffset, int m, int n0, int n, int k, int lda,
                  int ldaf, int ldb, int ldbf, int ldc) {
    return tier1_gemm(G_Q8Q4, b_id, bf_id, a, af, aoffset, b, bf, boffset, r, roffset, m, n0, n, k, lda, ldaf, ldb,
                      ldbf, ldc);
}
int jh_gemm_f32_q4(int64_t b_id, int64_t bf_id, const float* a, int aoffset, const float* bf, const uint8_t* b,
                   int boffset, float* r, int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldbf,
                   int ldc) {
    return tier1_gemm(G_F32Q4, b_id, bf_id, a, nullptr, aoffset, b, bf, boffset, r, roffset, m, n0, n, k, lda, 0,
                      ldb, ldbf, ldc);
}
int jh_gemm_f32(int64_t b_id, const float* a, int aoffset, const float* b, int boffset, float* r, int roffset, int m,
                int n0, int n, int k, int lda, int ldb, int ldc) {
    return tier1_gemm(G_F32, b_id, -1, a, nullptr, aoffset, b, nullptr, boffset, r, roffset, m, n0, n, k, lda, 0, ldb,
                      0, ldc);
}
int jh_gemm_bf16(int64_t b_id, const uint16_t* a, int aoffset, const uint16_t* b, int boffset, uint16_t* cr, float* r,
                 int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc) {
    return tier1_gemm(G_BF16, b_id, -1, a, nullptr, aoffset, b, nullptr, boffset, r, roffset, m, n0, n, k, lda, 0, ldb,
                      0, ldc, cr);
}
int jh_gemm_f32_bf16(int64_t b_id, const float* a, int aoffset, const uint16_t* b, int boffset, uint16_t* cr, float* r,
                     int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc) {
    return tier1_gemm(G_F32BF16, b_id, -1, a, nullptr, aoffset, b, nullptr, boffset, r, roffset, m, n0, n, k, lda, 0,
                      ldb, 0, ldc, cr);
}
int jh_gemm_f32_batch(int batch_num, const int64_t* b_ids, const float* a, int aoffset, const float* const* b, int boffset,
                      float* const* r, int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc) {
    if (batch_num < 0 || !r) return set_err(JH_ERR_INVALID, "gemm_f32_batch: bad argument");
    for (int i = 0; i < batch_num; i++)
        JHCHK(jh_gemm_f32(b_ids ? b_ids[i] : -1, a, aoffset, b ? b[i] : nullptr, boffset, r[i], roffset, m, n0, n, k, lda, ldb, ldc));
    return JH_OK;
}
int jh_gemm_bf16_batch(int batch_num, const int64_t* b_ids, const uint16_t* a, int aoffset, const uint16_t* const* b,
                       int boffset, uint16_t* const* cr, float* const* r, int roffset, int m, int n0, int n, int k, int lda,
                       int ldb, int ldc) {
    if (batch_num < 0 || (!r && !cr)) return set_err(JH_ERR_INVALID, "gemm_bf16_batch: bad argument");
    for (int i = 0; i < batch_num; i++)   // vector_simd.c:1256-1261
        JHCHK(jh_gemm_bf16(b_ids ? b_ids[i] : -1, a, aoffset, b ? b[i] : nullptr, boffset, cr ? cr[i] : nullptr, r ? r[i] : nullptr,
                           roffset, m, n0, n, k, lda, ldb, ldc));
    return JH_OK;
}
int jh_gemm_f32_bf16_batch(int batch_num, const int64_t* b_ids, const float* a, int aoffset, const uint16_t* const* b,
                           int boffset, uint16_t* const* cr, float* const* r, int roffset, int m, int n0, int n, int k,
                           int lda, int ldb, int ldc) {
    if (batch_num < 0 || (!r && !cr)) return set_err(JH_ERR_INVALID, "gemm_f32_bf16_batch: bad argument");
    for (int i = 0; i < batch_num; i++)   // vector_simd.c:1487-1492
        JHCHK(jh_gemm_f32_bf16(b_ids ? b_ids[i] : -1, a, aoffset, b ? b[i] : nullptr, boffset, cr ? cr[i] : nullptr,
                               r ? r[i] : nullptr, roffset, m, n0, n, k, lda, ldb, ldc));
    return JH_OK;
}
int jh_gemm_q8_q4_batch(int batch_num, const int64_t* b_ids, const int64_t* bf_ids, const float* af, const int8_t* a,
                        int aoffset, const float* const* bf, const uint8_t* const* b, int boffset, float* const* r,
                        int roffset, int m, int n0, int n, int k, int lda, int ldaf, int ldb, int ldbf, int ldc) {
    for (int i = 0; i < batch_num; i++)
        JHCHK(jh_gemm_q8_q4(b_ids ? b_ids[i] : -1, bf_ids ? bf_ids[i] : -1, af, a, aoffset, bf ? bf[i] : nullptr,
                            b ? b[i] : nullptr, boffset, r[i], roffset, m, n0, n, k, lda, ldaf, ldb, ldbf, ldc));
    return JH_OK;
}
int jh_gemm_f32_q4_batch(int batch_num, const int64_t* b_ids, const int64_t* bf_ids, const float* a, int aoffset,
                         const float* const* bf, const uint8_t* const* b, int boffset, float* const* r, int roffset,
                         int m, int n0, int n, int k, int lda, int ldb, int ldbf, int ldc) {
    for (int i = 0; i < batch_num; i++)
        JHCHK(jh_gemm_f32_q4(b_ids ? b_ids[i] : -1, bf_ids ? bf_ids[i] : -1, a, aoffset, bf ? bf[i] : nullptr,
                             b ? b[i] : nullptr, boffset, r[i], roffset, m, n0, n, k, lda, ldb, ldbf, ldc));
    return JH_OK;
}

int jh_accumulate_f32(float* a, const float* b, int offset, int length) {
    if (!b) return set_err(JH_ERR_INVALID, "accumulate: b is null");
    return tier1_ew<EW_ACC>(a, b, 0.f, offset, length);
}
int jh_maccumulate_f32(float* a, const float* b, int offset, int length) {
    if (!b) return set_err(JH_ERR_INVALID, "maccumulate: b is null");
    return tier1_ew<EW_MACC>(a, b, 0.f, offset, length);
}
int jh_scale_f32(float factor, float* a, int offset, int length) { return tier1_ew<EW_SCALE>(a, nullptr, factor, offset, length); }
int jh_silu_mul_f32(float* g, const float* u, int n) {
    if (!u) return set_err(JH_ERR_INVALID, "silu_mul: u is null");
    return tier1_ew<EW_SILU_MUL>(g, u, 0.f, 0, n);
}
int jh_saxpy_f32(float alpha, const float* x, float* y, int xoffset, int yoffset, int limit) {
    if (!x || !y || limit < 0) return set_err(JH_ERR_INVALID, "saxpy: bad argument");
    if (limit == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dY = nullptr, *dX = nullptr;
    JHCHK(dev_buf(0, (size_t)limit * 4, &dY));
    JHCHK(dev_buf(1, (size_t)limit * 4, &dX));
    HIPCHK(hipMemcpyAsync(dY, y + yoffset, (size_t)limit * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dX, x + xoffset, (size_t)limit * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL((ew_kernel<EW_SAXPY>), dim3((limit + 255) / 256), dim3(256), 0, st, (float*)dY, (const float*)dX, alpha, limit);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(y + yoffset, dY, (size_t)limit * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_saxpy_batch_f32(const float* alpha, const float* x, int ldx, float* y, int xoffset, int yoffset, int limit,
                       int aoffset, int xrowoffset, int batch_size) {
    if (!alpha || !x || !y || limit < 0 || batch_size < 0) return set_err(JH_ERR_INVALID, "saxpy_batch: bad argument");
    if (limit == 0 || batch_size == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dY = nullptr, *dX = nullptr, *dAl = nullptr;
    const size_t xelems = (size_t)ldx * (batch_size - 1) + limit;
    JHCHK(dev_buf(0, (size_t)limit * 4, &dY));
    JHCHK(dev_buf(1, xelems * 4, &dX));
    JHCHK(dev_buf(2, (size_t)batch_size * 4, &dAl));
    HIPCHK(hipMemcpyAsync(dY, y + yoffset, (size_t)limit * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dX, x + (size_t)xrowoffset * ldx + xoffset, xelems * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dAl, alpha + aoffset, (size_t)batch_size * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(saxpy_batch_kernel, dim3((limit + 127) / 128), dim3(128), 0, st, (const float*)dAl, (const float*)dX, ldx,
                       (float*)dY, limit, batch_size);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(y + yoffset, dY, (size_t)limit * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_accumulate_f32_q4(float* a, const uint8_t* nib_row, const float* scale_row, int offset, int length) {
    if (!a || !nib_row || !scale_row || length < 0) return set_err(JH_ERR_INVALID, "accumulate_q4: bad argument");
    if (length == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    const int end = offset + length;
    void *dA = nullptr, *dN = nullptr, *dS = nullptr;
    JHCHK(dev_buf(0, (size_t)end * 4, &dA));
    JHCHK(dev_buf(1, (size_t)(end + 31) / 32 * 16, &dN));
    JHCHK(dev_buf(2, (size_t)(end + 31) / 32 * 4, &dS));
    HIPCHK(hipMemcpyAsync((float*)dA + offset, a + offset, (size_t)length * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dN, nib_row, (size_t)(end + 31) / 32 * 16, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dS, scale_row, (size_t)(end + 31) / 32 * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(acc_q4_kernel, dim3((length + 255) / 256), dim3(256), 0, st, (float*)dA, (const uint8_t*)dN, (const float*)dS,
                       offset, length);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(a + offset, (float*)dA + offset, (size_t)length * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_quantize_q8(const float* x, int rows, int ldx, int offset, int length, int8_t* q, int ldq, float* d, int ldd) {
    if (!x || !q || !d || rows < 0 || length < 0 || (length % QB) || (offset % QB))
        return set_err(JH_ERR_INVALID, "quantize_q8: length/offset must be multiples of 32");
    if (rows == 0 || length == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dX = nullptr, *dQ = nullptr, *dD = nullptr;
    const size_t xe = (size_t)ldx * (rows - 1) + offset + length;
    const size_t qe = (size_t)ldq * (rows - 1) + offset + length;
    const size_t de = (size_t)ldd * (rows - 1) + (offset + length) / QB;
    JHCHK(dev_buf(0, xe * 4, &dX));
    JHCHK(dev_buf(1, qe, &dQ));
    JHCHK(dev_buf(2, de * 4, &dD));
    HIPCHK(hipMemcpyAsync(dX, x, xe * 4, hipMemcpyHostToDevice, st));
    const long long threads = (long long)rows * (length / QB) * 32;
    hipLaunchKernelGGL(quantize_q8_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const float*)dX, rows, ldx,
                       offset, length, (int8_t*)dQ, ldq, (float*)dD, ldd);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy2DAsync(q + offset, (size_t)ldq, (int8_t*)dQ + offset, (size_t)ldq, (size_t)length, rows, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpy2DAsync(d + offset / QB, (size_t)ldd * 4, (float*)dD + offset / QB, (size_t)ldd * 4, (size_t)(length / QB) * 4, rows,
                            hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_quantize_bf16(const float* x, int64_t n, uint16_t* out) {
    if (!x || !out || n < 0) return set_err(JH_ERR_INVALID, "quantize_bf16: bad argument");
    if (n == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dX = nullptr, *dO = nullptr;
    JHCHK(dev_buf(0, (size_t)n * 4, &dX));
    JHCHK(dev_buf(1, (size_t)n * 2, &dO));
    HIPCHK(hipMemcpyAsync(dX, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(quantize_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)dX, (long long)n, (uint16_t*)dO);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dO, (size_t)n * 2, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_rmsnorm_f32(const float* x, const float* w, float weight_adj, int n, float eps, float* out) {
    if (!x || !w || !out || n <= 0) return set_err(JH_ERR_INVALID, "rmsnorm: bad argument");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dX = nullptr, *dW = nullptr, *dO = nullptr;
    JHCHK(dev_buf(0, (size_t)n * 4, &dX));
    JHCHK(dev_buf(1, (size_t)n * 4, &dW));
    JHCHK(dev_buf(2, (size_t)n * 4, &dO));
    HIPCHK(hipMemcpyAsync(dX, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dW, w, (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(1), dim3(1024), 0, st, (const float*)dX, (const float*)dW, weight_adj, n, eps, (float*)dO);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dO, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
// GPT-2 family pieces (BASELINE config 0): LayerNorm (core/model/LayerNorm.java:41-67) over [offset, offset+length) of each
// of `rows` rows with leading dimension ld; divisor = embeddingLength.  GELU in place.
int jh_layernorm_f32(const float* x, const float* w, const float* b, int rows, int ld, int offset, int length, int divisor,
                     float eps, float* out) {
    if (!x || !w || !b || !out || rows <= 0 || ld <= 0 || offset < 0 || length <= 0 || offset + length > ld || divisor <= 0)
        return set_err(JH_ERR_INVALID, "layernorm: bad argument");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dX = nullptr, *dW = nullptr, *dB = nullptr, *dO = nullptr;
    const size_t nb = (size_t)rows * ld * 4;
    JHCHK(dev_buf(0, nb, &dX));
    JHCHK(dev_buf(1, (size_t)ld * 4, &dW));
    JHCHK(dev_buf(2, (size_t)ld * 4, &dB));
    JHCHK(dev_buf(4, nb, &dO));
    HIPCHK(hipMemcpyAsync(dX, x, nb, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dO, out, nb, hipMemcpyHostToDevice, st));   // columns outside the window keep the caller's values
    HIPCHK(hipMemcpyAsync(dW, w, (size_t)ld * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dB, b, (size_t)ld * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(64), 0, st, (const float*)dX, (const float*)dW, (const float*)dB, ld, offset, length,
                       divisor, eps, (float*)dO);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dO, nb, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_gelu_f32(float* x, int n) {
    if (!x || n <= 0) return set_err(JH_ERR_INVALID, "gelu: bad argument");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void* dX = nullptr;
    JHCHK(dev_buf(0, (size_t)n * 4, &dX));
    HIPCHK(hipMemcpyAsync(dX, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gelu_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (float*)dX, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(x, dX, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_softmax_f32(float* x, int offset, int length) {
    if (!x || length <= 0 || offset < 0) return set_err(JH_ERR_INVALID, "softmax: bad argument");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void* dX = nullptr;
    const size_t n = (size_t)offset + length;
    JHCHK(dev_buf(0, n * 4, &dX));
    HIPCHK(hipMemcpyAsync(dX, x, n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(softmax_kernel, dim3(1), dim3(1024), 0, st, (float*)dX, offset, length);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(x, dX, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}

// VectorMath.precomputeFreqsCis (core/math/VectorMath.java:148-165) -- host side, double cos/sin of the float angle.
int jh_rope_table(int dim, int end, double theta, double scaling, float* out) {
    if (!out || dim <= 0 || (dim & 1) || end <= 0) return set_err(JH_ERR_INVALID, "rope_table: bad argument");
    const int half = dim / 2;
    std::vector<float> freqs((size_t)half);
    float step = 0.0f;
    for (int i = 0; i < half; i++, step = (float)(step + 2.0))
        freqs[(size_t)i] = (float)((1.0 / pow(theta, (double)(step / (float)dim))) / scaling);
    for (int p = 0; p < end; p++) {
        const float t = (float)p;
        for (int i = 0; i < half; i++) {
            const float ang = t * freqs[(size_t)i];
            out[((size_t)p * half + i) * 2 + 0] = (float)cos((double)ang);
            out[((size_t)p * half + i) * 2 + 1] = (float)sin((double)ang);
        }
    }
    return JH_OK;
}
int jh_rope_apply_f32(float* q, float* k, const float* rope, int table_positions, int position, int n_heads, int n_kv_heads,
                      int head_size) {
    if (!q || !k || !rope || position < 0 || n_heads <= 0 || n_kv_heads <= 0 || n_heads % n_kv_heads || (head_size & 1))
        return set_err(JH_ERR_INVALID, "rope_apply: bad argument");
    // kv head h reads table row position + 2*h (CausalSelfAttention.java:260-283): the reference indexes past its table
    // (ArrayIndexOutOfBoundsException) for the last 2*(kvHeads-1) positions
    if ((long long)position + 2LL * (n_kv_heads - 1) >= (long long)table_positions)
        return set_err(JH_ERR_INVALID, "rope_apply: position + 2*(n_kv_heads-1) is beyond the RoPE table");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    const int half = head_size / 2;
    // rows of the table this call touches: [position*half, position*half + n_kv_heads*head_size)
    const size_t r0 = (size_t)position * half, rn = (size_t)n_kv_heads * head_size;
    void *dQ = nullptr, *dK = nullptr, *dR = nullptr;
    JHCHK(dev_buf(0, (size_t)n_heads * head_size * 4, &dQ));
    JHCHK(dev_buf(1, (size_t)n_kv_heads * head_size * 4, &dK));
    JHCHK(dev_buf(2, rn * 8, &dR));
    HIPCHK(hipMemcpyAsync(dQ, q, (size_t)n_heads * head_size * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dK, k, (size_t)n_kv_heads * head_size * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dR, rope + r0 * 2, rn * 8, hipMemcpyHostToDevice, st));
    const int threads = (n_heads + n_kv_heads) * half;
    hipLaunchKernelGGL(rope_kernel, dim3((threads + 255) / 256), dim3(256), 0, st, (float*)dQ, (float*)dK,
                       (const float*)dR - r0 * 2, position, n_heads, n_kv_heads, head_size);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(q, dQ, (size_t)n_heads * head_size * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(k, dK, (size_t)n_kv_heads * head_size * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
// KvBufferCache.computePageSize (core/tensor/KvBufferCache.java:224-280)
int jh_kv_page_geometry(int64_t max_page_bytes, int n_layers, int context_length, int kv_length, int dtype_size,
                        int32_t* out2) {
    if (!out2 || n_layers <= 0 || context_length <= 0 || kv_length <= 0) return set_err(JH_ERR_INVALID, "kv geometry: bad argument");
    const int64_t s = 2LL * dtype_size * kv_length;
    if (max_page_bytes <= s) return set_err(JH_ERR_INVALID, "maxPageSizeInBytes must be greater than the size of a single layer");
    int optL = 1, optC = 1;
    int64_t maxProduct = 0;
    for (int x = n_layers; x >= 1; x--) {
        const int64_t y = max_page_bytes / (x * s);
        if (y >= 1 && y <= context_length) {
            const int64_t product = x * y;
            if (product > maxProduct) { optL = x; optC = (int)y; maxProduct = product; }
            if (product < maxProduct) break;
        }
    }
    out2[0] = optL;
    out2[1] = optC;
    return JH_OK;
}

}  // extern "C"
