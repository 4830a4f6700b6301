// probe.hip -- part of libjlamahip.so (C ABI: include/jlama_hip.h).  Measurement hooks: attention phase timeline, GEMM / per-kernel probes bench.py and tools/ time with HIP events.
#include "jh_host.h"
#include "jh_launch.h"

extern "C" {

// Debug: phase timeline of ONE attention launch at position `pos` (layer 0 of the shard): out[split*16 + k] =
// wall_clock64 ticks (100 MHz) at phase k, -1 where not reached.  See JH_ATT_STAMP in jh_kernels.h.
int jh_debug_attn_timeline(jh_session* s, int pos, long long* out, int n) {
    if (!s || !out || n < 16 * 16) return set_err(JH_ERR_INVALID, "attn_timeline: need 256 slots");
    HIPCHK(hipSetDevice(s->m->device));
    long long* d = nullptr;
    HIPCHK(hipMalloc(&d, 256 * 8));
    HIPCHK(hipMemset(d, 0xff, 256 * 8));
    hipStream_t st = s->stream;
    for (int it = 0; it < 3; it++) {   // warm: the last launch's stamps are the ones reported
        hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, s->st, pos, 0, 0);
        s->attn_variant = attn_variant_for(s, pos);
        JHCHK(attn_launch(s, 0, st, false, d));
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(out, d, 256 * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d));
    return JH_OK;
}
static thread_local long long* g_gemv_dbg = nullptr;   // jh_debug_gemv_timeline: stamp buffer of the LAST launch of THIS thread's next jh_kernel_bench
// Phase stamps of one reference-order few-row GEMV launch (which: 0 q|k|v, 2 o, 4 down; the last layer's launch of a sweep over
// all layers, so its weights come from HBM): out[workgroup][wave][8] wall_clock64 ticks (100 MHz), -1 = not written.
int jh_debug_gemv_timeline(jh_session* s, int which, long long* out, int n) {
    if (!s || !out || !s->strict) return set_err(JH_ERR_INVALID, "gemv_timeline: a reference-order session");
    // the kernels stamp [workgroup][wave][8] without a bound: one 512-thread workgroup per CU at most (p16_plan / launch_gemv_t16)
    const int need = g_cu_count * (P16_THREADS / 64) * 8;
    if (n < need) return set_err(JH_ERR_INVALID, "gemv_timeline: needs " + std::to_string(need) + " slots (workgroups x 8 waves x 8 stamps)");
    if (which != 0 && which != 2 && which != 3 && which != 4) return set_err(JH_ERR_INVALID, "gemv_timeline: which = 0 (q|k|v), 2 (o), 3 (gate|up) or 4 (down)");
    HIPCHK(hipSetDevice(s->m->device));
    long long* d = nullptr;
    HIPCHK(hipMalloc(&d, (size_t)n * 8));
    HIPCHK(hipMemset(d, 0xff, (size_t)n * 8));
    g_gemv_dbg = d;
    double ms = 0; int64_t b = 0;
    const int rc = jh_kernel_bench(s, which, 2, &ms, &b);
    g_gemv_dbg = nullptr;
    if (rc == JH_OK) HIPCHK(hipMemcpy(out, d, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d));
    return rc;
}
// Device-resident timing of the batched (prefill) MFMA GEMMs: kind 0 = I8xQ4, 1 = BF16xBF16.  `copies` distinct weight
// matrices are cycled so the stream comes from HBM, not the Infinity Cache.  out_ms = average per GEMM.
int jh_gemm_bench(int kind, int m, int n, int k, int copies, int iters, double* out_ms) {
    if (!out_ms || m < 2 || m > 256 || (n % 32) || (k % 64) || copies < 1 || iters < 1) return set_err(JH_ERR_INVALID, "gemm_bench: bad shape");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    const bool q4 = (kind == 0 || kind == 2);   // kind 2 = I8xQ4 with both operands in MFMA-tiled order (prefill path)
    const size_t wbytes = q4 ? (size_t)n * k / 2 : (size_t)n * k * 2;
    const size_t sbytes = q4 ? (size_t)n * (k / QB) * 4 : 0;
    uint8_t *w = nullptr, *a = nullptr; float *ws = nullptr, *af = nullptr, *c = nullptr;
    HIPCHK(hipMalloc(&w, wbytes * copies)); HIPCHK(hipMemset(w, 0x37, wbytes * copies));
    if (sbytes) { HIPCHK(hipMalloc(&ws, sbytes * copies)); HIPCHK(hipMemset(ws, 0, sbytes * copies)); }
    HIPCHK(hipMalloc(&a, (size_t)(m + 32) * k * 2)); HIPCHK(hipMemset(a, 1, (size_t)(m + 32) * k * 2));
    HIPCHK(hipMalloc(&af, (size_t)(m + 32) * (k / QB) * 4)); HIPCHK(hipMemset(af, 0, (size_t)(m + 32) * (k / QB) * 4));
    HIPCHK(hipMalloc(&c, (size_t)m * n * 4));
    float* bf16_ws = nullptr;
    HIPCHK(hipMalloc(&bf16_ws, BF16_SPLITK_WS_BYTES));   // split-K workspace; kind 3 = BF16 with MFMA-ordered operands
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    int rc = JH_OK;
    for (int it = -1; it < iters && rc == JH_OK; it++) {
        if (it == 0) HIPCHK(hipEventRecord(e0, st));
        for (int l = 0; l < copies && rc == JH_OK; l++) {
            if (q4) {
                MfmaQ4Params g{(const int8_t*)a, af, w + l * wbytes, ws + l * (sbytes / 4), c, nullptr, m, 0, n, k, k, k / QB, k / 2, k / QB, n, 0};
                rc = launch_gemm_q8q4_mfma(g, st, kind == 2, bf16_ws, BF16_SPLITK_WS_BYTES);
            } else if (kind == 3) {
                MfmaBf16TileParams g{(const uint16_t*)a, (const uint16_t*)(w + l * wbytes), c, nullptr, m, n, k, n, bf16_ws, 1};
                rc = launch_gemm_bf16_tile(g, st);
            } else {
                MfmaGemmParams g{(const uint16_t*)a, (const uint16_t*)(w + l * wbytes), c, m, 0, n, k, k, k, n, 0, nullptr, bf16_ws, 1};
                rc = launch_gemm_bf16_mfma(g, st);
            }
        }
    }
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *out_ms = (double)ms / ((double)iters * copies);
    hipFree(w); if (ws) hipFree(ws); if (bf16_ws) hipFree(bf16_ws); hipFree(a); hipFree(af); hipFree(c); hipEventDestroy(e0); hipEventDestroy(e1);
    return rc;
}
int jh_session_synchronize(jh_session* s) {
    if (!s) return set_err(JH_ERR_INVALID, "session_synchronize: null");
    HIPCHK(hipSetDevice(s->m->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    return JH_OK;
}

// Roofline probe: launch ONE kernel kind of the decode step back-to-back over all of this shard's layers (so the
// weights stream from HBM, not from the 256 MiB Infinity Cache), `iters` sweeps, bracketed by hipEvents on the
// session's stream.  out_ms = average duration of one launch.
int jh_kernel_bench(jh_session* s, int which, int iters, double* out_ms, int64_t* out_bytes_per_launch) {
    if (!s || !out_ms || iters <= 0) return set_err(JH_ERR_INVALID, "kernel_bench: bad argument");
    if (s->m->c.weight_dtype != JH_DT_Q4) return set_err(JH_ERR_UNSUPPORTED, "kernel_bench: JQ4 models only");
    jh_model* m = s->m;
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = s->stream;
    const jh_config& c = m->c;
    const int E = c.embedding_length, H = c.hidden_length, hs = c.head_size;
    const int A = c.n_heads * hs, KV = c.n_kv_heads * hs;
    const int nl = c.layer_end - c.layer_start;
    JHCHK(ensure_strict_operands(s, st));
    JHCHK(refuse_order_free(s, "kernel_bench"));
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, s->st, s->max_ctx / 2, 0, 0);
    s->attn_variant = attn_variant_for(s, s->max_ctx / 2);
    const bool p16 = s->strict != 0;   // reference-order kernels (jh_t16.h / jh_p16.h) when the session is in that mode
    int launches = 0;
    for (int it = -1; it < iters; it++) {
        if (it == 0) HIPCHK(hipEventRecord(s->ev0, st));
        for (int li = c.layer_start; li < c.layer_end; li++) {
            const JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
            GemvParams p;
            memset(&p, 0, sizeof(p));
            if (it == iters - 1 && li == c.layer_end - 1) p.dbg = g_gemv_dbg;
            if (which == 9) {          // LM head (+ final norm, argmax partials): one weight, re-streamed per launch
                JHCHK(lmhead_launch(s, st));
            } else if (which == 0) {
                const JWeight& F = m->qkv[(size_t)li];
                p.w = (const uint8_t*)F.data; p.ws = F.scales; p.nrows = A + 2 * KV; p.out = s->qkv;
                p.K = E; p.ldb = E / 2; p.ldbf = E / QB;
                p.x = s->x; p.nw = (const float*)W[JH_W_NORM1].data; p.eps = c.rms_eps;
                if (p16) { JHCHK(use_p16t(p, F)); JHCHK((launch_gemv_i8q4_p16<PRO_RMS_Q8, EPI_STORE>(p, s->p16_depth, st))); }
                else JHCHK((launch_gemv_i8q4<PRO_RMS_Q8, EPI_STORE>(p, s->cfg_qkv, st)));
            } else if (which == 1) {
                JHCHK(attn_launch(s, li - c.layer_start, st, false));
            } else if (which == 2) {
                p.w = (const uint8_t*)W[JH_W_O].data; p.ws = W[JH_W_O].scales; p.nrows = E; p.out = s->x1;
                p.K = A; p.ldb = A / 2; p.ldbf = A / QB; p.x = s->attf; p.resid = s->x;
                if (p16) {
                    JHCHK(use_p16t(p, W[JH_W_O]));
                    JHCHK((launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_RESID>(p, s->p16_depth, st)));
                } else {
                    JHCHK((launch_gemv_i8q4<PRO_QUANT_Q8, EPI_RESID>(p, s->cfg_o, st)));
                }
            } else if (which == 3) {
                p.w = (const uint8_t*)W[JH_W_GATE].data; p.ws = W[JH_W_GATE].scales; p.nrows = H;
                p.w2 = (const uint8_t*)W[JH_W_UP].data; p.ws2 = W[JH_W_UP].scales;
                p.K = E; p.ldb = E / 2; p.ldbf = E / QB;
                p.x = s->x1; p.nw = (const float*)W[JH_W_NORM2].data; p.eps = c.rms_eps;
                p.out = s->hf;
                if (t16_gateup_ok(m, li) && (p16 || fast_gateup_t16(m))) {
                    JHCHK(ensure_gateup_t16(m, li, st));
                    p.w = m->gateup[(size_t)li].t16; p.ws = m->gateup[(size_t)li].t16_scales; p.w2 = nullptr; p.ws2 = nullptr;
                    JHCHK((launch_gemv_t16<PRO_RMS_Q8, EPI_SILU_MUL>(p, st)));
                }
                else if (p16) { JHCHK(use_p16t(p, W[JH_W_GATE])); p.w2 = W[JH_W_UP].p16t; JHCHK((launch_gemv_i8q4_p16<PRO_RMS_Q8, EPI_SILU_MUL>(p, s->p16_depth, st))); }
                else JHCHK((launch_gemv_i8q4<PRO_RMS_Q8, EPI_SILU_MUL>(p, s->cfg_gateup, st)));
            } else if (which == 4) {
                p.w = (const uint8_t*)W[JH_W_DOWN].data; p.ws = W[JH_W_DOWN].scales; p.nrows = E; p.out = s->x1;
                p.K = H; p.ldb = H / 2; p.ldbf = H / QB; p.x = s->hf; p.resid = s->x;
                if (p16) { JHCHK(use_p16t(p, W[JH_W_DOWN])); JHCHK((launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_RESID>(p, s->p16_depth, st))); }
                else JHCHK((launch_gemv_i8q4<PRO_QUANT_Q8, EPI_RESID>(p, s->cfg_down, st)));
            } else if (which >= 5 && which <= 8) {
                // the same GEMVs fed a pre-quantized activation row (PRO_Q8): what the fused prologue costs
                JHCHK(prefill_alloc(s));
                p.aq = s->pb_aq; p.ad = s->pb_ad;
                if (which == 5) {
                    const JWeight& F = m->qkv[(size_t)li];
                    p.w = (const uint8_t*)F.data; p.ws = F.scales; p.nrows = A + 2 * KV; p.out = s->qkv;
                    p.K = E; p.ldb = E / 2; p.ldbf = E / QB;
                    JHCHK((launch_gemv_i8q4<PRO_Q8, EPI_STORE>(p, s->cfg_qkv, st)));
                } else if (which == 6) {
                    p.w = (const uint8_t*)W[JH_W_O].data; p.ws = W[JH_W_O].scales; p.nrows = E; p.out = s->x1;
                    p.K = A; p.ldb = A / 2; p.ldbf = A / QB; p.resid = s->x;
                    JHCHK((launch_gemv_i8q4<PRO_Q8, EPI_RESID>(p, s->cfg_o, st)));
                } else if (which == 7) {
                    p.w = (const uint8_t*)W[JH_W_GATE].data; p.ws = W[JH_W_GATE].scales; p.nrows = H;
                    p.w2 = (const uint8_t*)W[JH_W_UP].data; p.ws2 = W[JH_W_UP].scales;
                    p.K = E; p.ldb = E / 2; p.ldbf = E / QB; p.out = s->hf;
                    JHCHK((launch_gemv_i8q4<PRO_Q8, EPI_SILU_MUL>(p, s->cfg_gateup, st)));
                } else {
                    p.w = (const uint8_t*)W[JH_W_DOWN].data; p.ws = W[JH_W_DOWN].scales; p.nrows = E; p.out = s->x1;
                    p.K = H; p.ldb = H / 2; p.ldbf = H / QB; p.resid = s->x;
                    JHCHK((launch_gemv_i8q4<PRO_Q8, EPI_RESID>(p, s->cfg_down, st)));
                }
            } else {
                return set_err(JH_ERR_INVALID, "kernel_bench: which in 0..9 (qkv, attn, oproj, gateup, down; 5..8 = the GEMVs with pre-quantized input; 9 = LM head)");
            }
            if (it >= 0) launches++;
        }
    }
    HIPCHK(hipEventRecord(s->ev1, st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    *out_ms = (double)ms / launches;
    if (out_bytes_per_launch) {
        const double bpw = 0.625;  // 0.5 B nibble + 4 B scale / 32 weights
        int64_t b = 0;
        if (which == 0) b = (int64_t)((double)(A + 2 * KV) * E * bpw);
        else if (which == 1) b = (int64_t)2 * (s->max_ctx / 2 + 1) * KV * 4 + (int64_t)2 * KV * 4;
        else if (which == 2) b = (int64_t)((double)E * A * bpw);
        else if (which == 3 || which == 7) b = (int64_t)((double)2 * H * E * bpw);
        else if (which == 9) b = (int64_t)((double)c.vocab_size * E * bpw);
        else if (which == 5) b = (int64_t)((double)(A + 2 * KV) * E * bpw);
        else if (which == 6) b = (int64_t)((double)E * A * bpw);
        else b = (int64_t)((double)E * H * bpw);
        *out_bytes_per_launch = b;
    }
    (void)nl;
    return JH_OK;
}


}  // extern "C"
