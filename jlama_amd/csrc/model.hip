// model.hip -- part of libjlamahip.so (C ABI: include/jlama_hip.h).  Models and sessions: weights in HBM, the operand copies of the reference-order kernels, KV pages, session state.
#include "jh_host.h"
#include "jh_launch.h"

bool is_global_slot(int which) { return which == JH_W_EMBED || which == JH_W_LMHEAD || which == JH_W_FINALNORM; }

thread_local int g_operand_packs = 0;   // operand copies this thread has queued a pack kernel for (ensure_strict_operands waits for them)
// the order-free sessions use the MFMA gate|up GEMV as well (option JH_FAST_GATEUP_T16=0: their own VALU kernel, for comparisons)
bool fast_gateup_t16(jh_model* m) { return opt_int("JH_FAST_GATEUP_T16", 1) != 0 && tiled_mode_for(m) == TILED_RESIDENT; }   // (a second copy: not under JH_TILED_COPY=transient)
// gate|up of layer li in T16 order (tile u = gate rows 8u..8u+7, up rows 8u..8u+7): made once, before any graph capture
bool t16_gateup_ok(const jh_model* m, int li) {
    const int enabled = opt_int("JH_T16", 1);
    const JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    const JWeight &G = W[JH_W_GATE], &U = W[JH_W_UP];
    return enabled && w_present(G) && w_present(U) && G.dtype == JH_DT_Q4 && U.dtype == JH_DT_Q4 && G.rows == U.rows && G.cols == U.cols &&
           G.rows % 8 == 0 && t16_shape_ok(G.cols);
}
int ensure_gateup_t16(jh_model* m, int li, hipStream_t st) {
    JWeight& F = m->gateup[(size_t)li];
    if (F.t16 || !t16_gateup_ok(m, li)) return JH_OK;
    if (m->strict_only) return set_err(JH_ERR_INVALID, "gate|up T16 copy: the row-major weights of this model were released (JH_STRICT_ONLY)");
    const JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    const JWeight &G = W[JH_W_GATE], &U = W[JH_W_UP];
    const int rows = 2 * G.rows, K = G.cols, nblk = K / QB, ntiles = rows / 16;
    F.dtype = G.dtype; F.rows = rows; F.cols = K;
    hipError_t e = hipMalloc((void**)&F.t16, t16_w_bytes(rows, K));
    if (e == hipSuccess) e = hipMalloc((void**)&F.t16_scales, t16_s_bytes(rows, K));
    if (e != hipSuccess) {   // never leave half a copy behind: later calls would take it for a finished one
        if (F.t16) hipFree(F.t16);
        F.t16 = nullptr; F.t16_scales = nullptr;
        return set_err(JH_ERR_OOM, "hipMalloc T16 gate|up copy");
    }
    g_operand_packs++;
    const long long threads = (long long)ntiles * (nblk / 4) * 16;
    hipLaunchKernelGGL(t16_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const i32x4*)G.data, (const float*)G.scales,
                       (const i32x4*)U.data, (const float*)U.scales, nblk, ntiles, 1, (i32x4*)F.t16, (f32x4t*)F.t16_scales);
    HIPCHK(hipGetLastError());
    return JH_OK;
}
// T16 copy (mode 0: tile u = rows 16u..16u+15) of a Q4 weight: the operand of the reference-order prompt GEMM on the MFMA
bool t16_weight_ok(const JWeight& W) { return w_present(W) && W.dtype == JH_DT_Q4 && W.rows % 16 == 0 && W.cols % 512 == 0; }
int ensure_t16(JWeight& W, hipStream_t st) {
    if (W.t16 || !t16_weight_ok(W) || !W.data) return JH_OK;
    const int nblk = W.cols / QB, ntiles = W.rows / 16;
    hipError_t e = hipMalloc((void**)&W.t16, t16_w_bytes(W.rows, W.cols));
    if (e == hipSuccess) e = hipMalloc((void**)&W.t16_scales, t16_s_bytes(W.rows, W.cols));
    if (e != hipSuccess) {
        if (W.t16) hipFree(W.t16);
        W.t16 = nullptr; W.t16_scales = nullptr;
        return set_err(JH_ERR_OOM, "hipMalloc T16 weight copy");
    }
    g_operand_packs++;
    const long long threads = (long long)ntiles * (nblk / 4) * 16;
    hipLaunchKernelGGL(t16_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const i32x4*)W.data, (const float*)W.scales,
                       (const i32x4*)nullptr, (const float*)nullptr, nblk, ntiles, 0, (i32x4*)W.t16, (f32x4t*)W.t16_scales);
    HIPCHK(hipGetLastError());
    return JH_OK;
}
// P16T copy of a Q4 weight (jh_p16.h: byte t of the 16 blocks of a group in one 16-byte chunk), made once
int ensure_p16t(JWeight& W, hipStream_t st) {
    if (W.p16t || !W.data) return JH_OK;
    if (W.dtype == JH_DT_BF16) {   // BF16T order (jh_bf16r.h): the 16-byte chunk t of a 128-element group = the next 8 links of chain t
        const size_t rb = bf16t_row_bytes(W.cols);
        if (hipMalloc((void**)&W.p16t, (size_t)((W.rows + 3) & ~3) * rb + 64) != hipSuccess) { W.p16t = nullptr; return set_err(JH_ERR_OOM, "hipMalloc BF16T weight copy"); }
        g_operand_packs++;
        const long long threads = (long long)W.rows * (long long)(rb / 16);
        hipLaunchKernelGGL(bf16t_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const uint16_t*)W.data, W.rows, W.cols, W.cols, W.p16t);
        HIPCHK(hipGetLastError());
        return JH_OK;
    }
    if (W.dtype != JH_DT_Q4) return JH_OK;
    const int nblk = W.cols / QB;
    const size_t rb = p16t_row_bytes(W.cols);
    if (hipMalloc((void**)&W.p16t, (size_t)((W.rows + 3) & ~3) * rb + 64) != hipSuccess) { W.p16t = nullptr; return set_err(JH_ERR_OOM, "hipMalloc P16T weight copy"); }
    g_operand_packs++;
    const long long threads = (long long)W.rows * (long long)(rb / 16);
    hipLaunchKernelGGL(p16t_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const uint8_t*)W.data, W.rows, nblk, W.cols / 2, W.p16t);
    HIPCHK(hipGetLastError());
    return JH_OK;
}
// the reference-order kernels' view of a weight: P16T nibbles + the checkpoint's scales
int use_p16t(GemvParams& p, const JWeight& W) {
    if (!W.p16t) return set_err(JH_ERR_INVALID, "reference-order GEMV: the weight has no P16T / BF16T copy (ensure_strict_operands)");
    p.w = W.p16t; p.ldb = (int)(W.dtype == JH_DT_BF16 ? bf16t_row_bytes(W.cols) : p16t_row_bytes(W.cols));
    return JH_OK;
}
// every operand copy a reference-order session of this shard will touch (allocation must not happen inside a graph capture)
static int ensure_strict_operands_locked(jh_session* s, hipStream_t st);
// The copies belong to the MODEL: every session of it (other streams, other host threads) reads them.  They are created under the
// model's lock, and the lock is released only once the pack kernels this call queued have FINISHED -- a second session then either
// waits here or finds complete copies; nobody launches a reference-order GEMV against a half-packed operand.
// JH_STRICT_ONLY=1: a model that serves reference-order sessions only keeps ONE copy of the projection nibbles per kernel family instead
// of those plus the checkpoint's row-major ones (70B in reference order held 2.56 x 43 GB).  Once the T16 / P16T
// copies of a projection are published, its row-major nibbles (and the order-free prompt GEMM's MFMA-ordered copy) are released; the
// block scales stay (the P16T kernels read them row-major), embedding / LM head / norms stay.  From then on the model refuses
// order-free sessions and weight replacement, loudly.
static void release_row_major(jh_model* m, JWeight& W, bool owns) {
    if (W.dropped || !W.data || W.dtype != JH_DT_Q4) return;
    if (owns) {
        m->dropped_bytes += (int64_t)W.rows * W.cols / 2;
        hipFree(W.data);
        if (W.tiled) { m->dropped_bytes += (int64_t)(tiled_w_bytes(W) + (W.tiled_scales ? tiled_s_bytes(W) : 0)); hipFree(W.tiled); if (W.tiled_scales) hipFree(W.tiled_scales); }
        W.tiled = nullptr; W.tiled_scales = nullptr;
    }
    W.data = nullptr; W.dropped = true;
}
static int drop_row_major_weights(jh_session* s) {
    jh_model* m = s->m;
    const bool gemm = prefill_t16_ok(s);
    for (int li = m->c.layer_start; li < m->c.layer_end; li++) {
        JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
        JWeight &F = m->qkv[(size_t)li], &GU = m->gateup[(size_t)li];
        auto covered = [&](const JWeight& w) { return w.p16t && (w.t16 || !gemm); };   // every reference-order kernel has its operand
        if (covered(F)) {
            release_row_major(m, F, true);
            release_row_major(m, W[JH_W_Q], false); release_row_major(m, W[JH_W_K], false); release_row_major(m, W[JH_W_V], false);   // slices of F
        }
        if (covered(W[JH_W_O])) release_row_major(m, W[JH_W_O], true);
        if (covered(W[JH_W_DOWN])) release_row_major(m, W[JH_W_DOWN], true);
        if (GU.t16) {   // decode GEMV and prompt GEMM both read the stacked T16 copy
            release_row_major(m, W[JH_W_GATE], true); release_row_major(m, W[JH_W_UP], true);
            if (GU.tiled) { m->dropped_bytes += (int64_t)(tiled_w_bytes(GU) + (GU.tiled_scales ? tiled_s_bytes(GU) : 0)); hipFree(GU.tiled); if (GU.tiled_scales) hipFree(GU.tiled_scales); GU.tiled = nullptr; GU.tiled_scales = nullptr; }
        }
    }
    m->strict_only = true;
    m->weights_version++;   // (no graph may keep a row-major pointer: order-free graphs of other sessions are dropped at their next use -- which now fails)
    return JH_OK;
}
int ensure_strict_operands(jh_session* s, hipStream_t st) {
    std::lock_guard<std::mutex> lk(s->m->op_mu);
    const int before = g_operand_packs;
    const int rc = ensure_strict_operands_locked(s, st);
    if (g_operand_packs != before) HIPCHK(hipStreamSynchronize(st));
    if (rc == JH_OK && s->strict && s->m->c.weight_dtype == JH_DT_Q4 && !s->m->strict_only && opt_int("JH_STRICT_ONLY", 0)) {
        HIPCHK(hipDeviceSynchronize());   // nobody is reading the row-major copies (another session's order-free decode would be the caller's bug)
        return drop_row_major_weights(s);
    }
    return rc;
}
int refuse_order_free(const jh_session* s, const char* what) {
    if (s->m->strict_only && !s->strict)
        return set_err(JH_ERR_UNSUPPORTED, std::string(what) + ": this model keeps reference-order operands only (JH_STRICT_ONLY released the row-major weights)");
    return JH_OK;
}
static int ensure_strict_operands_locked(jh_session* s, hipStream_t st) {
    jh_model* m = s->m;
    if (m->c.weight_dtype == JH_DT_BF16) {
        if (!s->strict) return JH_OK;
        for (int li = m->c.layer_start; li < m->c.layer_end; li++) {   // BF16T copies of every projection (jh_bf16r.h)
            JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
            JHCHK(ensure_p16t(m->qkv[(size_t)li], st));
            JHCHK(ensure_p16t(W[JH_W_O], st));
            JHCHK(ensure_p16t(W[JH_W_GATE], st));
            JHCHK(ensure_p16t(W[JH_W_UP], st));
            JHCHK(ensure_p16t(W[JH_W_DOWN], st));
        }
        JWeight* lmw = m->global_w[JH_W_LMHEAD].data ? &m->global_w[JH_W_LMHEAD] : &m->global_w[JH_W_EMBED];
        if (lmw->data && m->global_w[JH_W_FINALNORM].data) JHCHK(ensure_p16t(*lmw, st));
        return JH_OK;
    }
    if (m->c.weight_dtype != JH_DT_Q4) return JH_OK;
    if (!s->strict) {
        // the order-free sessions take the gate|up GEMV from jh_t16.h too (it is the faster kernel -- and bit-exact): its T16 copy only
        if (fast_gateup_t16(m))
            for (int li = m->c.layer_start; li < m->c.layer_end; li++)
                if (t16_gateup_ok(m, li)) JHCHK(ensure_gateup_t16(m, li, st));
        return JH_OK;
    }
    for (int li = m->c.layer_start; li < m->c.layer_end; li++) {
        JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
        JHCHK(ensure_p16t(m->qkv[(size_t)li], st));
        JHCHK(ensure_p16t(W[JH_W_O], st));
        JHCHK(ensure_p16t(W[JH_W_DOWN], st));
        if (t16_gateup_ok(m, li)) JHCHK(ensure_gateup_t16(m, li, st));
        if (prefill_t16_ok(s)) {   // prompt rows through gemm_t16_kernel: every projection in T16 order
            JHCHK(ensure_t16(m->qkv[(size_t)li], st));
            JHCHK(ensure_t16(W[JH_W_O], st));
            JHCHK(ensure_t16(W[JH_W_DOWN], st));
        }
        if (!t16_gateup_ok(m, li)) {   // the p16 form of the gate|up decode GEMV reads P16T order
            JHCHK(ensure_p16t(W[JH_W_GATE], st));
            JHCHK(ensure_p16t(W[JH_W_UP], st));
        }
    }
    JWeight* lm = m->global_w[JH_W_LMHEAD].data ? &m->global_w[JH_W_LMHEAD] : &m->global_w[JH_W_EMBED];   // (lm_head_weight)
    if (lm->data && m->global_w[JH_W_FINALNORM].data) JHCHK(ensure_p16t(*lm, st));
    return JH_OK;
}

extern "C" {

int jh_model_create(const jh_config* cfg, jh_model** out) {
    if (!cfg || !out) return set_err(JH_ERR_INVALID, "model_create: null");
    if (cfg->weight_dtype != JH_DT_Q4 && cfg->weight_dtype != JH_DT_BF16)
        return set_err(JH_ERR_UNSUPPORTED, "model_create: resident models are JQ4 (Q4 weights, I8 activations) or BF16 (BF16 weights and activations)");
    if (cfg->embedding_length % 256 || cfg->hidden_length % 32 || cfg->n_heads % cfg->n_kv_heads ||
        (cfg->head_size != 64 && cfg->head_size != 128) || cfg->layer_start < 0 || cfg->layer_end > cfg->n_layers ||
        cfg->layer_start >= cfg->layer_end)
        return set_err(JH_ERR_INVALID, "model_create: unsupported shape (E%256, H%32, head_size in {64,128})");
    JHCHK(ensure_ctx());
    jh_model* m = new jh_model();
    m->c = *cfg;
    m->device = tctx.device;
    m->layer_w.resize((size_t)cfg->n_layers * JH_W_COUNT);
    m->qkv.resize((size_t)cfg->n_layers);
    m->gateup.resize((size_t)cfg->n_layers);
    // Config ctor (core/safetensors/Config.java:270-274): table over the whole context
    // (+ ROPE_MARGIN rows: kv head h reads row position + 2*h, CausalSelfAttention.java:260-283; positions whose rows
    // would leave the reference's table are refused by check_positions(), the margin only keeps a stray read in bounds)
    const int half = cfg->head_size / 2;
    std::vector<float> table((size_t)(cfg->context_length + ROPE_MARGIN) * half * 2);
    jh_rope_table(cfg->head_size, cfg->context_length + ROPE_MARGIN, (double)cfg->rope_theta, (double)cfg->rope_scaling, table.data());
    hipError_t e = hipMalloc(&m->rope, table.size() * 4);
    if (e != hipSuccess) { delete m; return set_err(JH_ERR_OOM, "hipMalloc rope table"); }
    HIPCHK(hipMemcpy(m->rope, table.data(), table.size() * 4, hipMemcpyHostToDevice));
    m->attention_scale = (float)(1.0 / sqrt((double)cfg->head_size));  // CausalSelfAttention.java:134
    *out = m;
    return JH_OK;
}
int jh_model_destroy(jh_model* m) {
    if (!m) return JH_OK;
    hipSetDevice(m->device);
    for (size_t i = 0; i < m->layer_w.size(); i++) {
        const int which = (int)(i % JH_W_COUNT);
        if (which == JH_W_Q || which == JH_W_K || which == JH_W_V) continue;  // slices of the fused allocation
        auto& w = m->layer_w[i];
        if (w.data) hipFree(w.data);
        if (w.scales) hipFree(w.scales);
        if (w.tiled) hipFree(w.tiled);
        if (w.tiled_scales) hipFree(w.tiled_scales);
        if (w.p16t) hipFree(w.p16t);
        if (w.t16) hipFree(w.t16);
        if (w.t16_scales) hipFree(w.t16_scales);
    }
    for (auto& w : m->qkv) { if (w.data) hipFree(w.data); if (w.scales) hipFree(w.scales); if (w.tiled) hipFree(w.tiled); if (w.tiled_scales) hipFree(w.tiled_scales); if (w.p16t) hipFree(w.p16t); if (w.t16) hipFree(w.t16); if (w.t16_scales) hipFree(w.t16_scales); }
    for (auto& w : m->gateup) { if (w.tiled) hipFree(w.tiled); if (w.tiled_scales) hipFree(w.tiled_scales); if (w.t16) hipFree(w.t16); if (w.t16_scales) hipFree(w.t16_scales); }
    for (auto& w : m->global_w) { if (w.data) hipFree(w.data); if (w.scales) hipFree(w.scales); if (w.p16t) hipFree(w.p16t); }
    if (m->rope) hipFree(m->rope);
    delete m;
    return JH_OK;
}
int jh_model_set_weight(jh_model* m, int layer, int which, int dtype, const void* data, const float* scales, int rows,
                        int cols, int from_device) {
    if (!m || !data || which < 0 || which >= JH_W_COUNT) return set_err(JH_ERR_INVALID, "set_weight: bad argument");
    if (m->strict_only) return set_err(JH_ERR_UNSUPPORTED, "set_weight: this model released its row-major weights (JH_STRICT_ONLY); its weights are immutable");
    HIPCHK(hipSetDevice(m->device));
    JWeight* w;
    if (layer < 0) {
        if (!is_global_slot(which)) return set_err(JH_ERR_INVALID, "set_weight: slot needs a layer index");
        w = &m->global_w[which];
    } else {
        if (layer >= m->c.n_layers || is_global_slot(which)) return set_err(JH_ERR_INVALID, "set_weight: bad layer/slot");
        w = &m->layer_w[(size_t)layer * JH_W_COUNT + which];
    }
    size_t bytes, sbytes = 0;
    if (dtype == JH_DT_Q4) {
        if (!scales || cols % QB) return set_err(JH_ERR_INVALID, "set_weight: Q4 needs scales and cols%32==0");
        bytes = (size_t)rows * cols / 2;
        sbytes = (size_t)rows * (cols / QB) * 4;
    } else if (dtype == JH_DT_BF16) bytes = (size_t)rows * cols * 2;
    else if (dtype == JH_DT_F32) bytes = (size_t)rows * cols * 4;
    else return set_err(JH_ERR_UNSUPPORTED, "set_weight: dtype");
    const bool is_norm = (which == JH_W_NORM1 || which == JH_W_NORM2 || which == JH_W_FINALNORM);
    if (!is_norm && dtype != m->c.weight_dtype)
        return set_err(JH_ERR_UNSUPPORTED, "set_weight: matmul weights must have the model's weight_dtype (Q4 or BF16)");
    std::vector<float> widened;
    void* widened_dev = nullptr;
    if (is_norm && dtype == JH_DT_BF16) {
        // 1-D norm weights (BF16 on disk, never quantized: AbstractTensor.java:284) are widened to F32 once; exact.
        const size_t n = (size_t)rows * cols;
        if (from_device) {
            HIPCHK(hipMalloc(&widened_dev, n * 4));
            hipLaunchKernelGGL(widen_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const uint16_t*)data, (long long)n,
                               (float*)widened_dev);
            HIPCHK(hipGetLastError());
            HIPCHK(hipDeviceSynchronize());
            data = widened_dev;
        } else {
            widened.resize(n);
            const uint16_t* h = (const uint16_t*)data;
            for (size_t i = 0; i < n; i++) {
                const uint32_t u = ((uint32_t)h[i]) << 16;
                memcpy(&widened[i], &u, 4);
            }
            data = widened.data();
        }
        dtype = JH_DT_F32;
        bytes = n * 4;
    }
    const hipMemcpyKind kind = from_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    // the operand copies (T16 / P16T / BF16T) this call frees are shared by the model's sessions: replaced under the same lock that
    // ensure_strict_operands / the pack kernels hold.  Replacing a weight while ANOTHER session is decoding with it stays
    // the caller's bug, as in the reference (weights are immutable after load).
    std::lock_guard<std::mutex> op_lock(m->op_mu);
    if (m->strict_only) {   // (again, under the lock: the release happens under it)
        if (widened_dev) hipFree(widened_dev);
        return set_err(JH_ERR_UNSUPPORTED, "set_weight: this model released its row-major weights (JH_STRICT_ONLY); its weights are immutable");
    }
    if (layer >= 0 && (which == JH_W_Q || which == JH_W_K || which == JH_W_V)) {
        // q|k|v live stacked in one [A+2KV, E] allocation (CausalSelfAttention.java:161-171 issues three GEMVs over the
        // same activation; here they become one)
        const int A = m->c.n_heads * m->c.head_size, KV = m->c.n_kv_heads * m->c.head_size, E = m->c.embedding_length;
        const int want_rows = which == JH_W_Q ? A : KV;
        if (rows != want_rows || cols != E) return set_err(JH_ERR_INVALID, "set_weight: q/k/v shape");
        JWeight& f = m->qkv[(size_t)layer];
        const size_t row_bytes = dtype == JH_DT_Q4 ? (size_t)E / 2 : (size_t)E * 2;
        if (!f.data) {
            const size_t tot = (size_t)(A + 2 * KV);
            hipError_t e2 = hipMalloc(&f.data, tot * row_bytes + 64);
            if (e2 != hipSuccess) return set_err(JH_ERR_OOM, std::string("hipMalloc qkv: ") + hipGetErrorString(e2));
            if (dtype == JH_DT_Q4) {
                e2 = hipMalloc((void**)&f.scales, tot * (E / QB) * 4 + 64);
                if (e2 != hipSuccess) return set_err(JH_ERR_OOM, std::string("hipMalloc qkv scales: ") + hipGetErrorString(e2));
            }
            f.dtype = dtype; f.rows = (int)tot; f.cols = E;
        }
        if (f.tiled) { hipFree(f.tiled); hipFree(f.tiled_scales); f.tiled = nullptr; f.tiled_scales = nullptr; }
        if (f.p16t) { hipFree(f.p16t); f.p16t = nullptr; }
        if (f.t16) { hipFree(f.t16); hipFree(f.t16_scales); f.t16 = nullptr; f.t16_scales = nullptr; }
        const size_t row0 = which == JH_W_Q ? 0 : (which == JH_W_K ? (size_t)A : (size_t)(A + KV));
        uint8_t* dd = (uint8_t*)f.data + row0 * row_bytes;
        float* ds = f.scales ? f.scales + row0 * (E / QB) : nullptr;
        HIPCHK(hipMemcpy(dd, data, bytes, kind));
        if (sbytes) HIPCHK(hipMemcpy(ds, scales, sbytes, kind));
        if (!w->data) m->weight_bytes += (int64_t)(bytes + sbytes);
        w->data = dd; w->scales = ds; w->dtype = dtype; w->rows = rows; w->cols = cols;
        m->weights_version++;
        return JH_OK;
    }
    if (w->data) hipFree(w->data);
    if (w->scales) hipFree(w->scales);
    if (w->tiled) { hipFree(w->tiled); hipFree(w->tiled_scales); w->tiled = nullptr; w->tiled_scales = nullptr; }
    if (w->p16t) { hipFree(w->p16t); w->p16t = nullptr; }
    if (w->t16) { hipFree(w->t16); hipFree(w->t16_scales); w->t16 = nullptr; w->t16_scales = nullptr; }
    if (layer >= 0 && (which == JH_W_GATE || which == JH_W_UP)) {
        JWeight& gu = m->gateup[(size_t)layer];
        if (gu.tiled) { hipFree(gu.tiled); hipFree(gu.tiled_scales); gu.tiled = nullptr; gu.tiled_scales = nullptr; }
        if (gu.t16) { hipFree(gu.t16); hipFree(gu.t16_scales); gu.t16 = nullptr; gu.t16_scales = nullptr; }
    }
    w->data = nullptr; w->scales = nullptr;
    hipError_t e = hipMalloc(&w->data, bytes + 64);
    if (e != hipSuccess) return set_err(JH_ERR_OOM, std::string("hipMalloc weight: ") + hipGetErrorString(e));
    HIPCHK(hipMemcpy(w->data, data, bytes, kind));
    if (sbytes) {
        e = hipMalloc((void**)&w->scales, sbytes + 64);
        if (e != hipSuccess) return set_err(JH_ERR_OOM, std::string("hipMalloc scales: ") + hipGetErrorString(e));
        HIPCHK(hipMemcpy(w->scales, scales, sbytes, kind));
    }
    if (widened_dev) hipFree(widened_dev);
    w->dtype = dtype; w->rows = rows; w->cols = cols;
    if (!is_norm && which != JH_W_EMBED) m->weight_bytes += (int64_t)(bytes + sbytes);
    m->weights_version++;
    return JH_OK;
}
int64_t jh_model_weight_bytes(jh_model* m) { return m ? m->weight_bytes : 0; }
int64_t jh_model_released_bytes(jh_model* m) { return m ? m->dropped_bytes : 0; }
int64_t jh_model_tiled_bytes(jh_model* m) {
    if (!m) return 0;
    int64_t b = 0;
    auto add = [&](const JWeight& w) {
        if (w.tiled) b += (int64_t)(tiled_w_bytes(w) + (w.tiled_scales ? tiled_s_bytes(w) : 0));
        if (w.t16) b += (int64_t)(t16_w_bytes(w.rows, w.cols) + t16_s_bytes(w.rows, w.cols));
        if (w.p16t) b += (int64_t)((size_t)w.rows * (w.dtype == JH_DT_BF16 ? bf16t_row_bytes(w.cols) : p16t_row_bytes(w.cols)));
    };
    for (const JWeight& w : m->layer_w) add(w);
    for (const JWeight& w : m->qkv) add(w);
    for (const JWeight& w : m->gateup) add(w);
    for (const JWeight& w : m->global_w) add(w);
    return b;
}

static int session_init(jh_session* s, jh_model* m, int max_ctx, int64_t max_page_bytes);
int jh_session_create(jh_model* m, int max_ctx, int64_t max_page_bytes, jh_session** out) {
    if (!m || !out || max_ctx <= 0) return set_err(JH_ERR_INVALID, "session_create: bad argument");
    HIPCHK(hipSetDevice(m->device));
    if (max_ctx > m->c.context_length) max_ctx = m->c.context_length;
    jh_session* s = new jh_session();
    s->m = m;
    const int rc = session_init(s, m, max_ctx, max_page_bytes);
    if (rc != JH_OK) {   // a half-built session must not leak its stream / slabs (the error text survives the destroy)
        const std::string keep = g_err;
        jh_session_destroy(s);
        g_err = keep;
        return rc;
    }
    *out = s;
    return JH_OK;
}
static int session_init(jh_session* s, jh_model* m, int max_ctx, int64_t max_page_bytes) {
    const jh_config& c = m->c;
    HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    const int nl = c.layer_end - c.layer_start;
    const int KV = c.n_kv_heads * c.head_size, A = c.n_heads * c.head_size, E = c.embedding_length, H = c.hidden_length;
    int32_t geo[2];
    JHCHK(jh_kv_page_geometry(max_page_bytes > 0 ? max_page_bytes : (1 << 23), nl, c.context_length, KV, 4, geo));
    s->layers_per_page = geo[0];
    s->ctx_per_page = geo[1];
    s->n_layer_pages = (nl + geo[0] - 1) / geo[0];
    s->n_ctx_pages = (c.context_length + geo[1] - 1) / geo[1];
    s->n_ctx_alloc = (max_ctx + geo[1] - 1) / geo[1];
    s->max_ctx = max_ctx;
    s->pages_host.assign((size_t)s->n_layer_pages * s->n_ctx_pages, nullptr);
    // KvBufferCache pages (KvBufferCache.java:99-112: [layersPerPage, 2, ctxPerPage, kvLength] F32 each) carved out of
    // ONE slab: the attention kernel computes a row's address arithmetically instead of chasing a page pointer (a
    // dependent global load on its critical path).  The page table is still materialised for hosts / taps.
    const size_t page_elems = (size_t)geo[0] * 2 * geo[1] * KV;
    const size_t page_bytes = page_elems * 4;
    const size_t slab_bytes = page_bytes * s->n_layer_pages * s->n_ctx_alloc;
    {
        hipError_t e = hipMalloc(&s->kv_slab, slab_bytes);
        if (e != hipSuccess) return set_err(JH_ERR_OOM, "hipMalloc KV pages");
        HIPCHK(hipMemset(s->kv_slab, 0, slab_bytes));
    }
    for (int lp = 0; lp < s->n_layer_pages; lp++)
        for (int cp = 0; cp < s->n_ctx_alloc; cp++)
            s->pages_host[(size_t)lp * s->n_ctx_pages + cp] = s->kv_slab + ((size_t)lp * s->n_ctx_alloc + cp) * page_elems;
    s->page_elems = page_elems;
    HIPCHK(hipMalloc(&s->pages_dev, s->pages_host.size() * sizeof(float*)));
    HIPCHK(hipMemcpy(s->pages_dev, s->pages_host.data(), s->pages_host.size() * sizeof(float*), hipMemcpyHostToDevice));
    s->max_splits = opt_int("JH_ATTN_SPLITS", 16);
    if (s->max_splits < 1) s->max_splits = 1;
    s->chunk_cap = (max_ctx + s->max_splits - 1) / s->max_splits;
    if (s->chunk_cap < 32) s->chunk_cap = 32;
    if (s->chunk_cap < 128) s->chunk_cap = 128;
    HIPCHK(hipMalloc(&s->x, (size_t)E * 4));
    HIPCHK(hipMalloc(&s->x1, (size_t)E * 4));
    HIPCHK(hipMalloc(&s->qkv, (size_t)(A + 2 * KV) * 4));
    HIPCHK(hipMalloc(&s->attf, (size_t)A * 4));
    HIPCHK(hipMalloc(&s->tapq, (size_t)A * 4));
    HIPCHK(hipMalloc(&s->hf, (size_t)H * 4));
    HIPCHK(hipMalloc(&s->logits, (size_t)c.vocab_size * 4));
    HIPCHK(hipMalloc(&s->amax_v, 4096 * 4));
    HIPCHK(hipMalloc(&s->amax_i, 4096 * 4));
    // long contexts (> long_min rows) are bandwidth-bound and want the whole chip: up to long_splits slices (mid_splits up to
    // mid_max rows) -- 8100 rows: 518 vs 464 tok/s with 32 instead of 16, 4096 rows: 560 vs 537 with 24; <= 1024 rows lose with more than 16
    s->long_splits = opt_int("JH_ATTN_LONG_SPLITS", 32);
    s->long_min = opt_int("JH_ATTN_LONG_MIN", 2048);
    s->mid_splits = opt_int("JH_ATTN_MID_SPLITS", 24);
    s->mid_max = opt_int("JH_ATTN_MID_MAX", 6144);
    if (s->long_splits > 64) s->long_splits = 64;
    if (s->long_splits <= s->max_splits) s->long_splits = 0;   // no separate tier
    s->part_stride = s->max_splits > 4 ? s->max_splits : 4;
    if (s->long_splits > s->part_stride) s->part_stride = s->long_splits;
    // "direct" attention: contexts of up to 4 slices x 128 rows are combined by the o-projection's prologue
    s->direct_chunk = 128;
    s->direct_max = 0;   // ("direct" mode -- the o-projection's prologue combining the attention slices -- measured slower, profiles/NOTEBOOK_rounds_1-4.md 3: removed)
    HIPCHK(hipMalloc(&s->part_o, (size_t)c.n_heads * s->part_stride * c.head_size * 4));
    HIPCHK(hipMalloc(&s->part_ml, (size_t)c.n_heads * s->part_stride * 2 * 4));
    HIPCHK(hipMemset(s->part_o, 0, (size_t)c.n_heads * s->part_stride * c.head_size * 4));
    HIPCHK(hipMemset(s->part_ml, 0, (size_t)c.n_heads * s->part_stride * 2 * 4));
    HIPCHK(hipMalloc(&s->counters, (size_t)c.n_kv_heads * 4));
    HIPCHK(hipMemset(s->counters, 0, (size_t)c.n_kv_heads * 4));
    HIPCHK(hipMalloc(&s->st, sizeof(DecodeState)));
    HIPCHK(hipMemset(s->st, 0, sizeof(DecodeState)));
    HIPCHK(hipEventCreate(&s->ev0));
    HIPCHK(hipEventCreate(&s->ev1));
    JHCHK(ensure_out_tokens(s, 1024));
    const int cu = g_cu_count;
    // launch plans: see launch_gemv_i8q4 (env overrides are for tuning sweeps only)
    s->cfg_qkv = LaunchCfg{opt_int("JH_QKV_R", 0), opt_int("JH_QKV_WAVES", 0), cu * opt_int("JH_QKV_GRIDX", 1), opt_int("JH_QKV_PIPE", -1)};
    s->cfg_o = LaunchCfg{opt_int("JH_O_R", 0), opt_int("JH_O_WAVES", 0), cu * opt_int("JH_O_GRIDX", 1), opt_int("JH_O_PIPE", -1)};
    s->cfg_gateup = LaunchCfg{opt_int("JH_GATEUP_R", 0), opt_int("JH_GATEUP_WAVES", 0), cu * opt_int("JH_GATEUP_GRIDX", 1), opt_int("JH_GATEUP_PIPE", -1)};
    s->cfg_down = LaunchCfg{opt_int("JH_DOWN_R", 0), opt_int("JH_DOWN_WAVES", 0), cu * opt_int("JH_DOWN_GRIDX", 1), opt_int("JH_DOWN_PIPE", -1)};
    s->cfg_lm = LaunchCfg{opt_int("JH_LM_R", 2), opt_int("JH_LM_WAVES", 8), cu * opt_int("JH_LM_GRIDX", 2), 1};   // tools/sweep_lm.py
    if (s->cfg_lm.grid_cap > 4096) s->cfg_lm.grid_cap = 4096;
    s->prefill_batch_min = opt_int("JH_PREFILL_BATCH_MIN", 4);   // chunks of fewer rows go row by row; 0 disables batching
    s->prefill_attn_mfma_min = opt_int("JH_PREFILL_ATTN_MFMA_MIN", 384);   // -1: always the per-row kernel; 0: always the MFMA kernel
    s->graphs_version = m->weights_version;
    HIPCHK(hipMalloc(&s->eos_dev, (1 + JH_MAX_EOS) * sizeof(int)));
    HIPCHK(hipMemset(s->eos_dev, 0, (1 + JH_MAX_EOS) * sizeof(int)));
    HIPCHK(hipHostMalloc((void**)&s->st_host, 2 * sizeof(DecodeState), hipHostMallocDefault));
    memset(s->st_host, 0, 2 * sizeof(DecodeState));
    HIPCHK(hipEventCreateWithFlags(&s->ev_chunk[0], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&s->ev_chunk[1], hipEventDisableTiming));
    s->strict = opt_int("JH_STRICT_ORDER", 0) ? 1 : 0;
    s->p16_depth = opt_int("JH_P16_D", 8);   // upper bound of the prefetch depth (p16_depth_for)
    s->tokens_per_graph = opt_int("JH_TOKENS_PER_GRAPH", 4);   // 1, 2, 4, 8 or 16 (a divisor of the stop-token snapshot interval)
    if (s->tokens_per_graph != 2 && s->tokens_per_graph != 4 && s->tokens_per_graph != 8 && s->tokens_per_graph != 16) s->tokens_per_graph = 1;
    // reference-order attention: one slice of the context per 16 positions of max_ctx, at least 16, at most 256 (the slices that
    // lie beyond the current position return at once)
    s->p16_att_splits = opt_int("JH_P16_ATT_SPLITS", 0);
    if (s->p16_att_splits <= 0) s->p16_att_splits = (max_ctx + 15) / 16;
    if (s->p16_att_splits < 16) s->p16_att_splits = 16;
    if (s->p16_att_splits > 256) s->p16_att_splits = 256;
    // the id buffer of the device loop at its final size NOW: its address is baked into the captured decode graphs, and growing it
    // later (a decode_n(…, 1) that captures, then decode_n(…, K)) would drop and re-capture them inside the caller's timed region
    JHCHK(ensure_out_tokens(s, max_ctx));
    s->p16_sc_stride = (max_ctx + 63) & ~63;
    HIPCHK(hipMalloc(&s->p16_scores, (size_t)c.n_heads * s->p16_sc_stride * 4));
    if (s->strict && c.weight_dtype != JH_DT_Q4 && c.weight_dtype != JH_DT_BF16) return set_err(JH_ERR_UNSUPPORTED, "JH_STRICT_ORDER: reference-order kernels exist for JQ4 and BF16 models");
    if (s->strict) JHCHK(ensure_strict_operands(s, s->stream));   // a session that STARTS in reference order (JH_STRICT_ORDER=1): same copies as jh_session_set_strict makes
    if (!s->strict && !s->m->strict_only && !opt_int("JH_STRICT_ONLY", 0) && prefill_batch_ok(s)) {   // (a reference-order-only model never runs the order-free prompt GEMM)
        // the MFMA-ordered weight copies of the prefill GEMM are made here, once per model, not inside the first prompt
        JHCHK(ensure_all_tiled(s, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return JH_OK;
}
int jh_session_set_strict(jh_session* s, int on) {
    if (!s) return set_err(JH_ERR_INVALID, "set_strict: null");
    if (on && s->m->c.weight_dtype != JH_DT_Q4 && s->m->c.weight_dtype != JH_DT_BF16) return set_err(JH_ERR_UNSUPPORTED, "set_strict: reference-order kernels exist for JQ4 and BF16 models");
    HIPCHK(hipSetDevice(s->m->device));
    if (!on && s->m->strict_only) return set_err(JH_ERR_UNSUPPORTED, "set_strict(0): this model keeps reference-order operands only (JH_STRICT_ONLY released the row-major weights)");
    if ((on ? 1 : 0) != s->strict) {
        HIPCHK(hipStreamSynchronize(s->stream));
        s->strict = on ? 1 : 0;
        s->graphs_version = -1;   // the captured graphs hold the other mode's kernels
        drop_stale_graphs(s);
        JHCHK(ensure_strict_operands(s, s->stream));   // T16 copies of the weights the MFMA GEMVs read (jh_t16.h)
    }
    return JH_OK;
}
int jh_session_set_eos(jh_session* s, const int32_t* eos_ids, int n_eos) {
    if (!s || n_eos < 0 || (n_eos > 0 && !eos_ids)) return set_err(JH_ERR_INVALID, "set_eos: bad argument");
    if (n_eos > JH_MAX_EOS) return set_err(JH_ERR_INVALID, "set_eos: at most 16 stop tokens");
    if (n_eos == s->n_eos && (n_eos == 0 || memcmp(eos_ids, s->eos_host, (size_t)n_eos * sizeof(int)) == 0)) return JH_OK;   // unchanged
    HIPCHK(hipSetDevice(s->m->device));
    // the list lives in a fixed device buffer that finish_token_kernel reads at run time: no captured graph is invalidated.
    // Ordered on the session's stream behind whatever decode is still queued.
    int buf[1 + JH_MAX_EOS] = {0};
    buf[0] = n_eos;
    for (int i = 0; i < n_eos; i++) { buf[1 + i] = eos_ids[i]; s->eos_host[i] = eos_ids[i]; }
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(s->eos_dev, buf, sizeof(buf), hipMemcpyHostToDevice));
    s->n_eos = n_eos;
    return JH_OK;
}
int jh_decode_generated(jh_session* s, int32_t* out_n) {
    if (!s || !out_n) return set_err(JH_ERR_INVALID, "decode_generated: null");
    *out_n = s->generated;
    return JH_OK;
}
int jh_session_destroy(jh_session* s) {
    if (!s) return JH_OK;
    hipSetDevice(s->m->device);
    if (s->stream) hipStreamSynchronize(s->stream);
    for (int v = 0; v < N_ATTN_VARIANTS; v++) {
        if (s->exec_s[v]) hipGraphExecDestroy(s->exec_s[v]);
        if (s->graph_s[v]) hipGraphDestroy(s->graph_s[v]);
        if (s->exec[v]) hipGraphExecDestroy(s->exec[v]);
        if (s->exec_m[v]) { hipGraphExecDestroy(s->exec_m[v]); hipGraphDestroy(s->graph_m[v]); }
        if (s->graph[v]) hipGraphDestroy(s->graph[v]);
        if (s->row_exec[v]) hipGraphExecDestroy(s->row_exec[v]);
        if (s->row_graph[v]) hipGraphDestroy(s->row_graph[v]);
    }
    if (s->kv_slab) hipFree(s->kv_slab);
    void* bufs[] = {s->pages_dev, s->x, s->x1, s->qkv, s->attf, s->tapq, s->hf, s->logits,
                    s->amax_v, s->amax_i, s->part_o, s->part_ml, s->counters, s->st, s->out_tokens};
    for (void* b : bufs) if (b) hipFree(b);
    for (float* t : s->taps) if (t) hipFree(t);
    for (void* b : {(void*)s->pb_x, (void*)s->pb_x1, (void*)s->pb_qkv, (void*)s->pb_att, (void*)s->pb_g, (void*)s->pb_ad, (void*)s->pb_aq, (void*)s->pb_tok, (void*)s->pb_ws, (void*)s->pb_start, (void*)s->pb_att_o, (void*)s->pb_att_ml, (void*)s->tile_w, (void*)s->tile_s, (void*)s->p16_scores_b, (void*)s->pb_sel, (void*)s->pb_sad, (void*)s->pb_bfr}) if (b) hipFree(b);
    for (auto& kv : s->pb_graphs) hipGraphExecDestroy(kv.second);
    for (hipGraph_t g : s->pb_graph_src) hipGraphDestroy(g);
    if (s->ev0) hipEventDestroy(s->ev0);
    if (s->ev1) hipEventDestroy(s->ev1);
    for (hipEvent_t e : s->ev_chunk) if (e) hipEventDestroy(e);
    if (s->st_host) hipHostFree(s->st_host);
    if (s->eos_dev) hipFree(s->eos_dev);
    if (s->p16_scores) hipFree(s->p16_scores);
    if (s->prob) hipFree(s->prob);
    if (s->u_dev) hipFree(s->u_dev);
    if (s->pick) hipFree(s->pick);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
    return JH_OK;
}
int jh_session_page_info(jh_session* s, int32_t* out4) {
    if (!s || !out4) return set_err(JH_ERR_INVALID, "page_info: null");
    out4[0] = s->layers_per_page; out4[1] = s->ctx_per_page; out4[2] = s->n_layer_pages; out4[3] = s->n_ctx_pages;
    return JH_OK;
}
void* jh_session_stream(jh_session* s) { return s ? (void*)s->stream : nullptr; }

}  // extern "C"
