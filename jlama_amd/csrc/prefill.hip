// prefill.hip -- part of libjlamahip.so (C ABI: include/jlama_hip.h).  Prompt chunks (batchForward): MFMA GEMMs of the order-free path, the M-row reference-order forms, their attention.
#include "jh_host.h"
#include "jh_launch.h"

// ---- batched prefill -----------------------------------------------------------------------------------------------

// reference-order sessions: prompt rows through the M-row p16 GEMM (jh_p16.h) -- whole groups of 16 Q blocks in every K
bool prefill_p16_ok(jh_session* s) {
    const int enabled = opt_int("JH_P16_PREFILL", 1);
    const jh_config& c = s->m->c;
    if (!enabled || !s->strict || c.weight_dtype != JH_DT_Q4) return false;
    const int hs = c.head_size, A = c.n_heads * hs, group = c.n_heads / c.n_kv_heads;
    if (c.embedding_length % 512 || c.hidden_length % 512 || A % 512 || c.hidden_length > 32768 || c.embedding_length > 32768 || A > 32768) return false;
    return (hs == 128 || hs == 64) && (group == 1 || group == 2 || group == 4 || group == 8);
}
// ... and through the F16-MFMA form of that GEMM (jh_t16.h: gemm_t16_kernel) when every projection has whole T16 tiles
bool prefill_t16_ok(jh_session* s) {
    const int enabled = opt_int("JH_T16_PREFILL", 1);
    const jh_config& c = s->m->c;
    if (!enabled || !prefill_p16_ok(s)) return false;                 // E, H, A are multiples of 512 (whole chunks of 16 blocks)
    const int hs = c.head_size, A = c.n_heads * hs, KV = c.n_kv_heads * hs;
    return (A + 2 * KV) % 16 == 0 && c.embedding_length % 16 == 0 && c.hidden_length % 8 == 0 && t16_shape_ok(c.embedding_length);
}
// reference-order BF16 sessions: prompt rows through gemm_bf16r_kernel (jh_bf16r.h) -- whole pairs of 128-element groups in every K
bool prefill_bf16r_ok(jh_session* s) {
    const jh_config& c = s->m->c;
    if (!opt_int("JH_BF16R_PREFILL", 1) || !s->strict || c.weight_dtype != JH_DT_BF16) return false;
    const int hs = c.head_size, A = c.n_heads * hs, group = c.n_heads / c.n_kv_heads;
    if (c.embedding_length % 256 || c.hidden_length % 256 || A % 256) return false;
    return (hs == 128 || hs == 64) && (group == 1 || group == 2 || group == 4 || group == 8);
}
bool prefill_batch_ok(jh_session* s) {
    const jh_config& c = s->m->c;
    if (s->prefill_batch_min <= 0 || s->tap_layer >= 0) return false;
    if (s->strict && c.weight_dtype == JH_DT_BF16) return prefill_bf16r_ok(s);
    if (s->strict) return prefill_t16_ok(s);
    if (c.weight_dtype != JH_DT_Q4 && c.weight_dtype != JH_DT_BF16) return false;
    const int hs = c.head_size, A = c.n_heads * hs, KV = c.n_kv_heads * hs, group = c.n_heads / c.n_kv_heads;
    if (c.embedding_length % 64 || c.hidden_length % 64 || A % 64 || (A + 2 * KV) % 32) return false;
    if (c.weight_dtype == JH_DT_Q4 && (c.embedding_length % 256 || c.hidden_length % 256 || A % 256)) return false;   // tiled MFMA GEMMs only
    if (!((hs == 128 || hs == 64) && (group == 1 || group == 2 || group == 4 || group == 8))) return false;
    return true;
}
size_t prefill_attn_lds(const jh_config& c, int n_keys) {
    const int group = c.n_heads / c.n_kv_heads, hs = c.head_size;
    const int rps = PF_THREADS / (hs / 4);
    return ((size_t)rps * group * hs + (size_t)group * n_keys) * 4;
}
bool prefill_attn_mfma(const jh_session* s, int start_pos, int rows) {   // blockwise MFMA kernel for this chunk?
    return s->prefill_attn_mfma_min >= 0 && start_pos + rows >= s->prefill_attn_mfma_min && rows >= 2;
}
bool prefill_chunk_fits(jh_session* s, int start_pos, int rows) {   // per-row kernel: the score rows of the last position must fit in LDS
    return prefill_attn_mfma(s, start_pos, rows) || prefill_attn_lds(s->m->c, start_pos + rows) <= 150 * 1024;
}
int prefill_alloc(jh_session* s) {
    if (s->pb_rows) return JH_OK;
    const jh_config& c = s->m->c;
    const size_t E = c.embedding_length, H = c.hidden_length, A = (size_t)c.n_heads * c.head_size, KV = (size_t)c.n_kv_heads * c.head_size;
    size_t kmax = E > H ? E : H;
    if (A > kmax) kmax = A;
    const size_t R = PB_MAX_ROWS;
    HIPCHK(hipMalloc(&s->pb_x, R * E * 4));
    HIPCHK(hipMalloc(&s->pb_x1, R * E * 4));
    HIPCHK(hipMalloc(&s->pb_qkv, R * (A + 2 * KV) * 4));
    HIPCHK(hipMalloc(&s->pb_att, R * A * 4));
    HIPCHK(hipMalloc(&s->pb_g, R * 2 * H * 4));   // [rows][gate | up] when the fused gate|up GEMM runs, else gate [rows][H] + up behind it
    s->pb_u = s->pb_g + R * H;
    HIPCHK(hipMalloc(&s->pb_aq, R * kmax * (c.weight_dtype == JH_DT_BF16 ? 2 : 1)));   // Q8 codes, or BF16 rows for a BF16 model
    HIPCHK(hipMalloc(&s->pb_ad, R * (kmax / QB) * 4));
    HIPCHK(hipMalloc(&s->pb_tok, R * 4));
    HIPCHK(hipMalloc(&s->pb_start, 64));
    HIPCHK(hipMalloc(&s->pb_ws, BF16_SPLITK_WS_BYTES));   // split-K partials (BF16 tile GEMM, I8xQ4 LDS GEMM)
    HIPCHK(hipMalloc(&s->pb_att_o, R * c.n_heads * PF_MAX_SPLIT * c.head_size * 4));
    HIPCHK(hipMalloc(&s->pb_att_ml, R * c.n_heads * PF_MAX_SPLIT * 2 * 4));
    s->pb_rows = PB_MAX_ROWS;
    return JH_OK;
}
// The prefill GEMM wants both operands in MFMA order (gemm_q8q4_tile_kernel, TILED): possible when K % 128 == 0
bool prefill_tiled(jh_session* s, int K) {
    const int enabled = opt_int("JH_PREFILL_TILED", 1);
    const int nblk = K / QB;
    if (s->m->c.weight_dtype == JH_DT_BF16) return enabled && (K % 32) == 0;
    return enabled && s->m->c.weight_dtype == JH_DT_Q4 && nblk % 8 == 0 && (size_t)nblk * 128 <= 150 * 1024;
}
// one weight, row-major -> MFMA order, into wt (+ st for Q4)
int retile_launch(const JWeight& W, uint8_t* wt, float* st_out, hipStream_t st) {
    const bool bf = W.dtype == JH_DT_BF16;
    if ((!bf && W.dtype != JH_DT_Q4) || (W.rows % 32) || (W.cols % (bf ? 16 : QB))) return set_err(JH_ERR_INVALID, "tiled copy: shape");
    const int nch = bf ? W.cols / 8 : W.cols / QB;    // 16-byte chunks per row
    hipLaunchKernelGGL(retile16_kernel, dim3((unsigned)((nch + 63) / 64), (unsigned)(W.rows / 32)), dim3(256), 0, st, (const i32x4*)W.data,
                       bf ? (const float*)nullptr : (const float*)W.scales, W.rows, nch, (i32x4*)wt, bf ? (float*)nullptr : st_out);
    HIPCHK(hipGetLastError());
    return JH_OK;
}
size_t tiled_w_bytes(const JWeight& W) { return W.dtype == JH_DT_BF16 ? (size_t)W.rows * W.cols * 2 : (size_t)W.rows * (W.cols / QB) * 16; }
size_t tiled_s_bytes(const JWeight& W) { return W.dtype == JH_DT_BF16 ? 0 : (size_t)W.rows * (W.cols / QB) * 4; }
int tiled_mode_for(jh_model* m) {
    if (m->tiled_mode != TILED_UNSET) return m->tiled_mode;
    const int want = opt_int("JH_TILED_COPY", 0);   // 0 auto, 1 resident, 2 transient
    int mode = TILED_RESIDENT;
    if (want == 2) mode = TILED_TRANSIENT;
    else if (want != 1) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr < (size_t)m->weight_bytes + tot / 4) mode = TILED_TRANSIENT;
        (void)hipGetLastError();
    }
    m->tiled_mode = mode;
    return mode;
}
// resident re-tiled copy of a weight, made on first use (costs a second copy of the weights in HBM)
int ensure_tiled(JWeight& W, hipStream_t st) {
    if (W.tiled) return JH_OK;
    if ((W.dtype != JH_DT_Q4 && W.dtype != JH_DT_BF16) || (W.rows % 32) || (W.cols % QB)) return set_err(JH_ERR_INVALID, "tiled copy: shape");
    hipError_t e = hipMalloc((void**)&W.tiled, tiled_w_bytes(W));
    if (e == hipSuccess && W.dtype == JH_DT_Q4) e = hipMalloc((void**)&W.tiled_scales, tiled_s_bytes(W));
    if (e != hipSuccess) return set_err(JH_ERR_OOM, "hipMalloc tiled weight copy");
    return retile_launch(W, W.tiled, W.tiled_scales, st);
}
// the MFMA-ordered operand of W for the GEMM that is launched next on `st`: the resident copy, or the session's scratch filled now
int tiled_operand(jh_session* s, JWeight& W, hipStream_t st, const uint8_t** tw, const float** ts) {
    if (tiled_mode_for(s->m) == TILED_RESIDENT) {
        JHCHK(ensure_tiled(W, st));
        *tw = W.tiled; *ts = W.tiled_scales;
        return JH_OK;
    }
    if (tiled_w_bytes(W) > s->tile_w_bytes || tiled_s_bytes(W) > s->tile_s_bytes) return set_err(JH_ERR_INVALID, "tiled operand: scratch too small");
    JHCHK(retile_launch(W, s->tile_w, s->tile_s, st));
    *tw = s->tile_w; *ts = s->tile_s;
    return JH_OK;
}
// gate and up stacked along N in ONE MFMA-ordered operand: the prefill runs a single [rows, 2H] GEMM for both
// (MLPBlock.java:117-130 issues them over the same quantized activation), out[:, :H] = gate, out[:, H:] = up
bool gateup_fusable(jh_session* s, int li) {
    const JWeight* W = &s->m->layer_w[(size_t)li * JH_W_COUNT];
    const JWeight &G = W[JH_W_GATE], &U = W[JH_W_UP];
    return G.data && U.data && G.rows == U.rows && G.cols == U.cols && G.dtype == U.dtype && (G.rows % 32) == 0 &&
           (G.dtype == JH_DT_Q4 || G.dtype == JH_DT_BF16) && prefill_tiled(s, G.cols);
}
// fills (wt, st_out) with [gate ; up] in MFMA order
int retile_gateup(const JWeight& G, const JWeight& U, uint8_t* wt, float* st_out, hipStream_t st) {
    JHCHK(retile_launch(G, wt, st_out, st));
    return retile_launch(U, wt + tiled_w_bytes(G), st_out ? st_out + tiled_s_bytes(G) / 4 : nullptr, st);
}
int ensure_gateup_tiled(jh_session* s, int li, hipStream_t st) {
    jh_model* m = s->m;
    JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    JWeight& G = W[JH_W_GATE];
    JWeight& U = W[JH_W_UP];
    JWeight& F = m->gateup[(size_t)li];
    if (F.tiled || !gateup_fusable(s, li)) return JH_OK;
    F.dtype = G.dtype; F.rows = 2 * G.rows; F.cols = G.cols;
    hipError_t e = hipMalloc((void**)&F.tiled, tiled_w_bytes(F));
    if (e == hipSuccess && F.dtype == JH_DT_Q4) e = hipMalloc((void**)&F.tiled_scales, tiled_s_bytes(F));
    if (e != hipSuccess) return set_err(JH_ERR_OOM, "hipMalloc tiled gate|up copy");
    return retile_gateup(G, U, F.tiled, F.tiled_scales, st);
}
int gateup_operand(jh_session* s, int li, hipStream_t st, const uint8_t** tw, const float** ts) {
    jh_model* m = s->m;
    JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    if (tiled_mode_for(m) == TILED_RESIDENT) {
        JHCHK(ensure_gateup_tiled(s, li, st));
        *tw = m->gateup[(size_t)li].tiled; *ts = m->gateup[(size_t)li].tiled_scales;
        return JH_OK;
    }
    if (2 * tiled_w_bytes(W[JH_W_GATE]) > s->tile_w_bytes || 2 * tiled_s_bytes(W[JH_W_GATE]) > s->tile_s_bytes)
        return set_err(JH_ERR_INVALID, "tiled operand: scratch too small");
    JHCHK(retile_gateup(W[JH_W_GATE], W[JH_W_UP], s->tile_w, s->tile_s, st));
    *tw = s->tile_w; *ts = s->tile_s;
    return JH_OK;
}
// every weight the prefill GEMMs of this shard will touch (allocation must not happen inside a graph capture): the resident
// MFMA-ordered copies, or -- TILED_TRANSIENT -- the session's scratch, sized for the largest operand
int ensure_all_tiled(jh_session* s, hipStream_t st) {
    JHCHK(refuse_order_free(s, "MFMA-ordered prompt operands"));
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const bool resident = tiled_mode_for(m) == TILED_RESIDENT;
    size_t need_w = 0, need_s = 0;
    for (int li = c.layer_start; li < c.layer_end; li++) {
        JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
        JWeight* list[] = {&m->qkv[(size_t)li], &W[JH_W_O], &W[JH_W_DOWN], &W[JH_W_GATE], &W[JH_W_UP]};
        for (JWeight* w : list) {
            if (!(w->data && (w->dtype == JH_DT_Q4 || w->dtype == JH_DT_BF16) && prefill_tiled(s, w->cols) && (w->rows % 32) == 0)) continue;
            const bool gu = (w == &W[JH_W_GATE] || w == &W[JH_W_UP]);
            const bool fused = gu && gateup_fusable(s, li);
            if (resident) {
                if (!fused && !w->tiled) JHCHK(ensure_tiled(*w, st));
            } else {
                const size_t f = fused ? 2 : 1;
                if (f * tiled_w_bytes(*w) > need_w) need_w = f * tiled_w_bytes(*w);
                if (f * tiled_s_bytes(*w) > need_s) need_s = f * tiled_s_bytes(*w);
            }
        }
        if (resident) JHCHK(ensure_gateup_tiled(s, li, st));
    }
    if (!resident && (need_w > s->tile_w_bytes || need_s > s->tile_s_bytes)) {
        HIPCHK(hipStreamSynchronize(st));
        if (s->tile_w) hipFree(s->tile_w);
        if (s->tile_s) hipFree(s->tile_s);
        s->tile_w = nullptr; s->tile_s = nullptr; s->tile_w_bytes = s->tile_s_bytes = 0;
        if (hipMalloc((void**)&s->tile_w, need_w) != hipSuccess || (need_s && hipMalloc((void**)&s->tile_s, need_s) != hipSuccess))
            return set_err(JH_ERR_OOM, "hipMalloc tiled operand scratch");
        s->tile_w_bytes = need_w; s->tile_s_bytes = need_s;
        for (auto& kv : s->pb_graphs) hipGraphExecDestroy(kv.second);   // prefill graphs captured the old scratch address
        s->pb_graphs.clear();
        for (hipGraph_t g : s->pb_graph_src) hipGraphDestroy(g);
        s->pb_graph_src.clear();
    }
    return JH_OK;
}
template <int MODE>
int rows_quant_launch(jh_session* s, const float* x, int ldx, const float* x2, int ldx2, const float* nw, float eps, int K, int rows,
                      hipStream_t st) {
    RowsParams p{x, ldx, x2, ldx2, nw, eps, K, rows, s->pb_aq, prefill_tiled(s, K) ? -1 : K, s->pb_ad, K / QB, nullptr};
    const dim3 grid(rows, MODE == ROWS_RMS ? 1 : (K >= 8192 ? 4 : 2));   // few rows: split the independent blocks of a row over workgroups
    if (s->m->c.weight_dtype == JH_DT_BF16) hipLaunchKernelGGL((rows_bf16_kernel<MODE>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((rows_quant_kernel<MODE>), grid, dim3(256), 0, st, p);
    HIPCHK(hipGetLastError());
    return JH_OK;
}
// out[rows, N] = act[rows, K] x W^T (+ resid) with W given as an MFMA-ordered operand (tw, ts)
int prefill_gemm_operand(jh_session* s, int dtype, const uint8_t* tw, const float* ts, int N, int K, int rows, float* out, int ldc, const float* resid,
                         hipStream_t st) {
    if (dtype == JH_DT_BF16) {
        MfmaBf16TileParams g{(const uint16_t*)s->pb_aq, (const uint16_t*)tw, out, resid, rows, N, K, ldc, s->pb_ws, 1};
        return launch_gemm_bf16_tile(g, st);
    }
    MfmaQ4Params g{s->pb_aq, s->pb_ad, tw, ts, out, resid, rows, 0, N, K, K, K / QB, K / 2, K / QB, ldc, 0};
    return launch_gemm_q8q4_mfma(g, st, true, s->pb_ws, BF16_SPLITK_WS_BYTES);
}
// out[rows, N] = act[rows, K] x W[N, K]^T (+ resid): I8 x Q4 (exact integer MFMA) or BF16 x BF16 (MFMA), by model dtype
int prefill_gemm(jh_session* s, JWeight& W, int N, int K, int rows, float* out, int ldc, const float* resid, hipStream_t st) {
    const bool bf = s->m->c.weight_dtype == JH_DT_BF16;
    if (prefill_tiled(s, K) && (!bf || (N % 32) == 0)) {   // (a Q4 weight whose row count is no multiple of 32 is refused by the re-tiler)
        const uint8_t* tw = nullptr;
        const float* ts = nullptr;
        JHCHK(tiled_operand(s, W, st, &tw, &ts));
        return prefill_gemm_operand(s, bf ? JH_DT_BF16 : JH_DT_Q4, tw, ts, N, K, rows, out, ldc, resid, st);
    }
    if (bf) {
        MfmaGemmParams g{(const uint16_t*)s->pb_aq, (const uint16_t*)W.data, out, rows, 0, N, K, K, K, ldc, 0, resid, s->pb_ws, 1};
        return launch_gemm_bf16_mfma(g, st);
    }
    MfmaQ4Params g{s->pb_aq, s->pb_ad, (const uint8_t*)W.data, W.scales, out, resid, rows, 0, N, K, K, K / QB, K / 2, K / QB, ldc, 0};
    return launch_gemm_q8q4_mfma(g, st);
}
// nkeys_bound >= start_pos + rows sizes the score rows in LDS (the position itself is read from s->pb_start)
int prefill_attn_launch(jh_session* s, int rel, int nkeys_bound, int rows, bool mfma, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const int hs = c.head_size, A = c.n_heads * hs, KV = c.n_kv_heads * hs, group = c.n_heads / c.n_kv_heads;
    PrefillAttnParams p;
    memset(&p, 0, sizeof(p));
    p.qkv = s->pb_qkv; p.ldqkv = A + 2 * KV;
    p.rope = m->rope;
    p.kv_base = s->kv_slab + (size_t)(rel / s->layers_per_page) * s->n_ctx_alloc * s->page_elems;
    p.page_elems = (long long)s->page_elems;
    p.rel_layer_in_page = rel % s->layers_per_page;
    p.ctx_per_page = s->ctx_per_page;
    p.cpp_shift = -1;
    for (int sh = 0; sh < 30; sh++)
        if ((1 << sh) == s->ctx_per_page) p.cpp_shift = sh;
    p.n_heads = c.n_heads; p.n_kv_heads = c.n_kv_heads; p.head_size = hs; p.kv_head_offset = m->kv_head_offset;
    p.start_pos = s->pb_start; p.rows = rows; p.scale = m->attention_scale;
    p.out = s->pb_att; p.ldo = A;
    hipLaunchKernelGGL(rows_rope_kv_kernel, dim3(rows, 4), dim3(256), 0, st, p);   // 4 workgroups per row: the pair loop is latency-bound
    HIPCHK(hipGetLastError());
    if (mfma) {
        // blockwise causal attention on the matrix cores: (query tile of 32 rows) x (kv head) x (key-range split)
        const int qtiles = (rows + 31) / 32, tiles_bound = (nkeys_bound + 31) / 32;
        int S = 1;
        while (S < PF_MAX_SPLIT && qtiles * c.n_kv_heads * S < g_cu_count && tiles_bound / (2 * S) >= 8) S *= 2;
        PrefillMfmaExtra e{S, s->pb_att_o, s->pb_att_ml};
        const size_t lds_m = prefill_mfma_lds(hs, group);
        dim3 grid_m(qtiles * S, c.n_kv_heads), block_m(group * 64);
#define JH_PMFMA(HSV, GV)                                                                         \
    if (hs == HSV && group == GV) {                                                               \
        JHCHK(allow_lds((attn_prefill_mfma_kernel<HSV, GV>), lds_m));                             \
        hipLaunchKernelGGL((attn_prefill_mfma_kernel<HSV, GV>), grid_m, block_m, lds_m, st, p, e);  \
        HIPCHK(hipGetLastError());                                                                \
        if (S > 1) {                                                                              \
            hipLaunchKernelGGL(attn_prefill_combine_kernel, dim3(rows, c.n_heads), dim3(128), 0, st, p, e); \
            HIPCHK(hipGetLastError());                                                            \
        }                                                                                         \
        return JH_OK;                                                                             \
    }
        JH_PMFMA(128, 4) JH_PMFMA(128, 8) JH_PMFMA(64, 4) JH_PMFMA(128, 1) JH_PMFMA(128, 2) JH_PMFMA(64, 1) JH_PMFMA(64, 2) JH_PMFMA(64, 8)
#undef JH_PMFMA
        return set_err(JH_ERR_UNSUPPORTED, "prefill attention: unsupported head geometry");
    }
    const size_t lds = prefill_attn_lds(c, nkeys_bound);
    dim3 grid(c.n_kv_heads, rows), block(PF_THREADS);
#define JH_PATTN(HSV, GV)                                                                  \
    if (hs == HSV && group == GV) {                                                        \
        JHCHK(allow_lds(attn_prefill_kernel<HSV, GV>, lds));                               \
        hipLaunchKernelGGL((attn_prefill_kernel<HSV, GV>), grid, block, lds, st, p);       \
        HIPCHK(hipGetLastError());                                                         \
        return JH_OK;                                                                      \
    }
    JH_PATTN(128, 4) JH_PATTN(128, 8) JH_PATTN(64, 4) JH_PATTN(128, 1) JH_PATTN(128, 2) JH_PATTN(64, 1) JH_PATTN(64, 2) JH_PATTN(64, 8)
#undef JH_PATTN
    return set_err(JH_ERR_UNSUPPORTED, "prefill attention: unsupported head geometry");
}
// the layer loop of one chunk (also what the prefill graphs capture)
// One layer of a prompt chunk is two halves (a tensor-parallel shard's partial results are reduced between them): `resid` non-null
// adds the residual in the GEMM's epilogue (a whole model), null stores the bare projection (a shard's partial rows).
int prefill_weights_set(jh_session* s, int li) {
    jh_model* m = s->m;
    JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    JHCHK(refuse_order_free(s, "prompt chunk"));
    if (!w_present(m->qkv[(size_t)li]) || !w_present(W[JH_W_Q]) || !w_present(W[JH_W_K]) || !w_present(W[JH_W_V]) || !w_present(W[JH_W_O]) || !w_present(W[JH_W_GATE]) ||
        !w_present(W[JH_W_UP]) || !w_present(W[JH_W_DOWN]) || !W[JH_W_NORM1].data || !W[JH_W_NORM2].data)
        return set_err(JH_ERR_INVALID, "layer " + std::to_string(li) + ": weights not set");
    return JH_OK;
}
// preAttentionNorm + maybeQuantize, q|k|v projections, attention, maybeQuantize(valueBatch) + output projection
// (CausalSelfAttention.java:161-171, 364-376; residual TransformerBlock.java:185): rows of s->pb_x -> out
int prefill_attn_half(jh_session* s, int li, int rows, int nkeys_bound, bool attn_mfma, float* out, const float* resid, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const int E = c.embedding_length, hs = c.head_size, A = c.n_heads * hs, KV = c.n_kv_heads * hs;
    JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    JHCHK(prefill_weights_set(s, li));
    JHCHK((rows_quant_launch<ROWS_RMS>(s, s->pb_x, E, nullptr, 0, (const float*)W[JH_W_NORM1].data, c.rms_eps, E, rows, st)));
    JHCHK(prefill_gemm(s, m->qkv[(size_t)li], A + 2 * KV, E, rows, s->pb_qkv, A + 2 * KV, nullptr, st));
    JHCHK(prefill_attn_launch(s, li - c.layer_start, nkeys_bound, rows, attn_mfma, st));
    JHCHK((rows_quant_launch<ROWS_QUANT>(s, s->pb_att, A, nullptr, 0, nullptr, 0.f, A, rows, st)));
    return prefill_gemm(s, W[JH_W_O], E, A, rows, out, E, resid, st);
}
// preFFNorm + maybeQuantize, gate / up, SiLU*up + maybeQuantize, down (MLPBlock.java:117-158; residual TransformerBlock.java:203): rows of x1 -> out
int prefill_ffn_half(jh_session* s, int li, int rows, const float* x1, float* out, const float* resid, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const int E = c.embedding_length, H = c.hidden_length;
    JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    JHCHK((rows_quant_launch<ROWS_RMS>(s, x1, E, nullptr, 0, (const float*)W[JH_W_NORM2].data, c.rms_eps, E, rows, st)));
    if (gateup_fusable(s, li)) {   // one GEMM for gate|up: out[:, :H] = gate, out[:, H:] = up
        const uint8_t* tw = nullptr;
        const float* ts = nullptr;
        JHCHK(gateup_operand(s, li, st, &tw, &ts));
        JHCHK(prefill_gemm_operand(s, W[JH_W_GATE].dtype, tw, ts, 2 * H, E, rows, s->pb_g, 2 * H, nullptr, st));
        JHCHK((rows_quant_launch<ROWS_SILU_MUL>(s, s->pb_g, 2 * H, s->pb_g + H, 2 * H, nullptr, 0.f, H, rows, st)));
    } else {
        JHCHK(prefill_gemm(s, W[JH_W_GATE], H, E, rows, s->pb_g, H, nullptr, st));
        JHCHK(prefill_gemm(s, W[JH_W_UP], H, E, rows, s->pb_u, H, nullptr, st));
        JHCHK((rows_quant_launch<ROWS_SILU_MUL>(s, s->pb_g, H, s->pb_u, H, nullptr, 0.f, H, rows, st)));
    }
    return prefill_gemm(s, W[JH_W_DOWN], E, H, rows, out, E, resid, st);
}
int prefill_layers(jh_session* s, int rows, int nkeys_bound, bool attn_mfma, hipStream_t st) {
    const jh_config& c = s->m->c;
    for (int li = c.layer_start; li < c.layer_end; li++) {
        JHCHK(prefill_attn_half(s, li, rows, nkeys_bound, attn_mfma, s->pb_x1, s->pb_x, st));
        JHCHK(prefill_ffn_half(s, li, rows, s->pb_x1, s->pb_x, s->pb_x1, st));
        JHCHK(trace_sync("prefill layer", st));
    }
    return JH_OK;
}
// ---- the same chunk in reference order: one-hot selector operands per row + T16 GEMMs on the F16 MFMA (jh_t16.h), KV rows of the
// whole chunk, then scores / softmax + value chains of every row in one launch each (jh_p16.h)
// ---- the same GEMMs on the F16 MFMA (jh_t16.h): activations as one-hot selector operands, weights in T16 order
template <int PRO>
int rows_act_t16_launch(jh_session* s, const float* x, int ldx, const float* nw, float eps, int K, int rows, hipStream_t st) {
    RowsT16Params rp{x, ldx, nw, eps, K, (i32x4*)s->pb_sel, s->pb_sad, PB_MAX_ROWS};
    const size_t lds = lds_bytes_t16(K);
#define JH_ACT(UMV)                                                                                   \
    {                                                                                                 \
        JHCHK(allow_lds((rows_act_t16_kernel<PRO, UMV>), lds));                                       \
        hipLaunchKernelGGL((rows_act_t16_kernel<PRO, UMV>), dim3(rows), dim3(P16_THREADS), lds, st, rp); \
    }
    if (K <= 8192) JH_ACT(2) else if (K <= 16384) JH_ACT(4) else JH_ACT(8)
#undef JH_ACT
    HIPCHK(hipGetLastError());
    return JH_OK;
}
template <int EPI>
int gemm_t16_launch(jh_session* s, const uint8_t* w, const float* ws, int ntiles, int K, int rows, float* out, int ldc, const float* resid, int ldr,
                    hipStream_t st) {
    if (!w || !ws) return set_err(JH_ERR_INVALID, "reference-order GEMM: the weight has no T16 copy (ensure_strict_operands)");
    constexpr int MT = 8, CW = 4;                      // 4 waves x 2 tiles = 128 weight rows x 8 prompt rows per workgroup
    const int nslices = (ntiles + 2 * CW - 1) / (2 * CW), nrt = (rows + MT - 1) / MT;
    GemmT16Params g{(const i32x4*)w, (const f32x4t*)ws, ntiles, K, rows, (const i32x4*)s->pb_sel, s->pb_sad, PB_MAX_ROWS, out, ldc, resid, ldr, nslices, nrt};
    const size_t lds = lds_bytes_gemm_t16(MT);
    const int grid = ((nslices + 7) / 8) * 8 * nrt;
    if (opt_int("JH_T16_GEMM32", 1)) {                 // the 32x32x16 form (jh_t16.h: one MFMA issue and one scale product per tile pair)
        JHCHK(allow_lds((gemm_t16x_kernel<EPI, MT, CW>), lds));
        hipLaunchKernelGGL((gemm_t16x_kernel<EPI, MT, CW>), dim3(grid), dim3(CW * 64), lds, st, g);
    } else {
        JHCHK(allow_lds((gemm_t16_kernel<EPI, MT, CW>), lds));
        hipLaunchKernelGGL((gemm_t16_kernel<EPI, MT, CW>), dim3(grid), dim3(CW * 64), lds, st, g);
    }
    HIPCHK(hipGetLastError());
    return JH_OK;
}
int prefill_attn_p16_launch(jh_session* s, int rel, int rows, int start_pos, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const int hs = c.head_size, A = c.n_heads * hs, KV = c.n_kv_heads * hs, group = c.n_heads / c.n_kv_heads;
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.qkv = s->pb_qkv;
    p.rope = m->rope;
    p.kv_base = s->kv_slab + (size_t)(rel / s->layers_per_page) * s->n_ctx_alloc * s->page_elems;
    p.page_elems = (long long)s->page_elems;
    p.rel_layer_in_page = rel % s->layers_per_page;
    p.ctx_per_page = s->ctx_per_page;
    p.cpp_shift = -1;
    for (int sh = 0; sh < 30; sh++)
        if ((1 << sh) == s->ctx_per_page) p.cpp_shift = sh;
    p.n_heads = c.n_heads; p.n_kv_heads = c.n_kv_heads; p.head_size = hs; p.kv_head_offset = m->kv_head_offset;
    p.st = s->st; p.scale = m->attention_scale;
    p.outf = s->pb_att;
    p.batch_pos0 = start_pos; p.ldqkv = A + 2 * KV; p.ldo = A;
    // score lines [row tile][kv head][position][8 rows x group] (jh_p16.h "prompt rows in reference order")
    const int ztiles = (rows + P16_ROWS_TILE - 1) / P16_ROWS_TILE, units = ztiles * c.n_kv_heads, nch = P16_ROWS_TILE * group;
    const int nmax = start_pos + rows, stride = s->p16_sc_stride;
    float* sT = s->p16_scores_b;
    float* mx = sT + (size_t)PB_MAX_ROWS * c.n_heads * stride;
    float* sums = mx + (size_t)PB_MAX_ROWS * c.n_heads;
    const dim3 grid_s((nmax + P16_ROWS_TJ - 1) / P16_ROWS_TJ, c.n_kv_heads, ztiles);
    const dim3 grid_e(((size_t)nmax * nch + 1023) / 1024, units);
    const size_t lds_av = lds_bytes_rows_av_p16(hs, group);
    if (hs == 128) hipLaunchKernelGGL((rows_rope_kv_p16_kernel<128>), dim3(rows, c.n_kv_heads), dim3(128), 0, st, p);
    else hipLaunchKernelGGL((rows_rope_kv_p16_kernel<64>), dim3(rows, c.n_kv_heads), dim3(128), 0, st, p);
    HIPCHK(hipGetLastError());
#define JH_P16_ATTNB(HSV, GV)                                                                                                  \
    if (hs == HSV && group == GV) {                                                                                            \
        hipLaunchKernelGGL((rows_scores_p16_kernel<HSV, GV>), grid_s, dim3(256), 0, st, p, rows, sT, stride);                  \
        hipLaunchKernelGGL((rows_max_p16_kernel<GV>), dim3(units), dim3(256), 0, st, rows, start_pos, c.n_kv_heads, (const float*)sT, stride, mx); \
        hipLaunchKernelGGL((rows_exp_p16_kernel<GV>), grid_e, dim3(256), 0, st, rows, start_pos, c.n_kv_heads, sT, stride, (const float*)mx); \
        hipLaunchKernelGGL((rows_sum_p16_kernel<GV>), dim3(units), dim3(64), 0, st, rows, start_pos, c.n_kv_heads, (const float*)sT, stride, sums); \
        JHCHK(allow_lds((rows_av_p16_kernel<HSV, GV>), lds_av));                                                               \
        hipLaunchKernelGGL((rows_av_p16_kernel<HSV, GV>), dim3(units), dim3(P16_ROWS_AV_THREADS), lds_av, st, p, rows, (const float*)sT, stride, (const float*)sums); \
        HIPCHK(hipGetLastError());                                                                                             \
        return JH_OK;                                                                                                          \
    }
    JH_P16_ATTNB(128, 4) JH_P16_ATTNB(128, 8) JH_P16_ATTNB(64, 4) JH_P16_ATTNB(128, 1) JH_P16_ATTNB(128, 2) JH_P16_ATTNB(64, 1) JH_P16_ATTNB(64, 2) JH_P16_ATTNB(64, 8)
#undef JH_P16_ATTNB
    return set_err(JH_ERR_UNSUPPORTED, "attention: head_size must be 64 or 128 and heads/kv_heads in {1,2,4,8}");
}
// the two halves of a layer in reference order: pair sums on the F16 MFMA (gemm_t16_kernel), same chains and bits as the GEMVs
int prefill_p16_operands(jh_session* s) {
    const jh_config& c = s->m->c;
    if (c.weight_dtype == JH_DT_BF16) {
        if (!prefill_bf16r_ok(s))
            return set_err(JH_ERR_UNSUPPORTED, "reference-order prompt chunk: the model's shapes do not fit the BF16 M-row GEMM (rows go one at a time)");
        if (!s->pb_bfr) {
            const int E = c.embedding_length, H = c.hidden_length, A = c.n_heads * c.head_size;
            int kmax = E > H ? E : H;
            if (A > kmax) kmax = A;
            const size_t bytes = bfr_image_floats(PB_MAX_ROWS, kmax) * 4;
            if (hipMalloc((void**)&s->pb_bfr, bytes) != hipSuccess) return set_err(JH_ERR_OOM, "hipMalloc prompt activation image");
            HIPCHK(hipMemsetAsync(s->pb_bfr, 0, bytes, s->stream));   // rows past a ragged last tile are read (never stored): keep them finite
        }
        if (!s->p16_scores_b) {
            const hipError_t e = hipMalloc(&s->p16_scores_b, (size_t)PB_MAX_ROWS * c.n_heads * ((size_t)s->p16_sc_stride + 2) * 4);
            if (e != hipSuccess) return set_err(JH_ERR_OOM, "hipMalloc score rows of a prompt chunk");
        }
        return JH_OK;
    }
    if (!prefill_t16_ok(s))
        return set_err(JH_ERR_UNSUPPORTED, "reference-order prompt chunk: the model's shapes do not fit the T16 GEMM (rows go one at a time)");
    if (!s->pb_sel) {
        const size_t E = c.embedding_length, H = c.hidden_length, A = (size_t)c.n_heads * c.head_size;
        size_t kmax = E > H ? E : H;
        if (A > kmax) kmax = A;
        hipError_t e = hipMalloc((void**)&s->pb_sel, (size_t)PB_MAX_ROWS * (kmax / QB) * 256);
        if (e == hipSuccess) e = hipMalloc((void**)&s->pb_sad, (size_t)(kmax / QB) * PB_MAX_ROWS * 4);
        if (e != hipSuccess) return set_err(JH_ERR_OOM, "hipMalloc prompt selector operands");
    }
    if (!s->p16_scores_b) {
        const hipError_t e = hipMalloc(&s->p16_scores_b, (size_t)PB_MAX_ROWS * c.n_heads * ((size_t)s->p16_sc_stride + 2) * 4);   // score lines + maxima + sums
        if (e != hipSuccess) return set_err(JH_ERR_OOM, "hipMalloc score rows of a prompt chunk");
    }
    return JH_OK;
}
// Rows of a ragged last row tile.  A tile with one valid row costs what a full one costs; that is free while the machine has idle
// workgroup slots (2 per CU: 512), and a whole extra pass where the full tiles fill them exactly -- o / down of the 8B model at 128
// rows: 32 slices x 16 row tiles = 512 workgroups; the 17th tile took down from 279 to 439 us.  When the ragged tile would add
// such a pass to the E-row GEMMs and holds few rows, those rows go through the decode GEMVs instead (the same chains, hence the
// same bits; each does its own prologue from the F32 row): ~35 us per row and layer against ~180 us for the pass.  Measured on the
// 8B model (ms per prompt; rows = 129 / 130 / 131 / 132 / 134): all rows in the GEMM 34.7 / 34.9 / 35.0 / 35.1 / 35.2, tail rows
// 29.2 / 30.3 / 31.3 / 32.4 / 34.6.  Option JH_T16_TAIL_ROWS = most rows to divert (default 4; 0 = never).
static int t16_tail_rows(const jh_session* s, int rows) {
    const int r = rows % 8, lim = opt_int("JH_T16_TAIL_ROWS", 4);
    if (r < 1 || r > lim) return 0;
    const int nrt = (rows + 7) / 8, nsl = (s->m->c.embedding_length / 16 + 7) / 8, slots = 2 * g_cu_count;
    const bool extra_pass = (nsl * nrt + slots - 1) / slots > (nsl * (nrt - 1) + slots - 1) / slots;
    return (extra_pass || opt_int("JH_T16_TAIL_FORCE", 0)) ? r : 0;   // (_FORCE: tests reach the path on small shapes)
}
template <int PRO, int EPI>
static int tail_gemv_p16(jh_session* s, const JWeight& W, int K, int nrows, const float* x, const float* nw, float eps, float* out, const float* resid,
                         hipStream_t st) {
    GemvParams p;
    memset(&p, 0, sizeof(p));
    p.w = (const uint8_t*)W.data; p.ws = W.scales; p.nrows = nrows; p.out = out;
    p.K = K; p.ldb = K / 2; p.ldbf = K / QB;
    p.x = x; p.nw = nw; p.eps = eps; p.resid = resid;
    JHCHK(use_p16t(p, W));
    return launch_gemv_i8q4_p16<PRO, EPI>(p, s->p16_depth, st);
}
int prefill_attn_half_p16(jh_session* s, int li, int rows, int start_pos, float* out, const float* resid, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const int E = c.embedding_length, hs = c.head_size, A = c.n_heads * hs, KV = c.n_kv_heads * hs, Q = A + 2 * KV;
    JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    JWeight& F = m->qkv[(size_t)li];
    const float* n1 = (const float*)W[JH_W_NORM1].data;
    const int tail = t16_tail_rows(s, rows), rg = rows - tail;
    JHCHK(prefill_weights_set(s, li));
    if (rg) {
        JHCHK((rows_act_t16_launch<PRO_RMS_Q8>(s, s->pb_x, E, n1, c.rms_eps, E, rg, st)));
        JHCHK((gemm_t16_launch<EPI_STORE>(s, F.t16, F.t16_scales, Q / 16, E, rg, s->pb_qkv, Q, nullptr, 0, st)));
    }
    for (int r = rg; r < rows; r++)
        JHCHK((tail_gemv_p16<PRO_RMS_Q8, EPI_STORE>(s, F, E, Q, s->pb_x + (size_t)r * E, n1, c.rms_eps, s->pb_qkv + (size_t)r * Q, nullptr, st)));
    JHCHK(prefill_attn_p16_launch(s, li - c.layer_start, rows, start_pos, st));
    if (rg) {
        JHCHK((rows_act_t16_launch<PRO_QUANT_Q8>(s, s->pb_att, A, nullptr, 0.f, A, rg, st)));
        if (resid) JHCHK((gemm_t16_launch<EPI_RESID>(s, W[JH_W_O].t16, W[JH_W_O].t16_scales, E / 16, A, rg, out, E, resid, E, st)));
        else JHCHK((gemm_t16_launch<EPI_STORE>(s, W[JH_W_O].t16, W[JH_W_O].t16_scales, E / 16, A, rg, out, E, nullptr, 0, st)));
    }
    for (int r = rg; r < rows; r++) {
        const float* xr = s->pb_att + (size_t)r * A;
        if (resid) JHCHK((tail_gemv_p16<PRO_QUANT_Q8, EPI_RESID>(s, W[JH_W_O], A, E, xr, nullptr, 0.f, out + (size_t)r * E, resid + (size_t)r * E, st)));
        else JHCHK((tail_gemv_p16<PRO_QUANT_Q8, EPI_STORE>(s, W[JH_W_O], A, E, xr, nullptr, 0.f, out + (size_t)r * E, nullptr, st)));
    }
    return JH_OK;
}
int prefill_ffn_half_p16(jh_session* s, int li, int rows, const float* x1, float* out, const float* resid, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const int E = c.embedding_length, H = c.hidden_length;
    JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    const JWeight& GU = m->gateup[(size_t)li];
    const float* n2 = (const float*)W[JH_W_NORM2].data;
    const int tail = t16_tail_rows(s, rows), rg = rows - tail;
    if (rg) {
        JHCHK((rows_act_t16_launch<PRO_RMS_Q8>(s, x1, E, n2, c.rms_eps, E, rg, st)));
        JHCHK((gemm_t16_launch<EPI_SILU_MUL>(s, GU.t16, GU.t16_scales, H / 8, E, rg, s->pb_g, H, nullptr, 0, st)));
    }
    for (int r = rg; r < rows; r++) {   // the decode gate|up GEMV on the same T16 copy (layers.hip)
        GemvParams p;
        memset(&p, 0, sizeof(p));
        if (!GU.t16) return set_err(JH_ERR_INVALID, "reference-order prompt rows: gate|up has no T16 copy (ensure_strict_operands)");
        p.w = GU.t16; p.ws = GU.t16_scales; p.nrows = H;
        p.K = E; p.ldb = E / 2; p.ldbf = E / QB;
        p.x = x1 + (size_t)r * E; p.nw = n2; p.eps = c.rms_eps;
        p.out = s->pb_g + (size_t)r * H;
        JHCHK((launch_gemv_t16<PRO_RMS_Q8, EPI_SILU_MUL>(p, st)));
    }
    if (rg) {
        JHCHK((rows_act_t16_launch<PRO_QUANT_Q8>(s, s->pb_g, H, nullptr, 0.f, H, rg, st)));
        if (resid) JHCHK((gemm_t16_launch<EPI_RESID>(s, W[JH_W_DOWN].t16, W[JH_W_DOWN].t16_scales, E / 16, H, rg, out, E, resid, E, st)));
        else JHCHK((gemm_t16_launch<EPI_STORE>(s, W[JH_W_DOWN].t16, W[JH_W_DOWN].t16_scales, E / 16, H, rg, out, E, nullptr, 0, st)));
    }
    for (int r = rg; r < rows; r++) {
        const float* xr = s->pb_g + (size_t)r * H;
        if (resid) JHCHK((tail_gemv_p16<PRO_QUANT_Q8, EPI_RESID>(s, W[JH_W_DOWN], H, E, xr, nullptr, 0.f, out + (size_t)r * E, resid + (size_t)r * E, st)));
        else JHCHK((tail_gemv_p16<PRO_QUANT_Q8, EPI_STORE>(s, W[JH_W_DOWN], H, E, xr, nullptr, 0.f, out + (size_t)r * E, nullptr, st)));
    }
    return JH_OK;
}
// ---- the same two halves for a dense BF16 model in reference order (jh_bf16r.h): activation image per projection input, M-row chains
template <int PRO>
int rows_act_bf16r_launch(jh_session* s, const float* x, int ldx, const float* x2, int ldx2, const float* nw, float eps, int K, int rows, hipStream_t st) {
    RowsBfrParams rp{x, ldx, x2, ldx2, nw, eps, K, s->pb_bfr};
    hipLaunchKernelGGL((rows_act_bf16r_kernel<PRO>), dim3(rows), dim3(256), 0, st, rp);
    HIPCHK(hipGetLastError());
    return JH_OK;
}
template <int EPI>
int gemm_bf16r_launch(jh_session* s, const JWeight& W, int N, int K, int rows, float* out, int ldc, const float* resid, int ldr, hipStream_t st) {
    if (!W.p16t) return set_err(JH_ERR_INVALID, "reference-order GEMM: the weight has no BF16T copy (ensure_strict_operands)");
    const int nslices = (N + 63) / 64, nrt = (rows + BFR_MR - 1) / BFR_MR;
    GemmBfrParams g{W.p16t, (int)bf16t_row_bytes(K), N, K, rows, s->pb_bfr, out, ldc, resid, ldr, nslices, nrt};
    const size_t lds = lds_bytes_gemm_bf16r();
    JHCHK(allow_lds((gemm_bf16r_kernel<EPI>), lds));
    const int grid = ((nslices + 7) / 8) * 8 * nrt;
    hipLaunchKernelGGL((gemm_bf16r_kernel<EPI>), dim3(grid), dim3(BFR_WAVES * 64), lds, st, g);
    HIPCHK(hipGetLastError());
    return JH_OK;
}
int prefill_attn_half_bf16r(jh_session* s, int li, int rows, int start_pos, float* out, const float* resid, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const int E = c.embedding_length, hs = c.head_size, A = c.n_heads * hs, KV = c.n_kv_heads * hs;
    JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    JHCHK(prefill_weights_set(s, li));
    JHCHK((rows_act_bf16r_launch<PROB_RMS_BF16>(s, s->pb_x, E, nullptr, 0, (const float*)W[JH_W_NORM1].data, c.rms_eps, E, rows, st)));
    JHCHK((gemm_bf16r_launch<EPI_STORE>(s, m->qkv[(size_t)li], A + 2 * KV, E, rows, s->pb_qkv, A + 2 * KV, nullptr, 0, st)));
    JHCHK(prefill_attn_p16_launch(s, li - c.layer_start, rows, start_pos, st));
    JHCHK((rows_act_bf16r_launch<PROB_QUANT_BF16>(s, s->pb_att, A, nullptr, 0, nullptr, 0.f, A, rows, st)));
    if (resid) return gemm_bf16r_launch<EPI_RESID>(s, W[JH_W_O], E, A, rows, out, E, resid, E, st);
    return gemm_bf16r_launch<EPI_STORE>(s, W[JH_W_O], E, A, rows, out, E, nullptr, 0, st);
}
int prefill_ffn_half_bf16r(jh_session* s, int li, int rows, const float* x1, float* out, const float* resid, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const int E = c.embedding_length, H = c.hidden_length;
    JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    JHCHK((rows_act_bf16r_launch<PROB_RMS_BF16>(s, x1, E, nullptr, 0, (const float*)W[JH_W_NORM2].data, c.rms_eps, E, rows, st)));
    JHCHK((gemm_bf16r_launch<EPI_STORE>(s, W[JH_W_GATE], H, E, rows, s->pb_g, H, nullptr, 0, st)));
    JHCHK((gemm_bf16r_launch<EPI_STORE>(s, W[JH_W_UP], H, E, rows, s->pb_u, H, nullptr, 0, st)));
    JHCHK((rows_act_bf16r_launch<PROB_SILU_BF16>(s, s->pb_g, H, s->pb_u, H, nullptr, 0.f, H, rows, st)));   // silu(gate) * up, rounded to BF16
    if (resid) return gemm_bf16r_launch<EPI_RESID>(s, W[JH_W_DOWN], E, H, rows, out, E, resid, E, st);
    return gemm_bf16r_launch<EPI_STORE>(s, W[JH_W_DOWN], E, H, rows, out, E, nullptr, 0, st);
}
int prefill_layers_p16(jh_session* s, int rows, int start_pos, hipStream_t st) {
    const jh_config& c = s->m->c;
    JHCHK(prefill_p16_operands(s));
    if (c.weight_dtype == JH_DT_BF16) {
        for (int li = c.layer_start; li < c.layer_end; li++) {
            JHCHK(prefill_attn_half_bf16r(s, li, rows, start_pos, s->pb_x1, s->pb_x, st));
            JHCHK(prefill_ffn_half_bf16r(s, li, rows, s->pb_x1, s->pb_x, s->pb_x1, st));
            JHCHK(trace_sync("prefill layer (reference order, BF16)", st));
        }
        return JH_OK;
    }
    for (int li = c.layer_start; li < c.layer_end; li++) {
        JHCHK(prefill_attn_half_p16(s, li, rows, start_pos, s->pb_x1, s->pb_x, st));
        JHCHK(prefill_ffn_half_p16(s, li, rows, s->pb_x1, s->pb_x, s->pb_x1, st));
        JHCHK(trace_sync("prefill layer (reference order, MFMA)", st));
    }
    return JH_OK;
}
// One chunk of `rows` prompt rows at positions [start_pos, start_pos+rows) through this shard's layers.
int prefill_chunk(jh_session* s, const int32_t* tokens, const float* x_in, bool x_in_dev, int rows, int start_pos,
                  float* x_out, bool x_out_dev, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    JHCHK(prefill_alloc(s));
    const bool p16 = s->strict != 0;                      // reference order: prefill_batch_ok() admitted the session via prefill_p16_ok()
    if (!p16) JHCHK(ensure_all_tiled(s, st));
    const int E = c.embedding_length;
    if (tokens) {
        const JWeight& emb = m->global_w[JH_W_EMBED];
        HIPCHK(hipMemcpyAsync(s->pb_tok, tokens, (size_t)rows * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(embed_rows_kernel, dim3(rows), dim3(256), 0, st, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                           (const int*)s->pb_tok, E, s->pb_x);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemcpyAsync(s->pb_x, x_in, (size_t)rows * E * 4, x_in_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, st, s->pb_start, start_pos);
    HIPCHK(hipGetLastError());
    // ~15 launches per layer, many of them 5 us kernels: replay a captured graph of the layer loop.  The graph depends on
    // the chunk's row count (grids) and on an LDS bound for the score rows, not on the position: every full 256-row chunk
    // of a long prompt replays the same graph
    int bound = 1024;
    while (bound < start_pos + rows) bound *= 2;
    const bool attn_mfma = prefill_attn_mfma(s, start_pos, rows);
    if (!attn_mfma && !prefill_chunk_fits(s, 0, bound)) bound = start_pos + rows;
    const int use_graph = opt_int("JH_PREFILL_GRAPH", 1);
    if (p16) {
        JHCHK(prefill_layers_p16(s, rows, start_pos, st));   // positions are launch arguments here: launched directly, no graph
    } else if (use_graph && !opt_int("JH_TRACE", 0)) {
        drop_stale_graphs(s);
        const uint64_t key = (uint64_t)rows | ((uint64_t)(attn_mfma ? 1 : 0) << 16) | ((uint64_t)bound << 32);
        auto it = s->pb_graphs.find(key);
        if (it == s->pb_graphs.end()) {
            std::lock_guard<std::mutex> cap(g_capture_mu);
            HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int rc = prefill_layers(s, rows, bound, attn_mfma, st);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(st, &g);
            if (rc != JH_OK) { if (g) hipGraphDestroy(g); return rc; }
            if (e != hipSuccess) return set_err(JH_ERR_HIP, std::string("hipStreamEndCapture (prefill): ") + hipGetErrorString(e));
            hipGraphExec_t ex = nullptr;
            HIPCHK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
            s->pb_graph_src.push_back(g);
            it = s->pb_graphs.emplace(key, ex).first;
        }
        HIPCHK(hipGraphLaunch(it->second, st));
    } else {
        JHCHK(prefill_layers(s, rows, bound, attn_mfma, st));
    }
    // the chunk's last row is the session's current row (what sample() / the next shard's hand-off reads)
    HIPCHK(hipMemcpyAsync(s->x, s->pb_x + (size_t)(rows - 1) * E, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, s->st, start_pos + rows - 1, tokens ? tokens[rows - 1] : 0, 0);
    if (x_out)
        HIPCHK(hipMemcpyAsync(x_out, s->pb_x, (size_t)rows * E * 4, x_out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    return JH_OK;
}

