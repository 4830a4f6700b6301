// layers.hip -- part of libjlamahip.so (C ABI: include/jlama_hip.h).  One decoded row through the layers: the five launches of a TransformerBlock, LM head, argmax / next row.
#include "jh_host.h"
#include "jh_launch.h"

int attn_launch(jh_session* s, int rel, hipStream_t st, bool tap, long long* dbg) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    AttnParams p;
    memset(&p, 0, sizeof(p));
    const int lp = rel / s->layers_per_page;
    p.qkv = s->qkv;
    p.rope = m->rope;
    p.kv_base = s->kv_slab + (size_t)lp * s->n_ctx_alloc * s->page_elems;   // first context page of this layer page
    p.page_elems = (long long)s->page_elems;
    p.rel_layer_in_page = rel % s->layers_per_page;
    p.ctx_per_page = s->ctx_per_page;
    p.cpp_shift = -1;
    for (int sh = 0; sh < 30; sh++)
        if ((1 << sh) == s->ctx_per_page) p.cpp_shift = sh;
    p.n_heads = c.n_heads;
    p.n_kv_heads = c.n_kv_heads;
    p.head_size = c.head_size;
    p.kv_head_offset = m->kv_head_offset;
    p.st = s->st;
    p.scale = m->attention_scale;
    p.part_o = s->part_o;
    p.part_ml = s->part_ml;
    p.part_stride = s->part_stride;
    p.direct_max = s->direct_max;
    p.direct_chunk = s->direct_chunk;
    p.counters = s->counters;
    const bool long_v = s->attn_variant == 2;
    p.max_splits = long_v ? s->long_splits : s->max_splits;
    p.mid_splits = long_v ? s->mid_splits : 0;
    p.mid_max = long_v ? s->mid_max : 0;
    p.outf = s->attf;
    p.tap_q = tap ? s->tapq : nullptr;
    p.dbg = dbg;
    if (s->strict) {
        // reference order in two launches (jh_p16.h): scores of every position slice, then softmax + the value chains
        const int group = c.n_heads / c.n_kv_heads, hs = c.head_size;
        const size_t lds_av = lds_bytes_attn_p16(s->max_ctx);
        if (lds_av > 158 * 1024) return set_err(JH_ERR_UNSUPPORTED, "reference-order attention: the score row of max_ctx positions must fit in LDS");
        const dim3 grid_s(s->p16_att_splits, c.n_kv_heads), grid_v(hs / 32, c.n_heads);
        const int ru = p16_av_rows(s->max_ctx), ru2 = p16_av2_rows(s->max_ctx);
        // contexts of up to ~1 k positions: scores, softmax and value chains in ONE launch (attn_p16_fused_kernel); beyond that the K
        // rows of a head are too many for one workgroup to ingest and the scores stay spread over the chip (two launches)
        const int fused_max = opt_int("JH_P16_ATT_FUSED", 1024);
        const size_t lds_f = lds_bytes_attn_p16_fused(s->max_ctx, hs);
        const bool kv_small = (size_t)s->n_ctx_alloc * s->page_elems * 4 < ((size_t)4 << 30);   // the one-launch kernel addresses rows with 32-bit byte offsets
        if (s->max_ctx <= fused_max && lds_f <= 158 * 1024 && (hs == 128 || hs == 64) && c.n_heads % c.n_kv_heads == 0 && kv_small) {
            const dim3 grid_f(c.n_heads * (hs / 32));
#define JH_P16_FUSED(HSV, RV)                                                                                                   \
    if (hs == HSV && ru == RV) {                                                                                               \
        if (p.cpp_shift >= 0) {                                                                                                \
            JHCHK(allow_lds((attn_p16_fused_kernel<HSV, RV, true>), lds_f));                                                   \
            hipLaunchKernelGGL((attn_p16_fused_kernel<HSV, RV, true>), grid_f, dim3(P16_FUSED_THREADS), lds_f, st, p);           \
        } else {                                                                                                               \
            JHCHK(allow_lds((attn_p16_fused_kernel<HSV, RV, false>), lds_f));                                                  \
            hipLaunchKernelGGL((attn_p16_fused_kernel<HSV, RV, false>), grid_f, dim3(P16_FUSED_THREADS), lds_f, st, p);          \
        }                                                                                                                      \
        HIPCHK(hipGetLastError());                                                                                             \
        return JH_OK;                                                                                                          \
    }
            JH_P16_FUSED(128, 2) JH_P16_FUSED(128, 4) JH_P16_FUSED(128, 8) JH_P16_FUSED(128, 16)
            JH_P16_FUSED(64, 2) JH_P16_FUSED(64, 4) JH_P16_FUSED(64, 8) JH_P16_FUSED(64, 16)
#undef JH_P16_FUSED
        }
#define JH_P16_AV(HSV, RV)                                                                                                      \
    if (hs == HSV && ru2 == RV) {                                                                                               \
        JHCHK(allow_lds((attn_p16_av_kernel<HSV, RV>), lds_av));                                                               \
        hipLaunchKernelGGL((attn_p16_av_kernel<HSV, RV>), grid_v, dim3(P16_ATT_THREADS), lds_av, st, p, (const float*)s->p16_scores, s->p16_sc_stride, \
                           p16_av2_wcap(s->max_ctx), opt_int("JH_P16_AV_SEQ_MIN", P16_AV_SEQ_MIN));                            \
    }
#define JH_P16_ATTN(HSV, GV)                                                                                                   \
    if (hs == HSV && group == GV) {                                                                                            \
        hipLaunchKernelGGL((attn_p16_scores_kernel<HSV, GV>), grid_s, dim3(P16_ATT_THREADS), 0, st, p, s->p16_scores, s->p16_sc_stride); \
        HIPCHK(hipGetLastError());                                                                                             \
        JH_P16_AV(HSV, 2) JH_P16_AV(HSV, 4) JH_P16_AV(HSV, 8)                                                                   \
        HIPCHK(hipGetLastError());                                                                                             \
        return JH_OK;                                                                                                          \
    }
        JH_P16_ATTN(128, 4) JH_P16_ATTN(128, 8) JH_P16_ATTN(64, 4) JH_P16_ATTN(128, 1) JH_P16_ATTN(128, 2) JH_P16_ATTN(64, 1) JH_P16_ATTN(64, 2) JH_P16_ATTN(64, 8)
#undef JH_P16_ATTN
#undef JH_P16_AV
        return set_err(JH_ERR_UNSUPPORTED, "attention: head_size must be 64 or 128 and heads/kv_heads in {1,2,4,8}");
    }
    const int group = c.n_heads / c.n_kv_heads, hs = c.head_size;
    const int most = s->long_splits > s->max_splits ? s->long_splits : s->max_splits;
    const int sc_cap = s->chunk_cap > 2 * most ? s->chunk_cap : 2 * most;
    const size_t lds = ((size_t)group * hs + 2 * hs + (size_t)(ATT_THREADS * 4) * group + 2 * group + 4 + (size_t)group * sc_cap) * 4;
    const int gx = p.max_splits > 4 ? p.max_splits : 4;
    dim3 grid(gx, c.n_kv_heads), block(ATT_THREADS);
#define JH_ATTN(HSV, GV)                                                                   \
    if (hs == HSV && group == GV) {                                                        \
        if (s->attn_variant == 1) {                                                        \
            JHCHK(allow_lds((attn_decode_kernel<HSV, GV, 2>), lds));                       \
            hipLaunchKernelGGL((attn_decode_kernel<HSV, GV, 2>), grid, block, lds, st, p); \
        } else {                                                                           \
            JHCHK(allow_lds((attn_decode_kernel<HSV, GV, 8>), lds));                       \
            hipLaunchKernelGGL((attn_decode_kernel<HSV, GV, 8>), grid, block, lds, st, p); \
        }                                                                                  \
        HIPCHK(hipGetLastError());                                                         \
        return JH_OK;                                                                      \
    }
    JH_ATTN(128, 4) JH_ATTN(128, 8) JH_ATTN(64, 4) JH_ATTN(128, 1) JH_ATTN(128, 2) JH_ATTN(64, 1) JH_ATTN(64, 2) JH_ATTN(64, 8)
#undef JH_ATTN
    return set_err(JH_ERR_UNSUPPORTED, "attention: head_size must be 64 or 128 and heads/kv_heads in {1,2,4,8}");
}

int tap_copy(jh_session* s, int which, const float* src, int n, hipStream_t st) {
    if (!s->taps[which] || s->tap_len[which] < n) {
        if (s->taps[which]) HIPCHK(hipFree(s->taps[which]));
        HIPCHK(hipMalloc(&s->taps[which], (size_t)n * 4));
    }
    s->tap_len[which] = n;
    HIPCHK(hipMemcpyAsync(s->taps[which], src, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    return JH_OK;
}

// One TransformerBlock.forward (core/model/TransformerBlock.java:158-215) for the row described by s->st:
// 5 launches -- qkv(+rmsnorm+q8) | attention(+rope+kv write+q8) | o-proj(+residual) | gate/up(+rmsnorm+q8,+silu*up+q8) | down(+residual)
// in two halves, split where tensor-parallel shards synchronise (tensorReducer: CausalSelfAttention.java:378,
// MLPBlock.java:160).  resid == nullptr => the projection's partial result is stored WITHOUT the residual.
// o-proj / down of a tensor-parallel shard inside its token graph: the GEMV stores its partial row into every shard's slot and
// raises its workgroup flags itself (EPI_TP)
int tp_push_gemv(jh_session* s, GemvParams& p, const LaunchCfg& cfg, hipStream_t st) {
    TPPush* t = s->tp_push;
    p.tp_dst = t->dst; p.tp_flags = t->flags; p.tp_seq = t->seq; p.tp_n = t->n; p.tp_li = t->li; p.tp_L = t->L;
    if (s->strict) JHCHK((launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_TP>(p, s->p16_depth, st)));
    else JHCHK((launch_gemv_i8q4<PRO_QUANT_Q8, EPI_TP>(p, cfg, st)));
    if (g_last_gemv_grid > TP_MAX_FLAGS) return set_err(JH_ERR_UNSUPPORTED, "tensor-parallel push: the GEMV has more workgroups than flag words");
    t->grid = g_last_gemv_grid;
    return JH_OK;
}
int layer_attn_launch(jh_session* s, int li, hipStream_t st, bool tap, int pos_for_tap, float* out, const float* resid) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const int rel = li - c.layer_start;
    const JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    const int E = c.embedding_length, hs = c.head_size;
    const int A = c.n_heads * hs, KV = c.n_kv_heads * hs;
    if (tap) JHCHK(tap_copy(s, JH_TAP_INPUT_EMB, s->x, E, st));
    {   // q,k,v projections (CausalSelfAttention.java:161-171) with fused preAttentionNorm + maybeQuantize
        GemvParams p;
        memset(&p, 0, sizeof(p));
        const JWeight& F = m->qkv[(size_t)li];
        // every slot this half dereferences on the device (a partial checkpoint or a wrong layer range must be an error
        // code, not a GPU fault)
        JHCHK(refuse_order_free(s, "attention half"));
        if (!w_present(F) || !w_present(W[JH_W_Q]) || !w_present(W[JH_W_K]) || !w_present(W[JH_W_V]) || !w_present(W[JH_W_O]) || !W[JH_W_NORM1].data)
            return set_err(JH_ERR_INVALID, "layer " + std::to_string(li) + ": q/k/v/o/input_layernorm weights not set");
        p.w = (const uint8_t*)F.data; p.ws = F.scales; p.nrows = A + 2 * KV; p.out = s->qkv;
        p.K = E; p.ldb = E / 2; p.ldbf = E / QB;
        p.x = s->x; p.nw = (const float*)W[JH_W_NORM1].data; p.eps = c.rms_eps;
        if (c.weight_dtype == JH_DT_BF16 && s->strict) { JHCHK(use_p16t(p, F)); JHCHK((launch_gemv_bf16r<PROB_RMS_BF16, EPI_STORE, false>(p, nullptr, st))); }
        else if (c.weight_dtype == JH_DT_BF16) { p.ldb = E * 2; JHCHK((launch_gemv_bf16<PROB_RMS_BF16, EPI_STORE, false>(p, g_cu_count * 4, nullptr, st))); }
        else if (s->strict) { JHCHK(use_p16t(p, F)); JHCHK((launch_gemv_i8q4_p16<PRO_RMS_Q8, EPI_STORE>(p, s->p16_depth, st))); }
        else JHCHK((launch_gemv_i8q4<PRO_RMS_Q8, EPI_STORE>(p, s->cfg_qkv, st)));
        JHCHK(trace_sync("qkv", st));
    }
    if (tap) {
        JHCHK(tap_copy(s, JH_TAP_QUERY, s->qkv, A, st));
        JHCHK(tap_copy(s, JH_TAP_KEY, s->qkv + A, KV, st));
        JHCHK(tap_copy(s, JH_TAP_VALUE, s->qkv + A + KV, KV, st));
    }
    JHCHK(attn_launch(s, rel, st, tap));
    JHCHK(trace_sync("attn", st));
    if (tap) {
        JHCHK(tap_copy(s, JH_TAP_QUERY_ROPE, s->tapq, A, st));
        const int lp = rel / s->layers_per_page, cp = pos_for_tap / s->ctx_per_page, rc = pos_for_tap % s->ctx_per_page;
        const float* krow = s->pages_host[(size_t)lp * s->n_ctx_pages + cp] +
                            ((size_t)((rel % s->layers_per_page) * 2 + 0) * s->ctx_per_page + rc) * KV;
        JHCHK(tap_copy(s, JH_TAP_KEY_ROPE, krow, KV, st));
    }
    {   // output projection (:365-376) + residual (TransformerBlock.java:185)
        GemvParams p;
        memset(&p, 0, sizeof(p));
        p.w = (const uint8_t*)W[JH_W_O].data; p.ws = W[JH_W_O].scales; p.nrows = E; p.out = out;
        p.K = A; p.ldb = A / 2; p.ldbf = A / QB;
        p.x = s->attf; p.resid = resid;   // maybeQuantize(valueBatch) (:364) happens in the prologue
        if (c.weight_dtype == JH_DT_BF16 && s->strict) {
            JHCHK(use_p16t(p, W[JH_W_O]));
            if (resid) JHCHK((launch_gemv_bf16r<PROB_QUANT_BF16, EPI_RESID, false>(p, nullptr, st)));
            else JHCHK((launch_gemv_bf16r<PROB_QUANT_BF16, EPI_STORE, false>(p, nullptr, st)));
        } else if (c.weight_dtype == JH_DT_BF16) {
            p.ldb = A * 2;
            if (resid) JHCHK((launch_gemv_bf16<PROB_QUANT_BF16, EPI_RESID, false>(p, g_cu_count * 4, nullptr, st)));
            else JHCHK((launch_gemv_bf16<PROB_QUANT_BF16, EPI_STORE, false>(p, g_cu_count * 4, nullptr, st)));
        } else if (!resid && s->tp_push) {
            if (s->strict) JHCHK(use_p16t(p, W[JH_W_O]));
            JHCHK(tp_push_gemv(s, p, s->cfg_o, st));
        } else if (s->strict) {
            JHCHK(use_p16t(p, W[JH_W_O]));
            if (resid) JHCHK((launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_RESID>(p, s->p16_depth, st)));
            else JHCHK((launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_STORE>(p, s->p16_depth, st)));
        } else if (!resid) {
            JHCHK((launch_gemv_i8q4<PRO_QUANT_Q8, EPI_STORE>(p, s->cfg_o, st)));
        } else {
            JHCHK((launch_gemv_i8q4<PRO_QUANT_Q8, EPI_RESID>(p, s->cfg_o, st)));
        }
        JHCHK(trace_sync("oproj", st));
    }
    if (tap) {
        JHCHK(tap_copy(s, JH_TAP_AFTER_ATTENTION, s->attf, A, st));   // written by attention (ticket mode) or by the o-proj prologue
        JHCHK(tap_copy(s, 8, out, E, st));
    }
    return JH_OK;
}
// feed-forward half: reads s->x1, writes `out` (+ resid)
int layer_ffn_launch(jh_session* s, int li, hipStream_t st, bool tap, float* out, const float* resid) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    const int E = c.embedding_length, H = c.hidden_length;
    JHCHK(refuse_order_free(s, "feed-forward half"));
    if (!w_present(W[JH_W_GATE]) || !w_present(W[JH_W_UP]) || !w_present(W[JH_W_DOWN]) || !W[JH_W_NORM2].data)
        return set_err(JH_ERR_INVALID, "layer " + std::to_string(li) + ": gate/up/down/post_attention_layernorm weights not set");
    {   // gate/up (MLPBlock.java:117-142) with fused preFFNorm + maybeQuantize, SiLU*up + maybeQuantize
        GemvParams p;
        memset(&p, 0, sizeof(p));
        p.w = (const uint8_t*)W[JH_W_GATE].data; p.ws = W[JH_W_GATE].scales; p.nrows = H;
        p.w2 = (const uint8_t*)W[JH_W_UP].data; p.ws2 = W[JH_W_UP].scales;
        p.K = E; p.ldb = E / 2; p.ldbf = E / QB;
        p.x = s->x1; p.nw = (const float*)W[JH_W_NORM2].data; p.eps = c.rms_eps;
        p.out = s->hf;   // silu(gate)*up, F32; the down projection's prologue quantizes it (MLPBlock.java:144)
        if (c.weight_dtype == JH_DT_BF16 && s->strict) {
            JHCHK(use_p16t(p, W[JH_W_GATE]));
            if (!W[JH_W_UP].p16t) return set_err(JH_ERR_INVALID, "reference-order GEMV: up projection has no BF16T copy");
            p.w2 = W[JH_W_UP].p16t;
            JHCHK((launch_gemv_bf16r<PROB_RMS_BF16, EPI_SILU_MUL, false>(p, nullptr, st)));
        }
        else if (c.weight_dtype == JH_DT_BF16) { p.ldb = E * 2; JHCHK((launch_gemv_bf16<PROB_RMS_BF16, EPI_SILU_MUL, false>(p, g_cu_count * 4, nullptr, st))); }
        else if (t16_gateup_ok(m, li) && (s->strict || fast_gateup_t16(m))) {
            JHCHK(ensure_gateup_t16(m, li, st));   // (already there unless a weight was just replaced; never inside a capture: ensure_strict_operands)
            p.w = m->gateup[(size_t)li].t16; p.ws = m->gateup[(size_t)li].t16_scales; p.w2 = nullptr; p.ws2 = nullptr;
            JHCHK((launch_gemv_t16<PRO_RMS_Q8, EPI_SILU_MUL>(p, st)));
        }
        else if (s->strict) {
            JHCHK(use_p16t(p, W[JH_W_GATE]));
            if (!W[JH_W_UP].p16t) return set_err(JH_ERR_INVALID, "reference-order GEMV: up projection has no P16T copy");
            p.w2 = W[JH_W_UP].p16t;
            JHCHK((launch_gemv_i8q4_p16<PRO_RMS_Q8, EPI_SILU_MUL>(p, s->p16_depth, st)));
        }
        else JHCHK((launch_gemv_i8q4<PRO_RMS_Q8, EPI_SILU_MUL>(p, s->cfg_gateup, st)));
        JHCHK(trace_sync("gateup", st));
    }
    if (tap) JHCHK(tap_copy(s, 10, s->hf, H, st));
    {   // down projection (:147-158) + residual (TransformerBlock.java:203)
        GemvParams p;
        memset(&p, 0, sizeof(p));
        p.w = (const uint8_t*)W[JH_W_DOWN].data; p.ws = W[JH_W_DOWN].scales; p.nrows = E; p.out = out;
        p.K = H; p.ldb = H / 2; p.ldbf = H / QB;
        p.x = s->hf; p.resid = resid;
        if (c.weight_dtype == JH_DT_BF16 && s->strict) {
            JHCHK(use_p16t(p, W[JH_W_DOWN]));
            if (resid) JHCHK((launch_gemv_bf16r<PROB_QUANT_BF16, EPI_RESID, false>(p, nullptr, st)));
            else JHCHK((launch_gemv_bf16r<PROB_QUANT_BF16, EPI_STORE, false>(p, nullptr, st)));
        } else if (c.weight_dtype == JH_DT_BF16) {
            p.ldb = H * 2;
            if (resid) JHCHK((launch_gemv_bf16<PROB_QUANT_BF16, EPI_RESID, false>(p, g_cu_count * 4, nullptr, st)));
            else JHCHK((launch_gemv_bf16<PROB_QUANT_BF16, EPI_STORE, false>(p, g_cu_count * 4, nullptr, st)));
        } else if (!resid && s->tp_push) {
            if (s->strict) JHCHK(use_p16t(p, W[JH_W_DOWN]));
            JHCHK(tp_push_gemv(s, p, s->cfg_down, st));
        } else if (s->strict) {
            JHCHK(use_p16t(p, W[JH_W_DOWN]));
            if (resid) JHCHK((launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_RESID>(p, s->p16_depth, st)));
            else JHCHK((launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_STORE>(p, s->p16_depth, st)));
        } else if (resid) {
            JHCHK((launch_gemv_i8q4<PRO_QUANT_Q8, EPI_RESID>(p, s->cfg_down, st)));
        } else {
            JHCHK((launch_gemv_i8q4<PRO_QUANT_Q8, EPI_STORE>(p, s->cfg_down, st)));
        }
        JHCHK(trace_sync("down", st));
    }
    return JH_OK;
}
int layer_launch(jh_session* s, int li, hipStream_t st, bool tap, int pos_for_tap) {
    JHCHK(layer_attn_launch(s, li, st, tap, pos_for_tap, s->x1, s->x));
    JHCHK(layer_ffn_launch(s, li, st, tap, s->x, s->x1));
    if (tap) JHCHK(tap_copy(s, JH_TAP_POST_FF_RES, s->x, s->m->c.embedding_length, st));
    return JH_OK;
}

int layers_launch(jh_session* s, hipStream_t st, int pos_for_tap) {
    const jh_config& c = s->m->c;
    for (int li = c.layer_start; li < c.layer_end; li++)
        JHCHK(layer_launch(s, li, st, s->tap_layer == li, pos_for_tap));
    return JH_OK;
}

const JWeight* lm_head_weight(jh_model* m) {
    return m->global_w[JH_W_LMHEAD].data ? &m->global_w[JH_W_LMHEAD] : &m->global_w[JH_W_EMBED];
}

// AbstractModel.sample's device part (core/model/AbstractModel.java:443-469): final RMSNorm -> F32xQ4 LM head -> argmax partials
int lmhead_launch(jh_session* s, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const JWeight* w = lm_head_weight(m);
    if (!w->data || !m->global_w[JH_W_FINALNORM].data) return set_err(JH_ERR_INVALID, "sample: this shard has no output weights");
    GemvParams p;
    memset(&p, 0, sizeof(p));
    p.w = (const uint8_t*)w->data; p.ws = w->scales; p.nrows = c.vocab_size; p.out = s->logits;
    p.K = c.embedding_length; p.ldb = p.K / 2; p.ldbf = p.K / QB;
    p.x = s->x; p.nw = (const float*)m->global_w[JH_W_FINALNORM].data;
    p.eps = c.rms_eps;
    p.amax_part = s->amax_v; p.amax_idx = s->amax_i;
    int grid = 0;
    if (w->dtype == JH_DT_BF16 && s->strict) {
        JHCHK(use_p16t(p, *w));
        JHCHK((launch_gemv_bf16r<PROB_RMS_F32, EPI_STORE, true>(p, &grid, st)));       // F32 x BF16 in GemmerF32BF16's order (PTO:1511-1538)
    } else if (w->dtype == JH_DT_BF16) {
        p.ldb = p.K * 2;
        JHCHK((launch_gemv_bf16<PROB_RMS_F32, EPI_STORE, true>(p, 4096, &grid, st)));   // F32 x BF16 (GemmerF32BF16)
    } else if (s->strict) {
        JHCHK(use_p16t(p, *w));
        JHCHK((launch_gemv_f32q4_p16<PRO_RMS_F32>(p, &grid, st)));
    } else {
        JHCHK((launch_gemv_f32q4<PRO_RMS_F32>(p, s->cfg_lm, &grid, st)));
    }
    s->lm_grid = grid;
    return JH_OK;
}

int finish_launch(jh_session* s, hipStream_t st, int do_embed, float temperature) {
    jh_model* m = s->m;
    const JWeight& e = m->global_w[JH_W_EMBED];
    const int V = m->c.vocab_size;
    if (temperature != 0.0f) {   // AbstractModel.java:471-489 on the device: exponentials, then the two sequential float accumulations
        hipLaunchKernelGGL(sample_exp_kernel, dim3((V + 255) / 256), dim3(256), 0, st, (const float*)s->logits, V, (const float*)s->amax_v, s->lm_grid,
                           temperature, s->prob);
        float* sum = s->prob + (((size_t)V + 3) & ~(size_t)3);
        const size_t lds = lds_bytes_sample(SAMPLE_T_DEFAULT, SAMPLE_E_DEFAULT);
        JHCHK(allow_lds((sample_sum_kernel<SAMPLE_T_DEFAULT, SAMPLE_E_DEFAULT>), lds));
        JHCHK(allow_lds((sample_pick_kernel<SAMPLE_T_DEFAULT, SAMPLE_E_DEFAULT>), lds));
        hipLaunchKernelGGL((sample_sum_kernel<SAMPLE_T_DEFAULT, SAMPLE_E_DEFAULT>), dim3(1), dim3(SAMPLE_T_DEFAULT), lds, st, (const float*)s->prob, V, (const DecodeState*)s->st, sum);
        hipLaunchKernelGGL(sample_norm_kernel, dim3((V + 255) / 256), dim3(256), 0, st, s->prob, V, (const DecodeState*)s->st, (const float*)sum);
        hipLaunchKernelGGL((sample_pick_kernel<SAMPLE_T_DEFAULT, SAMPLE_E_DEFAULT>), dim3(1), dim3(SAMPLE_T_DEFAULT), lds, st, (const float*)s->prob, V, (const float*)s->u_dev, (const DecodeState*)s->st, s->pick);
    }
    hipLaunchKernelGGL(finish_token_kernel, dim3(1), dim3(256), 0, st, (const float*)s->amax_v, (const int*)s->amax_i, s->lm_grid,
                       s->st, s->out_tokens, (const void*)e.data, (const float*)e.scales, e.dtype, m->c.embedding_length, s->x,
                       (do_embed && e.data) ? 1 : 0, (const int*)s->eos_dev, temperature != 0.0f ? (const int*)s->pick : (const int*)nullptr);
    HIPCHK(hipGetLastError());
    return JH_OK;
}

int ensure_out_tokens(jh_session* s, int n) {
    if (s->out_cap >= n) return JH_OK;
    if (s->out_tokens) HIPCHK(hipFree(s->out_tokens));
    HIPCHK(hipMalloc(&s->out_tokens, (size_t)n * sizeof(int)));
    s->out_cap = n;
    for (int v = 0; v < N_ATTN_VARIANTS; v++) {   // out_tokens pointer is baked into the captured graphs
        if (s->exec[v]) { hipGraphExecDestroy(s->exec[v]); s->exec[v] = nullptr; hipGraphDestroy(s->graph[v]); s->graph[v] = nullptr; }
        if (s->exec_m[v]) { hipGraphExecDestroy(s->exec_m[v]); s->exec_m[v] = nullptr; hipGraphDestroy(s->graph_m[v]); s->graph_m[v] = nullptr; }
        if (s->exec_s[v]) { hipGraphExecDestroy(s->exec_s[v]); s->exec_s[v] = nullptr; hipGraphDestroy(s->graph_s[v]); s->graph_s[v] = nullptr; }
    }
    return JH_OK;
}

