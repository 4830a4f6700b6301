// pipeline.hip -- part of libjlamahip.so (C ABI: include/jlama_hip.h).  Layer split: the one-process N-device pipeline host and the stream-ordered stage step of the rank-per-GPU host.
#include "jh_host.h"
#include "jh_launch.h"

extern "C" {

// ---- one-process layer-sharded pipeline ------------------------------------------------------------------------------
// 1 = device `dev` can address `peer`'s memory directly (enabled now or earlier), 0 = it cannot
static int enable_peer(int dev, int peer) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dev, peer) != hipSuccess || !can) { (void)hipGetLastError(); return 0; }
    hipSetDevice(dev);
    const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
    (void)hipGetLastError();
    return (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ? 1 : 0;
}
struct jh_pipeline {
    std::vector<int> peer_ok;             // per hop k-1 -> k (index k, [0] = the token's way back): 1 direct peer access, 0 staged copies, -1 same device
    std::vector<jh_session*> st;          // stages in order
    std::vector<float*> hop;              // per stage: [PB_MAX_ROWS, E] F32 on the stage's device (prefill hand-off landing zone)
    std::vector<hipEvent_t> done;         // per stage: its part of the current row / chunk is complete
    int pending_n = 0;
};
int jh_pipeline_create(jh_session* const* stages, int n_stages, jh_pipeline** out) {
    if (!stages || n_stages <= 0 || !out) return set_err(JH_ERR_INVALID, "pipeline_create: bad argument");
    for (int k = 0; k < n_stages; k++) {
        if (!stages[k]) return set_err(JH_ERR_INVALID, "pipeline_create: null stage");
        const jh_config& c = stages[k]->m->c;
        const jh_config& c0 = stages[0]->m->c;
        if (c.embedding_length != c0.embedding_length || c.n_layers != c0.n_layers)
            return set_err(JH_ERR_INVALID, "pipeline_create: stages belong to different models");
        if (k > 0 && c.layer_start != stages[k - 1]->m->c.layer_end)
            return set_err(JH_ERR_INVALID, "pipeline_create: stage layer ranges must be contiguous and in order");
    }
    if (stages[0]->m->c.layer_start != 0 || stages[n_stages - 1]->m->c.layer_end != stages[0]->m->c.n_layers)
        return set_err(JH_ERR_INVALID, "pipeline_create: stages must cover layers [0, n_layers)");
    if (!stages[0]->m->global_w[JH_W_EMBED].data) return set_err(JH_ERR_INVALID, "pipeline_create: the first stage needs the embedding table");
    jh_model* ml = stages[n_stages - 1]->m;
    if (!lm_head_weight(ml)->data || !ml->global_w[JH_W_FINALNORM].data)
        return set_err(JH_ERR_INVALID, "pipeline_create: the last stage needs final norm and LM head");
    jh_pipeline* p = new jh_pipeline();
    const size_t E = (size_t)stages[0]->m->c.embedding_length;
    for (int k = 0; k < n_stages; k++) {
        jh_session* s = stages[k];
        p->st.push_back(s);
        hipSetDevice(s->m->device);
        float* h = nullptr;
        hipEvent_t ev = nullptr;
        if (hipMalloc(&h, (size_t)PB_MAX_ROWS * E * 4) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            p->hop.push_back(h); p->done.push_back(ev);
            jh_pipeline_destroy(p);
            return set_err(JH_ERR_OOM, "pipeline_create: hop buffers");
        }
        p->hop.push_back(h);
        p->done.push_back(ev);
        // direct xGMI copies between neighbouring stages; the outcome is recorded per hop (jh_pipeline_peer_access): without peer
        // access hipMemcpyPeerAsync still works, staged through the host -- correct, but not the xGMI hop the design counts on
        p->peer_ok.push_back(-1);
        if (k > 0 && stages[k - 1]->m->device != s->m->device) p->peer_ok[k] = enable_peer(s->m->device, stages[k - 1]->m->device);
    }
    if (n_stages > 1 && stages[0]->m->device != stages[n_stages - 1]->m->device)   // the token id's way back
        p->peer_ok[0] = enable_peer(stages[0]->m->device, stages[n_stages - 1]->m->device);
    *out = p;
    return JH_OK;
}
int jh_pipeline_peer_access(jh_pipeline* p, int32_t* out, int n) {
    if (!p || !out || n < (int)p->st.size()) return set_err(JH_ERR_INVALID, "pipeline_peer_access: need one slot per stage");
    for (size_t k = 0; k < p->st.size(); k++) out[k] = p->peer_ok[k];
    return (int)p->st.size();
}
int jh_pipeline_destroy(jh_pipeline* p) {
    if (!p) return JH_OK;
    for (size_t k = 0; k < p->st.size(); k++) {
        hipSetDevice(p->st[k]->m->device);
        hipStreamSynchronize(p->st[k]->stream);
        if (k < p->hop.size() && p->hop[k]) hipFree(p->hop[k]);
        if (k < p->done.size() && p->done[k]) hipEventDestroy(p->done[k]);
    }
    delete p;
    return JH_OK;
}
int jh_pipeline_prefill(jh_pipeline* p, const int32_t* tokens, int n, int start_pos, int32_t* first_token) {
    if (!p || !tokens || n <= 0 || !first_token) return set_err(JH_ERR_INVALID, "pipeline_prefill: bad argument");
    const int N = (int)p->st.size();
    const size_t E = (size_t)p->st[0]->m->c.embedding_length;
    for (int done = 0; done < n; done += PB_MAX_ROWS) {
        const int rows = n - done < PB_MAX_ROWS ? n - done : PB_MAX_ROWS;
        for (int k = 0; k < N; k++) {
            jh_session* s = p->st[k];
            HIPCHK(hipSetDevice(s->m->device));
            // a stage may overwrite its hop buffer only after the next stage has pulled the previous chunk out of it
            if (done > 0 && k + 1 < N) HIPCHK(hipStreamWaitEvent(s->stream, p->done[k + 1], 0));
            if (k > 0) {
                jh_session* prev = p->st[k - 1];
                HIPCHK(hipStreamWaitEvent(s->stream, p->done[k - 1], 0));
                // the previous stage left its [rows, E] output in ITS hop buffer; pull it across
                HIPCHK(hipMemcpyPeerAsync(p->hop[k], s->m->device, p->hop[k - 1], prev->m->device, (size_t)rows * E * 4, s->stream));
            }
            // in place on the stage's own hop buffer: input rows -> output rows (stream-ordered inside jh_forward_device)
            JHCHK(jh_forward_device(s, k == 0 ? tokens + done : nullptr, k == 0 ? nullptr : p->hop[k], rows, start_pos + done, p->hop[k]));
            HIPCHK(hipEventRecord(p->done[k], s->stream));
        }
    }
    jh_session* last = p->st[N - 1];
    HIPCHK(hipSetDevice(last->m->device));
    return jh_sample(last, 0.0f, 0.5f, first_token, nullptr);
}
int jh_pipeline_decode_n_async(jh_pipeline* p, int32_t first_token, int start_pos, int n) {
    if (!p || n <= 0 || start_pos < 0) return set_err(JH_ERR_INVALID, "pipeline_decode_n: bad argument");
    const int N = (int)p->st.size();
    jh_session* s0 = p->st[0];
    jh_session* sl = p->st[N - 1];
    for (int k = 0; k < N; k++) {
        if (start_pos + n > p->st[k]->max_ctx) return set_err(JH_ERR_INVALID, "pipeline_decode_n: positions beyond a stage's max_ctx");
        JHCHK(check_positions(p->st[k], start_pos + n - 1));
    }
    if (first_token < 0 || first_token >= s0->m->c.vocab_size) return set_err(JH_ERR_INVALID, "pipeline_decode_n: token id out of range");
    if (N == 1) { p->pending_n = n; return jh_decode_n_async(s0, first_token, start_pos, n); }
    const size_t E = (size_t)s0->m->c.embedding_length;
    const JWeight& emb = s0->m->global_w[JH_W_EMBED];
    HIPCHK(hipSetDevice(sl->m->device));
    JHCHK(ensure_out_tokens(sl, n));
    // graphs first (a capture costs milliseconds and must not sit inside the queued loop)
    for (int k = 0; k < N; k++) {
        jh_session* s = p->st[k];
        HIPCHK(hipSetDevice(s->m->device));
        for (int v = 0; v < N_ATTN_VARIANTS; v++) {
            if (!attn_variant_in_range(s, v, start_pos, start_pos + n - 1)) continue;
            if (k == N - 1) JHCHK(build_graph(s, v)); else JHCHK(build_row_graph(s, v));
        }
    }
    HIPCHK(hipSetDevice(s0->m->device));
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, s0->stream, s0->st, start_pos, first_token, 0);
    HIPCHK(hipSetDevice(sl->m->device));
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, sl->stream, sl->st, start_pos, first_token, 0);
    HIPCHK(hipEventRecord(sl->ev0, sl->stream));
    for (int i = 0; i < n; i++) {
        const int pos = start_pos + i;
        for (int k = 0; k < N; k++) {
            jh_session* s = p->st[k];
            HIPCHK(hipSetDevice(s->m->device));
            hipStream_t st = s->stream;
            const int v = attn_variant_for(s, pos);
            if (k == 0) {
                if (i > 0) {   // the id sampled by the last stage for the previous position
                    HIPCHK(hipStreamWaitEvent(st, p->done[N - 1], 0));
                    HIPCHK(hipMemcpyPeerAsync(&s->st->token, s->m->device, &sl->st->token, sl->m->device, sizeof(int), st));
                }
                hipLaunchKernelGGL(set_pos_kernel, dim3(1), dim3(1), 0, st, s->st, pos);
                hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, st, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                                   (const DecodeState*)s->st, (int)E, s->x);
                HIPCHK(hipGetLastError());
                HIPCHK(hipGraphLaunch(s->row_exec[v], st));
            } else {
                HIPCHK(hipStreamWaitEvent(st, p->done[k - 1], 0));
                HIPCHK(hipMemcpyPeerAsync(s->x, s->m->device, p->st[k - 1]->x, p->st[k - 1]->m->device, E * 4, st));
                hipLaunchKernelGGL(set_pos_kernel, dim3(1), dim3(1), 0, st, s->st, pos);
                HIPCHK(hipGetLastError());
                HIPCHK(hipGraphLaunch(k == N - 1 ? s->exec[v] : s->row_exec[v], st));   // last stage: layers + LM head + argmax
            }
            HIPCHK(hipEventRecord(p->done[k], st));
        }
    }
    HIPCHK(hipSetDevice(sl->m->device));
    HIPCHK(hipEventRecord(sl->ev1, sl->stream));
    sl->pending_n = n;
    p->pending_n = n;
    return JH_OK;
}
int jh_pipeline_decode_wait(jh_pipeline* p, int32_t* out_tokens, int n) {
    if (!p) return set_err(JH_ERR_INVALID, "pipeline_decode_wait: null");
    jh_session* sl = p->st.back();
    for (jh_session* s : p->st) { HIPCHK(hipSetDevice(s->m->device)); HIPCHK(hipStreamSynchronize(s->stream)); }
    HIPCHK(hipSetDevice(sl->m->device));
    p->pending_n = 0;
    return jh_decode_wait(sl, out_tokens, n);
}

// ---- one pipeline stage per process (rank-per-GPU hosts): one decode row of THIS shard, stream-ordered end to end ----------
// The caller's transport (RCCL send/recv issued on the session's stream, jh_session_stream) delivers x_in_dev / token_dev and
// ships x_out_dev / token_out_dev; nothing here touches the host, so a rank can queue its ticks ahead of the GPU.
int jh_stage_decode_async(jh_session* s, const int32_t* token_dev, const float* x_in_dev, int pos, float* x_out_dev, int32_t* token_out_dev) {
    if (!s || pos < 0) return set_err(JH_ERR_INVALID, "stage_decode: bad argument");
    if (pos + 1 > s->max_ctx) return set_err(JH_ERR_INVALID, "stage_decode: position beyond the session's max_ctx");
    JHCHK(check_positions(s, pos));
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const bool first = c.layer_start == 0, last = c.layer_end == c.n_layers;
    const JWeight& emb = m->global_w[JH_W_EMBED];
    if (first && (!token_dev || !emb.data)) return set_err(JH_ERR_INVALID, "stage_decode: the first stage needs a token word and the embedding table");
    if (!first && !x_in_dev) return set_err(JH_ERR_INVALID, "stage_decode: a later stage needs the previous stage's row");
    if (last && (!token_out_dev || !lm_head_weight(m)->data || !m->global_w[JH_W_FINALNORM].data))
        return set_err(JH_ERR_INVALID, "stage_decode: the last stage needs final norm, LM head and a token destination");
    if (!last && !x_out_dev) return set_err(JH_ERR_INVALID, "stage_decode: this stage needs a destination for its row");
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = s->stream;
    const size_t E = (size_t)c.embedding_length;
    const int v = attn_variant_for(s, pos);
    if (last) { JHCHK(ensure_out_tokens(s, 1)); JHCHK(build_graph(s, v)); }
    else JHCHK(build_row_graph(s, v));
    hipLaunchKernelGGL(set_state_dev_kernel, dim3(1), dim3(1), 0, st, s->st, pos, first ? token_dev : nullptr);
    if (first)
        hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, st, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                           (const DecodeState*)s->st, (int)E, s->x);
    else
        HIPCHK(hipMemcpyAsync(s->x, x_in_dev, E * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipGetLastError());
    HIPCHK(hipGraphLaunch(last ? s->exec[v] : s->row_exec[v], st));   // last stage: layers + LM head + argmax -> st->token
    if (last) {
        hipLaunchKernelGGL(store_token_kernel, dim3(1), dim3(1), 0, st, (const DecodeState*)s->st, token_out_dev);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemcpyAsync(x_out_dev, s->x, E * 4, hipMemcpyDeviceToDevice, st));
    }
    return JH_OK;
}


}  // extern "C"
