// tp.hip -- part of libjlamahip.so (C ABI: include/jlama_hip.h).  Tensor parallel (head split): shard half-layers, the one-process group and the rank-per-process host meeting inside kernels.
#include "jh_host.h"
#include "jh_launch.h"

extern "C" {

// ---- tensor-parallel (head-split) shard: the model is created with its LOCAL head counts / hidden length and holds
// the matching row / column windows of the weights (DistributedContext.java:79-98); the caller all-reduces the partial
// [E] results between the halves (tensorReducer, CausalSelfAttention.java:378, MLPBlock.java:160).  All calls are
// asynchronous on the session's stream; pointers are device pointers.
int jh_model_set_kv_head_offset(jh_model* m, int kv_head_offset) {
    if (!m || kv_head_offset < 0) return set_err(JH_ERR_INVALID, "set_kv_head_offset: bad argument");
    m->kv_head_offset = kv_head_offset;
    return JH_OK;
}
int jh_tp_set_row(jh_session* s, int32_t token, const float* x_dev, int pos) {
    if (!s || pos < 0 || pos >= s->max_ctx) return set_err(JH_ERR_INVALID, "tp_set_row: bad argument");
    JHCHK(check_positions(s, pos));
    jh_model* m = s->m;
    if (!x_dev) {   // validate before anything is queued
        if (!m->global_w[JH_W_EMBED].data) return set_err(JH_ERR_INVALID, "tp_set_row: this shard has no embedding table");
        if (token < 0 || token >= m->c.vocab_size) return set_err(JH_ERR_INVALID, "tp_set_row: token id out of range");
    }
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = s->stream;
    const int E = m->c.embedding_length;
    s->attn_variant = attn_variant_for(s, pos);
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, s->st, pos, token >= 0 ? token : 0, 0);
    if (x_dev) {
        HIPCHK(hipMemcpyAsync(s->x, x_dev, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
    } else {
        const JWeight& emb = m->global_w[JH_W_EMBED];
        if (!emb.data) return set_err(JH_ERR_INVALID, "tp_set_row: this shard has no embedding table");
        if (token < 0 || token >= m->c.vocab_size) return set_err(JH_ERR_INVALID, "tp_set_row: token id out of range");
        hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, st, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                           (const DecodeState*)s->st, E, s->x);
    }
    HIPCHK(hipGetLastError());
    return JH_OK;
}
int jh_tp_attn(jh_session* s, int layer, float* partial_out_dev) {
    if (!s || !partial_out_dev || layer < s->m->c.layer_start || layer >= s->m->c.layer_end) return set_err(JH_ERR_INVALID, "tp_attn: bad argument");
    HIPCHK(hipSetDevice(s->m->device));
    return layer_attn_launch(s, layer, s->stream, false, 0, partial_out_dev, nullptr);
}
int jh_tp_ffn(jh_session* s, int layer, const float* reduced_attn_dev, float* partial_out_dev) {
    if (!s || !reduced_attn_dev || !partial_out_dev || layer < s->m->c.layer_start || layer >= s->m->c.layer_end)
        return set_err(JH_ERR_INVALID, "tp_ffn: bad argument");
    HIPCHK(hipSetDevice(s->m->device));
    const int E = s->m->c.embedding_length;
    // residual (TransformerBlock.java:185) after the reduction: x1 = x + sum_shards(o-proj partial)
    hipLaunchKernelGGL(add_rows_kernel, dim3((E + 255) / 256), dim3(256), 0, s->stream, (const float*)s->x, reduced_attn_dev, s->x1, E);
    HIPCHK(hipGetLastError());
    return layer_ffn_launch(s, layer, s->stream, false, partial_out_dev, nullptr);
}
int jh_tp_finish_layer(jh_session* s, const float* reduced_ffn_dev) {
    if (!s || !reduced_ffn_dev) return set_err(JH_ERR_INVALID, "tp_finish_layer: bad argument");
    HIPCHK(hipSetDevice(s->m->device));
    const int E = s->m->c.embedding_length;
    // residual (TransformerBlock.java:203): x = x1 + sum_shards(down partial)
    hipLaunchKernelGGL(add_rows_kernel, dim3((E + 255) / 256), dim3(256), 0, s->stream, (const float*)s->x1, reduced_ffn_dev, s->x, E);
    HIPCHK(hipGetLastError());
    return JH_OK;
}
// ---- the same halves over a chunk of prompt rows (AbstractModel.batchForward on a head-split shard: the reducer then sums
// [rows, E], CausalSelfAttention.java:378 / MLPBlock.java:160): one meeting per half-layer and chunk instead of one per row
int jh_tp_rows_max(jh_session* s) {
    if (!s) return 0;
    return prefill_batch_ok(s) ? PB_MAX_ROWS : 0;
}
int jh_tp_set_rows(jh_session* s, const int32_t* tokens, const float* x_dev, int n, int start_pos) {
    if (!s || (!tokens && !x_dev) || n <= 0 || n > PB_MAX_ROWS || start_pos < 0 || start_pos + n > s->max_ctx)
        return set_err(JH_ERR_INVALID, "tp_set_rows: bad argument (at most 256 rows per chunk)");
    if (!prefill_batch_ok(s)) return set_err(JH_ERR_UNSUPPORTED, "tp_set_rows: this shard's shapes have no batched path (jh_tp_rows_max == 0): feed rows with jh_tp_set_row");
    if (!s->strict && !prefill_chunk_fits(s, start_pos, n))
        return set_err(JH_ERR_UNSUPPORTED, "tp_set_rows: the score rows of this chunk do not fit the per-row attention kernel (feed these rows with jh_tp_set_row)");
    JHCHK(check_positions(s, start_pos + n - 1));
    jh_model* m = s->m;
    const JWeight& emb = m->global_w[JH_W_EMBED];
    if (tokens) {
        if (!emb.data) return set_err(JH_ERR_INVALID, "tp_set_rows: this shard has no embedding table");
        for (int i = 0; i < n; i++)
            if (tokens[i] < 0 || tokens[i] >= m->c.vocab_size) return set_err(JH_ERR_INVALID, "tp_set_rows: token id out of range");
    }
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = s->stream;
    JHCHK(ensure_strict_operands(s, st));
    JHCHK(prefill_alloc(s));
    if (s->strict) JHCHK(prefill_p16_operands(s));
    else JHCHK(ensure_all_tiled(s, st));
    const int E = m->c.embedding_length;
    if (tokens) {
        HIPCHK(hipMemcpyAsync(s->pb_tok, tokens, (size_t)n * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(embed_rows_kernel, dim3(n), dim3(256), 0, st, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                           (const int*)s->pb_tok, E, s->pb_x);
    } else {
        HIPCHK(hipMemcpyAsync(s->pb_x, x_dev, (size_t)n * E * 4, hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, st, s->pb_start, start_pos);
    HIPCHK(hipGetLastError());
    s->tp_rows = n;
    s->tp_pos0 = start_pos;
    s->tp_last_token = tokens ? tokens[n - 1] : 0;
    return JH_OK;
}
int jh_tp_attn_rows(jh_session* s, int layer, float* partial_out_dev) {
    if (!s || !partial_out_dev || layer < s->m->c.layer_start || layer >= s->m->c.layer_end || s->tp_rows <= 0)
        return set_err(JH_ERR_INVALID, "tp_attn_rows: bad argument (jh_tp_set_rows first)");
    HIPCHK(hipSetDevice(s->m->device));
    const int rows = s->tp_rows, pos0 = s->tp_pos0;
    if (s->strict && s->m->c.weight_dtype == JH_DT_BF16) return prefill_attn_half_bf16r(s, layer, rows, pos0, partial_out_dev, nullptr, s->stream);
    if (s->strict) return prefill_attn_half_p16(s, layer, rows, pos0, partial_out_dev, nullptr, s->stream);
    int bound = 1024;
    while (bound < pos0 + rows) bound *= 2;
    const bool attn_mfma = prefill_attn_mfma(s, pos0, rows);
    if (!attn_mfma && !prefill_chunk_fits(s, 0, bound)) bound = pos0 + rows;
    return prefill_attn_half(s, layer, rows, bound, attn_mfma, partial_out_dev, nullptr, s->stream);
}
int jh_tp_ffn_rows(jh_session* s, int layer, const float* reduced_attn_dev, float* partial_out_dev) {
    if (!s || !reduced_attn_dev || !partial_out_dev || layer < s->m->c.layer_start || layer >= s->m->c.layer_end || s->tp_rows <= 0)
        return set_err(JH_ERR_INVALID, "tp_ffn_rows: bad argument (jh_tp_set_rows first)");
    HIPCHK(hipSetDevice(s->m->device));
    const int cnt = s->tp_rows * s->m->c.embedding_length;
    // residual (TransformerBlock.java:185) after the reduction: x1 = x + sum_shards(o-proj partial), every row of the chunk
    hipLaunchKernelGGL(add_rows_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s->stream, (const float*)s->pb_x, reduced_attn_dev, s->pb_x1, cnt);
    HIPCHK(hipGetLastError());
    if (s->strict && s->m->c.weight_dtype == JH_DT_BF16) return prefill_ffn_half_bf16r(s, layer, s->tp_rows, s->pb_x1, partial_out_dev, nullptr, s->stream);
    if (s->strict) return prefill_ffn_half_p16(s, layer, s->tp_rows, s->pb_x1, partial_out_dev, nullptr, s->stream);
    return prefill_ffn_half(s, layer, s->tp_rows, s->pb_x1, partial_out_dev, nullptr, s->stream);
}
int jh_tp_finish_layer_rows(jh_session* s, const float* reduced_ffn_dev) {
    if (!s || !reduced_ffn_dev || s->tp_rows <= 0) return set_err(JH_ERR_INVALID, "tp_finish_layer_rows: bad argument");
    HIPCHK(hipSetDevice(s->m->device));
    const int cnt = s->tp_rows * s->m->c.embedding_length;
    hipLaunchKernelGGL(add_rows_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s->stream, (const float*)s->pb_x1, reduced_ffn_dev, s->pb_x, cnt);   // :203
    HIPCHK(hipGetLastError());
    return JH_OK;
}
// after the last layer: the chunk's last row becomes the session's current row (what sample() reads); rows_out_dev (optional) gets all rows
int jh_tp_finish_rows(jh_session* s, float* rows_out_dev) {
    if (!s || s->tp_rows <= 0) return set_err(JH_ERR_INVALID, "tp_finish_rows: no chunk in flight");
    HIPCHK(hipSetDevice(s->m->device));
    const int E = s->m->c.embedding_length, rows = s->tp_rows;
    hipStream_t st = s->stream;
    HIPCHK(hipMemcpyAsync(s->x, s->pb_x + (size_t)(rows - 1) * E, (size_t)E * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, s->st, s->tp_pos0 + rows - 1, s->tp_last_token, 0);
    if (rows_out_dev) HIPCHK(hipMemcpyAsync(rows_out_dev, s->pb_x, (size_t)rows * E * 4, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipGetLastError());
    s->tp_rows = 0;
    return JH_OK;
}
// ---- one-process tensor-parallel group ---------------------------------------------------------------------------------
static int tp_enable_peer(int dev, int peer) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dev, peer) != hipSuccess || !can) { (void)hipGetLastError(); return 0; }
    hipSetDevice(dev);
    const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
    (void)hipGetLastError();
    return (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ? 1 : 0;
}
struct jh_tp_group {
    std::vector<jh_session*> sh;
    std::vector<float*> part, red, slots;      // per shard, on its device: [E], [E], [2 rounds][N][E]
    std::vector<float**> peers;                // per shard, on its device: [2 rounds][N] destination pointers of ITS slot on every shard
    std::vector<hipEvent_t> evA, evB, evTok;
    // graph-replayed decode (no host inside a token): flag words per (round, producing shard, workgroup), the producers' pointer
    // tables into every shard's flags, the token mailboxes, a per-shard token counter, and one captured graph per attention variant
    int nwg = 0;
    std::vector<unsigned*> flags;              // per shard: [2][N][TP_MAX_FLAGS] (a producer launch uses as many words as it has workgroups)
    std::vector<unsigned**> peers_f;           // per shard: [2][N] -> ITS flag row on every shard
    std::vector<TPMail*> mail;                 // per shard (shard 0's is unused)
    TPMail** mails_dev = nullptr;              // on shard 0's device: the other shards' mailboxes
    std::vector<unsigned*> seq;                // per shard: tokens replayed so far
    std::vector<hipGraph_t> graph[N_ATTN_VARIANTS];
    std::vector<hipGraphExec_t> exec[N_ATTN_VARIANTS];
    int graphs_strict = -1, graphs_version = -1;
    bool graph_ok = true;                      // false after a wait timed out once: this group stays on the event-ordered loop
    int timeouts = 0;                          // meetings that ran into their bound so far (jh_tp_group_status)
    int last_mode = 0;                         // what the last decode_n ran on: 1 graph replay per shard and token, 2 event-ordered host loop
    int fused_push = -1, flags_per_launch = 0; // launch plan of the pushing GEMVs as captured (EPI_TP or scatter kernel; flag words polled per producer)
    bool fresh_graphs = true;                  // the first replay after a capture uploads the graphs: its waits get a longer bound
    int plan_flags[2] = {-1, -1};              // flag words per producer launch (0 = scatter kernel) of the o-proj / down meeting: one plan for ALL shards
    // one process per shard (jh_tp_rank_*): only shard `local` lives here, the others' slot / flag / mailbox buffers are mapped
    // through hipIpc handles (slots_of / flags_of / mail_of[j] = shard j's buffer as addressable from this process)
    int local = -1;
    bool connected = false;                    // jh_tp_rank_connect filled the pointer tables
    std::vector<float*> slots_of;
    std::vector<unsigned*> flags_of;
    std::vector<TPMail*> mail_of;
    std::vector<void*> ipc_open;               // mappings to close
    // shards that share ONE device (loopback runs): each gets a stream with its own CU mask for the life of the group -- a hardware
    // queue of its own (the runtime multiplexes plain streams over a few queues; two shards on one queue cannot meet inside
    // kernels) and CUs no other shard's spinning kernel can occupy
    std::vector<hipStream_t> masked, unmasked;
    // prompt chunks (jh_tp_group_forward): per shard [256 rows][E] partial / reduced rows, [2 rounds][N][256][E] slots + pointer tables
    std::vector<float*> part_rows, red_rows, slots_rows;
    std::vector<float**> peers_rows;
};
static void tp_mask_streams(jh_tp_group* g) {
    const size_t N = g->sh.size();
    g->masked.assign(N, nullptr);
    g->unmasked.assign(N, nullptr);
    if (!opt_int("JH_TP_CU_MASK", 1)) return;
    for (size_t k = 0; k < N; k++) {
        if (!g->sh[k] || g->masked[k]) continue;
        const int dev = g->sh[k]->m->device;
        std::vector<size_t> same;
        for (size_t j = 0; j < N; j++) if (g->sh[j] && g->sh[j]->m->device == dev) same.push_back(j);
        if (same.size() < 2) continue;
        hipDeviceProp_t prop;
        if (hipSetDevice(dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); continue; }
        const int cus = prop.multiProcessorCount, words = (cus + 31) / 32;
        for (size_t i = 0; i < same.size(); i++) {
            std::vector<uint32_t> mask((size_t)words, 0u);
            for (int cu = 0; cu < cus; cu++) if ((size_t)cu % same.size() == i) mask[cu >> 5] |= 1u << (cu & 31);
            hipStream_t ns = nullptr;
            if (hipExtStreamCreateWithCUMask(&ns, (uint32_t)words, mask.data()) != hipSuccess) { (void)hipGetLastError(); continue; }
            jh_session* s = g->sh[same[i]];
            hipStreamSynchronize(s->stream);
            g->unmasked[same[i]] = s->stream;
            g->masked[same[i]] = ns;
            s->stream = ns;
        }
    }
}
static void tp_unmask_streams(jh_tp_group* g) {
    for (size_t k = 0; k < g->masked.size() && k < g->sh.size(); k++) {
        if (!g->masked[k] || !g->sh[k]) continue;
        hipSetDevice(g->sh[k]->m->device);
        hipStreamSynchronize(g->masked[k]);
        g->sh[k]->stream = g->unmasked[k];
        hipStreamDestroy(g->masked[k]);
        g->masked[k] = nullptr;
    }
}
// memory that kernels of several devices meet in: fine-grained (coherent at system scope inside a kernel) where the runtime
// offers it, plain device memory otherwise (enough when all shards share one device)
static hipError_t tp_shared_malloc(void** p, size_t bytes) {
    hipError_t e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(p, bytes); }
    return e;
}
int jh_tp_group_destroy(jh_tp_group* g) {
    if (!g) return JH_OK;
    for (size_t k = 0; k < g->sh.size(); k++) {
        if (!g->sh[k]) continue;               // rank mode: the other shards live in other processes
        hipSetDevice(g->sh[k]->m->device);
        hipStreamSynchronize(g->sh[k]->stream);
        if (k < g->part.size() && g->part[k]) hipFree(g->part[k]);
        if (k < g->red.size() && g->red[k]) hipFree(g->red[k]);
        if (k < g->slots.size() && g->slots[k]) hipFree(g->slots[k]);
        if (k < g->flags.size() && g->flags[k]) hipFree(g->flags[k]);
        if (k < g->peers_f.size() && g->peers_f[k]) hipFree(g->peers_f[k]);
        if (k < g->mail.size() && g->mail[k]) hipFree(g->mail[k]);
        if (k < g->seq.size() && g->seq[k]) hipFree(g->seq[k]);
        for (int v = 0; v < N_ATTN_VARIANTS; v++) {
            if (k < g->exec[v].size() && g->exec[v][k]) hipGraphExecDestroy(g->exec[v][k]);
            if (k < g->graph[v].size() && g->graph[v][k]) hipGraphDestroy(g->graph[v][k]);
        }
        if (k < g->peers.size() && g->peers[k]) hipFree(g->peers[k]);
        if (k < g->evA.size() && g->evA[k]) hipEventDestroy(g->evA[k]);
        if (k < g->evB.size() && g->evB[k]) hipEventDestroy(g->evB[k]);
        if (k < g->evTok.size() && g->evTok[k]) hipEventDestroy(g->evTok[k]);
    }
    for (size_t k = 0; k < g->part_rows.size(); k++) {
        if (!g->sh[k]) continue;
        hipSetDevice(g->sh[k]->m->device);
        if (g->part_rows[k]) hipFree(g->part_rows[k]);
        if (k < g->red_rows.size() && g->red_rows[k]) hipFree(g->red_rows[k]);
        if (k < g->slots_rows.size() && g->slots_rows[k]) hipFree(g->slots_rows[k]);
        if (k < g->peers_rows.size() && g->peers_rows[k]) hipFree(g->peers_rows[k]);
    }
    tp_unmask_streams(g);
    for (void* p : g->ipc_open) hipIpcCloseMemHandle(p);
    if (g->mails_dev) {
        if (g->sh[0]) hipSetDevice(g->sh[0]->m->device);
        hipFree(g->mails_dev);
    }
    delete g;
    return JH_OK;
}
int jh_tp_group_create(jh_session* const* shards, int n_shards, jh_tp_group** out) {
    if (!shards || n_shards <= 0 || n_shards > 64 || !out) return set_err(JH_ERR_INVALID, "tp_group_create: bad argument");
    for (int k = 0; k < n_shards; k++) {
        if (!shards[k]) return set_err(JH_ERR_INVALID, "tp_group_create: null shard");
        const jh_config &c = shards[k]->m->c, &c0 = shards[0]->m->c;
        if (c.embedding_length != c0.embedding_length || c.n_layers != c0.n_layers || c.layer_start != 0 || c.layer_end != c.n_layers)
            return set_err(JH_ERR_INVALID, "tp_group_create: shards must be head-split shards of ONE model holding all layers");
        if (!shards[k]->m->global_w[JH_W_EMBED].data) return set_err(JH_ERR_INVALID, "tp_group_create: every shard needs the embedding table");
    }
    jh_tp_group* g = new jh_tp_group();
    bool ok_peer = true;
    const int N = n_shards;
    const size_t E = (size_t)shards[0]->m->c.embedding_length;
    bool ok = true;
    for (int k = 0; k < N && ok; k++) {
        g->sh.push_back(shards[k]);
        hipSetDevice(shards[k]->m->device);
        float *p = nullptr, *r = nullptr, *sl = nullptr;
        float** pe = nullptr;
        hipEvent_t a = nullptr, b = nullptr, t = nullptr;
        g->nwg = (int)((E + 255) / 256);
        unsigned *fl = nullptr, *sq = nullptr;
        unsigned** pf = nullptr;
        TPMail* ml = nullptr;
        ok = hipMalloc(&p, E * 4) == hipSuccess && hipMalloc(&r, E * 4) == hipSuccess && tp_shared_malloc((void**)&sl, 2 * (size_t)N * E * 4) == hipSuccess &&
             tp_shared_malloc((void**)&fl, 2 * (size_t)N * TP_MAX_FLAGS * 4) == hipSuccess && hipMemset(fl, 0, 2 * (size_t)N * TP_MAX_FLAGS * 4) == hipSuccess &&
             hipMalloc(&pf, 2 * (size_t)N * sizeof(unsigned*)) == hipSuccess && tp_shared_malloc((void**)&ml, sizeof(TPMail)) == hipSuccess &&
             hipMemset(ml, 0, sizeof(TPMail)) == hipSuccess && hipMalloc(&sq, 64) == hipSuccess && hipMemset(sq, 0, 64) == hipSuccess &&
             hipMalloc(&pe, 2 * (size_t)N * sizeof(float*)) == hipSuccess && hipEventCreateWithFlags(&a, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&b, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&t, hipEventDisableTiming) == hipSuccess;
        g->part.push_back(p); g->red.push_back(r); g->slots.push_back(sl); g->peers.push_back(pe);
        g->flags.push_back(fl); g->peers_f.push_back(pf); g->mail.push_back(ml); g->seq.push_back(sq);
        for (int v = 0; v < N_ATTN_VARIANTS; v++) { g->graph[v].push_back(nullptr); g->exec[v].push_back(nullptr); }
        g->evA.push_back(a); g->evB.push_back(b); g->evTok.push_back(t);
        for (int j = 0; j < k; j++)   // direct peer stores both ways
            if (shards[j]->m->device != shards[k]->m->device) {
                // peer STORES need direct access both ways; without it the group cannot work (no staged fallback for kernels)
                if (!tp_enable_peer(shards[k]->m->device, shards[j]->m->device) || !tp_enable_peer(shards[j]->m->device, shards[k]->m->device))
                    ok_peer = false;
            }
    }
    if (!ok) { jh_tp_group_destroy(g); return set_err(JH_ERR_OOM, "tp_group_create: buffers"); }
    if (!ok_peer) { jh_tp_group_destroy(g); return set_err(JH_ERR_UNSUPPORTED, "tp_group_create: the shards' devices cannot address each other's memory (peer access)"); }
    for (int k = 0; k < N; k++) {   // shard k's slot on shard j, round r:  slots[j] + (r*N + k)*E
        std::vector<float*> h(2 * (size_t)N);
        for (int r = 0; r < 2; r++)
            for (int j = 0; j < N; j++) h[(size_t)r * N + j] = g->slots[j] + ((size_t)r * N + k) * E;
        std::vector<unsigned*> hf(2 * (size_t)N);   // shard k's flag row on shard j, round r: flags[j] + (r*N + k)*TP_MAX_FLAGS
        for (int r = 0; r < 2; r++)
            for (int j = 0; j < N; j++) hf[(size_t)r * N + j] = g->flags[j] + ((size_t)r * N + k) * TP_MAX_FLAGS;
        hipSetDevice(shards[k]->m->device);
        if (hipMemcpy(g->peers[k], h.data(), h.size() * sizeof(float*), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(g->peers_f[k], hf.data(), hf.size() * sizeof(unsigned*), hipMemcpyHostToDevice) != hipSuccess) {
            jh_tp_group_destroy(g);
            return set_err(JH_ERR_HIP, "tp_group_create: peer table upload");
        }
    }
    if (N > 1) {
        std::vector<TPMail*> hm;
        for (int k = 1; k < N; k++) hm.push_back(g->mail[k]);
        hipSetDevice(shards[0]->m->device);
        if (hipMalloc(&g->mails_dev, hm.size() * sizeof(TPMail*)) != hipSuccess ||
            hipMemcpy(g->mails_dev, hm.data(), hm.size() * sizeof(TPMail*), hipMemcpyHostToDevice) != hipSuccess) {
            jh_tp_group_destroy(g);
            return set_err(JH_ERR_HIP, "tp_group_create: mailbox table upload");
        }
    }
    tp_mask_streams(g);
    *out = g;
    return JH_OK;
}
namespace {
// all layers of the row every shard currently holds in s->x at position `pos` (state words already set)
int tp_group_layers(jh_tp_group* g, int pos) {
    const int N = (int)g->sh.size();
    const int E = g->sh[0]->m->c.embedding_length, L = g->sh[0]->m->c.n_layers;
    const dim3 eg((E + 255) / 256), eb(256);
    for (int li = 0; li < L; li++) {
        for (int k = 0; k < N; k++) {
            jh_session* s = g->sh[k];
            HIPCHK(hipSetDevice(s->m->device));
            s->attn_variant = attn_variant_for(s, pos);
            JHCHK(layer_attn_launch(s, li, s->stream, false, 0, g->part[k], nullptr));
            hipLaunchKernelGGL(tp_scatter_kernel, eg, eb, 0, s->stream, (const float*)g->part[k], (float* const*)g->peers[k], N, E);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(g->evA[k], s->stream));
        }
        for (int j = 0; j < N; j++) {
            jh_session* s = g->sh[j];
            HIPCHK(hipSetDevice(s->m->device));
            for (int k = 0; k < N; k++) if (k != j) HIPCHK(hipStreamWaitEvent(s->stream, g->evA[k], 0));
            hipLaunchKernelGGL(tp_sum_kernel, eg, eb, 0, s->stream, (const float*)g->slots[j], N, E, g->red[j], (size_t)E);
            hipLaunchKernelGGL(add_rows_kernel, eg, eb, 0, s->stream, (const float*)s->x, (const float*)g->red[j], s->x1, E);   // TransformerBlock.java:185
            HIPCHK(hipGetLastError());
            JHCHK(layer_ffn_launch(s, li, s->stream, false, g->part[j], nullptr));
            hipLaunchKernelGGL(tp_scatter_kernel, eg, eb, 0, s->stream, (const float*)g->part[j], (float* const*)(g->peers[j] + N), N, E);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(g->evB[j], s->stream));
        }
        for (int j = 0; j < N; j++) {
            jh_session* s = g->sh[j];
            HIPCHK(hipSetDevice(s->m->device));
            for (int k = 0; k < N; k++) if (k != j) HIPCHK(hipStreamWaitEvent(s->stream, g->evB[k], 0));
            hipLaunchKernelGGL(tp_sum_kernel, eg, eb, 0, s->stream, (const float*)(g->slots[j] + (size_t)N * E), N, E, g->red[j], (size_t)E);
            hipLaunchKernelGGL(add_rows_kernel, eg, eb, 0, s->stream, (const float*)s->x1, (const float*)g->red[j], s->x, E);   // :203
            HIPCHK(hipGetLastError());
        }
    }
    return JH_OK;
}
}  // namespace
namespace {
// One token of shard k as a captured graph (attention variant v): [wait for the row | embed]  ->  per layer: attention half,
// scatter + flags, wait + sum + residual, feed-forward half, scatter + flags, wait + sum + residual  ->  [LM head, argmax, next
// row, publish] -> count the token.  Nothing in it depends on the host: the position / token / sequence number are device words.
int tp_build_graph(jh_tp_group* g, int k, int v) {
    jh_session* s = g->sh[k];
    const int strict_key = s->strict;
    if (g->graphs_strict != strict_key || g->graphs_version != s->m->weights_version) {
        for (int vv = 0; vv < N_ATTN_VARIANTS; vv++)
            for (size_t j = 0; j < g->sh.size(); j++) {
                if (g->exec[vv][j]) { hipGraphExecDestroy(g->exec[vv][j]); g->exec[vv][j] = nullptr; }
                if (g->graph[vv][j]) { hipGraphDestroy(g->graph[vv][j]); g->graph[vv][j] = nullptr; }
            }
        g->graphs_strict = strict_key;
        g->graphs_version = s->m->weights_version;
        g->plan_flags[0] = g->plan_flags[1] = -1;
    }
    if (g->exec[v][k]) return JH_OK;
    JHCHK(ensure_strict_operands(s, s->stream));
    g->fresh_graphs = true;
    const int N = (int)g->sh.size();
    const jh_config& c = s->m->c;
    const int E = c.embedding_length, L = c.n_layers;
    const dim3 eg(g->nwg), eb(256);
    HIPCHK(hipSetDevice(s->m->device));
    hipStream_t st = s->stream;
    s->attn_variant = v;
    std::lock_guard<std::mutex> cap(g_capture_mu);
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = JH_OK;
    if (k > 0) {
        const JWeight& emb = s->m->global_w[JH_W_EMBED];
        hipLaunchKernelGGL(tp_wait_token_kernel, dim3(1), dim3(1), 0, st, (const TPMail*)g->mail[k], g->seq[k], s->st);
        hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, st, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                           (const DecodeState*)s->st, E, s->x);
    }
    // the o-proj / down GEMVs push their partial rows and raise the flags themselves (EPI_TP) where a kernel for it exists;
    // otherwise (grid == 0: BF16 model, first-generation strict kernels) a scatter launch follows the GEMV
    const int tp_fuse = opt_int("JH_TP_FUSE", 1);
    TPPush push[2];
    for (int r = 0; r < 2; r++) push[r] = TPPush{(float* const*)(g->peers[k] + (size_t)r * N), (unsigned* const*)(g->peers_f[k] + (size_t)r * N), g->seq[k], N, 0, L, 0};
    auto meet = [&](int r, int li, const float* resid, float* out) {
        const float* slots = g->slots[k] + (size_t)r * N * E;
        const unsigned* flags = g->flags[k] + (size_t)r * N * TP_MAX_FLAGS;
        // the consumer polls as many flag words per producer as ITS OWN GEMV launch raised: every shard must have planned the same
        // launch (same kernel family, CU count, push mode) or the sums would read slots before the last producer workgroup stored
        if (g->plan_flags[r] < 0) g->plan_flags[r] = push[r].grid;
        else if (g->plan_flags[r] != push[r].grid && rc == JH_OK)
            rc = set_err(JH_ERR_INVALID, "tensor-parallel group: shard " + std::to_string(k) + " planned " + std::to_string(push[r].grid) +
                                             " flag words per launch where an earlier shard planned " + std::to_string(g->plan_flags[r]) +
                                             " (different kernel mode or CU count between the shards)");
        g->fused_push = push[r].grid > 0 ? 1 : 0;
        g->flags_per_launch = push[r].grid > 0 ? push[r].grid : g->nwg;
        if (push[r].grid > 0) {
            hipLaunchKernelGGL(tp_sum_wait_all_kernel, eg, eb, 0, st, slots, flags, N, E, push[r].grid, TP_MAX_FLAGS, g->seq[k], li, L, resid, out);
        } else {
            hipLaunchKernelGGL(tp_scatter_flag_kernel, eg, eb, 0, st, (const float*)g->part[k], push[r].dst, push[r].flags, N, E, (const unsigned*)g->seq[k], li, L);
            hipLaunchKernelGGL(tp_sum_wait_kernel, eg, eb, 0, st, slots, flags, N, E, TP_MAX_FLAGS, g->seq[k], li, L, resid, out);
        }
    };
    for (int li = 0; li < L && rc == JH_OK; li++) {
        push[0].li = push[1].li = li;
        push[0].grid = push[1].grid = 0;
        s->tp_push = tp_fuse ? &push[0] : nullptr;
        rc = layer_attn_launch(s, li, st, false, 0, g->part[k], nullptr);
        s->tp_push = nullptr;
        if (rc != JH_OK) break;
        meet(0, li, s->x, s->x1);
        s->tp_push = tp_fuse ? &push[1] : nullptr;
        rc = layer_ffn_launch(s, li, st, false, g->part[k], nullptr);
        s->tp_push = nullptr;
        if (rc != JH_OK) break;
        meet(1, li, s->x1, s->x);
    }
    if (rc == JH_OK && k == 0) {
        rc = lmhead_launch(s, st);
        if (rc == JH_OK) rc = finish_launch(s, st, 1);
        if (rc == JH_OK && N > 1)
            hipLaunchKernelGGL(tp_publish_token_kernel, dim3(1), dim3(64), 0, st, (const DecodeState*)s->st, (TPMail* const*)g->mails_dev, N - 1,
                               (const unsigned*)g->seq[0]);
    }
    hipLaunchKernelGGL(tp_bump_seq_kernel, dim3(1), dim3(1), 0, st, g->seq[k]);
    hipGraph_t gr = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &gr);
    if (rc != JH_OK) { if (gr) hipGraphDestroy(gr); return rc; }
    if (e != hipSuccess) return set_err(JH_ERR_HIP, std::string("hipStreamEndCapture (tp): ") + hipGetErrorString(e));
    g->graph[v][k] = gr;
    HIPCHK(hipGraphInstantiate(&g->exec[v][k], gr, nullptr, nullptr, 0));
    return JH_OK;
}
}  // namespace
namespace {
int tp_group_rows_alloc(jh_tp_group* g) {
    const size_t N = g->sh.size();
    if (g->part_rows.size() == N) return JH_OK;
    const size_t RE = (size_t)PB_MAX_ROWS * g->sh[0]->m->c.embedding_length;
    g->part_rows.assign(N, nullptr); g->red_rows.assign(N, nullptr); g->slots_rows.assign(N, nullptr); g->peers_rows.assign(N, nullptr);
    for (size_t k = 0; k < N; k++) {
        HIPCHK(hipSetDevice(g->sh[k]->m->device));
        if (hipMalloc(&g->part_rows[k], RE * 4) != hipSuccess || hipMalloc(&g->red_rows[k], RE * 4) != hipSuccess ||
            tp_shared_malloc((void**)&g->slots_rows[k], 2 * N * RE * 4) != hipSuccess || hipMalloc(&g->peers_rows[k], 2 * N * sizeof(float*)) != hipSuccess) {
            g->part_rows.clear();
            return set_err(JH_ERR_OOM, "tp_group_forward: buffers for a chunk of prompt rows");
        }
    }
    for (size_t k = 0; k < N; k++) {   // shard k's slot on shard j, round r:  slots_rows[j] + (r*N + k)*RE
        std::vector<float*> h(2 * N);
        for (size_t r = 0; r < 2; r++)
            for (size_t j = 0; j < N; j++) h[r * N + j] = g->slots_rows[j] + (r * N + k) * RE;
        HIPCHK(hipSetDevice(g->sh[k]->m->device));
        HIPCHK(hipMemcpy(g->peers_rows[k], h.data(), h.size() * sizeof(float*), hipMemcpyHostToDevice));
    }
    return JH_OK;
}
// one chunk of prompt rows through all layers: the shards' partial [rows, E] results meet once per half-layer (event-ordered;
// sums in shard order like the row loop)
int tp_group_rows(jh_tp_group* g, const int32_t* tokens, int rows, int start_pos) {
    const int N = (int)g->sh.size();
    const int E = g->sh[0]->m->c.embedding_length, L = g->sh[0]->m->c.n_layers, cnt = rows * E;
    const size_t RE = (size_t)PB_MAX_ROWS * E;
    const dim3 eg((cnt + 255) / 256), eb(256);
    JHCHK(tp_group_rows_alloc(g));
    for (jh_session* s : g->sh) JHCHK(jh_tp_set_rows(s, tokens, nullptr, rows, start_pos));
    for (int li = 0; li < L; li++) {
        for (int k = 0; k < N; k++) {
            jh_session* s = g->sh[k];
            JHCHK(jh_tp_attn_rows(s, li, g->part_rows[k]));
            hipLaunchKernelGGL(tp_scatter_kernel, eg, eb, 0, s->stream, (const float*)g->part_rows[k], (float* const*)g->peers_rows[k], N, cnt);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(g->evA[k], s->stream));
        }
        for (int j = 0; j < N; j++) {
            jh_session* s = g->sh[j];
            HIPCHK(hipSetDevice(s->m->device));
            for (int k = 0; k < N; k++) if (k != j) HIPCHK(hipStreamWaitEvent(s->stream, g->evA[k], 0));
            hipLaunchKernelGGL(tp_sum_kernel, eg, eb, 0, s->stream, (const float*)g->slots_rows[j], N, cnt, g->red_rows[j], RE);
            HIPCHK(hipGetLastError());
            JHCHK(jh_tp_ffn_rows(s, li, g->red_rows[j], g->part_rows[j]));
            hipLaunchKernelGGL(tp_scatter_kernel, eg, eb, 0, s->stream, (const float*)g->part_rows[j], (float* const*)(g->peers_rows[j] + N), N, cnt);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(g->evB[j], s->stream));
        }
        for (int j = 0; j < N; j++) {
            jh_session* s = g->sh[j];
            HIPCHK(hipSetDevice(s->m->device));
            for (int k = 0; k < N; k++) if (k != j) HIPCHK(hipStreamWaitEvent(s->stream, g->evB[k], 0));
            hipLaunchKernelGGL(tp_sum_kernel, eg, eb, 0, s->stream, (const float*)(g->slots_rows[j] + (size_t)N * RE), N, cnt, g->red_rows[j], RE);
            HIPCHK(hipGetLastError());
            JHCHK(jh_tp_finish_layer_rows(s, g->red_rows[j]));
        }
    }
    for (jh_session* s : g->sh) JHCHK(jh_tp_finish_rows(s, nullptr));
    return JH_OK;
}
}  // namespace
int jh_tp_group_forward(jh_tp_group* g, const int32_t* tokens, int n, int start_pos) {
    if (!g || !tokens || n <= 0 || start_pos < 0) return set_err(JH_ERR_INVALID, "tp_group_forward: bad argument");
    for (jh_session* s : g->sh) {
        if (start_pos + n > s->max_ctx) return set_err(JH_ERR_INVALID, "tp_group_forward: position beyond a shard's max_ctx");
        JHCHK(check_positions(s, start_pos + n - 1));
    }
    for (int i = 0; i < n; i++)
        if (tokens[i] < 0 || tokens[i] >= g->sh[0]->m->c.vocab_size) return set_err(JH_ERR_INVALID, "tp_group_forward: token id out of range");
    // chunks of >= prefill_batch_min rows go through the layers together (one meeting per half-layer and chunk, the reducer's
    // [batch, E] of CausalSelfAttention.java:378 / MLPBlock.java:160) when every shard has the batched path; the rest row by row
    int done = 0;
    bool rows_ok = true;
    for (jh_session* s : g->sh) rows_ok = rows_ok && prefill_batch_ok(s);
    while (rows_ok && n - done >= g->sh[0]->prefill_batch_min) {
        const int rows = n - done < PB_MAX_ROWS ? n - done : PB_MAX_ROWS;
        bool fits = true;
        for (jh_session* s : g->sh) fits = fits && (s->strict || prefill_chunk_fits(s, start_pos + done, rows));
        if (!fits) break;
        JHCHK(tp_group_rows(g, tokens + done, rows, start_pos + done));
        done += rows;
    }
    for (int i = done; i < n; i++) {
        for (jh_session* s : g->sh) JHCHK(jh_tp_set_row(s, tokens[i], nullptr, start_pos + i));
        JHCHK(tp_group_layers(g, start_pos + i));
    }
    for (jh_session* s : g->sh) { HIPCHK(hipSetDevice(s->m->device)); HIPCHK(hipStreamSynchronize(s->stream)); }
    return JH_OK;
}
int jh_tp_group_sample(jh_tp_group* g, int32_t* next_token) {
    if (!g || !next_token) return set_err(JH_ERR_INVALID, "tp_group_sample: null");
    return jh_sample(g->sh[0], 0.0f, 0.5f, next_token, nullptr);
}
int jh_tp_group_decode_n(jh_tp_group* g, int32_t first_token, int start_pos, int n, int32_t* out_tokens) {
    if (!g || n <= 0 || start_pos < 0 || !out_tokens) return set_err(JH_ERR_INVALID, "tp_group_decode_n: bad argument");
    const int N = (int)g->sh.size();
    jh_session* s0 = g->sh[0];
    for (jh_session* s : g->sh) {
        if (start_pos + n > s->max_ctx) return set_err(JH_ERR_INVALID, "tp_group_decode_n: positions beyond a shard's max_ctx");
        JHCHK(check_positions(s, start_pos + n - 1));
    }
    if (first_token < 0 || first_token >= s0->m->c.vocab_size) return set_err(JH_ERR_INVALID, "tp_group_decode_n: token id out of range");
    if (!lm_head_weight(s0->m)->data || !s0->m->global_w[JH_W_FINALNORM].data) return set_err(JH_ERR_INVALID, "tp_group_decode_n: shard 0 needs the output weights");
    HIPCHK(hipSetDevice(s0->m->device));
    JHCHK(ensure_out_tokens(s0, n));
    const int E = s0->m->c.embedding_length;
    const int tp_graph = opt_int("JH_TP_GRAPH", 1);
    if (tp_graph && g->graph_ok) {
        // ---- one graph replay per shard and token, the shards meet in kernels (tp_build_graph)
        for (int v = 0; v < N_ATTN_VARIANTS; v++)
            if (attn_variant_in_range(s0, v, start_pos, start_pos + n - 1))
                for (int k = 0; k < N; k++) JHCHK(tp_build_graph(g, k, v));
        for (int k = 0; k < N; k++) {
            jh_session* s = g->sh[k];
            HIPCHK(hipSetDevice(s->m->device));
            HIPCHK(hipStreamSynchronize(s->stream));            // counters below are read on the host
            // the first replay after a capture uploads every shard's graph: its waits are bounded by 2 s instead of 50 ms, so that a
            // slow upload (or a descheduled host thread between the per-shard launches) is not mistaken for a missing peer
            const unsigned bound = g->fresh_graphs ? 200000000u : 0u;
            HIPCHK(hipMemcpy((char*)g->seq[k] + 8, &bound, 4, hipMemcpyHostToDevice));
        }
        for (int k = 0; k < N; k++) {
            jh_session* s = g->sh[k];
            const JWeight& emb = s->m->global_w[JH_W_EMBED];
            HIPCHK(hipSetDevice(s->m->device));
            if (k == 0) {
                hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, s->stream, s->st, start_pos, first_token, 0);
                hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, s->stream, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                                   (const DecodeState*)s->st, E, s->x);
            } else {
                // the first row reaches the other shards through their mailbox, like every later one: seq = this shard's counter
                unsigned cur = 0;
                HIPCHK(hipMemcpy(&cur, g->seq[k], sizeof(cur), hipMemcpyDeviceToHost));
                TPMail m0{first_token, start_pos, cur, 0};
                HIPCHK(hipMemcpy(g->mail[k], &m0, sizeof(m0), hipMemcpyHostToDevice));
                hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, s->stream, s->st, start_pos, first_token, 0);
            }
            HIPCHK(hipGetLastError());
        }
        for (int i = 0; i < n; i++) {
            const int v = attn_variant_for(s0, start_pos + i);
            for (int k = 0; k < N; k++) {
                jh_session* s = g->sh[k];
                HIPCHK(hipSetDevice(s->m->device));
                HIPCHK(hipGraphLaunch(g->exec[v][k], s->stream));
            }
        }
        for (jh_session* s : g->sh) { HIPCHK(hipSetDevice(s->m->device)); HIPCHK(hipStreamSynchronize(s->stream)); }
        bool timed_out = false;
        for (int k = 0; k < N; k++) {      // a wait that timed out (tp_wait_ge): the shards' streams did not run side by side
            unsigned w2[2] = {0, 0};
            HIPCHK(hipSetDevice(g->sh[k]->m->device));
            HIPCHK(hipMemcpy(w2, g->seq[k], sizeof(w2), hipMemcpyDeviceToHost));
            if (w2[1]) timed_out = true;
        }
        if (timed_out) {
            // A wait ran into its bound: the shards' kernels did not run side by side.  Seen when several shards share ONE device
            // and their streams were mapped onto the same hardware queue (the runtime multiplexes streams over GPU_MAX_HW_QUEUES
            // = 4 queues): a spinning kernel then blocks the very kernel it waits for.  With one shard per device every stream has
            // its own queue.  Recover: reset the meeting points, stay on the event-ordered loop for this group, redo the call.
            for (int k = 0; k < N; k++) {
                HIPCHK(hipSetDevice(g->sh[k]->m->device));
                HIPCHK(hipMemset(g->flags[k], 0, 2 * (size_t)N * TP_MAX_FLAGS * 4));
                HIPCHK(hipMemset(g->seq[k], 0, 64));
                HIPCHK(hipMemset(g->mail[k], 0, sizeof(TPMail)));
            }
            g->graph_ok = false;
            g->timeouts++;
            if (opt_int("JH_TP_LOUD", 0))
                return set_err(JH_ERR_HIP, "tp_group_decode_n: a shard waited for a peer that never arrived (streams serialised on one hardware queue?)");
            fprintf(stderr, "[jlama-hip] tensor-parallel group: a meeting timed out (%d so far); this group continues on the event-ordered host loop "
                            "(jh_tp_group_status reports it; JH_TP_LOUD=1 makes it an error)\n", g->timeouts);
            return jh_tp_group_decode_n(g, first_token, start_pos, n, out_tokens);
        }
        g->last_mode = 1;
        g->fresh_graphs = false;
        HIPCHK(hipSetDevice(s0->m->device));
        DecodeState hs;
        HIPCHK(hipMemcpy(&hs, s0->st, sizeof(hs), hipMemcpyDeviceToHost));
        s0->generated = hs.step < n ? hs.step : n;
        HIPCHK(hipMemcpy(out_tokens, s0->out_tokens, (size_t)s0->generated * sizeof(int), hipMemcpyDeviceToHost));
        return JH_OK;
    }
    g->last_mode = 2;
    for (int k = 0; k < N; k++) {   // row of the first token on every shard; shard 0's step counter starts at 0
        jh_session* s = g->sh[k];
        const JWeight& emb = s->m->global_w[JH_W_EMBED];
        HIPCHK(hipSetDevice(s->m->device));
        hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, s->stream, s->st, start_pos, first_token, 0);
        hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, s->stream, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                           (const DecodeState*)s->st, E, s->x);
        HIPCHK(hipGetLastError());
    }
    for (int i = 0; i < n; i++) {
        const int pos = start_pos + i;
        JHCHK(tp_group_layers(g, pos));
        // shard 0 samples (Coordinator.java:184): final norm -> LM head -> argmax; finish_token_kernel advances its state
        // and embeds the next row; the id then travels to the other shards, which embed it themselves
        HIPCHK(hipSetDevice(s0->m->device));
        JHCHK(lmhead_launch(s0, s0->stream));
        JHCHK(finish_launch(s0, s0->stream, 1));
        HIPCHK(hipEventRecord(g->evTok[0], s0->stream));
        for (int k = 1; k < N && i + 1 < n; k++) {
            jh_session* s = g->sh[k];
            const JWeight& emb = s->m->global_w[JH_W_EMBED];
            HIPCHK(hipSetDevice(s->m->device));
            HIPCHK(hipStreamWaitEvent(s->stream, g->evTok[0], 0));
            HIPCHK(hipMemcpyPeerAsync(&s->st->token, s->m->device, &s0->st->token, s0->m->device, sizeof(int), s->stream));
            hipLaunchKernelGGL(set_pos_kernel, dim3(1), dim3(1), 0, s->stream, s->st, pos + 1);
            hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, s->stream, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                               (const DecodeState*)s->st, E, s->x);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(g->evTok[k], s->stream));
        }
        // shard 0 must not overwrite its token word (next finish) before the peers copied it: it waits for their copies
        if (i + 1 < n) {
            HIPCHK(hipSetDevice(s0->m->device));
            for (int k = 1; k < N; k++) HIPCHK(hipStreamWaitEvent(s0->stream, g->evTok[k], 0));
        }
    }
    for (jh_session* s : g->sh) { HIPCHK(hipSetDevice(s->m->device)); HIPCHK(hipStreamSynchronize(s->stream)); }
    HIPCHK(hipSetDevice(s0->m->device));
    // stop tokens (jh_session_set_eos on shard 0): its state froze at the step that sampled one; the other shards ran the queued
    // rows on (their KV tail past the stop is never read again).  Only the ids up to and including the stop token are valid.
    DecodeState hs;
    HIPCHK(hipMemcpy(&hs, s0->st, sizeof(hs), hipMemcpyDeviceToHost));
    s0->generated = hs.step < n ? hs.step : n;
    HIPCHK(hipMemcpy(out_tokens, s0->out_tokens, (size_t)s0->generated * sizeof(int), hipMemcpyDeviceToHost));
    return JH_OK;
}

// ---- The same group with ONE PROCESS PER SHARD (the reference's one-Worker-per-range shape; rank-per-GPU launches): a rank holds
// its shard only and maps the other ranks' slot / flag / mailbox buffers through hipIpc handles the host side exchanges (192
// bytes per rank, any transport: torch.distributed all_gather in jlama_amd/distributed.py).  The token graph is the group's
// (tp_build_graph): partial rows pushed into every rank's slot by the o-proj / down GEMVs, flags, shard-ordered sums, the
// sampled id through mailboxes -- no collective library on the data path, no host inside a token.
int jh_tp_rank_create(jh_session* shard, int rank, int n_ranks, jh_tp_group** out) {
    if (!shard || !out || n_ranks <= 0 || n_ranks > 64 || rank < 0 || rank >= n_ranks) return set_err(JH_ERR_INVALID, "tp_rank_create: bad argument");
    const jh_config& c = shard->m->c;
    if (c.layer_start != 0 || c.layer_end != c.n_layers) return set_err(JH_ERR_INVALID, "tp_rank_create: the shard must hold all layers (head split)");
    if (!shard->m->global_w[JH_W_EMBED].data) return set_err(JH_ERR_INVALID, "tp_rank_create: every shard needs the embedding table");
    HIPCHK(hipSetDevice(shard->m->device));
    jh_tp_group* g = new jh_tp_group();
    const int N = n_ranks, k = rank;
    const size_t E = (size_t)c.embedding_length;
    g->local = k;
    g->nwg = (int)((E + 255) / 256);
    g->sh.assign(N, nullptr); g->part.assign(N, nullptr); g->red.assign(N, nullptr); g->slots.assign(N, nullptr);
    g->peers.assign(N, nullptr); g->flags.assign(N, nullptr); g->peers_f.assign(N, nullptr); g->mail.assign(N, nullptr); g->seq.assign(N, nullptr);
    g->slots_of.assign(N, nullptr); g->flags_of.assign(N, nullptr); g->mail_of.assign(N, nullptr);
    for (int v = 0; v < N_ATTN_VARIANTS; v++) { g->graph[v].assign(N, nullptr); g->exec[v].assign(N, nullptr); }
    g->sh[k] = shard;
    // the buffers other ranks' kernels store into: fine-grained like the one-process group's (what RCCL shares over IPC too)
    const bool ok = hipMalloc(&g->part[k], E * 4) == hipSuccess && tp_shared_malloc((void**)&g->slots[k], 2 * (size_t)N * E * 4) == hipSuccess &&
                    tp_shared_malloc((void**)&g->flags[k], 2 * (size_t)N * TP_MAX_FLAGS * 4) == hipSuccess &&
                    hipMemset(g->flags[k], 0, 2 * (size_t)N * TP_MAX_FLAGS * 4) == hipSuccess &&
                    tp_shared_malloc((void**)&g->mail[k], 4096) == hipSuccess && hipMemset(g->mail[k], 0, 4096) == hipSuccess &&
                    hipMalloc(&g->seq[k], 64) == hipSuccess && hipMemset(g->seq[k], 0, 64) == hipSuccess &&
                    hipMalloc(&g->peers[k], 2 * (size_t)N * sizeof(float*)) == hipSuccess &&
                    hipMalloc(&g->peers_f[k], 2 * (size_t)N * sizeof(unsigned*)) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); jh_tp_group_destroy(g); return set_err(JH_ERR_OOM, "tp_rank_create: buffers"); }
    g->slots_of[k] = g->slots[k]; g->flags_of[k] = g->flags[k]; g->mail_of[k] = g->mail[k];
    {   // waits of a rank are bounded by seconds, not the group's 50 ms: the ranks start their graphs independently
        const unsigned bound = 300000000u;   // 3 s of the 100 MHz wall clock
        HIPCHK(hipMemcpy((char*)g->seq[k] + 8, &bound, 4, hipMemcpyHostToDevice));
    }
    HIPCHK(hipDeviceSynchronize());
    *out = g;
    return JH_OK;
}
int jh_tp_rank_handles(jh_tp_group* g, void* out192) {
    if (!g || g->local < 0 || !out192) return set_err(JH_ERR_INVALID, "tp_rank_handles: bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
    const int k = g->local;
    HIPCHK(hipSetDevice(g->sh[k]->m->device));
    hipIpcMemHandle_t h[3];
    HIPCHK(hipIpcGetMemHandle(&h[0], g->slots[k]));
    HIPCHK(hipIpcGetMemHandle(&h[1], g->flags[k]));
    HIPCHK(hipIpcGetMemHandle(&h[2], g->mail[k]));
    memcpy(out192, h, sizeof(h));
    return JH_OK;
}
int jh_tp_rank_connect(jh_tp_group* g, const void* all_handles) {
    if (!g || g->local < 0 || !all_handles) return set_err(JH_ERR_INVALID, "tp_rank_connect: bad argument");
    if (g->connected) return set_err(JH_ERR_INVALID, "tp_rank_connect: already connected");
    const int N = (int)g->sh.size(), k = g->local;
    const size_t E = (size_t)g->sh[k]->m->c.embedding_length;
    HIPCHK(hipSetDevice(g->sh[k]->m->device));
    const hipIpcMemHandle_t* h = (const hipIpcMemHandle_t*)all_handles;
    for (int j = 0; j < N; j++) {
        if (j == k) continue;
        void* p[3] = {nullptr, nullptr, nullptr};
        for (int i = 0; i < 3; i++) {
            const hipError_t e = hipIpcOpenMemHandle(&p[i], h[3 * j + i], hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { (void)hipGetLastError(); return set_err(JH_ERR_HIP, std::string("tp_rank_connect: hipIpcOpenMemHandle: ") + hipGetErrorString(e)); }
            g->ipc_open.push_back(p[i]);
        }
        g->slots_of[j] = (float*)p[0]; g->flags_of[j] = (unsigned*)p[1]; g->mail_of[j] = (TPMail*)p[2];
    }
    std::vector<float*> hs(2 * (size_t)N);     // this shard's slot on shard j, round r:  slots_of[j] + (r*N + k)*E
    std::vector<unsigned*> hf(2 * (size_t)N);
    for (int r = 0; r < 2; r++)
        for (int j = 0; j < N; j++) {
            hs[(size_t)r * N + j] = g->slots_of[j] + ((size_t)r * N + k) * E;
            hf[(size_t)r * N + j] = g->flags_of[j] + ((size_t)r * N + k) * TP_MAX_FLAGS;
        }
    HIPCHK(hipMemcpy(g->peers[k], hs.data(), hs.size() * sizeof(float*), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(g->peers_f[k], hf.data(), hf.size() * sizeof(unsigned*), hipMemcpyHostToDevice));
    if (k == 0 && N > 1) {
        std::vector<TPMail*> hm;
        for (int j = 1; j < N; j++) hm.push_back(g->mail_of[j]);
        HIPCHK(hipMalloc(&g->mails_dev, hm.size() * sizeof(TPMail*)));
        HIPCHK(hipMemcpy(g->mails_dev, hm.data(), hm.size() * sizeof(TPMail*), hipMemcpyHostToDevice));
    }
    g->connected = true;
    return JH_OK;
}
// n greedy steps from the row every rank holds (s->x is NOT used: rank 0 embeds first_token, the others receive it through their
// mailbox like every later id).  Every rank calls it with the same arguments; out_tokens (HOST [n]) is filled on rank 0 only.
int jh_tp_rank_decode_n(jh_tp_group* g, int32_t first_token, int start_pos, int n, int32_t* out_tokens) {
    if (!g || g->local < 0 || n <= 0 || start_pos < 0) return set_err(JH_ERR_INVALID, "tp_rank_decode_n: bad argument");
    const int N = (int)g->sh.size(), k = g->local;
    jh_session* s = g->sh[k];
    if (k == 0 && !out_tokens) return set_err(JH_ERR_INVALID, "tp_rank_decode_n: rank 0 needs out_tokens");
    if (start_pos + n > s->max_ctx) return set_err(JH_ERR_INVALID, "tp_rank_decode_n: positions beyond max_ctx");
    JHCHK(check_positions(s, start_pos + n - 1));
    if (first_token < 0 || first_token >= s->m->c.vocab_size) return set_err(JH_ERR_INVALID, "tp_rank_decode_n: token id out of range");
    if (k == 0 && (!lm_head_weight(s->m)->data || !s->m->global_w[JH_W_FINALNORM].data)) return set_err(JH_ERR_INVALID, "tp_rank_decode_n: rank 0 needs the output weights");
    if (!g->connected) return set_err(JH_ERR_INVALID, "tp_rank_decode_n: jh_tp_rank_connect has not been called");
    HIPCHK(hipSetDevice(s->m->device));
    if (k == 0) JHCHK(ensure_out_tokens(s, n));
    for (int v = 0; v < N_ATTN_VARIANTS; v++)
        if (attn_variant_in_range(s, v, start_pos, start_pos + n - 1)) JHCHK(tp_build_graph(g, k, v));
    HIPCHK(hipStreamSynchronize(s->stream));
    const JWeight& emb = s->m->global_w[JH_W_EMBED];
    const int E = s->m->c.embedding_length;
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, s->stream, s->st, start_pos, first_token, 0);
    if (k == 0) {
        hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, s->stream, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                           (const DecodeState*)s->st, E, s->x);
    } else {
        unsigned cur = 0;
        HIPCHK(hipMemcpy(&cur, g->seq[k], sizeof(cur), hipMemcpyDeviceToHost));
        TPMail m0{first_token, start_pos, cur, 0};
        HIPCHK(hipMemcpy(g->mail[k], &m0, sizeof(m0), hipMemcpyHostToDevice));
    }
    HIPCHK(hipGetLastError());
    for (int i = 0; i < n; i++) HIPCHK(hipGraphLaunch(g->exec[attn_variant_for(s, start_pos + i)][k], s->stream));
    // A rank k > 0 meets rank 0 for the last time in the final layer's meeting, BEFORE rank 0's LM head / finish / publish of the
    // last token: without this wait it could return, and a later call could write its mailbox {first_token, seq} from the host
    // while that publish (same sequence number) is still in flight and then overwrites token and position.  One more wait on the
    // mailbox for the sequence number of that publish (this rank's counter after its n bumps): the call returns only once it landed.
    if (k > 0) hipLaunchKernelGGL(tp_wait_token_kernel, dim3(1), dim3(1), 0, s->stream, (const TPMail*)g->mail[k], g->seq[k], s->st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s->stream));
    unsigned w2[2] = {0, 0};
    HIPCHK(hipMemcpy(w2, g->seq[k], sizeof(w2), hipMemcpyDeviceToHost));
    if (w2[1]) {
        // sequence numbers, flags and mailboxes have diverged across the ranks: clearing only the error word would leave every later
        // call broken in silence.  The group is dead until it is re-created and re-connected.
        g->connected = false;
        g->timeouts++;
        return set_err(JH_ERR_HIP, "tp_rank_decode_n: this rank waited for a peer that never arrived (is every rank decoding the same steps?); "
                                   "the group is disconnected -- destroy and re-create it on every rank");
    }
    if (k == 0) {
        DecodeState hs;
        HIPCHK(hipMemcpy(&hs, s->st, sizeof(hs), hipMemcpyDeviceToHost));
        s->generated = hs.step < n ? hs.step : n;
        HIPCHK(hipMemcpy(out_tokens, s->out_tokens, (size_t)s->generated * sizeof(int), hipMemcpyDeviceToHost));
    }
    return JH_OK;
}

int jh_tp_group_status(jh_tp_group* g, int32_t* out, int n) {
    if (!g || !out || n <= 0) return set_err(JH_ERR_INVALID, "tp_group_status: bad argument");
    const int32_t v[6] = {g->last_mode, g->timeouts, g->graph_ok ? 1 : 0, g->fused_push, g->flags_per_launch, g->connected ? 1 : 0};
    for (int i = 0; i < n && i < 6; i++) out[i] = v[i];
    return 6;
}
// What the meeting protocol of a rank depends on besides the model shape: every rank must poll the flag count the producers raise
// and agree on push vs scatter mode -- both follow from (kernel family, CU count, push option, shard shape), which this folds into
// one word.  The host compares the words of all ranks before the first jh_tp_rank_decode_n (distributed.tp_generate_ipc does).
int jh_tp_rank_signature(jh_tp_group* g, int64_t* out) {
    if (!g || g->local < 0 || !out) return set_err(JH_ERR_INVALID, "tp_rank_signature: bad argument");
    const jh_session* s = g->sh[g->local];
    const jh_config& c = s->m->c;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h ^= v; h *= 1099511628211ull; };
    mix((uint64_t)g_cu_count); mix((uint64_t)s->strict); mix((uint64_t)opt_int("JH_TP_FUSE", 1)); mix((uint64_t)c.weight_dtype);
    mix((uint64_t)c.embedding_length); mix((uint64_t)c.n_layers); mix((uint64_t)(c.n_heads * c.head_size)); mix((uint64_t)c.hidden_length);
    mix((uint64_t)g->sh.size());
    // what shapes the o-proj / down launches, i.e. the number of flag words a producer raises and a consumer polls: the launch
    // planners' inputs (explicit options land in these), the kernel family switches, and -- once the token graphs exist -- the
    // planned flag counts themselves
    for (const LaunchCfg* lc : {&s->cfg_o, &s->cfg_down}) { mix((uint64_t)lc->R); mix((uint64_t)lc->waves); mix((uint64_t)lc->grid_cap); mix((uint64_t)lc->pipe); }
    mix((uint64_t)s->p16_depth); mix((uint64_t)opt_int("JH_T16", 1));
    mix((uint64_t)(g->plan_flags[0] + 1)); mix((uint64_t)(g->plan_flags[1] + 1));
    *out = (int64_t)(h & 0x7fffffffffffffffull);
    return JH_OK;
}


}  // extern "C"
