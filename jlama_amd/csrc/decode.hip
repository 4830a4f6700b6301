// decode.hip -- part of libjlamahip.so (C ABI: include/jlama_hip.h).  forward / sample / the device decode loop: captured token graphs, stop tokens, temperature sampling.
#include "jh_host.h"
#include "jh_launch.h"

// The RoPE row of kv head h at position p is table row p + 2*h (global head index): the reference's table has
// context_length rows and Java throws ArrayIndexOutOfBoundsException beyond it -- same positions refused here.
int check_positions(const jh_session* s, int last_pos) {
    const jh_model* m = s->m;
    const long long last_row = (long long)last_pos + 2LL * (m->kv_head_offset + m->c.n_kv_heads - 1);
    if (last_row >= (long long)m->c.context_length)
        return set_err(JH_ERR_INVALID, "position " + std::to_string(last_pos) + ": RoPE row position + 2*(kvHeads-1) is beyond the model's context_length (the reference's table ends there)");
    return JH_OK;
}

extern "C" {

// captured graphs hold raw device pointers of the weights: drop them all if a weight was replaced since the capture
void drop_stale_graphs(jh_session* s) {
    if (s->graphs_version == s->m->weights_version) return;
    for (int v = 0; v < N_ATTN_VARIANTS; v++) {
        if (s->exec_s[v]) { hipGraphExecDestroy(s->exec_s[v]); s->exec_s[v] = nullptr; }
        if (s->graph_s[v]) { hipGraphDestroy(s->graph_s[v]); s->graph_s[v] = nullptr; }
        if (s->exec[v]) { hipGraphExecDestroy(s->exec[v]); s->exec[v] = nullptr; }
        if (s->exec_m[v]) { hipGraphExecDestroy(s->exec_m[v]); s->exec_m[v] = nullptr; hipGraphDestroy(s->graph_m[v]); s->graph_m[v] = nullptr; }
        if (s->graph[v]) { hipGraphDestroy(s->graph[v]); s->graph[v] = nullptr; }
        if (s->row_exec[v]) { hipGraphExecDestroy(s->row_exec[v]); s->row_exec[v] = nullptr; }
        if (s->row_graph[v]) { hipGraphDestroy(s->row_graph[v]); s->row_graph[v] = nullptr; }
    }
    for (auto& kv : s->pb_graphs) hipGraphExecDestroy(kv.second);
    s->pb_graphs.clear();
    for (hipGraph_t g : s->pb_graph_src) hipGraphDestroy(g);
    s->pb_graph_src.clear();
    s->graphs_version = s->m->weights_version;
}
// which attention variant serves position pos: slices of <= 32 rows need only 2 prefetched row steps
int attn_variant_for(const jh_session* s, int pos) {
    if (s->direct_max == 0 && pos + 1 <= s->max_splits * 32) return 1;
    if (s->long_splits > 0 && pos + 1 > s->long_min) return 2;
    return 0;
}
// does any position of [first, last] use attention variant v?  (variants change at most twice along the context)
bool attn_variant_in_range(const jh_session* s, int v, int first, int last) {
    const int edges[4] = {first, last, s->max_splits * 32, s->long_min};   // positions next to the two thresholds
    for (int e : edges)
        for (int d = -1; d <= 1; d++) {
            const int pos = e + d;
            if (pos >= first && pos <= last && attn_variant_for(s, pos) == v) return true;
        }
    return false;
}
int build_row_graph(jh_session* s, int v) {
    drop_stale_graphs(s);
    JHCHK(ensure_strict_operands(s, s->stream));
    if (s->row_exec[v]) return JH_OK;
    s->attn_variant = v;
    hipStream_t st = s->stream;
    std::lock_guard<std::mutex> cap(g_capture_mu);
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = layers_launch(s, st, 0);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != JH_OK) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return set_err(JH_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    s->row_graph[v] = g;
    HIPCHK(hipGraphInstantiate(&s->row_exec[v], g, nullptr, nullptr, 0));
    return JH_OK;
}
int forward_impl(jh_session* s, const int32_t* tokens, const float* x_in, bool x_in_dev, int n, int start_pos,
                        float* x_out, bool x_out_dev) {
    if (!s || n <= 0 || start_pos < 0 || (!tokens && !x_in)) return set_err(JH_ERR_INVALID, "forward: bad argument");
    if (start_pos + n > s->max_ctx) return set_err(JH_ERR_INVALID, "forward: position beyond the session's max_ctx");
    JHCHK(check_positions(s, start_pos + n - 1));
    jh_model* m = s->m;
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = s->stream;
    JHCHK(ensure_strict_operands(s, st));   // (no-op unless a reference-order session is missing an operand copy)
    const int E = m->c.embedding_length;
    const JWeight& emb = m->global_w[JH_W_EMBED];
    if (tokens && !emb.data) return set_err(JH_ERR_INVALID, "forward: this shard has no embedding table");
    if (tokens)
        for (int i = 0; i < n; i++)
            if (tokens[i] < 0 || tokens[i] >= m->c.vocab_size) return set_err(JH_ERR_INVALID, "forward: token id out of range");
    // Chunks of >= prefill_batch_min rows take the batched path (MFMA GEMMs over all rows, AbstractModel.java:295-312);
    // the rest -- and every call while a tap layer is set -- goes one position at a time (batchForwardSlow order,
    // :282-290; per-row arithmetic is the same, attention is per position there too, CausalSelfAttention.java:199).
    int done = 0;
    if (prefill_batch_ok(s)) {
        while (n - done >= s->prefill_batch_min) {
            const int rows = n - done < PB_MAX_ROWS ? n - done : PB_MAX_ROWS;
            if (!s->strict && !prefill_chunk_fits(s, start_pos + done, rows)) break;
            JHCHK(prefill_chunk(s, tokens ? tokens + done : nullptr, x_in ? x_in + (size_t)done * E : nullptr, x_in_dev, rows,
                                start_pos + done, x_out ? x_out + (size_t)done * E : nullptr, x_out_dev, st));
            done += rows;
        }
    }
    for (int i = done; i < n; i++) {
        hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, s->st, start_pos + i, tokens ? tokens[i] : 0, 0);
        if (tokens) {
            hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, st, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                               (const DecodeState*)s->st, E, s->x);
        } else {
            HIPCHK(hipMemcpyAsync(s->x, x_in + (size_t)i * E, (size_t)E * 4, x_in_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        }
        HIPCHK(hipGetLastError());
        if (s->tap_layer < 0 && !opt_int("JH_NO_GRAPH", 0)) {
            // the layers read the position from the device-resident state, so ONE captured graph serves every row:
            // a pipeline stage pays 1 launch per tick instead of 5 per layer
            const int v = attn_variant_for(s, start_pos + i);
            JHCHK(build_row_graph(s, v));
            HIPCHK(hipGraphLaunch(s->row_exec[v], st));
        } else {
            s->attn_variant = attn_variant_for(s, start_pos + i);
            JHCHK(layers_launch(s, st, start_pos + i));
        }
        if (x_out)
            HIPCHK(hipMemcpyAsync(x_out + (size_t)i * E, s->x, (size_t)E * 4, x_out_dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    }
    if (!x_out_dev) HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_forward(jh_session* s, const int32_t* tokens, const float* x_in, int n, int start_pos, float* x_out) {
    return forward_impl(s, tokens, x_in, false, n, start_pos, x_out, false);
}
int jh_forward_device(jh_session* s, const int32_t* tokens, const float* x_in_dev, int n, int start_pos, float* x_out_dev) {
    return forward_impl(s, tokens, x_in_dev, true, n, start_pos, x_out_dev, true);
}

int jh_session_get_row(jh_session* s, float* out, int to_device) {
    if (!s || !out) return set_err(JH_ERR_INVALID, "get_row: null");
    HIPCHK(hipSetDevice(s->m->device));
    HIPCHK(hipMemcpyAsync(out, s->x, (size_t)s->m->c.embedding_length * 4, to_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s->stream));
    if (!to_device) HIPCHK(hipStreamSynchronize(s->stream));
    return JH_OK;
}

int jh_get_logits(jh_session* s, float* out_v) {
    if (!s || !out_v) return set_err(JH_ERR_INVALID, "get_logits: null");
    HIPCHK(hipSetDevice(s->m->device));
    HIPCHK(hipMemcpyAsync(out_v, s->logits, (size_t)s->m->c.vocab_size * 4, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return JH_OK;
}

int jh_sample(jh_session* s, float temperature, float u, int32_t* next_token, float* logits_out) {
    if (!s || !next_token) return set_err(JH_ERR_INVALID, "sample: null");
    HIPCHK(hipSetDevice(s->m->device));
    hipStream_t st = s->stream;
    JHCHK(lmhead_launch(s, st));
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, s->st, 0, 0, 0);
    JHCHK(finish_launch(s, st, 0));
    int tok = 0;
    HIPCHK(hipMemcpyAsync(&tok, s->out_tokens, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    const int V = s->m->c.vocab_size;
    if (logits_out || temperature != 0.0f) {
        std::vector<float> tmp;
        float* lg = logits_out;
        if (!lg) { tmp.resize((size_t)V); lg = tmp.data(); }
        JHCHK(jh_get_logits(s, lg));
        if (temperature != 0.0f) {
            // AbstractModel.java:475-489 (host side: the uniform comes from the caller; sequential float sums)
            std::vector<float> pr((size_t)V);
            const double maxv = (double)lg[tok];
            float sum = 0;
            for (int i = 0; i < V; i++) {
                const float v = (float)exp(((double)lg[i] - maxv) / (double)temperature);
                sum += v;
                pr[(size_t)i] = v;
            }
            float acc = 0;
            int pick = V - 1;
            for (int i = 0; i < V; i++) {
                acc += pr[(size_t)i] / sum;
                if (acc >= u) { pick = i; break; }
            }
            tok = pick;
        }
    }
    *next_token = tok;
    return JH_OK;
}

int jh_decode_step(jh_session* s, int32_t token, int pos, int32_t* next_token) {
    if (!s || !next_token) return set_err(JH_ERR_INVALID, "decode_step: null");
    JHCHK(jh_forward(s, &token, nullptr, 1, pos, nullptr));
    return jh_sample(s, 0.0f, 0.5f, next_token, nullptr);
}

int build_graph(jh_session* s, int v, float temperature) {
    drop_stale_graphs(s);
    JHCHK(ensure_strict_operands(s, s->stream));
    const bool sampled = temperature != 0.0f;
    if (sampled && s->sampled_temp != temperature) {   // the temperature is a kernel argument of the captured graphs
        for (int vv = 0; vv < N_ATTN_VARIANTS; vv++)
            if (s->exec_s[vv]) { hipGraphExecDestroy(s->exec_s[vv]); s->exec_s[vv] = nullptr; hipGraphDestroy(s->graph_s[vv]); s->graph_s[vv] = nullptr; }
        s->sampled_temp = temperature;
    }
    if (sampled ? s->exec_s[v] != nullptr : s->exec[v] != nullptr) return JH_OK;
    s->attn_variant = v;
    hipStream_t st = s->stream;
    const jh_config& c = s->m->c;
    const int saved_tap = s->tap_layer;
    s->tap_layer = -1;
    std::lock_guard<std::mutex> cap(g_capture_mu);
    HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = layers_launch(s, st, 0);
    const bool has_out = lm_head_weight(s->m)->data && s->m->global_w[JH_W_FINALNORM].data;
    if (rc == JH_OK && has_out) rc = lmhead_launch(s, st);
    if (rc == JH_OK && has_out) rc = finish_launch(s, st, 1, temperature);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    s->tap_layer = saved_tap;
    if (rc != JH_OK) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return set_err(JH_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    if (sampled) {
        s->graph_s[v] = g;
        HIPCHK(hipGraphInstantiate(&s->exec_s[v], g, nullptr, nullptr, 0));
        return JH_OK;
    }
    s->graph[v] = g;
    HIPCHK(hipGraphInstantiate(&s->exec[v], g, nullptr, nullptr, 0));
    size_t n_nodes = 0;                                   // what the graph actually holds (kernel nodes; no memcpy / memset nodes are captured)
    HIPCHK(hipGraphGetNodes(g, nullptr, &n_nodes));
    s->kernels_per_token = (int)n_nodes;
    // the same token tokens_per_graph times in ONE graph (greedy loop only)
    // The multi-token graph is an optimisation on top of a working single-token graph: if its capture or instantiation fails the
    // session keeps decoding one token per launch (tokens_per_graph = 1) instead of failing decode_n.
    if (has_out && s->tokens_per_graph > 1 && !s->exec_m[v]) {
        struct TapGuard { jh_session* s; int v; ~TapGuard() { s->tap_layer = v; } } restore{s, saved_tap};   // every exit path
        s->tap_layer = -1;
        hipGraph_t gm = nullptr;
        hipGraphExec_t xm = nullptr;
        bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            rc = JH_OK;
            for (int t = 0; t < s->tokens_per_graph && rc == JH_OK; t++) {
                rc = layers_launch(s, st, 0);
                if (rc == JH_OK) rc = lmhead_launch(s, st);
                if (rc == JH_OK) rc = finish_launch(s, st, 1, 0.0f);
            }
            ok = hipStreamEndCapture(st, &gm) == hipSuccess && rc == JH_OK && gm != nullptr;
            if (ok) ok = hipGraphInstantiate(&xm, gm, nullptr, nullptr, 0) == hipSuccess;
        }
        if (ok) {
            s->graph_m[v] = gm;
            s->exec_m[v] = xm;
        } else {
            if (gm) hipGraphDestroy(gm);
            (void)hipGetLastError();
            s->tokens_per_graph = 1;
            fprintf(stderr, "[jh] the %d-tokens-per-launch graph could not be built (%s): decoding one token per launch\n", s->tokens_per_graph,
                    rc != JH_OK ? g_err.c_str() : "capture / instantiate failed");
        }
    }
    return JH_OK;
}

int decode_n_async_impl(jh_session* s, int32_t first_token, int start_pos, int n, float temperature, const float* u);
int jh_decode_n_async(jh_session* s, int32_t first_token, int start_pos, int n) {
    return decode_n_async_impl(s, first_token, start_pos, n, 0.0f, nullptr);
}
// The same loop with AbstractModel.sample's temperature branch on the device: u[i] is the uniform of the i-th sampled token (the
// reference draws ThreadLocalRandom.nextFloat() per call, AbstractModel.java:594 -- not seedable, hence the caller's array).
int jh_decode_n_sampled(jh_session* s, int32_t first_token, int start_pos, int n, float temperature, const float* u, int32_t* out_tokens) {
    if (temperature != 0.0f && !u) return set_err(JH_ERR_INVALID, "decode_n_sampled: temperature > 0 needs n uniforms");
    JHCHK(decode_n_async_impl(s, first_token, start_pos, n, temperature, u));
    return jh_decode_wait(s, out_tokens, n);
}
int decode_n_async_impl(jh_session* s, int32_t first_token, int start_pos, int n, float temperature, const float* u) {
    if (!s || n <= 0 || start_pos < 0) return set_err(JH_ERR_INVALID, "decode_n: bad argument");
    if (start_pos + n > s->max_ctx) return set_err(JH_ERR_INVALID, "decode_n: positions beyond the session's max_ctx");
    JHCHK(check_positions(s, start_pos + n - 1));
    jh_model* m = s->m;
    HIPCHK(hipSetDevice(m->device));
    const JWeight& emb = m->global_w[JH_W_EMBED];
    if (!emb.data || !lm_head_weight(m)->data) return set_err(JH_ERR_INVALID, "decode_n: needs embedding and output weights on this shard");
    if (first_token < 0 || first_token >= m->c.vocab_size) return set_err(JH_ERR_INVALID, "decode_n: token id out of range");
    JHCHK(ensure_out_tokens(s, n));
    hipStream_t st = s->stream;
    const bool sampled = temperature != 0.0f;
    if (sampled) {
        if (!s->prob) HIPCHK(hipMalloc(&s->prob, ((size_t)m->c.vocab_size + 8) * 4));   // exponentials, then their float sum
        if (!s->pick) HIPCHK(hipMalloc(&s->pick, 64));
        if (s->u_cap < n) {
            HIPCHK(hipStreamSynchronize(st));
            if (s->u_dev) HIPCHK(hipFree(s->u_dev));
            HIPCHK(hipMalloc(&s->u_dev, (size_t)n * 4));
            s->u_cap = n;
            for (int v = 0; v < N_ATTN_VARIANTS; v++)   // the buffer's address is baked into the sampled graphs
                if (s->exec_s[v]) { hipGraphExecDestroy(s->exec_s[v]); s->exec_s[v] = nullptr; hipGraphDestroy(s->graph_s[v]); s->graph_s[v] = nullptr; }
        }
        HIPCHK(hipMemcpyAsync(s->u_dev, u, (size_t)n * 4, hipMemcpyHostToDevice, st));
    }
    const bool use_graph = !opt_int("JH_NO_GRAPH", 0);
    if (use_graph) {   // capture the graph variants this call needs before the timed region (a capture costs milliseconds)
        for (int v = 0; v < N_ATTN_VARIANTS; v++)
            if (attn_variant_in_range(s, v, start_pos, start_pos + n - 1)) JHCHK(build_graph(s, v, temperature));
    }
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, s->st, start_pos, first_token, 0);
    hipLaunchKernelGGL(embed_kernel, dim3(1), dim3(256), 0, st, (const void*)emb.data, (const float*)emb.scales, emb.dtype,
                       (const DecodeState*)s->st, m->c.embedding_length, s->x);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(s->ev0, st));
    // Stop tokens: the device freezes its state at the step that samples one (finish_token_kernel); the host keeps two
    // chunks of replays queued and, before queueing a third, looks at the state snapshot taken after the first -- the GPU
    // never idles and at most 2*EOS_CHUNK steps are replayed for nothing.
    constexpr int EOS_CHUNK = 16;
    int launched = 0, chunk = 0;
    const int T = s->tokens_per_graph;                       // (a divisor of EOS_CHUNK: a multi-token launch never straddles a snapshot)
    for (int i = 0; i < n; i++) {
        const int v = attn_variant_for(s, start_pos + i);   // the host knows every token's position in advance
        if (use_graph) {
            JHCHK(build_graph(s, v, temperature));
            if (!sampled && T > 1 && s->exec_m[v] && i + T <= n && launched % T == 0 && attn_variant_for(s, start_pos + i + T - 1) == v) {   // (each variant is one range of positions)
                HIPCHK(hipGraphLaunch(s->exec_m[v], st));   // T tokens, one launch
                i += T - 1;
                launched += T - 1;
            } else {
                HIPCHK(hipGraphLaunch(sampled ? s->exec_s[v] : s->exec[v], st));
            }
        } else {
            const int saved = s->tap_layer;
            s->tap_layer = -1;
            s->attn_variant = v;
            int rc = layers_launch(s, st, 0);
            if (rc == JH_OK) rc = lmhead_launch(s, st);
            if (rc == JH_OK) rc = finish_launch(s, st, 1, temperature);
            s->tap_layer = saved;
            JHCHK(rc);
        }
        launched++;
        if (s->n_eos > 0 && launched % EOS_CHUNK == 0 && i + 1 < n) {
            const int slot = chunk & 1;
            if (chunk >= 2) {   // snapshot taken two chunks ago lives in this slot
                HIPCHK(hipEventSynchronize(s->ev_chunk[slot]));
                if (s->st_host[slot].done) break;
            }
            HIPCHK(hipMemcpyAsync(&s->st_host[slot], s->st, sizeof(DecodeState), hipMemcpyDeviceToHost, st));
            HIPCHK(hipEventRecord(s->ev_chunk[slot], st));
            chunk++;
        }
    }
    HIPCHK(hipEventRecord(s->ev1, st));
    s->pending_n = launched;
    return JH_OK;
}
int jh_decode_wait(jh_session* s, int32_t* out_tokens, int n) {
    if (!s) return set_err(JH_ERR_INVALID, "decode_wait: null");
    HIPCHK(hipSetDevice(s->m->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->pending_n > 0) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        s->ms_per_token = (double)ms / s->pending_n;
        DecodeState hs;
        HIPCHK(hipMemcpy(&hs, s->st, sizeof(hs), hipMemcpyDeviceToHost));
        s->generated = hs.step < s->pending_n ? hs.step : s->pending_n;   // fewer than queued only after a stop token
    }
    if (out_tokens && n > 0) {
        if (n > s->generated) n = s->generated;
        if (n > 0) HIPCHK(hipMemcpy(out_tokens, s->out_tokens, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    }
    s->pending_n = 0;
    return JH_OK;
}
int jh_decode_n(jh_session* s, int32_t first_token, int start_pos, int n, int32_t* out_tokens) {
    JHCHK(jh_decode_n_async(s, first_token, start_pos, n));
    return jh_decode_wait(s, out_tokens, n);
}
int jh_decode_stats(jh_session* s, double* ms_per_token, int32_t* kernels_per_token) {
    if (!s) return set_err(JH_ERR_INVALID, "decode_stats: null");
    if (ms_per_token) *ms_per_token = s->ms_per_token;
    if (kernels_per_token) *kernels_per_token = s->kernels_per_token;
    return JH_OK;
}
int jh_set_tap_layer(jh_session* s, int layer) {
    if (!s) return set_err(JH_ERR_INVALID, "set_tap_layer: null");
    s->tap_layer = layer;
    return JH_OK;
}
int jh_get_tap(jh_session* s, int which, float* out, int n) {
    if (!s || !out || which < 0 || which >= TAP_SLOTS || !s->taps[which]) return set_err(JH_ERR_INVALID, "get_tap: not recorded");
    HIPCHK(hipSetDevice(s->m->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    const int m = s->tap_len[which] < n ? s->tap_len[which] : n;
    HIPCHK(hipMemcpy(out, s->taps[which], (size_t)m * 4, hipMemcpyDeviceToHost));
    return m;
}


}  // extern "C"
