/*
 * Panama FFM bindings of libjlamahip.so, Tier 2: the resident model / session / multi-GPU entry points of include/jlama_hip.h
 * (what LlamaModel delegates to so that weights, activations, KV pages and the greedy loop stay in HBM -- INTEGRATION.md section 3).
 * Same shape as NativeHip.java (Tier 1) and as jextract output; every C parameter is int / long / float / pointer and
 * jh_config is passed by pointer (a MemorySegment laid out as JhConfig.LAYOUT below), so there are no structs by value.
 * tests/test_java_binding.py checks every descriptor here against the header (argument count and kinds).
 */
package com.github.tjake.jlama.tensor.operations.cnative;

import static java.lang.foreign.ValueLayout.ADDRESS;
import static java.lang.foreign.ValueLayout.JAVA_FLOAT;
import static java.lang.foreign.ValueLayout.JAVA_INT;
import static java.lang.foreign.ValueLayout.JAVA_LONG;

import java.lang.foreign.FunctionDescriptor;
import java.lang.foreign.Linker;
import java.lang.foreign.MemoryLayout;
import java.lang.foreign.MemorySegment;
import java.lang.foreign.SymbolLookup;
import java.lang.invoke.MethodHandle;

public final class NativeHipModel {
    private NativeHipModel() {}

    private static final Linker LINKER = Linker.nativeLinker();
    private static final SymbolLookup LOOKUP = SymbolLookup.loaderLookup().or(LINKER.defaultLookup());

    private static MethodHandle h(String name, MemoryLayout res, MemoryLayout... args) {
        MemorySegment sym = LOOKUP.find(name).orElseThrow(() -> new UnsatisfiedLinkError("unresolved symbol: " + name));
        return LINKER.downcallHandle(sym, res == null ? FunctionDescriptor.ofVoid(args) : FunctionDescriptor.of(res, args));
    }

    private static MemoryLayout[] sig(String s) {
        // i = int, l = long, f = float, p = pointer
        MemoryLayout[] out = new MemoryLayout[s.length()];
        for (int k = 0; k < s.length(); k++) {
            switch (s.charAt(k)) {
                case 'i': out[k] = JAVA_INT; break;
                case 'l': out[k] = JAVA_LONG; break;
                case 'f': out[k] = JAVA_FLOAT; break;
                default: out[k] = ADDRESS;
            }
        }
        return out;
    }

    private static RuntimeException rethrow(Throwable t) {
        if (t instanceof RuntimeException) return (RuntimeException) t;
        if (t instanceof Error) throw (Error) t;
        return new AssertionError("should not reach here", t);
    }

    // define JH_TAP_POST_FF_RES 11

int jh_model_create(const jh_config* cfg, jh_model** out)
    private static final MethodHandle jh_model_create = h("jh_model_create", JAVA_INT, sig("pp"));
    // int jh_model_destroy(jh_model* m)
    private static final MethodHandle jh_model_destroy = h("jh_model_destroy", JAVA_INT, sig("p"));
    // int jh_model_set_weight(jh_model* m, int layer, int which, int dtype, const void* data, const float* scales, int rows, int cols, int from_device)
    private static final MethodHandle jh_model_set_weight = h("jh_model_set_weight", JAVA_INT, sig("piiippiii"));
    // int jh_model_set_kv_head_offset(jh_model* m, int kv_head_offset)
    private static final MethodHandle jh_model_set_kv_head_offset = h("jh_model_set_kv_head_offset", JAVA_INT, sig("pi"));
    // int jh_session_create(jh_model* m, int max_ctx, int64_t max_page_bytes, jh_session** out)
    private static final MethodHandle jh_session_create = h("jh_session_create", JAVA_INT, sig("pilp"));
    // int jh_session_destroy(jh_session* s)
    private static final MethodHandle jh_session_destroy = h("jh_session_destroy", JAVA_INT, sig("p"));
    // int jh_session_page_info(jh_session* s, int32_t* out4)
    private static final MethodHandle jh_session_page_info = h("jh_session_page_info", JAVA_INT, sig("pp"));
    // int jh_session_set_eos(jh_session* s, const int32_t* eos_ids, int n_eos)
    private static final MethodHandle jh_session_set_eos = h("jh_session_set_eos", JAVA_INT, sig("ppi"));
    // int jh_session_set_strict(jh_session* s, int on)
    private static final MethodHandle jh_session_set_strict = h("jh_session_set_strict", JAVA_INT, sig("pi"));
    // int jh_session_synchronize(jh_session* s)
    private static final MethodHandle jh_session_synchronize = h("jh_session_synchronize", JAVA_INT, sig("p"));
    // void* jh_session_stream(jh_session* s)
    private static final MethodHandle jh_session_stream = h("jh_session_stream", ADDRESS, sig("p"));
    // int jh_forward(jh_session* s, const int32_t* tokens, const float* x_in, int n, int start_pos, float* x_out)
    private static final MethodHandle jh_forward = h("jh_forward", JAVA_INT, sig("pppiip"));
    // int jh_forward_device(jh_session* s, const int32_t* tokens, const float* x_in_dev, int n, int start_pos, float* x_out_dev)
    private static final MethodHandle jh_forward_device = h("jh_forward_device", JAVA_INT, sig("pppiip"));
    // int jh_sample(jh_session* s, float temperature, float u, int32_t* next_token, float* logits_out)
    private static final MethodHandle jh_sample = h("jh_sample", JAVA_INT, sig("pffpp"));
    // int jh_decode_step(jh_session* s, int32_t token, int pos, int32_t* next_token)
    private static final MethodHandle jh_decode_step = h("jh_decode_step", JAVA_INT, sig("piip"));
    // int jh_decode_n(jh_session* s, int32_t first_token, int start_pos, int n, int32_t* out_tokens)
    private static final MethodHandle jh_decode_n = h("jh_decode_n", JAVA_INT, sig("piiip"));
    // int jh_decode_n_sampled(jh_session* s, int32_t first_token, int start_pos, int n, float temperature, const float* u, int32_t* out_tokens)
    private static final MethodHandle jh_decode_n_sampled = h("jh_decode_n_sampled", JAVA_INT, sig("piiifpp"));
    // int jh_decode_n_async(jh_session* s, int32_t first_token, int start_pos, int n)
    private static final MethodHandle jh_decode_n_async = h("jh_decode_n_async", JAVA_INT, sig("piii"));
    // int jh_decode_wait(jh_session* s, int32_t* out_tokens, int n)
    private static final MethodHandle jh_decode_wait = h("jh_decode_wait", JAVA_INT, sig("ppi"));
    // int jh_decode_generated(jh_session* s, int32_t* out_n)
    private static final MethodHandle jh_decode_generated = h("jh_decode_generated", JAVA_INT, sig("pp"));
    // int jh_decode_stats(jh_session* s, double* ms_per_token, int32_t* kernels_per_token)
    private static final MethodHandle jh_decode_stats = h("jh_decode_stats", JAVA_INT, sig("ppp"));
    // int jh_set_tap_layer(jh_session* s, int layer)
    private static final MethodHandle jh_set_tap_layer = h("jh_set_tap_layer", JAVA_INT, sig("pi"));
    // int jh_get_tap(jh_session* s, int which, float* out, int n)
    private static final MethodHandle jh_get_tap = h("jh_get_tap", JAVA_INT, sig("pipi"));
    // int jh_stage_decode_async(jh_session* s, const int32_t* token_dev, const float* x_in_dev, int pos, float* x_out_dev, int32_t* token_out_dev)
    private static final MethodHandle jh_stage_decode_async = h("jh_stage_decode_async", JAVA_INT, sig("pppipp"));
    // int jh_abi_config_layout(int32_t* out, int n)
    private static final MethodHandle jh_abi_config_layout = h("jh_abi_config_layout", JAVA_INT, sig("pi"));
    // int jh_tp_rank_create(jh_session* shard, int rank, int n_ranks, jh_tp_group** out)
    private static final MethodHandle jh_tp_rank_create = h("jh_tp_rank_create", JAVA_INT, sig("piip"));
    // int jh_tp_rank_handles(jh_tp_group* g, void* out192)
    private static final MethodHandle jh_tp_rank_handles = h("jh_tp_rank_handles", JAVA_INT, sig("pp"));
    // int jh_tp_rank_connect(jh_tp_group* g, const void* all_handles)
    private static final MethodHandle jh_tp_rank_connect = h("jh_tp_rank_connect", JAVA_INT, sig("pp"));
    // int jh_tp_rank_decode_n(jh_tp_group* g, int32_t first_token, int start_pos, int n, int32_t* out_tokens)
    private static final MethodHandle jh_tp_rank_decode_n = h("jh_tp_rank_decode_n", JAVA_INT, sig("piiip"));
    // int jh_tp_group_status(jh_tp_group* g, int32_t* out, int n)
    private static final MethodHandle jh_tp_group_status = h("jh_tp_group_status", JAVA_INT, sig("ppi"));
    // int jh_tp_rank_signature(jh_tp_group* g, int64_t* out)
    private static final MethodHandle jh_tp_rank_signature = h("jh_tp_rank_signature", JAVA_INT, sig("pp"));
    // int jh_set_option(const char* name, int32_t value)
    private static final MethodHandle jh_set_option = h("jh_set_option", JAVA_INT, sig("pi"));
    // int jh_clear_options(void)
    private static final MethodHandle jh_clear_options = h("jh_clear_options", JAVA_INT, sig(""));
    // int64_t jh_model_tiled_bytes(jh_model* m)
    private static final MethodHandle jh_model_tiled_bytes = h("jh_model_tiled_bytes", JAVA_LONG, sig("p"));
    // int64_t jh_model_released_bytes(jh_model* m)
    private static final MethodHandle jh_model_released_bytes = h("jh_model_released_bytes", JAVA_LONG, sig("p"));
    // int jh_pipeline_peer_access(jh_pipeline* p, int32_t* out, int n)
    private static final MethodHandle jh_pipeline_peer_access = h("jh_pipeline_peer_access", JAVA_INT, sig("ppi"));
    // int jh_pipeline_create(jh_session* const* stages, int n_stages, jh_pipeline** out)
    private static final MethodHandle jh_pipeline_create = h("jh_pipeline_create", JAVA_INT, sig("pip"));
    // int jh_pipeline_destroy(jh_pipeline* p)
    private static final MethodHandle jh_pipeline_destroy = h("jh_pipeline_destroy", JAVA_INT, sig("p"));
    // int jh_pipeline_prefill(jh_pipeline* p, const int32_t* tokens, int n, int start_pos, int32_t* first_token)
    private static final MethodHandle jh_pipeline_prefill = h("jh_pipeline_prefill", JAVA_INT, sig("ppiip"));
    // int jh_pipeline_decode_n_async(jh_pipeline* p, int32_t first_token, int start_pos, int n)
    private static final MethodHandle jh_pipeline_decode_n_async = h("jh_pipeline_decode_n_async", JAVA_INT, sig("piii"));
    // int jh_pipeline_decode_wait(jh_pipeline* p, int32_t* out_tokens, int n)
    private static final MethodHandle jh_pipeline_decode_wait = h("jh_pipeline_decode_wait", JAVA_INT, sig("ppi"));
    // int jh_tp_set_row(jh_session* s, int32_t token, const float* x_dev, int pos)
    private static final MethodHandle jh_tp_set_row = h("jh_tp_set_row", JAVA_INT, sig("pipi"));
    // int jh_tp_attn(jh_session* s, int layer, float* partial_out_dev)
    private static final MethodHandle jh_tp_attn = h("jh_tp_attn", JAVA_INT, sig("pip"));
    // int jh_tp_ffn(jh_session* s, int layer, const float* reduced_attn_dev, float* partial_out_dev)
    private static final MethodHandle jh_tp_ffn = h("jh_tp_ffn", JAVA_INT, sig("pipp"));
    // int jh_tp_finish_layer(jh_session* s, const float* reduced_ffn_dev)
    private static final MethodHandle jh_tp_finish_layer = h("jh_tp_finish_layer", JAVA_INT, sig("pp"));
    // int jh_tp_rows_max(jh_session* s)
    private static final MethodHandle jh_tp_rows_max = h("jh_tp_rows_max", JAVA_INT, sig("p"));
    // int jh_tp_set_rows(jh_session* s, const int32_t* tokens, const float* x_dev, int n, int start_pos)
    private static final MethodHandle jh_tp_set_rows = h("jh_tp_set_rows", JAVA_INT, sig("pppii"));
    // int jh_tp_attn_rows(jh_session* s, int layer, float* partial_out_dev)
    private static final MethodHandle jh_tp_attn_rows = h("jh_tp_attn_rows", JAVA_INT, sig("pip"));
    // int jh_tp_ffn_rows(jh_session* s, int layer, const float* reduced_attn_dev, float* partial_out_dev)
    private static final MethodHandle jh_tp_ffn_rows = h("jh_tp_ffn_rows", JAVA_INT, sig("pipp"));
    // int jh_tp_finish_layer_rows(jh_session* s, const float* reduced_ffn_dev)
    private static final MethodHandle jh_tp_finish_layer_rows = h("jh_tp_finish_layer_rows", JAVA_INT, sig("pp"));
    // int jh_tp_finish_rows(jh_session* s, float* rows_out_dev)
    private static final MethodHandle jh_tp_finish_rows = h("jh_tp_finish_rows", JAVA_INT, sig("pp"));
    // int jh_tp_group_create(jh_session* const* shards, int n_shards, jh_tp_group** out)
    private static final MethodHandle jh_tp_group_create = h("jh_tp_group_create", JAVA_INT, sig("pip"));
    // int jh_tp_group_destroy(jh_tp_group* g)
    private static final MethodHandle jh_tp_group_destroy = h("jh_tp_group_destroy", JAVA_INT, sig("p"));
    // int jh_tp_group_forward(jh_tp_group* g, const int32_t* tokens, int n, int start_pos)
    private static final MethodHandle jh_tp_group_forward = h("jh_tp_group_forward", JAVA_INT, sig("ppii"));
    // int jh_tp_group_sample(jh_tp_group* g, int32_t* next_token)
    private static final MethodHandle jh_tp_group_sample = h("jh_tp_group_sample", JAVA_INT, sig("pp"));
    // int jh_tp_group_decode_n(jh_tp_group* g, int32_t first_token, int start_pos, int n, int32_t* out_tokens)
    private static final MethodHandle jh_tp_group_decode_n = h("jh_tp_group_decode_n", JAVA_INT, sig("piiip"));

    public static int jh_model_create(MemorySegment cfg, MemorySegment out) {
        try { return (int) jh_model_create.invokeExact(cfg, out); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_model_destroy(MemorySegment m) {
        try { return (int) jh_model_destroy.invokeExact(m); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_model_set_weight(MemorySegment m, int layer, int which, int dtype, MemorySegment data, MemorySegment scales, int rows, int cols, int from_device) {
        try { return (int) jh_model_set_weight.invokeExact(m, layer, which, dtype, data, scales, rows, cols, from_device); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_model_set_kv_head_offset(MemorySegment m, int kv_head_offset) {
        try { return (int) jh_model_set_kv_head_offset.invokeExact(m, kv_head_offset); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_session_create(MemorySegment m, int max_ctx, long max_page_bytes, MemorySegment out) {
        try { return (int) jh_session_create.invokeExact(m, max_ctx, max_page_bytes, out); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_session_destroy(MemorySegment s) {
        try { return (int) jh_session_destroy.invokeExact(s); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_session_page_info(MemorySegment s, MemorySegment out4) {
        try { return (int) jh_session_page_info.invokeExact(s, out4); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_session_set_eos(MemorySegment s, MemorySegment eos_ids, int n_eos) {
        try { return (int) jh_session_set_eos.invokeExact(s, eos_ids, n_eos); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_session_set_strict(MemorySegment s, int on) {
        try { return (int) jh_session_set_strict.invokeExact(s, on); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_session_synchronize(MemorySegment s) {
        try { return (int) jh_session_synchronize.invokeExact(s); } catch (Throwable t) { throw rethrow(t); }
    }

    public static MemorySegment jh_session_stream(MemorySegment s) {
        try { return (MemorySegment) jh_session_stream.invokeExact(s); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_forward(MemorySegment s, MemorySegment tokens, MemorySegment x_in, int n, int start_pos, MemorySegment x_out) {
        try { return (int) jh_forward.invokeExact(s, tokens, x_in, n, start_pos, x_out); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_forward_device(MemorySegment s, MemorySegment tokens, MemorySegment x_in_dev, int n, int start_pos, MemorySegment x_out_dev) {
        try { return (int) jh_forward_device.invokeExact(s, tokens, x_in_dev, n, start_pos, x_out_dev); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_sample(MemorySegment s, float temperature, float u, MemorySegment next_token, MemorySegment logits_out) {
        try { return (int) jh_sample.invokeExact(s, temperature, u, next_token, logits_out); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_decode_step(MemorySegment s, int token, int pos, MemorySegment next_token) {
        try { return (int) jh_decode_step.invokeExact(s, token, pos, next_token); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_decode_n(MemorySegment s, int first_token, int start_pos, int n, MemorySegment out_tokens) {
        try { return (int) jh_decode_n.invokeExact(s, first_token, start_pos, n, out_tokens); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_decode_n_sampled(MemorySegment s, int first_token, int start_pos, int n, float temperature, MemorySegment u, MemorySegment out_tokens) {
        try { return (int) jh_decode_n_sampled.invokeExact(s, first_token, start_pos, n, temperature, u, out_tokens); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_decode_n_async(MemorySegment s, int first_token, int start_pos, int n) {
        try { return (int) jh_decode_n_async.invokeExact(s, first_token, start_pos, n); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_decode_wait(MemorySegment s, MemorySegment out_tokens, int n) {
        try { return (int) jh_decode_wait.invokeExact(s, out_tokens, n); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_decode_generated(MemorySegment s, MemorySegment out_n) {
        try { return (int) jh_decode_generated.invokeExact(s, out_n); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_decode_stats(MemorySegment s, MemorySegment ms_per_token, MemorySegment kernels_per_token) {
        try { return (int) jh_decode_stats.invokeExact(s, ms_per_token, kernels_per_token); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_set_tap_layer(MemorySegment s, int layer) {
        try { return (int) jh_set_tap_layer.invokeExact(s, layer); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_get_tap(MemorySegment s, int which, MemorySegment out, int n) {
        try { return (int) jh_get_tap.invokeExact(s, which, out, n); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_stage_decode_async(MemorySegment s, MemorySegment token_dev, MemorySegment x_in_dev, int pos, MemorySegment x_out_dev, MemorySegment token_out_dev) {
        try { return (int) jh_stage_decode_async.invokeExact(s, token_dev, x_in_dev, pos, x_out_dev, token_out_dev); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_abi_config_layout(MemorySegment out, int n) {
        try { return (int) jh_abi_config_layout.invokeExact(out, n); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_rank_create(MemorySegment shard, int rank, int n_ranks, MemorySegment out) {
        try { return (int) jh_tp_rank_create.invokeExact(shard, rank, n_ranks, out); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_rank_handles(MemorySegment g, MemorySegment out192) {
        try { return (int) jh_tp_rank_handles.invokeExact(g, out192); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_rank_connect(MemorySegment g, MemorySegment all_handles) {
        try { return (int) jh_tp_rank_connect.invokeExact(g, all_handles); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_rank_decode_n(MemorySegment g, int first_token, int start_pos, int n, MemorySegment out_tokens) {
        try { return (int) jh_tp_rank_decode_n.invokeExact(g, first_token, start_pos, n, out_tokens); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_group_status(MemorySegment g, MemorySegment out, int n) {
        try { return (int) jh_tp_group_status.invokeExact(g, out, n); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_rank_signature(MemorySegment g, MemorySegment out) {
        try { return (int) jh_tp_rank_signature.invokeExact(g, out); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_set_option(MemorySegment name, int value) {
        try { return (int) jh_set_option.invokeExact(name, value); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_clear_options() {
        try { return (int) jh_clear_options.invokeExact(); } catch (Throwable t) { throw rethrow(t); }
    }

    public static long jh_model_tiled_bytes(MemorySegment m) {
        try { return (long) jh_model_tiled_bytes.invokeExact(m); } catch (Throwable t) { throw rethrow(t); }
    }

    public static long jh_model_released_bytes(MemorySegment m) {
        try { return (long) jh_model_released_bytes.invokeExact(m); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_pipeline_peer_access(MemorySegment p, MemorySegment out, int n) {
        try { return (int) jh_pipeline_peer_access.invokeExact(p, out, n); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_pipeline_create(MemorySegment stages, int n_stages, MemorySegment out) {
        try { return (int) jh_pipeline_create.invokeExact(stages, n_stages, out); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_pipeline_destroy(MemorySegment p) {
        try { return (int) jh_pipeline_destroy.invokeExact(p); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_pipeline_prefill(MemorySegment p, MemorySegment tokens, int n, int start_pos, MemorySegment first_token) {
        try { return (int) jh_pipeline_prefill.invokeExact(p, tokens, n, start_pos, first_token); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_pipeline_decode_n_async(MemorySegment p, int first_token, int start_pos, int n) {
        try { return (int) jh_pipeline_decode_n_async.invokeExact(p, first_token, start_pos, n); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_pipeline_decode_wait(MemorySegment p, MemorySegment out_tokens, int n) {
        try { return (int) jh_pipeline_decode_wait.invokeExact(p, out_tokens, n); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_set_row(MemorySegment s, int token, MemorySegment x_dev, int pos) {
        try { return (int) jh_tp_set_row.invokeExact(s, token, x_dev, pos); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_attn(MemorySegment s, int layer, MemorySegment partial_out_dev) {
        try { return (int) jh_tp_attn.invokeExact(s, layer, partial_out_dev); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_ffn(MemorySegment s, int layer, MemorySegment reduced_attn_dev, MemorySegment partial_out_dev) {
        try { return (int) jh_tp_ffn.invokeExact(s, layer, reduced_attn_dev, partial_out_dev); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_finish_layer(MemorySegment s, MemorySegment reduced_ffn_dev) {
        try { return (int) jh_tp_finish_layer.invokeExact(s, reduced_ffn_dev); } catch (Throwable t) { throw rethrow(t); }
    }
    public static int jh_tp_rows_max(MemorySegment s) {
        try { return (int) jh_tp_rows_max.invokeExact(s); } catch (Throwable t) { throw rethrow(t); }
    }
    public static int jh_tp_set_rows(MemorySegment s, MemorySegment tokens, MemorySegment x_dev, int n, int start_pos) {
        try { return (int) jh_tp_set_rows.invokeExact(s, tokens, x_dev, n, start_pos); } catch (Throwable t) { throw rethrow(t); }
    }
    public static int jh_tp_attn_rows(MemorySegment s, int layer, MemorySegment partial_out_dev) {
        try { return (int) jh_tp_attn_rows.invokeExact(s, layer, partial_out_dev); } catch (Throwable t) { throw rethrow(t); }
    }
    public static int jh_tp_ffn_rows(MemorySegment s, int layer, MemorySegment reduced_attn_dev, MemorySegment partial_out_dev) {
        try { return (int) jh_tp_ffn_rows.invokeExact(s, layer, reduced_attn_dev, partial_out_dev); } catch (Throwable t) { throw rethrow(t); }
    }
    public static int jh_tp_finish_layer_rows(MemorySegment s, MemorySegment reduced_ffn_dev) {
        try { return (int) jh_tp_finish_layer_rows.invokeExact(s, reduced_ffn_dev); } catch (Throwable t) { throw rethrow(t); }
    }
    public static int jh_tp_finish_rows(MemorySegment s, MemorySegment rows_out_dev) {
        try { return (int) jh_tp_finish_rows.invokeExact(s, rows_out_dev); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_group_create(MemorySegment shards, int n_shards, MemorySegment out) {
        try { return (int) jh_tp_group_create.invokeExact(shards, n_shards, out); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_group_destroy(MemorySegment g) {
        try { return (int) jh_tp_group_destroy.invokeExact(g); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_group_forward(MemorySegment g, MemorySegment tokens, int n, int start_pos) {
        try { return (int) jh_tp_group_forward.invokeExact(g, tokens, n, start_pos); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_group_sample(MemorySegment g, MemorySegment next_token) {
        try { return (int) jh_tp_group_sample.invokeExact(g, next_token); } catch (Throwable t) { throw rethrow(t); }
    }

    public static int jh_tp_group_decode_n(MemorySegment g, int first_token, int start_pos, int n, MemorySegment out_tokens) {
        try { return (int) jh_tp_group_decode_n.invokeExact(g, first_token, start_pos, n, out_tokens); } catch (Throwable t) { throw rethrow(t); }
    }

}
