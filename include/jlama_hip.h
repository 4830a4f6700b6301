/*
 * jlama_hip.h -- C ABI of libjlamahip.so, the MI355X (gfx950) tensor backend for Jlama.
 *
 * Plain `extern "C"`: only int32/int64/float/pointer parameters, no structs by value, no
 * callbacks -- bindable from Panama FFM (jextract), a JNI shim, cgo or ctypes alike.  The
 * reference's own native providers are bound the same way
 * (jlama-native/src/main/c/simd/jextract_vector_simd.sh:6-28,
 *  jlama-native/src/main/java22/.../cnative/NativeSimd.java:117-189).
 *
 * Two tiers:
 *   Tier 1 -- drop-in for the Java interface `TensorOperations`
 *             (jlama-core/.../tensor/operations/TensorOperations.java:25-161).  Same offset /
 *             stride conventions as the reference's C SIMD library
 *             (jlama-native/src/main/c/simd/vector_simd.h:22-38) so a Java
 *             `HipTensorOperations` can copy the marshaling of
 *             jlama-native/.../NativeSimdTensorOperations.java:84-232.  Host pointers in, host
 *             pointers out; weights may be pre-registered in HBM
 *             (cf. NativeGPUTensorOperations.registerModelTensor :104-151, vector_gpu.h:11).
 *   Tier 2 -- device-resident model/session: activations, paged KV and the decode loop stay
 *             in HBM (what the metric is measured on).  Call order = TransformerBlock.forward
 *             (jlama-core/.../model/TransformerBlock.java:158-215).
 *
 * Errors: every function returns 0 on success or a negative JH_ERR_* code; jh_last_error()
 * gives the message for the calling thread.  Nothing aborts.  JH_ERR_UNSUPPORTED lets the
 * Java wrapper throw UnsupportedOperationException exactly where Panama does
 * (PanamaTensorOperations.java:125-142).
 */
#ifndef JLAMA_HIP_H
#define JLAMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JH_OK 0
#define JH_ERR_NO_DEVICE (-1)   /* constructor of the provider must throw => fallback (TensorOperationsProvider.java:50-87) */
#define JH_ERR_OOM (-2)         /* registration failed => caller keeps the tensor on CPU (NativeGPUTensorOperations.java:111-150) */
#define JH_ERR_UNSUPPORTED (-3) /* dtype pair / shape not supported */
#define JH_ERR_INVALID (-4)     /* bad argument */
#define JH_ERR_HIP (-5)         /* HIP runtime error; see jh_last_error() */

/* DType tags (jlama-core/.../safetensors/DType.java:18-61 subset on the hot path) */
#define JH_DT_F32 0
#define JH_DT_BF16 1
#define JH_DT_I8 2
#define JH_DT_Q4 3

/* ------------------------------------------------------------------ runtime / provider facts */

/* Select `device` (HIP ordinal) for the calling thread's later calls and initialise the runtime.
 * out_info (may be NULL): [0]=free HBM bytes, [1]=CU count, [2]=device count, [3]=LDS bytes/CU.
 * Replaces init_gpu (jlama-native/src/main/c/gpu/vector_gpu.h:9). */
int jh_init(int device, int64_t* out_info);
/* TensorOperations.name() :28 */
const char* jh_name(void);
/* TensorOperations.parallelSplitSize() :30 -- 1: every GEMM arrives whole
 * (NativeGPUTensorOperations.java:98-101). */
int jh_parallel_split_size(void);
/* TensorOperations.preferredWorkingQuantizedType() :32-34 -> JH_DT_I8 */
int jh_preferred_working_qtype(void);
const char* jh_last_error(void);
/* Hash of the sources this binary was compiled from (lets the host side detect a stale library). */
const char* jh_source_hash(void);
/* Layout of jh_config as THIS binary was compiled: out[0] = sizeof(jh_config), out[1..] = offsetof of its fields in declaration
 * order (at most n ints are written; returns the number of ints available = 1 + field count).  A binding checks its own struct
 * layout against it at load time (ctypes: jlama_amd/_native.py; FFM: HipResidentLlama's static initialiser). */
int jh_abi_config_layout(int32_t* out, int n);
/* Block until all work queued by this thread's stream has finished. */
int jh_synchronize(void);
/* Process options.  The library reads NO tuning knob from the process environment (a JVM host would inherit whatever its launcher
 * exported); an option exists only after the host sets it here.  The one exception is a documented handful that jh_init copies
 * from the environment once: JH_TRACE, JH_NO_GRAPH, JH_STRICT_ORDER, JH_TILED_COPY (auto|resident|transient), JH_TP_LOUD.
 * Everything else is a constant chosen by the launch planners; names and meanings of the test / tuning options are listed in
 * tools/README.md.  No reference counterpart (the reference has no such switches). */
int jh_set_option(const char* name, int32_t value);
int jh_clear_options(void);

/* ------------------------------------------------------------------ Tier 1: TensorOperations */

/* registerModelTensor (TensorOperations.java:39; vector_gpu.h:11 register_tensor): copy a weight
 * buffer (nibbles, scales, bf16 or f32 data -- raw bytes) to HBM once.  Returns id >= 0 or
 * JH_ERR_OOM / JH_ERR_HIP.  The host buffer is not retained. */
int64_t jh_register_tensor(const void* host, int64_t bytes);
int jh_unregister_tensor(int64_t id);

/* batchDotProduct, I8 activations x Q4 weights -> F32.  Replaces gemm_q8_q4
 * (vector_simd.h:22).  Identical argument meaning:
 *   r[ldc*i + j - roffset] = sum_blk af[ldaf*i + aoffset/32 + blk] * bf[ldbf*j + (boffset*2)/32 + blk]
 *                            * sum_t a[lda*i + aoffset + 32*blk + t] * (nib(b[ldb*j + boffset + 16*blk], t) - 8)
 *   for i in [0,m), j in [n0, n0+n).  aoffset in elements, boffset in BYTES of the nibble
 *   buffer, ldb in bytes, ldaf/ldbf in floats (vector_simd.c:300-305,344).
 * b_id/bf_id: ids from jh_register_tensor, or -1 to use the host pointers b/bf. */
int jh_gemm_q8_q4(int64_t b_id, int64_t bf_id, const float* af, const int8_t* a, int aoffset, const float* bf,
                  const uint8_t* b, int boffset, float* r, int roffset, int m, int n0, int n, int k, int lda,
                  int ldaf, int ldb, int ldbf, int ldc);
/* F32 x Q4 -> F32.  Replaces gemm_f32_q4 (vector_simd.h:30). */
int jh_gemm_f32_q4(int64_t b_id, int64_t bf_id, const float* a, int aoffset, const float* bf, const uint8_t* b,
                   int boffset, float* r, int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldbf,
                   int ldc);
/* F32 x F32 -> F32.  Replaces gemm_f32 (vector_simd.h:26). */
int jh_gemm_f32(int64_t b_id, const float* a, int aoffset, const float* b, int boffset, float* r, int roffset,
                int m, int n0, int n, int k, int lda, int ldb, int ldc);
/* BF16 x BF16 -> F32 (r) or BF16 (cr) -- replaces gemm_bf16 (vector_simd.h:34).  Exactly one of cr / r is used, as in
 * the reference (NativeSimdTensorOperations.java:113-131 passes cr = result when result.dType()==BF16, else NULL;
 * vector_simd.c:1060-1064): cr != NULL => the results are rounded to BF16 and stored at cr[ldc*i + j - roffset], r is
 * ignored.  Rounding is FloatConversions.float32ToBFloat16 (RNE, FloatConversions.java:35-60 -- what the Panama provider
 * writes, PTO:1305-1311); the reference C helper fp32_to_bf16 (vector_simd.c:22-38) truncates its result to `char`
 * and is not reproduced. */
int jh_gemm_bf16(int64_t b_id, const uint16_t* a, int aoffset, const uint16_t* b, int boffset, uint16_t* cr, float* r,
                 int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc);
/* F32 x BF16 -> F32 (r) or BF16 (cr) -- replaces gemm_f32_bf16 (vector_simd.h:38). */
int jh_gemm_f32_bf16(int64_t b_id, const float* a, int aoffset, const uint16_t* b, int boffset, uint16_t* cr, float* r,
                     int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc);
/* `_batch` forms (dotProductBatchChunk, TensorOperations.java:86-99; vector_simd.h:23,31): the same A against
 * batch_num weight tensors, results into batch_num result buffers. */
int jh_gemm_q8_q4_batch(int batch_num, const int64_t* b_ids, const int64_t* bf_ids, const float* af,
                        const int8_t* a, int aoffset, const float* const* bf, const uint8_t* const* b, int boffset,
                        float* const* r, int roffset, int m, int n0, int n, int k, int lda, int ldaf, int ldb,
                        int ldbf, int ldc);
int jh_gemm_f32_q4_batch(int batch_num, const int64_t* b_ids, const int64_t* bf_ids, const float* a, int aoffset,
                         const float* const* bf, const uint8_t* const* b, int boffset, float* const* r, int roffset,
                         int m, int n0, int n, int k, int lda, int ldb, int ldbf, int ldc);
/* the remaining `_batch` entry points of the reference library (vector_simd.h:27,35,39); b_ids / cr may be NULL */
int jh_gemm_f32_batch(int batch_num, const int64_t* b_ids, const float* a, int aoffset, const float* const* b, int boffset,
                      float* const* r, int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc);
int jh_gemm_bf16_batch(int batch_num, const int64_t* b_ids, const uint16_t* a, int aoffset, const uint16_t* const* b,
                       int boffset, uint16_t* const* cr, float* const* r, int roffset, int m, int n0, int n, int k, int lda,
                       int ldb, int ldc);
int jh_gemm_f32_bf16_batch(int batch_num, const int64_t* b_ids, const float* a, int aoffset, const uint16_t* const* b,
                           int boffset, uint16_t* const* cr, float* const* r, int roffset, int m, int n0, int n, int k,
                           int lda, int ldb, int ldc);

/* accumulate (TensorOperations.java:104): a[i] += b[i], i in [offset, offset+length).  F32 += F32
 * (PanamaTensorOperations.java:2281-2295). */
int jh_accumulate_f32(float* a, const float* b, int offset, int length);
/* F32 += Q4 row (PanamaTensorOperations.java:2297-2325): a[i] += (nib-8)*scale.  nib/scales point at the row. */
int jh_accumulate_f32_q4(float* a, const uint8_t* nib_row, const float* scale_row, int offset, int length);
/* maccumulate (:109): a[i] *= b[i] */
int jh_maccumulate_f32(float* a, const float* b, int offset, int length);
/* scale (:140): a[i] *= factor */
int jh_scale_f32(float factor, float* a, int offset, int length);
/* saxpy (:114): y[yoffset+t] = fma(x[xoffset+t], alpha, y[yoffset+t]) */
int jh_saxpy_f32(float alpha, const float* x, float* y, int xoffset, int yoffset, int limit);
/* batched saxpy (:119-135, Panama :2648-2698): y += sum_n alpha[aoffset+n] * x[(xrowoffset+n)*ldx + xoffset ..],
 * one fma chain per element over rows in ascending order. */
int jh_saxpy_batch_f32(const float* alpha, const float* x, int ldx, float* y, int xoffset, int yoffset, int limit,
                       int aoffset, int xrowoffset, int batch_size);
/* quantize (:145-149) F32 -> I8 with the Panama-512 semantics (PanamaTensorOperations.java:1684-1723):
 * d = max|x|/127, q = (int8) trunc(x*(127/max|x|) + 0.5f).  x: [rows, ldx]; columns [offset, offset+length). */
int jh_quantize_q8(const float* x, int rows, int ldx, int offset, int length, int8_t* q, int ldq, float* d,
                   int ldd);
/* quantize F32 -> BF16, round-to-nearest-even (FloatConversions.java:35-60). */
int jh_quantize_bf16(const float* x, int64_t n, uint16_t* out);

/* Ops the reference keeps in scalar Java OUTSIDE TensorOperations; exported so a Java caller can stop
 * bouncing activations (last bullet) and so each device kernel has a host-buffer parity hook. */
/* RMSNorm.forward (jlama-core/.../model/RMSNorm.java:33-56).  w: F32 norm weights. */
int jh_rmsnorm_f32(const float* x, const float* w, float weight_adj, int n, float eps, float* out);
/* VectorMath.softMax (jlama-core/.../math/VectorMath.java:69-90) in place on x[offset, offset+length). */
int jh_softmax_f32(float* x, int offset, int length);
/* SiLU(x)*up (ActivationFunction.java:31 + MLPBlock.java:132-142): g[i] = silu(g[i]) * u[i]. */
int jh_silu_mul_f32(float* g, const float* u, int n);
/* GPT-2 family (BASELINE.json configs[0]): LayerNorm.forward (jlama-core/.../model/LayerNorm.java:41-67) over columns
 * [offset, offset+length) of `rows` rows (leading dimension ld, divisor = embeddingLength; float sums in index order),
 * and the tanh-GELU of ActivationFunction.java:32-34 (double), in place. */
int jh_layernorm_f32(const float* x, const float* w, const float* b, int rows, int ld, int offset, int length, int divisor,
                     float eps, float* out);
int jh_gelu_f32(float* x, int n);
/* VectorMath.precomputeFreqsCis (VectorMath.java:148-165): out [end*dim/2][2] = (cos, sin). Host-side. */
int jh_rope_table(int dim, int end, double theta, double scaling, float* out);
/* RoPE rotation of one q row [n_heads*head_size] and one k row [n_kv_heads*head_size] at `position`
 * (CausalSelfAttention.java:247-286, GQA branch incl. the per-kv-head table offset: kv head h reads the table row of
 * position + 2*h).  rope = jh_rope_table output covering `table_positions` positions; a position whose rows would run
 * past the table (position + 2*(n_kv_heads-1) >= table_positions) is JH_ERR_INVALID -- the reference throws
 * ArrayIndexOutOfBounds there. */
int jh_rope_apply_f32(float* q, float* k, const float* rope, int table_positions, int position, int n_heads, int n_kv_heads,
                      int head_size);
/* KvBufferCache.computePageSize (jlama-core/.../tensor/KvBufferCache.java:224-280). out2 = {layersPerPage, ctxPerPage}. */
int jh_kv_page_geometry(int64_t max_page_bytes, int n_layers, int context_length, int kv_length, int dtype_size,
                        int32_t* out2);

/* ------------------------------------------------------------------ Tier 2: resident model */

typedef struct jh_model jh_model;
typedef struct jh_session jh_session;

/* Config fields the path needs (jlama-core/.../safetensors/Config.java:253-274, LlamaConfig.java:29-41,
 * DistributedContext.java:75-77).  Passed by pointer. */
typedef struct jh_config {
    int32_t embedding_length; /* E */
    int32_t hidden_length;    /* H */
    int32_t n_heads, n_kv_heads, head_size;
    int32_t n_layers;         /* total layers of the model */
    int32_t vocab_size;
    int32_t context_length;
    int32_t weight_dtype;     /* JH_DT_Q4 => I8 activations; JH_DT_BF16 => BF16 activations (AbstractModel.java:119-169) */
    int32_t layer_start, layer_end; /* this shard's layers [start,end) */
    float rms_eps;
    float rope_theta;
    float rope_scaling;
} jh_config;

/* weight slots (LlamaModel.java:102-147,152-173) */
#define JH_W_Q 0
#define JH_W_K 1
#define JH_W_V 2
#define JH_W_O 3
#define JH_W_GATE 4
#define JH_W_UP 5
#define JH_W_DOWN 6
#define JH_W_NORM1 7      /* input_layernorm */
#define JH_W_NORM2 8      /* post_attention_layernorm */
#define JH_W_EMBED 9      /* model.embed_tokens (layer = -1) */
#define JH_W_LMHEAD 10    /* lm_head (layer = -1); absent => tied to EMBED (LlamaModel.java:155-158) */
#define JH_W_FINALNORM 11 /* model.norm (layer = -1) */
#define JH_W_COUNT 12

/* tap ids = DebugSupport.debug names (TransformerBlock.java:165-205, CausalSelfAttention.java:194-196,309-310,359) */
#define JH_TAP_INPUT_EMB 0
#define JH_TAP_QUERY 2
#define JH_TAP_KEY 3
#define JH_TAP_VALUE 4
#define JH_TAP_QUERY_ROPE 5
#define JH_TAP_KEY_ROPE 6
#define JH_TAP_AFTER_ATTENTION 7
#define JH_TAP_ATTN_RES 8        /* post_attn + residual (TransformerBlock.java:185) */
#define JH_TAP_FF_H 10           /* silu(gate)*up (MLPBlock.java:132-142) */
#define JH_TAP_POST_FF_RES 11

int jh_model_create(const jh_config* cfg, jh_model** out);
int jh_model_destroy(jh_model* m);
/* Upload one weight.  data/scales are HOST pointers unless from_device != 0 (then device pointers, copied
 * device-to-device).  Q4: data = nibbles [rows, cols/2] bytes, scales = F32 [rows, cols/32] (the JQ4 layout:
 * `<name>` + `<name>.qb`, Weights.java:159-171).  Norm weights: F32 or BF16 [1, cols], scales NULL. */
int jh_model_set_weight(jh_model* m, int layer, int which, int dtype, const void* data, const float* scales,
                        int rows, int cols, int from_device);
/* Bytes of weights resident in HBM (for the roofline's algorithmic-bytes accounting). */
int64_t jh_model_weight_bytes(jh_model* m);
/* Bytes of the SECOND, MFMA-ordered copy of the projection weights the batched prefill keeps resident (0 until the first
 * prefill made it; always 0 with JH_TILED_COPY=transient, where each GEMM's operand is rebuilt in a per-session scratch). */
int64_t jh_model_tiled_bytes(jh_model* m);
/* Bytes of row-major projection nibbles (and MFMA-ordered prompt copies) RELEASED because the process option JH_STRICT_ONLY=1 was
 * set when a reference-order session of this JQ4 model published its T16 / P16T operand copies: those copies are then the only
 * ones, the model serves reference-order sessions only (order-free use, set_strict(0) and set_weight fail with JH_ERR_UNSUPPORTED).
 * 0 = the option never took effect.  (No reference counterpart: the Java host keeps one heap copy of a checkpoint.) */
int64_t jh_model_released_bytes(jh_model* m);

/* One KV buffer (KvBufferCache.getKvBuffer, KvBufferCache.java:58-60): pages of max_page_bytes (0 => 8 MiB)
 * shaped [layersPerPage, 2, ctxPerPage, kvLength] F32, enough pages for positions [0, max_ctx). */
int jh_session_create(jh_model* m, int max_ctx, int64_t max_page_bytes, jh_session** out);
int jh_session_destroy(jh_session* s);
/* out4 = {layersPerPage, ctxPerPage, nLayerPages, nCtxPages} */
int jh_session_page_info(jh_session* s, int32_t* out4);

/* ---- Tensor-parallel (head-split) shard, SURVEY.md 8(e)/f2.  Replaces the reference's model-shard workers
 * (DistributedContext.java:79-98; the partial results are combined by tensorReducer, CausalSelfAttention.java:378,
 * MLPBlock.java:160).  The shard's jh_config carries its LOCAL n_heads / n_kv_heads / hidden_length and the weights are
 * the matching windows: q,k,v,gate,up by rows, o,down by K columns.  Between the halves the CALLER sums the partial
 * [E] F32 vectors over shards (RCCL all-reduce).  Asynchronous on the session's stream; device pointers. */
int jh_model_set_kv_head_offset(jh_model* m, int kv_head_offset);   /* global index of local kv head 0 (RoPE rows, CausalSelfAttention.java:260-283) */
int jh_tp_set_row(jh_session* s, int32_t token, const float* x_dev, int pos);  /* x = embedding row of token (or x_dev), position pos */
int jh_tp_attn(jh_session* s, int layer, float* partial_out_dev);  /* norm, q|k|v, KV write+RoPE, attention, o-proj partial (no residual) */
int jh_tp_ffn(jh_session* s, int layer, const float* reduced_attn_dev, float* partial_out_dev);  /* x1 = x + reduced; norm, gate/up, SiLU*up, down partial */
int jh_tp_finish_layer(jh_session* s, const float* reduced_ffn_dev);  /* x = x1 + reduced */
int jh_session_get_row(jh_session* s, float* out, int to_device);   /* the session's current row x [E] */
/* The same halves over a CHUNK of prompt rows (AbstractModel.batchForward on a shard, AbstractModel.java:295-312: the reducer then
 * sums [rows, E] once per half-layer instead of [E] once per row).  jh_tp_rows_max: rows a chunk may hold (256), 0 if this shard's
 * shapes have no batched path (then feed rows with jh_tp_set_row).  partial / reduced buffers are [n, E] F32, row-major, device. */
int jh_tp_rows_max(jh_session* s);
int jh_tp_set_rows(jh_session* s, const int32_t* tokens, const float* x_dev, int n, int start_pos);  /* rows = embedding rows of tokens (host ids) or x_dev [n, E] */
int jh_tp_attn_rows(jh_session* s, int layer, float* partial_out_dev);
int jh_tp_ffn_rows(jh_session* s, int layer, const float* reduced_attn_dev, float* partial_out_dev);
int jh_tp_finish_layer_rows(jh_session* s, const float* reduced_ffn_dev);
int jh_tp_finish_rows(jh_session* s, float* rows_out_dev);   /* after the last layer: last row -> current row (sample reads it); rows_out_dev [n, E] optional */

/* ---- One-process tensor-parallel group: N head-split shard sessions (one per device; loopback on one device allowed)
 * driven by one host thread with no host synchronisation inside a layer.  The two reductions of a layer
 * (tensorReducer, CausalSelfAttention.java:378 / MLPBlock.java:160; combine of JlamaService.java:300-376) are one-shot:
 * every shard writes its [E] partial straight into a slot of every peer's buffer (peer stores over xGMI), an event orders
 * the write against the readers, and each shard sums the N slots locally in SHARD ORDER 0..N-1 -- the order of the
 * oracle's lock-step restatement, so results do not depend on arrival order. */
typedef struct jh_tp_group jh_tp_group;
int jh_tp_group_create(jh_session* const* shards, int n_shards, jh_tp_group** out);
int jh_tp_group_destroy(jh_tp_group* g);
/* rows of token ids at positions [start_pos, start_pos+n), one position at a time through all layers on every shard */
int jh_tp_group_forward(jh_tp_group* g, const int32_t* tokens, int n, int start_pos);
/* greedy sample on shard 0 (every shard holds the same residual stream) */
int jh_tp_group_sample(jh_tp_group* g, int32_t* next_token);
/* n greedy decode steps, everything queued without a host round trip; out_tokens: HOST [n] */
int jh_tp_group_decode_n(jh_tp_group* g, int32_t first_token, int start_pos, int n, int32_t* out_tokens);
/* The same group with ONE PROCESS PER SHARD (one rank per GPU; the reference's one-Worker-per-range shape, Worker.java:193-248):
 * a rank holds its shard only and addresses the other ranks' slot / flag / mailbox buffers through hipIpc mappings.
 *   create  -> this rank's buffers;   handles -> 192 bytes (three hipIpcMemHandle_t) for the host side to exchange (any
 *   transport: jlama_amd/distributed.py all_gathers them);   connect <- the n_ranks * 192 bytes of all ranks, in rank order;
 *   decode_n: every rank calls it with the same (first_token, start_pos, n) after the prompt rows (jh_tp_attn / jh_tp_ffn /
 *   jh_tp_finish_layer with the host's all-reduce, as before): one captured graph per token and rank, partial rows pushed into
 *   every rank's slot by the o-proj / down GEMVs, shard-ordered sums, sampled id through mailboxes -- no collective library on
 *   the data path, nothing on the host inside a token.  out_tokens (HOST [n]) is filled on rank 0 only.  A rank returns only
 *   after rank 0's publish of the LAST token has reached its mailbox, so back-to-back calls need no barrier between them; a
 *   meeting that times out disconnects the group (destroy and re-create it on every rank).  Destroy with jh_tp_group_destroy. */
int jh_tp_rank_create(jh_session* shard, int rank, int n_ranks, jh_tp_group** out);
/* Status of a tensor-parallel group (either host): out[0] = what the last decode_n ran on (1: one captured graph per shard and
 * token with in-kernel meetings, 2: the event-ordered host loop, 0: nothing yet), out[1] = meetings that timed out so far (a
 * time-out moves a one-process group to the host loop for good and is reported on stderr; JH_TP_LOUD=1 makes it an error; a rank
 * of the process-per-shard host is disconnected by it), out[2] = 1 while the graph path is in use, out[3] = 1 when the o-proj /
 * down GEMVs push their partial rows themselves (0: separate scatter launches), out[4] = flag words polled per producer launch,
 * out[5] = connected (ranks).  Returns the number of words available (6); at most n are written. */
int jh_tp_group_status(jh_tp_group* g, int32_t* out, int n);
/* One word every rank of a process-per-shard group must agree on before the first jh_tp_rank_decode_n (kernel family, CU count,
 * push mode, shard shape: what the flag-polling protocol depends on); the host all-gathers and compares them. */
int jh_tp_rank_signature(jh_tp_group* g, int64_t* out);
int jh_tp_rank_handles(jh_tp_group* g, void* out192);
int jh_tp_rank_connect(jh_tp_group* g, const void* all_handles);
int jh_tp_rank_decode_n(jh_tp_group* g, int32_t first_token, int start_pos, int n, int32_t* out_tokens);

/* ---- One-process layer-sharded pipeline (BASELINE north_star: "one-process layer sharding across the GPUs of
 * a single node"): stage k is a session of a model created with its own [layer_start, layer_end) on the device that was
 * current (jh_init) at creation; the first stage's model holds the embedding table, the last one final norm + LM head.
 * One host thread queues every stage; a hop is a hipMemcpyPeerAsync of the [n, E] F32 activation (16 KiB per decode token)
 * over xGMI ordered by events -- the PassRecord hand-off of jlama-net/.../Worker.java:193-248 without a host round trip --
 * and the sampled token id travels back to stage 0 the same way (Coordinator.java:184).  Stages may share a device (loopback). */
typedef struct jh_pipeline jh_pipeline;
int jh_pipeline_create(jh_session* const* stages, int n_stages, jh_pipeline** out);
int jh_pipeline_destroy(jh_pipeline* p);
/* Per stage k: how the hop INTO it travels -- 1 direct peer access over xGMI, 0 staged copies (hipDeviceEnablePeerAccess
 * refused), -1 same device as its predecessor; slot 0 = the sampled id's way back to the first stage.  Returns the stage count. */
int jh_pipeline_peer_access(jh_pipeline* p, int32_t* out, int n);
/* batchForward of n prompt rows at [start_pos, start_pos+n) through all stages, then sample (temperature 0) on the last. */
int jh_pipeline_prefill(jh_pipeline* p, const int32_t* tokens, int n, int start_pos, int32_t* first_token);
/* n greedy decode steps; returns after queueing (every stage's work is stream-ordered), _wait fetches the ids. */
int jh_pipeline_decode_n_async(jh_pipeline* p, int32_t first_token, int start_pos, int n);
int jh_pipeline_decode_wait(jh_pipeline* p, int32_t* out_tokens, int n);

/* One pipeline stage per PROCESS (rank-per-GPU hosts, the shape of the reference's cluster: one Worker per layer range,
 * jlama-net/.../Worker.java:193-248; the coordinator samples and feeds the id back, Coordinator.java:184).
 * jh_stage_decode_async queues ONE decode row of this shard on the session's stream and returns: a first stage
 * (layer_start == 0) reads the token id from DEVICE memory (token_dev) and embeds it, later stages take the previous stage's
 * [E] F32 row (x_in_dev); the last stage (layer_end == n_layers) runs final norm + LM head + greedy argmax and stores the id
 * to token_out_dev, the others store their output row to x_out_dev.  No host synchronisation: the caller's transport (RCCL
 * send/recv enqueued on jh_session_stream) orders the hops, so a rank queues ticks ahead of its GPU.  */
int jh_stage_decode_async(jh_session* s, const int32_t* token_dev, const float* x_in_dev, int pos, float* x_out_dev,
                          int32_t* token_out_dev);

/* batchForward (AbstractModel.java:295-312): run rows through this shard's layers at positions
 * [start_pos, start_pos+n).  tokens != NULL: rows come from the embedding table (first shard);
 * else x_in (HOST, [n,E] F32) is the previous shard's output.  x_out (HOST [n,E], may be NULL) receives the
 * shard output ("PassRecord.tensor", jlama-net/.../Worker.java:193-196). */
int jh_forward(jh_session* s, const int32_t* tokens, const float* x_in, int n, int start_pos, float* x_out);
/* Same with DEVICE activations (layer-sharded pipelines hand HBM buffers to RCCL send/recv directly). */
int jh_forward_device(jh_session* s, const int32_t* tokens, const float* x_in_dev, int n, int start_pos,
                      float* x_out_dev);
/* AbstractModel.sample (AbstractModel.java:443-491) on the last forwarded row: final norm -> LM head -> argmax
 * (temperature 0) or softmax-sample with the caller's uniform u.  logits_out (HOST [V]) may be NULL. */
int jh_sample(jh_session* s, float temperature, float u, int32_t* next_token, float* logits_out);
/* One decode iteration of AbstractModel.generate (:590-621): forward(token,pos) + sample at T=0. */
int jh_decode_step(jh_session* s, int32_t token, int pos, int32_t* next_token);
/* n greedy decode iterations chained on the device (argmax feeds the next embedding lookup without a host
 * round trip; one hipGraph replay per token).  out_tokens: HOST [n].  Timing region of the metric. */
int jh_decode_n(jh_session* s, int32_t first_token, int start_pos, int n, int32_t* out_tokens);
/* The same device-resident loop with AbstractModel.sample's temperature branch (core/model/AbstractModel.java:471-489):
 * softmax((logit - max) / temperature) with a FLOAT running sum in index order, inverse CDF against u[i], the uniform of the
 * i-th sampled token (the reference draws ThreadLocalRandom.nextFloat() per call, :594 -- the caller supplies them).  One graph
 * replay per token, no host round trip; temperature == 0 is jh_decode_n.  Honours jh_session_set_eos like jh_decode_n. */
int jh_decode_n_sampled(jh_session* s, int32_t first_token, int start_pos, int n, float temperature, const float* u, int32_t* out_tokens);
/* Stop tokens of the device loop (Config.eosTokens; AbstractModel.java:600-603: generation ends with the step that
 * samples one).  n_eos == 0 clears the set.  With a non-empty set jh_decode_n stops feeding the GPU once a stop token
 * was sampled (checked every few steps without draining the queue); the ids up to and including the stop token are
 * returned and jh_decode_generated() reports how many. */
int jh_session_set_eos(jh_session* s, const int32_t* eos_ids, int n_eos);
/* Tokens the last jh_decode_n / jh_decode_wait produced (== n unless a stop token ended the loop early). */
int jh_decode_generated(jh_session* s, int32_t* out_n);
/* Verification mode: every float accumulation of the decode path in the reference's Panama-512 order (16-lane
 * accumulators over ascending K, halving-tree lane reduction, sequential softmax sum, position-ordered saxpy), so that
 * results are bit-identical to the reference arithmetic instead of equal up to summation order.  Slower (no MFMA
 * prefill, byte-granular weight reads).  Default from the environment: JH_STRICT_ORDER=1.  JQ4 models only. */
int jh_session_set_strict(jh_session* s, int on);
/* Same, but returns after queueing; jh_decode_wait() fetches the tokens.  Lets a caller bracket the loop with
 * its own device events. */
int jh_decode_n_async(jh_session* s, int32_t first_token, int start_pos, int n);
int jh_decode_wait(jh_session* s, int32_t* out_tokens, int n);
/* Logits of the last jh_sample/jh_decode_step, HOST [V]. */
int jh_get_logits(jh_session* s, float* out_v);
/* Stage taps: record the named intermediate of `layer` during the next jh_forward of ONE row. */
int jh_set_tap_layer(jh_session* s, int layer);
int jh_get_tap(jh_session* s, int which, float* out, int n);
/* The HIP stream the session launches on (hipStream_t as void*), so callers can record their own events. */
void* jh_session_stream(jh_session* s);
/* Throughput probe of the batched (prefill) MFMA GEMMs with device-resident operands: kind 0 = I8xQ4, 1 = BF16xBF16;
 * `copies` weight matrices are cycled so they stream from HBM.  out_ms = average milliseconds per GEMM. */
int jh_gemm_bench(int kind, int m, int n, int k, int copies, int iters, double* out_ms);
/* Debug aid: wall-clock (100 MHz) phase stamps of one decode-attention launch at `pos`; out[split*16 + phase]. */
int jh_debug_attn_timeline(jh_session* s, int pos, long long* out, int n);
/* Phase stamps of one reference-order few-row GEMV launch (which: 0 q|k|v, 2 o, 3 gate|up, 4 down): out[workgroup][wave][8], 100 MHz ticks, -1 = unused. */
int jh_debug_gemv_timeline(jh_session* s, int which, long long* out, int n);
/* Wait for everything queued on the session's stream. */
int jh_session_synchronize(jh_session* s);
/* Roofline probe: average duration (ms) of ONE launch of decode kernel `which` (0 qkv, 1 attention, 2 o-proj,
 * 3 gate/up, 4 down), launched back-to-back over all of the shard's layers for `iters` sweeps so the weights stream
 * from HBM, timed with hipEvents on the session's stream; bytes_per_launch = that launch's algorithmic bytes. */
int jh_kernel_bench(jh_session* s, int which, int iters, double* out_ms, int64_t* out_bytes_per_launch);
/* Average duration (ms) of the decode graph replays timed with hipEvents inside the last jh_decode_n, and the
 * number of kernels per replay. */
int jh_decode_stats(jh_session* s, double* ms_per_token, int32_t* kernels_per_token);

#ifdef __cplusplus
}
#endif
#endif /* JLAMA_HIP_H */
