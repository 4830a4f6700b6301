// jh_host.h -- what the translation units of libjlamahip.so share: error / context plumbing, process options, the resident model
// and session structs, and the prototypes of the launch helpers that cross file boundaries.  Internal: the library's contract is
// include/jlama_hip.h (plain C).  Built with -fvisibility=hidden: only the C ABI below is exported.
#pragma once
#pragma GCC visibility push(default)
#include "../../include/jlama_hip.h"
#pragma GCC visibility pop

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "jh_kernels.h"

using namespace jh;

// ------------------------------------------------------------------------------------------------ errors / context  (core.hip)
extern thread_local std::string g_err;
int set_err(int code, const std::string& msg);
#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return set_err(JH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + \
                                           std::to_string(__LINE__));                                    \
    } while (0)
#define JHCHK(expr)            \
    do {                       \
        int _r = (expr);       \
        if (_r != JH_OK) return _r; \
    } while (0)

constexpr int NSCRATCH = 8;
struct ThreadCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    void* scratch[NSCRATCH] = {nullptr};
    size_t cap[NSCRATCH] = {0};
};
extern thread_local ThreadCtx tctx;
extern int g_default_device;
extern int g_cu_count;
int ensure_ctx();
int dev_buf(int slot, size_t bytes, void** out);
struct RegTensor {
    void* ptr;
    int64_t bytes;
    int device;
};
extern std::mutex g_reg_mu;
extern std::unordered_map<int64_t, RegTensor> g_reg;
extern int64_t g_next_id;
const void* reg_ptr(int64_t id);
// process options (jh_set_option; five names copied from the environment once by jh_init)
int opt_int(const char* name, int dflt);
void options_from_environment_once();
template <typename K>
int allow_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return JH_OK;
}
extern std::mutex g_capture_mu;   // one hipGraph capture at a time per process
int trace_sync(const char* what, hipStream_t st);
int nb_for(int K);
struct LaunchCfg { int R, waves, grid_cap, pipe; };   // 0 / -1 = let the planner decide
extern thread_local int g_last_gemv_grid;              // workgroups of the GEMV launched last on this thread (EPI_TP: its flag count)
struct P16Plan { int grid, per, tw; };
constexpr size_t BF16_SPLITK_WS_BYTES = (size_t)8 * 256 * 16384 * 4;   // S x 256 rows x N floats with S*N <= 8*16384 (enforced by the launcher)
constexpr int PB_MAX_ROWS = 256;   // rows per prompt chunk = the MFMA GEMM's M limit (8 tiles of 32)
constexpr int PF_MAX_SPLIT = 8;    // key-range splits of the MFMA prefill attention

// ------------------------------------------------------------------------------------------------ Tier 2: resident model / session
struct JWeight {
    int dtype = -1;
    void* data = nullptr;
    float* scales = nullptr;
    int rows = 0, cols = 0;
    uint8_t* tiled = nullptr;        // Q4 only: resident copy in MFMA order for the prefill GEMM (made at first use)
    float* tiled_scales = nullptr;
    uint8_t* t16 = nullptr;          // Q4 only: resident copy in T16 order (jh_t16.h) for the reference-order MFMA GEMV
    float* t16_scales = nullptr;
    uint8_t* p16t = nullptr;         // resident copy for the reference-order GEMVs: Q4 in P16T order (jh_p16.h), BF16 in BF16T order (jh_bf16r.h)
    bool dropped = false;            // Q4 only: the row-major nibbles were released (JH_STRICT_ONLY); the T16 / P16T copies ARE the weight (scales stay)
};
static inline bool w_present(const JWeight& W) { return W.data || W.dropped; }
struct jh_model {
    jh_config c;
    int device;
    std::vector<JWeight> layer_w;  // [n_layers][JH_W_COUNT]; Q/K/V entries alias slices of qkv[layer]
    std::vector<JWeight> qkv;      // [n_layers] q|k|v stacked along N in ONE allocation => one GEMV, no tensor switch
    std::vector<JWeight> gateup;   // [n_layers] prefill only: gate|up stacked along N in MFMA order (`tiled`), one GEMM for both
    JWeight global_w[JH_W_COUNT];
    float* rope = nullptr;
    float attention_scale;
    int64_t weight_bytes = 0;
    int kv_head_offset = 0;   // tensor-parallel shard (jh_model_set_kv_head_offset)
    int weights_version = 0;  // bumped by jh_model_set_weight: sessions drop graphs that captured the old device pointers
    bool strict_only = false; // JH_STRICT_ONLY took effect: row-major projection nibbles released, reference-order sessions only, weights immutable
    int64_t dropped_bytes = 0;
    int tiled_mode = 0;       // TILED_*: where the prefill GEMM's MFMA-ordered weight operand lives (decided at the first prefill)
    std::mutex op_mu;         // the operand copies (T16 / P16T / BF16T) are per model and shared by its sessions: made under this lock, published packed
};
// The prefill GEMM reads its weight operand in MFMA order.  RESIDENT keeps a second, re-tiled copy of every projection weight in
// HBM (2x the checkpoint; the default while the device has room: this part has 288 GB).  TRANSIENT keeps only the row-major
// weights and rebuilds the operand of each GEMM in a per-session scratch (the largest single weight) right in front of it:
// 1.0x the checkpoint, at the price of one extra read + write of the weights per prompt chunk (retile16_kernel).
// JH_TILED_COPY = auto | resident | transient;  auto = resident when the copy leaves >= 1/4 of the device memory free.
enum { TILED_UNSET = 0, TILED_RESIDENT = 1, TILED_TRANSIENT = 2 };
// Where the o-proj / down GEMV of a tensor-parallel shard delivers its partial row when it runs inside the group's token graph
// (EPI_TP): this shard's slot on every shard + one flag word per workgroup; `grid` returns the flag count of the launch (0 = the
// GEMV could not push -- BF16 model, first-generation strict kernels -- and the caller adds the scatter launch).
constexpr int TP_MAX_FLAGS = 4096;
struct TPPush { float* const* dst; unsigned* const* flags; const unsigned* seq; int n, li, L; int grid; };
enum { TAP_SLOTS = 12 };
constexpr int JH_MAX_EOS = 16;   // stop tokens per session (Config.eosTokens holds 1-3 in practice)
constexpr int N_ATTN_VARIANTS = 3;
struct jh_session {
    jh_model* m;
    hipStream_t stream = nullptr;
    int layers_per_page = 0, ctx_per_page = 0, n_layer_pages = 0, n_ctx_pages = 0, n_ctx_alloc = 0;
    std::vector<float*> pages_host;
    float** pages_dev = nullptr;
    float* kv_slab = nullptr;
    size_t page_elems = 0;
    int max_ctx = 0, max_splits = 32, chunk_cap = 32;
    int long_splits = 32, long_min = 2048, mid_splits = 24, mid_max = 6144;   // attention variant 2: more slices for long contexts
    // activations
    float *x = nullptr, *x1 = nullptr, *qkv = nullptr, *attf = nullptr, *hf = nullptr;
    float *logits = nullptr, *amax_v = nullptr, *part_o = nullptr, *part_ml = nullptr, *tapq = nullptr;
    int part_stride = 16, direct_max = 512, direct_chunk = 128;
    int* amax_i = nullptr;
    unsigned* counters = nullptr;
    DecodeState* st = nullptr;
    int* out_tokens = nullptr;
    int out_cap = 0;
    int lm_grid = 0;
    // graphs exist per attention variant (0: PRE=8 row steps prefetched, 1: PRE=2 for short contexts, 2: long contexts -- more slices)
    int attn_variant = 0;
    int graphs_version = 0;   // jh_model::weights_version the cached graphs were captured against
    hipGraph_t graph[N_ATTN_VARIANTS] = {nullptr, nullptr, nullptr};
    hipGraphExec_t exec[N_ATTN_VARIANTS] = {nullptr, nullptr, nullptr};
    // the greedy loop replays JH_TOKENS_PER_GRAPH tokens per launch where it can (~8 us pass between two graph launches, only a kernel
    // boundary between two tokens inside one graph): same nodes, same order, the position is a device word
    hipGraph_t graph_m[N_ATTN_VARIANTS] = {nullptr, nullptr, nullptr};
    hipGraphExec_t exec_m[N_ATTN_VARIANTS] = {nullptr, nullptr, nullptr};
    int tokens_per_graph = 4;
    hipGraph_t row_graph[N_ATTN_VARIANTS] = {nullptr, nullptr, nullptr};   // this shard's layers only: single-row forward (pipeline stages)
    hipGraphExec_t row_exec[N_ATTN_VARIANTS] = {nullptr, nullptr, nullptr};
    int pending_n = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double ms_per_token = 0;
    int kernels_per_token = 0;
    int tap_layer = -1;
    float* taps[TAP_SLOTS] = {nullptr};
    int tap_len[TAP_SLOTS] = {0};
    LaunchCfg cfg_qkv, cfg_o, cfg_gateup, cfg_down, cfg_lm;
    // batched prefill (lazily allocated): chunk rows x {x, x1, qkv, att, gate, up} F32 + Q8 codes / block scales
    int pb_rows = 0;
    float *pb_x = nullptr, *pb_x1 = nullptr, *pb_qkv = nullptr, *pb_att = nullptr, *pb_g = nullptr, *pb_u = nullptr, *pb_ad = nullptr;
    int8_t* pb_aq = nullptr;
    float* pb_ws = nullptr;   // split-K workspace of the BF16 prefill GEMM
    float *pb_att_o = nullptr, *pb_att_ml = nullptr;   // key-range split partials of the MFMA prefill attention
    struct TPPush* tp_push = nullptr;                  // set while a tensor-parallel token graph is captured: o-proj / down push their partials
    float* p16_scores_b = nullptr;                     // reference-order prefill: score rows of a whole chunk [rows][n_heads][p16_sc_stride]
    uint8_t* pb_sel = nullptr;                         // reference-order prefill on the MFMA (gemm_t16_kernel): one-hot selector operands of a chunk's rows
    float* pb_sad = nullptr;                           // ... and their block scales [nblk][PB_MAX_ROWS]
    float* pb_bfr = nullptr;                           // reference-order prefill of a BF16 model: the activation image of a chunk (jh_bf16r.h)
    uint8_t* tile_w = nullptr;                         // TILED_TRANSIENT: scratch for ONE weight in MFMA order (+ its scales)
    float* tile_s = nullptr;
    size_t tile_w_bytes = 0, tile_s_bytes = 0;
    int prefill_attn_mfma_min = 384;   // chunks whose newest position reaches this many keys take attn_prefill_mfma_kernel
    int* pb_tok = nullptr;
    int* pb_start = nullptr;  // device word: start position of the chunk being prefilled
    std::map<uint64_t, hipGraphExec_t> pb_graphs;   // captured layer loops, key = rows | key-count bucket << 32
    std::vector<hipGraph_t> pb_graph_src;
    int prefill_batch_min = 4;
    int tp_rows = 0, tp_pos0 = 0, tp_last_token = 0;   // chunk of prompt rows a tensor-parallel host is walking through the layers (jh_tp_set_rows)
    int strict = 0;           // jh_session_set_strict: reference-order kernels (jh_p16.h)
    // temperature sampling inside the device loop: exp((l - max)/T) of every logit, the caller's uniforms, the picked id; the
    // decode graphs of this mode are captured per temperature (a kernel argument)
    float* prob = nullptr;
    float* u_dev = nullptr;
    int u_cap = 0;
    int* pick = nullptr;
    float sampled_temp = 0.0f;
    hipGraph_t graph_s[N_ATTN_VARIANTS] = {nullptr, nullptr, nullptr};
    hipGraphExec_t exec_s[N_ATTN_VARIANTS] = {nullptr, nullptr, nullptr};
    float* p16_scores = nullptr;   // [n_heads][p16_sc_stride] scaled attention scores between the two reference-order attention launches
    int p16_sc_stride = 0, p16_att_splits = 16, p16_depth = 8;
    // stop tokens (jh_session_set_eos): device copy for finish_token_kernel + host-side feeding control
    int* eos_dev = nullptr;   // fixed buffer [count, id0, id1, ...] read by finish_token_kernel: changing the list re-captures nothing
    int n_eos = 0;
    int eos_host[JH_MAX_EOS] = {0};
    DecodeState* st_host = nullptr;   // pinned: state snapshots the host polls between chunks of graph replays
    hipEvent_t ev_chunk[2] = {nullptr, nullptr};
    int generated = 0;
};


constexpr int ROPE_MARGIN = 512;

// ------------------------------------------------------------------------------------------------ prototypes by file
// model.hip
bool is_global_slot(int which);
extern thread_local int g_operand_packs;
bool fast_gateup_t16(jh_model* m);
bool t16_gateup_ok(const jh_model* m, int li);
int ensure_gateup_t16(jh_model* m, int li, hipStream_t st);
bool t16_weight_ok(const JWeight& W);
int ensure_t16(JWeight& W, hipStream_t st);
int ensure_p16t(JWeight& W, hipStream_t st);
int use_p16t(GemvParams& p, const JWeight& W);
int ensure_strict_operands(jh_session* s, hipStream_t st);
int refuse_order_free(const jh_session* s, const char* what);   // JH_STRICT_ONLY models: order-free use fails loudly
// layers.hip
int attn_launch(jh_session* s, int rel, hipStream_t st, bool tap, long long* dbg = nullptr);
int tap_copy(jh_session* s, int which, const float* src, int n, hipStream_t st);
int tp_push_gemv(jh_session* s, GemvParams& p, const LaunchCfg& cfg, hipStream_t st);
int layer_attn_launch(jh_session* s, int li, hipStream_t st, bool tap, int pos_for_tap, float* out, const float* resid);
int layer_ffn_launch(jh_session* s, int li, hipStream_t st, bool tap, float* out, const float* resid);
int layer_launch(jh_session* s, int li, hipStream_t st, bool tap, int pos_for_tap);
int layers_launch(jh_session* s, hipStream_t st, int pos_for_tap);
const JWeight* lm_head_weight(jh_model* m);
int lmhead_launch(jh_session* s, hipStream_t st);
int finish_launch(jh_session* s, hipStream_t st, int do_embed, float temperature = 0.0f);
int ensure_out_tokens(jh_session* s, int n);
// prefill.hip
bool prefill_p16_ok(jh_session* s);
bool prefill_t16_ok(jh_session* s);
bool prefill_bf16r_ok(jh_session* s);
bool prefill_batch_ok(jh_session* s);
size_t prefill_attn_lds(const jh_config& c, int n_keys);
bool prefill_attn_mfma(const jh_session* s, int start_pos, int rows);
bool prefill_chunk_fits(jh_session* s, int start_pos, int rows);
int prefill_alloc(jh_session* s);
size_t tiled_w_bytes(const JWeight& W);
size_t tiled_s_bytes(const JWeight& W);
int tiled_mode_for(jh_model* m);
int ensure_all_tiled(jh_session* s, hipStream_t st);
int prefill_p16_operands(jh_session* s);
int prefill_attn_half(jh_session* s, int li, int rows, int nkeys_bound, bool attn_mfma, float* out, const float* resid, hipStream_t st);
int prefill_ffn_half(jh_session* s, int li, int rows, const float* x1, float* out, const float* resid, hipStream_t st);
int prefill_attn_half_p16(jh_session* s, int li, int rows, int start_pos, float* out, const float* resid, hipStream_t st);
int prefill_ffn_half_p16(jh_session* s, int li, int rows, const float* x1, float* out, const float* resid, hipStream_t st);
int prefill_attn_half_bf16r(jh_session* s, int li, int rows, int start_pos, float* out, const float* resid, hipStream_t st);
int prefill_ffn_half_bf16r(jh_session* s, int li, int rows, const float* x1, float* out, const float* resid, hipStream_t st);
int prefill_chunk(jh_session* s, const int32_t* tokens, const float* x_in, bool x_in_dev, int rows, int start_pos, float* x_out, bool x_out_dev,
                  hipStream_t st);
// decode.hip
int check_positions(const jh_session* s, int last_pos);
extern "C" {   // (defined between the C ABI entry points of decode.hip; hidden like everything that is not in include/jlama_hip.h)
void drop_stale_graphs(jh_session* s);
int attn_variant_for(const jh_session* s, int pos);
bool attn_variant_in_range(const jh_session* s, int v, int first, int last);
int build_row_graph(jh_session* s, int v);
int build_graph(jh_session* s, int v, float temperature = 0.0f);
int forward_impl(jh_session* s, const int32_t* tokens, const float* x_in, bool x_in_dev, int n, int start_pos, float* x_out, bool x_out_dev);
}
