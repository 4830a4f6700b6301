// core.hip -- part of libjlamahip.so (C ABI: include/jlama_hip.h).  Errors, per-thread context, process options, jh_init, registered tensors.
#include "jh_host.h"

// ------------------------------------------------------------------------------------------------ errors / context
thread_local std::string g_err;
int set_err(int code, const std::string& msg) {
    g_err = msg;
    return code;
}


thread_local ThreadCtx tctx;
int g_default_device = 0;
int g_cu_count = 256;

int ensure_ctx() {
    if (tctx.device >= 0) {
        HIPCHK(hipSetDevice(tctx.device));
        return JH_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return set_err(JH_ERR_NO_DEVICE, "no HIP device (hipGetDeviceCount)");
    int dev = g_default_device < n ? g_default_device : 0;
    HIPCHK(hipSetDevice(dev));
    HIPCHK(hipStreamCreateWithFlags(&tctx.stream, hipStreamNonBlocking));
    tctx.device = dev;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cu_count = prop.multiProcessorCount;
    return JH_OK;
}

int dev_buf(int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    if (tctx.cap[slot] < bytes) {
        if (tctx.scratch[slot]) HIPCHK(hipFree(tctx.scratch[slot]));
        tctx.scratch[slot] = nullptr;
        tctx.cap[slot] = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&tctx.scratch[slot], want);
        if (e != hipSuccess) return set_err(JH_ERR_OOM, std::string("hipMalloc scratch: ") + hipGetErrorString(e));
        tctx.cap[slot] = want;
    }
    *out = tctx.scratch[slot];
    return JH_OK;
}

std::mutex g_reg_mu;
std::unordered_map<int64_t, RegTensor> g_reg;
int64_t g_next_id = 1;

const void* reg_ptr(int64_t id) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_reg.find(id);
    return it == g_reg.end() ? nullptr : it->second.ptr;
}

// Process options.  The library does NOT read tuning knobs from the process environment (a Java host would inherit whatever its
// launcher exported): an option exists only after jh_set_option() -- the host's explicit decision, used by tests and tools/ -- with
// one exception, the documented handful that jh_init copies from the environment ONCE (JH_ENV_OPTIONS below).  Everything else is
// a constant chosen by the launch planners.
std::mutex g_opt_mu;
std::map<std::string, int> g_opts;
std::map<std::string, int> g_env_opts;   // what jh_init copied from the environment: jh_clear_options() falls back to these
int opt_int(const char* name, int dflt) {
    std::lock_guard<std::mutex> lk(g_opt_mu);
    auto it = g_opts.find(name);
    return it == g_opts.end() ? dflt : it->second;
}
// every name opt_int() is asked for anywhere in the library (tests/test_abi.py keeps this list and the call sites in step): a
// misspelt option is an error, not a silent no-op
const char* const JH_KNOWN_OPTIONS[] = {
    "JH_ATTN_LONG_MIN", "JH_ATTN_LONG_SPLITS", "JH_ATTN_MID_MAX", "JH_ATTN_MID_SPLITS", "JH_ATTN_SPLITS", "JH_BF16_CW2", "JH_BF16_CWB",
    "JH_BF16_LDS", "JH_BF16R_DEEP", "JH_BF16R_PREFILL", "JH_BF16_S", "JH_BF16_W8", "JH_DOWN_GRIDX", "JH_DOWN_PIPE", "JH_DOWN_R", "JH_DOWN_WAVES", "JH_FAST_GATEUP_T16",
    "JH_GATEUP_GRIDX", "JH_GATEUP_PIPE", "JH_GATEUP_R", "JH_GATEUP_WAVES", "JH_GEMM_CW", "JH_GEMM_LDS", "JH_GEMM_LDS_CT",
    "JH_GEMM_LDS_CW", "JH_GEMM_LDS_PK", "JH_GEMM_LDS_S", "JH_GEMM_S", "JH_GEMM_Z", "JH_GEMV_PIPE", "JH_GEMV_R", "JH_GEMV_WAVES",
    "JH_LM_GRIDX", "JH_LM_R", "JH_LM_WAVES", "JH_NO_GRAPH", "JH_O_GRIDX", "JH_O_PIPE", "JH_O_R", "JH_O_WAVES",
    "JH_P16_ATT_FUSED", "JH_P16_ATT_SPLITS", "JH_P16_AV_SEQ_MIN", "JH_P16_D", "JH_P16_PREFILL", "JH_PREFILL_ATTN_MFMA_MIN", "JH_PREFILL_BATCH_MIN", "JH_PREFILL_GRAPH", "JH_T16_GEMM32", "JH_T16_TAIL_FORCE", "JH_T16_TAIL_ROWS",
    "JH_PREFILL_TILED", "JH_QKV_GRIDX", "JH_QKV_PIPE", "JH_QKV_R", "JH_QKV_WAVES", "JH_STRICT_ONLY", "JH_STRICT_ORDER", "JH_T16",
    "JH_T16_PREFILL", "JH_TIER1_GENERIC", "JH_TOKENS_PER_GRAPH", "JH_TILED_COPY", "JH_TP_CU_MASK", "JH_TP_FUSE", "JH_TP_GRAPH", "JH_TP_LOUD",
    "JH_TRACE",
};
// JH_TRACE=1          synchronize + report after every launch (debugging)
// JH_NO_GRAPH=1       decode without hipGraph replay (debugging)
// JH_STRICT_ORDER=1   new sessions start in reference order (default 0: order-free kernels; jh_session_set_strict switches a session)
// JH_TILED_COPY=auto|resident|transient   where the order-free prefill GEMM's MFMA-ordered weight operand lives
// JH_TP_LOUD=1        a tensor-parallel meeting that times out is an error instead of a (reported) fall-back to the event loop
const char* const JH_ENV_OPTIONS[] = {"JH_TRACE", "JH_NO_GRAPH", "JH_STRICT_ORDER", "JH_TILED_COPY", "JH_TP_LOUD"};
void options_from_environment_once() {
    static bool done = false;
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (done) return;
    done = true;
    for (const char* name : JH_ENV_OPTIONS) {
        const char* v = getenv(name);
        if (!v || !*v || g_opts.count(name)) continue;
        int val = atoi(v);
        if (!strcmp(name, "JH_TILED_COPY")) val = !strcmp(v, "resident") ? 1 : !strcmp(v, "transient") ? 2 : 0;
        g_opts[name] = val;
        g_env_opts[name] = val;
    }
}


std::mutex g_capture_mu;   // one hipGraph capture at a time per process (captures are rare; concurrent ones from different host threads are fragile)
int trace_sync(const char* what, hipStream_t st) {
    if (!opt_int("JH_TRACE", 0)) return JH_OK;   // read when asked (no latch: jh_set_option / jh_clear_options take effect at once)
    fprintf(stderr, "[jh] %s ...", what);
    fflush(stderr);
    hipError_t e = hipStreamSynchronize(st);
    fprintf(stderr, " %s\n", hipGetErrorString(e));
    fflush(stderr);
    return e == hipSuccess ? JH_OK : set_err(JH_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

int nb_for(int K) {
    const int nblk = K / QB;
    if (nblk % 64) return 0;
    const int nb = nblk / 64;
    return (nb == 1 || nb == 2 || nb == 4 || nb == 7) ? nb : 0;
}

thread_local int g_last_gemv_grid = 0;                // workgroups of the GEMV launched last on this thread (EPI_TP: its flag count)

// ------------------------------------------------------------------------------------------------ runtime facts
extern "C" {

int jh_init(int device, int64_t* out_info) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) return set_err(JH_ERR_NO_DEVICE, "no HIP device (hipGetDeviceCount)");
    if (device < 0 || device >= n) return set_err(JH_ERR_INVALID, "device ordinal out of range");
    options_from_environment_once();
    g_default_device = device;
    if (tctx.device != device) {
        tctx.device = -1;  // re-create the per-thread stream on the new device
        tctx.stream = nullptr;
        for (int i = 0; i < NSCRATCH; i++) { tctx.scratch[i] = nullptr; tctx.cap[i] = 0; }
    }
    JHCHK(ensure_ctx());
    if (out_info) {
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        out_info[0] = (int64_t)fr;
        out_info[1] = prop.multiProcessorCount;
        out_info[2] = n;
        out_info[3] = (int64_t)prop.maxSharedMemoryPerMultiProcessor;
    }
    return JH_OK;
}
int jh_set_option(const char* name, int32_t value) {
    if (!name || !*name) return set_err(JH_ERR_INVALID, "set_option: null name");
    bool known = false;
    for (const char* k : JH_KNOWN_OPTIONS) known = known || !strcmp(k, name);
    if (!known) return set_err(JH_ERR_INVALID, std::string("set_option: the library has no option named ") + name);
    std::lock_guard<std::mutex> lk(g_opt_mu);
    g_opts[name] = value;
    return JH_OK;
}
int jh_clear_options(void) {
    std::lock_guard<std::mutex> lk(g_opt_mu);
    g_opts = g_env_opts;   // explicit options go; the process-wide environment snapshot of jh_init stays in force
    return JH_OK;
}
const char* jh_name(void) { return "HIP CDNA4 (gfx950) Operations"; }
int jh_parallel_split_size(void) { return 1; }
int jh_preferred_working_qtype(void) { return JH_DT_I8; }
const char* jh_last_error(void) { return g_err.c_str(); }
#ifndef JH_SRC_HASH
#define JH_SRC_HASH "unknown"
#endif
static const char g_src_hash_marker[] = "JHSRCHASH:" JH_SRC_HASH;   // the host side finds it by scanning the file (no dlopen)
const char* jh_source_hash(void) { return g_src_hash_marker + 10; }
int jh_abi_config_layout(int32_t* out, int n) {
    const int32_t v[] = {(int32_t)sizeof(jh_config),
                         (int32_t)offsetof(jh_config, embedding_length), (int32_t)offsetof(jh_config, hidden_length),
                         (int32_t)offsetof(jh_config, n_heads), (int32_t)offsetof(jh_config, n_kv_heads), (int32_t)offsetof(jh_config, head_size),
                         (int32_t)offsetof(jh_config, n_layers), (int32_t)offsetof(jh_config, vocab_size), (int32_t)offsetof(jh_config, context_length),
                         (int32_t)offsetof(jh_config, weight_dtype), (int32_t)offsetof(jh_config, layer_start), (int32_t)offsetof(jh_config, layer_end),
                         (int32_t)offsetof(jh_config, rms_eps), (int32_t)offsetof(jh_config, rope_theta), (int32_t)offsetof(jh_config, rope_scaling)};
    const int cnt = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; out && i < n && i < cnt; i++) out[i] = v[i];
    return cnt;
}
int jh_synchronize(void) {
    JHCHK(ensure_ctx());
    HIPCHK(hipStreamSynchronize(tctx.stream));
    return JH_OK;
}

// ------------------------------------------------------------------------------------------------ Tier 1
int64_t jh_register_tensor(const void* host, int64_t bytes) {
    if (!host || bytes <= 0) return set_err(JH_ERR_INVALID, "jh_register_tensor: null/empty");
    int rc = ensure_ctx();
    if (rc != JH_OK) return rc;
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, (size_t)bytes + 64);
    if (e != hipSuccess) return set_err(JH_ERR_OOM, std::string("hipMalloc weight: ") + hipGetErrorString(e));
    e = hipMemcpy(d, host, (size_t)bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(d);
        return set_err(JH_ERR_HIP, std::string("hipMemcpy weight: ") + hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(g_reg_mu);
    int64_t id = g_next_id++;
    g_reg[id] = RegTensor{d, bytes, tctx.device};
    return id;
}
int jh_unregister_tensor(int64_t id) {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    auto it = g_reg.find(id);
    if (it == g_reg.end()) return set_err(JH_ERR_INVALID, "unknown tensor id");
    hipFree(it->second.ptr);
    g_reg.erase(it);
    return JH_OK;
}

}  // extern "C"
