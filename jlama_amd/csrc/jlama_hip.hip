// jlama_hip.hip -- C ABI of libjlamahip.so (see include/jlama_hip.h for the contract and the reference
// interfaces each entry point replaces).  HIP runtime only: no torch, no oracle, no CPU fallback -- every
// compute entry point fails with JH_ERR_NO_DEVICE / JH_ERR_HIP when no MI355X is usable.
#include "../../include/jlama_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <map>
#include <vector>

#include "jh_kernels.h"
#include "jh_t16.h"
#include "jh_bf16r.h"

using namespace jh;

// ------------------------------------------------------------------------------------------------ errors / context
static thread_local std::string g_err;
static int set_err(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
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

namespace {

constexpr int NSCRATCH = 8;
struct ThreadCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    void* scratch[NSCRATCH] = {nullptr};
    size_t cap[NSCRATCH] = {0};
};
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

struct RegTensor {
    void* ptr;
    int64_t bytes;
    int device;
};
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
    "JH_ATTN_LONG_MIN", "JH_ATTN_LONG_SPLITS", "JH_ATTN_MID_MAX", "JH_ATTN_MID_SPLITS", "JH_ATTN_SPLITS", "JH_BF16_CWB",
    "JH_BF16_LDS", "JH_BF16R_PREFILL", "JH_BF16_S", "JH_DOWN_GRIDX", "JH_DOWN_PIPE", "JH_DOWN_R", "JH_DOWN_WAVES", "JH_FAST_GATEUP_T16",
    "JH_GATEUP_GRIDX", "JH_GATEUP_PIPE", "JH_GATEUP_R", "JH_GATEUP_WAVES", "JH_GEMM_CW", "JH_GEMM_LDS", "JH_GEMM_LDS_CT",
    "JH_GEMM_LDS_CW", "JH_GEMM_LDS_PK", "JH_GEMM_LDS_S", "JH_GEMM_S", "JH_GEMM_Z", "JH_GEMV_PIPE", "JH_GEMV_R", "JH_GEMV_WAVES",
    "JH_LM_GRIDX", "JH_LM_R", "JH_LM_WAVES", "JH_NO_GRAPH", "JH_O_GRIDX", "JH_O_PIPE", "JH_O_R", "JH_O_WAVES",
    "JH_P16_ATT_SPLITS", "JH_P16_D", "JH_P16_PREFILL", "JH_PREFILL_ATTN_MFMA_MIN", "JH_PREFILL_BATCH_MIN", "JH_PREFILL_GRAPH",
    "JH_PREFILL_TILED", "JH_QKV_GRIDX", "JH_QKV_PIPE", "JH_QKV_R", "JH_QKV_WAVES", "JH_STRICT_ORDER", "JH_T16",
    "JH_T16_PREFILL", "JH_TIER1_GENERIC", "JH_TILED_COPY", "JH_TP_CU_MASK", "JH_TP_FUSE", "JH_TP_GRAPH", "JH_TP_LOUD",
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

template <typename K>
int allow_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return JH_OK;
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

// ---- launchers -------------------------------------------------------------------------------------------
struct LaunchCfg { int R, waves, grid_cap, pipe; };   // 0 / -1 = let the planner decide
thread_local int g_last_gemv_grid = 0;                // workgroups of the GEMV launched last on this thread (EPI_TP: its flag count)

// (R, NB, PIPE) instantiations of gemv_i8q4_kernel
#define JH_GEMV_COMBOS(X)                                                                                         \
    X(1, 1, 0) X(1, 2, 0) X(1, 4, 0) X(1, 7, 0) X(2, 1, 0) X(2, 2, 0) X(2, 4, 0) X(2, 7, 0) X(4, 1, 0) X(4, 2, 0)  \
    X(4, 4, 0) X(8, 1, 0) X(8, 2, 0) X(14, 2, 0)                                                                  \
    X(2, 2, 1) X(4, 2, 1) X(2, 4, 1) X(4, 1, 1) X(8, 1, 1) X(2, 7, 1) X(1, 0, 1) X(2, 0, 1) X(4, 0, 1)

template <int PRO, int EPI>
int launch_gemv_i8q4_combo(const GemvParams& p, int R, int NB, int PIPE, int grid, int threads, hipStream_t st) {
    const size_t lds = lds_bytes_i8(p.K);
#define X(RV, NBV, PV)                                                                                          \
    if (R == RV && NB == NBV && PIPE == PV) {                                                                   \
        if constexpr (!(EPI == EPI_SILU_MUL && ((RV) & 1))) {                                                   \
            JHCHK(allow_lds(gemv_i8q4_kernel<PRO, EPI, RV, NBV, PV>, lds));                                     \
            hipLaunchKernelGGL((gemv_i8q4_kernel<PRO, EPI, RV, NBV, PV>), dim3(grid), dim3(threads), lds, st, p); \
            g_last_gemv_grid = grid;                                                                            \
            HIPCHK(hipGetLastError());                                                                          \
            return JH_OK;                                                                                       \
        }                                                                                                       \
    }
    JH_GEMV_COMBOS(X)
#undef X
    return set_err(JH_ERR_INVALID, "gemv: no kernel instantiation for R=" + std::to_string(R) + " NB=" + std::to_string(NB) +
                                       " PIPE=" + std::to_string(PIPE));
}

// Plan a GEMV launch.  Goal (tools/membw.hip calibration): every CU gets one workgroup whose waves each hold
// R*NB >= ~8 independent 16-byte loads, ALL issued before the activation prologue (PIPE=0).  When a wave's share of
// rows does not fit in registers, fall back to the software-pipelined loop (PIPE=1).
template <int PRO, int EPI>
int launch_gemv_i8q4(const GemvParams& p, LaunchCfg cfg, hipStream_t st) {
    const int total = (EPI == EPI_SILU_MUL) ? 2 * p.nrows : p.nrows;
    const int nb = nb_for(p.K);
    const int cu = g_cu_count;
    auto divides = [&](int R) {
        if (EPI == EPI_SILU_MUL) return (R % 2 == 0) && (p.nrows % (R / 2) == 0);
        return p.nrows % R == 0;
    };
    int R = cfg.R, pipe = cfg.pipe, waves = cfg.waves;
    if (nb == 0) {
        pipe = 1;
        if (R <= 0) R = 2;
        while (R > 1 && !divides(R)) R >>= 1;
        if (EPI == EPI_SILU_MUL && R < 2) return set_err(JH_ERR_INVALID, "gate/up GEMV needs an even hidden length");
        if (waves <= 0) waves = 8;
    } else if (R <= 0 || pipe < 0) {
        // Measured on MI355X (tools/gemv_lab2.hip, gate/up-sized matrix): issuing a wave's whole weight stream before the
        // prologue ("single shot", PIPE=0) only helps when the prologue is trivial -- a CU keeps ~32-64 KB of loads in
        // flight, so a wave that queues more than that is still ISSUING loads when its prologue should already run.
        // The software-pipelined loop with ~8 KB per wave prefetched (R*NB = 4..8) is the better default.
        if (pipe < 0) {
            // small/medium GEMVs (<= ~200 KB of weights per CU) run best single-shot, the big gate/up one pipelined
            const double bytes_per_cu = (double)total * p.K * 0.625 / cu;
            pipe = bytes_per_cu <= 200e3 ? 0 : 1;
        }
        if (R <= 0) {
            if (pipe >= 1) R = (nb == 1) ? 4 : 2;
            else {
                const int need = (total + cu * 16 - 1) / (cu * 16);
                static const int oneshot[] = {1, 2, 4, 8, 14};
                R = 1;
                for (int r : oneshot) {
                    const bool inst = (r == 1 || r == 2) || (r == 4 && nb <= 4) || (r == 8 && nb <= 2) || (r == 14 && nb == 2);
                    if (inst && r >= need && r * nb <= 28 && divides(r)) { R = r; break; }
                }
            }
        }
        while (R > 2 && !divides(R)) R >>= 1;
    }
    const int ngroups = total / R;
    int grid, threads;
    if (pipe == 0) {
        // VGPR budget: 1024-thread blocks are capped at 128 registers
        int wmax = (R * nb <= 4) ? 16 : 8;   // must match gemv_i8q4_kernel __launch_bounds__
        if (waves <= 0) {
            waves = (ngroups + cu - 1) / cu;
            if (waves < 4) waves = 4;
            if (waves > wmax) waves = wmax;
        }
        if (waves > wmax) waves = wmax;
        grid = (ngroups + waves - 1) / waves;
    } else {
        const int wmax = (nb > 0 && R * nb <= 4) ? 16 : 8;   // must match gemv_i8q4_kernel __launch_bounds__
        if (waves <= 0) waves = 8;
        if (waves > wmax) waves = wmax;
        grid = (ngroups + waves - 1) / waves;
        const int cap = cfg.grid_cap > 0 ? cfg.grid_cap : cu;
        if (grid > cap) grid = cap;
    }
    if (grid < 1) grid = 1;
    threads = waves * 64;
    return launch_gemv_i8q4_combo<PRO, EPI>(p, R, nb, pipe, grid, threads, st);
}

template <int PRO, int EPI, bool ARGMAX>
int launch_gemv_bf16(const GemvParams& p, int grid_cap, int* grid_out, hipStream_t st) {
    const size_t lds = lds_bytes_bf(p.K);
    const int total = (EPI == EPI_SILU_MUL) ? 2 * p.nrows : p.nrows;
    constexpr int R = (EPI == EPI_SILU_MUL) ? 2 : 1;   // 8 x 16-byte loads per row already keep a wave's queue full
    const int ngroups = total / R, waves = 8;
    int grid = (ngroups + waves - 1) / waves;
    if (grid > grid_cap) grid = grid_cap;
    if (grid < 1) grid = 1;
    if (grid_out) *grid_out = grid;
    JHCHK(allow_lds(gemv_bf16_kernel<PRO, EPI, R, ARGMAX>, lds));
    hipLaunchKernelGGL((gemv_bf16_kernel<PRO, EPI, R, ARGMAX>), dim3(grid), dim3(waves * 64), lds, st, p);
    HIPCHK(hipGetLastError());
    return JH_OK;
}

template <int MT>
int launch_gemm_q8q4_mfma_mt(const MfmaQ4Params& g, hipStream_t st) {
    const int tiles = g.n / 32;
    const int nblk = g.k / QB;
    // split-K so that small N still fills the chip: one workgroup per 32-column tile, KSPLIT waves each
    const int ksplit = (tiles >= g_cu_count * 4 || nblk < 8) ? 1 : (tiles >= g_cu_count * 2 || nblk < 16 ? 2 : 4);
    const size_t stage = (size_t)MT * 32 * MQ_ASTRIDE + 2 * (size_t)MT * 32 * 4;
    const size_t red = ksplit > 1 ? (size_t)ksplit * MT * 16 * 64 * 4 : 0;
    size_t lds = (size_t)ksplit * 2 * stage;
    if (red > lds) lds = red;
    if (ksplit == 4) { JHCHK(allow_lds(gemm_q8q4_mfma_kernel<MT, 4>, lds)); hipLaunchKernelGGL((gemm_q8q4_mfma_kernel<MT, 4>), dim3(tiles), dim3(256), lds, st, g); }
    else if (ksplit == 2) { JHCHK(allow_lds(gemm_q8q4_mfma_kernel<MT, 2>, lds)); hipLaunchKernelGGL((gemm_q8q4_mfma_kernel<MT, 2>), dim3(tiles), dim3(128), lds, st, g); }
    else { JHCHK(allow_lds(gemm_q8q4_mfma_kernel<MT, 1>, lds)); hipLaunchKernelGGL((gemm_q8q4_mfma_kernel<MT, 1>), dim3(tiles), dim3(64), lds, st, g); }
    HIPCHK(hipGetLastError());
    return JH_OK;
}
template <int S, bool TILED, int CW>
int launch_gemm_q8q4_tile(const MfmaQ4Params& g, int mtiles, hipStream_t st) {
    const int cgroups = g.n / (32 * CW), gg = (cgroups + 7) / 8;
    const size_t lds_scales = (size_t)(g.k / QB) * 32 * 4, lds_red = S > 1 ? (size_t)CW * S * 16 * 64 * 4 : 0;
    const size_t lds = lds_scales > lds_red ? lds_scales : lds_red;
    JHCHK(allow_lds((gemm_q8q4_tile_kernel<S, TILED, CW>), lds));
    hipLaunchKernelGGL((gemm_q8q4_tile_kernel<S, TILED, CW>), dim3(8 * mtiles * gg), dim3(S * CW * 64), lds, st, g, mtiles);
    HIPCHK(hipGetLastError());
    return JH_OK;
}

// gemm_q8q4_lds_kernel: one row tile x CW*CT column tiles per workgroup, A staged through LDS; K split over grid.y when the
// output alone does not fill the chip (partials in ws, summed in ascending K order by splitk_reduce_kernel)
// second pass of a split-K GEMM (partials summed in ascending K order): four columns per thread where the shapes allow (the
// one-float-per-thread form spent 19 us on the 2 x 14.8 MB of the BF16 gate|up GEMM at 129 rows: 64-bit divisions per element)
int launch_splitk_reduce(const float* ws, int S, int m, int n, int n0, float* c, int ldc, int roffset, const float* resid, hipStream_t st) {
    const int coff = n0 - roffset;
    if (n % 4 == 0 && ldc % 4 == 0 && coff % 4 == 0 && ((uintptr_t)ws | (uintptr_t)c | (uintptr_t)resid) % 16 == 0) {
        hipLaunchKernelGGL(splitk_reduce4_kernel, dim3((unsigned)((n / 4 + 255) / 256), (unsigned)m), dim3(256), 0, st, (const f32x4*)ws, S, m, n / 4,
                           (f32x4*)c, ldc / 4, coff / 4, (const f32x4*)resid);
    } else {
        const size_t tot = (size_t)m * n;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, ws, S, m, n, n0, c, ldc, roffset, resid);
    }
    HIPCHK(hipGetLastError());
    return JH_OK;
}

template <int CW, int CT, int S, bool PK = false>
int launch_gemm_q8q4_lds(const MfmaQ4Params& g, int mtiles, float* ws, size_t ws_bytes, hipStream_t st) {
    const int nblk = g.k / QB;
    const int cgroups = g.n / (32 * CW * CT), gg = (cgroups + 7) / 8;
    int Z = 1;
    const int z_env = opt_int("JH_GEMM_Z", 0);
    while (Z < 8 && (long long)mtiles * cgroups * Z < (long long)g_cu_count * 2 && nblk % (16 * S * Z) == 0 &&
           ws && (size_t)(2 * Z) * g.m * g.n * 4 <= ws_bytes) Z *= 2;
    if (z_env > 0 && nblk % (8 * S * z_env) == 0 && (z_env == 1 || (ws && (size_t)z_env * g.m * g.n * 4 <= ws_bytes))) Z = z_env;
    const int nbz = nblk / Z;
    size_t lds = (size_t)nbz * 128 + (size_t)S * 2 * 4 * 1024;
    const size_t red = S > 1 ? (size_t)CW * S * CT * 16 * 64 * 4 : 0;
    if (red > lds) lds = red;
    JHCHK(allow_lds((gemm_q8q4_lds_kernel<CW, CT, S, PK>), lds));
    hipLaunchKernelGGL((gemm_q8q4_lds_kernel<CW, CT, S, PK>), dim3(8 * mtiles * gg, Z), dim3(CW * S * 64), lds, st, g, mtiles, nbz, Z > 1 ? ws : nullptr);
    HIPCHK(hipGetLastError());
    if (Z > 1) JHCHK(launch_splitk_reduce((const float*)ws, Z, g.m, g.n, g.n0, g.c, g.ldc, g.roffset, g.resid, st));
    return JH_OK;
}

// tiled = both operands already in MFMA order (see gemm_q8q4_tile_kernel)
int launch_gemm_q8q4_mfma(const MfmaQ4Params& g, hipStream_t st, bool tiled = false, float* ws = nullptr, size_t ws_bytes = 0) {
    const int mt = (g.m + 31) / 32;
    const int nblk = g.k / QB;
    if (nblk % 8 == 0 && (size_t)nblk * 128 <= 150 * 1024 && (tiled || (g.lda % 16) == 0) && true) {
        // one 32x32 output tile per wave; split K over S waves until the chip has >= ~8 waves per CU; CW column tiles per
        // workgroup share the A tile through L1 (S*CW <= 8 waves: several workgroups per CU keep the CUs evenly loaded)
        // gemm_q8q4_lds_kernel (A through LDS, K split over the waves of a workgroup) for the GEMMs with enough work per column
        // group -- gate|up and down: 71 vs 78 us and 45 vs 56 us at M = 129, 96 vs 102 and 54 vs 67 at M = 256 -- the tile kernel
        // below for q|k|v and the o-projection (21 vs 30 us).  JH_GEMM_LDS = 0 / 1 forces one of them.
        const int lds_env = opt_int("JH_GEMM_LDS", -1);   // read per call: the tests flip it within one process
        const bool lds_auto = (long long)(g.n / 32) * nblk >= (long long)896 * 64;
        if (tiled && (lds_env > 0 || (lds_env < 0 && lds_auto))) {
            const int cw_l = opt_int("JH_GEMM_LDS_CW", 4), ct_l = opt_int("JH_GEMM_LDS_CT", 1), s_l = opt_int("JH_GEMM_LDS_S", 2);
            int CWL = cw_l, CTL = ct_l > 2 ? 2 : ct_l, SL = s_l;
            while (CTL > 1 && g.n % (32 * CWL * CTL)) CTL >>= 1;
            while (CWL > 1 && g.n % (32 * CWL * CTL)) CWL >>= 1;
            while (SL > 1 && nblk % (8 * SL)) SL >>= 1;
            while (SL > 1 && (size_t)nblk * 128 + (size_t)SL * 8192 > 150 * 1024) SL >>= 1;   // scale slice + the A rings of the SL K slices
            const bool lds_fits = (size_t)nblk * 128 + (size_t)SL * 8192 <= 150 * 1024;
            if (!lds_fits) CWL = -1;   // no instantiation below matches: the tile kernel takes it
            if (CWL == 4 && CTL == 1 && SL == 2 && opt_int("JH_GEMM_LDS_PK", 0)) return launch_gemm_q8q4_lds<4, 1, 2, true>(g, mt, ws, ws_bytes, st);
#define JH_LDS(CV, TV, SV) if (CWL == CV && CTL == TV && SL == SV) return launch_gemm_q8q4_lds<CV, TV, SV>(g, mt, ws, ws_bytes, st);
            JH_LDS(4, 1, 1) JH_LDS(4, 1, 2) JH_LDS(4, 1, 4) JH_LDS(2, 1, 2) JH_LDS(2, 1, 4) JH_LDS(2, 1, 8) JH_LDS(4, 2, 1) JH_LDS(4, 2, 2) JH_LDS(2, 2, 2)
            JH_LDS(1, 1, 4) JH_LDS(1, 1, 8) JH_LDS(2, 1, 1) JH_LDS(1, 1, 1) JH_LDS(1, 1, 2)
#undef JH_LDS
        }
        const long long tiles = (long long)mt * (g.n / 32);
        int S = 1;
        while (S < 8 && tiles * S < (long long)g_cu_count * 8 && nblk % (16 * S) == 0) S *= 2;   // nblk/S stays a multiple of 8
        const int cw_env = opt_int("JH_GEMM_CW", 0), s_env = opt_int("JH_GEMM_S", 0);
        if (s_env > 0 && nblk % (8 * s_env) == 0) S = s_env;
        int CW = 1;
        if (tiled) {
            CW = S <= 4 ? 4 : 2;                       // S*CW <= 16 waves
            if (cw_env > 0 && cw_env * S <= 16) CW = cw_env;
            while (CW > 1 && (g.n % (32 * CW)) != 0) CW >>= 1;
        }
#define JH_TILE(SV, CV) if (S == SV && CW == CV) return tiled ? launch_gemm_q8q4_tile<SV, true, CV>(g, mt, st) : launch_gemm_q8q4_tile<SV, false, 1>(g, mt, st);
        JH_TILE(1, 1) JH_TILE(2, 1) JH_TILE(4, 1) JH_TILE(8, 1) JH_TILE(1, 2) JH_TILE(2, 2) JH_TILE(4, 2) JH_TILE(1, 4) JH_TILE(2, 4) JH_TILE(8, 2) JH_TILE(1, 8) JH_TILE(4, 4)
#undef JH_TILE
        return set_err(JH_ERR_INVALID, "tile GEMM: no instantiation for this (S, CW)");
    }
    if (tiled) return set_err(JH_ERR_UNSUPPORTED, "tiled I8xQ4 GEMM needs K % 256 == 0");
    switch (mt) {
        case 1: return launch_gemm_q8q4_mfma_mt<1>(g, st);
        case 2: return launch_gemm_q8q4_mfma_mt<2>(g, st);
        case 3: case 4: return launch_gemm_q8q4_mfma_mt<4>(g, st);
        case 5: case 6: return launch_gemm_q8q4_mfma_mt<6>(g, st);
        // M > 192 with K % 256 != 0: the all-of-M-per-wave kernel would need 8 accumulator tiles and spills (156 VGPRs
        // to scratch when it existed); such shapes take the generic kernel instead
        default: return set_err(JH_ERR_UNSUPPORTED, "I8xQ4 MFMA GEMM: M > 192 needs K % 256 == 0");
    }
}

template <int MT>
int launch_gemm_bf16_mfma_mt(const MfmaGemmParams& g0, hipStream_t st) {
    MfmaGemmParams g = g0;
    const int tiles = g.n / 32;
    // 4 column tiles (waves) per workgroup share one staged A slice (A is 6x the W bytes per slice at M=192, so a
    // single-wave workgroup spends its time re-staging A); parallelism comes from splitting K over workgroup rows
    // (partials to the caller's workspace, splitk_reduce_kernel adds them in K order)
    const int waves = tiles >= 4 ? 4 : (tiles >= 2 ? 2 : 1);
    const int grid = (tiles + waves - 1) / waves;
    int S = 1;
    if (g.ws) {
        const int nslices = g.k / MG_KS;
        while (S < 16 && grid * S < g_cu_count * 3 && nslices % (2 * S) == 0 && nslices / (2 * S) >= 2 &&
               (size_t)(2 * S) * g.n <= (size_t)8 * 16384 && (g.n <= 8192 || 2 * S * g.m <= 640)) S *= 2;   // wide outputs: the reduce pass costs S*M*N*8 bytes
    }
    g.nsplit = S;
    const size_t lds = (size_t)2 * MT * 32 * MG_ASTRIDE;
    if (waves == 4) { JHCHK(allow_lds(gemm_bf16_mfma_kernel<MT, 4>, lds)); hipLaunchKernelGGL((gemm_bf16_mfma_kernel<MT, 4>), dim3(grid, S), dim3(256), lds, st, g); }
    else if (waves == 2) { JHCHK(allow_lds(gemm_bf16_mfma_kernel<MT, 2>, lds)); hipLaunchKernelGGL((gemm_bf16_mfma_kernel<MT, 2>), dim3(grid, S), dim3(128), lds, st, g); }
    else { JHCHK(allow_lds(gemm_bf16_mfma_kernel<MT, 1>, lds)); hipLaunchKernelGGL((gemm_bf16_mfma_kernel<MT, 1>), dim3(grid, S), dim3(64), lds, st, g); }
    HIPCHK(hipGetLastError());
    if (S > 1) JHCHK(launch_splitk_reduce((const float*)g.ws, S, g.m, g.n, g.n0, g.c, g.ldc, g.roffset, g.resid, st));
    return JH_OK;
}
constexpr size_t BF16_SPLITK_WS_BYTES = (size_t)8 * 256 * 16384 * 4;   // S x 256 rows x N floats with S*N <= 8*16384 (enforced by the launcher)

int launch_gemm_bf16_mfma(const MfmaGemmParams& g, hipStream_t st) {
    const int mt = (g.m + 31) / 32;
    switch (mt) {
        case 1: return launch_gemm_bf16_mfma_mt<1>(g, st);
        case 2: return launch_gemm_bf16_mfma_mt<2>(g, st);
        case 3: case 4: return launch_gemm_bf16_mfma_mt<4>(g, st);
        case 5: case 6: return launch_gemm_bf16_mfma_mt<6>(g, st);
        default: return launch_gemm_bf16_mfma_mt<8>(g, st);
    }
}

template <int MT, int CWB>
int launch_gemm_bf16_tile_mc(MfmaBf16TileParams g, int S, hipStream_t st) {
    const int tiles = g.n / 32, grid = (tiles + CWB - 1) / CWB;
    g.nsplit = S;
    // A through LDS (gemm_bf16_lds_kernel) when the K range of a workgroup row divides into double chunks of 4 slices
    const int lds_env = opt_int("JH_BF16_LDS", 1);
    if (lds_env && (tiles % CWB) == 0 && ((g.k / 16 / S) % 8) == 0) {
        const size_t lds = (size_t)2 * MT * 4 * 1024;
        JHCHK(allow_lds((gemm_bf16_lds_kernel<MT, CWB>), lds));
        hipLaunchKernelGGL((gemm_bf16_lds_kernel<MT, CWB>), dim3(grid, S), dim3(CWB * 64), lds, st, g);
    } else {
        hipLaunchKernelGGL((gemm_bf16_tile_kernel<MT, CWB>), dim3(grid, S), dim3(CWB * 64), 0, st, g);
    }
    HIPCHK(hipGetLastError());
    if (S > 1) JHCHK(launch_splitk_reduce((const float*)g.ws, S, g.m, g.n, 0, g.c, g.ldc, 0, g.resid, st));
    return JH_OK;
}
// both operands in MFMA order (gemm_bf16_tile_kernel); n % 32 == 0, k % 16 == 0
int launch_gemm_bf16_tile(const MfmaBf16TileParams& g, hipStream_t st) {
    const int mt = (g.m + 31) / 32, tiles = g.n / 32, nks = g.k / 16;
    // waves per workgroup (column tiles sharing the A fragments through L1) vs workgroups: want >= ~2 workgroups per CU
    // before splitting K, because the split's reduce pass moves S*M*N*8 bytes
    const int cwb_env = opt_int("JH_BF16_CWB", 0), s_env = opt_int("JH_BF16_S", 0);
    int cwb = 8;
    while (cwb > 1 && ((tiles % cwb) != 0 || tiles / cwb < g_cu_count * 2)) cwb >>= 1;
    if (cwb < 4 && tiles % 4 == 0 && nks >= 512) cwb = 4;      // long K, few tiles: measured best (tools/gemm_bench.py)
    if (cwb < 2 && tiles % 2 == 0) cwb = 2;
    if (cwb_env > 0 && tiles % cwb_env == 0) cwb = cwb_env;
    int S = 1;
    const bool lds_kernel = opt_int("JH_BF16_LDS", 1) != 0 && nks % 8 == 0;
    if (lds_kernel && cwb_env <= 0) {
        // gemm_bf16_lds_kernel (tools/bf16_exp.sh sweeps, profiles/r02i_*): the waves of a workgroup share the staged A chunk, so
        // 4 column tiles per workgroup (8 when the A chunk is 7-8 row tiles); K split until the launch has a workgroup per CU and
        // either 1.5 per CU or <= 64 k slices per workgroup
        cwb = (mt >= 7 && tiles % 8 == 0) ? 8 : (tiles % 4 == 0 ? 4 : (tiles % 2 == 0 ? 2 : 1));
        if (g.ws) {
            auto fits = [&](int s2) {
                return nks % (8 * s2) == 0 && (size_t)s2 * g.n <= (size_t)8 * 16384 && (g.n <= 8192 || s2 * g.m <= 512);
            };
            while (S < 16 && fits(2 * S)) {
                const int wgs = (tiles / cwb) * S;
                if (wgs >= g_cu_count && (2 * wgs >= 3 * g_cu_count || nks / S <= 64)) break;
                S *= 2;
            }
        }
    } else if (g.ws) {
        while (S < 16 && (tiles / cwb) * S < g_cu_count * 2 && nks % (4 * S) == 0 && nks / (2 * S) >= 8 &&   // nks/S stays even
               (size_t)(2 * S) * g.n <= (size_t)8 * 16384 && (g.n <= 8192 || 2 * S * g.m <= 640)) S *= 2;
    }
    if (s_env > 0 && g.ws && nks % (2 * s_env) == 0 && (size_t)s_env * g.n <= (size_t)8 * 16384) S = s_env;
    if ((nks / S) % 2) return set_err(JH_ERR_UNSUPPORTED, "tiled BF16 GEMM needs K % 32 == 0");
#define JH_BT(MV) { if (cwb == 8) return launch_gemm_bf16_tile_mc<MV, 8>(g, S, st); if (cwb == 4) return launch_gemm_bf16_tile_mc<MV, 4>(g, S, st); \
                    if (cwb == 2) return launch_gemm_bf16_tile_mc<MV, 2>(g, S, st); return launch_gemm_bf16_tile_mc<MV, 1>(g, S, st); }
    switch (mt) {
        case 1: JH_BT(1)
        case 2: JH_BT(2)
        case 3: JH_BT(3)
        case 4: JH_BT(4)
        case 5: JH_BT(5)
        case 6: JH_BT(6)
        default: JH_BT(8)
    }
#undef JH_BT
}

template <int PRO, int R>
int launch_gemv_f32q4_r(const GemvParams& p, int grid, int threads, hipStream_t st) {
    const size_t lds = lds_bytes_f32(p.K);
    const int nb = nb_for(p.K);
    if (nb == 1) {
        JHCHK(allow_lds(gemv_f32q4_kernel<PRO, R, 1>, lds));
        hipLaunchKernelGGL((gemv_f32q4_kernel<PRO, R, 1>), dim3(grid), dim3(threads), lds, st, p);
    } else if (nb == 2) {
        JHCHK(allow_lds(gemv_f32q4_kernel<PRO, R, 2>, lds));
        hipLaunchKernelGGL((gemv_f32q4_kernel<PRO, R, 2>), dim3(grid), dim3(threads), lds, st, p);
    } else {
        JHCHK(allow_lds(gemv_f32q4_kernel<PRO, R, 0>, lds));
        hipLaunchKernelGGL((gemv_f32q4_kernel<PRO, R, 0>), dim3(grid), dim3(threads), lds, st, p);
    }
    HIPCHK(hipGetLastError());
    return JH_OK;
}
template <int PRO>
int launch_gemv_f32q4(const GemvParams& p, LaunchCfg cfg, int* grid_out, hipStream_t st) {
    int R = cfg.R <= 0 ? 4 : cfg.R;
    if (R > 4) R = 4;
    while (R > 1 && p.nrows % R) R >>= 1;
    const int ngroups = p.nrows / R;
    int grid = (ngroups + cfg.waves - 1) / cfg.waves;
    if (grid > cfg.grid_cap) grid = cfg.grid_cap;
    if (grid < 1) grid = 1;
    if (grid_out) *grid_out = grid;
    if (R == 4) return launch_gemv_f32q4_r<PRO, 4>(p, grid, cfg.waves * 64, st);
    if (R == 2) return launch_gemv_f32q4_r<PRO, 2>(p, grid, cfg.waves * 64, st);
    return launch_gemv_f32q4_r<PRO, 1>(p, grid, cfg.waves * 64, st);
}

}  // namespace

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

namespace {
// Shared Tier-1 GEMM driver.  a_es/b_es: element size in bytes of A / B storage rows (Q4: ldb already in bytes).
int tier1_gemm(int kind, int64_t b_id, int64_t bf_id, const void* a, const float* af, int aoffset, const void* b,
               const float* bf, int boffset, float* r, int roffset, int m, int n0, int n, int k, int lda, int ldaf,
               int ldb, int ldbf, int ldc, uint16_t* cr = nullptr) {
    if (m < 0 || n < 0 || k < 0 || (!r && !cr) || !a) return set_err(JH_ERR_INVALID, "gemm: bad argument");
    const bool q4 = (kind == G_Q8Q4 || kind == G_F32Q4);
    if (q4 && (k % QB)) return set_err(JH_ERR_INVALID, "gemm: K must be a multiple of 32 for Q4/Q8 blocks");
    if (m == 0 || n == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    const size_t a_es = (kind == G_Q8Q4) ? 1 : (kind == G_BF16 ? 2 : 4);
    const size_t b_es = q4 ? 1 : ((kind == G_BF16 || kind == G_F32BF16) ? 2 : 4);
    // ---- A (+ scales)
    const size_t a_elems = (size_t)lda * (m - 1) + aoffset + k;
    void *dA = nullptr, *dAf = nullptr;
    JHCHK(dev_buf(0, a_elems * a_es, &dA));
    HIPCHK(hipMemcpyAsync(dA, a, a_elems * a_es, hipMemcpyHostToDevice, st));
    if (kind == G_Q8Q4) {
        if (!af) return set_err(JH_ERR_INVALID, "gemm_q8_q4: af is null");
        const size_t af_elems = (size_t)ldaf * (m - 1) + aoffset / QB + k / QB;
        JHCHK(dev_buf(1, af_elems * 4, &dAf));
        HIPCHK(hipMemcpyAsync(dAf, af, af_elems * 4, hipMemcpyHostToDevice, st));
    }
    // ---- B (+ scales): registered (whole tensor resident) or copied rows [n0, n0+n)
    const uint8_t* dB = nullptr;
    const float* dBf = nullptr;
    if (b_id >= 0) {
        dB = (const uint8_t*)reg_ptr(b_id);
        if (!dB) return set_err(JH_ERR_INVALID, "gemm: unknown b_id");
    } else {
        if (!b) return set_err(JH_ERR_INVALID, "gemm: b is null and not registered");
        const size_t row0 = (size_t)ldb * n0 * b_es;
        const size_t bytes = ((size_t)ldb * (n - 1) + boffset + (q4 ? k / 2 : k)) * b_es;
        void* t = nullptr;
        JHCHK(dev_buf(2, bytes, &t));
        HIPCHK(hipMemcpyAsync(t, (const uint8_t*)b + row0, bytes, hipMemcpyHostToDevice, st));
        dB = (const uint8_t*)t - row0;  // so that kernel-side ldb*j indexing lands in the copied window
    }
    if (q4) {
        if (bf_id >= 0) {
            dBf = (const float*)reg_ptr(bf_id);
            if (!dBf) return set_err(JH_ERR_INVALID, "gemm: unknown bf_id");
        } else {
            if (!bf) return set_err(JH_ERR_INVALID, "gemm: bf is null and not registered");
            const size_t row0 = (size_t)ldbf * n0;
            const size_t elems = (size_t)ldbf * (n - 1) + (boffset * 2) / QB + k / QB;
            void* t = nullptr;
            JHCHK(dev_buf(3, elems * 4, &t));
            HIPCHK(hipMemcpyAsync(t, bf + row0, elems * 4, hipMemcpyHostToDevice, st));
            dBf = (const float*)t - row0;
        }
    }
    // ---- R
    const long long cmin = (long long)n0 - roffset;
    if (cmin < 0) return set_err(JH_ERR_INVALID, "gemm: n0 - roffset < 0");
    const size_t r_elems = (size_t)ldc * (m - 1) + (size_t)cmin + n;
    void* dR = nullptr;
    JHCHK(dev_buf(4, r_elems * 4, &dR));

    bool fast = (m == 1) && q4 && (aoffset % QB == 0) && (boffset % 16 == 0) && (ldb % 16 == 0) &&
                !opt_int("JH_TIER1_GENERIC", 0);
    if (fast) {
        GemvParams p;
        memset(&p, 0, sizeof(p));
        p.nrows = n;
        p.K = k;
        p.ldb = ldb;
        p.ldbf = ldbf;
        p.w = dB + (size_t)ldb * n0 + boffset;
        p.ws = dBf + (size_t)ldbf * n0 + (boffset * 2) / QB;
        p.out = (float*)dR + cmin;
        LaunchCfg cfg{opt_int("JH_GEMV_R", 0), opt_int("JH_GEMV_WAVES", 0), 0, opt_int("JH_GEMV_PIPE", -1)};
        LaunchCfg cfgf{opt_int("JH_GEMV_R", 0), 8, g_cu_count * 2, 1};
        if (kind == G_Q8Q4) {
            p.aq = (const int8_t*)dA + aoffset;
            p.ad = (const float*)dAf + aoffset / QB;
            JHCHK((launch_gemv_i8q4<PRO_Q8, EPI_STORE>(p, cfg, st)));
        } else {
            p.x = (const float*)dA + aoffset;
            if (aoffset % 4) fast = false;
            else JHCHK((launch_gemv_f32q4<PRO_F32>(p, cfgf, nullptr, st)));
        }
    }
    if (!fast && kind == G_Q8Q4 && m >= 2 && m <= 256 && (n % 32) == 0 && (aoffset % QB) == 0 && (boffset % 16) == 0 && (lda % 16) == 0 &&
        (ldb % 16) == 0 && !opt_int("JH_TIER1_GENERIC", 0)) {
        // batched I8 x Q4 GEMM on the matrix cores (prefill shape), exact integer block sums
        MfmaQ4Params g;
        g.a = (const int8_t*)dA + aoffset; g.af = (const float*)dAf + aoffset / QB;
        g.w = dB + boffset; g.ws = dBf + (boffset * 2) / QB; g.c = (float*)dR; g.resid = nullptr;
        g.m = m; g.n0 = n0; g.n = n; g.k = k; g.lda = lda; g.ldaf = ldaf; g.ldb = ldb; g.ldbf = ldbf; g.ldc = ldc; g.roffset = roffset;
        const int rcm = launch_gemm_q8q4_mfma(g, st);
        if (rcm == JH_OK) fast = true;
        else if (rcm != JH_ERR_UNSUPPORTED) return rcm;
    }
    if (!fast && kind == G_BF16 && m >= 2 && m <= 256 && (k % MG_KS) == 0 && (n % 32) == 0 && (aoffset % 8) == 0 && (boffset % 8) == 0 &&
        (lda % 8) == 0 && (ldb % 8) == 0 && !opt_int("JH_TIER1_GENERIC", 0)) {
        // batched BF16 GEMM on the matrix cores (prefill shape)
        MfmaGemmParams g;
        g.a = (const uint16_t*)dA + aoffset; g.w = (const uint16_t*)dB + boffset; g.c = (float*)dR;
        g.m = m; g.n0 = n0; g.n = n; g.k = k; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.roffset = roffset; g.resid = nullptr;
        g.ws = nullptr; g.nsplit = 1;
        { void* wsp = nullptr; JHCHK(dev_buf(7, BF16_SPLITK_WS_BYTES, &wsp)); g.ws = (float*)wsp; }
        JHCHK(launch_gemm_bf16_mfma(g, st));
        fast = true;
    }
    if (!fast) {
        GemmParams g{dA, (const float*)dAf, dB, dBf, (float*)dR, aoffset, boffset, roffset, m, n0, n, k,
                     lda, ldaf, ldb, ldbf, ldc};
        if (!q4) { g.ldb = ldb; }
        const long long waves = (long long)m * n;
        const int grid = (int)((waves + 3) / 4);
        switch (kind) {
            case G_Q8Q4: hipLaunchKernelGGL((gemm_generic_kernel<G_Q8Q4>), dim3(grid), dim3(256), 0, st, g); break;
            case G_F32Q4: hipLaunchKernelGGL((gemm_generic_kernel<G_F32Q4>), dim3(grid), dim3(256), 0, st, g); break;
            case G_F32: hipLaunchKernelGGL((gemm_generic_kernel<G_F32>), dim3(grid), dim3(256), 0, st, g); break;
            case G_BF16: hipLaunchKernelGGL((gemm_generic_kernel<G_BF16>), dim3(grid), dim3(256), 0, st, g); break;
            default: hipLaunchKernelGGL((gemm_generic_kernel<G_F32BF16>), dim3(grid), dim3(256), 0, st, g); break;
        }
        HIPCHK(hipGetLastError());
    }
    if (cr) {   // BF16 result tensor (vector_simd.c:1060-1064): round on the device, ship 2 bytes per element
        void* dC = nullptr;
        JHCHK(dev_buf(5, r_elems * 2, &dC));
        hipLaunchKernelGGL(store_bf16_2d_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)m), dim3(256), 0, st, (const float*)dR + cmin,
                           (uint16_t*)dC + cmin, n, ldc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy2DAsync(cr + cmin, (size_t)ldc * 2, (uint16_t*)dC + cmin, (size_t)ldc * 2, (size_t)n * 2, m, hipMemcpyDeviceToHost, st));
    } else {
        HIPCHK(hipMemcpy2DAsync(r + cmin, (size_t)ldc * 4, (float*)dR + cmin, (size_t)ldc * 4, (size_t)n * 4, m,
                                hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}

// element-wise Tier-1 helper: a (in/out) and b windows copied, op applied, a copied back
template <int OP>
int tier1_ew(float* a, const float* b, float f, int offset, int length) {
    if (length < 0 || !a) return set_err(JH_ERR_INVALID, "elementwise: bad argument");
    if (length == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dA = nullptr, *dB = nullptr;
    JHCHK(dev_buf(0, (size_t)length * 4, &dA));
    HIPCHK(hipMemcpyAsync(dA, a + offset, (size_t)length * 4, hipMemcpyHostToDevice, st));
    if (b) {
        JHCHK(dev_buf(1, (size_t)length * 4, &dB));
        HIPCHK(hipMemcpyAsync(dB, b + offset, (size_t)length * 4, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL((ew_kernel<OP>), dim3((length + 255) / 256), dim3(256), 0, st, (float*)dA, (const float*)dB, f, length);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(a + offset, dA, (size_t)length * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
}  // namespace

extern "C" {

int jh_gemm_q8_q4(int64_t b_id, int64_t bf_id, const float* af, const int8_t* a, int aoffset, const float* bf,
                  const uint8_t* b, int boffset, float* r, int roffset, int m, int n0, int n, int k, int lda,
                  int ldaf, int ldb, int ldbf, int ldc) {
    return tier1_gemm(G_Q8Q4, b_id, bf_id, a, af, aoffset, b, bf, boffset, r, roffset, m, n0, n, k, lda, ldaf, ldb,
                      ldbf, ldc);
}
int jh_gemm_f32_q4(int64_t b_id, int64_t bf_id, const float* a, int aoffset, const float* bf, const uint8_t* b,
                   int boffset, float* r, int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldbf,
                   int ldc) {
    return tier1_gemm(G_F32Q4, b_id, bf_id, a, nullptr, aoffset, b, bf, boffset, r, roffset, m, n0, n, k, lda, 0,
                      ldb, ldbf, ldc);
}
int jh_gemm_f32(int64_t b_id, const float* a, int aoffset, const float* b, int boffset, float* r, int roffset, int m,
                int n0, int n, int k, int lda, int ldb, int ldc) {
    return tier1_gemm(G_F32, b_id, -1, a, nullptr, aoffset, b, nullptr, boffset, r, roffset, m, n0, n, k, lda, 0, ldb,
                      0, ldc);
}
int jh_gemm_bf16(int64_t b_id, const uint16_t* a, int aoffset, const uint16_t* b, int boffset, uint16_t* cr, float* r,
                 int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc) {
    return tier1_gemm(G_BF16, b_id, -1, a, nullptr, aoffset, b, nullptr, boffset, r, roffset, m, n0, n, k, lda, 0, ldb,
                      0, ldc, cr);
}
int jh_gemm_f32_bf16(int64_t b_id, const float* a, int aoffset, const uint16_t* b, int boffset, uint16_t* cr, float* r,
                     int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc) {
    return tier1_gemm(G_F32BF16, b_id, -1, a, nullptr, aoffset, b, nullptr, boffset, r, roffset, m, n0, n, k, lda, 0,
                      ldb, 0, ldc, cr);
}
int jh_gemm_f32_batch(int batch_num, const int64_t* b_ids, const float* a, int aoffset, const float* const* b, int boffset,
                      float* const* r, int roffset, int m, int n0, int n, int k, int lda, int ldb, int ldc) {
    if (batch_num < 0 || !r) return set_err(JH_ERR_INVALID, "gemm_f32_batch: bad argument");
    for (int i = 0; i < batch_num; i++)
        JHCHK(jh_gemm_f32(b_ids ? b_ids[i] : -1, a, aoffset, b ? b[i] : nullptr, boffset, r[i], roffset, m, n0, n, k, lda, ldb, ldc));
    return JH_OK;
}
int jh_gemm_bf16_batch(int batch_num, const int64_t* b_ids, const uint16_t* a, int aoffset, const uint16_t* const* b,
                       int boffset, uint16_t* const* cr, float* const* r, int roffset, int m, int n0, int n, int k, int lda,
                       int ldb, int ldc) {
    if (batch_num < 0 || (!r && !cr)) return set_err(JH_ERR_INVALID, "gemm_bf16_batch: bad argument");
    for (int i = 0; i < batch_num; i++)   // vector_simd.c:1256-1261
        JHCHK(jh_gemm_bf16(b_ids ? b_ids[i] : -1, a, aoffset, b ? b[i] : nullptr, boffset, cr ? cr[i] : nullptr, r ? r[i] : nullptr,
                           roffset, m, n0, n, k, lda, ldb, ldc));
    return JH_OK;
}
int jh_gemm_f32_bf16_batch(int batch_num, const int64_t* b_ids, const float* a, int aoffset, const uint16_t* const* b,
                           int boffset, uint16_t* const* cr, float* const* r, int roffset, int m, int n0, int n, int k,
                           int lda, int ldb, int ldc) {
    if (batch_num < 0 || (!r && !cr)) return set_err(JH_ERR_INVALID, "gemm_f32_bf16_batch: bad argument");
    for (int i = 0; i < batch_num; i++)   // vector_simd.c:1487-1492
        JHCHK(jh_gemm_f32_bf16(b_ids ? b_ids[i] : -1, a, aoffset, b ? b[i] : nullptr, boffset, cr ? cr[i] : nullptr,
                               r ? r[i] : nullptr, roffset, m, n0, n, k, lda, ldb, ldc));
    return JH_OK;
}
int jh_gemm_q8_q4_batch(int batch_num, const int64_t* b_ids, const int64_t* bf_ids, const float* af, const int8_t* a,
                        int aoffset, const float* const* bf, const uint8_t* const* b, int boffset, float* const* r,
                        int roffset, int m, int n0, int n, int k, int lda, int ldaf, int ldb, int ldbf, int ldc) {
    for (int i = 0; i < batch_num; i++)
        JHCHK(jh_gemm_q8_q4(b_ids ? b_ids[i] : -1, bf_ids ? bf_ids[i] : -1, af, a, aoffset, bf ? bf[i] : nullptr,
                            b ? b[i] : nullptr, boffset, r[i], roffset, m, n0, n, k, lda, ldaf, ldb, ldbf, ldc));
    return JH_OK;
}
int jh_gemm_f32_q4_batch(int batch_num, const int64_t* b_ids, const int64_t* bf_ids, const float* a, int aoffset,
                         const float* const* bf, const uint8_t* const* b, int boffset, float* const* r, int roffset,
                         int m, int n0, int n, int k, int lda, int ldb, int ldbf, int ldc) {
    for (int i = 0; i < batch_num; i++)
        JHCHK(jh_gemm_f32_q4(b_ids ? b_ids[i] : -1, bf_ids ? bf_ids[i] : -1, a, aoffset, bf ? bf[i] : nullptr,
                             b ? b[i] : nullptr, boffset, r[i], roffset, m, n0, n, k, lda, ldb, ldbf, ldc));
    return JH_OK;
}

int jh_accumulate_f32(float* a, const float* b, int offset, int length) {
    if (!b) return set_err(JH_ERR_INVALID, "accumulate: b is null");
    return tier1_ew<EW_ACC>(a, b, 0.f, offset, length);
}
int jh_maccumulate_f32(float* a, const float* b, int offset, int length) {
    if (!b) return set_err(JH_ERR_INVALID, "maccumulate: b is null");
    return tier1_ew<EW_MACC>(a, b, 0.f, offset, length);
}
int jh_scale_f32(float factor, float* a, int offset, int length) { return tier1_ew<EW_SCALE>(a, nullptr, factor, offset, length); }
int jh_silu_mul_f32(float* g, const float* u, int n) {
    if (!u) return set_err(JH_ERR_INVALID, "silu_mul: u is null");
    return tier1_ew<EW_SILU_MUL>(g, u, 0.f, 0, n);
}
int jh_saxpy_f32(float alpha, const float* x, float* y, int xoffset, int yoffset, int limit) {
    if (!x || !y || limit < 0) return set_err(JH_ERR_INVALID, "saxpy: bad argument");
    if (limit == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dY = nullptr, *dX = nullptr;
    JHCHK(dev_buf(0, (size_t)limit * 4, &dY));
    JHCHK(dev_buf(1, (size_t)limit * 4, &dX));
    HIPCHK(hipMemcpyAsync(dY, y + yoffset, (size_t)limit * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dX, x + xoffset, (size_t)limit * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL((ew_kernel<EW_SAXPY>), dim3((limit + 255) / 256), dim3(256), 0, st, (float*)dY, (const float*)dX, alpha, limit);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(y + yoffset, dY, (size_t)limit * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_saxpy_batch_f32(const float* alpha, const float* x, int ldx, float* y, int xoffset, int yoffset, int limit,
                       int aoffset, int xrowoffset, int batch_size) {
    if (!alpha || !x || !y || limit < 0 || batch_size < 0) return set_err(JH_ERR_INVALID, "saxpy_batch: bad argument");
    if (limit == 0 || batch_size == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dY = nullptr, *dX = nullptr, *dAl = nullptr;
    const size_t xelems = (size_t)ldx * (batch_size - 1) + limit;
    JHCHK(dev_buf(0, (size_t)limit * 4, &dY));
    JHCHK(dev_buf(1, xelems * 4, &dX));
    JHCHK(dev_buf(2, (size_t)batch_size * 4, &dAl));
    HIPCHK(hipMemcpyAsync(dY, y + yoffset, (size_t)limit * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dX, x + (size_t)xrowoffset * ldx + xoffset, xelems * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dAl, alpha + aoffset, (size_t)batch_size * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(saxpy_batch_kernel, dim3((limit + 127) / 128), dim3(128), 0, st, (const float*)dAl, (const float*)dX, ldx,
                       (float*)dY, limit, batch_size);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(y + yoffset, dY, (size_t)limit * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_accumulate_f32_q4(float* a, const uint8_t* nib_row, const float* scale_row, int offset, int length) {
    if (!a || !nib_row || !scale_row || length < 0) return set_err(JH_ERR_INVALID, "accumulate_q4: bad argument");
    if (length == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    const int end = offset + length;
    void *dA = nullptr, *dN = nullptr, *dS = nullptr;
    JHCHK(dev_buf(0, (size_t)end * 4, &dA));
    JHCHK(dev_buf(1, (size_t)(end + 31) / 32 * 16, &dN));
    JHCHK(dev_buf(2, (size_t)(end + 31) / 32 * 4, &dS));
    HIPCHK(hipMemcpyAsync((float*)dA + offset, a + offset, (size_t)length * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dN, nib_row, (size_t)(end + 31) / 32 * 16, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dS, scale_row, (size_t)(end + 31) / 32 * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(acc_q4_kernel, dim3((length + 255) / 256), dim3(256), 0, st, (float*)dA, (const uint8_t*)dN, (const float*)dS,
                       offset, length);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(a + offset, (float*)dA + offset, (size_t)length * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_quantize_q8(const float* x, int rows, int ldx, int offset, int length, int8_t* q, int ldq, float* d, int ldd) {
    if (!x || !q || !d || rows < 0 || length < 0 || (length % QB) || (offset % QB))
        return set_err(JH_ERR_INVALID, "quantize_q8: length/offset must be multiples of 32");
    if (rows == 0 || length == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dX = nullptr, *dQ = nullptr, *dD = nullptr;
    const size_t xe = (size_t)ldx * (rows - 1) + offset + length;
    const size_t qe = (size_t)ldq * (rows - 1) + offset + length;
    const size_t de = (size_t)ldd * (rows - 1) + (offset + length) / QB;
    JHCHK(dev_buf(0, xe * 4, &dX));
    JHCHK(dev_buf(1, qe, &dQ));
    JHCHK(dev_buf(2, de * 4, &dD));
    HIPCHK(hipMemcpyAsync(dX, x, xe * 4, hipMemcpyHostToDevice, st));
    const long long threads = (long long)rows * (length / QB) * 32;
    hipLaunchKernelGGL(quantize_q8_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const float*)dX, rows, ldx,
                       offset, length, (int8_t*)dQ, ldq, (float*)dD, ldd);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy2DAsync(q + offset, (size_t)ldq, (int8_t*)dQ + offset, (size_t)ldq, (size_t)length, rows, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpy2DAsync(d + offset / QB, (size_t)ldd * 4, (float*)dD + offset / QB, (size_t)ldd * 4, (size_t)(length / QB) * 4, rows,
                            hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_quantize_bf16(const float* x, int64_t n, uint16_t* out) {
    if (!x || !out || n < 0) return set_err(JH_ERR_INVALID, "quantize_bf16: bad argument");
    if (n == 0) return JH_OK;
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dX = nullptr, *dO = nullptr;
    JHCHK(dev_buf(0, (size_t)n * 4, &dX));
    JHCHK(dev_buf(1, (size_t)n * 2, &dO));
    HIPCHK(hipMemcpyAsync(dX, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(quantize_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)dX, (long long)n, (uint16_t*)dO);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dO, (size_t)n * 2, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_rmsnorm_f32(const float* x, const float* w, float weight_adj, int n, float eps, float* out) {
    if (!x || !w || !out || n <= 0) return set_err(JH_ERR_INVALID, "rmsnorm: bad argument");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dX = nullptr, *dW = nullptr, *dO = nullptr;
    JHCHK(dev_buf(0, (size_t)n * 4, &dX));
    JHCHK(dev_buf(1, (size_t)n * 4, &dW));
    JHCHK(dev_buf(2, (size_t)n * 4, &dO));
    HIPCHK(hipMemcpyAsync(dX, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dW, w, (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(1), dim3(1024), 0, st, (const float*)dX, (const float*)dW, weight_adj, n, eps, (float*)dO);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dO, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
// GPT-2 family pieces (BASELINE config 0): LayerNorm (core/model/LayerNorm.java:41-67) over [offset, offset+length) of each
// of `rows` rows with leading dimension ld; divisor = embeddingLength.  GELU in place.
int jh_layernorm_f32(const float* x, const float* w, const float* b, int rows, int ld, int offset, int length, int divisor,
                     float eps, float* out) {
    if (!x || !w || !b || !out || rows <= 0 || ld <= 0 || offset < 0 || length <= 0 || offset + length > ld || divisor <= 0)
        return set_err(JH_ERR_INVALID, "layernorm: bad argument");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void *dX = nullptr, *dW = nullptr, *dB = nullptr, *dO = nullptr;
    const size_t nb = (size_t)rows * ld * 4;
    JHCHK(dev_buf(0, nb, &dX));
    JHCHK(dev_buf(1, (size_t)ld * 4, &dW));
    JHCHK(dev_buf(2, (size_t)ld * 4, &dB));
    JHCHK(dev_buf(4, nb, &dO));
    HIPCHK(hipMemcpyAsync(dX, x, nb, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dO, out, nb, hipMemcpyHostToDevice, st));   // columns outside the window keep the caller's values
    HIPCHK(hipMemcpyAsync(dW, w, (size_t)ld * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dB, b, (size_t)ld * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(64), 0, st, (const float*)dX, (const float*)dW, (const float*)dB, ld, offset, length,
                       divisor, eps, (float*)dO);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dO, nb, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_gelu_f32(float* x, int n) {
    if (!x || n <= 0) return set_err(JH_ERR_INVALID, "gelu: bad argument");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void* dX = nullptr;
    JHCHK(dev_buf(0, (size_t)n * 4, &dX));
    HIPCHK(hipMemcpyAsync(dX, x, (size_t)n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gelu_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (float*)dX, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(x, dX, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
int jh_softmax_f32(float* x, int offset, int length) {
    if (!x || length <= 0 || offset < 0) return set_err(JH_ERR_INVALID, "softmax: bad argument");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    void* dX = nullptr;
    const size_t n = (size_t)offset + length;
    JHCHK(dev_buf(0, n * 4, &dX));
    HIPCHK(hipMemcpyAsync(dX, x, n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(softmax_kernel, dim3(1), dim3(1024), 0, st, (float*)dX, offset, length);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(x, dX, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}

// VectorMath.precomputeFreqsCis (core/math/VectorMath.java:148-165) -- host side, double cos/sin of the float angle.
int jh_rope_table(int dim, int end, double theta, double scaling, float* out) {
    if (!out || dim <= 0 || (dim & 1) || end <= 0) return set_err(JH_ERR_INVALID, "rope_table: bad argument");
    const int half = dim / 2;
    std::vector<float> freqs((size_t)half);
    float step = 0.0f;
    for (int i = 0; i < half; i++, step = (float)(step + 2.0))
        freqs[(size_t)i] = (float)((1.0 / pow(theta, (double)(step / (float)dim))) / scaling);
    for (int p = 0; p < end; p++) {
        const float t = (float)p;
        for (int i = 0; i < half; i++) {
            const float ang = t * freqs[(size_t)i];
            out[((size_t)p * half + i) * 2 + 0] = (float)cos((double)ang);
            out[((size_t)p * half + i) * 2 + 1] = (float)sin((double)ang);
        }
    }
    return JH_OK;
}
int jh_rope_apply_f32(float* q, float* k, const float* rope, int table_positions, int position, int n_heads, int n_kv_heads,
                      int head_size) {
    if (!q || !k || !rope || position < 0 || n_heads <= 0 || n_kv_heads <= 0 || n_heads % n_kv_heads || (head_size & 1))
        return set_err(JH_ERR_INVALID, "rope_apply: bad argument");
    // kv head h reads table row position + 2*h (CausalSelfAttention.java:260-283): the reference indexes past its table
    // (ArrayIndexOutOfBoundsException) for the last 2*(kvHeads-1) positions
    if ((long long)position + 2LL * (n_kv_heads - 1) >= (long long)table_positions)
        return set_err(JH_ERR_INVALID, "rope_apply: position + 2*(n_kv_heads-1) is beyond the RoPE table");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    const int half = head_size / 2;
    // rows of the table this call touches: [position*half, position*half + n_kv_heads*head_size)
    const size_t r0 = (size_t)position * half, rn = (size_t)n_kv_heads * head_size;
    void *dQ = nullptr, *dK = nullptr, *dR = nullptr;
    JHCHK(dev_buf(0, (size_t)n_heads * head_size * 4, &dQ));
    JHCHK(dev_buf(1, (size_t)n_kv_heads * head_size * 4, &dK));
    JHCHK(dev_buf(2, rn * 8, &dR));
    HIPCHK(hipMemcpyAsync(dQ, q, (size_t)n_heads * head_size * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dK, k, (size_t)n_kv_heads * head_size * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dR, rope + r0 * 2, rn * 8, hipMemcpyHostToDevice, st));
    const int threads = (n_heads + n_kv_heads) * half;
    hipLaunchKernelGGL(rope_kernel, dim3((threads + 255) / 256), dim3(256), 0, st, (float*)dQ, (float*)dK,
                       (const float*)dR - r0 * 2, position, n_heads, n_kv_heads, head_size);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(q, dQ, (size_t)n_heads * head_size * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(k, dK, (size_t)n_kv_heads * head_size * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return JH_OK;
}
// KvBufferCache.computePageSize (core/tensor/KvBufferCache.java:224-280)
int jh_kv_page_geometry(int64_t max_page_bytes, int n_layers, int context_length, int kv_length, int dtype_size,
                        int32_t* out2) {
    if (!out2 || n_layers <= 0 || context_length <= 0 || kv_length <= 0) return set_err(JH_ERR_INVALID, "kv geometry: bad argument");
    const int64_t s = 2LL * dtype_size * kv_length;
    if (max_page_bytes <= s) return set_err(JH_ERR_INVALID, "maxPageSizeInBytes must be greater than the size of a single layer");
    int optL = 1, optC = 1;
    int64_t maxProduct = 0;
    for (int x = n_layers; x >= 1; x--) {
        const int64_t y = max_page_bytes / (x * s);
        if (y >= 1 && y <= context_length) {
            const int64_t product = x * y;
            if (product > maxProduct) { optL = x; optC = (int)y; maxProduct = product; }
            if (product < maxProduct) break;
        }
    }
    out2[0] = optL;
    out2[1] = optC;
    return JH_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ Tier 2
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
};
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

static int attn_variant_for(const jh_session* s, int pos);
static void drop_stale_graphs(jh_session* s);
constexpr int ROPE_MARGIN = 512;
// The RoPE row of kv head h at position p is table row p + 2*h (global head index): the reference's table has
// context_length rows and Java throws ArrayIndexOutOfBoundsException beyond it -- same positions refused here.
static int check_positions(const jh_session* s, int last_pos) {
    const jh_model* m = s->m;
    const long long last_row = (long long)last_pos + 2LL * (m->kv_head_offset + m->c.n_kv_heads - 1);
    if (last_row >= (long long)m->c.context_length)
        return set_err(JH_ERR_INVALID, "position " + std::to_string(last_pos) + ": RoPE row position + 2*(kvHeads-1) is beyond the model's context_length (the reference's table ends there)");
    return JH_OK;
}

namespace {

bool is_global_slot(int which) { return which == JH_W_EMBED || which == JH_W_LMHEAD || which == JH_W_FINALNORM; }

// ---- reference-order launchers (jh_p16.h).  A wave serves 4 weight rows ("row quad"); the plan gives every CU the same
// number of row quads: one 512-thread workgroup per CU, `tw` of its 8 waves own `per` row quads each (the others help
// with the activation prologue only -- a 16-lane row per chain caps the useful waves at rows / 4).
struct P16Plan { int grid, per, tw; };
P16Plan p16_plan(int nrows, int wgs_per_cu, int nwaves = 8) {
    const int nq = (nrows + 3) / 4;
    int grid = g_cu_count * wgs_per_cu;
    if (grid > nq) grid = nq;
    if (grid < 1) grid = 1;
    const int q_wg = (nq + grid - 1) / grid;
    const int per = (q_wg + nwaves - 1) / nwaves;
    const int tw = (q_wg + per - 1) / per;
    return P16Plan{grid, per, tw};
}
// prefetch depth D = groups (16 Q blocks) in flight per lane: the largest of {8, 7, 4, 2, 1} that divides the groups of a row, so
// that a pass ends exactly at the end of a ring block (K = 4096: 8 = the whole row; 14336: 7; 2048: 4)
int p16_depth_for(int K, int want) {
    const int G = (K / QB + 15) / 16;
    static const int ds[] = {8, 7, 4, 2, 1};
    for (int d : ds)
        if (d <= want && G % d == 0) return d;
    return 1;
}
// UM = 8-element units of the activation row per thread, all held in registers (no load loop in the kernel): 2 (K <= 8192) / 4 for
// the RMSNorm prologues, 4 (K <= 16384) / 8 for the plain-quantize ones
template <int PRO, int EPI, int D>
int launch_gemv_i8q4_p16_d(const GemvParams& p, const P16Plan& pl, bool wide, hipStream_t st) {
    const size_t lds = lds_bytes_p16(p.K);
    constexpr int UM_LO = (PRO == PRO_RMS_Q8) ? 2 : 4;
    (void)wide;
    if (p.K <= UM_LO * 4096) {
        JHCHK(allow_lds((gemv_i8q4_p16_kernel<PRO, EPI, D, UM_LO>), lds));
        hipLaunchKernelGGL((gemv_i8q4_p16_kernel<PRO, EPI, D, UM_LO>), dim3(pl.grid), dim3(P16_THREADS), lds, st, p, pl.per, pl.tw);
    } else if (p.K <= 2 * UM_LO * 4096) {
        JHCHK(allow_lds((gemv_i8q4_p16_kernel<PRO, EPI, D, 2 * UM_LO>), lds));
        hipLaunchKernelGGL((gemv_i8q4_p16_kernel<PRO, EPI, D, 2 * UM_LO>), dim3(pl.grid), dim3(P16_THREADS), lds, st, p, pl.per, pl.tw);
    } else {
        return set_err(JH_ERR_UNSUPPORTED, "reference-order GEMV: K = " + std::to_string(p.K) + " exceeds the register-resident activation row");
    }
    HIPCHK(hipGetLastError());
    g_last_gemv_grid = pl.grid;
    return JH_OK;
}
template <int PRO, int EPI>
int launch_gemv_i8q4_p16(const GemvParams& p, int depth, hipStream_t st) {
    // more than 8 row quads per CU (gate|up): two workgroups per CU, so that every SIMD has 3-4 waves to issue from -- the kernel
    // is as much VALU- as HBM-bound, and a wave alone issues one instruction per ~4 cycles
    const P16Plan pl = p16_plan(p.nrows, 1, 8);
    const bool wide = false;
    // ring depth by bytes in flight per CU (tw waves x D KiB): ~32 KiB is what a CU sustains; deeper rings only cost registers
    // (measured: q|k|v and gate|up with 6-7 task waves 4 > 8, the o- and down-projections with 4 task waves 8 / 7 > 4)
    if (pl.tw >= 6 && depth > 4) depth = 4;
    switch (p16_depth_for(p.K, depth)) {
        case 8: return launch_gemv_i8q4_p16_d<PRO, EPI, 8>(p, pl, wide, st);
        case 7: return launch_gemv_i8q4_p16_d<PRO, EPI, 7>(p, pl, wide, st);
        case 4: return launch_gemv_i8q4_p16_d<PRO, EPI, 4>(p, pl, wide, st);
        case 2: return launch_gemv_i8q4_p16_d<PRO, EPI, 2>(p, pl, wide, st);
        default: return launch_gemv_i8q4_p16_d<PRO, EPI, 1>(p, pl, wide, st);
    }
}
template <int PRO, int D>
int launch_gemv_f32q4_p16_d(const GemvParams& p, const P16Plan& pl, hipStream_t st) {
    const size_t lds = lds_bytes_f32_p16(p.K);
    if (p.K <= 8192) {
        JHCHK(allow_lds((gemv_f32q4_p16_kernel<PRO, D, 2>), lds));
        hipLaunchKernelGGL((gemv_f32q4_p16_kernel<PRO, D, 2>), dim3(pl.grid), dim3(P16_THREADS), lds, st, p, pl.per, pl.tw);
    } else if (p.K <= 16384) {
        JHCHK(allow_lds((gemv_f32q4_p16_kernel<PRO, D, 4>), lds));
        hipLaunchKernelGGL((gemv_f32q4_p16_kernel<PRO, D, 4>), dim3(pl.grid), dim3(P16_THREADS), lds, st, p, pl.per, pl.tw);
    } else {
        return set_err(JH_ERR_UNSUPPORTED, "reference-order LM head: K exceeds the register-resident activation row");
    }
    HIPCHK(hipGetLastError());
    return JH_OK;
}
template <int PRO>
int launch_gemv_f32q4_p16(const GemvParams& p, int* grid_out, hipStream_t st) {
    const P16Plan pl = p16_plan(p.nrows, 2);   // two workgroups per CU; the argmax partial buffers hold 4096 entries
    if (grid_out) *grid_out = pl.grid;
    switch (p16_depth_for(p.K, 8)) {
        case 8: return launch_gemv_f32q4_p16_d<PRO, 8>(p, pl, st);
        case 7: return launch_gemv_f32q4_p16_d<PRO, 7>(p, pl, st);
        case 4: return launch_gemv_f32q4_p16_d<PRO, 4>(p, pl, st);
        case 2: return launch_gemv_f32q4_p16_d<PRO, 2>(p, pl, st);
        default: return launch_gemv_f32q4_p16_d<PRO, 1>(p, pl, st);
    }
}

// ---- reference-order GEMV on the integer MFMA (jh_t16.h): one wave per 16-row tile, one 512-thread workgroup per CU
// K % 256 == 0 (whole q steps of 4 blocks, an even number of them), K <= 8192 (register-resident activation row at 512 threads)
bool t16_shape_ok(int K) { return K % 256 == 0 && K <= 8192 && lds_bytes_t16(K) <= 150 * 1024; }
template <int PRO, int EPI>
int launch_gemv_t16(const GemvParams& p, hipStream_t st) {
    constexpr int NT = 512;
    const int ntiles = (EPI == EPI_SILU_MUL) ? p.nrows / 8 : p.nrows / 16;
    const int nq = p.K / QB / 4;
    int cus = g_cu_count < ntiles ? g_cu_count : ntiles;
    if (cus < 1) cus = 1;
    const int t_cu = (ntiles + cus - 1) / cus;
    const int tpw = (t_cu + 7) / 8;
    const int aw = (t_cu + tpw - 1) / tpw;
    const int grid = (ntiles + aw * tpw - 1) / (aw * tpw);
    const size_t lds = lds_bytes_t16(p.K);
#define JH_T16_LAUNCH(DV, UMV)                                                                                     \
    do {                                                                                                           \
        JHCHK(allow_lds((gemv_t16_kernel<PRO, EPI, DV, UMV, NT>), lds));                                           \
        hipLaunchKernelGGL((gemv_t16_kernel<PRO, EPI, DV, UMV, NT>), dim3(grid), dim3(NT), lds, st, p, tpw, aw);   \
    } while (0)
    if (nq % 4 == 0) {
        if (p.K <= 4096) JH_T16_LAUNCH(4, 1); else JH_T16_LAUNCH(4, 2);
    } else {
        if (p.K <= 4096) JH_T16_LAUNCH(2, 1); else JH_T16_LAUNCH(2, 2);
    }
#undef JH_T16_LAUNCH
    HIPCHK(hipGetLastError());
    g_last_gemv_grid = grid;
    return JH_OK;
}
int tiled_mode_for(jh_model* m);
thread_local int g_operand_packs = 0;   // operand copies this thread has queued a pack kernel for (ensure_strict_operands waits for them)
// the order-free sessions use the MFMA gate|up GEMV as well (option JH_FAST_GATEUP_T16=0: their own VALU kernel, for comparisons)
bool fast_gateup_t16(jh_model* m) { return opt_int("JH_FAST_GATEUP_T16", 1) != 0 && tiled_mode_for(m) == TILED_RESIDENT; }   // (a second copy: not under JH_TILED_COPY=transient)
// gate|up of layer li in T16 order (tile u = gate rows 8u..8u+7, up rows 8u..8u+7): made once, before any graph capture
bool t16_gateup_ok(const jh_model* m, int li) {
    const int enabled = opt_int("JH_T16", 1);
    const JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    const JWeight &G = W[JH_W_GATE], &U = W[JH_W_UP];
    return enabled && G.data && U.data && G.dtype == JH_DT_Q4 && U.dtype == JH_DT_Q4 && G.rows == U.rows && G.cols == U.cols &&
           G.rows % 8 == 0 && t16_shape_ok(G.cols);
}
int ensure_gateup_t16(jh_model* m, int li, hipStream_t st) {
    JWeight& F = m->gateup[(size_t)li];
    if (F.t16 || !t16_gateup_ok(m, li)) return JH_OK;
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
bool t16_weight_ok(const JWeight& W) { return W.data && W.dtype == JH_DT_Q4 && W.rows % 16 == 0 && W.cols % 512 == 0; }
int ensure_t16(JWeight& W, hipStream_t st) {
    if (W.t16 || !t16_weight_ok(W)) return JH_OK;
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
        if (hipMalloc((void**)&W.p16t, (size_t)W.rows * rb + 64) != hipSuccess) { W.p16t = nullptr; return set_err(JH_ERR_OOM, "hipMalloc BF16T weight copy"); }
        g_operand_packs++;
        const long long threads = (long long)W.rows * (long long)(rb / 16);
        hipLaunchKernelGGL(bf16t_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const uint16_t*)W.data, W.rows, W.cols, W.cols, W.p16t);
        HIPCHK(hipGetLastError());
        return JH_OK;
    }
    if (W.dtype != JH_DT_Q4) return JH_OK;
    const int nblk = W.cols / QB;
    const size_t rb = p16t_row_bytes(W.cols);
    if (hipMalloc((void**)&W.p16t, (size_t)W.rows * rb + 64) != hipSuccess) { W.p16t = nullptr; return set_err(JH_ERR_OOM, "hipMalloc P16T weight copy"); }
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
// ---- reference-order GEMV of a dense BF16 weight (jh_bf16r.h): same work split as the p16 kernels
template <int PRO, int EPI, bool ARGMAX>
int launch_gemv_bf16r(const GemvParams& p, int* grid_out, hipStream_t st) {
    if (p.K % 32 || p.K > 32768) return set_err(JH_ERR_UNSUPPORTED, "reference-order BF16 GEMV: K must be a multiple of 32, at most 32768");
    const P16Plan pl = p16_plan(p.nrows, ARGMAX ? 2 : 1);   // LM head: two workgroups per CU (argmax partial buffers hold 4096 entries)
    if (grid_out) *grid_out = pl.grid;
    const int G = (p.K + BF16R_GROUP - 1) / BF16R_GROUP;
    const size_t lds = lds_bytes_bfr(p.K);
#define JH_BFR(DV, UMV)                                                                                                          \
    do {                                                                                                                         \
        JHCHK(allow_lds((gemv_bf16r_kernel<PRO, EPI, ARGMAX, DV, UMV>), lds));                                                   \
        hipLaunchKernelGGL((gemv_bf16r_kernel<PRO, EPI, ARGMAX, DV, UMV>), dim3(pl.grid), dim3(P16_THREADS), lds, st, p, pl.per, pl.tw); \
    } while (0)
#define JH_BFR_D(UMV)                                                                                                            \
    do {                                                                                                                         \
        if (G % 8 == 0) JH_BFR(8, UMV); else if (G % 4 == 0) JH_BFR(4, UMV); else if (G % 2 == 0) JH_BFR(2, UMV); else JH_BFR(1, UMV); \
    } while (0)
    if (p.K <= 8192) JH_BFR_D(2); else if (p.K <= 16384) JH_BFR_D(4); else JH_BFR_D(8);
#undef JH_BFR_D
#undef JH_BFR
    HIPCHK(hipGetLastError());
    g_last_gemv_grid = pl.grid;
    return JH_OK;
}
bool prefill_t16_ok(jh_session* s);
bool prefill_bf16r_ok(jh_session* s);
// every operand copy a reference-order session of this shard will touch (allocation must not happen inside a graph capture)
int ensure_strict_operands_locked(jh_session* s, hipStream_t st);
// The copies belong to the MODEL: every session of it (other streams, other host threads) reads them.  They are created under the
// model's lock, and the lock is released only once the pack kernels this call queued have FINISHED -- a second session then either
// waits here or finds complete copies; nobody launches a reference-order GEMV against a half-packed operand.
int ensure_strict_operands(jh_session* s, hipStream_t st) {
    std::lock_guard<std::mutex> lk(s->m->op_mu);
    const int before = g_operand_packs;
    const int rc = ensure_strict_operands_locked(s, st);
    if (g_operand_packs != before) HIPCHK(hipStreamSynchronize(st));
    return rc;
}
int ensure_strict_operands_locked(jh_session* s, hipStream_t st) {
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

int attn_launch(jh_session* s, int rel, hipStream_t st, bool tap, long long* dbg = nullptr) {
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
        const int ru = p16_av_rows(s->max_ctx);
#define JH_P16_AV(HSV, RV)                                                                                                      \
    if (hs == HSV && ru == RV) {                                                                                               \
        JHCHK(allow_lds((attn_p16_av_kernel<HSV, RV>), lds_av));                                                               \
        hipLaunchKernelGGL((attn_p16_av_kernel<HSV, RV>), grid_v, dim3(P16_ATT_THREADS), lds_av, st, p, (const float*)s->p16_scores, s->p16_sc_stride); \
    }
#define JH_P16_ATTN(HSV, GV)                                                                                                   \
    if (hs == HSV && group == GV) {                                                                                            \
        hipLaunchKernelGGL((attn_p16_scores_kernel<HSV, GV>), grid_s, dim3(P16_ATT_THREADS), 0, st, p, s->p16_scores, s->p16_sc_stride); \
        HIPCHK(hipGetLastError());                                                                                             \
        JH_P16_AV(HSV, 2) JH_P16_AV(HSV, 4) JH_P16_AV(HSV, 8) JH_P16_AV(HSV, 16)                                                \
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
        if (!F.data || !W[JH_W_Q].data || !W[JH_W_K].data || !W[JH_W_V].data || !W[JH_W_O].data || !W[JH_W_NORM1].data)
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
    if (!W[JH_W_GATE].data || !W[JH_W_UP].data || !W[JH_W_DOWN].data || !W[JH_W_NORM2].data)
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

// ---- batched prefill -----------------------------------------------------------------------------------------------
constexpr int PB_MAX_ROWS = 256;   // rows per chunk = the MFMA GEMM's M limit (8 tiles of 32)
constexpr int PF_MAX_SPLIT = 8;    // key-range splits of the MFMA prefill attention

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
static size_t tiled_w_bytes(const JWeight& W) { return W.dtype == JH_DT_BF16 ? (size_t)W.rows * W.cols * 2 : (size_t)W.rows * (W.cols / QB) * 16; }
static size_t tiled_s_bytes(const JWeight& W) { return W.dtype == JH_DT_BF16 ? 0 : (size_t)W.rows * (W.cols / QB) * 4; }
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
    if (!m->qkv[(size_t)li].data || !W[JH_W_Q].data || !W[JH_W_K].data || !W[JH_W_V].data || !W[JH_W_O].data || !W[JH_W_GATE].data || !W[JH_W_UP].data ||
        !W[JH_W_DOWN].data || !W[JH_W_NORM1].data || !W[JH_W_NORM2].data)
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
    JHCHK(allow_lds((gemm_t16_kernel<EPI, MT, CW>), lds));
    const int grid = ((nslices + 7) / 8) * 8 * nrt;
    hipLaunchKernelGGL((gemm_t16_kernel<EPI, MT, CW>), dim3(grid), dim3(CW * 64), lds, st, g);
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
int prefill_attn_half_p16(jh_session* s, int li, int rows, int start_pos, float* out, const float* resid, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const int E = c.embedding_length, hs = c.head_size, A = c.n_heads * hs, KV = c.n_kv_heads * hs;
    JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    JWeight& F = m->qkv[(size_t)li];
    JHCHK(prefill_weights_set(s, li));
    JHCHK((rows_act_t16_launch<PRO_RMS_Q8>(s, s->pb_x, E, (const float*)W[JH_W_NORM1].data, c.rms_eps, E, rows, st)));
    JHCHK((gemm_t16_launch<EPI_STORE>(s, F.t16, F.t16_scales, (A + 2 * KV) / 16, E, rows, s->pb_qkv, A + 2 * KV, nullptr, 0, st)));
    JHCHK(prefill_attn_p16_launch(s, li - c.layer_start, rows, start_pos, st));
    JHCHK((rows_act_t16_launch<PRO_QUANT_Q8>(s, s->pb_att, A, nullptr, 0.f, A, rows, st)));
    if (resid) return gemm_t16_launch<EPI_RESID>(s, W[JH_W_O].t16, W[JH_W_O].t16_scales, E / 16, A, rows, out, E, resid, E, st);
    return gemm_t16_launch<EPI_STORE>(s, W[JH_W_O].t16, W[JH_W_O].t16_scales, E / 16, A, rows, out, E, nullptr, 0, st);
}
int prefill_ffn_half_p16(jh_session* s, int li, int rows, const float* x1, float* out, const float* resid, hipStream_t st) {
    jh_model* m = s->m;
    const jh_config& c = m->c;
    const int E = c.embedding_length, H = c.hidden_length;
    JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
    const JWeight& GU = m->gateup[(size_t)li];
    JHCHK((rows_act_t16_launch<PRO_RMS_Q8>(s, x1, E, (const float*)W[JH_W_NORM2].data, c.rms_eps, E, rows, st)));
    JHCHK((gemm_t16_launch<EPI_SILU_MUL>(s, GU.t16, GU.t16_scales, H / 8, E, rows, s->pb_g, H, nullptr, 0, st)));
    JHCHK((rows_act_t16_launch<PRO_QUANT_Q8>(s, s->pb_g, H, nullptr, 0.f, H, rows, st)));
    if (resid) return gemm_t16_launch<EPI_RESID>(s, W[JH_W_DOWN].t16, W[JH_W_DOWN].t16_scales, E / 16, H, rows, out, E, resid, E, st);
    return gemm_t16_launch<EPI_STORE>(s, W[JH_W_DOWN].t16, W[JH_W_DOWN].t16_scales, E / 16, H, rows, out, E, nullptr, 0, st);
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

int finish_launch(jh_session* s, hipStream_t st, int do_embed, float temperature = 0.0f) {
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
        if (s->exec_s[v]) { hipGraphExecDestroy(s->exec_s[v]); s->exec_s[v] = nullptr; hipGraphDestroy(s->graph_s[v]); s->graph_s[v] = nullptr; }
    }
    return JH_OK;
}

}  // namespace

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
    s->direct_max = 0;   // ("direct" mode -- the o-projection's prologue combining the attention slices -- measured slower, DESIGN.md 3: removed)
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
    if (!s->strict && prefill_batch_ok(s)) {
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
// Debug: phase timeline of ONE attention launch at position `pos` (layer 0 of the shard): out[split*16 + k] =
// wall_clock64 ticks (100 MHz) at phase k, -1 where not reached.  See JH_ATT_STAMP in jh_kernels.h.
int jh_debug_attn_timeline(jh_session* s, int pos, long long* out, int n) {
    if (!s || !out || n < 16 * 16) return set_err(JH_ERR_INVALID, "attn_timeline: need 256 slots");
    HIPCHK(hipSetDevice(s->m->device));
    long long* d = nullptr;
    HIPCHK(hipMalloc(&d, 256 * 8));
    HIPCHK(hipMemset(d, 0xff, 256 * 8));
    hipStream_t st = s->stream;
    for (int it = 0; it < 3; it++) {   // warm: the last launch's stamps are the ones reported
        hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, s->st, pos, 0, 0);
        s->attn_variant = attn_variant_for(s, pos);
        JHCHK(attn_launch(s, 0, st, false, d));
    }
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(out, d, 256 * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(d));
    return JH_OK;
}
// Device-resident timing of the batched (prefill) MFMA GEMMs: kind 0 = I8xQ4, 1 = BF16xBF16.  `copies` distinct weight
// matrices are cycled so the stream comes from HBM, not the Infinity Cache.  out_ms = average per GEMM.
int jh_gemm_bench(int kind, int m, int n, int k, int copies, int iters, double* out_ms) {
    if (!out_ms || m < 2 || m > 256 || (n % 32) || (k % 64) || copies < 1 || iters < 1) return set_err(JH_ERR_INVALID, "gemm_bench: bad shape");
    JHCHK(ensure_ctx());
    hipStream_t st = tctx.stream;
    const bool q4 = (kind == 0 || kind == 2);   // kind 2 = I8xQ4 with both operands in MFMA-tiled order (prefill path)
    const size_t wbytes = q4 ? (size_t)n * k / 2 : (size_t)n * k * 2;
    const size_t sbytes = q4 ? (size_t)n * (k / QB) * 4 : 0;
    uint8_t *w = nullptr, *a = nullptr; float *ws = nullptr, *af = nullptr, *c = nullptr;
    HIPCHK(hipMalloc(&w, wbytes * copies)); HIPCHK(hipMemset(w, 0x37, wbytes * copies));
    if (sbytes) { HIPCHK(hipMalloc(&ws, sbytes * copies)); HIPCHK(hipMemset(ws, 0, sbytes * copies)); }
    HIPCHK(hipMalloc(&a, (size_t)(m + 32) * k * 2)); HIPCHK(hipMemset(a, 1, (size_t)(m + 32) * k * 2));
    HIPCHK(hipMalloc(&af, (size_t)(m + 32) * (k / QB) * 4)); HIPCHK(hipMemset(af, 0, (size_t)(m + 32) * (k / QB) * 4));
    HIPCHK(hipMalloc(&c, (size_t)m * n * 4));
    float* bf16_ws = nullptr;
    HIPCHK(hipMalloc(&bf16_ws, BF16_SPLITK_WS_BYTES));   // split-K workspace; kind 3 = BF16 with MFMA-ordered operands
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    int rc = JH_OK;
    for (int it = -1; it < iters && rc == JH_OK; it++) {
        if (it == 0) HIPCHK(hipEventRecord(e0, st));
        for (int l = 0; l < copies && rc == JH_OK; l++) {
            if (q4) {
                MfmaQ4Params g{(const int8_t*)a, af, w + l * wbytes, ws + l * (sbytes / 4), c, nullptr, m, 0, n, k, k, k / QB, k / 2, k / QB, n, 0};
                rc = launch_gemm_q8q4_mfma(g, st, kind == 2, bf16_ws, BF16_SPLITK_WS_BYTES);
            } else if (kind == 3) {
                MfmaBf16TileParams g{(const uint16_t*)a, (const uint16_t*)(w + l * wbytes), c, nullptr, m, n, k, n, bf16_ws, 1};
                rc = launch_gemm_bf16_tile(g, st);
            } else {
                MfmaGemmParams g{(const uint16_t*)a, (const uint16_t*)(w + l * wbytes), c, m, 0, n, k, k, k, n, 0, nullptr, bf16_ws, 1};
                rc = launch_gemm_bf16_mfma(g, st);
            }
        }
    }
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *out_ms = (double)ms / ((double)iters * copies);
    hipFree(w); if (ws) hipFree(ws); if (bf16_ws) hipFree(bf16_ws); hipFree(a); hipFree(af); hipFree(c); hipEventDestroy(e0); hipEventDestroy(e1);
    return rc;
}
int jh_session_synchronize(jh_session* s) {
    if (!s) return set_err(JH_ERR_INVALID, "session_synchronize: null");
    HIPCHK(hipSetDevice(s->m->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    return JH_OK;
}

// Roofline probe: launch ONE kernel kind of the decode step back-to-back over all of this shard's layers (so the
// weights stream from HBM, not from the 256 MiB Infinity Cache), `iters` sweeps, bracketed by hipEvents on the
// session's stream.  out_ms = average duration of one launch.
int jh_kernel_bench(jh_session* s, int which, int iters, double* out_ms, int64_t* out_bytes_per_launch) {
    if (!s || !out_ms || iters <= 0) return set_err(JH_ERR_INVALID, "kernel_bench: bad argument");
    if (s->m->c.weight_dtype != JH_DT_Q4) return set_err(JH_ERR_UNSUPPORTED, "kernel_bench: JQ4 models only");
    jh_model* m = s->m;
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = s->stream;
    const jh_config& c = m->c;
    const int E = c.embedding_length, H = c.hidden_length, hs = c.head_size;
    const int A = c.n_heads * hs, KV = c.n_kv_heads * hs;
    const int nl = c.layer_end - c.layer_start;
    JHCHK(ensure_strict_operands(s, st));
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, st, s->st, s->max_ctx / 2, 0, 0);
    s->attn_variant = attn_variant_for(s, s->max_ctx / 2);
    const bool p16 = s->strict != 0;   // reference-order kernels (jh_t16.h / jh_p16.h) when the session is in that mode
    int launches = 0;
    for (int it = -1; it < iters; it++) {
        if (it == 0) HIPCHK(hipEventRecord(s->ev0, st));
        for (int li = c.layer_start; li < c.layer_end; li++) {
            const JWeight* W = &m->layer_w[(size_t)li * JH_W_COUNT];
            GemvParams p;
            memset(&p, 0, sizeof(p));
            if (which == 9) {          // LM head (+ final norm, argmax partials): one weight, re-streamed per launch
                JHCHK(lmhead_launch(s, st));
            } else if (which == 0) {
                const JWeight& F = m->qkv[(size_t)li];
                p.w = (const uint8_t*)F.data; p.ws = F.scales; p.nrows = A + 2 * KV; p.out = s->qkv;
                p.K = E; p.ldb = E / 2; p.ldbf = E / QB;
                p.x = s->x; p.nw = (const float*)W[JH_W_NORM1].data; p.eps = c.rms_eps;
                if (p16) { JHCHK(use_p16t(p, F)); JHCHK((launch_gemv_i8q4_p16<PRO_RMS_Q8, EPI_STORE>(p, s->p16_depth, st))); }
                else JHCHK((launch_gemv_i8q4<PRO_RMS_Q8, EPI_STORE>(p, s->cfg_qkv, st)));
            } else if (which == 1) {
                JHCHK(attn_launch(s, li - c.layer_start, st, false));
            } else if (which == 2) {
                p.w = (const uint8_t*)W[JH_W_O].data; p.ws = W[JH_W_O].scales; p.nrows = E; p.out = s->x1;
                p.K = A; p.ldb = A / 2; p.ldbf = A / QB; p.x = s->attf; p.resid = s->x;
                if (p16) {
                    JHCHK(use_p16t(p, W[JH_W_O]));
                    JHCHK((launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_RESID>(p, s->p16_depth, st)));
                } else {
                    JHCHK((launch_gemv_i8q4<PRO_QUANT_Q8, EPI_RESID>(p, s->cfg_o, st)));
                }
            } else if (which == 3) {
                p.w = (const uint8_t*)W[JH_W_GATE].data; p.ws = W[JH_W_GATE].scales; p.nrows = H;
                p.w2 = (const uint8_t*)W[JH_W_UP].data; p.ws2 = W[JH_W_UP].scales;
                p.K = E; p.ldb = E / 2; p.ldbf = E / QB;
                p.x = s->x1; p.nw = (const float*)W[JH_W_NORM2].data; p.eps = c.rms_eps;
                p.out = s->hf;
                if (t16_gateup_ok(m, li) && (p16 || fast_gateup_t16(m))) {
                    JHCHK(ensure_gateup_t16(m, li, st));
                    p.w = m->gateup[(size_t)li].t16; p.ws = m->gateup[(size_t)li].t16_scales; p.w2 = nullptr; p.ws2 = nullptr;
                    JHCHK((launch_gemv_t16<PRO_RMS_Q8, EPI_SILU_MUL>(p, st)));
                }
                else if (p16) { JHCHK(use_p16t(p, W[JH_W_GATE])); p.w2 = W[JH_W_UP].p16t; JHCHK((launch_gemv_i8q4_p16<PRO_RMS_Q8, EPI_SILU_MUL>(p, s->p16_depth, st))); }
                else JHCHK((launch_gemv_i8q4<PRO_RMS_Q8, EPI_SILU_MUL>(p, s->cfg_gateup, st)));
            } else if (which == 4) {
                p.w = (const uint8_t*)W[JH_W_DOWN].data; p.ws = W[JH_W_DOWN].scales; p.nrows = E; p.out = s->x1;
                p.K = H; p.ldb = H / 2; p.ldbf = H / QB; p.x = s->hf; p.resid = s->x;
                if (p16) { JHCHK(use_p16t(p, W[JH_W_DOWN])); JHCHK((launch_gemv_i8q4_p16<PRO_QUANT_Q8, EPI_RESID>(p, s->p16_depth, st))); }
                else JHCHK((launch_gemv_i8q4<PRO_QUANT_Q8, EPI_RESID>(p, s->cfg_down, st)));
            } else if (which >= 5 && which <= 8) {
                // the same GEMVs fed a pre-quantized activation row (PRO_Q8): what the fused prologue costs
                JHCHK(prefill_alloc(s));
                p.aq = s->pb_aq; p.ad = s->pb_ad;
                if (which == 5) {
                    const JWeight& F = m->qkv[(size_t)li];
                    p.w = (const uint8_t*)F.data; p.ws = F.scales; p.nrows = A + 2 * KV; p.out = s->qkv;
                    p.K = E; p.ldb = E / 2; p.ldbf = E / QB;
                    JHCHK((launch_gemv_i8q4<PRO_Q8, EPI_STORE>(p, s->cfg_qkv, st)));
                } else if (which == 6) {
                    p.w = (const uint8_t*)W[JH_W_O].data; p.ws = W[JH_W_O].scales; p.nrows = E; p.out = s->x1;
                    p.K = A; p.ldb = A / 2; p.ldbf = A / QB; p.resid = s->x;
                    JHCHK((launch_gemv_i8q4<PRO_Q8, EPI_RESID>(p, s->cfg_o, st)));
                } else if (which == 7) {
                    p.w = (const uint8_t*)W[JH_W_GATE].data; p.ws = W[JH_W_GATE].scales; p.nrows = H;
                    p.w2 = (const uint8_t*)W[JH_W_UP].data; p.ws2 = W[JH_W_UP].scales;
                    p.K = E; p.ldb = E / 2; p.ldbf = E / QB; p.out = s->hf;
                    JHCHK((launch_gemv_i8q4<PRO_Q8, EPI_SILU_MUL>(p, s->cfg_gateup, st)));
                } else {
                    p.w = (const uint8_t*)W[JH_W_DOWN].data; p.ws = W[JH_W_DOWN].scales; p.nrows = E; p.out = s->x1;
                    p.K = H; p.ldb = H / 2; p.ldbf = H / QB; p.resid = s->x;
                    JHCHK((launch_gemv_i8q4<PRO_Q8, EPI_RESID>(p, s->cfg_down, st)));
                }
            } else {
                return set_err(JH_ERR_INVALID, "kernel_bench: which in 0..9 (qkv, attn, oproj, gateup, down; 5..8 = the GEMVs with pre-quantized input; 9 = LM head)");
            }
            if (it >= 0) launches++;
        }
    }
    HIPCHK(hipEventRecord(s->ev1, st));
    HIPCHK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    *out_ms = (double)ms / launches;
    if (out_bytes_per_launch) {
        const double bpw = 0.625;  // 0.5 B nibble + 4 B scale / 32 weights
        int64_t b = 0;
        if (which == 0) b = (int64_t)((double)(A + 2 * KV) * E * bpw);
        else if (which == 1) b = (int64_t)2 * (s->max_ctx / 2 + 1) * KV * 4 + (int64_t)2 * KV * 4;
        else if (which == 2) b = (int64_t)((double)E * A * bpw);
        else if (which == 3 || which == 7) b = (int64_t)((double)2 * H * E * bpw);
        else if (which == 9) b = (int64_t)((double)c.vocab_size * E * bpw);
        else if (which == 5) b = (int64_t)((double)(A + 2 * KV) * E * bpw);
        else if (which == 6) b = (int64_t)((double)E * A * bpw);
        else b = (int64_t)((double)E * H * bpw);
        *out_bytes_per_launch = b;
    }
    (void)nl;
    return JH_OK;
}

// captured graphs hold raw device pointers of the weights: drop them all if a weight was replaced since the capture
static void drop_stale_graphs(jh_session* s) {
    if (s->graphs_version == s->m->weights_version) return;
    for (int v = 0; v < N_ATTN_VARIANTS; v++) {
        if (s->exec_s[v]) { hipGraphExecDestroy(s->exec_s[v]); s->exec_s[v] = nullptr; }
        if (s->graph_s[v]) { hipGraphDestroy(s->graph_s[v]); s->graph_s[v] = nullptr; }
        if (s->exec[v]) { hipGraphExecDestroy(s->exec[v]); s->exec[v] = nullptr; }
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
static int attn_variant_for(const jh_session* s, int pos) {
    if (s->direct_max == 0 && pos + 1 <= s->max_splits * 32) return 1;
    if (s->long_splits > 0 && pos + 1 > s->long_min) return 2;
    return 0;
}
// does any position of [first, last] use attention variant v?  (variants change at most twice along the context)
static bool attn_variant_in_range(const jh_session* s, int v, int first, int last) {
    const int edges[4] = {first, last, s->max_splits * 32, s->long_min};   // positions next to the two thresholds
    for (int e : edges)
        for (int d = -1; d <= 1; d++) {
            const int pos = e + d;
            if (pos >= first && pos <= last && attn_variant_for(s, pos) == v) return true;
        }
    return false;
}
static int build_row_graph(jh_session* s, int v) {
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
static int forward_impl(jh_session* s, const int32_t* tokens, const float* x_in, bool x_in_dev, int n, int start_pos,
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

static int build_graph(jh_session* s, int v, float temperature = 0.0f) {
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
    const bool p16_two_launch_attn = s->strict != 0;
    const int per_layer = 5 + (p16_two_launch_attn ? 1 : 0);
    s->kernels_per_token = (c.layer_end - c.layer_start) * per_layer + (has_out ? 2 : 0);
    return JH_OK;
}

static int decode_n_async_impl(jh_session* s, int32_t first_token, int start_pos, int n, float temperature, const float* u);
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
static int decode_n_async_impl(jh_session* s, int32_t first_token, int start_pos, int n, float temperature, const float* u) {
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
    for (int i = 0; i < n; i++) {
        const int v = attn_variant_for(s, start_pos + i);   // the host knows every token's position in advance
        if (use_graph) {
            JHCHK(build_graph(s, v, temperature));
            HIPCHK(hipGraphLaunch(sampled ? s->exec_s[v] : s->exec[v], st));
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
