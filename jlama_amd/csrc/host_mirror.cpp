// host_mirror.cpp -- libjlamahost.so: the reference's HOST, as it is, above the Tier-1 C ABI.
//
// north_star: "the Java host (jlama-core AbstractModel.generate() / TransformerBlock forward) stays as-is and calls through ... the
// TensorOperations provider".  There is no JDK on either box, so this file restates that host in C++ (the reference is compiled
// code): the SAME op sequence, buffers and call arguments as
//   AbstractModel.forward / batchForward / sample / generate   core/model/AbstractModel.java:267-329,443-491,515-646
//   TransformerBlock.forward                                   core/model/TransformerBlock.java:158-215
//   CausalSelfAttention.forward (GQA branch)                   core/model/CausalSelfAttention.java:145-385
//   MLPBlock.forward                                           core/model/MLPBlock.java:105-166
//   LlamaModel (embedding, maybeQuantize, LM head)             core/model/llama/LlamaModel.java:67-184
//   KvBufferCache pages                                        core/tensor/KvBufferCache.java:224-352
// and where Java calls a TensorOperations method this file calls the C entry point the provider binds (include/jlama_hip.h,
// java/.../HipTensorOperations.java): batchDotProduct / dotProductChunk / dotProductBatchChunk -> jh_gemm_*[_batch],
// quantize -> jh_quantize_q8 / jh_quantize_bf16, accumulate / maccumulate / scale / saxpy -> jh_*_f32.  What is plain Java in the
// reference (RMSNorm.forward, the RoPE rotation, VectorMath.softMax, ActivationFunction.eval, the KV copy, argmax) is plain C++ here,
// on the host, exactly where Java has it.  `elementwise_on_device` = 0 reproduces the provider's default (`jlama.hip_elementwise`
// unset: element-wise methods go to the Panama delegate on the host, HipTensorOperations.java:45-57,315-407) -- only the GEMMs cross
// the boundary; 1 sends every provider method to the device.
//
// This library is NOT the backend: it is the caller.  It owns host memory only, holds no device pointer, and links nothing but the
// exported C ABI of libjlamahip.so.  tests/test_gpu_host_mirror.py checks its ids / logits bit for bit against the oracle (with
// JH_STRICT_ORDER=1) and bench.py times it (`tier1_host_tokens_per_s`): the measured cost of keeping the host as it is.
#include <omp.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/jlama_hip.h"

#define JHOST_API extern "C" __attribute__((visibility("default")))

namespace {
thread_local std::string g_herr;
int herr(int code, const std::string& m) { g_herr = m; return code; }
#define HCHK(expr)                                                                                        \
    do {                                                                                                  \
        int _r = (expr);                                                                                  \
        if (_r < 0) return herr(_r, std::string(#expr) + ": " + jh_last_error());                         \
    } while (0)

constexpr int QB = 32;

struct HWeight {          // a model tensor as the Java host holds it: a host buffer + (Q4) its blockF + the ids registerModelTensor got
    int dtype = -1;
    const void* data = nullptr;
    const float* scales = nullptr;
    int rows = 0, cols = 0;
    int64_t id = -1, sid = -1;
};

inline float bf16_widen(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
}  // namespace

struct jhost {
    jh_config c;
    int ew_device = 0, threads = 1;
    std::vector<HWeight> lw;          // [n_layers][JH_W_COUNT]
    HWeight gw[JH_W_COUNT];
    std::vector<float> rope;          // Config.ropeFreqs: [context_length * head_size / 2][2]
    float attention_scale = 0.0f;
    int layers_per_page = 0, ctx_per_page = 0, n_layer_pages = 0, n_ctx_pages = 0;
    std::vector<std::vector<float>> pages;   // KvBufferPage tensors [layersPerPage, 2, ctxPerPage, kvLength], allocated on first touch
    long long provider_calls = 0;     // C-ABI calls issued (what a JVM would issue as FFM downcalls)
    double ms_provider = 0.0;
};

namespace {
struct CallTimer {   // wall time inside the C ABI (per calling thread; summed at the end of a parallel region)
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// ---- the host's own scalar code --------------------------------------------------------------------------------------------
// RMSNorm.forward core/model/RMSNorm.java:33-56: float square, DOUBLE sum in index order, /E, +eps, 1/sqrt in double, cast, w*(ss*x)
void rmsnorm_row(const float* x, const HWeight& w, int E, float eps, float* out) {
    double ss = 0.0;
    for (int j = 0; j < E; j++) { const float v = x[j]; ss += v * v; }
    ss /= E;
    ss += eps;
    ss = 1.0 / std::sqrt(ss);
    const float f = (float)ss;
    for (int j = 0; j < E; j++) {
        const float wj = w.dtype == JH_DT_BF16 ? bf16_widen(((const uint16_t*)w.data)[j]) : ((const float*)w.data)[j];
        out[j] = (0.0f + wj) * (f * x[j]);
    }
}
// VectorMath.softMax core/math/VectorMath.java:69-90
void softmax_row(float* x, int offset, int length) {
    const int size = offset + length;
    float max_val = x[offset];
    for (int i = offset + 1; i < size; i++) if (x[i] > max_val) max_val = x[i];
    float sum = 0.0f;
    for (int i = offset; i < size; i++) { x[i] = (float)std::exp((double)(x[i] - max_val)); sum += x[i]; }
    for (int i = 0; i < size; i++) x[i] = x[i] / sum;
}
// ActivationFunction.eval(SILU) core/math/ActivationFunction.java:31
inline float silu(float x) { return (float)((double)x * (1.0 / (1.0 + std::exp((double)(-x))))); }
// VectorMath.precomputeFreqsCis core/math/VectorMath.java:148-165 is a provider-independent table: the backend's own restatement is
// exported for exactly this use (jh_rope_table, host-side double cos / sin)

// ---- TensorOperations methods: the device entry point, or (elementwise_on_device = 0) what the Panama delegate computes ---------
struct Ops {
    jhost* h;
    long long calls = 0;
    double ms = 0.0;
    template <typename F> int call(F&& f) { CallTimer t; const int rc = f(); ms += t.ms(); calls++; return rc; }

    // quantize(t, I8, 0, K) PTO:1684-1723 / quantize(t, BF16) PTO:1624-1628
    int quantize_q8(const float* x, int rows, int K, int8_t* q, float* d) {
        if (h->ew_device) return call([&] { return jh_quantize_q8(x, rows, K, 0, K, q, K, d, K / QB); });
        for (int r = 0; r < rows; r++)
            for (int b = 0; b < K / QB; b++) {
                const float* xb = x + (size_t)r * K + (size_t)b * QB;
                float m = 0.0f;
                for (int t = 0; t < QB; t++) { const float a = std::fabs(xb[t]); if (a > m) m = a; }
                const float dd = m / 127.0f, id = (m != 0.0f) ? 127.0f / m : 0.0f;
                for (int t = 0; t < QB; t++) { float v = xb[t] * id; v = v + 0.5f; q[(size_t)r * K + b * QB + t] = (int8_t)(int)v; }
                d[(size_t)r * (K / QB) + b] = dd;
            }
        return JH_OK;
    }
    int quantize_bf16(const float* x, int64_t n, uint16_t* out) {
        if (h->ew_device) return call([&] { return jh_quantize_bf16(x, n, out); });
        for (int64_t i = 0; i < n; i++) {   // FloatConversions.float32ToBFloat16 core/math/FloatConversions.java:35-60 (RNE)
            uint32_t u; memcpy(&u, &x[i], 4);
            if ((u & 0x7F800000u) == 0x7F800000u) { out[i] = (u & 0x7FFFFFu) ? (uint16_t)0x7FC0 : (uint16_t)(u >> 16); continue; }
            out[i] = (uint16_t)((u + (0x7FFFu + ((u >> 16) & 1u))) >> 16);
        }
        return JH_OK;
    }
    int accumulate(float* a, const float* b, int offset, int length) {
        if (h->ew_device) return call([&] { return jh_accumulate_f32(a, b, offset, length); });
        for (int i = offset; i < offset + length; i++) a[i] = a[i] + b[i];
        return JH_OK;
    }
    int maccumulate(float* a, const float* b, int offset, int length) {
        if (h->ew_device) return call([&] { return jh_maccumulate_f32(a, b, offset, length); });
        for (int i = offset; i < offset + length; i++) a[i] = a[i] * b[i];
        return JH_OK;
    }
    int scale(float f, float* a, int offset, int length) {
        if (h->ew_device) return call([&] { return jh_scale_f32(f, a, offset, length); });
        for (int i = offset; i < offset + length; i++) a[i] = a[i] * f;
        return JH_OK;
    }
    // saxpy(alpha-tensor, x, y, xoffset, yoffset, limit, aOffset, xRowOffset, batchSize) PTO:2648-2698: one fma chain per element over rows
    int saxpy_batch(const float* alpha, const float* x, int ldx, float* y, int xoffset, int yoffset, int limit, int aoffset, int xrow, int n) {
        if (h->ew_device) return call([&] { return jh_saxpy_batch_f32(alpha, x, ldx, y, xoffset, yoffset, limit, aoffset, xrow, n); });
        const int ub = limit - (limit % 16);
        for (int r = 0; r < n; r++) {
            const float al = alpha[aoffset + r];
            const float* xr = x + (size_t)(xrow + r) * ldx;
            int t = 0;
            for (; t < ub; t++) y[yoffset + t] = std::fmaf(xr[xoffset + t], al, y[yoffset + t]);
            for (; t < limit; t++) y[yoffset + t] = y[yoffset + t] + (al * xr[xoffset + t]);
        }
        return JH_OK;
    }
};

// a quantized activation batch (LlamaModel.maybeQuantize core/model/llama/LlamaModel.java:176-184 + the dtype policy of AbstractModel.java:119-169)
struct Act {
    const float* f = nullptr;
    std::vector<int8_t> q;
    std::vector<float> d;
    std::vector<uint16_t> hb;
    int K = 0;
};
int maybe_quantize(Ops& ops, const jh_config& c, const float* x, int B, int K, Act& a) {
    a.f = x; a.K = K;
    if (c.weight_dtype == JH_DT_Q4) {
        a.q.resize((size_t)B * K); a.d.resize((size_t)B * (K / QB));
        return ops.quantize_q8(x, B, K, a.q.data(), a.d.data());
    }
    if (c.weight_dtype == JH_DT_BF16) {
        a.hb.resize((size_t)B * K);
        return ops.quantize_bf16(x, (int64_t)B * K, a.hb.data());
    }
    return JH_OK;
}
// dotProductChunk(result, a, w, colOff, K, 0, w.rows): one whole-range chunk because parallelSplitSize() == 1 (VectorMath.pchunk)
int weight_gemm(Ops& ops, const Act& a, int B, const HWeight& w, int colOff, int K, float* r, int ldc) {
    const int N = w.rows;
    if (w.dtype == JH_DT_Q4 && !a.q.empty())
        return ops.call([&] { return jh_gemm_q8_q4(w.id, w.sid, a.d.data(), a.q.data(), colOff, w.scales, (const uint8_t*)w.data, colOff / 2, r, 0, B, 0, N, K,
                                                   a.K, a.K / QB, w.cols / 2, w.cols / QB, ldc); });
    if (w.dtype == JH_DT_Q4)
        return ops.call([&] { return jh_gemm_f32_q4(w.id, w.sid, a.f, colOff, w.scales, (const uint8_t*)w.data, colOff / 2, r, 0, B, 0, N, K, a.K, w.cols / 2,
                                                    w.cols / QB, ldc); });
    if (w.dtype == JH_DT_BF16 && !a.hb.empty())
        return ops.call([&] { return jh_gemm_bf16(w.id, a.hb.data(), colOff, (const uint16_t*)w.data, colOff, nullptr, r, 0, B, 0, N, K, a.K, w.cols, ldc); });
    if (w.dtype == JH_DT_BF16)
        return ops.call([&] { return jh_gemm_f32_bf16(w.id, a.f, colOff, (const uint16_t*)w.data, colOff, nullptr, r, 0, B, 0, N, K, a.K, w.cols, ldc); });
    return ops.call([&] { return jh_gemm_f32(w.id, a.f, colOff, (const float*)w.data, colOff, r, 0, B, 0, N, K, a.K, w.cols, ldc); });
}

float* kv_row(jhost* h, int rel_layer, int idx, int pos) {   // KvBuffer.getTensorForPosition KvBufferCache.java:307-321
    const int KV = h->c.n_kv_heads * h->c.head_size;
    const int lp = rel_layer / h->layers_per_page, cp = pos / h->ctx_per_page;
    const int rl = rel_layer % h->layers_per_page, rc = pos % h->ctx_per_page;
    std::vector<float>& pg = h->pages[(size_t)lp * h->n_ctx_pages + cp];
    if (pg.empty()) pg.assign((size_t)h->layers_per_page * 2 * h->ctx_per_page * KV, 0.0f);
    return pg.data() + (((size_t)rl * 2 + idx) * h->ctx_per_page + rc) * KV;
}

// LlamaModel.loadInputWeights core/model/llama/LlamaModel.java:67-98: a Q4 table row stays Q4 (every consumer reads (nib-8)*scale)
void embed_row(jhost* h, int token, float* out) {
    const HWeight& w = h->gw[JH_W_EMBED];
    const int E = h->c.embedding_length;
    if (w.dtype == JH_DT_Q4) {
        const uint8_t* nr = (const uint8_t*)w.data + (size_t)token * (E / 2);
        const float* sr = w.scales + (size_t)token * (E / QB);
        for (int j = 0; j < E; j++) {
            const int blk = j / QB, in = j % QB;
            const uint8_t b0 = nr[blk * 16 + (in & 15)];
            const int x = in < 16 ? (b0 & 0x0F) - 8 : ((b0 >> 4) & 0x0F) - 8;
            out[j] = (float)x * sr[blk];
        }
    } else if (w.dtype == JH_DT_BF16) {
        for (int j = 0; j < E; j++) out[j] = bf16_widen(((const uint16_t*)w.data)[(size_t)token * E + j]);
    } else {
        memcpy(out, (const float*)w.data + (size_t)token * E, sizeof(float) * (size_t)E);
    }
}

// CausalSelfAttention.forward (GQA branch) core/model/CausalSelfAttention.java:145-385.  x: [B, E] in; att_out: [B, E] out (no residual)
int attention_forward(jhost* h, Ops& ops, int li, const float* x, int B, int start_pos, float* att_out) {
    const jh_config& c = h->c;
    const int E = c.embedding_length, hs = c.head_size, A = c.n_heads * hs, KV = c.n_kv_heads * hs, half = hs / 2;
    const int group = c.n_heads / c.n_kv_heads, rel = li - c.layer_start;
    const HWeight* W = &h->lw[(size_t)li * JH_W_COUNT];
    std::vector<float> ln((size_t)B * E), q((size_t)B * A), k((size_t)B * KV), v((size_t)B * KV), val((size_t)B * A, 0.0f);
    for (int b = 0; b < B; b++) rmsnorm_row(x + (size_t)b * E, W[JH_W_NORM1], E, c.rms_eps, ln.data() + (size_t)b * E);   // preAttentionNorm TransformerBlock.java:167
    Act qa;
    HCHK(maybe_quantize(ops, c, ln.data(), B, E, qa));                                                                   // :172
    HCHK(weight_gemm(ops, qa, B, W[JH_W_Q], 0, E, q.data(), A));                                                         // :161-171, three dotProductChunk calls
    HCHK(weight_gemm(ops, qa, B, W[JH_W_K], 0, E, k.data(), KV));
    HCHK(weight_gemm(ops, qa, B, W[JH_W_V], 0, E, v.data(), KV));
    const int max_ctx_alloc = ((start_pos + B) / h->ctx_per_page + 1) * h->ctx_per_page;
    const int T = h->threads;
    std::vector<float> attn_all((size_t)max_ctx_alloc * T);
    std::vector<Ops> tops((size_t)T, Ops{h});
    for (int bi = 0, position = start_pos; bi < B; bi++, position++) {
        float* key = kv_row(h, rel, 0, position);
        float* vrow = kv_row(h, rel, 1, position);
        memcpy(key, k.data() + (size_t)bi * KV, sizeof(float) * (size_t)KV);     // :226-241 copyFrom
        memcpy(vrow, v.data() + (size_t)bi * KV, sizeof(float) * (size_t)KV);
        float* query = q.data() + (size_t)bi * A;
        float* value = val.data() + (size_t)bi * A;
        // RoPE :247-286: table row position*half + g with g over kvHead*hs + [0, half) => effective position pos + 2*kvHead; the
        // reference throws past its table (ArrayIndexOutOfBounds), mirrored as an error
        if ((long long)position + 2LL * (c.n_kv_heads - 1) >= (long long)c.context_length)
            return herr(JH_ERR_INVALID, "RoPE: position + 2*(n_kv_heads-1) is beyond the table (CausalSelfAttention.java:260-283)");
        const size_t poffset = (size_t)position * half;
        for (int hd = 0; hd < c.n_heads; hd++) {
            const int offset = hd * hs, goffset = (hd / group) * hs;
            for (int i = offset, gg = goffset; i < offset + half; i++, gg++) {
                const float q0 = query[i], q1 = query[i + half];
                const float fcr = h->rope[(poffset + gg) * 2], fci = h->rope[(poffset + gg) * 2 + 1];
                query[i] = q0 * fcr - q1 * fci;
                query[i + half] = q0 * fci + q1 * fcr;
            }
        }
        for (int hd = 0; hd < c.n_kv_heads; hd++) {
            const int offset = hd * hs;
            for (int i = offset; i < offset + half; i++) {
                const float k0 = key[i], k1 = key[i + half];
                const float fcr = h->rope[(poffset + i) * 2], fci = h->rope[(poffset + i) * 2 + 1];
                key[i] = k0 * fcr - k1 * fci;
                key[i + half] = k0 * fci + k1 * fcr;
            }
        }
        const int npages = position / h->ctx_per_page + 1;
        for (int pg = 0; pg < npages; pg++) (void)kv_row(h, rel, 0, pg * h->ctx_per_page);   // materialise before the parallel loop
        int rc_all = JH_OK;
        std::string err_all;
        // VectorMath.pfor(headStart, headEnd, ...) core/math/VectorMath.java:34-36: heads on the ForkJoin workers, each calling the provider
#pragma omp parallel for schedule(static) num_threads(T) if (T > 1)
        for (int hd = 0; hd < c.n_heads; hd++) {
            const int tid = omp_get_thread_num();
            Ops& o = tops[(size_t)tid];
            float* attn = attn_all.data() + (size_t)max_ctx_alloc * tid;
            const int xoffset = (hd / group) * hs, yoffset = hd * hs;
            int rc = JH_OK;
            for (int pg = 0; pg < npages && rc >= 0; pg++) {
                const int len = h->ctx_per_page, off = pg * len, size = pg == npages - 1 ? (position + 1) - off : len;
                const float* kpage = kv_row(h, rel, 0, off);
                // batchDotProduct(attn, query, kvp[i], yoffset, xoffset, headSize, offset, 0, size) :328-329 -> roffset = -rRowOffset
                rc = o.call([&] { return jh_gemm_f32(-1, query, yoffset, kpage, xoffset, attn, -off, 1, 0, size, hs, A, KV, max_ctx_alloc); });
            }
            if (rc >= 0) rc = o.scale(h->attention_scale, attn, 0, position + 1);   // :332
            if (rc >= 0) {
                softmax_row(attn, 0, position + 1);                                 // :345 (plain Java)
                for (int pg = 0; pg < npages && rc >= 0; pg++) {
                    const int len = h->ctx_per_page, off = pg * len, size = pg == npages - 1 ? (position + 1) - off : len;
                    const float* vpage = kv_row(h, rel, 1, off);
                    rc = o.saxpy_batch(attn, vpage, KV, value, xoffset, yoffset, hs, off, 0, size);   // :349-354
                }
            }
            if (rc < 0) {
#pragma omp critical
                { rc_all = rc; err_all = jh_last_error(); }
            }
        }
        if (rc_all < 0) return herr(rc_all, "attention head: " + err_all);
    }
    for (Ops& o : tops) { ops.calls += o.calls; ops.ms += o.ms / (T > 1 ? T : 1); }
    Act va;
    HCHK(maybe_quantize(ops, c, val.data(), B, A, va));                          // :364
    HCHK(weight_gemm(ops, va, B, W[JH_W_O], 0, A, att_out, E));                  // :365-376
    return JH_OK;
}

// MLPBlock.forward core/model/MLPBlock.java:105-166.  att_res: [B, E] in; ff: [B, E] out (no residual)
int mlp_forward(jhost* h, Ops& ops, int li, const float* att_res, int B, float* ff) {
    const jh_config& c = h->c;
    const int E = c.embedding_length, H = c.hidden_length;
    const HWeight* W = &h->lw[(size_t)li * JH_W_COUNT];
    std::vector<float> ln((size_t)B * E), g((size_t)B * H), u((size_t)B * H);
    for (int b = 0; b < B; b++) rmsnorm_row(att_res + (size_t)b * E, W[JH_W_NORM2], E, c.rms_eps, ln.data() + (size_t)b * E);   // preFFNorm TransformerBlock.java:187
    Act fa;
    HCHK(maybe_quantize(ops, c, ln.data(), B, E, fa));                                                                         // :192
    {   // dotProductBatchChunk({buf, buf2}, lnemb, {fullyConnected, upProjection}, 0, E, 0, H) :117-125 -> the `_batch` entry point
        const HWeight& G = W[JH_W_GATE];
        const HWeight& U = W[JH_W_UP];
        const int64_t ids[2] = {G.id, U.id}, sids[2] = {G.sid, U.sid};
        float* rs[2] = {g.data(), u.data()};
        if (G.dtype == JH_DT_Q4) {
            const float* bfs[2] = {G.scales, U.scales};
            const uint8_t* bs[2] = {(const uint8_t*)G.data, (const uint8_t*)U.data};
            HCHK(ops.call([&] { return jh_gemm_q8_q4_batch(2, ids, sids, fa.d.data(), fa.q.data(), 0, bfs, bs, 0, rs, 0, B, 0, H, E, E, E / QB, E / 2, E / QB, H); }));
        } else if (G.dtype == JH_DT_BF16) {
            const uint16_t* bs[2] = {(const uint16_t*)G.data, (const uint16_t*)U.data};
            HCHK(ops.call([&] { return jh_gemm_bf16_batch(2, ids, fa.hb.data(), 0, bs, 0, nullptr, rs, 0, B, 0, H, E, E, E, H); }));
        } else {
            const float* bs[2] = {(const float*)G.data, (const float*)U.data};
            HCHK(ops.call([&] { return jh_gemm_f32_batch(2, ids, fa.f, 0, bs, 0, rs, 0, B, 0, H, E, E, E, H); }));
        }
    }
    for (size_t t = 0; t < (size_t)B * H; t++) g[t] = silu(g[t]);                  // :132-140 ActivationFunction.eval per element (plain Java)
    for (int b = 0; b < B; b++) HCHK(ops.maccumulate(g.data() + (size_t)b * H, u.data() + (size_t)b * H, 0, H));   // :141
    Act ha;
    HCHK(maybe_quantize(ops, c, g.data(), B, H, ha));                              // :144
    HCHK(weight_gemm(ops, ha, B, W[JH_W_DOWN], 0, H, ff, E));                      // :147-158
    return JH_OK;
}
}  // namespace

JHOST_API const char* jhost_last_error(void) { return g_herr.c_str(); }

// elementwise_on_device: 0 = provider default (element-wise methods on the Panama delegate), 1 = every provider method on the device.
// threads: workers of the heads' pfor (the reference: PhysicalCoreExecutor, cores / 2).
JHOST_API int jhost_create(const jh_config* cfg, int elementwise_on_device, int threads, int64_t max_page_bytes, jhost** out) {
    if (!cfg || !out) return herr(JH_ERR_INVALID, "jhost_create: null argument");
    jhost* h = new jhost();
    h->c = *cfg;
    h->ew_device = elementwise_on_device ? 1 : 0;
    h->threads = threads > 0 ? threads : 1;
    h->lw.resize((size_t)cfg->n_layers * JH_W_COUNT);
    const int half = cfg->head_size / 2;
    h->rope.resize((size_t)cfg->context_length * half * 2);
    int rc = jh_rope_table(cfg->head_size, cfg->context_length, (double)cfg->rope_theta, (double)cfg->rope_scaling, h->rope.data());   // Config ctor core/safetensors/Config.java:270-274
    if (rc < 0) { delete h; return herr(rc, std::string("jh_rope_table: ") + jh_last_error()); }
    h->attention_scale = (float)(1.0 / std::sqrt((double)cfg->head_size));   // CausalSelfAttention.java:134
    const int KV = cfg->n_kv_heads * cfg->head_size, nl = cfg->layer_end - cfg->layer_start;
    int32_t geo[2] = {0, 0};
    rc = jh_kv_page_geometry(max_page_bytes > 0 ? max_page_bytes : (1 << 23), nl, cfg->context_length, KV, 4, geo);
    if (rc < 0) { delete h; return herr(rc, std::string("jh_kv_page_geometry: ") + jh_last_error()); }
    h->layers_per_page = geo[0]; h->ctx_per_page = geo[1];
    h->n_layer_pages = (nl + geo[0] - 1) / geo[0];
    h->n_ctx_pages = (cfg->context_length + geo[1] - 1) / geo[1];
    h->pages.resize((size_t)h->n_layer_pages * h->n_ctx_pages);
    *out = h;
    return JH_OK;
}
JHOST_API void jhost_destroy(jhost* h) {
    if (!h) return;
    for (HWeight& w : h->lw) { if (w.id >= 0) jh_unregister_tensor(w.id); if (w.sid >= 0) jh_unregister_tensor(w.sid); }
    for (HWeight& w : h->gw) { if (w.id >= 0) jh_unregister_tensor(w.id); if (w.sid >= 0) jh_unregister_tensor(w.sid); }
    delete h;
}
// Host pointers are borrowed (the Java host owns mmap'd weights).  Projection weights, the LM head / tied table are registered like
// registerModelTensor does from the block constructors (CausalSelfAttention.java:139-142, MLPBlock.java:97-101, LlamaModel.java:160):
// Q4 registers nibbles AND blockF; on JH_ERR_OOM the tensor stays on the host and is shipped per call.
JHOST_API int jhost_set_weight(jhost* h, int layer, int which, int dtype, const void* data, const float* scales, int rows, int cols) {
    if (!h || which < 0 || which >= JH_W_COUNT || !data) return herr(JH_ERR_INVALID, "jhost_set_weight: bad argument");
    HWeight& w = layer < 0 ? h->gw[which] : h->lw[(size_t)layer * JH_W_COUNT + which];
    w.dtype = dtype; w.data = data; w.scales = scales; w.rows = rows; w.cols = cols;
    const bool matrix = which != JH_W_NORM1 && which != JH_W_NORM2 && which != JH_W_FINALNORM;
    const bool lm = which == JH_W_LMHEAD || (which == JH_W_EMBED);   // the embedding table doubles as the LM head when tied (LlamaModel.java:155-158)
    if (matrix && (layer >= 0 || lm)) {
        const int64_t bytes = dtype == JH_DT_Q4 ? (int64_t)rows * cols / 2 : dtype == JH_DT_BF16 ? (int64_t)rows * cols * 2 : (int64_t)rows * cols * 4;
        const int64_t id = jh_register_tensor(data, bytes);
        if (id < 0 && id != JH_ERR_OOM) return herr((int)id, std::string("jh_register_tensor: ") + jh_last_error());
        w.id = id >= 0 ? id : -1;
        if (dtype == JH_DT_Q4 && w.id >= 0) {
            const int64_t sid = jh_register_tensor(scales, (int64_t)rows * (cols / QB) * 4);
            if (sid < 0 && sid != JH_ERR_OOM) return herr((int)sid, std::string("jh_register_tensor: ") + jh_last_error());
            if (sid < 0) { jh_unregister_tensor(w.id); w.id = -1; } else w.sid = sid;
        }
    }
    return JH_OK;
}

// AbstractModel.forward / batchForward core/model/AbstractModel.java:267-329: B rows through layers [layer_start, layer_end).
// tokens != NULL: rows come from the embedding table; else x holds them.  x: [B, E] in/out.
JHOST_API int jhost_forward(jhost* h, const int32_t* tokens, float* x, int B, int start_pos) {
    if (!h || !x || B <= 0) return herr(JH_ERR_INVALID, "jhost_forward: bad argument");
    const jh_config& c = h->c;
    const int E = c.embedding_length;
    Ops ops{h};
    if (tokens) for (int b = 0; b < B; b++) embed_row(h, tokens[b], x + (size_t)b * E);
    std::vector<float> att_out((size_t)B * E), ff((size_t)B * E);
    for (int li = c.layer_start; li < c.layer_end; li++) {
        int rc = attention_forward(h, ops, li, x, B, start_pos, att_out.data());
        if (rc < 0) return rc;
        for (int b = 0; b < B; b++) HCHK(ops.accumulate(att_out.data() + (size_t)b * E, x + (size_t)b * E, 0, E));   // TransformerBlock.java:185
        rc = mlp_forward(h, ops, li, att_out.data(), B, ff.data());
        if (rc < 0) return rc;
        for (int b = 0; b < B; b++) HCHK(ops.accumulate(ff.data() + (size_t)b * E, att_out.data() + (size_t)b * E, 0, E));   // :203
        memcpy(x, ff.data(), sizeof(float) * (size_t)B * E);
    }
    h->provider_calls += ops.calls;
    h->ms_provider += ops.ms;
    return JH_OK;
}

// AbstractModel.sample core/model/AbstractModel.java:443-491 at temperature 0: final RMSNorm (plain Java) -> LM head through the provider
// (the F32 row is NOT re-quantized, :443-449) -> argmax with strict > (first maximum wins, :455-469).  logits: [V] out.
JHOST_API int jhost_sample(jhost* h, const float* last_row, float* logits, int32_t* token) {
    if (!h || !last_row || !logits || !token) return herr(JH_ERR_INVALID, "jhost_sample: bad argument");
    const jh_config& c = h->c;
    const int E = c.embedding_length, V = c.vocab_size;
    std::vector<float> emb((size_t)E);
    rmsnorm_row(last_row, h->gw[JH_W_FINALNORM], E, c.rms_eps, emb.data());
    Ops ops{h};
    Act a;
    a.f = emb.data(); a.K = E;
    const HWeight& w = h->gw[JH_W_LMHEAD].data ? h->gw[JH_W_LMHEAD] : h->gw[JH_W_EMBED];
    HCHK(weight_gemm(ops, a, 1, w, 0, E, logits, V));
    int maxi = INT32_MIN;
    double maxv = -INFINITY;
    for (int i = 0; i < V; i++) { const float v = logits[i]; if (v > maxv) { maxi = i; maxv = v; } }
    *token = maxi;
    h->provider_calls += ops.calls;
    h->ms_provider += ops.ms;
    return JH_OK;
}

// AbstractModel.generate core/model/AbstractModel.java:515-646 at the token-id level, temperature 0: batchForward(prompt) in chunks of
// jlama.max_batch_size = 256 (:57,:304) -> sample -> decode loop.  times_ms: [0] prompt, [1] decode (clock starts after the first
// sampled token, :589), [2] wall time inside the C ABI over the whole call, [3] provider calls.
JHOST_API int jhost_generate(jhost* h, const int32_t* prompt, int n_prompt, int n_gen, int32_t* out_tokens, float* logits_last, double* times_ms) {
    if (!h || !prompt || n_prompt <= 0 || n_gen <= 0 || !out_tokens) return herr(JH_ERR_INVALID, "jhost_generate: bad argument");
    const int E = h->c.embedding_length, V = h->c.vocab_size, MAXB = 256;
    std::vector<float> x((size_t)MAXB * E), logits((size_t)V);
    const long long calls0 = h->provider_calls;
    const double ms0 = h->ms_provider;
    const auto t0 = std::chrono::steady_clock::now();
    int lastB = 0;
    for (int i = 0; i < n_prompt; i += MAXB) {
        const int B = n_prompt - i < MAXB ? n_prompt - i : MAXB;
        int rc = jhost_forward(h, prompt + i, x.data(), B, i);
        if (rc < 0) return rc;
        lastB = B;
    }
    int32_t next = 0;
    int rc = jhost_sample(h, x.data() + (size_t)(lastB - 1) * E, logits.data(), &next);
    if (rc < 0) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    int n = 0;
    out_tokens[n++] = next;
    for (int pos = n_prompt; n < n_gen; pos++) {
        const int32_t tok = next;
        rc = jhost_forward(h, &tok, x.data(), 1, pos);
        if (rc < 0) return rc;
        rc = jhost_sample(h, x.data(), logits.data(), &next);
        if (rc < 0) return rc;
        out_tokens[n++] = next;
    }
    const auto t2 = std::chrono::steady_clock::now();
    if (logits_last) memcpy(logits_last, logits.data(), sizeof(float) * (size_t)V);
    if (times_ms) {
        times_ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
        times_ms[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
        times_ms[2] = h->ms_provider - ms0;
        times_ms[3] = (double)(h->provider_calls - calls0);
    }
    return n;
}
JHOST_API int jhost_page_info(jhost* h, int32_t* out4) {
    if (!h || !out4) return herr(JH_ERR_INVALID, "jhost_page_info: bad argument");
    out4[0] = h->layers_per_page; out4[1] = h->ctx_per_page; out4[2] = h->n_layer_pages; out4[3] = h->n_ctx_pages;
    return JH_OK;
}
