"""ctypes binding of the parity ORACLE (oracle/jlama_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (jlama_amd / libjlamahip.so) never does; see DESIGN.md "Oracle".

Also binds oracle/_ref/libjlama_ref_*.so = the reference's own C SIMD GEMM
(jlama-native/src/main/c/simd/vector_simd.c compiled as-is by oracle/Makefile), used to
validate the restated GEMMs and as the CPU baseline ("reference native-SIMD GEMM + restated
Java ops").
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DT_F32, DT_BF16, DT_I8, DT_Q4 = 0, 1, 2, 3
(W_Q, W_K, W_V, W_O, W_GATE, W_UP, W_DOWN, W_NORM1, W_NORM2, W_EMBED, W_LMHEAD, W_FINALNORM) = range(12)
# GPT-2 family extras (core/model/gpt2/GPT2Model.java:53-129): biases, LayerNorm biases, learned position embeddings
(W_QB, W_KB, W_VB, W_OB, W_GATEB, W_DOWNB, W_NORM1B, W_NORM2B, W_WPE, W_FINALNORMB) = range(12, 22)
ARCH_LLAMA, ARCH_GPT2 = 0, 1
TAPS = ["input_emb", "ln_emb", "query", "key", "value", "query+rope", "key+rope", "after_attention",
        "post_attn", "pre_ff_norm", "post_ff", "post_ff_res"]

# flags the Java wrapper passes on an AVX-512 Linux host
# (jlama-native/.../NativeSimdTensorOperations.java:55-63, vector_simd.h:13-15)
REF_FLAGS = 4 + 2  # HAS_AVX2 | HAS_F16C


class Config(C.Structure):
    _fields_ = [("embedding_length", C.c_int32), ("hidden_length", C.c_int32), ("n_heads", C.c_int32),
                ("n_kv_heads", C.c_int32), ("head_size", C.c_int32), ("n_layers", C.c_int32),
                ("vocab_size", C.c_int32), ("context_length", C.c_int32), ("weight_dtype", C.c_int32),
                ("layer_start", C.c_int32), ("layer_end", C.c_int32), ("rms_eps", C.c_float),
                ("rope_theta", C.c_float), ("rope_scaling", C.c_float)]


def build(force=False):
    so = os.path.join(HERE, "_build", "libjlama_oracle.so")
    src = os.path.join(HERE, "jlama_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return so


_lib = None
_ref = None


def available_cpus():
    """CPUs this process may actually use: affinity mask AND cgroup quota (what Java's availableProcessors() reports;
    the reference sizes its pool from it, PhysicalCoreExecutor.java:27)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return max(1, n)


if "OMP_NUM_THREADS" not in os.environ:
    os.environ["OMP_NUM_THREADS"] = str(available_cpus())  # before libgomp loads: never oversubscribe a cgroup quota


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.jo_bf16_to_f32.restype = C.c_float
        _lib.jo_bf16_to_f32.argtypes = [C.c_uint16]
        _lib.jo_f32_to_bf16.restype = C.c_uint16
        _lib.jo_f32_to_bf16.argtypes = [C.c_float]
        _lib.jo_silu.restype = C.c_float
        _lib.jo_silu.argtypes = [C.c_float]
        _lib.jo_gelu.restype = C.c_float
        _lib.jo_gelu.argtypes = [C.c_float]
        _lib.jo_model_create.restype = C.c_void_p
        _lib.jo_session_create.restype = C.c_void_p
        _lib.jo_session_create.argtypes = [C.c_void_p, C.c_int64]
    return _lib


def _cpu_flags():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":")[1].split())
    except OSError:
        pass
    return set()


def ref_lib():
    """The reference's C SIMD GEMM library, or None when oracle/_ref was never built."""
    global _ref
    if _ref is None:
        fl = _cpu_flags()
        names = []
        if {"avx512f", "avx512vl", "avx512bw", "avx512_vnni"} <= fl:
            names.append("libjlama_ref_avx512.so")
        if "avx2" in fl and "fma" in fl:
            names.append("libjlama_ref_avx2.so")
        for n in names:
            p = os.path.join(HERE, "_ref", n)
            if os.path.exists(p):
                _ref = C.CDLL(p)
                _ref._path = p
                break
        else:
            _ref = False
    return _ref or None


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t) if a is not None else None


# ----------------------------------------------------------------------------- quantizers
def q4_quantize(x):
    """F32 [rows, cols] -> (nibbles uint8 [rows, cols/2], scales f32 [rows, cols/32])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, cols = x.shape
    assert cols % 32 == 0
    nib = np.empty((rows, cols // 2), dtype=np.uint8)
    sc = np.empty((rows, cols // 32), dtype=np.float32)
    lib().jo_q4_quantize(_p(x), C.c_int64(rows), C.c_int(cols), _p(nib), _p(sc))
    return nib, sc


def q4_dequantize(nib, sc):
    rows, half = nib.shape
    out = np.empty((rows, half * 2), dtype=np.float32)
    lib().jo_q4_dequantize(_p(nib), _p(sc), C.c_int64(rows), C.c_int(half * 2), _p(out))
    return out


def q8_quantize(x):
    """Panama quantizeQ8_512: F32 [rows, K] -> (int8 [rows,K], f32 [rows,K/32])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, K = x.shape
    q = np.empty((rows, K), dtype=np.int8)
    d = np.empty((rows, K // 32), dtype=np.float32)
    lib().jo_q8_quantize(_p(x), rows, K, 0, K, _p(q), K, _p(d), K // 32)
    return q, d


def bf16_quantize(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().jo_bf16_quantize(_p(x), C.c_int64(x.size), _p(out))
    return out


def bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


# ----------------------------------------------------------------------------- GEMMs
def gemm_i8q4(aq, ad, bn, bs, M=None, aColOff=0, bColOff=0, K=None, rRowOff=0, bRowOff=0, N=None, out=None):
    M = aq.shape[0] if M is None else M
    K = aq.shape[1] if K is None else K
    N = bn.shape[0] if N is None else N
    if out is None:
        out = np.zeros((M, rRowOff + bRowOff + N), dtype=np.float32)
    lib().jo_gemm_i8q4(_p(aq), _p(ad), aq.shape[1], ad.shape[1], _p(bn), _p(bs), bn.shape[1], bs.shape[1],
                       _p(out), out.shape[1], M, aColOff, bColOff, K, rRowOff, bRowOff, N)
    return out


def gemm_f32q4(a, bn, bs, M=None, aColOff=0, bColOff=0, K=None, rRowOff=0, bRowOff=0, N=None, out=None):
    M = a.shape[0] if M is None else M
    K = a.shape[1] if K is None else K
    N = bn.shape[0] if N is None else N
    if out is None:
        out = np.zeros((M, rRowOff + bRowOff + N), dtype=np.float32)
    lib().jo_gemm_f32q4(_p(a), a.shape[1], _p(bn), _p(bs), bn.shape[1], bs.shape[1], _p(out), out.shape[1], M,
                        aColOff, bColOff, K, rRowOff, bRowOff, N)
    return out


def _gemm_dense(fn, a, b, M, aColOff, bColOff, K, rRowOff, bRowOff, N, out):
    M = a.shape[0] if M is None else M
    K = a.shape[1] if K is None else K
    N = b.shape[0] if N is None else N
    if out is None:
        out = np.zeros((M, rRowOff + bRowOff + N), dtype=np.float32)
    fn(_p(a), a.shape[1], _p(b), b.shape[1], _p(out), out.shape[1], M, aColOff, bColOff, K, rRowOff, bRowOff, N)
    return out


def gemm_f32(a, b, M=None, aColOff=0, bColOff=0, K=None, rRowOff=0, bRowOff=0, N=None, out=None):
    return _gemm_dense(lib().jo_gemm_f32, a, b, M, aColOff, bColOff, K, rRowOff, bRowOff, N, out)


def gemm_bf16(a, b, M=None, aColOff=0, bColOff=0, K=None, rRowOff=0, bRowOff=0, N=None, out=None):
    return _gemm_dense(lib().jo_gemm_bf16, a, b, M, aColOff, bColOff, K, rRowOff, bRowOff, N, out)


def gemm_f32bf16(a, b, M=None, aColOff=0, bColOff=0, K=None, rRowOff=0, bRowOff=0, N=None, out=None):
    return _gemm_dense(lib().jo_gemm_f32bf16, a, b, M, aColOff, bColOff, K, rRowOff, bRowOff, N, out)


def gemm_naive(adt, a, af, bdt, b, bf, M, aColOff, bColOff, K, rRowOff, bRowOff, N, lda=None, ldb=None):
    """NaiveTensorOperations.batchDotProduct (the reference tests' control implementation)."""
    out = np.zeros((M, rRowOff + N), dtype=np.float32)
    lda = lda if lda is not None else (a.shape[1] * (2 if adt == DT_Q4 else 1))
    ldb = ldb if ldb is not None else (b.shape[1] * (2 if bdt == DT_Q4 else 1))
    ldaf = af.shape[1] if af is not None else 0
    ldbf = bf.shape[1] if bf is not None else 0
    lib().jo_gemm_naive(adt, _p(a), _p(af), lda, ldaf, bdt, _p(b), _p(bf), ldb, ldbf, _p(out), out.shape[1], M,
                        aColOff, bColOff, K, rRowOff, bRowOff, N)
    return out


# reference C library wrappers (argument marshaling follows NativeSimdTensorOperations.java:96-107,204-223)
def ref_gemm_q8_q4(aq, ad, bn, bs, aColOff=0, bColOff=0, K=None, rRowOff=0, bRowOff=0, N=None):
    r = ref_lib()
    M = aq.shape[0]
    K = aq.shape[1] if K is None else K
    N = bn.shape[0] if N is None else N
    out = np.zeros((M, rRowOff + bRowOff + N), dtype=np.float32)
    r.gemm_q8_q4(REF_FLAGS, _p(ad), _p(aq), aColOff, _p(bs), _p(bn), bColOff // 2, _p(out), -rRowOff, M, bRowOff, N,
                 K, aq.shape[1], ad.shape[1], bn.shape[1], bs.shape[1], out.shape[1])
    return out


def ref_gemm_f32_q4(a, bn, bs, aColOff=0, bColOff=0, K=None, rRowOff=0, bRowOff=0, N=None):
    r = ref_lib()
    M = a.shape[0]
    K = a.shape[1] if K is None else K
    N = bn.shape[0] if N is None else N
    out = np.zeros((M, rRowOff + bRowOff + N), dtype=np.float32)
    r.gemm_f32_q4(REF_FLAGS, _p(a), aColOff, _p(bs), _p(bn), bColOff // 2, _p(out), -rRowOff, M, bRowOff, N, K,
                  a.shape[1], bn.shape[1], bs.shape[1], out.shape[1])
    return out


def ref_gemm_f32(a, b, aColOff=0, bColOff=0, K=None, rRowOff=0, bRowOff=0, N=None):
    r = ref_lib()
    M = a.shape[0]
    K = a.shape[1] if K is None else K
    N = b.shape[0] if N is None else N
    out = np.zeros((M, rRowOff + bRowOff + N), dtype=np.float32)
    r.gemm_f32(REF_FLAGS, _p(a), aColOff, _p(b), bColOff, _p(out), -rRowOff, M, bRowOff, N, K, a.shape[1], b.shape[1],
               out.shape[1])
    return out


def ref_gemm_bf16(a, b, aColOff=0, bColOff=0, K=None, rRowOff=0, bRowOff=0, N=None):
    """gemm_bf16 of the reference C library with F32 output (rs = NULL), M = 1 ONLY: gemm_bf16_512 indexes A with the
    COLUMN tile index (`params.lda * (jj + ni)`, vector_simd.c:1189) -- a reference bug that is harmless exactly when
    the A row stride passed is 0, which is legitimate for a single row."""
    r = ref_lib()
    assert a.shape[0] == 1, "reference gemm_bf16 is only usable for M = 1 (vector_simd.c:1189)"
    K = a.shape[1] if K is None else K
    N = b.shape[0] if N is None else N
    out = np.zeros((1, rRowOff + bRowOff + N), dtype=np.float32)
    r.gemm_bf16(REF_FLAGS, _p(a), aColOff, _p(b), bColOff, None, _p(out), -rRowOff, 1, bRowOff, N, K, 0, b.shape[1],
                out.shape[1])
    return out


def ref_gemm_f32_bf16(a, b, aColOff=0, bColOff=0, K=None, rRowOff=0, bRowOff=0, N=None):
    r = ref_lib()
    M = a.shape[0]
    K = a.shape[1] if K is None else K
    N = b.shape[0] if N is None else N
    out = np.zeros((M, rRowOff + bRowOff + N), dtype=np.float32)
    r.gemm_f32_bf16(REF_FLAGS, _p(a), aColOff, _p(b), bColOff, None, _p(out), -rRowOff, M, bRowOff, N, K, a.shape[1],
                    b.shape[1], out.shape[1])
    return out


# ----------------------------------------------------------------------------- small ops
def rmsnorm(x, w, eps, weight_adj=0.0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty_like(x)
    lib().jo_rmsnorm(_p(x), _p(w), C.c_float(weight_adj), x.size, C.c_float(eps), _p(out))
    return out


def layernorm(x, w, b, eps, offset=0, length=None, divisor=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    x = x.reshape(1, -1) if x.ndim == 1 else x
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    length = x.shape[1] - offset if length is None else length
    out = x.copy()
    for r in range(x.shape[0]):
        lib().jo_layernorm(_p(x[r]), _p(w), _p(b), offset, length, x.shape[1] if divisor is None else divisor, C.c_float(eps), _p(out[r]))
    return out


def gelu(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return np.array([lib().jo_gelu(C.c_float(v)) for v in x.reshape(-1)], dtype=np.float32).reshape(x.shape)


def rope_table(dim, end, theta, scaling=1.0):
    out = np.empty((end * (dim // 2), 2), dtype=np.float32)
    lib().jo_rope_table(dim, end, C.c_double(theta), C.c_double(scaling), _p(out))
    return out


def softmax(x, offset, length):
    x = np.ascontiguousarray(x, dtype=np.float32).copy()
    lib().jo_softmax(_p(x), offset, length)
    return x


def silu(x):
    x = np.asarray(x, dtype=np.float32)
    return np.array([lib().jo_silu(C.c_float(float(v))) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


def saxpy_batch(alpha, x, y, xoffset, yoffset, limit, aOffset, xRowOffset, batchSize):
    y = np.ascontiguousarray(y, dtype=np.float32).copy()
    lib().jo_saxpy_batch_f32(_p(alpha), _p(x), x.shape[1], _p(y), xoffset, yoffset, limit, aOffset, xRowOffset,
                             batchSize)
    return y


def kv_page_geometry(max_page_bytes, n_layers, context_length, kv_length, dtype_size=4):
    a, b = C.c_int(), C.c_int()
    lib().jo_kv_page_geometry(C.c_int64(max_page_bytes), n_layers, context_length, kv_length, dtype_size,
                              C.byref(a), C.byref(b))
    return a.value, b.value


# ----------------------------------------------------------------------------- whole model
class OracleModel:
    """CPU restatement of AbstractModel.generate()/forward() for Llama-family models."""

    def __init__(self, cfg: dict, weights: dict, layer_range=None, kv_head_offset=0, arch=ARCH_LLAMA):
        """kv_head_offset: for a tensor-parallel shard (cfg carries the LOCAL head counts / hidden length), the global
        index of its first kv head (DistributedContext.groupHeadStart) -- RoPE table rows are indexed globally.
        arch=ARCH_GPT2: LayerNorm + biases, wte + wpe embeddings, GELU MLP without up-projection, no RoPE, LM head = wte
        (BASELINE.json configs[0], the reference's CPU plumbing case)."""
        L = cfg["n_layers"]
        ls, le = layer_range if layer_range else (0, L)
        self.cfg = cfg
        self.c = Config(cfg["embedding_length"], cfg["hidden_length"], cfg["n_heads"], cfg["n_kv_heads"],
                        cfg["head_size"], L, cfg["vocab_size"], cfg["context_length"], cfg["weight_dtype"], ls, le,
                        cfg["rms_eps"], cfg["rope_theta"], cfg.get("rope_scaling", 1.0))
        self.m = C.c_void_p(lib().jo_model_create(C.byref(self.c)))
        self._keep = weights  # borrowed pointers
        for (layer, which), w in weights.items():
            data, scales, dtype = w["data"], w.get("scales"), w["dtype"]
            rows, cols = w["shape"]
            rc = lib().jo_model_set_weight(self.m, layer, which, dtype, _p(data), _p(scales), rows, cols)
            assert rc == 0
        if kv_head_offset:
            lib().jo_model_set_kv_head_offset(self.m, int(kv_head_offset))
        if arch:
            lib().jo_model_set_arch(self.m, int(arch))

    def embed_rows(self, tokens):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        x = np.empty((tokens.size, self.cfg["embedding_length"]), dtype=np.float32)
        lib().jo_embed_rows(self.m, _p(tokens), tokens.size, _p(x))
        return x

    def use_reference_gemm(self, nthreads):
        r = ref_lib()
        if r is None:
            raise RuntimeError("oracle/_ref not built")
        q = C.cast(r.gemm_q8_q4, C.c_void_p)
        f = C.cast(r.gemm_f32_q4, C.c_void_p)
        lib().jo_model_set_ref_gemm(self.m, q, f, REF_FLAGS, nthreads)

    def session(self, max_page_bytes=1 << 23):
        return OracleSession(self, max_page_bytes)

    def sample(self, last_row, temperature=0.0, u=0.5):
        logits = np.empty(self.cfg["vocab_size"], dtype=np.float32)
        last_row = np.ascontiguousarray(last_row, dtype=np.float32)
        tok = lib().jo_sample(self.m, _p(last_row), C.c_float(temperature), C.c_float(u), _p(logits))
        return tok, logits

    def __del__(self):
        try:
            lib().jo_model_destroy(self.m)
        except Exception:
            pass


def forward_tp(sessions, tokens, start_pos, x=None):
    """All tensor-parallel shards in one process, partials summed in shard order (jo_forward_tp)."""
    E = sessions[0].model.cfg["embedding_length"]
    arr = (C.c_void_p * len(sessions))(*[s.s for s in sessions])
    if x is None:
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        x = np.empty((tokens.size, E), dtype=np.float32)
        lib().jo_forward_tp(arr, len(sessions), _p(tokens), _p(x), tokens.size, start_pos)
    else:
        x = np.ascontiguousarray(x, dtype=np.float32).copy()
        lib().jo_forward_tp(arr, len(sessions), None, _p(x), x.shape[0], start_pos)
    return x


class OracleSession:
    def __init__(self, model, max_page_bytes):
        self.model = model
        self.s = C.c_void_p(lib().jo_session_create(model.m, C.c_int64(max_page_bytes)))

    def page_info(self):
        out = (C.c_int * 4)()
        lib().jo_session_page_info(self.s, out)
        return tuple(out)

    def forward(self, tokens, start_pos, x=None):
        """Rows of token ids (or an [B,E] activation for non-first layer shards) -> [B,E]."""
        E = self.model.cfg["embedding_length"]
        if x is None:
            tokens = np.ascontiguousarray(tokens, dtype=np.int32)
            B = tokens.size
            x = np.empty((B, E), dtype=np.float32)
            lib().jo_forward(self.s, _p(tokens), _p(x), B, start_pos)
        else:
            x = np.ascontiguousarray(x, dtype=np.float32).copy()
            lib().jo_forward(self.s, None, _p(x), x.shape[0], start_pos)
        return x

    # tensor-parallel halves of one layer (partial [B,E] results, no residual): the caller sums over shards
    def tp_attn(self, layer, x, start_pos):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        assert lib().jo_tp_attn(self.s, layer, _p(x), x.shape[0], start_pos, _p(out)) == 0
        return out

    def tp_ffn(self, layer, att_res):
        att_res = np.ascontiguousarray(att_res, dtype=np.float32)
        out = np.empty_like(att_res)
        assert lib().jo_tp_ffn(self.s, layer, _p(att_res), att_res.shape[0], _p(out)) == 0
        return out

    def set_tap_layer(self, layer):
        lib().jo_session_set_tap_layer(self.s, layer)

    def tap(self, name, n):
        out = np.empty(n, dtype=np.float32)
        got = lib().jo_session_get_tap(self.s, TAPS.index(name), _p(out), n)
        assert got == n, (name, got, n)
        return out

    def generate(self, prompt, n_gen, temperature=0.0):
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        out = np.empty(n_gen, dtype=np.int32)
        logits = np.empty(self.model.cfg["vocab_size"], dtype=np.float32)
        times = (C.c_double * 2)()
        n = lib().jo_generate(self.s, _p(prompt), prompt.size, n_gen, C.c_float(temperature), _p(out), _p(logits),
                              times)
        return out[:n], logits, (times[0], times[1])

    def __del__(self):
        try:
            lib().jo_session_destroy(self.s)
        except Exception:
            pass
