"""The reference's host "as is" above the Tier-1 C ABI: ctypes face of libjlamahost.so (csrc/host_mirror.cpp).

``HostAsIsModel`` = AbstractModel + TransformerBlock + CausalSelfAttention + MLPBlock of the reference restated in C++
(no JDK here), calling the provider entry points (`jh_gemm_*`, `jh_quantize_*`, `jh_accumulate_f32`, ...) exactly
where the Java host calls ``TensorOperations`` and doing in host code what Java does in host code (RMSNorm, RoPE,
softmax, SiLU, KV copy, argmax).  It exists to (a) prove the drop-in boundary end to end -- greedy ids and logits bit
for bit against the oracle with ``JH_STRICT_ORDER=1`` -- and (b) MEASURE what keeping the host unchanged costs
(``bench.py``: ``tier1_host_tokens_per_s``), which until round 6 was an estimate.  Nothing here is on the Tier-2 path.
"""
import ctypes as C

import numpy as np

from . import _native as N


class HostAsIsModel:
    def __init__(self, cfg: dict, weights: dict, elementwise_on_device=False, threads=1, layer_range=None, max_page_bytes=0, device=0):
        N.init(device)                       # raises without a GPU: the provider's constructor would throw and Jlama would pick another provider
        self._L = N.host_lib()
        L = cfg["n_layers"]
        ls, le = layer_range if layer_range else (0, L)
        self.cfg = dict(cfg)
        self.c = N.Config(cfg["embedding_length"], cfg["hidden_length"], cfg["n_heads"], cfg["n_kv_heads"], cfg["head_size"], L,
                          cfg["vocab_size"], cfg["context_length"], cfg["weight_dtype"], ls, le, cfg["rms_eps"], cfg["rope_theta"],
                          cfg.get("rope_scaling", 1.0))
        self.h = C.c_void_p()
        self._check(self._L.jhost_create(C.byref(self.c), 1 if elementwise_on_device else 0, int(threads), int(max_page_bytes), C.byref(self.h)))
        self._keep = []                      # the host owns its weights (Java: mmap'd safetensors); the C side borrows the pointers
        for (layer, slot), w in weights.items():
            if layer >= 0 and not (ls <= layer < le):
                continue
            d = np.ascontiguousarray(w["data"])
            s = np.ascontiguousarray(w["scales"]) if w.get("scales") is not None else None
            self._keep.append((d, s))
            rows, cols = w["shape"]
            self._check(self._L.jhost_set_weight(self.h, layer, slot, w["dtype"], N.ptr(d), N.ptr(s), rows, cols))

    def _check(self, rc):
        if rc is not None and rc < 0:
            msg = self._L.jhost_last_error().decode(errors="replace")
            if rc == N.JH_ERR_UNSUPPORTED:
                raise N.UnsupportedOperation(rc, msg)
            raise N.JhError(rc, msg)
        return rc

    def page_info(self):
        out = (C.c_int32 * 4)()
        self._check(self._L.jhost_page_info(self.h, out))
        return tuple(out)

    def forward(self, tokens, start_pos=0):
        """AbstractModel.batchForward of one chunk (<= 256 rows): returns the [n, E] output rows."""
        tokens = np.ascontiguousarray(tokens, dtype=np.int32).reshape(-1)
        x = np.empty((tokens.size, self.cfg["embedding_length"]), dtype=np.float32)
        self._check(self._L.jhost_forward(self.h, N.ptr(tokens), N.ptr(x), tokens.size, int(start_pos)))
        return x

    def sample(self, row):
        """AbstractModel.sample at temperature 0: (token id, logits)."""
        row = np.ascontiguousarray(row, dtype=np.float32).reshape(-1)
        logits = np.empty(self.cfg["vocab_size"], dtype=np.float32)
        tok = C.c_int32()
        self._check(self._L.jhost_sample(self.h, N.ptr(row), N.ptr(logits), C.byref(tok)))
        return tok.value, logits

    def generate(self, prompt, n_gen):
        """AbstractModel.generate at temperature 0.  Returns dict(tokens, logits of the last step, prompt_ms, decode_ms,
        provider_ms = wall time inside the C ABI, provider_calls)."""
        prompt = np.ascontiguousarray(prompt, dtype=np.int32).reshape(-1)
        out = np.empty(n_gen, dtype=np.int32)
        logits = np.empty(self.cfg["vocab_size"], dtype=np.float32)
        t = (C.c_double * 4)()
        n = self._check(self._L.jhost_generate(self.h, N.ptr(prompt), prompt.size, int(n_gen), N.ptr(out), N.ptr(logits), t))
        return {"tokens": out[:n], "logits": logits, "prompt_ms": t[0], "decode_ms": t[1], "provider_ms": t[2], "provider_calls": int(t[3])}

    def close(self):
        if self.h:
            self._L.jhost_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
