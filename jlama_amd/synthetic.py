"""Synthetic JQ4 Llama-family weights (no network => no checkpoints): the recipe of SURVEY.md 8(d).

For every weight tensor draw F32 N(0, sigma^2) (sigma = 0.02 for embeddings / LM head, 1/sqrt(K) for
projections), norm weights = 1 + N(0, 0.01^2) stored as BF16, then quantize with the restated
Q4ByteBufferTensor.processBlock (jq4.quantize_q4).  Everything except 1-row tensors / names containing "norm" is
Q4 (jlama-cli/.../QuantizeCommand.java:36-38, AbstractTensor.java:284).  Seeds: default_rng(seed + tensorIndex).
"""
import numpy as np

from . import jq4
from ._native import (DT_BF16, DT_Q4, W_DOWN, W_EMBED, W_FINALNORM, W_GATE, W_K, W_LMHEAD, W_NORM1, W_NORM2, W_O,
                      W_Q, W_UP, W_V)

BASE_SEED = 0x4A4C414D41  # "JLAMA"

LLAMA3_8B = dict(embedding_length=4096, hidden_length=14336, n_heads=32, n_kv_heads=8, head_size=128, n_layers=32,
                 vocab_size=128256, context_length=8192, weight_dtype=DT_Q4, rms_eps=1e-5, rope_theta=500000.0,
                 rope_scaling=1.0, bos_token=128000)
LLAMA32_1B = dict(embedding_length=2048, hidden_length=8192, n_heads=32, n_kv_heads=8, head_size=64, n_layers=16,
                  vocab_size=128256, context_length=131072, weight_dtype=DT_Q4, rms_eps=1e-5, rope_theta=500000.0,
                  rope_scaling=1.0, bos_token=128000, tied=True)
LLAMA3_70B = dict(embedding_length=8192, hidden_length=28672, n_heads=64, n_kv_heads=8, head_size=128, n_layers=80,
                  vocab_size=128256, context_length=8192, weight_dtype=DT_Q4, rms_eps=1e-5, rope_theta=500000.0,
                  rope_scaling=1.0, bos_token=128000)
MISTRAL_7B = dict(embedding_length=4096, hidden_length=14336, n_heads=32, n_kv_heads=8, head_size=128, n_layers=32,
                  vocab_size=32768, context_length=32768, weight_dtype=DT_BF16, rms_eps=1e-5, rope_theta=1000000.0,
                  rope_scaling=1.0, bos_token=1)
TINY = dict(embedding_length=256, hidden_length=512, n_heads=4, n_kv_heads=2, head_size=64, n_layers=2,
            vocab_size=512, context_length=256, weight_dtype=DT_Q4, rms_eps=1e-5, rope_theta=10000.0,
            rope_scaling=1.0, bos_token=1)
SMALL = dict(embedding_length=512, hidden_length=1024, n_heads=8, n_kv_heads=2, head_size=128, n_layers=3,
             vocab_size=1024, context_length=512, weight_dtype=DT_Q4, rms_eps=1e-5, rope_theta=500000.0,
             rope_scaling=1.0, bos_token=1)


def layer_shapes(cfg):
    E, H = cfg["embedding_length"], cfg["hidden_length"]
    A, KV = cfg["n_heads"] * cfg["head_size"], cfg["n_kv_heads"] * cfg["head_size"]
    return {W_Q: (A, E), W_K: (KV, E), W_V: (KV, E), W_O: (E, A), W_GATE: (H, E), W_UP: (H, E), W_DOWN: (E, H)}


def _q4(rng, rows, cols, sigma, quantize=None):
    x = rng.standard_normal((rows, cols), dtype=np.float32) * np.float32(sigma)
    nib, sc = (quantize or jq4.quantize_q4)(x)
    return {"dtype": DT_Q4, "data": nib, "scales": sc, "shape": (rows, cols)}


def _bf16(rng, rows, cols, sigma, quantize=None):
    x = rng.standard_normal((rows, cols), dtype=np.float32) * np.float32(sigma)
    return {"dtype": DT_BF16, "data": jq4.f32_to_bf16(x), "scales": None, "shape": (rows, cols)}


def _norm(rng, E):
    w = (1.0 + rng.standard_normal(E, dtype=np.float32) * np.float32(0.01)).astype(np.float32)
    return {"dtype": DT_BF16, "data": jq4.f32_to_bf16(w).reshape(1, E), "scales": None, "shape": (1, E)}


def make_weights(cfg, seed=0, layers=None, quantize=None):
    """dict {(layer or -1, slot): {dtype, data, scales, shape}} for the given layer range (default: all).
    quantize: optional F32 [rows, cols] -> (nibbles, scales) callable replacing jq4.quantize_q4 (tests pass a compiled,
    bit-identical quantizer to build multi-layer real-shape models in seconds)."""
    E, V, L = cfg["embedding_length"], cfg["vocab_size"], cfg["n_layers"]
    ls, le = layers if layers else (0, L)
    out = {}
    idx = 0

    def rng_for(i):
        return np.random.default_rng(BASE_SEED + seed * 100003 + i)

    _mk = globals()["_q4"] if cfg["weight_dtype"] == DT_Q4 else _bf16   # BF16 models: RNE of the F32 draw

    def _q4(rng, rows, cols, sigma):
        return _mk(rng, rows, cols, sigma, quantize)

    out[(-1, W_EMBED)] = _q4(rng_for(idx), V, E, 0.02); idx += 1
    for li in range(L):
        shapes = layer_shapes(cfg)
        for slot in (W_Q, W_K, W_V, W_O, W_GATE, W_UP, W_DOWN):
            r, c = shapes[slot]
            if ls <= li < le:
                out[(li, slot)] = _q4(rng_for(idx), r, c, 1.0 / np.sqrt(c))
            idx += 1
        for slot in (W_NORM1, W_NORM2):
            if ls <= li < le:
                out[(li, slot)] = _norm(rng_for(idx), E)
            idx += 1
    out[(-1, W_FINALNORM)] = _norm(rng_for(idx), E); idx += 1
    if not cfg.get("tied"):
        out[(-1, W_LMHEAD)] = _q4(rng_for(idx), V, E, 0.02)
    idx += 1
    return out


def prompt_tokens(cfg, n=128, seed=1234):
    """128 token ids uniform in [0,V) with BOS prepended => the 129-row prefill of AbstractModel.java:549-555."""
    rng = np.random.default_rng(seed)
    toks = rng.integers(0, cfg["vocab_size"], size=n, dtype=np.int64).astype(np.int32)
    return np.concatenate([[np.int32(cfg["bos_token"])], toks]).astype(np.int32)


def weight_bytes(cfg):
    """Algorithmic bytes of weights read per decoded token: Q4 = 0.5 B nibble + 4 B scale / 32 = 0.625 B/weight, BF16 = 2 B
    (SURVEY.md 8d); the embedding table is a one-row lookup and excluded, the LM head (or tied table) is read once."""
    E, V, L = cfg["embedding_length"], cfg["vocab_size"], cfg["n_layers"]
    per_layer = sum(r * c for r, c in layer_shapes(cfg).values())
    return int((L * per_layer + V * E) * (0.625 if cfg["weight_dtype"] == DT_Q4 else 2.0))


def kv_bytes_per_position(cfg):
    return 2 * cfg["n_layers"] * cfg["n_kv_heads"] * cfg["head_size"] * 4
