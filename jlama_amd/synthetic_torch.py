"""Full-size synthetic JQ4 weights generated ON the GPU (torch is plumbing here: RNG + device memory).

Same recipe as synthetic.make_weights but drawn with torch's device generator, because 7.5 G normals
take minutes on the host.  quantize_q4 mirrors jq4.quantize_q4 / Q4ByteBufferTensor.processBlock op for op in
float32, so bytes produced here can be copied to the host and fed to the oracle unchanged.
"""
import numpy as np
import torch

from ._native import (DT_BF16, DT_Q4, W_DOWN, W_EMBED, W_FINALNORM, W_GATE, W_K, W_LMHEAD, W_NORM1, W_NORM2, W_O, W_Q,
                      W_UP, W_V)
from .synthetic import BASE_SEED, layer_shapes

_MINV = 1.4e-45


def quantize_q4(x: torch.Tensor):
    rows, cols = x.shape
    nb = cols // 32
    xb = x.view(rows, nb, 32)
    ax = xb.abs()
    amax, idx = ax.max(dim=2)  # first occurrence of the max
    signed = torch.gather(xb, 2, idx.unsqueeze(-1)).squeeze(-1)
    mx = torch.where(amax > _MINV, signed, torch.full_like(signed, _MINV))
    scale = mx / -8.0
    iscale = torch.where(scale != 0, 1.0 / scale, torch.zeros_like(scale))
    f = xb * iscale.unsqueeze(-1) + 8.5
    q = torch.clamp(f.to(torch.int32), max=15).to(torch.uint8)
    nib = q[..., :16] | (q[..., 16:] << 4)
    return nib.reshape(rows, cols // 2).contiguous(), scale.reshape(rows, nb).contiguous()


def _q4(gen, rows, cols, sigma, device, chunk_rows=16384):
    nibs, scs = [], []
    for r0 in range(0, rows, chunk_rows):
        r = min(chunk_rows, rows - r0)
        x = torch.randn((r, cols), generator=gen, device=device, dtype=torch.float32) * sigma
        n, s = quantize_q4(x)
        nibs.append(n); scs.append(s)
    return {"dtype": DT_Q4, "data": torch.cat(nibs), "scales": torch.cat(scs), "shape": (rows, cols)}


def _bf16(gen, rows, cols, sigma, device, chunk_rows=16384):
    parts = []
    for r0 in range(0, rows, chunk_rows):
        r = min(chunk_rows, rows - r0)
        x = torch.randn((r, cols), generator=gen, device=device, dtype=torch.float32) * sigma
        parts.append(x.to(torch.bfloat16).view(torch.int16))
    return {"dtype": DT_BF16, "data": torch.cat(parts).contiguous(), "scales": None, "shape": (rows, cols)}


def _norm(gen, E, device):
    w = 1.0 + torch.randn(E, generator=gen, device=device, dtype=torch.float32) * 0.01
    h = w.to(torch.bfloat16).view(torch.int16).view(1, E).contiguous()  # RNE, same as FloatConversions.float32ToBFloat16
    return {"dtype": DT_BF16, "data": h, "scales": None, "shape": (1, E)}


def make_weights(cfg, seed=0, layers=None, device="cuda", need_embed=True, need_head=True):
    E, V, L = cfg["embedding_length"], cfg["vocab_size"], cfg["n_layers"]
    ls, le = layers if layers else (0, L)
    gen = torch.Generator(device=device)
    out = {}
    _q4 = globals()["_q4"] if cfg["weight_dtype"] == DT_Q4 else _bf16

    def reseed(i):
        gen.manual_seed(BASE_SEED + seed * 100003 + i)
        return gen

    idx = 0
    if need_embed:
        out[(-1, W_EMBED)] = _q4(reseed(idx), V, E, 0.02, device)
    idx += 1
    shapes = layer_shapes(cfg)
    for li in range(L):
        for slot in (W_Q, W_K, W_V, W_O, W_GATE, W_UP, W_DOWN):
            r, c = shapes[slot]
            if ls <= li < le:
                out[(li, slot)] = _q4(reseed(idx), r, c, 1.0 / float(np.sqrt(c)), device)
            idx += 1
        for slot in (W_NORM1, W_NORM2):
            if ls <= li < le:
                out[(li, slot)] = _norm(reseed(idx), E, device)
            idx += 1
    if need_head:
        out[(-1, W_FINALNORM)] = _norm(reseed(idx), E, device)
    idx += 1
    if need_head and not cfg.get("tied"):
        out[(-1, W_LMHEAD)] = _q4(reseed(idx), V, E, 0.02, device)
    return out


def to_host(weights):
    """Copy a device weight dict to numpy (for the CPU baseline / oracle), same keys."""
    out = {}
    for k, w in weights.items():
        d = w["data"].cpu().numpy()
        if w["dtype"] == DT_BF16:
            d = d.view(np.uint16)
        out[k] = {"dtype": w["dtype"], "data": np.ascontiguousarray(d),
                  "scales": None if w["scales"] is None else np.ascontiguousarray(w["scales"].cpu().numpy()),
                  "shape": w["shape"]}
    return out
