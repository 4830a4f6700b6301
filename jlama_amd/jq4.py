"""JQ4 host formats: the Q4 weight layout, the Q8 activation layout, BF16 -- as numpy arrays.

Restates (host side, vectorised) what the reference's storage classes define, so weights can be produced and
inspected without a JVM:

* Q4: jlama-core/.../tensor/Q4ByteBufferTensor.java:36-38,66-120,179-202 -- block = 32 consecutive elements of a
  row; 16 bytes per block, byte j: low nibble = element j, high nibble = element j+16; value = (nibble-8)*scale;
  scale is F32 in a separate row-major [rows, cols/32] tensor (the `.qb` sibling on disk, Weights.java:159-171).
* Q8: jlama-core/.../tensor/Q8ByteBufferTensor.java:39-40,140-146 -- int8 values + F32 scale per 32.
* BF16: jlama-core/.../math/FloatConversions.java:31-60.

The device-side quantizers (jh_quantize_q8 etc.) are the product; these helpers only build inputs.
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np

from ._native import DT_BF16, DT_F32, DT_I8, DT_Q4

BLOCK = 32
_FLOAT_MIN_VALUE = np.float32(1.4e-45)  # Java Float.MIN_VALUE


def quantize_q4(x):
    """Q4ByteBufferTensor.processBlock (Q4ByteBufferTensor.java:66-106), vectorised.

    x: float32 [rows, cols] -> (nibbles uint8 [rows, cols/2], scales float32 [rows, cols/32])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, cols = x.shape
    assert cols % BLOCK == 0
    nb = cols // BLOCK
    xb = x.reshape(rows, nb, BLOCK)
    ax = np.abs(xb)
    idx = ax.argmax(axis=2)  # first strictly-greater max wins == first occurrence of the max
    amax = np.take_along_axis(ax, idx[..., None], axis=2)[..., 0]
    signed = np.take_along_axis(xb, idx[..., None], axis=2)[..., 0]
    mx = np.where(amax > _FLOAT_MIN_VALUE, signed, _FLOAT_MIN_VALUE).astype(np.float32)
    with np.errstate(divide="ignore", under="ignore"):
        scale = (mx / np.float32(-8.0)).astype(np.float32)
        iscale = np.where(scale != 0, np.float32(1.0) / scale, np.float32(0.0)).astype(np.float32)
    f = (xb * iscale[..., None]).astype(np.float32) + np.float32(8.5)
    q = np.minimum(f.astype(np.int32).astype(np.int8), 15).astype(np.uint8)  # (byte) Math.min(15, (byte)(f + 8.5f))
    nib = (q[..., :16] | (q[..., 16:] << 4)).astype(np.uint8)
    return nib.reshape(rows, cols // 2), scale.reshape(rows, nb)


def dequantize_q4(nib, scales):
    """Q4ByteBufferTensor.get (:179-197): (nibble - 8) * scale."""
    rows, half = nib.shape
    nb = half // 16
    b = nib.reshape(rows, nb, 16)
    lo = (b & 0x0F).astype(np.int32) - 8
    hi = ((b >> 4) & 0x0F).astype(np.int32) - 8
    v = np.concatenate([lo, hi], axis=2).astype(np.float32) * scales.reshape(rows, nb, 1)
    return v.reshape(rows, half * 2).astype(np.float32)


def f32_to_bf16(x):
    """FloatConversions.float32ToBFloat16 (:35-60): round-to-nearest-even, NaN -> 0x7fc0."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    rounded = ((u + (0x7FFF + ((u >> 16) & 1))) >> 16).astype(np.uint16)
    nan = ((u & 0x7F800000) == 0x7F800000) & ((u & 0x7FFFFF) != 0)
    inf = ((u & 0x7F800000) == 0x7F800000) & ((u & 0x7FFFFF) == 0)
    out = np.where(nan, np.uint16(0x7FC0), rounded)
    out = np.where(inf, (u >> 16).astype(np.uint16), out)
    return out.astype(np.uint16)


def bf16_to_f32(h):
    return (np.ascontiguousarray(h, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


@dataclass
class Tensor:
    """Minimal stand-in for AbstractTensor (jlama-core/.../tensor/AbstractTensor.java:45-323): a dense 2-D
    row-major tensor with a dtype tag, plus the `blockF` scale tensor for I8/Q4."""
    dtype: int
    data: np.ndarray            # F32: float32 [r,c]; BF16: uint16 [r,c]; I8: int8 [r,c]; Q4: uint8 [r,c/2]
    scales: Optional[np.ndarray] = None  # blockF, float32 [r, c/32]
    reg_ids: Optional[tuple] = None      # ids from jh_register_tensor (data, scales)

    @property
    def rows(self):
        return self.data.shape[0]

    @property
    def cols(self):
        return self.data.shape[1] * (2 if self.dtype == DT_Q4 else 1)

    @property
    def stride(self):  # AbstractTensor.getStride (:215-217), elements between rows
        return self.cols

    @staticmethod
    def f32(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return Tensor(DT_F32, a.reshape(1, -1) if a.ndim == 1 else a)

    @staticmethod
    def bf16(a):
        a = np.ascontiguousarray(a)
        if a.dtype != np.uint16:
            a = f32_to_bf16(a)
        return Tensor(DT_BF16, a.reshape(1, -1) if a.ndim == 1 else a)

    @staticmethod
    def q4(a):
        nib, sc = quantize_q4(a)
        return Tensor(DT_Q4, nib, sc)

    @staticmethod
    def i8(q, d):
        return Tensor(DT_I8, np.ascontiguousarray(q, dtype=np.int8), np.ascontiguousarray(d, dtype=np.float32))

    @staticmethod
    def zeros(rows, cols):
        return Tensor(DT_F32, np.zeros((rows, cols), dtype=np.float32))
