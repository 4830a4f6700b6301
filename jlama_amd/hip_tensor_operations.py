"""HipTensorOperations -- Python mirror of the Java provider a Jlama maintainer would add.

Same method names, argument meaning and error behaviour as ``TensorOperations``
(jlama-core/.../tensor/operations/TensorOperations.java:25-161); the marshaling of ``batchDotProduct`` copies
jlama-native/.../NativeSimdTensorOperations.java:84-232 (dense tensors: no sparse windows).  Unsupported dtype
pairs raise :class:`UnsupportedOperation` where Panama throws UnsupportedOperationException
(PanamaTensorOperations.java:125-142).  The constructor raises when no GPU is usable, which is what lets
``TensorOperationsProvider`` fall through to the next provider (TensorOperationsProvider.java:50-87).
"""
import ctypes as C

import numpy as np

from . import _native as N
from .jq4 import Tensor
from ._native import DT_BF16, DT_F32, DT_I8, DT_Q4


class HipTensorOperations:
    def __init__(self, device=0):
        self.info = N.init(device)  # raises JhError(JH_ERR_NO_DEVICE) without a GPU
        self._lib = N.lib()

    # ---- provider facts ------------------------------------------------------------------------------
    def name(self):
        return self._lib.jh_name().decode()

    def parallelSplitSize(self):
        return self._lib.jh_parallel_split_size()

    def preferredWorkingQuantizedType(self):
        return self._lib.jh_preferred_working_qtype()

    def registerModelTensor(self, t: Tensor):
        """Upload a weight once (NativeGPUTensorOperations.registerModelTensor :104-151: Q4 registers nibbles AND
        blockF).  On OOM the tensor simply stays unregistered and later calls ship it per call."""
        if t.reg_ids is not None:
            return
        try:
            d = N.check(self._lib.jh_register_tensor(N.ptr(t.data), t.data.nbytes))
            s = N.check(self._lib.jh_register_tensor(N.ptr(t.scales), t.scales.nbytes)) if t.scales is not None else -1
            t.reg_ids = (d, s)
        except N.JhError as e:
            if e.code != N.JH_ERR_OOM:
                raise

    # ---- batchDotProduct (TensorOperations.java:62-72) --------------------------------------------------
    def batchDotProduct(self, result: Tensor, a: Tensor, b: Tensor, aColumnOffset, bColumnOffset, columnLength,
                        rRowOffset=0, bRowOffset=0, rowChunkSize=None):
        if rowChunkSize is None:
            rowChunkSize = b.rows
        M, Nn, K = a.rows, rowChunkSize, columnLength
        aOffset = aColumnOffset
        bOffset = bColumnOffset
        rOffset = -rRowOffset  # NativeSimdTensorOperations.java:105 (dense: no sparse offsets)
        bid, bfid = b.reg_ids if b.reg_ids else (-1, -1)
        L = self._lib
        # BF16 result tensors exist for the BF16-weight GEMMs only (NativeSimdTensorOperations.java:113-131,156-174:
        # cr = result when result.dType()==BF16, else NULL)
        bf16_out = result.dtype == DT_BF16 and b.dtype == DT_BF16 and a.dtype in (DT_BF16, DT_F32)
        if result.dtype != DT_F32 and not bf16_out:
            raise N.UnsupportedOperation(N.JH_ERR_UNSUPPORTED, "result must be F32 (or BF16 for BF16 weights)")
        r = None if bf16_out else N.ptr(result.data)
        cr = N.ptr(result.data) if bf16_out else None
        if a.dtype == DT_I8 and b.dtype == DT_Q4:
            rc = L.jh_gemm_q8_q4(bid, bfid, N.ptr(a.scales), N.ptr(a.data), aOffset, N.ptr(b.scales), N.ptr(b.data),
                                 bOffset // 2, r, rOffset, M, bRowOffset, Nn, K, a.stride, a.scales.shape[1],
                                 b.stride // 2, b.scales.shape[1], result.stride)
        elif a.dtype == DT_F32 and b.dtype == DT_Q4:
            rc = L.jh_gemm_f32_q4(bid, bfid, N.ptr(a.data), aOffset, N.ptr(b.scales), N.ptr(b.data), bOffset // 2, r,
                                  rOffset, M, bRowOffset, Nn, K, a.stride, b.stride // 2, b.scales.shape[1],
                                  result.stride)
        elif a.dtype == DT_F32 and b.dtype == DT_F32:
            rc = L.jh_gemm_f32(bid, N.ptr(a.data), aOffset, N.ptr(b.data), bOffset, r, rOffset, M, bRowOffset, Nn, K,
                               a.stride, b.stride, result.stride)
        elif a.dtype == DT_BF16 and b.dtype == DT_BF16:
            rc = L.jh_gemm_bf16(bid, N.ptr(a.data), aOffset, N.ptr(b.data), bOffset, cr, r, rOffset, M, bRowOffset, Nn, K,
                                a.stride, b.stride, result.stride)
        elif a.dtype == DT_F32 and b.dtype == DT_BF16:
            rc = L.jh_gemm_f32_bf16(bid, N.ptr(a.data), aOffset, N.ptr(b.data), bOffset, cr, r, rOffset, M, bRowOffset, Nn,
                                    K, a.stride, b.stride, result.stride)
        else:
            raise N.UnsupportedOperation(N.JH_ERR_UNSUPPORTED, f"dtype pair {a.dtype} x {b.dtype}")
        N.check(rc)

    def dotProductChunk(self, result, a, b, columnOffset, columnLimit, rowOffset, rowChunkSize):
        # TensorOperations.java:74-84
        self.batchDotProduct(result, a, b, columnOffset, columnOffset, columnLimit, 0, rowOffset, rowChunkSize)

    def dotProductBatchChunk(self, results, a, bs, offset, limit, chunkStart, chunkSize):
        """TensorOperations.java:86-99 through the `_batch` entry points: arrays of weight / result pointers filled like
        MemorySegmentSupport.setupBatch (jlama-native/.../NativeSimdTensorOperations.java:236-334)."""
        assert len(results) == len(bs) and len(bs) > 0
        nb = len(bs)
        dt_b = bs[0].dtype
        if any(b.dtype != dt_b or b.stride != bs[0].stride for b in bs) or any(r.stride != results[0].stride for r in results):
            for r, b in zip(results, bs):   # heterogeneous batch: one call each
                self.dotProductChunk(r, a, b, offset, limit, chunkStart, chunkSize)
            return
        L = self._lib
        M, K = a.rows, limit
        arr_p = lambda xs: (C.c_void_p * nb)(*[x.ctypes.data for x in xs])
        arr_l = lambda xs: (C.c_int64 * nb)(*xs)
        ids = arr_l([b.reg_ids[0] if b.reg_ids else -1 for b in bs])
        b_ptr = arr_p([b.data for b in bs])
        bf16_out = all(r.dtype == DT_BF16 for r in results) and dt_b == DT_BF16
        if not bf16_out and any(r.dtype != DT_F32 for r in results):
            raise N.UnsupportedOperation(N.JH_ERR_UNSUPPORTED, "result must be F32 (or BF16 for BF16 weights)")
        r_ptr = None if bf16_out else arr_p([r.data for r in results])
        cr_ptr = arr_p([r.data for r in results]) if bf16_out else None
        ldc = results[0].stride
        if a.dtype == DT_I8 and dt_b == DT_Q4:
            sid = arr_l([b.reg_ids[1] if b.reg_ids else -1 for b in bs])
            rc = L.jh_gemm_q8_q4_batch(nb, ids, sid, N.ptr(a.scales), N.ptr(a.data), offset, arr_p([b.scales for b in bs]), b_ptr,
                                       offset // 2, r_ptr, 0, M, chunkStart, chunkSize, K, a.stride, a.scales.shape[1],
                                       bs[0].stride // 2, bs[0].scales.shape[1], ldc)
        elif a.dtype == DT_F32 and dt_b == DT_Q4:
            sid = arr_l([b.reg_ids[1] if b.reg_ids else -1 for b in bs])
            rc = L.jh_gemm_f32_q4_batch(nb, ids, sid, N.ptr(a.data), offset, arr_p([b.scales for b in bs]), b_ptr, offset // 2,
                                        r_ptr, 0, M, chunkStart, chunkSize, K, a.stride, bs[0].stride // 2,
                                        bs[0].scales.shape[1], ldc)
        elif a.dtype == DT_F32 and dt_b == DT_F32:
            rc = L.jh_gemm_f32_batch(nb, ids, N.ptr(a.data), offset, b_ptr, offset, r_ptr, 0, M, chunkStart, chunkSize, K,
                                     a.stride, bs[0].stride, ldc)
        elif a.dtype == DT_BF16 and dt_b == DT_BF16:
            rc = L.jh_gemm_bf16_batch(nb, ids, N.ptr(a.data), offset, b_ptr, offset, cr_ptr, r_ptr, 0, M, chunkStart, chunkSize,
                                      K, a.stride, bs[0].stride, ldc)
        elif a.dtype == DT_F32 and dt_b == DT_BF16:
            rc = L.jh_gemm_f32_bf16_batch(nb, ids, N.ptr(a.data), offset, b_ptr, offset, cr_ptr, r_ptr, 0, M, chunkStart,
                                          chunkSize, K, a.stride, bs[0].stride, ldc)
        else:
            raise N.UnsupportedOperation(N.JH_ERR_UNSUPPORTED, f"dtype pair {a.dtype} x {dt_b}")
        N.check(rc)

    def dotProduct(self, a, b, aoffset=0, boffset=0, limit=None):
        # TensorOperations.java:41-49
        limit = a.cols if limit is None else limit
        r = Tensor.zeros(1, 1)
        self.batchDotProduct(r, a, b, aoffset, boffset, limit, 0, 0, 1)
        return float(r.data[0, 0])

    # ---- element-wise ------------------------------------------------------------------------------------
    def accumulate(self, a: Tensor, b: Tensor, offset, length):
        # PanamaTensorOperations.java:2150-2218: per row of a; b broadcast when it has one row
        for ai in range(a.rows):
            bi = ai if b.rows > 1 else 0
            if b.dtype == DT_F32:
                N.check(self._lib.jh_accumulate_f32(N.ptr(a.data[ai]), N.ptr(b.data[bi]), offset, length))
            elif b.dtype == DT_Q4:
                N.check(self._lib.jh_accumulate_f32_q4(N.ptr(a.data[ai]), N.ptr(b.data[bi]), N.ptr(b.scales[bi]), offset,
                                                       length))
            else:
                raise N.UnsupportedOperation(N.JH_ERR_UNSUPPORTED, "accumulate dtype")

    def maccumulate(self, a: Tensor, b: Tensor, offset, length):
        if a.dtype != DT_F32 or b.dtype != DT_F32:
            raise N.UnsupportedOperation(N.JH_ERR_UNSUPPORTED, "maccumulate dtype")
        for ai in range(a.rows):
            bi = ai if b.rows > 1 else 0
            N.check(self._lib.jh_maccumulate_f32(N.ptr(a.data[ai]), N.ptr(b.data[bi]), offset, length))

    def scale(self, factor, a: Tensor, offset, length):
        if a.dtype != DT_F32:
            raise N.UnsupportedOperation(N.JH_ERR_UNSUPPORTED, "scale dtype")
        for ai in range(a.rows):
            N.check(self._lib.jh_scale_f32(float(factor), N.ptr(a.data[ai]), offset, length))

    def saxpy(self, alpha, x: Tensor, y: Tensor, xoffset, yoffset, limit, aOffset=None, xRowOffset=None,
              batchSize=None):
        if x.dtype != DT_F32 or y.dtype != DT_F32:
            raise N.UnsupportedOperation(N.JH_ERR_UNSUPPORTED, "saxpy dtype")
        if isinstance(alpha, Tensor):  # batched form, TensorOperations.java:119-135
            N.check(self._lib.jh_saxpy_batch_f32(N.ptr(alpha.data), N.ptr(x.data), x.stride, N.ptr(y.data), xoffset,
                                                 yoffset, limit, aOffset, xRowOffset, batchSize))
        else:
            N.check(self._lib.jh_saxpy_f32(float(alpha), N.ptr(x.data), N.ptr(y.data), xoffset, yoffset, limit))

    def quantize(self, t: Tensor, qtype, offset, length):
        # PanamaTensorOperations.java:1595-1622
        if t.dtype != DT_F32:
            raise N.UnsupportedOperation(N.JH_ERR_UNSUPPORTED, "quantize source dtype")
        if qtype == DT_I8:
            q = np.zeros((t.rows, t.cols), dtype=np.int8)
            d = np.zeros((t.rows, t.cols // 32), dtype=np.float32)
            N.check(self._lib.jh_quantize_q8(N.ptr(t.data), t.rows, t.stride, offset, length, N.ptr(q), t.cols, N.ptr(d),
                                             t.cols // 32))
            return Tensor.i8(q, d)
        if qtype == DT_BF16:
            out = np.zeros((t.rows, t.cols), dtype=np.uint16)
            N.check(self._lib.jh_quantize_bf16(N.ptr(t.data), t.data.size, N.ptr(out)))
            return Tensor(DT_BF16, out)
        raise N.UnsupportedOperation(N.JH_ERR_UNSUPPORTED, f"F32 => {qtype}")

    # ---- ops outside the Java interface but on the path ------------------------------------------------
    def rmsnorm(self, x, w, eps, weight_adj=0.0):
        x = np.ascontiguousarray(x, dtype=np.float32)
        w = np.ascontiguousarray(w, dtype=np.float32)
        out = np.empty_like(x)
        N.check(self._lib.jh_rmsnorm_f32(N.ptr(x), N.ptr(w), weight_adj, x.size, eps, N.ptr(out)))
        return out

    def softmax(self, x, offset, length):
        x = np.ascontiguousarray(x, dtype=np.float32).copy()
        N.check(self._lib.jh_softmax_f32(N.ptr(x), offset, length))
        return x

    def silu_mul(self, g, u):
        g = np.ascontiguousarray(g, dtype=np.float32).copy()
        u = np.ascontiguousarray(u, dtype=np.float32)
        N.check(self._lib.jh_silu_mul_f32(N.ptr(g), N.ptr(u), g.size))
        return g

    def layer_norm(self, x, w, b, eps, offset=0, length=None, divisor=None):
        """LayerNorm.forward (core/model/LayerNorm.java:41-67) per row of x [rows, E]; GPT-2 family."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        x = x.reshape(1, -1) if x.ndim == 1 else x
        rows, ld = x.shape
        length = ld - offset if length is None else length
        out = x.copy()
        N.check(self._lib.jh_layernorm_f32(N.ptr(x), N.ptr(np.ascontiguousarray(w, dtype=np.float32)),
                                           N.ptr(np.ascontiguousarray(b, dtype=np.float32)), rows, ld, offset, length,
                                           ld if divisor is None else divisor, eps, N.ptr(out)))
        return out

    def gelu(self, x):
        """ActivationFunction.eval(GELU) (core/math/ActivationFunction.java:32-34)."""
        x = np.ascontiguousarray(x, dtype=np.float32).copy()
        N.check(self._lib.jh_gelu_f32(N.ptr(x), x.size))
        return x

    def rope_table(self, dim, end, theta, scaling=1.0):
        out = np.empty((end * (dim // 2), 2), dtype=np.float32)
        N.check(self._lib.jh_rope_table(dim, end, theta, scaling, N.ptr(out)))
        return out

    def rope_apply(self, q, k, rope, position, n_heads, n_kv_heads, head_size):
        q = np.ascontiguousarray(q, dtype=np.float32).copy()
        k = np.ascontiguousarray(k, dtype=np.float32).copy()
        N.check(self._lib.jh_rope_apply_f32(N.ptr(q), N.ptr(k), N.ptr(rope), rope.shape[0] // (head_size // 2), position,
                                            n_heads, n_kv_heads, head_size))
        return q, k
