#!/usr/bin/env python3
"""Golden vectors from the REFERENCE ITSELF: jlama-native/src/main/c/simd/vector_simd.c compiled as-is into
oracle/_ref/libjlama_ref_avx512.so (oracle/Makefile), called exactly as NativeSimdTensorOperations.java:96-107,204-223
marshals it (flags = HAS_AVX2|HAS_F16C => the _512 kernels).  Run in the build container (needs /root/reference for the
build and an AVX-512 host); the output, tests/golden/ref_gemm_vectors.npz, is committed so that the GPU tests -- which
run where /root/reference does not exist -- compare libjlamahip.so with outputs of the reference.

M = 1 only: the reference's tiler leaves output corners unwritten for M > 5 (tests/test_oracle.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jlama_amd import jq4  # noqa: E402
from oracle import oracle as O  # noqa: E402

assert O.ref_lib() is not None, "build oracle/_ref first (make -C oracle)"
rng = np.random.default_rng(20240923)
N, K = 192, 1024
w = (rng.standard_normal((N, K), dtype=np.float32) * np.float32(1.0 / 32.0))
nib, sc = jq4.quantize_q4(w)
x = rng.uniform(-1.0, 3.0, size=(1, K)).astype(np.float32)
aq, ad = O.q8_quantize(x)                       # Panama quantizer restatement (pinned separately, test_oracle.py)
out = {"nib": nib, "scales": sc, "x": x, "aq": aq, "ad": ad}
# full-K I8xQ4 and F32xQ4 GEMV
out["q8q4_full"] = O.ref_gemm_q8_q4(aq, ad, nib, sc)
out["f32q4_full"] = O.ref_gemm_f32_q4(x, nib, sc)
# column window K=512 at offset 512, rows [32, 32+64) (dotProductChunk form, TestOperations.java:714-757)
out["q8q4_window"] = O.ref_gemm_q8_q4(aq, ad, nib, sc, aColOff=512, bColOff=512, K=512, bRowOff=32, N=64)
out["f32q4_window"] = O.ref_gemm_f32_q4(x, nib, sc, aColOff=512, bColOff=512, K=512, bRowOff=32, N=64)
# dense F32 (attention-score form: K = head size 128)
kpage = rng.standard_normal((48, 256), dtype=np.float32)
q = rng.standard_normal((1, 256), dtype=np.float32)
out["f32_q"], out["f32_kpage"] = q, kpage
out["f32_scores"] = O.ref_gemm_f32(q, kpage, aColOff=128, bColOff=128, K=128)
# BF16 weights (config 4 family): BF16 x BF16 and F32 x BF16, M = 1, F32 output (gemm_bf16 / gemm_f32_bf16,
# vector_simd.c:1226-1263,1457-1492; the BF16-activation kernel only with a zero A row stride, see oracle.ref_gemm_bf16)
wb = O.bf16_quantize(w)
xb = O.bf16_quantize(x)
out["w_bf16"], out["x_bf16"] = wb, xb
out["bf16_full"] = O.ref_gemm_bf16(xb, wb)
out["f32bf16_full"] = O.ref_gemm_f32_bf16(x, wb)
out["bf16_window"] = O.ref_gemm_bf16(xb, wb, aColOff=512, bColOff=512, K=512, bRowOff=32, N=64)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_gemm_vectors.npz"), **out)
print({k: v.shape for k, v in out.items()})
