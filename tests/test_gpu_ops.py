"""GPU parity, operator level: HipTensorOperations (through the C ABI) vs the oracle, mirroring the cases of the
reference's TestOperations.java (:111-777): every dtype pair, offsets 512/512/512 (:151-187), result offsets
(:533-552), chunked N (:714-757), batch 1 and 32 -- seeded.  Tolerances (written here):
integer/byte outputs bit-exact; I8xQ4 <= 1e-5 relative per element (integer-exact block sums, only the float
accumulation order differs); F32/BF16 accumulations <= 1e-4 relative to the row scale."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZE, ROWS, BATCH = 1024, 128, 32  # TestOperations.java:46-48


@pytest.fixture(scope="module")
def ops(gpu):
    from jlama_amd.hip_tensor_operations import HipTensorOperations
    return HipTensorOperations()


def _acts(rng, m, k):
    return rng.uniform(-1, 100, (m, k)).astype(np.float32)


def _wts(rng, n, k):
    return rng.uniform(0, 1, (n, k)).astype(np.float32)


def _close(got, want, rel):
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got - want).max() / scale
    assert err <= rel, err


def test_provider_facts(ops):
    assert ops.parallelSplitSize() == 1
    assert ops.preferredWorkingQuantizedType() == 2
    assert "HIP" in ops.name()
    assert ops.info["cu_count"] >= 64


@pytest.mark.parametrize("m", [1, BATCH])
def test_quantize_q8_bit_exact(ops, oracle, m):
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(1)
    x = _acts(rng, m, SIZE)
    x[0, 64:96] = 0.0
    x[0, 5] = -33.3
    q = ops.quantize(Tensor.f32(x), 2, 0, SIZE)
    oq, od = oracle.q8_quantize(x)
    np.testing.assert_array_equal(q.data, oq)
    np.testing.assert_array_equal(q.scales.view(np.uint32), od.view(np.uint32))
    # N(0,1) values too (negatives exercise the truncation rule)
    x = rng.standard_normal((m, SIZE)).astype(np.float32)
    q = ops.quantize(Tensor.f32(x), 2, 0, SIZE)
    oq, od = oracle.q8_quantize(x)
    np.testing.assert_array_equal(q.data, oq)
    np.testing.assert_array_equal(q.scales, od)


def test_quantize_bf16_bit_exact(ops, oracle):
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(2)
    x = rng.standard_normal((4, SIZE)).astype(np.float32)
    x[0, :4] = [1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, np.inf, -0.0]
    h = ops.quantize(Tensor.f32(x), 1, 0, SIZE)
    np.testing.assert_array_equal(h.data, oracle.bf16_quantize(x))


@pytest.mark.parametrize("m", [1, BATCH])
@pytest.mark.parametrize("registered", [False, True])
def test_batch_dot_product_i8_q4(ops, oracle, m, registered):
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(3 + m)
    a, w = _acts(rng, m, SIZE), _wts(rng, ROWS, SIZE)
    A = ops.quantize(Tensor.f32(a), 2, 0, SIZE)
    B = Tensor.q4(w)
    if registered:
        ops.registerModelTensor(B)
        assert B.reg_ids and B.reg_ids[0] >= 0
    R = Tensor.zeros(m, ROWS)
    ops.batchDotProduct(R, A, B, 0, 0, SIZE)
    want = oracle.gemm_i8q4(A.data, A.scales, B.data, B.scales)
    _close(R.data, want, 1e-5)
    # the reference's own bound: sum within 1% of the Naive control (TestOperations.java:128-139)
    ctl = oracle.gemm_naive(oracle.DT_I8, A.data, A.scales, oracle.DT_Q4, B.data, B.scales, m, 0, 0, SIZE, 0, 0, ROWS)
    assert abs(R.data.sum() - ctl.sum()) <= 0.01 * abs(ctl.sum())
    # offsets 512/512/512 (:151-187), chunked N with a row offset (:714-757)
    R = Tensor.zeros(m, ROWS)
    ops.batchDotProduct(R, A, B, 512, 512, 512, 0, 32, 64)
    want = oracle.gemm_i8q4(A.data, A.scales, B.data, B.scales, aColOff=512, bColOff=512, K=512, bRowOff=32, N=64,
                            out=np.zeros((m, ROWS), np.float32))
    _close(R.data[:, 32:96], want[:, 32:96], 1e-5)
    assert (R.data[:, :32] == 0).all() and (R.data[:, 96:] == 0).all()
    # result offset with bRowOffset == 0 (the attention-page form, CausalSelfAttention.java:329)
    R = Tensor.zeros(m, 2 * ROWS)
    ops.batchDotProduct(R, A, B, 0, 0, SIZE, ROWS, 0, ROWS)
    want = oracle.gemm_i8q4(A.data, A.scales, B.data, B.scales, rRowOff=ROWS, out=np.zeros((m, 2 * ROWS), np.float32))
    _close(R.data, want, 1e-5)


@pytest.mark.parametrize("m", [1, BATCH])
def test_batch_dot_product_f32_q4(ops, oracle, m):
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(5 + m)
    a, w = _acts(rng, m, SIZE), _wts(rng, ROWS, SIZE)
    A, B = Tensor.f32(a), Tensor.q4(w)
    R = Tensor.zeros(m, ROWS)
    ops.batchDotProduct(R, A, B, 0, 0, SIZE)
    _close(R.data, oracle.gemm_f32q4(a, B.data, B.scales), 1e-4)
    R = Tensor.zeros(m, ROWS)
    ops.dotProductChunk(R, A, B, 512, 512, 64, 32)
    want = oracle.gemm_f32q4(a, B.data, B.scales, aColOff=512, bColOff=512, K=512, bRowOff=64, N=32,
                             out=np.zeros((m, ROWS), np.float32))
    _close(R.data, want, 1e-4)


@pytest.mark.parametrize("pair", ["f32_f32", "bf16_bf16", "f32_bf16"])
@pytest.mark.parametrize("m", [1, BATCH])
def test_batch_dot_product_dense(ops, oracle, pair, m):
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(9)
    a, w = _acts(rng, m, SIZE), _wts(rng, ROWS, SIZE)
    if pair == "f32_f32":
        A, B, want = Tensor.f32(a), Tensor.f32(w), oracle.gemm_f32(a, w)
    elif pair == "bf16_bf16":
        A, B = Tensor.bf16(a), Tensor.bf16(w)
        want = oracle.gemm_bf16(A.data, B.data)
    else:
        A, B = Tensor.f32(a), Tensor.bf16(w)
        want = oracle.gemm_f32bf16(a, B.data)
    R = Tensor.zeros(m, ROWS)
    ops.batchDotProduct(R, A, B, 0, 0, SIZE)
    _close(R.data, want, 1e-4)
    # attention-score shape: q_h . K_page^T with column windows and a result offset (CausalSelfAttention.java:324-330)
    if pair == "f32_f32":
        R = Tensor.zeros(1, 96)
        ops.batchDotProduct(R, Tensor.f32(a[:1]), B, 256, 128, 128, 32, 0, 40)
        want = oracle.gemm_f32(a[:1], w, aColOff=256, bColOff=128, K=128, rRowOff=32, N=40, out=np.zeros((1, 96), np.float32))
        _close(R.data, want, 1e-4)


def test_unsupported_pairs_raise(ops):
    from jlama_amd import _native as N
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(0)
    A = Tensor.bf16(_acts(rng, 1, SIZE))
    B = Tensor.q4(_wts(rng, ROWS, SIZE))
    with pytest.raises(N.UnsupportedOperation):  # TestOperations.java:140-142 treats this as "skip"
        ops.batchDotProduct(Tensor.zeros(1, ROWS), A, B, 0, 0, SIZE)
    # K not a multiple of the block size
    r = N.lib().jh_gemm_f32_q4(-1, -1, N.ptr(np.zeros(48, np.float32)), 0, N.ptr(B.scales), N.ptr(B.data), 0,
                               N.ptr(np.zeros(4, np.float32)), 0, 1, 0, 4, 48, 48, SIZE // 2, SIZE // 32, 4)
    assert r == N.JH_ERR_INVALID
    assert b"multiple of 32" in N.lib().jh_last_error()
    # empty shapes are a no-op
    assert N.lib().jh_scale_f32(2.0, N.ptr(np.zeros(4, np.float32)), 0, 0) == 0


def test_elementwise_bit_exact(ops, oracle):
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(11)
    a, b = _acts(rng, BATCH, SIZE), _acts(rng, BATCH, SIZE)
    A = Tensor.f32(a.copy())
    ops.accumulate(A, Tensor.f32(b), 128, 512)
    want = a.copy(); want[:, 128:640] = a[:, 128:640] + b[:, 128:640]
    np.testing.assert_array_equal(A.data, want)
    A = Tensor.f32(a.copy())
    ops.accumulate(A, Tensor.f32(b[:1]), 0, SIZE)  # broadcast b
    np.testing.assert_array_equal(A.data, a + b[:1])
    A = Tensor.f32(a.copy())
    ops.maccumulate(A, Tensor.f32(b), 0, SIZE)
    np.testing.assert_array_equal(A.data, a * b)
    A = Tensor.f32(a.copy())
    ops.scale(3.3, A, 512, 256)
    want = a.copy(); want[:, 512:768] = a[:, 512:768] * np.float32(3.3)
    np.testing.assert_array_equal(A.data, want)
    # F32 += Q4 row (layer-0 residual of a JQ4 model, PanamaTensorOperations.java:2297-2325)
    B = Tensor.q4(_wts(rng, 4, SIZE))
    A = Tensor.f32(a[:1].copy())
    ops.accumulate(A, Tensor(3, B.data[2:3], B.scales[2:3]), 0, SIZE)
    np.testing.assert_array_equal(A.data[0], a[0] + oracle.q4_dequantize(B.data[2:3], B.scales[2:3])[0])
    # saxpy: fma
    y = _acts(rng, 1, SIZE); x = _acts(rng, 1, SIZE)
    Y = Tensor.f32(y.copy())
    ops.saxpy(0.37, Tensor.f32(x), Y, 64, 128, 256)
    want = y.copy()
    import ctypes
    oracle.lib().jo_saxpy_f32(ctypes.c_float(0.37), x.ctypes.data_as(ctypes.c_void_p), want.ctypes.data_as(ctypes.c_void_p), 64, 128, 256)
    np.testing.assert_array_equal(Y.data, want)
    # batched saxpy: fma chain over rows in ascending order (attention's AV step)
    alpha = rng.random((1, 48)).astype(np.float32)
    xs = rng.standard_normal((40, 256)).astype(np.float32)
    Y = Tensor.zeros(1, 512)
    ops.saxpy(Tensor.f32(alpha), Tensor.f32(xs), Y, 128, 256, 128, 8, 3, 33)
    want = oracle.saxpy_batch(alpha[0], xs, np.zeros(512, np.float32), 128, 256, 128, 8, 3, 33)
    np.testing.assert_array_equal(Y.data[0], want)


def test_norm_softmax_silu_rope(ops, oracle):
    rng = np.random.default_rng(13)
    x = rng.standard_normal(4096).astype(np.float32) * 3
    w = (1 + 0.01 * rng.standard_normal(4096)).astype(np.float32)
    np.testing.assert_array_equal(ops.rmsnorm(x, w, 1e-5), oracle.rmsnorm(x, w, 1e-5))  # double-precision reduce
    s = rng.standard_normal(385).astype(np.float32) * 5
    got, want = ops.softmax(s, 0, 385), oracle.softmax(s, 0, 385)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-12)  # float sum order differs (parallel vs sequential)
    g = rng.standard_normal(2048).astype(np.float32) * 4
    u = rng.standard_normal(2048).astype(np.float32)
    got = ops.silu_mul(g, u)
    want = (oracle.silu(g) * u).astype(np.float32)
    assert (np.abs(got.view(np.int32) - want.view(np.int32)) <= 1).all()  # device exp vs libm exp: <= 1 ulp
    assert (got == want).mean() > 0.999
    table = ops.rope_table(128, 512, 500000.0)
    np.testing.assert_array_equal(table, oracle.rope_table(128, 512, 500000.0))
    q = rng.standard_normal(32 * 128).astype(np.float32)
    k = rng.standard_normal(8 * 128).astype(np.float32)
    gq, gk = ops.rope_apply(q, k, table, 37, 32, 8, 128)
    # oracle rotation incl. the per-kv-head table offset (effective position pos + 2*kvHead)
    half = 64
    wq, wk = q.copy(), k.copy()
    for h in range(32):
        for i in range(half):
            c, sn = table[37 * half + (h // 4) * 128 + i]
            q0, q1 = q[h * 128 + i], q[h * 128 + i + half]
            wq[h * 128 + i] = np.float32(q0 * c) - np.float32(q1 * sn)
            wq[h * 128 + i + half] = np.float32(q0 * sn) + np.float32(q1 * c)
    for h in range(8):
        for i in range(half):
            c, sn = table[37 * half + h * 128 + i]
            k0, k1 = k[h * 128 + i], k[h * 128 + i + half]
            wk[h * 128 + i] = np.float32(k0 * c) - np.float32(k1 * sn)
            wk[h * 128 + i + half] = np.float32(k0 * sn) + np.float32(k1 * c)
    np.testing.assert_array_equal(gq, wq)
    np.testing.assert_array_equal(gk, wk)


def test_gemv_at_model_shapes_and_linearity(ops, oracle):
    """Llama-3-8B projection shapes through the fast M=1 kernel (down-proj K=14336 uses 7 blocks per lane), plus a
    size-independent property at full size: C(a) is linear in the activation scales."""
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(17)
    for (n, k) in [(1024, 4096), (512, 14336), (256, 2048), (128, 8192), (96, 1600)]:
        w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        a = rng.standard_normal((1, k)).astype(np.float32)
        B = Tensor.q4(w)
        A = ops.quantize(Tensor.f32(a), 2, 0, k)
        R = Tensor.zeros(1, n)
        ops.batchDotProduct(R, A, B, 0, 0, k)
        _close(R.data, oracle.gemm_i8q4(A.data, A.scales, B.data, B.scales), 1e-5)
        R2 = Tensor.zeros(1, n)
        A2 = Tensor.i8(A.data, A.scales * np.float32(2.0))
        ops.batchDotProduct(R2, A2, B, 0, 0, k)
        np.testing.assert_array_equal(R2.data, R.data * np.float32(2.0))  # power-of-two scaling is exact
        Rf = Tensor.zeros(1, n)
        ops.batchDotProduct(Rf, Tensor.f32(a), B, 0, 0, k)
        _close(Rf.data, oracle.gemm_f32q4(a, B.data, B.scales), 1e-4)


@pytest.mark.parametrize("m,n,k", [(129, 1024, 4096), (32, 128, 1024), (256, 256, 512), (5, 64, 256), (129, 4096, 1024)])
def test_batched_bf16_gemm_on_mfma(ops, oracle, m, n, k):
    """Prefill-shaped BF16 x BF16 -> F32 GEMM (GemmerBF16) through jh_gemm_bf16: M >= 2 takes the
    v_mfma_f32_32x32x16_bf16 kernel.  Asymmetric operands (cdna_hip_programming.md §3: a symmetric B hides a
    row/col swap); products are exact in F32, only the accumulation order differs => 1e-4 of the row scale."""
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(m * 7 + n)
    a = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) * np.linspace(0.5, 2.0, n)[:, None]).astype(np.float32)
    A, B = Tensor.bf16(a), Tensor.bf16(w)
    R = Tensor.zeros(m, n)
    ops.batchDotProduct(R, A, B, 0, 0, k)
    want = oracle.gemm_bf16(A.data, B.data)
    _close(R.data, want, 1e-4)
    # the identity probe: A = I (bf16 exact) must return W's leading block, transposed -- catches fragment-layout swaps
    if m >= 32 and k >= 64:
        eye = np.zeros((m, k), dtype=np.float32)
        eye[np.arange(min(m, k)), np.arange(min(m, k))] = 1.0
        R2 = Tensor.zeros(m, n)
        ops.batchDotProduct(R2, Tensor.bf16(eye), B, 0, 0, k)
        wb = oracle.bf16_to_f32(B.data)
        np.testing.assert_array_equal(R2.data[: min(m, k)], wb[:, : min(m, k)].T)
    # column window + row chunk + result offset through the same kernel
    if n >= 128 and k >= 512:
        R3 = Tensor.zeros(m, n)
        ops.batchDotProduct(R3, A, B, 256, 256, 256, 0, 32, 64)
        want3 = oracle.gemm_bf16(A.data, B.data, aColOff=256, bColOff=256, K=256, bRowOff=32, N=64, out=np.zeros((m, n), np.float32))
        _close(R3.data[:, 32:96], want3[:, 32:96], 1e-4)
        assert (R3.data[:, :32] == 0).all() and (R3.data[:, 96:] == 0).all()


@pytest.mark.parametrize("m,n,k", [(129, 1024, 4096), (32, 128, 1024), (192, 256, 512), (256, 64, 256), (5, 64, 64), (2, 4096, 1024)])
def test_batched_i8q4_gemm_on_mfma(ops, oracle, m, n, k):
    """Prefill-shaped I8 x Q4 -> F32 GEMM (GemmerI8Q4_512 2x2 tile / gemm_q8_q4) through jh_gemm_q8_q4: M >= 2 takes the
    v_mfma_i32_32x32x32_i8 kernel (one Q block per MFMA => exact integer block sums).  1e-5 of the row scale."""
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(m * 11 + n)
    a = rng.uniform(-1, 100, (m, k)).astype(np.float32)
    a[1] = rng.standard_normal(k)          # a row with negatives (sum(a_block) of both signs)
    w = (rng.uniform(0, 1, (n, k)) * np.linspace(0.5, 2.0, n)[:, None]).astype(np.float32)
    w[::3] *= -1
    A = ops.quantize(Tensor.f32(a), 2, 0, k)
    B = Tensor.q4(w)
    R = Tensor.zeros(m, n)
    ops.batchDotProduct(R, A, B, 0, 0, k)
    want = oracle.gemm_i8q4(A.data, A.scales, B.data, B.scales)
    _close(R.data, want, 1e-5)
    if k >= 512 and n >= 128:      # window: column offset 256, rows [32, 96)
        R3 = Tensor.zeros(m, n)
        ops.batchDotProduct(R3, A, B, 256, 256, 256, 0, 32, 64)
        want3 = oracle.gemm_i8q4(A.data, A.scales, B.data, B.scales, aColOff=256, bColOff=256, K=256, bRowOff=32, N=64,
                                 out=np.zeros((m, n), np.float32))
        _close(R3.data[:, 32:96], want3[:, 32:96], 1e-5)
        assert (R3.data[:, :32] == 0).all() and (R3.data[:, 96:] == 0).all()


def test_against_vectors_from_the_reference_library(ops):
    """tests/golden/ref_gemm_vectors.npz = outputs of the reference's OWN C SIMD GEMM (vector_simd.c compiled as-is,
    flags = the _512 kernels; generated by tests/golden/make_ref_gemm_vectors.py in the build container).  No oracle in
    this test: libjlamahip.so vs the reference.  I8xQ4: identical integer block sums, float grouping differs (8 lanes of
    4-element sums in C vs a 64-lane DPP tree here) => 1e-5; F32xQ4 / F32xF32 => 1e-4 of the row scale.  The device
    Q8 quantizer must reproduce the codes the reference consumed, bit for bit."""
    import os
    from jlama_amd import _native as N
    from jlama_amd.jq4 import Tensor
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_gemm_vectors.npz"))
    B = Tensor(N.DT_Q4, np.ascontiguousarray(g["nib"]), np.ascontiguousarray(g["scales"]))
    X = Tensor.f32(g["x"])
    A = ops.quantize(X, 2, 0, X.cols)
    np.testing.assert_array_equal(A.data, g["aq"])
    np.testing.assert_array_equal(A.scales, g["ad"])
    n, k = B.rows, B.cols
    R = Tensor.zeros(1, n)
    ops.batchDotProduct(R, A, B, 0, 0, k)
    _close(R.data, g["q8q4_full"], 1e-5)
    R = Tensor.zeros(1, n)
    ops.batchDotProduct(R, X, B, 0, 0, k)
    _close(R.data, g["f32q4_full"], 1e-4)
    for kind, src, tol in (("q8q4_window", A, 1e-5), ("f32q4_window", X, 1e-4)):
        R = Tensor.zeros(1, 96)
        ops.batchDotProduct(R, src, B, 512, 512, 512, 0, 32, 64)
        _close(R.data[:, 32:96], g[kind][:, 32:96], tol)
        assert (R.data[:, :32] == 0).all()
    Q, KP = Tensor.f32(g["f32_q"]), Tensor.f32(g["f32_kpage"])
    R = Tensor.zeros(1, 48)
    ops.batchDotProduct(R, Q, KP, 128, 128, 128)
    _close(R.data, g["f32_scores"], 1e-4)
    # BF16 weights (config 4): gemm_bf16 / gemm_f32_bf16 of the reference library, M = 1, F32 output; the device BF16
    # quantizer must reproduce the activation codes the reference consumed
    WB = Tensor(N.DT_BF16, np.ascontiguousarray(g["w_bf16"]))
    XB = ops.quantize(X, N.DT_BF16, 0, X.cols)
    np.testing.assert_array_equal(XB.data, g["x_bf16"])
    R = Tensor.zeros(1, n)
    ops.batchDotProduct(R, XB, WB, 0, 0, k)
    _close(R.data, g["bf16_full"], 1e-4)
    R = Tensor.zeros(1, n)
    ops.batchDotProduct(R, X, WB, 0, 0, k)
    _close(R.data, g["f32bf16_full"], 1e-4)
    R = Tensor.zeros(1, 96)
    ops.batchDotProduct(R, XB, WB, 512, 512, 512, 0, 32, 64)
    _close(R.data[:, 32:96], g["bf16_window"][:, 32:96], 1e-4)


def test_gpt2_family_pieces(ops, oracle):
    """BASELINE configs[0] (GPT-2 small: E=768, F32 weights, LayerNorm + bias, tanh-GELU): the pieces outside the GEMM.
    LayerNorm is bit-exact (float sums in index order, as LayerNorm.java:47-53; no FMA in the affine step); GELU is
    evaluated in double and cast (ActivationFunction.java:32-34): bit-exact against the libm restatement.  The F32xF32
    GEMM with bias accumulate is covered by test_batch_dot_product_dense / test_elementwise_bit_exact."""
    rng = np.random.default_rng(77)
    E = 768
    x = (rng.standard_normal((5, E)) * 3 + 0.5).astype(np.float32)
    w = (1 + rng.standard_normal(E) * 0.1).astype(np.float32)
    b = (rng.standard_normal(E) * 0.1).astype(np.float32)
    np.testing.assert_array_equal(ops.layer_norm(x, w, b, 1e-5), oracle.layernorm(x, w, b, 1e-5))
    # the offset/length form (forward(input, offset, length)): statistics over the window, divisor stays E
    got = ops.layer_norm(x, w, b, 1e-5, offset=256, length=256)
    want = oracle.layernorm(x, w, b, 1e-5, offset=256, length=256)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got[:, :256], x[:, :256])
    g = (rng.standard_normal(4096) * 4).astype(np.float32)
    np.testing.assert_array_equal(ops.gelu(g), oracle.gelu(g))
    assert ops.gelu(np.zeros(3, np.float32)).tolist() == [0.0, 0.0, 0.0]


@pytest.mark.parametrize("pair", ["bf16_bf16", "f32_bf16"])
@pytest.mark.parametrize("m", [1, 5, 129])
def test_bf16_result_tensor(ops, oracle, pair, m):
    """The `cr` output of gemm_bf16 / gemm_f32_bf16 (vector_simd.h:34,38; NativeSimdTensorOperations.java:113-131 passes it
    when result.dType()==BF16): results rounded with FloatConversions.float32ToBFloat16 (RNE).  The rounding of a value
    within float-ordering noise of a BF16 tie may differ, so: identical to the rounded F32 result of the SAME kernel, and
    within one BF16 ulp of the oracle's."""
    from jlama_amd.jq4 import Tensor
    from jlama_amd import _native as N
    rng = np.random.default_rng(31 + m)
    k, n = 512, 96
    a = rng.standard_normal((m, k)).astype(np.float32)
    w = rng.standard_normal((n, k)).astype(np.float32)
    A = Tensor.bf16(a) if pair == "bf16_bf16" else Tensor.f32(a)
    B = Tensor.bf16(w)
    Rf = Tensor.zeros(m, n)
    ops.batchDotProduct(Rf, A, B, 0, 0, k)
    Rb = Tensor(N.DT_BF16, np.zeros((m, n), dtype=np.uint16))
    ops.batchDotProduct(Rb, A, B, 0, 0, k)
    np.testing.assert_array_equal(Rb.data, oracle.bf16_quantize(Rf.data))
    want = oracle.gemm_bf16(A.data, B.data) if pair == "bf16_bf16" else oracle.gemm_f32bf16(a, B.data)
    assert np.abs(Rb.data.astype(np.int32) - oracle.bf16_quantize(want).astype(np.int32)).max() <= 1
    # window form: only columns [32, 96) are written, the rest of the BF16 result keeps the caller's bytes
    Rw = Tensor(N.DT_BF16, np.full((m, n), 0x1234, dtype=np.uint16))
    ops.batchDotProduct(Rw, A, B, 256, 256, 256, 0, 32, 64)
    assert (Rw.data[:, :32] == 0x1234).all()
    ref = np.zeros((m, n), np.float32)
    Rr = Tensor.f32(ref)
    ops.batchDotProduct(Rr, A, B, 256, 256, 256, 0, 32, 64)
    np.testing.assert_array_equal(Rw.data[:, 32:], oracle.bf16_quantize(Rr.data[:, 32:]))


@pytest.mark.parametrize("pair", ["i8_q4", "f32_q4", "f32_f32", "bf16_bf16", "f32_bf16"])
def test_dot_product_batch_chunk_entry_points(ops, oracle, pair):
    """dotProductBatchChunk (TensorOperations.java:86-99) through every `_batch` entry point of the reference library
    (vector_simd.h:23,27,31,35,39): one activation against several weight tensors, pointer arrays like
    MemorySegmentSupport.setupBatch -- each result equals the single call's, bit for bit."""
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(41)
    m, k, n = 3, 512, 64
    a = _acts(rng, m, k)
    ws = [_wts(rng, n, k) * s for s in (1.0, -0.5, 2.0)]
    if pair == "i8_q4":
        A, Bs = ops.quantize(Tensor.f32(a), 2, 0, k), [Tensor.q4(w) for w in ws]
    elif pair == "f32_q4":
        A, Bs = Tensor.f32(a), [Tensor.q4(w) for w in ws]
    elif pair == "f32_f32":
        A, Bs = Tensor.f32(a), [Tensor.f32(w) for w in ws]
    elif pair == "bf16_bf16":
        A, Bs = Tensor.bf16(a), [Tensor.bf16(w) for w in ws]
    else:
        A, Bs = Tensor.f32(a), [Tensor.bf16(w) for w in ws]
    ops.registerModelTensor(Bs[1])                      # a mix of registered and host-resident weights
    Rs = [Tensor.zeros(m, n) for _ in ws]
    ops.dotProductBatchChunk(Rs, A, Bs, 128, 256, 16, 32)
    for R, B in zip(Rs, Bs):
        single = Tensor.zeros(m, n)
        ops.dotProductChunk(single, A, B, 128, 256, 16, 32)
        np.testing.assert_array_equal(R.data, single.data)
        assert np.abs(R.data[:, 16:48]).max() > 0 and (R.data[:, :16] == 0).all() and (R.data[:, 48:] == 0).all()


def test_tier1_calls_are_reentrant(ops, oracle):
    """The reference calls batchDotProduct from many pfor workers at once (heads of one attention step,
    CausalSelfAttention.java:314; VectorMath.pfor): Tier-1 keeps a stream + scratch per calling thread.  8 threads, each
    issuing the score GEMM of its own heads (F32 x F32, column windows into shared q / K tensors) plus an I8 x Q4 GEMV
    against a registered weight, many times over; every result must equal the single-threaded one bit for bit."""
    import threading
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(51)
    heads, hs, ctx = 32, 128, 96
    q = Tensor.f32(rng.standard_normal((1, heads * hs)).astype(np.float32))
    kpage = Tensor.f32(rng.standard_normal((ctx, 8 * hs)).astype(np.float32))
    w = Tensor.q4(_wts(rng, 256, 1024))
    ops.registerModelTensor(w)
    aq = ops.quantize(Tensor.f32(_acts(rng, 1, 1024)), 2, 0, 1024)

    def scores(h, out):
        ops.batchDotProduct(out, q, kpage, h * hs, (h // 4) * hs, hs, 0, 0, ctx)

    want = []
    for h in range(heads):
        r = Tensor.zeros(1, ctx)
        scores(h, r)
        want.append(r.data.copy())
    wg = Tensor.zeros(1, 256)
    ops.batchDotProduct(wg, aq, w, 0, 0, 1024)
    errors = []

    def worker(t):
        try:
            for rep in range(20):
                for h in range(t, heads, 8):
                    r = Tensor.zeros(1, ctx)
                    scores(h, r)
                    if not np.array_equal(r.data, want[h]):
                        errors.append(("scores", t, h, rep))
                g = Tensor.zeros(1, 256)
                ops.batchDotProduct(g, aq, w, 0, 0, 1024)
                if not np.array_equal(g.data, wg.data):
                    errors.append(("gemv", t, rep))
        except Exception as e:   # noqa: BLE001
            errors.append(("exc", t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
