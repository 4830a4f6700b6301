"""CPU tests that PIN the oracle (oracle/jlama_oracle.c) before anything trusts it:

* the reference's own known-answer vectors (RoPE table, TestCorrectness.java:92-115),
* the reference's own C SIMD GEMM library compiled as-is (oracle/_ref), bit-exact for F32xQ4 / F32xF32,
* the reference tests' control implementation (NaiveTensorOperations) within the reference's own 1% bound
  (TestOperations.java:128-139) -- in practice ~1e-6,
* BF16 round-trip bound (TestCorrectness.java:137-145), KV page geometries quoted in SURVEY.md 8(a7).
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_rope_table_known_answers(oracle):
    kat = json.load(open(os.path.join(HERE, "golden", "rope_kat.json")))
    t = oracle.rope_table(kat["head_dim"], kat["context"], kat["theta"], kat["scaling"])
    half = kat["head_dim"] // 2
    np.testing.assert_allclose(t[1 * half:(1 + 1) * half, 1], kat["sin_pos1"], atol=kat["tolerance"])
    np.testing.assert_allclose(t[64 * half:65 * half, 1], kat["sin_pos64"], atol=kat["tolerance"])
    # cos^2 + sin^2 == 1 everywhere (size-independent property)
    np.testing.assert_allclose((t.astype(np.float64) ** 2).sum(axis=1), 1.0, atol=1e-6)


def test_bf16_round_trip(oracle):
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, 4096).astype(np.float32)
    h = oracle.bf16_quantize(x)
    back = oracle.bf16_to_f32(h)
    assert np.abs(back - x).max() < 0.01  # TestCorrectness.java:137-145
    # RNE ties: 1.0 + 2^-8 is exactly half way between two bf16 values -> even mantissa
    tie = np.array([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8], dtype=np.float32)
    hh = oracle.bf16_quantize(tie)
    assert hh[0] == 0x3F80 and hh[1] == 0x3F82


def test_kv_page_geometry(oracle):
    # Llama-3-8B 32x32, Llama-3.2-1B 16x128, Llama-3-70B 80x12, 10-layer shard 10x102, GPT-2 11x124
    assert oracle.kv_page_geometry(1 << 23, 32, 8192, 1024) == (32, 32)
    assert oracle.kv_page_geometry(1 << 23, 16, 131072, 512) == (16, 128)
    assert oracle.kv_page_geometry(1 << 23, 80, 8192, 1024) == (80, 12)
    assert oracle.kv_page_geometry(1 << 23, 10, 8192, 1024) == (10, 102)
    assert oracle.kv_page_geometry(1 << 23, 12, 1024, 768) == (11, 124)


def test_q4_layout_and_quantizer(oracle):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((8, 256)).astype(np.float32)
    x[0, :32] = 0.0                      # all-zero block: scale -0.0, every nibble 8
    x[1, 5] = -7.5                       # negative max => positive scale
    nib, sc = oracle.q4_quantize(x)
    assert nib.shape == (8, 128) and sc.shape == (8, 8)
    assert (nib[0, :16] == 0x88).all() and sc[0, 0] == 0.0
    deq = oracle.q4_dequantize(nib, sc)
    # value = (nibble-8)*scale, byte j: low nibble = elem j, high = elem j+16 (Q4ByteBufferTensor.java:88-106)
    b = nib[2, :16]
    manual = np.concatenate([(b & 15).astype(np.int32) - 8, (b >> 4).astype(np.int32) - 8]) * sc[2, 0]
    np.testing.assert_array_equal(deq[2, :32], manual.astype(np.float32))
    # quantization error bounded by one step; the max-abs element maps to nibble 0 exactly
    step = np.abs(np.repeat(sc, 32, axis=1))
    assert (np.abs(deq - x) <= step * 1.0001 + 1e-12).all()


def test_numpy_q4_quantizer_matches_oracle(oracle):
    from jlama_amd import jq4
    rng = np.random.default_rng(1)
    for scale in (1.0, 0.02, 1e-30, 100.0):
        x = (rng.standard_normal((16, 512)) * scale).astype(np.float32)
        x[3, 64:96] = 0
        n1, s1 = oracle.q4_quantize(x)
        n2, s2 = jq4.quantize_q4(x)
        np.testing.assert_array_equal(n1, n2)
        np.testing.assert_array_equal(s1.view(np.uint32), s2.view(np.uint32))
        np.testing.assert_array_equal(oracle.q4_dequantize(n1, s1), jq4.dequantize_q4(n2, s2))
    h1 = oracle.bf16_quantize(x)
    np.testing.assert_array_equal(h1, jq4.f32_to_bf16(x))


def test_q8_quantizer_truncation_semantics(oracle):
    # q = (byte)(x*id + 0.5f): truncation toward zero AFTER adding 0.5 => -0.7*... rounds toward zero for negatives
    x = np.zeros((1, 32), dtype=np.float32)
    x[0, 0] = 127.0
    x[0, 1] = -1.4      # -1.4 + 0.5 = -0.9 -> 0   (round-to-nearest would give -1)
    x[0, 2] = -1.6      # -1.6 + 0.5 = -1.1 -> -1  (round-to-nearest would give -2)
    x[0, 3] = 1.5       # 2.0 -> 2
    x[0, 4] = -127.0    # -126.5 -> -126
    q, d = oracle.q8_quantize(x)
    assert d[0, 0] == np.float32(1.0)
    assert list(q[0, :5]) == [127, 0, -1, 2, -126]
    z, dz = oracle.q8_quantize(np.zeros((1, 32), dtype=np.float32))
    assert (z == 0).all() and dz[0, 0] == 0.0


def _operands(rng, M, N, K):
    # value ranges of TestOperations.java:94-109: activations U(-1,100), weights U(0,1)
    a = rng.uniform(-1, 100, (M, K)).astype(np.float32)
    w = rng.uniform(0, 1, (N, K)).astype(np.float32)
    return a, w


@pytest.mark.parametrize("M", [1, 2, 5])
def test_gemm_against_reference_library(oracle, M):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference sources absent)")
    rng = np.random.default_rng(10 + M)
    K, N = 1024, 128  # SIZE / ROWS of TestOperations.java:46-48
    a, w = _operands(rng, M, N, K)
    bn, bs = oracle.q4_quantize(w)
    aq, ad = oracle.q8_quantize(a)
    # F32xQ4 and F32xF32: same 16-lane fma structure as the C twins => bit-exact
    np.testing.assert_array_equal(oracle.gemm_f32q4(a, bn, bs), oracle.ref_gemm_f32_q4(a, bn, bs))
    np.testing.assert_array_equal(oracle.gemm_f32(a, w), oracle.ref_gemm_f32(a, w))
    # I8xQ4: identical integers, 16 (Panama) vs 8 (C) float lanes => rounding-level difference only
    r1, r2 = oracle.gemm_i8q4(aq, ad, bn, bs), oracle.ref_gemm_q8_q4(aq, ad, bn, bs)
    assert np.abs(r1 - r2).max() <= 2e-6 * np.abs(r1).max()
    # windows: column offsets 512 (TestOperations.java:151-187) and result offsets
    r1 = oracle.gemm_i8q4(aq, ad, bn, bs, aColOff=512, bColOff=512, K=512, rRowOff=0, bRowOff=32, N=64)
    r2 = oracle.ref_gemm_q8_q4(aq, ad, bn, bs, aColOff=512, bColOff=512, K=512, rRowOff=0, bRowOff=32, N=64)
    assert np.abs(r1 - r2).max() <= 2e-6 * np.abs(r1).max()


def test_reference_library_tiler_leaves_corner_uncomputed(oracle):
    """Documented reference quirk: nc/simd/vector_simd.c:65-184 `gemm()` recursion never visits
    [mp,m) x [np,n) -- with M=8, N=128 the 3x3 corner stays zero.  This is why the compiled reference is used
    as an oracle for M <= 5 only, and why decode (M=1) is the pinned case."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    a, w = _operands(rng, 8, 128, 1024)
    bn, bs = oracle.q4_quantize(w)
    ref = oracle.ref_gemm_f32_q4(a, bn, bs)
    ours = oracle.gemm_f32q4(a, bn, bs)
    assert (ref[5:, 125:] == 0).all()
    np.testing.assert_array_equal(ref[:5], ours[:5])
    np.testing.assert_array_equal(ref[5:, :125], ours[5:, :125])


@pytest.mark.parametrize("M", [1, 32])
def test_gemm_against_naive_control(oracle, M):
    """The reference's own test contract: every provider within 1% of NaiveTensorOperations
    (TestOperations.java:128-139, sums compared).  We hold 1e-5 per element."""
    O = oracle
    rng = np.random.default_rng(20 + M)
    K, N = 1024, 128
    a, w = _operands(rng, M, N, K)
    bn, bs = O.q4_quantize(w)
    aq, ad = O.q8_quantize(a)
    ah, wh = O.bf16_quantize(a), O.bf16_quantize(w)
    cases = [
        (O.gemm_i8q4(aq, ad, bn, bs), O.gemm_naive(O.DT_I8, aq, ad, O.DT_Q4, bn, bs, M, 0, 0, K, 0, 0, N)),
        (O.gemm_f32q4(a, bn, bs), O.gemm_naive(O.DT_F32, a, None, O.DT_Q4, bn, bs, M, 0, 0, K, 0, 0, N)),
        (O.gemm_f32(a, w), O.gemm_naive(O.DT_F32, a, None, O.DT_F32, w, None, M, 0, 0, K, 0, 0, N)),
        (O.gemm_bf16(ah, wh), O.gemm_naive(O.DT_BF16, ah, None, O.DT_BF16, wh, None, M, 0, 0, K, 0, 0, N)),
        (O.gemm_f32bf16(a, wh), O.gemm_naive(O.DT_F32, a, None, O.DT_BF16, wh, None, M, 0, 0, K, 0, 0, N)),
    ]
    for got, ctl in cases:
        assert abs(got.sum() - ctl.sum()) <= 0.01 * abs(ctl.sum())          # the reference's bound
        assert np.abs(got - ctl).max() <= 1e-5 * np.abs(ctl).max()          # ours


def test_small_ops(oracle):
    O = oracle
    rng = np.random.default_rng(7)
    x = rng.standard_normal(512).astype(np.float32)
    w = (1 + 0.01 * rng.standard_normal(512)).astype(np.float32)
    y = O.rmsnorm(x, w, 1e-5)
    ref = w.astype(np.float64) * x / np.sqrt((x.astype(np.float64) ** 2).mean() + 1e-5)
    np.testing.assert_allclose(y, ref, rtol=3e-7)
    s = O.softmax(rng.standard_normal(300).astype(np.float32) * 4, 0, 300)
    assert abs(s.sum() - 1) < 1e-5 and (s >= 0).all()
    v = np.array([-20, -1, 0, 1, 20], dtype=np.float32)
    np.testing.assert_allclose(O.silu(v), v / (1 + np.exp(-v.astype(np.float64))), rtol=1e-7)
    # batched saxpy == fma chain over rows
    alpha = rng.random(10).astype(np.float32)
    xs = rng.standard_normal((10, 64)).astype(np.float32)
    out = O.saxpy_batch(alpha, xs, np.zeros(64, np.float32), 0, 0, 64, 0, 0, 10)
    np.testing.assert_allclose(out, (alpha[:, None].astype(np.float64) * xs).sum(0), rtol=1e-5, atol=1e-6)


def test_model_forward_is_causal_and_batch_equals_sequential(oracle):
    """Whole-model restatement: prefill as one batch == token-by-token (batchForwardSlow, AbstractModel.java:282-290),
    logits finite, greedy generation deterministic."""
    from jlama_amd import synthetic as S
    cfg = dict(S.TINY)
    w = S.make_weights(cfg, seed=0)
    m = oracle.OracleModel(cfg, w)
    prompt = S.prompt_tokens(cfg, n=12, seed=3)
    s1 = m.session()
    full = s1.forward(prompt, 0)
    s2 = m.session()
    rows = [s2.forward(prompt[i:i + 1], i)[0] for i in range(prompt.size)]
    np.testing.assert_array_equal(full, np.stack(rows))
    tok, logits = m.sample(full[-1])
    assert np.isfinite(logits).all() and 0 <= tok < cfg["vocab_size"]
    g1, _, _ = m.session().generate(prompt, 8)
    g2, _, _ = m.session().generate(prompt, 8)
    np.testing.assert_array_equal(g1, g2)
    assert g1[0] == tok


def test_reference_gemm_driven_model_matches_panama_order_at_m1(oracle):
    """bench.py's cpu_baseline leg drives the reference's compiled C GEMM from the restated decode loop.  At M = 1
    (decode, and prompts fed row by row) it must agree with the Panama-order restatement to rounding."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    from jlama_amd import synthetic as S
    cfg = dict(S.SMALL)
    w = S.make_weights(cfg, seed=1)
    m1, m2 = oracle.OracleModel(cfg, w), oracle.OracleModel(cfg, w)
    m2.use_reference_gemm(2)
    p = S.prompt_tokens(cfg, n=7, seed=2)
    s1, s2 = m1.session(), m2.session()
    for i, t in enumerate(p):
        x1, x2 = s1.forward([t], i), s2.forward([t], i)
    assert np.abs(x1 - x2).max() <= 1e-5 * np.abs(x1).max()
    (t1, l1), (t2, l2) = m1.sample(x1[-1]), m2.sample(x2[-1])
    assert t1 == t2 and np.abs(l1 - l2).max() <= 1e-5


def test_oracle_matches_committed_reference_vectors(oracle):
    """The committed outputs of the reference's C GEMM (tests/golden/ref_gemm_vectors.npz) pin the oracle even where
    oracle/_ref cannot be rebuilt (no /root/reference): F32xQ4 and F32xF32 bit-exact (same 16-lane FMA order and
    halving-tree reduce), I8xQ4 within 2e-6 of the row scale (integer sums identical; the C kernel groups 4-element
    int32 sums over 8 lanes, Panama/the oracle 2-element sums over 16)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_gemm_vectors.npz"))
    aq, ad = oracle.q8_quantize(g["x"])
    np.testing.assert_array_equal(aq, g["aq"])
    np.testing.assert_array_equal(ad, g["ad"])
    got = oracle.gemm_i8q4(g["aq"], g["ad"], g["nib"], g["scales"])
    assert np.abs(got - g["q8q4_full"]).max() <= 2e-6 * np.abs(g["q8q4_full"]).max()
    np.testing.assert_array_equal(oracle.gemm_f32q4(g["x"], g["nib"], g["scales"]), g["f32q4_full"])
    got = oracle.gemm_f32q4(g["x"], g["nib"], g["scales"], aColOff=512, bColOff=512, K=512, bRowOff=32, N=64,
                            out=np.zeros((1, 96), np.float32))
    np.testing.assert_array_equal(got[:, 32:96], g["f32q4_window"][:, 32:96])
    np.testing.assert_array_equal(oracle.gemm_f32(g["f32_q"], g["f32_kpage"], aColOff=128, bColOff=128, K=128), g["f32_scores"])
    # BF16 legs (GemmerBF16 / GemmerF32BF16 restatements) against the reference library's gemm_bf16 / gemm_f32_bf16, M = 1:
    # bit-exact (elements t then t+16 into 16 lanes, halving-tree reduce); activations' BF16 codes = RNE of the F32 row
    np.testing.assert_array_equal(oracle.bf16_quantize(g["x"]), g["x_bf16"])
    np.testing.assert_array_equal(oracle.gemm_bf16(g["x_bf16"], g["w_bf16"]), g["bf16_full"])
    np.testing.assert_array_equal(oracle.gemm_f32bf16(g["x"], g["w_bf16"]), g["f32bf16_full"])
    got = oracle.gemm_bf16(g["x_bf16"], g["w_bf16"], aColOff=512, bColOff=512, K=512, bRowOff=32, N=64, out=np.zeros((1, 96), np.float32))
    np.testing.assert_array_equal(got[:, 32:96], g["bf16_window"][:, 32:96])


def test_layernorm_and_gelu_restatements(oracle):
    """LayerNorm.java:41-67 / ActivationFunction.java:32-34 against straightforward numpy float32 / float64 evaluations."""
    rng = np.random.default_rng(5)
    E = 96
    x = rng.standard_normal((2, E)).astype(np.float32)
    w = rng.standard_normal(E).astype(np.float32)
    b = rng.standard_normal(E).astype(np.float32)
    got = oracle.layernorm(x, w, b, 1e-5)
    for r in range(2):
        s = np.float32(0); q = np.float32(0)
        for v in x[r]:
            s = np.float32(s + v); q = np.float32(q + np.float32(v * v))
        mean = np.float32(s / np.float32(E))
        var = np.float32(np.float32(q / np.float32(E)) - np.float32(mean * mean))
        inv = np.float32(np.float32(1.0) / np.float32(np.sqrt(np.float64(np.float32(var + np.float32(1e-5))))))
        want = (((x[r] - mean).astype(np.float32) * inv).astype(np.float32) * w).astype(np.float32) + b
        np.testing.assert_array_equal(got[r], want.astype(np.float32))
    g = rng.standard_normal(64).astype(np.float32) * 3
    v = g.astype(np.float64)
    want = (0.5 * v * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (v + 0.044715 * v ** 3)))).astype(np.float32)
    assert np.abs(oracle.gelu(g) - want).max() <= 1e-6


def test_simd_bodies_equal_the_scalar_text(oracle):
    """The oracle's optional AVX2 bodies of the two big GEMV loops (what makes a full-size 8B oracle run affordable) must
    reproduce the scalar restatement bit for bit: same lanes, same order, same roundings."""
    O = oracle
    L = O.lib()
    rng = np.random.default_rng(77)
    for (M, N, K) in [(1, 96, 4096), (3, 40, 1024), (2, 33, 14336)]:
        x = (rng.standard_normal((M, K)) * rng.uniform(0.01, 30)).astype(np.float32)
        w = rng.standard_normal((N, K)).astype(np.float32)
        bn, bs = O.q4_quantize(w)
        aq, ad = O.q8_quantize(x)
        p = O._p
        a = np.zeros((M, N), np.float32); b = np.zeros((M, N), np.float32)
        L.jo_gemm_i8q4(p(aq), p(ad), K, K // 32, p(bn), p(bs), K // 2, K // 32, p(a), N, M, 0, 0, K, 0, 0, N)
        L.jo_gemm_i8q4_scalar(p(aq), p(ad), K, K // 32, p(bn), p(bs), K // 2, K // 32, p(b), N, M, 0, 0, K, 0, 0, N)
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
        a[:] = 0; b[:] = 0
        L.jo_gemm_f32q4(p(x), K, p(bn), p(bs), K // 2, K // 32, p(a), N, M, 0, 0, K, 0, 0, N)
        L.jo_gemm_f32q4_scalar(p(x), K, p(bn), p(bs), K // 2, K // 32, p(b), N, M, 0, 0, K, 0, 0, N)
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
        # windows: column offsets and a row offset, as the attention / sharded callers use them
        a[:] = 0; b[:] = 0
        L.jo_gemm_i8q4(p(aq), p(ad), K, K // 32, p(bn), p(bs), K // 2, K // 32, p(a), N, M, 512, 512, 256, 0, 8, 16)
        L.jo_gemm_i8q4_scalar(p(aq), p(ad), K, K // 32, p(bn), p(bs), K // 2, K // 32, p(b), N, M, 512, 512, 256, 0, 8, 16)
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
        # dense BF16 (GemmerBF16 / GemmerF32BF16 restated): what makes a full-size Mistral-7B oracle run affordable
        from jlama_amd import jq4
        xb, wb = jq4.f32_to_bf16(x), jq4.f32_to_bf16(w)
        for fast, slow, act in ((L.jo_gemm_bf16, L.jo_gemm_bf16_scalar, xb), (L.jo_gemm_f32bf16, L.jo_gemm_f32bf16_scalar, x)):
            a[:] = 0; b[:] = 0
            fast(p(act), K, p(wb), K, p(a), N, M, 0, 0, K, 0, 0, N)
            slow(p(act), K, p(wb), K, p(b), N, M, 0, 0, K, 0, 0, N)
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
            a[:] = 0; b[:] = 0
            fast(p(act), K, p(wb), K, p(a), N, M, 512, 512, 256, 0, 8, 16)
            slow(p(act), K, p(wb), K, p(b), N, M, 512, 512, 256, 0, 8, 16)
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
