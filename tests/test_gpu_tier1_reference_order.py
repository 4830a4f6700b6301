"""Tier 1 (the reference's own operator API) in REFERENCE ORDER, and a whole model through it with the host "as is".

`jh_set_option("JH_STRICT_ORDER", 1)` makes every `jh_gemm_*` of the provider API accumulate in the Panama-512 order
(gemm_reford_kernel, jh_p16.h).  Bars, all EXACT (no tolerance):

* F32xQ4, F32xF32, BF16xBF16, F32xBF16 bit-equal to tests/golden/ref_gemm_vectors.npz = outputs of the reference's
  own compiled C GEMM (vector_simd.c as it lies; no oracle in the loop), windows included (vector_simd.c:300-305,344);
* I8xQ4 bit-equal to the restated Panama provider (the C twin sums 8 lanes of 4-element int32 groups, SURVEY app. A.3,
  so the reference BINARY is not the bit-level checker for this one pair; it agrees to 2e-6);
* every dtype pair at M in {1, 5, 32}, offsets 512/512/512, row chunks, result offsets: bit-equal to the oracle;
* libjlamahost.so (csrc/host_mirror.cpp: TransformerBlock / CausalSelfAttention / MLPBlock / AbstractModel restated
  in C++, calling ONLY the provider entry points + plain host code where Java has plain Java) generates the oracle's
  greedy ids with bit-identical logits -- TINY, SMALL and a 3-layer Llama-3-8B shape, element-wise provider methods
  on the device and on the host delegate, heads on 1 and 4 pfor workers.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_gemm_vectors.npz")


@pytest.fixture()
def ops(gpu):
    from jlama_amd import _native as N
    from jlama_amd.hip_tensor_operations import HipTensorOperations
    o = HipTensorOperations()
    N.set_option("JH_STRICT_ORDER", 1)      # cleared after each test by conftest
    return o


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_reference_order_equals_the_reference_binary(ops):
    """No oracle in this test: libjlamahip.so against the vectors the compiled reference produced."""
    from jlama_amd import _native as N
    from jlama_amd.jq4 import Tensor
    g = np.load(GOLDEN)
    B = Tensor(N.DT_Q4, np.ascontiguousarray(g["nib"]), np.ascontiguousarray(g["scales"]))
    X = Tensor.f32(g["x"])
    n, k = B.rows, B.cols
    for registered in (False, True):
        if registered:
            ops.registerModelTensor(B)
        R = Tensor.zeros(1, n)
        ops.batchDotProduct(R, X, B, 0, 0, k)
        np.testing.assert_array_equal(_bits(R.data), _bits(g["f32q4_full"]))
        R = Tensor.zeros(1, 96)
        ops.batchDotProduct(R, X, B, 512, 512, 512, 0, 32, 64)
        np.testing.assert_array_equal(_bits(R.data[:, 32:96]), _bits(g["f32q4_window"][:, 32:96]))
        assert (R.data[:, :32] == 0).all()
    Q, KP = Tensor.f32(g["f32_q"]), Tensor.f32(g["f32_kpage"])
    R = Tensor.zeros(1, 48)
    ops.batchDotProduct(R, Q, KP, 128, 128, 128)
    np.testing.assert_array_equal(_bits(R.data), _bits(g["f32_scores"]))
    WB = Tensor(N.DT_BF16, np.ascontiguousarray(g["w_bf16"]))
    XB = ops.quantize(X, N.DT_BF16, 0, X.cols)
    np.testing.assert_array_equal(XB.data, g["x_bf16"])
    R = Tensor.zeros(1, n)
    ops.batchDotProduct(R, XB, WB, 0, 0, k)
    np.testing.assert_array_equal(_bits(R.data), _bits(g["bf16_full"]))
    R = Tensor.zeros(1, n)
    ops.batchDotProduct(R, X, WB, 0, 0, k)
    np.testing.assert_array_equal(_bits(R.data), _bits(g["f32bf16_full"]))
    R = Tensor.zeros(1, 96)
    ops.batchDotProduct(R, XB, WB, 512, 512, 512, 0, 32, 64)
    np.testing.assert_array_equal(_bits(R.data[:, 32:96]), _bits(g["bf16_window"][:, 32:96]))
    # I8 x Q4: the codes the reference consumed, then the Panama order (2e-6 from the C twin's grouping, by construction)
    A = ops.quantize(X, 2, 0, X.cols)
    np.testing.assert_array_equal(A.data, g["aq"])
    R = Tensor.zeros(1, n)
    ops.batchDotProduct(R, A, B, 0, 0, k)
    assert np.abs(R.data - g["q8q4_full"]).max() <= 1e-5 * np.abs(g["q8q4_full"]).max()


@pytest.mark.parametrize("pair", ["i8_q4", "f32_q4", "f32_f32", "bf16_bf16", "f32_bf16"])
@pytest.mark.parametrize("m", [1, 5, 32])
def test_reference_order_equals_the_oracle_every_pair_and_window(ops, oracle, pair, m):
    from jlama_amd.jq4 import Tensor
    rng = np.random.default_rng(600 + m)
    k, n = 1024, 128
    a = rng.uniform(-1, 100, (m, k)).astype(np.float32)
    a[m // 2] = rng.standard_normal(k)
    w = rng.uniform(0, 1, (n, k)).astype(np.float32)
    w[::3] *= -1
    if pair == "i8_q4":
        A, B = ops.quantize(Tensor.f32(a), 2, 0, k), Tensor.q4(w)
        ref = lambda **kw: oracle.gemm_i8q4(A.data, A.scales, B.data, B.scales, **kw)
    elif pair == "f32_q4":
        A, B = Tensor.f32(a), Tensor.q4(w)
        ref = lambda **kw: oracle.gemm_f32q4(a, B.data, B.scales, **kw)
    elif pair == "f32_f32":
        A, B = Tensor.f32(a), Tensor.f32(w)
        ref = lambda **kw: oracle.gemm_f32(a, w, **kw)
    elif pair == "bf16_bf16":
        A, B = Tensor.bf16(a), Tensor.bf16(w)
        ref = lambda **kw: oracle.gemm_bf16(A.data, B.data, **kw)
    else:
        A, B = Tensor.f32(a), Tensor.bf16(w)
        ref = lambda **kw: oracle.gemm_f32bf16(a, B.data, **kw)
    for registered in (False, True):
        if registered:
            ops.registerModelTensor(B)
        R = Tensor.zeros(m, n)
        ops.batchDotProduct(R, A, B, 0, 0, k)
        np.testing.assert_array_equal(_bits(R.data), _bits(ref()))
        # offsets 512/512/512 (TestOperations.java:151-187), rows [32, 96)
        R = Tensor.zeros(m, n)
        ops.batchDotProduct(R, A, B, 512, 512, 512, 0, 32, 64)
        want = ref(aColOff=512, bColOff=512, K=512, bRowOff=32, N=64, out=np.zeros((m, n), np.float32))
        np.testing.assert_array_equal(_bits(R.data), _bits(want))
        # result offset (TestOperations.java:533-552): rows [0, 64) of B written at columns [64, 128)
        R = Tensor.zeros(m, n)
        ops.batchDotProduct(R, A, B, 0, 0, k, 64, 0, 64)
        want = ref(rRowOff=64, bRowOff=0, N=64, out=np.zeros((m, n), np.float32))
        np.testing.assert_array_equal(_bits(R.data), _bits(want))


def test_reference_order_refuses_what_panama_cannot_do(ops):
    """Panama reads whole 16-lane vectors (PTO:1086-1102 has no tail loop): K % 16 != 0 has no reference order to follow."""
    from jlama_amd import _native as N
    from jlama_amd.jq4 import Tensor
    A, B = Tensor.f32(np.ones((1, 40), np.float32)), Tensor.f32(np.ones((4, 40), np.float32))
    with pytest.raises(N.UnsupportedOperation):
        ops.batchDotProduct(Tensor.zeros(1, 4), A, B, 0, 0, 40)


def _host_vs_oracle(oracle, cfg, w, prompt, n_gen, **kw):
    from jlama_amd import _native as N
    from jlama_amd.host_mirror import HostAsIsModel
    N.set_option("JH_STRICT_ORDER", 1)
    om = oracle.OracleModel(cfg, w)
    want, logits_o, _ = om.session().generate(prompt, n_gen)
    hm = HostAsIsModel(cfg, w, **kw)
    res = hm.generate(prompt, n_gen)
    hm.close()
    return res, want, logits_o


@pytest.mark.parametrize("ew_device,threads", [(True, 1), (False, 1), (True, 4), (False, 4)])
def test_host_as_is_through_the_provider_api_tiny(gpu, oracle, ew_device, threads):
    from jlama_amd import synthetic as S
    cfg = dict(S.TINY)
    w = S.make_weights(cfg, seed=0)
    prompt = S.prompt_tokens(cfg, n=40, seed=7)     # 41 rows: crosses a KV page of the TINY geometry or not, both are exercised by SMALL below
    res, want, logits_o = _host_vs_oracle(oracle, cfg, w, prompt, 24, elementwise_on_device=ew_device, threads=threads)
    np.testing.assert_array_equal(res["tokens"], want)
    np.testing.assert_array_equal(_bits(res["logits"]), _bits(logits_o))
    assert res["provider_calls"] > 0


def test_host_as_is_small_crossing_kv_pages(gpu, oracle):
    """SMALL (3 layers, 8 heads / 2 kv heads, hs 128) with 2 KiB KV pages... the page size is forced small so that the per-page
    score / saxpy calls of CausalSelfAttention.java:324-354 run over several pages with a ragged last one."""
    from jlama_amd import _native as N, synthetic as S
    from jlama_amd.host_mirror import HostAsIsModel
    cfg = dict(S.SMALL)
    w = S.make_weights(cfg, seed=3)
    prompt = S.prompt_tokens(cfg, n=37, seed=11)
    N.set_option("JH_STRICT_ORDER", 1)
    page_bytes = 3 * 2 * 4 * cfg["n_kv_heads"] * cfg["head_size"] * 16     # 3 layers x 16 positions per page
    om = oracle.OracleModel(cfg, w)
    so = om.session(max_page_bytes=page_bytes)
    want, logits_o, _ = so.generate(prompt, 20)
    hm = HostAsIsModel(cfg, w, elementwise_on_device=True, threads=2, max_page_bytes=page_bytes)
    assert hm.page_info()[:2] == tuple(so.page_info()[:2]) and hm.page_info()[1] == 16
    res = hm.generate(prompt, 20)
    np.testing.assert_array_equal(res["tokens"], want)
    np.testing.assert_array_equal(_bits(res["logits"]), _bits(logits_o))
    # the same model through the RESIDENT path (Tier 2, reference order): the two tiers agree bit for bit
    from jlama_amd.model import HipLlamaModel
    m2 = HipLlamaModel(cfg, w)
    s2 = m2.session(128)
    s2.set_strict(True)
    r2 = s2.generate(prompt, prompt.size + 19)
    np.testing.assert_array_equal(r2["tokens"], res["tokens"])
    np.testing.assert_array_equal(_bits(s2.logits()), _bits(res["logits"]))
    hm.close()
    m2.close()


def test_host_as_is_three_layers_of_llama3_8b(gpu, oracle):
    """The metric's own shapes (E 4096, H 14336, 32 / 8 heads of 128, V 128256), 3 layers: prompt of 33 rows (crosses the 32-position
    KV page of the 8B geometry... with 3 layers the geometry search gives its own page size, asserted equal to the oracle's) + 6 steps."""
    from jlama_amd import synthetic as S
    cfg = dict(S.LLAMA3_8B)
    cfg["n_layers"] = 3
    w = S.make_weights(cfg, seed=1, quantize=oracle.q4_quantize)
    prompt = S.prompt_tokens(cfg, n=32, seed=5)
    res, want, logits_o = _host_vs_oracle(oracle, cfg, w, prompt, 6, elementwise_on_device=False, threads=4)
    np.testing.assert_array_equal(res["tokens"], want)
    np.testing.assert_array_equal(_bits(res["logits"]), _bits(logits_o))


def test_host_as_is_bf16_model(gpu, oracle):
    """Config 4's dtype flow through the provider API: quantize(F32 -> BF16), BF16xBF16 projections, F32xBF16 LM head."""
    from jlama_amd import _native as N, synthetic as S
    cfg = dict(S.SMALL)
    cfg["weight_dtype"] = N.DT_BF16
    w = S.make_weights(cfg, seed=2)
    prompt = S.prompt_tokens(cfg, n=20, seed=13)
    res, want, logits_o = _host_vs_oracle(oracle, cfg, w, prompt, 12, elementwise_on_device=True, threads=1)
    np.testing.assert_array_equal(res["tokens"], want)
    np.testing.assert_array_equal(_bits(res["logits"]), _bits(logits_o))


def test_host_as_is_order_free_kernels_stay_inside_the_envelope(gpu, oracle):
    """Without JH_STRICT_ORDER the same host runs on the order-free Tier-1 kernels: ids equal until a near-tie, logits inside the
    Q8 noise floor of tests/test_gpu_model.py (4e-2 absolute on these shapes)."""
    from jlama_amd import synthetic as S
    from jlama_amd.host_mirror import HostAsIsModel
    cfg = dict(S.TINY)
    w = S.make_weights(cfg, seed=0)
    prompt = S.prompt_tokens(cfg, n=16, seed=1)
    om = oracle.OracleModel(cfg, w)
    want, logits_o, _ = om.session().generate(prompt, 16)
    hm = HostAsIsModel(cfg, w, elementwise_on_device=True, threads=1)
    res = hm.generate(prompt, 16)
    hm.close()
    agree = int((res["tokens"] == want).cumprod().sum())
    assert agree >= 4, (res["tokens"], want)
    if agree == want.size:
        assert np.abs(res["logits"] - logits_o).max() <= 4e-2
