"""Reference order for the dense BF16 path (BASELINE configs[3], jlama_amd/csrc/jh_bf16r.h): every float accumulation of a
BF16 session in the order of the reference's Panama provider -- GemmerBF16 (PanamaTensorOperations.java:1279-1311: 16 lanes,
two fmas per 32-element step, halving tree) for the projections, GemmerF32BF16 (:1511-1538) for the LM head, the
reference-order attention of jh_p16.h in between.  With the order fixed there is no tolerance: stage taps of every layer,
prompt rows, logits and greedy ids equal the oracle BIT FOR BIT."""
import numpy as np
import pytest

from jlama_amd import _native as _N

pytestmark = pytest.mark.gpu


def _bf16_pair(cfg, seed, oracle):
    from jlama_amd import synthetic as S
    from jlama_amd.model import HipLlamaModel
    cfg = dict(cfg)
    cfg["weight_dtype"] = _N.DT_BF16
    w = S.make_weights(cfg, seed=seed)
    return cfg, HipLlamaModel(cfg, w), oracle.OracleModel(cfg, w), w


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("cfgname", ["TINY", "SMALL"])
def test_bf16_reference_order_is_bit_identical(gpu, oracle, cfgname):
    from jlama_amd import synthetic as S
    cfg, hm, om, w = _bf16_pair(getattr(S, cfgname), 1, oracle)
    prompt = S.prompt_tokens(cfg, n=40, seed=11)          # 41 positions: crosses a KV context page for SMALL
    E, A, KV = cfg["embedding_length"], cfg["n_heads"] * cfg["head_size"], cfg["n_kv_heads"] * cfg["head_size"]
    for layer in range(cfg["n_layers"]):
        hs, os_ = hm.session(64), om.session()
        hs.set_strict(True)
        hs.set_tap_layer(layer)
        os_.set_tap_layer(layer)
        for i, t in enumerate(prompt[:9]):
            hs.forward([t], i, want_output=False)
            os_.forward([t], i)
        for name, n in [("input_emb", E), ("query", A), ("key", KV), ("value", KV), ("query+rope", A), ("key+rope", KV),
                        ("after_attention", A), ("post_ff_res", E)]:
            np.testing.assert_array_equal(_bits(hs.tap(name, n)), _bits(os_.tap(name, n)), err_msg=f"layer {layer} tap {name}")
    hs, os_ = hm.session(200), om.session()
    hs.set_strict(True)
    out_h, out_o = hs.batch_forward(prompt, 0), os_.forward(prompt, 0)
    np.testing.assert_array_equal(_bits(out_h), _bits(out_o))
    tok_h, lh = hs.sample(0.0, 0.5, want_logits=True)
    tok_o, lo = om.sample(out_o[-1])
    np.testing.assert_array_equal(_bits(lh), _bits(lo))
    assert tok_h == tok_o
    n_gen = 100
    ids_h = hs.decode_n(tok_h, prompt.size, n_gen)        # hipGraph replay of the reference-order kernels
    lh_last = hs.logits()
    ids_o, tok = [], tok_o
    for i in range(n_gen):
        xo = os_.forward([tok], prompt.size + i)
        tok, lo = om.sample(xo[-1])
        ids_o.append(tok)
    np.testing.assert_array_equal(ids_h, np.array(ids_o, dtype=np.int32))
    np.testing.assert_array_equal(_bits(lh_last), _bits(lo))
    # back to the order-free kernels on the same session (graphs are re-captured): close, not equal
    hs.set_strict(False)
    hs.batch_forward(prompt, 0)
    _, lf = hs.sample(0.0, 0.5, want_logits=True)
    first_logits = om.sample(out_o[-1])[1]
    assert np.abs(lf - first_logits).max() <= 2e-2 * max(1.0, float(np.abs(first_logits).max()))


def test_bf16_reference_order_real_shapes(gpu, oracle):
    """Mistral-7B's shapes (E 4096, H 14336: 32 / 112 groups of 128 per row, GQA 4, head size 128), three layers, reduced
    vocabulary: 37 prompt rows (crosses a 32-row KV context page), logits and 24 greedy ids bit-identical."""
    from jlama_amd import synthetic as S
    base = dict(S.MISTRAL_7B)
    base.update(n_layers=3, vocab_size=2048, context_length=512, bos_token=1)
    cfg, hm, om, w = _bf16_pair(base, 7, oracle)
    prompt = S.prompt_tokens(cfg, n=36, seed=8)
    hs, os_ = hm.session(96), om.session()
    hs.set_strict(True)
    got, want = hs.batch_forward(prompt, 0), os_.forward(prompt, 0)
    np.testing.assert_array_equal(_bits(got), _bits(want))
    th, lh = hs.sample(0.0, 0.5, want_logits=True)
    to, lo = om.sample(want[-1])
    np.testing.assert_array_equal(_bits(lh), _bits(lo))
    assert th == to
    ids = hs.decode_n(th, prompt.size, 24)
    tok = to
    for i, g in enumerate(ids):
        xo = os_.forward([tok], prompt.size + i)
        tok, lo = om.sample(xo[-1])
        assert g == tok, (i, g, tok)
    np.testing.assert_array_equal(_bits(hs.logits()), _bits(lo))


def test_bf16_prompt_in_batches_is_the_row_path_bit_for_bit(gpu, oracle):
    """A BF16 reference-order session walks prompts in chunks as well (gemm_bf16r_kernel: the M-row form of the same
    chains); rows, KV pages and everything decoded afterwards equal the one-position-at-a-time path bit for bit -- across
    a chunk boundary (300 = 256 + 44 rows), a ragged last row tile and a continuation at start_pos > 0."""
    from jlama_amd import synthetic as S
    cfg, hm, om, w = _bf16_pair(S.SMALL, 3, oracle)
    prompt = S.prompt_tokens(cfg, n=300, seed=19)
    _N.set_option("JH_PREFILL_BATCH_MIN", "0")            # rows one at a time
    s_row = hm.session(512)
    s_row.set_strict(True)
    rows = s_row.forward(prompt, 0)
    _N.clear_options()
    s_bat = hm.session(512)
    s_bat.set_strict(True)
    bat = s_bat.forward(prompt, 0)
    np.testing.assert_array_equal(_bits(bat), _bits(rows))
    s_two = hm.session(512)
    s_two.set_strict(True)
    a = s_two.forward(prompt[:100], 0)
    b = s_two.forward(prompt[100:], 100)
    np.testing.assert_array_equal(_bits(np.concatenate([a, b])), _bits(rows))
    want = om.session().forward(prompt[:48], 0)
    np.testing.assert_array_equal(_bits(bat[:48]), _bits(want))
    firsts, logits = [], []
    for s in (s_row, s_bat, s_two):
        t, l = s.sample(0.0, 0.5, want_logits=True)
        firsts.append(t); logits.append(l)
    assert firsts[0] == firsts[1] == firsts[2]
    np.testing.assert_array_equal(_bits(logits[0]), _bits(logits[1]))
    np.testing.assert_array_equal(_bits(logits[0]), _bits(logits[2]))
    ids = [list(s.decode_n(firsts[0], prompt.size, 24)) for s in (s_row, s_bat, s_two)]
    assert ids[0] == ids[1] == ids[2]
    np.testing.assert_array_equal(_bits(s_row.logits()), _bits(s_bat.logits()))
