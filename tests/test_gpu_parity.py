"""GPU parity, the two instruments that take the Q8 quantizer's step function out of the comparison:

1. STRICT ORDER (jh_session_set_strict): every float accumulation of the decode path in the reference's Panama-512 order
   (jlama_amd/csrc/jh_t16.h, jh_p16.h).  With identical summation order there is no noise floor to hide behind: stage taps of
   EVERY layer, logits and greedy ids must equal the oracle BIT FOR BIT.
2. PER-LAYER TEACHER FORCING of the fast kernels: layer l of the oracle is fed the GPU's own `input_emb` tap of layer l
   (all positions), so each layer is compared in isolation -- no cascade from earlier layers.  Within ONE layer the fast
   kernels can still differ from the oracle by a single I8 code flip downstream of a 1e-7 summation-order difference
   (o-projection input, down-projection input); such a flip moves the layer output by (max|a|/127)*|w| -- measured
   2e-4 .. 6e-3 of the row scale on these models, and two flips in one layer add.  An addressing bug (wrong KV page for
   rel_layer > 0, RoPE row off by one, wrong head) moves it by O(1).  So: every (layer, row) <= 1.5e-2 of the row scale,
   and most of them (no flip) <= 1e-5, with every layer's median <= 1e-5.
"""
import numpy as np
import pytest

from jlama_amd import _native as _N

pytestmark = pytest.mark.gpu

FLIP_TOL = 1.5e-2    # one or two I8 code flips inside a layer (see module docstring)
NOFLIP_TOL = 1e-5    # float summation order only


def _pair(cfg, seed, oracle, layer_range=None):
    from jlama_amd import synthetic as S
    from jlama_amd.model import HipLlamaModel
    w = S.make_weights(cfg, seed=seed, quantize=oracle.q4_quantize)   # compiled quantizer, bit-equal to jq4 (test_oracle.py)
    return HipLlamaModel(cfg, w, layer_range=layer_range), oracle.OracleModel(cfg, w, layer_range=layer_range), w


_REAL = {}


def _real(name, oracle):
    """Benchmark-config shapes (E, H, heads, head size) with few layers and a reduced vocabulary; built once per module."""
    if name not in _REAL:
        from jlama_amd import synthetic as S
        cfg = dict(getattr(S, name))
        cfg.update(n_layers=1 if name == "LLAMA3_70B" else 3, vocab_size=2048, context_length=512, bos_token=1)
        cfg.pop("tied", None)
        _REAL[name] = (cfg,) + _pair(cfg, 7, oracle)
    return _REAL[name]


def _rowrel(got, want):
    """max |diff| per row, relative to the row's max |want|."""
    return np.abs(got - want).max(axis=-1) / (np.abs(want).max(axis=-1) + 1e-30)


def layer_teacher_forced(hm, oracle, cfg, w, prompt, max_ctx, strict=False):
    """For every layer l: run the GPU over the prompt one position at a time with the taps of layer l armed, collect its
    input rows and output rows, then run the ORACLE's layer l alone on those input rows.  Returns rel[l, row]."""
    E, L = cfg["embedding_length"], cfg["n_layers"]
    rel = np.zeros((L, prompt.size))
    for layer in range(L):
        hs = hm.session(max_ctx)
        if strict:
            hs.set_strict(True)
        hs.set_tap_layer(layer)
        xin, xout = [], []
        for i, t in enumerate(prompt):
            hs.forward([t], i, want_output=False)
            xin.append(hs.tap("input_emb", E))
            xout.append(hs.tap("post_ff_res", E))
        hs.close()
        om_l = oracle.OracleModel(cfg, w, layer_range=(layer, layer + 1))
        want = om_l.session().forward(None, 0, x=np.stack(xin))
        got = np.stack(xout)
        if strict:
            np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32), err_msg=f"layer {layer}")
        rel[layer] = _rowrel(got, want)
    return rel


@pytest.mark.parametrize("cfgname", ["TINY", "SMALL"])
def test_every_layer_teacher_forced(gpu, oracle, cfgname):
    from jlama_amd import synthetic as S
    cfg = dict(getattr(S, cfgname))
    hm, om, w = _pair(cfg, 0, oracle)
    prompt = S.prompt_tokens(cfg, n=40, seed=5)        # 41 positions: crosses a KV context page (32 rows) for SMALL
    rel = layer_teacher_forced(hm, oracle, cfg, w, prompt, 64)
    assert rel.max() <= FLIP_TOL, (rel.max(), np.unravel_index(rel.argmax(), rel.shape))
    assert (rel <= NOFLIP_TOL).mean() >= 0.5, (rel <= NOFLIP_TOL).mean()
    for layer in range(cfg["n_layers"]):                # no layer may be systematically worse than the others
        assert np.median(rel[layer]) <= NOFLIP_TOL or rel[layer].min() <= NOFLIP_TOL, (layer, np.median(rel[layer]))


@pytest.mark.parametrize("name", ["LLAMA32_1B", "LLAMA3_8B"])
def test_every_layer_teacher_forced_real_shapes(gpu, oracle, name):
    """The kernel instantiations the benchmark configs use (head size 64 / 128, NB = 1 / 2 / 7 blocks per lane), three
    layers so that rel_layer > 0 pages and the per-layer weight pointers are exercised; reduced vocabulary."""
    from jlama_amd import synthetic as S
    cfg, hm, om, w = _real(name, oracle)
    prompt = S.prompt_tokens(cfg, n=35, seed=8)
    rel = layer_teacher_forced(hm, oracle, cfg, w, prompt, 64)
    assert rel.max() <= FLIP_TOL, (rel.max(), np.unravel_index(rel.argmax(), rel.shape))
    assert (rel <= NOFLIP_TOL).mean() >= 0.5
    for layer in range(cfg["n_layers"]):
        assert np.median(rel[layer]) <= NOFLIP_TOL or rel[layer].min() <= NOFLIP_TOL, (layer, np.median(rel[layer]))


@pytest.mark.parametrize("cfgname", ["TINY", "SMALL"])
def test_strict_order_is_bit_identical(gpu, oracle, cfgname):
    """Strict order: taps of every layer, the forward output of every prompt row, the logits of every decode step and the
    greedy ids -- all equal to the oracle bit for bit, free-running (no teacher forcing needed: nothing diverges)."""
    from jlama_amd import synthetic as S
    cfg = dict(getattr(S, cfgname))
    hm, om, w = _pair(cfg, 1, oracle)
    prompt = S.prompt_tokens(cfg, n=40, seed=11)
    E, A, KV = cfg["embedding_length"], cfg["n_heads"] * cfg["head_size"], cfg["n_kv_heads"] * cfg["head_size"]
    # stage taps, every layer, positions 0..8
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
            np.testing.assert_array_equal(hs.tap(name, n).view(np.uint32), os_.tap(name, n).view(np.uint32),
                                          err_msg=f"layer {layer} tap {name}")
    # prompt (row by row on the GPU, one batch in the oracle: same per-row arithmetic), then free-running greedy decode
    hs, os_ = hm.session(200), om.session()
    hs.set_strict(True)
    out_h, out_o = hs.batch_forward(prompt, 0), os_.forward(prompt, 0)
    np.testing.assert_array_equal(out_h.view(np.uint32), out_o.view(np.uint32))
    tok_h, lh = hs.sample(0.0, 0.5, want_logits=True)
    tok_o, lo = om.sample(out_o[-1])
    np.testing.assert_array_equal(lh.view(np.uint32), lo.view(np.uint32))
    assert tok_h == tok_o
    n_gen = 100
    ids_h = hs.decode_n(tok_h, prompt.size, n_gen)                 # hipGraph replay of the strict kernels
    lh_last = hs.logits()
    ids_o, tok = [], tok_o
    for i in range(n_gen):
        xo = os_.forward([tok], prompt.size + i)
        tok, lo = om.sample(xo[-1])
        ids_o.append(tok)
    np.testing.assert_array_equal(ids_h, np.array(ids_o, dtype=np.int32))
    np.testing.assert_array_equal(lh_last.view(np.uint32), lo.view(np.uint32))
    # the fast kernels on the same session (mode switch re-captures the graphs): same ids wherever the decision margin
    # is above the Q8 noise floor -- i.e. summation order is the ONLY difference between the two modes
    hs.set_strict(False)
    hs.batch_forward(prompt, 0)
    fast_first = hs.sample()
    ids_f = hs.decode_n(fast_first, prompt.size, 20)
    agree = int((np.concatenate([[fast_first], ids_f]) == np.concatenate([[tok_o], ids_o[:20]])).cumprod().sum())
    assert agree >= 8, agree


@pytest.mark.parametrize("name", ["LLAMA32_1B", "LLAMA3_8B", "LLAMA3_70B"])
def test_strict_order_real_shapes(gpu, oracle, name):
    """Strict order at the benchmark configs' shapes (reduced vocabulary): bit-identical prompt rows, logits and 24 greedy
    ids; and every layer teacher-forced in strict mode is bit-identical too."""
    from jlama_amd import synthetic as S
    cfg, hm, om, w = _real(name, oracle)
    prompt = S.prompt_tokens(cfg, n=36, seed=8)      # 37 rows: crosses a 32-row KV context page
    hs, os_ = hm.session(96), om.session()
    hs.set_strict(True)
    got, want = hs.batch_forward(prompt, 0), os_.forward(prompt, 0)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    th, lh = hs.sample(0.0, 0.5, want_logits=True)
    to, lo = om.sample(want[-1])
    np.testing.assert_array_equal(lh.view(np.uint32), lo.view(np.uint32))
    assert th == to
    ids = hs.decode_n(th, prompt.size, 24)
    tok = to
    for i, g in enumerate(ids):
        xo = os_.forward([tok], prompt.size + i)
        tok, lo = om.sample(xo[-1])
        assert g == tok, (i, g, tok)
    np.testing.assert_array_equal(hs.logits().view(np.uint32), lo.view(np.uint32))
    if name != "LLAMA3_70B":
        layer_teacher_forced(hm, oracle, cfg, w, prompt[:12], 64, strict=True)


def test_reference_order_prefill_in_batches_is_the_row_path_bit_for_bit(gpu, oracle, monkeypatch):
    """Reference-order sessions run prompts in chunks (jh_t16.h: gemm_t16_kernel on the F16 MFMA; jh_p16.h: rows_*_p16_kernel, the
    attention of 8-row tiles).  Every (prompt row, weight row) pair keeps its own 16-lane chain fed in the same order, every
    (row, head) its own score dots, float sum and value chains, so the rows, the KV
    pages they leave behind and everything decoded afterwards must equal the one-position-at-a-time path bit for bit --
    across a chunk boundary (300 rows = 256 + 44), a ragged last row tile, KV context pages, and a continuation at
    start_pos > 0."""
    from jlama_amd import synthetic as S
    cfg = dict(S.SMALL)
    hm, om, w = _pair(cfg, 3, oracle)
    prompt = S.prompt_tokens(cfg, n=300, seed=19)
    _N.set_option("JH_PREFILL_BATCH_MIN", "0")            # rows one at a time
    s_row = hm.session(512)
    s_row.set_strict(True)
    rows = s_row.forward(prompt, 0)
    _N.clear_options()
    s_bat = hm.session(512)
    s_bat.set_strict(True)
    bat = s_bat.forward(prompt, 0)
    np.testing.assert_array_equal(bat.view(np.uint32), rows.view(np.uint32))
    s_two = hm.session(512)
    s_two.set_strict(True)
    a = s_two.forward(prompt[:100], 0)
    b = s_two.forward(prompt[100:], 100)
    np.testing.assert_array_equal(np.concatenate([a, b]).view(np.uint32), rows.view(np.uint32))
    # one or two rows of a ragged last row tile leave the GEMM for the decode GEMVs (prefill.hip: t16_tail_rows): 129 = 16 tiles + 1,
    # 170 = 21 tiles + 2, then a single row with no tile at all; JH_T16_TAIL_ROWS=0 keeps every row in the GEMM
    # (the library diverts them only where the ragged tile would cost an extra pass over the machine's workgroup slots -- never
    # on this small shape: JH_T16_TAIL_FORCE)
    _N.set_option("JH_T16_TAIL_FORCE", "1")
    s_tail = hm.session(512)
    s_tail.set_strict(True)
    parts = [s_tail.forward(prompt[:129], 0), s_tail.forward(prompt[129:299], 129), s_tail.forward(prompt[299:], 299)]
    np.testing.assert_array_equal(np.concatenate(parts).view(np.uint32), rows.view(np.uint32))
    s_tail4 = hm.session(512)
    s_tail4.set_strict(True)                                   # 3 and 4 tail rows (the default limit)
    parts = [s_tail4.forward(prompt[:131], 0), s_tail4.forward(prompt[131:231], 131)]
    np.testing.assert_array_equal(np.concatenate(parts).view(np.uint32), rows[:231].view(np.uint32))
    s_tail4.close()
    _N.clear_options()
    _N.set_option("JH_T16_TAIL_ROWS", "0")
    s_gemm = hm.session(512)
    s_gemm.set_strict(True)
    np.testing.assert_array_equal(s_gemm.forward(prompt[:129], 0).view(np.uint32), rows[:129].view(np.uint32))
    s_gemm.close()
    _N.clear_options()
    _N.set_option("JH_T16_GEMM32", "0")                    # the 16x16x32 form of the GEMM (kept for A/B): same bits
    s_16 = hm.session(512)
    s_16.set_strict(True)
    np.testing.assert_array_equal(s_16.forward(prompt[:129], 0).view(np.uint32), rows[:129].view(np.uint32))
    s_16.close()
    _N.clear_options()
    _N.set_option("JH_T16_TAIL_FORCE", "1")
    _N.set_option("JH_PREFILL_BATCH_MIN", "1")            # a 2-row chunk in the batched path: tail rows only, no GEMM launch at all
    s_few = hm.session(512)
    s_few.set_strict(True)
    few = [s_few.forward(prompt[:40], 0), s_few.forward(prompt[40:42], 40), s_few.forward(prompt[42:43], 42)]
    np.testing.assert_array_equal(np.concatenate(few).view(np.uint32), rows[:43].view(np.uint32))
    s_few.close()
    _N.clear_options()
    want = om.session().forward(prompt[:48], 0)                 # the oracle on the first rows (it is the slow one)
    np.testing.assert_array_equal(bat[:48].view(np.uint32), want.view(np.uint32))
    firsts, logits = [], []
    for s in (s_row, s_bat, s_two, s_tail):
        t, l = s.sample(0.0, 0.5, want_logits=True)
        firsts.append(t); logits.append(l)
    assert firsts[0] == firsts[1] == firsts[2] == firsts[3]
    for l in logits[1:]:
        np.testing.assert_array_equal(logits[0].view(np.uint32), l.view(np.uint32))
    ids = [list(s.decode_n(firsts[0], prompt.size, 24)) for s in (s_row, s_bat, s_two, s_tail)]   # reads the KV pages of all 300 positions
    assert ids[0] == ids[1] == ids[2] == ids[3]
    np.testing.assert_array_equal(s_row.logits().view(np.uint32), s_bat.logits().view(np.uint32))


def test_reference_order_only_model_releases_the_row_major_weights(gpu, oracle):
    """JH_STRICT_ONLY=1: once a reference-order session has published the T16 / P16T operand copies, the row-major
    projection nibbles are released -- those copies are then the weight.  Same rows, logits and ids bit for bit; HBM really comes
    back; order-free use, leaving reference order and replacing a weight fail loudly instead of reading freed memory."""
    from jlama_amd import synthetic as S
    from jlama_amd.model import HipLlamaModel
    cfg, hm, om, w = _real("LLAMA3_8B", oracle)                  # whole T16 tiles: the prompt goes through the MFMA GEMM
    prompt = S.prompt_tokens(cfg, n=41, seed=5)                  # 41 = 5 row tiles + 1 tail row through the decode GEMVs
    s0 = hm.session(256)
    s0.set_strict(True)
    rows0 = s0.forward(prompt, 0)
    t0, l0 = s0.sample(0.0, 0.5, want_logits=True)
    ids0 = list(s0.decode_n(t0, prompt.size, 12))
    s0.close()
    _N.set_option("JH_STRICT_ONLY", "1")
    try:
        m2 = HipLlamaModel(cfg, w)
        s = m2.session(256)
        assert m2.released_bytes() == 0
        s.set_strict(True)                                       # publishes the copies, then releases the row-major nibbles
        E, H = cfg["embedding_length"], cfg["hidden_length"]
        A, KV = cfg["n_heads"] * cfg["head_size"], cfg["n_kv_heads"] * cfg["head_size"]
        nibbles = cfg["n_layers"] * ((A + 2 * KV) * E + E * A + 3 * H * E) // 2
        assert m2.released_bytes() >= nibbles                    # (+ the MFMA-ordered prompt copies, had an order-free session made them)
        rows = s.forward(prompt, 0)
        np.testing.assert_array_equal(rows.view(np.uint32), rows0.view(np.uint32))
        t, l = s.sample(0.0, 0.5, want_logits=True)
        assert t == t0
        np.testing.assert_array_equal(l.view(np.uint32), l0.view(np.uint32))
        assert list(s.decode_n(t, prompt.size, 12)) == ids0
        s_other = m2.session(256)                                # sessions are born order-free: using one must fail, not fault
        with pytest.raises(_N.UnsupportedOperation, match="JH_STRICT_ONLY"):
            s_other.forward(prompt[:4], 0)
        with pytest.raises(_N.UnsupportedOperation, match="JH_STRICT_ONLY"):
            s_other.decode_n(1, 0, 2)
        s_other.set_strict(True)                                 # ... and entering reference order makes it usable
        np.testing.assert_array_equal(s_other.forward(prompt, 0).view(np.uint32), rows0.view(np.uint32))
        with pytest.raises(_N.UnsupportedOperation, match="JH_STRICT_ONLY"):
            s.set_strict(False)
        with pytest.raises(_N.UnsupportedOperation, match="immutable"):
            key = next(k for k in w if k[0] == 0)
            m2.set_weight(key[0], key[1], w[key])
        s_other.close(); s.close()
    finally:
        _N.clear_options()


def test_strict_mode_accepts_bf16_models(gpu):
    """Reference order exists for both weight formats of BASELINE's configs: JQ4 (jh_t16.h / jh_p16.h) and dense BF16
    (jh_bf16r.h; bit-identity asserted in tests/test_gpu_bf16_reference_order.py)."""
    from jlama_amd import _native as N, synthetic as S
    from jlama_amd.model import HipLlamaModel
    cfg = dict(S.TINY)
    cfg["weight_dtype"] = N.DT_BF16
    hs = HipLlamaModel(cfg, S.make_weights(cfg, seed=0)).session(16)
    hs.set_strict(True)
    hs.set_strict(False)


def test_device_loop_stops_at_eos(gpu, oracle):
    """AbstractModel.java:600-603: the step that samples a stop token is the last one.  The device loop freezes its state
    there (finish_token_kernel) and the host stops feeding; the ids up to and including the stop token come back."""
    from jlama_amd import synthetic as S
    cfg = dict(S.TINY)
    hm, om, _ = _pair(cfg, 2, oracle)
    prompt = S.prompt_tokens(cfg, n=20, seed=2)
    s = hm.session(250)
    s.batch_forward(prompt, 0)
    first = s.sample()
    free = s.decode_n(first, prompt.size, 200)
    assert free.size == 200 and s.decode_generated() == 200
    for k in (3, 37, 150):                      # inside the first chunk, after two chunks, deep into the loop
        stop = int(free[k])
        kfirst = int(np.nonzero(free == stop)[0][0])      # the id may occur earlier than step k
        s2 = hm.session(250)
        s2.set_eos([stop, cfg["vocab_size"] - 1])
        s2.batch_forward(prompt, 0)
        assert s2.sample() == first
        got = s2.decode_n(first, prompt.size, 200)
        assert got.size == kfirst + 1 and s2.decode_generated() == kfirst + 1, (k, kfirst, got.size)
        np.testing.assert_array_equal(got, free[:kfirst + 1])
        # the session is reusable afterwards and the set can be cleared
        s2.set_eos([])
        again = s2.decode_n(first, prompt.size, 50)
        np.testing.assert_array_equal(again, free[:50])
    # generate(): same contract at the host mirror's level
    res = hm.session(250).generate(prompt, prompt.size + 200, eos_tokens=(int(free[37]),))
    kf = int(np.nonzero(free == free[37])[0][0])
    if first != int(free[37]):
        np.testing.assert_array_equal(res["tokens"], np.concatenate([[first], free[:kf + 1]]))


def test_first_sampled_token_is_never_tested_against_the_stop_set(gpu, oracle):
    """AbstractModel.java:573-603: the token sampled from the prompt is always forwarded; only ids sampled INSIDE the loop
    are compared with eosTokens.  Device loop, host loop and the reference order (oracle ids with the same rule) agree, and an
    unchanged stop list re-captures nothing."""
    from jlama_amd import synthetic as S
    cfg = dict(S.TINY)
    hm, om, _ = _pair(cfg, 2, oracle)
    prompt = S.prompt_tokens(cfg, n=20, seed=2)
    want, _, _ = om.session().generate(prompt, 60)            # free-running oracle ids: want[0] is sampled from the prompt
    first = int(want[0])
    # expected: stop at the first LATER occurrence of `first`, or run to the end
    later = np.nonzero(want[1:] == first)[0]
    expect = want[:later[0] + 2] if later.size else want
    for device_loop in (True, False):
        s = hm.session(128)
        s.set_strict(True)
        res = s.generate(prompt, prompt.size + 59, eos_tokens=(first,), on_device_loop=device_loop)
        np.testing.assert_array_equal(res["tokens"], expect, err_msg=f"device_loop={device_loop}")
        assert res["tokens_generated"] == expect.size - 1
        # eos_tokens=None is an empty set
        res2 = s.generate(prompt, prompt.size + 9, eos_tokens=None, on_device_loop=device_loop)
        np.testing.assert_array_equal(res2["tokens"], want[:10])
        s.close()


@pytest.mark.parametrize("vocab,T", [(512, 0.8), (20000, 0.8), (128256, 0.8), (128256, 0.05), (50257, 4.0)])
def test_device_loop_temperature_sampling_follows_the_reference_rule(gpu, oracle, vocab, T):
    """AbstractModel.sample with temperature > 0 (AbstractModel.java:471-489) inside the device loop (jh_decode_n_sampled): the
    float running sum in index order, the inverse CDF against the caller's uniform.  In reference order the sampled ids equal the
    oracle's jo_sample sequence bit for bit.  The device evaluates both accumulations with 1024 lanes (jh_seqsum.h; its integer
    rounding rule is checked on the host by tests/test_seqsum.py); vocabularies beyond 8192 ids take several scan iterations,
    T = 0.05 makes most exponentials denormal or zero, T = 4 makes the sum cross many binades."""
    from jlama_amd import synthetic as S
    cfg = dict(S.TINY)
    cfg["vocab_size"] = vocab
    hm, om, _ = _pair(cfg, 11, oracle)
    prompt = S.prompt_tokens(cfg, n=9, seed=4)
    n = 24
    u = np.random.default_rng(5).random(n + 1).astype(np.float32)
    osess = om.session()
    x = osess.forward(prompt, 0)
    tok, _ = om.sample(x[-1], T, float(u[0]))
    want = [tok]
    for i in range(n):
        x = osess.forward([tok], prompt.size + i)
        tok, _ = om.sample(x[-1], T, float(u[i + 1]))
        want.append(tok)
    s = hm.session(64)
    s.set_strict(True)
    s.batch_forward(prompt, 0)
    first = s.sample(T, float(u[0]))
    assert first == want[0]
    got = s.decode_n_sampled(first, prompt.size, n, T, u[1:])
    np.testing.assert_array_equal(got, np.array(want[1:], dtype=np.int32))
    assert len(set(want)) > 4                                   # the uniforms really pick different ids (not an argmax in disguise)
    # the same through generate(): uniforms drawn from the rng in the host loop's order
    class Rng:
        def __init__(self, vals): self.v = list(vals)
        def random(self): return self.v.pop(0)
    res_d = s.generate(prompt, prompt.size + n, temperature=T, rng=Rng(u.tolist()), on_device_loop=True)
    res_h = hm.session(64)
    res_h.set_strict(True)
    res_h = res_h.generate(prompt, prompt.size + n, temperature=T, rng=Rng(u.tolist()), on_device_loop=False)
    np.testing.assert_array_equal(res_d["tokens"], np.array(want, dtype=np.int32))
    np.testing.assert_array_equal(res_h["tokens"], res_d["tokens"])
    s.close()


def test_positions_at_the_context_tail_are_refused(gpu):
    """kv head h reads RoPE table row position + 2*h (CausalSelfAttention.java:260-283); the reference's table has
    context_length rows and Java throws ArrayIndexOutOfBounds for the last 2*(kvHeads-1) positions.  Same positions are
    an error code here (never an out-of-bounds read of the device table)."""
    from jlama_amd import _native as N, synthetic as S
    from jlama_amd.hip_tensor_operations import HipTensorOperations
    from jlama_amd.model import HipLlamaModel
    cfg = dict(S.TINY)                               # context 256, 2 kv heads => last usable position 253
    hm = HipLlamaModel(cfg, S.make_weights(cfg, seed=0))
    s = hm.session(cfg["context_length"])
    last_ok = cfg["context_length"] - 1 - 2 * (cfg["n_kv_heads"] - 1)
    s.forward([3], last_ok, want_output=False)
    with pytest.raises(N.JhError) as e:
        s.forward([3], last_ok + 1, want_output=False)
    assert e.value.code == N.JH_ERR_INVALID
    with pytest.raises(N.JhError) as e:
        s.decode_n(3, last_ok - 3, 6)              # within max_ctx, but the last positions' RoPE rows leave the table
    assert e.value.code == N.JH_ERR_INVALID
    ops = HipTensorOperations()
    table = ops.rope_table(64, 16, 10000.0)
    q = np.ones(4 * 64, np.float32); k = np.ones(2 * 64, np.float32)
    ops.rope_apply(q, k, table, 13, 4, 2, 64)
    with pytest.raises(N.JhError):
        ops.rope_apply(q, k, table, 14, 4, 2, 64)


def test_missing_weights_are_an_error_code_not_a_gpu_fault(gpu):
    from jlama_amd import _native as N, synthetic as S
    from jlama_amd.model import HipLlamaModel
    cfg = dict(S.TINY)
    w = S.make_weights(cfg, seed=0)
    for drop in [(1, S.W_DOWN), (0, S.W_NORM2), (1, S.W_O), (0, S.W_NORM1)]:
        part = {k: v for k, v in w.items() if k != drop}
        s = HipLlamaModel(cfg, part).session(32)
        with pytest.raises(N.JhError) as e:
            s.forward([1, 2, 3, 4, 5, 6], 0)       # batched prefill path
        assert e.value.code == N.JH_ERR_INVALID
        with pytest.raises(N.JhError) as e:
            s.forward([1], 0)                      # decode path
        assert e.value.code == N.JH_ERR_INVALID


@pytest.mark.gpu
def test_one_launch_reference_order_attention_equals_the_two_launch_form_past_one_value_tile(gpu, oracle):
    """attn_p16_fused_kernel keeps a 512-position V tile in LDS; contexts between 513 and 1,024 positions take a second tile (and
    whole 64-link chunks + a 16-link tail in both).  The two-launch form (attn_p16_scores_kernel + attn_p16_av_kernel, selected by
    JH_P16_ATT_FUSED=0) is an independent implementation of the same sums: greedy ids and logits must agree bit for bit while the
    context grows from 540 to 600 positions."""
    from jlama_amd import synthetic as S
    cfg = dict(S.SMALL)
    cfg["context_length"] = 1024
    hm, om, w = _pair(cfg, 5, oracle)
    prompt = S.prompt_tokens(cfg, n=539, seed=23)
    out = []
    for fused, seq_min in ((1024, 0), (0, 0), (0, 512)):      # one launch; two launches with the wave's sum chain; ... with the parallel exact sum
        _N.set_option("JH_P16_ATT_FUSED", str(fused))
        if seq_min:
            _N.set_option("JH_P16_AV_SEQ_MIN", str(seq_min))
        s = hm.session(1024)
        s.set_strict(True)
        s.forward(prompt, 0)
        t, l0 = s.sample(0.0, 0.5, want_logits=True)
        ids = list(s.decode_n(t, prompt.size, 60))
        out.append((t, l0.copy(), ids, s.logits().copy()))
        s.close()
        _N.clear_options()
    for other in out[1:]:
        assert out[0][0] == other[0]
        np.testing.assert_array_equal(out[0][1].view(np.uint32), other[1].view(np.uint32))
        assert out[0][2] == other[2]
        np.testing.assert_array_equal(out[0][3].view(np.uint32), other[3].view(np.uint32))
