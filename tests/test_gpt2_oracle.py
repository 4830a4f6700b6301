"""BASELINE.json configs[0]: GPT-2 small, F32 weights, greedy decode on the CPU path -- the reference's plumbing case
(no GPU).  The oracle restates GPT2Model (core/model/gpt2/GPT2Model.java:53-129) on top of the shared block code:
wte + wpe embeddings, LayerNorm with bias (LayerNorm.java:41-67), q/k/v/o and MLP biases (accumulate after the GEMM,
CausalSelfAttention.java:183-191,378-380, MLPBlock.java:128-130,163), tanh-GELU, no RoPE, LM head = wte.

Pinned by an independent float64 numpy evaluation of the GPT-2 architecture (HF conventions: c_attn / c_fc / c_proj
stored [in, out]; the reference transposes them at load, :79-80,97-100) -- the oracle's F32 result must agree to 1e-4
and the greedy ids must be equal wherever the float64 margin is clear; plus the size-independent invariants of the KV
path (row-by-row == batched, bit for bit)."""
import numpy as np
import pytest

GPT2_SMALL = dict(embedding_length=768, hidden_length=3072, n_heads=12, n_kv_heads=12, head_size=64, n_layers=12,
                  vocab_size=50257, context_length=1024, weight_dtype=0, rms_eps=1e-5, rope_theta=10000.0,
                  rope_scaling=1.0, bos_token=50256)


def make_gpt2(cfg, seed, O):
    """Synthetic GPT-2 checkpoint in HF layout (weights N(0, 0.02^2), biases small, LN weights ~1) and the
    oracle's weight dict (matrices transposed to [out, in] and c_attn split in three, as GPT2Model.java:79-80 does)."""
    rng = np.random.default_rng(seed)
    E, H, L, V, C = cfg["embedding_length"], cfg["hidden_length"], cfg["n_layers"], cfg["vocab_size"], cfg["context_length"]
    f = lambda *s, sd=0.02: (rng.standard_normal(s) * sd).astype(np.float32)
    hf = {"wte": f(V, E), "wpe": f(C, E, sd=0.01), "ln_f.w": (1 + f(E, sd=0.05)), "ln_f.b": f(E, sd=0.01)}
    w = {(-1, O.W_EMBED): dict(dtype=O.DT_F32, data=hf["wte"], scales=None, shape=(V, E)),
         (-1, O.W_WPE): dict(dtype=O.DT_F32, data=hf["wpe"], scales=None, shape=(C, E)),
         (-1, O.W_FINALNORM): dict(dtype=O.DT_F32, data=hf["ln_f.w"].reshape(1, E), scales=None, shape=(1, E)),
         (-1, O.W_FINALNORMB): dict(dtype=O.DT_F32, data=hf["ln_f.b"].reshape(1, E), scales=None, shape=(1, E))}
    for li in range(L):
        h = {"ln_1.w": 1 + f(E, sd=0.05), "ln_1.b": f(E, sd=0.01), "ln_2.w": 1 + f(E, sd=0.05), "ln_2.b": f(E, sd=0.01),
             "c_attn.w": f(E, 3 * E), "c_attn.b": f(3 * E, sd=0.01), "attn.c_proj.w": f(E, E), "attn.c_proj.b": f(E, sd=0.01),
             "c_fc.w": f(E, H), "c_fc.b": f(H, sd=0.01), "mlp.c_proj.w": f(H, E), "mlp.c_proj.b": f(E, sd=0.01)}
        hf[li] = h
        wt = np.ascontiguousarray(h["c_attn.w"].T)                        # [3E, E]
        mats = {O.W_Q: wt[:E], O.W_K: wt[E:2 * E], O.W_V: wt[2 * E:], O.W_O: h["attn.c_proj.w"].T,
                O.W_GATE: h["c_fc.w"].T, O.W_DOWN: h["mlp.c_proj.w"].T}
        for slot, m in mats.items():
            m = np.ascontiguousarray(m, dtype=np.float32)
            w[(li, slot)] = dict(dtype=O.DT_F32, data=m, scales=None, shape=m.shape)
        vecs = {O.W_QB: h["c_attn.b"][:E], O.W_KB: h["c_attn.b"][E:2 * E], O.W_VB: h["c_attn.b"][2 * E:],
                O.W_OB: h["attn.c_proj.b"], O.W_GATEB: h["c_fc.b"], O.W_DOWNB: h["mlp.c_proj.b"],
                O.W_NORM1: h["ln_1.w"], O.W_NORM1B: h["ln_1.b"], O.W_NORM2: h["ln_2.w"], O.W_NORM2B: h["ln_2.b"]}
        for slot, v in vecs.items():
            v = np.ascontiguousarray(v, dtype=np.float32).reshape(1, -1)
            w[(li, slot)] = dict(dtype=O.DT_F32, data=v, scales=None, shape=v.shape)
    return hf, w


def gpt2_float64(cfg, hf, tokens):
    """Independent evaluation of the architecture in float64 (HF GPT2 semantics), all positions at once."""
    E, nh, hs, L = cfg["embedding_length"], cfg["n_heads"], cfg["head_size"], cfg["n_layers"]
    d = lambda a: np.asarray(a, dtype=np.float64)
    T = len(tokens)
    x = d(hf["wte"])[tokens] + d(hf["wpe"])[:T]

    def ln(v, w, b):
        mu = v.mean(-1, keepdims=True)
        var = ((v - mu) ** 2).mean(-1, keepdims=True)
        return (v - mu) / np.sqrt(var + cfg["rms_eps"]) * d(w) + d(b)

    for li in range(L):
        h = hf[li]
        a = ln(x, h["ln_1.w"], h["ln_1.b"]) @ d(h["c_attn.w"]) + d(h["c_attn.b"])
        q, k, v = a[:, :E], a[:, E:2 * E], a[:, 2 * E:]
        out = np.zeros((T, E))
        for hd in range(nh):
            sl = slice(hd * hs, (hd + 1) * hs)
            sc = q[:, sl] @ k[:, sl].T / np.sqrt(hs)
            sc = np.where(np.tril(np.ones((T, T), bool)), sc, -np.inf)
            sc = np.exp(sc - sc.max(-1, keepdims=True))
            out[:, sl] = (sc / sc.sum(-1, keepdims=True)) @ v[:, sl]
        x = x + out @ d(h["attn.c_proj.w"]) + d(h["attn.c_proj.b"])
        m = ln(x, h["ln_2.w"], h["ln_2.b"]) @ d(h["c_fc.w"]) + d(h["c_fc.b"])
        m = 0.5 * m * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (m + 0.044715 * m ** 3)))
        x = x + m @ d(h["mlp.c_proj.w"]) + d(h["mlp.c_proj.b"])
    return x, ln(x, hf["ln_f.w"], hf["ln_f.b"]) @ d(hf["wte"]).T


@pytest.mark.parametrize("shape", ["tiny", "small_2_layers"])
def test_gpt2_forward_matches_float64_architecture(oracle, shape):
    O = oracle
    cfg = dict(GPT2_SMALL)
    if shape == "tiny":
        cfg.update(embedding_length=64, hidden_length=256, n_heads=4, n_kv_heads=4, head_size=16, n_layers=3, vocab_size=211,
                   context_length=64)
    else:
        cfg.update(n_layers=2, vocab_size=1024, context_length=128)     # GPT-2 small's E / H / heads
    hf, w = make_gpt2(cfg, 5, O)
    rng = np.random.default_rng(9)
    tokens = rng.integers(0, cfg["vocab_size"], size=23).astype(np.int32)
    om = O.OracleModel(cfg, w, arch=O.ARCH_GPT2)
    x = om.session().forward(tokens, 0)
    want_x, want_logits = gpt2_float64(cfg, hf, tokens)
    assert np.abs(x - want_x).max() <= 1e-4 * np.abs(want_x).max()
    for row in (0, 7, 22):
        tok, logits = om.sample(x[row])
        assert np.abs(logits - want_logits[row]).max() <= 1e-4 * np.abs(want_logits[row]).max()
        top2 = np.partition(want_logits[row], -2)[-2:]
        if top2[1] - top2[0] > 1e-3:
            assert tok == int(np.argmax(want_logits[row]))
    # KV path invariants: one position at a time == one batch, bit for bit (same per-row arithmetic)
    s2 = om.session()
    rows = np.concatenate([s2.forward([t], i) for i, t in enumerate(tokens)])
    np.testing.assert_array_equal(rows.view(np.uint32), x.view(np.uint32))


def test_gpt2_small_greedy_decode_full_size(oracle):
    """configs[0] itself: GPT-2 small (12 layers, E = 768, V = 50257, tied LM head) greedy decode, 12-token prompt + 8
    generated tokens through AbstractModel.generate's restatement; the float64 evaluation of the SAME final context must
    reproduce every generated id whose margin is clear."""
    O = oracle
    cfg = dict(GPT2_SMALL)
    hf, w = make_gpt2(cfg, 11, O)
    om = O.OracleModel(cfg, w, arch=O.ARCH_GPT2)
    rng = np.random.default_rng(3)
    prompt = np.concatenate([[cfg["bos_token"]], rng.integers(0, cfg["vocab_size"], size=11)]).astype(np.int32)
    ids, logits, _ = om.session().generate(prompt, 9)
    assert ids.size == 9
    ctx = np.concatenate([prompt, ids[:-1]])
    _, want_logits = gpt2_float64(cfg, hf, ctx)
    for i, tok in enumerate(ids):
        row = want_logits[prompt.size - 1 + i]
        top2 = np.partition(row, -2)[-2:]
        if top2[1] - top2[0] > 1e-3:
            assert tok == int(np.argmax(row)), (i, tok, int(np.argmax(row)))
    assert np.abs(logits - want_logits[-1]).max() <= 2e-4 * np.abs(want_logits[-1]).max()
