import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jlama_amd import _native as N, synthetic as S
from jlama_amd.model import HipLlamaModel
from oracle import oracle
oracle.lib()
cfg = dict(getattr(S, sys.argv[1] if len(sys.argv) > 1 else "TINY"))
N.init(0)
w = S.make_weights(cfg, seed=1, quantize=oracle.q4_quantize)
hm, om = HipLlamaModel(cfg, w), oracle.OracleModel(cfg, w)
E, A, H = cfg["embedding_length"], cfg["n_heads"] * cfg["head_size"], cfg["hidden_length"]
prompt = S.prompt_tokens(cfg, n=4, seed=11)
hs, os_ = hm.session(64), om.session()
hs.set_strict(True); hs.set_tap_layer(0); os_.set_tap_layer(0)
for i, t in enumerate(prompt[:3]):
    hs.forward([t], i, want_output=False); os_.forward([t], i)
for name, n in [("query", A), ("after_attention", A), ("attn_res", E), ("ff_h", H), ("post_ff_res", E)]:
    try:
        g, o = hs.tap(name, n), os_.tap(name, n)
    except Exception as e:
        print(name, "n/a", e); continue
    bad = np.nonzero(g.view(np.uint32) != o.view(np.uint32))[0]
    print(name, "mismatch", bad.size, "of", n, "first", bad[:12], "max abs", np.abs(g - o).max())
    if bad.size: print("   got", g[bad[:6]], "want", o[bad[:6]])
