import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from jlama_amd import synthetic as S
from jlama_amd.model import HipLlamaModel
from oracle import oracle as O
cfg = dict(getattr(S, sys.argv[1] if len(sys.argv) > 1 else "SMALL"))
w = S.make_weights(cfg, seed=1)
hm, om = HipLlamaModel(cfg, w), O.OracleModel(cfg, w)
prompt = S.prompt_tokens(cfg, n=40, seed=11)
hs, os_ = hm.session(200), om.session()
xh = hs.batch_forward(prompt, 0); xo = os_.forward(prompt, 0)
print("prefill rel", np.abs(xh - xo).max() / np.abs(xo).max())
tok, lh = hs.sample(0.0, 0.5, want_logits=True); _, lo = om.sample(xo[-1]); print("first", np.abs(lh - lo).max())
E = cfg["embedding_length"]
for i in range(100):
    pos = prompt.size + i
    xh = hs.forward([tok], pos); xo = os_.forward([tok], pos)
    nh, lh = hs.sample(0.0, 0.5, want_logits=True); no, lo = om.sample(xo[-1])
    # oracle logits from the GPU's hidden row: isolates lm-head differences from trunk differences
    _, lo_from_h = om.sample(xh[-1])
    print(i, pos, "x rel %.2e" % (np.abs(xh - xo).max() / np.abs(xo).max()), "logit %.2e" % np.abs(lh - lo).max(), "lmhead-only %.2e" % np.abs(lh - lo_from_h).max(), nh == no, flush=True)
    tok = nh
