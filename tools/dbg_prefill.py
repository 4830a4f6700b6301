import os, numpy as np, sys
sys.path.insert(0, "/root/repo")
from jlama_amd import synthetic as S
from jlama_amd.model import HipLlamaModel
from oracle import oracle
cfg = dict(S.SMALL)
if len(sys.argv) > 1 and sys.argv[1] == "BF16":
    from jlama_amd import _native as N
    cfg["weight_dtype"] = N.DT_BF16
w = S.make_weights(cfg, seed=21)
hm = HipLlamaModel(cfg, w); om = oracle.OracleModel(cfg, w)
prompt = S.prompt_tokens(cfg, n=300, seed=22)
want = om.session().forward(prompt, 0)
os.environ["JH_PREFILL_BATCH_MIN"] = "0"
rows = hm.session(512).forward(prompt, 0)
del os.environ["JH_PREFILL_BATCH_MIN"]
bat = hm.session(512).forward(prompt, 0)
for name, a, b in (("bat-want", bat, want), ("rows-want", rows, want), ("bat-rows", bat, rows)):
    d = np.abs(a - b)
    print(name, "max", d.max(), "rel", d.max() / np.abs(b).max(), "median", np.median(d), "worst row", d.max(axis=1).argmax(), "per-row max first 8", d.max(axis=1)[:8])
