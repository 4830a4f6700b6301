import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jlama_amd import _native as N, distributed as D, synthetic as S
from jlama_amd.model import HipLlamaModel, HipTPGroup
cfg = dict(getattr(S, os.environ.get("TP_CONFIG", "LLAMA32_1B")))
if os.environ.get("TP_LAYERS"):
    cfg["n_layers"] = int(os.environ["TP_LAYERS"])
N.init(0)
N.options_from_env()   # tools only: JH_* environment variables become explicit library options
w = S.make_weights(cfg, seed=0)
prompt = S.prompt_tokens(cfg, n=16, seed=3)
size = int(os.environ.get("TP_SIZE", "2"))
models = []
for r in range(size):
    lc, off = D.tp_shard_config(cfg, r, size)
    models.append(HipLlamaModel(lc, D.tp_shard_weights(cfg, w, r, size), kv_head_offset=off))
g = HipTPGroup(models, 128)
g.forward(prompt, 0)
f2 = g.sample()
for n in (1, 2, 4, 16, 64):
    try:
        t0 = time.perf_counter(); got = g.decode_n(f2, prompt.size, n); dt = time.perf_counter() - t0
        print(f"n={n}: ok {n/dt:.1f} tok/s ids {got[:6].tolist()}", flush=True)
    except Exception as e:
        print(f"n={n}: FAILED {e}", flush=True)
        break
