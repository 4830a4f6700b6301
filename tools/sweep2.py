#!/usr/bin/env python3
"""Per-kernel probe at the planner's defaults + a few overrides (8-layer Llama-3-8B slice)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
cfg = dict(getattr(S, os.environ.get("SWEEP_CFG", "LLAMA3_8B"))); cfg["n_layers"] = int(os.environ.get("SWEEP_LAYERS", "8"))
N.init(0)
N.options_from_env()   # tools only: JH_* environment variables become explicit library options
m = HipLlamaModel(cfg, ST.make_weights(cfg, seed=0, device="cuda"))
names = ["qkv", "attn", "oproj", "gateup", "down", "qkv_preq", "oproj_preq", "gateup_preq", "down_preq"]
def run(env, k, ctx=512):
    for kk, v in env.items(): os.environ[kk] = str(v)
    s = m.session(ctx); ms, b = s.kernel_bench(k, 5); s.close()
    for kk in env: del os.environ[kk]
    return round(ms * 1e3, 2), round(b / ms / 1e6, 0)
variants = json.loads(os.environ.get("SWEEP_VARIANTS", "[{}]"))
for k in [int(x) for x in os.environ.get("SWEEP_KERNELS", "0,1,2,3,4").split(",")]:
    out = []
    for v in variants:
        try: out.append((run(v, k), v))
        except Exception as e: out.append(("ERR " + str(e)[:60], v))
    print(names[k], out, flush=True)
