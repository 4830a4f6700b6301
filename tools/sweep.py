#!/usr/bin/env python3
"""Sweep launch configs of the decode kernels with the jh_kernel_bench probe (8-layer Llama-3-8B slice)."""
import itertools, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
cfgname = os.environ.get("SWEEP_CFG", "LLAMA3_8B")
cfg = dict(getattr(S, cfgname)); cfg["n_layers"] = int(os.environ.get("SWEEP_LAYERS", "8"))
N.init(0)
N.options_from_env()   # tools only: JH_* environment variables become explicit library options
w = ST.make_weights(cfg, seed=0, device="cuda", need_embed=True, need_head=True)
m = HipLlamaModel(cfg, w)
names = ["qkv", "attn", "oproj", "gateup", "down"]
which = [int(x) for x in os.environ.get("SWEEP_KERNELS", "0,1,2,3,4").split(",")]
def run(env, k, pos=None):
    for kk, v in env.items(): os.environ[kk] = str(v)
    s = m.session(512 if pos is None else 2 * pos)
    ms, b = s.kernel_bench(k, 5)
    s.close()
    for kk in env: del os.environ[kk]
    return ms * 1e3, b
for k in which:
    res = []
    if k == 1:
        for splits in (8, 16, 32):
            for pos in (130, 256, 384):
                us, b = run({"JH_ATTN_SPLITS": splits}, k, pos)
                res.append((round(us, 2), splits, pos))
    else:
        pre = {0: "QKV", 2: "O", 3: "GATEUP", 4: "DOWN"}[k]
        for R, waves, gx in itertools.product((2, 4, 8), (4, 8), (1, 2, 4)):
            try:
                us, b = run({f"JH_{pre}_R": R, f"JH_{pre}_WAVES": waves, f"JH_{pre}_GRIDX": gx}, k)
            except Exception as e:
                continue
            res.append((round(us, 2), R, waves, gx, round(b / us / 1e3, 1)))
    res.sort()
    print(names[k], json.dumps(res[:6]), "worst", res[-1], flush=True)
