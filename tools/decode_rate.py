#!/usr/bin/env python3
"""Reference-order / order-free decode rate of one config without the per-kernel probes (which are JQ4-only): for A/B runs of two
library builds on one box (JH_LIB, tools/build_variant.py).  usage: decode_rate.py [CONFIG] [steps] [strict|fast]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
if os.environ.get("JH_LIB"):
    N.LIB_PATH = os.path.abspath(os.environ["JH_LIB"])
config = sys.argv[1] if len(sys.argv) > 1 else "LLAMA3_8B"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
mode = sys.argv[3] if len(sys.argv) > 3 else "strict"
cfg = dict(getattr(S, config))
torch.cuda.set_device(0); N.init(0); N.options_from_env()
model = HipLlamaModel(cfg, ST.make_weights(cfg, seed=0, device="cuda"))
prompt = S.prompt_tokens(cfg, n=int(os.environ.get("PROMPT", "128")), seed=1234)
s = model.session(prompt.size + steps + 8)
s.batch_forward(prompt, 0)
first = s.sample()
if mode == "strict":
    s.set_strict(True)
s.decode_n(first, prompt.size, 4); s.synchronize()
t0 = time.perf_counter()
toks = s.decode_n(first, prompt.size, steps)
dt = time.perf_counter() - t0
print(json.dumps({"config": config, "mode": mode, "steps": steps, "tok_s": round(steps / dt, 1), "first_ids": [int(t) for t in toks[:6]]}))
