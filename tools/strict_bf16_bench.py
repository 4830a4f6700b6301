#!/usr/bin/env python3
"""Reference-order (jh_bf16r.h) vs order-free decode rate of a dense BF16 model on one GPU.  usage: strict_bf16_bench.py [steps] [fast,strict]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ("fast", "strict")
cfg = dict(S.MISTRAL_7B)
torch.cuda.set_device(0)
N.init(0)
N.options_from_env()
model = HipLlamaModel(cfg, ST.make_weights(cfg, seed=0, device="cuda"))
prompt = S.prompt_tokens(cfg, n=128, seed=1234)
out = {}
for mode in modes:
    s = model.session(prompt.size + steps + 8)
    s.batch_forward(prompt, 0)
    first = s.sample()
    if mode == "strict":
        s.set_strict(True)
    s.decode_n(first, prompt.size, 4)
    s.synchronize()
    t0 = time.perf_counter()
    s.decode_n(first, prompt.size, steps)
    dt = time.perf_counter() - t0
    ev_ms, kernels = s.decode_stats()
    out[mode] = {"tok_s": round(steps / dt, 1), "event_ms_per_token": round(ev_ms, 4), "kernels_per_token": kernels}
    s.close()
print(json.dumps(out))
