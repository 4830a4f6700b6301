#!/usr/bin/env python3
"""One decode kernel kind of the reference-order path repeated over all layers, for rocprofv3 --pmc / --kernel-trace.
usage: p16_pmc.py [which=3] [iters=4] [fast|strict]   (which: 0 qkv, 1 attention, 2 o, 3 gate|up, 4 down, 9 LM head)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
which = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mode = sys.argv[3] if len(sys.argv) > 3 else "strict"
cfg = dict(getattr(S, os.environ.get("CONFIG", "LLAMA3_8B")))
torch.cuda.set_device(0); N.init(0); N.options_from_env()
model = HipLlamaModel(cfg, ST.make_weights(cfg, seed=0, device="cuda"))
s = model.session(400)
s.batch_forward(S.prompt_tokens(cfg, n=8, seed=1), 0)
if mode == "strict":
    s.set_strict(True)
ms, b = s.kernel_bench(which, iters)
print(f"which={which} {mode}: {ms*1e3:.2f} us, {b/(ms*1e-3)/1e9:.0f} GB/s")
