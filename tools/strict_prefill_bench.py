"""Reference-order (Panama-512 summation order) prompt processing: the M-row p16 GEMM path against the one-position-at-a-time path,
full-size model, the metric's 129-row prompt.  Usage: [TP_CONFIG=LLAMA3_8B] python tools/strict_prefill_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
name = os.environ.get("TP_CONFIG", "LLAMA3_8B")
cfg = dict(getattr(S, name))
N.init(0)
N.options_from_env()   # tools only: JH_* environment variables become explicit library options
w = ST.make_weights(cfg, seed=0, device=torch.device("cuda", 0))
model = HipLlamaModel(cfg, w)
prompt = S.prompt_tokens(cfg, n=128, seed=1234)
res = {}
modes = (("batched", None),) if os.environ.get("SPB_SKIP_ROWS") else (("batched", None), ("row by row", "0"))
for mode, env in modes:
    if env is None:
        os.environ.pop("JH_PREFILL_BATCH_MIN", None)
    else:
        os.environ["JH_PREFILL_BATCH_MIN"] = env
    s = model.session(512)
    s.set_strict(True)
    s.batch_forward(prompt, 0); s.synchronize()          # warm (allocations)
    t0 = time.perf_counter()
    out = s.batch_forward(prompt, 0)
    s.synchronize()
    dt = time.perf_counter() - t0
    tok, logits = s.sample(0.0, 0.5, want_logits=True)
    res[mode] = (out, tok, logits)
    print(f"{name} reference order, {prompt.size}-row prompt, {mode}: {dt * 1e3:8.2f} ms", flush=True)
    s.close()
if len(res) < 2:
    sys.exit(0)
a, b = res["batched"], res["row by row"]
print("rows bit-identical:", bool(np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))), " first token equal:", a[1] == b[1],
      " logits bit-identical:", bool(np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))))
