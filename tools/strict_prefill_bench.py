"""Reference-order (Panama-512 summation order) prompt processing: the batched path (gemm_t16_kernel + rows_*_p16_kernel) against
the one-position-at-a-time path, full-size model, the metric's 129-row prompt (SPB_ROWS=n for another length; SPB_SKIP_ROWS=1
times the batched path only).  Usage: [TP_CONFIG=LLAMA3_8B] python tools/strict_prefill_bench.py"""
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
nrows = int(os.environ.get("SPB_ROWS", "129"))
prompt = S.prompt_tokens(cfg, n=nrows - 1, seed=1234)
res = {}
modes = (("batched", None),) if os.environ.get("SPB_SKIP_ROWS") else (("batched", None), ("row by row", "0"))
for mode, env in modes:
    if env is not None:
        N.set_option("JH_PREFILL_BATCH_MIN", env)
    s = model.session(max(512, nrows + 64))
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
    if env is not None:
        N.clear_options()
        N.options_from_env()
if len(res) < 2:
    sys.exit(0)
a, b = res["batched"], res["row by row"]
print("rows bit-identical:", bool(np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))), " first token equal:", a[1] == b[1],
      " logits bit-identical:", bool(np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))))
