"""N independent sessions decoding concurrently on ONE GPU (each on its own stream, one hipGraph replay per token each): the
kernels of different sessions fill each other's launch edges and ramps.  Not the metric (batch-1 decode), a serving data point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
cfg = dict(getattr(S, os.environ.get("MS_CONFIG", "LLAMA3_8B")))
N.init(0)
N.options_from_env()   # tools only: JH_* environment variables become explicit library options
model = HipLlamaModel(cfg, ST.make_weights(cfg, seed=0, device="cuda"))
prompt = S.prompt_tokens(cfg, n=128, seed=1234)
steps = int(os.environ.get("MS_STEPS", "128"))
for n in (1, 2, 3, 4, 8):
    ss = [model.session(prompt.size + steps + 8) for _ in range(n)]
    firsts = []
    for s in ss:
        s.batch_forward(prompt, 0)
        firsts.append(s.sample())
        s.decode_n(firsts[-1], prompt.size, 1)      # capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, f in zip(ss, firsts):
        s.decode_n_async(f, prompt.size, steps)
    outs = [s.decode_wait(steps) for s in ss]
    dt = time.perf_counter() - t0
    same = all(np.array_equal(outs[0], o) for o in outs)
    print(f"{n} sessions: {n * steps / dt:8.1f} tok/s aggregate, {steps / dt:7.1f} per session, identical ids: {same}", flush=True)
    for s in ss:
        s.close()
