"""Single-stream decode rate of the one-process tensor-parallel group (jh_tp_group_*) with all head-split shards on ONE device:
what the host-launched (not graph-captured) halves + the peer-write reductions cost against the un-sharded model."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jlama_amd import _native as N, distributed as D, synthetic as S
from jlama_amd.model import HipLlamaModel, HipTPGroup
cfg = dict(getattr(S, os.environ.get("TP_CONFIG", "LLAMA32_1B")))
N.init(0)
w = S.make_weights(cfg, seed=0)
prompt = S.prompt_tokens(cfg, n=16, seed=3)
steps = 64
full = HipLlamaModel(cfg, w).session(128)
full.batch_forward(prompt, 0)
f = full.sample()
full.decode_n(f, prompt.size, 4)
t0 = time.perf_counter(); ref = full.decode_n(f, prompt.size, steps); dt = time.perf_counter() - t0
print(f"un-sharded, graph-captured loop: {steps / dt:8.1f} tok/s", flush=True)
for size in (2, 4):
    models = []
    for r in range(size):
        lc, off = D.tp_shard_config(cfg, r, size)
        models.append(HipLlamaModel(lc, D.tp_shard_weights(cfg, w, r, size), kv_head_offset=off))
    g = HipTPGroup(models, 128)
    g.forward(prompt, 0)
    f2 = g.sample()
    g.decode_n(f2, prompt.size, 4)
    t0 = time.perf_counter(); got = g.decode_n(f2, prompt.size, steps); dt = time.perf_counter() - t0
    agree = int(np.argmin(got == ref)) if not (got == ref).all() else steps
    print(f"TP group, {size} shards on one device (host-launched halves): {steps / dt:8.1f} tok/s; first token {f2 == f}, ids equal for {agree} steps", flush=True)
    g.close()
