"""Single-stream decode rate of the one-process tensor-parallel group (jh_tp_group_*) with all head-split shards on ONE device
against the un-sharded model: graph-replayed shards that meet in kernels (default) and the event-ordered host loop (JH_TP_GRAPH=0).
All shards on one device need one HARDWARE queue per shard stream (a spinning kernel blocks a queue it shares with the kernel it
waits for): GPU_MAX_HW_QUEUES is raised before HIP initialises; with one shard per device this does not arise."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jlama_amd import _native as N, distributed as D, synthetic as S
from jlama_amd.model import HipLlamaModel, HipTPGroup
cfg = dict(getattr(S, os.environ.get("TP_CONFIG", "LLAMA32_1B")))
N.init(0)
N.options_from_env()   # tools only: JH_* environment variables become explicit library options
w = S.make_weights(cfg, seed=0)
prompt = S.prompt_tokens(cfg, n=int(os.environ.get("TP_PROMPT", "128")), seed=3)   # 129 rows: the metric's prompt
steps = 64
full = HipLlamaModel(cfg, w).session(prompt.size + 128)
full.batch_forward(prompt, 0)
t0 = time.perf_counter(); full.batch_forward(prompt, 0); full.synchronize(); dt = time.perf_counter() - t0
print(f"un-sharded prompt of {prompt.size} rows: {dt * 1e3:8.2f} ms", flush=True)
f = full.sample()
full.decode_n(f, prompt.size, 4)
t0 = time.perf_counter(); ref = full.decode_n(f, prompt.size, steps); dt = time.perf_counter() - t0
print(f"un-sharded, graph-captured loop: {steps / dt:8.1f} tok/s", flush=True)
full.close()
for size in (2, 4):
    models = []
    for r in range(size):
        lc, off = D.tp_shard_config(cfg, r, size)
        models.append(HipLlamaModel(lc, D.tp_shard_weights(cfg, w, r, size), kv_head_offset=off))
    g = HipTPGroup(models, prompt.size + 128)
    g.forward(prompt, 0)
    t0 = time.perf_counter(); g.forward(prompt, 0); dtp = time.perf_counter() - t0       # (forward returns with the shard streams drained)
    N.set_option("JH_PREFILL_BATCH_MIN", 0)
    g1 = HipTPGroup(models, prompt.size + 128)
    g1.forward(prompt[:16], 0)
    t0 = time.perf_counter(); g1.forward(prompt, 0); dtr = time.perf_counter() - t0
    g1.close()
    N.clear_options(); N.options_from_env()
    print(f"TP group, {size} shards: prompt of {prompt.size} rows {dtp * 1e3:8.2f} ms in chunks (one meeting per half-layer), {dtr * 1e3:8.2f} ms row by row", flush=True)
    f2 = g.sample()
    g.decode_n(f2, prompt.size, 4)
    t0 = time.perf_counter(); got = g.decode_n(f2, prompt.size, steps); dt = time.perf_counter() - t0
    agree = int(np.argmin(got == ref)) if not (got == ref).all() else steps
    mode = "event-ordered host loop" if os.environ.get("JH_TP_GRAPH") == "0" else (
        "graph replay per shard, kernels meet on flags, " + ("separate scatter launches" if os.environ.get("JH_TP_FUSE") == "0" else "o-proj / down push their partial rows"))
    print(f"TP group, {size} shards on one device ({mode}): {steps / dt:8.1f} tok/s; first token {f2 == f}, ids equal for {agree} steps", flush=True)
    g.close()
