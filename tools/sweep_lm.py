import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
import torch
cfg = dict(S.LLAMA3_8B); cfg["n_layers"] = 1
N.init(0)
N.options_from_env()   # tools only: JH_* environment variables become explicit library options
m = HipLlamaModel(cfg, ST.make_weights(cfg, seed=0, device="cuda"))
def run(env):
    for k, v in env.items(): os.environ[k] = str(v)
    s = m.session(64); s.forward([5], 0, want_output=False)
    for _ in range(3): s.sample()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): s.sample()
    dt = (time.perf_counter() - t0) / 20
    s.close()
    for k in env: del os.environ[k]
    return round(dt * 1e6, 1)
base = run({})
for R in (1, 2, 4):
    for waves in (4, 8):
        for gx in (1, 2, 4, 8):
            print(R, waves, gx, run({"JH_LM_R": R, "JH_LM_WAVES": waves, "JH_LM_GRIDX": gx}), flush=True)
print("default", base)
