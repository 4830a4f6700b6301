"""Cost of temperature sampling inside the device loop (jh_decode_n_sampled: sample_exp_kernel + sample_pick_kernel per token)
against the greedy loop, V = 128256 on a tiny trunk so that the difference is the sampler.  SB_T=0.8 SB_N=256.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jlama_amd import _native as N, synthetic as S
from jlama_amd.model import HipLlamaModel
N.init(0)
N.options_from_env()
cfg = dict(S.TINY)
cfg["vocab_size"] = int(os.environ.get("SB_V", "128256"))
T, n = float(os.environ.get("SB_T", "0.8")), int(os.environ.get("SB_N", "200"))
model = HipLlamaModel(cfg, S.make_weights(cfg, seed=3))
prompt = S.prompt_tokens(cfg, n=8, seed=1)
s = model.session(cfg["context_length"])
u = np.random.default_rng(1).random(n).astype(np.float32)
res = {}
for name in ("greedy", "sampled", "greedy", "sampled"):
    s.batch_forward(prompt, 0)
    first = s.sample()
    s.synchronize()
    t0 = time.perf_counter()
    ids = s.decode_n(first, prompt.size, n) if name == "greedy" else s.decode_n_sampled(first, prompt.size, n, T, u)
    s.synchronize()
    res[name] = (time.perf_counter() - t0) / n * 1e6
    print(f"{name:8s} {res[name]:8.1f} us/token  distinct ids {len(set(ids.tolist()))}", flush=True)
print(f"V = {cfg['vocab_size']}, T = {T}: sampling adds {res['sampled'] - res['greedy']:.1f} us per token")
