"""Prefill of one 129-row prompt, repeated -- run under `rocprofv3 --kernel-trace --stats` to get the per-kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: torch bundles its own HIP runtime)
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
cfg = dict(getattr(S, os.environ.get("PF_CFG", "LLAMA3_8B"))); cfg["n_layers"] = int(os.environ.get("PF_LAYERS", "8"))
N.init(0)
N.options_from_env()   # tools only: JH_* environment variables become explicit library options
m = HipLlamaModel(cfg, ST.make_weights(cfg, seed=0, device="cuda"))
prompt = S.prompt_tokens(cfg, n=int(os.environ.get("PF_ROWS", "128")), seed=1)
s = m.session(prompt.size + 8)
for _ in range(6):
    s.forward(prompt, 0, want_output=False)
s.synchronize()
