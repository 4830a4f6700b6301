"""129-row reference-order prefill of Mistral-7B BF16 (gemm_bf16r_kernel), wall time per call."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
cfg = dict(S.MISTRAL_7B)
torch.cuda.set_device(0); N.init(0); N.options_from_env()
model = HipLlamaModel(cfg, ST.make_weights(cfg, seed=0, device="cuda"))
prompt = S.prompt_tokens(cfg, n=128, seed=1234)
s = model.session(200); s.set_strict(True)
s.batch_forward(prompt, 0); s.sample()
for _ in range(2):
    t0 = time.perf_counter(); s.batch_forward(prompt, 0); s.sample(); print("strict BF16 129-row prefill ms", round((time.perf_counter() - t0) * 1e3, 2))
