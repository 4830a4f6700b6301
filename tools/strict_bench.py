#!/usr/bin/env python3
"""Reference-order (jh_p16.h) vs order-free kernels on one GPU: decode tok/s and per-kernel probes (HIP events on the
session stream), full-size synthetic model.  usage: strict_bench.py [CONFIG] [steps] [fast,strict]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
    from jlama_amd.model import HipLlamaModel
    if os.environ.get("JH_LIB"):           # tools only: A/B another build of the library on the SAME box (rates move ~5 % box to box)
        N.LIB_PATH = os.path.abspath(os.environ["JH_LIB"])
    config = sys.argv[1] if len(sys.argv) > 1 else "LLAMA3_8B"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    cfg = dict(getattr(S, config))
    torch.cuda.set_device(0)
    N.init(0)
    N.options_from_env()   # tools only: JH_* environment variables become explicit library options
    w = ST.make_weights(cfg, seed=0, device="cuda")
    model = HipLlamaModel(cfg, w)
    prompt = S.prompt_tokens(cfg, n=int(os.environ.get("PROMPT", "128")), seed=1234)   # PROMPT=2048 / 8100: long-context decode
    out = {"config": config, "steps": steps}
    names = ["qkv", "attention", "o_proj", "gate_up", "down"]
    modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ("fast", "strict")
    for mode in modes:
        s = model.session(prompt.size + steps + 8)
        s.batch_forward(prompt, 0)           # fast prefill in both modes: this tool measures kernels, not parity
        first = s.sample()
        if mode == "strict":
            s.set_strict(True)
        s.decode_n(first, prompt.size, 4)
        s.synchronize()
        t0 = time.perf_counter()
        toks = s.decode_n(first, prompt.size, steps)
        dt = time.perf_counter() - t0
        ev_ms, kernels = s.decode_stats()
        probe = {}
        for i, nm in enumerate(names + [None] * 4 + ["lm_head"]):
            if nm is None:
                continue
            ms, b = s.kernel_bench(i, 3)
            probe[nm] = {"us": round(ms * 1e3, 2), "GBps": round(b / (ms * 1e-3) / 1e9, 1)}
        out[mode] = {"tok_s": round(steps / dt, 1), "event_ms_per_token": round(ev_ms, 4), "kernels_per_token": kernels,
                     "kernels": probe, "first_ids": [int(t) for t in toks[:8]]}
        s.close()
    free, total = torch.cuda.mem_get_info()
    out["hbm"] = {"weights_GB": round(model.weight_bytes() / 1e9, 2), "operand_copies_GB": round(model.tiled_bytes() / 1e9, 2),
                  "released_GB": round(model.released_bytes() / 1e9, 2), "device_used_GB": round((total - free) / 1e9, 2)}   # JH_STRICT_ONLY=1: row-major nibbles released
    print(json.dumps(out))


if __name__ == "__main__":
    main()
