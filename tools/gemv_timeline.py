#!/usr/bin/env python3
"""Where the time of one reference-order few-row GEMV launch goes: phase stamps (wall_clock64) of every wave.
usage: gemv_timeline.py [which=4]   (0 q|k|v, 2 o, 4 down); CONFIG=LLAMA3_8B"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
which = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dict(getattr(S, os.environ.get("CONFIG", "LLAMA3_8B")))
cfg["n_layers"] = int(os.environ.get("LAYERS", "8"))
torch.cuda.set_device(0); N.init(0); N.options_from_env()
m = HipLlamaModel(cfg, ST.make_weights(cfg, seed=0, device="cuda"))
s = m.session(400)
s.batch_forward(S.prompt_tokens(cfg, n=8, seed=1), 0)
s.set_strict(True)
names = ["entry", "ring requested", "operands in LDS", "main loop done", "last block done", "stored"]
for rep in range(2):
    out = np.full(1024 * 8 * 8, -1, dtype=np.int64)
    N.check(N.lib().jh_debug_gemv_timeline(s.h, which, N.ptr(out), out.size))
t = out.reshape(1024, 8, 8)
used = t[:, :, 0] >= 0
base = t[:, :, 0][used].min()
us = (t - base) / 100.0
work = t[:, :, 5] >= 0          # waves that own rows
print(f"which={which}: {used.any(axis=1).sum()} workgroups, {used.sum()} waves, {work.sum()} with rows; kernel span {us[:, :, :6][t[:, :, :6] >= 0].max():.2f} us")
v6 = us[:, :, 6][work & (t[:, :, 6] >= 0)]
if v6.size:
    print(f"  {'row requested':18s} min {v6.min():6.2f}  p10 {np.percentile(v6, 10):6.2f}  median {np.median(v6):6.2f}  p90 {np.percentile(v6, 90):6.2f}  max {v6.max():6.2f}")
for k, nm in enumerate(names):
    sel = work if k != 2 else used
    v = us[:, :, k][sel & (t[:, :, k] >= 0)]
    if v.size:
        print(f"  {nm:18s} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
w = us[work]
for k in range(1, 6):
    d = w[:, k] - w[:, k - 1]
    print(f"  phase {names[k - 1]} -> {names[k]}: median {np.median(d):.2f} us, max {d.max():.2f}")
# who finishes late: by XCD (workgroup x runs on XCD x & 7), by wave slot, and the slowest / fastest workgroups
fin = np.where(t[:, :, 3] >= 0, us[:, :, 3], np.nan)
wgs = np.nonzero(used.any(axis=1))[0]
print("  main loop done by XCD     :", " ".join(f"{np.nanmedian(fin[wgs[wgs % 8 == x]]):.2f}/{np.nanmax(fin[wgs[wgs % 8 == x]]):.2f}" for x in range(8)), "(median/max)")
print("  main loop done by wave    :", " ".join(f"{np.nanmedian(fin[wgs, w]):.2f}" for w in range(8) if np.isfinite(fin[wgs, w]).any()))
wgmax = np.nanmax(fin[wgs], axis=1)
order = np.argsort(wgmax)
print("  fastest workgroups        :", " ".join(f"{wgs[i]}:{wgmax[i]:.2f}" for i in order[:8]))
print("  slowest workgroups        :", " ".join(f"{wgs[i]}:{wgmax[i]:.2f}" for i in order[-12:]))
print("  workgroup finish quantiles:", " ".join(f"{np.percentile(wgmax, q):.2f}" for q in (0, 10, 25, 50, 75, 90, 100)))
