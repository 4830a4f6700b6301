#!/usr/bin/env python3
"""Condense a tools/profile_round.sh run into the record bench.py reads: the dominant kernel's rocprofv3 average duration,
its HBM traffic from the FETCH_SIZE / WRITE_SIZE passes (gfx950 correction of MI355X_MICROARCH.md HBM section: FETCH_SIZE
counts 1/2 of a wide coalesced streaming read), and the source hash of the binary that was profiled.
Usage: dominant_kernel_json.py <prefix of the *_kernel_trace_stats.md / *_pmc_*.md files> <CONFIG>"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jlama_amd import _native as N, synthetic as S  # noqa: E402

prefix, cfgname = sys.argv[1], sys.argv[2]
cfg = getattr(S, cfgname)


def rows(path):
    out = []
    for line in open(path):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) >= 4 and cells[0] and not set(cells[0]) <= set("-") and cells[0] != "kernel":
            out.append(cells)
    return out


def is_gateup(name):   # gemv_t16_kernel<PRO_RMS_Q8=1, EPI_SILU_MUL=2, ...> (both decode loops since round 4), gemv_i8q4_kernel<1, 2, ...> where T16 does not fit
    return re.match(r"gemv_t16_kernel<1, 2,", name) is not None or re.match(r"gemv_i8q4_kernel<1, 2,", name) is not None


trace = [r for r in rows(prefix + "_kernel_trace_stats.md") if is_gateup(r[0])]
fetch = [r for r in rows(prefix + "_pmc_fetch_size.md") if is_gateup(r[0])]
write = [r for r in rows(prefix + "_pmc_write_size.md") if is_gateup(r[0])]
E, H = cfg["embedding_length"], cfg["hidden_length"]
algo = int(2 * H * E * 0.625)
rec = {"config": cfgname, "source_hash": N.built_hash(), "kernel": trace[0][0] if trace else None,
       "us_per_launch_rocprof": float(trace[0][2]) if trace else None,
       "FETCH_SIZE_KB_per_dispatch": float(fetch[0][3]) if fetch else None,
       "WRITE_SIZE_KB_per_dispatch": float(write[0][3]) if write else None,
       "algorithmic_bytes_per_launch": algo,
       "correction": "gfx950: FETCH_SIZE reports 1/2 of a wide coalesced (16 B/lane) streaming read; traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024"}
if fetch:
    rec["traffic_bytes_per_launch"] = (2 * rec["FETCH_SIZE_KB_per_dispatch"] + (rec["WRITE_SIZE_KB_per_dispatch"] or 0.0)) * 1024
    rec["ratio_to_algorithmic"] = rec["traffic_bytes_per_launch"] / algo


# the same kernel of the reference-order path (jh_t16.h / jh_p16.h)
def is_gateup_p16(name):   # gemv_t16_kernel<PRO_RMS_Q8=1, EPI_SILU_MUL=2, ...> (jh_t16.h), or the p16 form where the shape rules T16 out
    return re.match(r"gemv_t16_kernel<1, 2,", name) is not None or re.match(r"gemv_i8q4_p16_kernel<1, 2,", name) is not None


t16 = [r for r in rows(prefix + "_kernel_trace_stats.md") if is_gateup_p16(r[0])]
f16 = [r for r in rows(prefix + "_pmc_fetch_size.md") if is_gateup_p16(r[0])]
w16 = [r for r in rows(prefix + "_pmc_write_size.md") if is_gateup_p16(r[0])]
if t16:
    ro = {"kernel": t16[0][0], "us_per_launch_rocprof": float(t16[0][2]),
          "FETCH_SIZE_KB_per_dispatch": float(f16[0][3]) if f16 else None, "WRITE_SIZE_KB_per_dispatch": float(w16[0][3]) if w16 else None}
    if f16:
        ro["traffic_bytes_per_launch"] = (2 * ro["FETCH_SIZE_KB_per_dispatch"] + (ro["WRITE_SIZE_KB_per_dispatch"] or 0.0)) * 1024
        ro["ratio_to_algorithmic"] = ro["traffic_bytes_per_launch"] / algo
    rec["reference_order"] = ro
print(json.dumps(rec, indent=1))
