#!/usr/bin/env python3
"""Scan the gfx950 ISA of one translation unit for loops that end by COPYING registers behind s_waitcnt vmcnt(...): the sign of a
prefetch ring whose refills landed in fresh registers (hipcc then closes the loop by moving the ring back, waiting for the
youngest load each time round -- what made the reference-order few-row GEMVs latency-bound until round 5).
usage: isa_ring_copies.py <unit, e.g. gemv_ref> [min copies = 4]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit = sys.argv[1] if len(sys.argv) > 1 else "gemv_ref"
mincp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
src = os.path.join(ROOT, "jlama_amd", "csrc", unit + ".hip")
asm = f"/tmp/_isa_{unit}.s"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value",
                "-S", "--offload-device-only", src, "-o", asm], check=True, capture_output=True)
fn, block, found = None, [], {}
def flush():
    global block
    movs = [l for l in block if re.match(r"\s*v_mov_b32_e32 v\d+, v\d+\s*$", l) or re.match(r"\s*v_mov_b64_e32 v\[\d+:\d+\], v\[\d+:\d+\]\s*$", l)]
    waits = [l for l in block if "s_waitcnt" in l and "vmcnt" in l]
    ends_in_branch = any(re.match(r"\s*s_c?branch", l) for l in block[-3:])
    # a copy block: (almost) nothing but moves and waits, closed by a branch
    if fn and len(movs) >= mincp and waits and ends_in_branch and len(movs) + len(waits) >= 0.7 * len(block):
        found.setdefault(fn, []).append((len(movs), [w.strip() for w in waits]))
    block = []
for line in open(asm):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        flush(); fn = m.group(1); continue
    if re.match(r"^\.LBB", line):
        flush(); continue
    if line.strip() and not line.strip().startswith(";"):
        block.append(line.rstrip())
flush()
if not found:
    print(f"{unit}: no loop ends by copying a prefetch ring")
for f, hits in found.items():
    name = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip()
    for n, waits in hits:
        print(f"{name[:110]}: {n} copies behind {waits}")
