#!/usr/bin/env python3
"""Scan the gfx950 ISA of a unit for reference-order streaming GEMVs whose PREFETCH RING IS DRAINED IN FRONT OF THE PROLOGUE BARRIER.
The kernels request R non-temporal ring loads, then run the activation prologue (RMSNorm / Q8 into LDS, one workgroup barrier) while
those loads are in flight; every wait between the last ring request and the first s_barrier may only be for the (older) activation
loads, i.e. `s_waitcnt vmcnt(N)` with N >= R.  Round 6: a build whose streaming code was byte-identical ran the o-projection 12 %
slower because hipcc had given a next-row offset the register of a pending ring load -- an `s_waitcnt vmcnt(0)` right behind the fill.
usage: isa_ring_drain.py [--file asm.s | unit]   (kernels: *_p16_kernel / gemv_bf16r_kernel / gemv_t16_kernel)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--file":
    asm, unit = sys.argv[2], os.path.basename(sys.argv[2])
else:
    unit = sys.argv[1] if len(sys.argv) > 1 else "gemv_ref"
    asm = f"/tmp/_isa_{unit}.s"
    if not os.path.exists(asm) or os.path.getmtime(asm) < os.path.getmtime(os.path.join(ROOT, "jlama_amd", "csrc", "jh_p16.h")):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value",
                        "-S", "--offload-device-only", os.path.join(ROOT, "jlama_amd", "csrc", unit + ".hip"), "-o", asm], check=True, capture_output=True)
bad, seen = [], 0
fn, lines = None, []
def check():
    global seen
    if not fn or not re.search(r"p16_kernel|gemv_bf16r_kernel|gemv_t16_kernel", fn):
        return
    first_bar = next((i for i, l in enumerate(lines) if re.match(r"\s*s_barrier", l)), None)
    if first_bar is None:
        return
    ring = [i for i, l in enumerate(lines[:first_bar]) if re.match(r"\s*(global|buffer)_load_\w+ .*\bnt\b", l)]
    if not ring:
        return
    seen += 1
    R, last = len(ring), ring[-1]
    for i in range(last + 1, first_bar):
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", lines[i])
        if m and int(m.group(1)) < R:
            bad.append((fn, R, lines[i].strip(), i - last))
for line in open(asm):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        check(); fn, lines = m.group(1), []
        continue
    if line.strip() and not line.strip().startswith(";"):
        lines.append(line.rstrip())
check()
for f, R, w, d in bad:
    name = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip()
    print(f"{name[:120]}: `{w}` {d} instructions behind the last of {R} ring requests, in front of the prologue barrier")
print(f"{unit}: {seen} streaming kernels checked, {len(bad)} drain their ring in front of the prologue")
sys.exit(1 if bad else 0)
