#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage for libjlamahip (VGPR/SGPR/scratch/occupancy per kernel)."""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# usage: kernel_resources.py [kernel name filter] [translation unit = gemv_ref]   (units: jlama_amd/_native.py UNITS)
unit = sys.argv[2] if len(sys.argv) > 2 else "gemv_ref"
src = os.path.join(ROOT, "jlama_amd", "csrc", unit + ".hip")
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-c",
                      "-Wno-unused-value", "-Rpass-analysis=kernel-resource-usage", src, "-o", "/tmp/_jh_res.o"],
                     capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur); continue
    for key in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None and key not in cur:
            cur[key] = int(m.group(1))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'kernel':90s} VGPR SGPR scratch occ")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:90]:90s} {r.get('VGPRs',0):4d} {r.get('TotalSGPRs',0):4d} {r.get('ScratchSize [bytes/lane]',0):7d} {r.get('Occupancy [waves/SIMD]',0):3d}")
