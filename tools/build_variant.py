#!/usr/bin/env python3
"""Build a VARIANT of libjlamahip.so into tools/ab/<name>.so with extra -D flags (lab builds for same-box A/B runs: rates move
~5 % box to box, so two builds are compared inside ONE gpurun call; tools/strict_bench.py takes the library from $JH_LIB).
usage: build_variant.py <name> [-DFLAG ...]"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jlama_amd import _native as N
name, extra = sys.argv[1], sys.argv[2:]
obj = os.path.join(ROOT, "tools", "ab", "obj_" + name)
os.makedirs(obj, exist_ok=True)
def cc(u):
    cmd = ["hipcc"] + N.CFLAGS + extra + ["-c", os.path.join(N.CSRC, u + ".hip"), "-o", os.path.join(obj, u + ".o")]
    if u == "core":
        cmd.insert(-3, '-DJH_SRC_HASH="variant-%s"' % name)
    subprocess.check_call(cmd)
with ThreadPoolExecutor(max_workers=os.cpu_count()) as ex:
    list(ex.map(cc, N.UNITS))
out = os.path.join(ROOT, "tools", "ab", name + ".so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [os.path.join(obj, u + ".o") for u in N.UNITS])
print(out)
