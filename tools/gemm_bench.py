import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
from jlama_amd import _native as N
N.init(0)
N.options_from_env()   # tools only: JH_* environment variables become explicit library options
for kind, name, bpw in ((2, "I8xQ4t", 0.625), (3, "BF16t", 2.0), (0, "I8xQ4", 0.625), (1, "BF16", 2.0)):
    shapes = ((129, 4096, 4096), (129, 28672, 4096), (129, 4096, 14336), (256, 28672, 4096), (32, 28672, 4096))
    if os.environ.get("GB_MODEL_SHAPES"):
        shapes = ((129, 6144, 4096), (129, 4096, 4096), (129, 14336, 4096), (129, 4096, 14336), (256, 14336, 4096))
    if os.environ.get("GB_PREFILL_SHAPES"):   # the four GEMMs of a Llama-3-8B prefill layer at 129 and 256 rows
        shapes = tuple((m, n, k) for m in (129, 256) for (n, k) in ((6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)))
    if os.environ.get("GB_KINDS") and str(kind) not in os.environ["GB_KINDS"].split(","):
        continue
    for (m, n, k) in shapes:
        copies = max(1, int(600e6 / (n * k * bpw)))
        ms = C.c_double()
        N.check(N.lib().jh_gemm_bench(kind, m, n, k, copies, 3, C.byref(ms)))
        fl = 2.0 * m * n * k
        print(f"{name:6s} M={m:3d} N={n:5d} K={k:5d}: {ms.value*1e3:8.1f} us  {fl/ms.value/1e9:8.1f} TFLOP/s  weights {n*k*bpw/ms.value/1e6:7.1f} GB/s", flush=True)
