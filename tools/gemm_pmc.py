"""One tiled prefill GEMM shape repeated, for rocprofv3 --pmc / --kernel-trace (GP_M, GP_N, GP_K; GP_KIND 2 = I8xQ4, 3 = BF16; JH_GEMM_* / JH_BF16_* select the kernel)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jlama_amd import _native as N
N.init(0)
N.options_from_env()   # tools only: JH_* environment variables become explicit library options
m, n, k = int(os.environ.get("GP_M", 129)), int(os.environ.get("GP_N", 28672)), int(os.environ.get("GP_K", 4096))
ms = C.c_double()
N.check(N.lib().jh_gemm_bench(int(os.environ.get("GP_KIND", 2)), m, n, k, max(1, int(600e6 / (n * k * 0.625))), 2, C.byref(ms)))
print(f"M={m} N={n} K={k}: {ms.value*1e3:.1f} us")
