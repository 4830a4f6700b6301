import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
from jlama_amd.model import HipLlamaModel
cfg = dict(S.LLAMA3_8B); cfg["n_layers"] = 2
if os.environ.get("JH_LIB"):   # A/B another build of the library on the same box (tools/build_variant.py)
    N.LIB_PATH = os.path.abspath(os.environ["JH_LIB"])
N.init(0)
N.options_from_env()   # tools only: JH_* environment variables become explicit library options
m = HipLlamaModel(cfg, ST.make_weights(cfg, seed=0, device="cuda"))
names = ["entry", "loads issued", "rope done", "scores", "softmax", "PV", "reduced", "published", "ticket", "combined"]
strict = os.environ.get("ATT_STRICT")            # reference order: the one-launch kernel (attn_p16_fused_kernel) stamps its own phases
if strict:
    names = ["entry", "loads issued", "rope barrier", "scores", "max+exp", "tile filed", "sum", "values stored", "w0 filed", "sum barrier", "divided", "chain start", "chain done", "-", "-", "-"]
for env in ({},):
    os.environ.update(env)
    s = m.session(int(os.environ.get("ATT_CTX", "1024")))
    if strict:
        s.set_strict(True)
    for pos in tuple(int(x) for x in os.environ.get("ATT_POS", "384").split(",")):
        out = np.zeros(256, dtype=np.int64)
        N.check(N.lib().jh_debug_attn_timeline(s.h, pos, N.ptr(out), 256))
        t = out.reshape(16, 16)
        base = t[t > 0].min()
        print(env, "pos", pos)
        for sp in range(16):
            if t[sp, 0] <= 0: continue
            row = [(names[k], round((t[sp, k] - base) / 100.0, 2)) for k in range(len(names)) if t[sp, k] > 0]
            print("  split", sp, " ".join(f"{n}={v}" for n, v in row))
    s.close()
    for k in env: del os.environ[k]
