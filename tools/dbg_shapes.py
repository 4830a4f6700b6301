import sys, os, numpy as np, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
from jlama_amd import synthetic as S, _native as N
from jlama_amd.model import HipLlamaModel
from jlama_amd.hip_tensor_operations import HipTensorOperations
from jlama_amd.jq4 import Tensor
ops = HipTensorOperations()
rng = np.random.default_rng(0)
for (n,k) in [(256,2048),(128,8192),(512,14336),(1024,4096)]:
    w=(rng.standard_normal((n,k))/np.sqrt(k)).astype(np.float32); a=rng.standard_normal((1,k)).astype(np.float32)
    B=Tensor.q4(w); A=ops.quantize(Tensor.f32(a),2,0,k); R=Tensor.zeros(1,n)
    ops.batchDotProduct(R,A,B,0,0,k); print("gemv i8q4", n,k, float(np.abs(R.data).max()), flush=True)
    Rf=Tensor.zeros(1,n); ops.batchDotProduct(Rf,Tensor.f32(a),B,0,0,k); print("gemv f32q4", n,k, float(np.abs(Rf.data).max()), flush=True)
name = sys.argv[1]
cfg = dict(getattr(S, name)); cfg.update(n_layers=1, vocab_size=2048, context_length=512, bos_token=1); cfg.pop("tied", None)
w = S.make_weights(cfg, seed=7)
hm = HipLlamaModel(cfg, w); print("model ok", flush=True)
hs = hm.session(128); print("session ok", hs.page_info(), flush=True)
hs.set_tap_layer(0)
x = hs.forward([5], 0); print("fwd ok", float(np.abs(x).max()), flush=True)
for nm,n in [("query",cfg["n_heads"]*cfg["head_size"]),("after_attention",cfg["n_heads"]*cfg["head_size"]),("attn_res",cfg["embedding_length"]),("ff_h",cfg["hidden_length"]),("post_ff_res",cfg["embedding_length"])]:
    print(nm, float(np.abs(hs.tap(nm,n)).max()), flush=True)
print(hs.sample(0.0,0.5), flush=True)
prompt = S.prompt_tokens(cfg, n=40, seed=8)
hs2 = hm.session(128)
for i, t in enumerate(prompt):
    x = hs2.forward([t], i)
    print("row", i, float(np.abs(x).max()), flush=True)
print("batch", flush=True)
hs3 = hm.session(128)
x = hs3.batch_forward(prompt, 0)
print("batch ok", x.shape, flush=True)
