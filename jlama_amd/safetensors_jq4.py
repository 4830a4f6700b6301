"""JQ4 safetensors checkpoints -> device memory: the on-disk format either side of the hot path.

Restates the reference's reader (host code, no arithmetic):

* container: 8-byte little-endian header length, JSON header ``{name: {dtype, shape, data_offsets}, "__metadata__": {..}}``,
  then the data section; offsets are relative to the end of the header
  (jlama-core/.../safetensors/SafeTensorSupport.java:54-100; header length < 0 or > 1 GiB is rejected, :57-68);
* Jlama's extra dtype tags (core/safetensors/DType.java:46-50): ``"Q4"`` with the LOGICAL shape ``[rows, cols]`` but
  ``rows*cols/2`` data bytes (Q4ByteBufferTensor layout, see jq4.py), block scales in the sibling tensor ``"<name>.qb"``
  F32 ``[rows, cols/32]`` (Weights.java:159-171; written by SafeTensorSupport.quantizeModel :262-266); ``"I8"`` uses the
  same ``.qb`` convention (:172-180);
* row windows for layer/TP shards: ``offset = data_offsets[0] + shardOffset*columnLength`` with the column length
  halved for Q4 (Weights.getLoadOffsets :101-120, "Hack for Q4");
* multi-file checkpoints: ``model.safetensors.index.json`` ``weight_map`` name -> file (SafeTensorIndex.java:87-119);
* Llama tensor names and config.json fields: core/model/llama/LlamaModel.java:67-173, LlamaConfig.java:29-57.

Tensors are returned as numpy views of an ``np.memmap`` (read-only, zero copy), i.e. exactly the bytes
``jh_model_set_weight`` uploads; `load_llama` hands them to `HipLlamaModel` so a checkpoint goes page cache -> HBM with
no intermediate copy.  The writer exists so tests / synthetic checkpoints can be produced without a JVM; it emits what
quantizeModel emits (header order = write order, no alignment padding).
"""
import json
import os
import struct

import numpy as np

from . import _native as N
from . import synthetic as S

MAX_HEADER_LENGTH = 1 << 30  # SafeTensorSupport.java:55

_DTYPES = {"F32": (np.float32, N.DT_F32), "BF16": (np.uint16, N.DT_BF16), "F16": (np.uint16, None),
           "Q4": (np.uint8, N.DT_Q4), "I8": (np.int8, N.DT_I8)}


class TensorInfo:
    """core/safetensors/TensorInfo.java:24-45 -- dtype tag, logical shape, [begin, end) in the data section."""

    def __init__(self, dtype, shape, data_offsets):
        self.dtype = dtype
        self.shape = tuple(int(x) for x in shape)
        self.data_offsets = (int(data_offsets[0]), int(data_offsets[1]))

    def nbytes_expected(self):
        n = 1
        for d in self.shape:
            n *= d
        if self.dtype == "Q4":
            return n // 2
        return n * {"F32": 4, "BF16": 2, "F16": 2, "I8": 1}[self.dtype]

    def __repr__(self):
        return f"TensorInfo{{dType={self.dtype}, shape={list(self.shape)}, dataOffsets={list(self.data_offsets)}}}"


def parse_header(buf):
    """SafeTensorSupport.readTensorInfoMap (:54-100).  buf: bytes-like starting at file offset 0.
    Returns (infos sorted by data_offsets[0], metadata, data_start)."""
    if len(buf) < 8:
        raise ValueError("safetensors: file shorter than the 8-byte header length")
    (hlen,) = struct.unpack_from("<q", buf, 0)
    if hlen < 0:
        raise ValueError(f"Header length cannot be negative: {hlen}")
    if hlen > MAX_HEADER_LENGTH:
        raise ValueError(f"Header length {hlen} exceeds the maximum allowed length {MAX_HEADER_LENGTH}.")
    if len(buf) < 8 + hlen:
        raise ValueError("safetensors: truncated header")
    root = json.loads(bytes(buf[8:8 + hlen]).decode("utf-8"))
    infos, metadata = {}, {}
    for k, v in root.items():
        if k.lower() == "__metadata__":
            metadata = {str(a): str(b) for a, b in v.items()}
        else:
            if v["dtype"] not in _DTYPES:
                raise ValueError(f"Unsupported Tensor type: {v['dtype']} for {k}")
            infos[k] = TensorInfo(v["dtype"], v["shape"], v["data_offsets"])
    infos = dict(sorted(infos.items(), key=lambda kv: kv[1].data_offsets[0]))  # Comparable by dataOffsets[0] (:83-86)
    return infos, metadata, 8 + hlen


class SafeTensorsFile:
    """One .safetensors file, memory-mapped read-only (Weights.java:44-99)."""

    def __init__(self, path):
        self.path = path
        self.mm = np.memmap(path, dtype=np.uint8, mode="r")
        head = bytes(self.mm[:min(len(self.mm), 8)])
        (hlen,) = struct.unpack_from("<q", head, 0) if len(head) == 8 else (-1,)
        self.infos, self.metadata, self.data_start = parse_header(self.mm[:8 + max(0, min(hlen, MAX_HEADER_LENGTH))]
                                                                  if 0 <= hlen <= MAX_HEADER_LENGTH else head)
        for name, ti in self.infos.items():
            b, e = ti.data_offsets
            if b < 0 or e < b or self.data_start + e > len(self.mm):
                raise ValueError(f"safetensors: {name} data_offsets {ti.data_offsets} outside the file")
            if e - b != ti.nbytes_expected():
                raise ValueError(f"safetensors: {name} has {e - b} bytes, {ti.dtype}{list(ti.shape)} needs {ti.nbytes_expected()}")

    def names(self):
        return list(self.infos)

    def _raw(self, name, row_window=None):
        """bytes of `name`, optionally only rows [r0, r0+n) (Weights.getLoadOffsets :101-120)."""
        ti = self.infos.get(name)
        if ti is None:
            raise KeyError(f"{name} not found in weights")
        if len(ti.shape) < 1:
            raise ValueError(f"Invalid shape dimensions {len(ti.shape)} encountered for {name}")
        b, e = ti.data_offsets
        shape = ti.shape
        if row_window is not None:
            if len(ti.shape) != 2:
                raise ValueError(f"Invalid shape dimensions {len(ti.shape)} encountered for {name} with offset")
            r0, n = row_window
            rows, cols = ti.shape
            if r0 < 0 or n < 0 or r0 + n > rows:
                raise ValueError(f"row window {row_window} outside {name} with {rows} rows")
            elem = {"F32": 4, "BF16": 2, "F16": 2, "I8": 1, "Q4": 1}[ti.dtype]
            column_length = cols * elem
            if ti.dtype == "Q4":
                column_length //= 2  # "Hack for Q4" (:111-112)
            b = ti.data_offsets[0] + r0 * column_length
            e = b + n * column_length
            shape = (n, cols)
        return ti, shape, self.mm[self.data_start + b:self.data_start + e]

    def load(self, name, row_window=None):
        """-> {dtype, data, scales, shape} in the form HipLlamaModel.set_weight takes (views of the mmap, zero copy).
        Q4 / I8 pull their block scales from "<name>.qb" with the same row window (Weights.java:159-180)."""
        ti, shape, raw = self._raw(name, row_window)
        np_dt, jh_dt = _DTYPES[ti.dtype]
        if jh_dt is None:
            raise ValueError(f"Unsupported Tensor type: {ti.dtype} for {name}")
        scales = None
        if ti.dtype == "Q4":
            rows, cols = shape
            data = raw.view(np.uint8).reshape(rows, cols // 2)
        else:
            data = raw.view(np_dt).reshape(shape)
        if ti.dtype in ("Q4", "I8"):
            qi, qshape, qraw = self._raw(name + ".qb", row_window)
            if qi.dtype != "F32" or tuple(qshape) != (shape[0], shape[1] // 32):
                raise ValueError(f"{name}.qb must be F32 [{shape[0]}, {shape[1] // 32}], found {qi}")
            scales = qraw.view(np.float32).reshape(qshape)
        if len(shape) == 1:   # 1-D norm weights are used as one row
            shape = (1, shape[0])
            data = data.reshape(shape)
        return {"dtype": jh_dt, "data": data, "scales": scales, "shape": tuple(shape)}


class Checkpoint:
    """A model directory: model.safetensors, or model.safetensors.index.json + shards (SafeTensorIndex.java:87-119)."""

    def __init__(self, model_dir):
        self.dir = model_dir
        self.files = {}
        self.where = {}
        idx = os.path.join(model_dir, "model.safetensors.index.json")
        if os.path.exists(idx):
            with open(idx) as f:
                wm = json.load(f)["weight_map"]
            for name, fn in wm.items():
                if fn not in self.files:
                    self.files[fn] = SafeTensorsFile(os.path.join(model_dir, fn))
            for fn, sf in self.files.items():   # allTensorInfoMap: every tensor of every listed file, incl. ".qb" siblings
                for name in sf.names():
                    self.where[name] = fn
        else:
            fn = "model.safetensors"
            self.files[fn] = SafeTensorsFile(os.path.join(model_dir, fn))
            for name in self.files[fn].names():
                self.where[name] = fn

    def names(self):
        return list(self.where)

    def info(self, name):
        return self.files[self.where[name]].infos[name]

    def load(self, name, row_window=None):
        if name not in self.where:
            raise KeyError(f"{name} not found in weights")
        return self.files[self.where[name]].load(name, row_window)

    def model_dtype(self):
        """majority dtype over tensors (Weights.findDType, Weights.java:56-68), as a jh dtype id."""
        count = {}
        for name in self.where:
            d = self.info(name).dtype
            count[d] = count.get(d, 0) + 1
        return _DTYPES[max(count, key=count.get)][1]


def llama_config(model_dir):
    """config.json -> the cfg dict of synthetic.py (LlamaConfig.java:29-57, Config.java:253-274)."""
    with open(os.path.join(model_dir, "config.json")) as f:
        c = json.load(f)
    E, heads = int(c["hidden_size"]), int(c["num_attention_heads"])
    rs = c.get("rope_scaling")
    scaling = float(rs["factor"]) if rs and rs.get("rope_type") == "linear" else 1.0   # only "linear" is honoured (:56)
    eos = c.get("eos_token_id")
    if eos is None:                       # LlamaConfig requires eos_token_id (an int or a list, LlamaConfig.java:36-40)
        eos_tokens = []
    else:
        eos_tokens = [int(t) for t in eos] if isinstance(eos, list) else [int(eos)]
    # headSize: the reference derives it as embeddingLength / numberOfHeads for Llama models (LlamaConfig passes no headSize,
    # Config.java:95) and does not read a `head_dim` key.  EXTENSION (documented in DESIGN.md): an explicit `head_dim` is
    # honoured, as HF does -- checkpoints where the two agree (every published Llama-3 / Mistral one) load identically
    head_size = int(c["head_dim"]) if "head_dim" in c else E // heads
    return dict(embedding_length=E, hidden_length=int(c["intermediate_size"]), n_heads=heads,
                n_kv_heads=int(c.get("num_key_value_heads", heads)), head_size=head_size,
                n_layers=int(c["num_hidden_layers"]), vocab_size=int(c["vocab_size"]),
                context_length=int(c["max_position_embeddings"]), rms_eps=float(c["rms_norm_eps"]),
                rope_theta=float(c.get("rope_theta") or 10000.0), rope_scaling=scaling,
                bos_token=int(c.get("bos_token_id", 1)), eos_tokens=eos_tokens,
                tied=bool(c.get("tie_word_embeddings", False)))


_LAYER_NAMES = {S.W_Q: "self_attn.q_proj.weight", S.W_K: "self_attn.k_proj.weight", S.W_V: "self_attn.v_proj.weight",
                S.W_O: "self_attn.o_proj.weight", S.W_GATE: "mlp.gate_proj.weight", S.W_UP: "mlp.up_proj.weight",
                S.W_DOWN: "mlp.down_proj.weight", S.W_NORM1: "input_layernorm.weight",
                S.W_NORM2: "post_attention_layernorm.weight"}


def tensor_name(layer, slot):
    """HF names read by LlamaModel.loadTransformerBlockWeights / loadInputWeights / loadOutputWeights (:67-173)."""
    if layer < 0:
        return {S.W_EMBED: "model.embed_tokens.weight", S.W_FINALNORM: "model.norm.weight", S.W_LMHEAD: "lm_head.weight"}[slot]
    return f"model.layers.{layer}.{_LAYER_NAMES[slot]}"


def load_llama_weights(model_dir, layer_range=None):
    """(cfg, weights) for HipLlamaModel: only this shard's layers are touched (DistributedContext.java:75-77); the
    embedding table goes to the first shard, final norm + LM head to the last (LlamaModel.java:67-98,152-173).
    lm_head.weight absent => tied to the embedding table (:155-158)."""
    cfg = llama_config(model_dir)
    ck = Checkpoint(model_dir)
    L = cfg["n_layers"]
    ls, le = layer_range if layer_range else (0, L)
    qd = ck.info(tensor_name(ls, S.W_Q)).dtype
    if qd not in ("Q4", "BF16"):          # resident models are JQ4 (I8 activations) or BF16 (include/jlama_hip.h: jh_model_create)
        raise ValueError(f"{tensor_name(ls, S.W_Q)} is {qd}: this loader handles JQ4 ('Q4' + '.qb' scales) and BF16 checkpoints; "
                         "quantize F32/F16 checkpoints with the reference's `jlama quantize` first")
    cfg["weight_dtype"] = N.DT_Q4 if qd == "Q4" else N.DT_BF16
    for slot in (S.W_NORM1, S.W_NORM2):
        nd = ck.info(tensor_name(ls, slot)).dtype
        if nd not in ("BF16", "F32"):
            raise ValueError(f"{tensor_name(ls, slot)} is {nd}: norm weights must be BF16 or F32")
    w = {}
    tied = "lm_head.weight" not in ck.where
    cfg["tied"] = tied
    if ls == 0 or (tied and le == L):
        w[(-1, S.W_EMBED)] = ck.load(tensor_name(-1, S.W_EMBED))
    for li in range(ls, le):
        for slot in _LAYER_NAMES:
            w[(li, slot)] = ck.load(tensor_name(li, slot))
    if le == L:
        w[(-1, S.W_FINALNORM)] = ck.load(tensor_name(-1, S.W_FINALNORM))
        if not tied:
            w[(-1, S.W_LMHEAD)] = ck.load(tensor_name(-1, S.W_LMHEAD))
    return cfg, w


def load_llama(model_dir, layer_range=None, device=0):
    """Checkpoint directory -> resident HipLlamaModel (mmap'd bytes go straight to jh_model_set_weight)."""
    from .model import HipLlamaModel
    cfg, w = load_llama_weights(model_dir, layer_range)
    return HipLlamaModel(cfg, w, layer_range=layer_range, device=device)


# ------------------------------------------------------------------------------------------------ writer (tests / tools)
_TAGS = {N.DT_F32: "F32", N.DT_BF16: "BF16", N.DT_Q4: "Q4", N.DT_I8: "I8"}


def write_safetensors(path, tensors, metadata=None):
    """tensors: {name: {dtype, data, scales, shape}} -> one file in quantizeModel's layout (SafeTensorSupport.java:226-311):
    data in insertion order, each Q4/I8 tensor immediately followed by its ".qb" F32 scales; header written after."""
    header, blobs, pos = {}, [], 0

    def put(name, tag, shape, arr):
        nonlocal pos
        b = np.ascontiguousarray(arr).tobytes()
        header[name] = {"dtype": tag, "shape": [int(x) for x in shape], "data_offsets": [pos, pos + len(b)]}
        blobs.append(b)
        pos += len(b)

    for name, t in tensors.items():
        tag = _TAGS[t["dtype"]]
        shape = t.get("disk_shape", t["shape"])
        put(name, tag, shape, t["data"])
        if tag in ("Q4", "I8"):
            put(name + ".qb", "F32", (t["shape"][0], t["shape"][1] // 32), t["scales"])
    if metadata:
        header["__metadata__"] = {str(k): str(v) for k, v in metadata.items()}
    hb = json.dumps(header, separators=(",", ":")).encode("utf-8")
    with open(path, "wb") as f:
        f.write(struct.pack("<q", len(hb)))
        f.write(hb)
        for b in blobs:
            f.write(b)


def write_llama_checkpoint(model_dir, cfg, weights, shards=1):
    """Synthetic weights (synthetic.make_weights) -> a checkpoint directory a Jlama user would recognise:
    config.json + model.safetensors (or an index + `shards` files)."""
    os.makedirs(model_dir, exist_ok=True)
    conf = {"model_type": "llama", "architectures": ["LlamaForCausalLM"], "hidden_size": cfg["embedding_length"],
            "intermediate_size": cfg["hidden_length"], "num_attention_heads": cfg["n_heads"],
            "num_key_value_heads": cfg["n_kv_heads"], "head_dim": cfg["head_size"], "num_hidden_layers": cfg["n_layers"],
            "rms_norm_eps": cfg["rms_eps"], "vocab_size": cfg["vocab_size"], "max_position_embeddings": cfg["context_length"],
            "bos_token_id": cfg["bos_token"], "eos_token_id": cfg.get("eos_tokens", [2])[0], "hidden_act": "silu",
            "rope_theta": cfg["rope_theta"], "tie_word_embeddings": bool(cfg.get("tied", False))}
    with open(os.path.join(model_dir, "config.json"), "w") as f:
        json.dump(conf, f)
    named = {}
    for (layer, slot), t in weights.items():
        t = dict(t)
        if slot in (S.W_NORM1, S.W_NORM2, S.W_FINALNORM):
            t["disk_shape"] = (t["shape"][1],)   # norm weights are 1-D on disk
        named[tensor_name(layer, slot)] = t
    if shards <= 1:
        write_safetensors(os.path.join(model_dir, "model.safetensors"), named, {"format": "pt"})
        return
    names = list(named)
    per = (len(names) + shards - 1) // shards
    weight_map = {}
    for i in range(shards):
        part = {n: named[n] for n in names[i * per:(i + 1) * per]}
        fn = f"model-{i + 1:05d}-of-{shards:05d}.safetensors"
        write_safetensors(os.path.join(model_dir, fn), part, {"format": "pt"})
        for n in part:
            weight_map[n] = fn
    with open(os.path.join(model_dir, "model.safetensors.index.json"), "w") as f:
        json.dump({"metadata": {}, "weight_map": weight_map}, f)
