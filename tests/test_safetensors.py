"""JQ4 safetensors reader -- CPU tests of the host-side format code.

Known answer: the reference's own parser test (jlama-tests/.../safetensors/TestParser.java:41-69): header bytes
0x59 + the JSON below + four floats must parse to a 2x2 F32 tensor [[1,2],[3,4]] with metadata foo=bar."""
import json
import os
import struct

import numpy as np
import pytest

from jlama_amd import _native as N, jq4, safetensors_jq4 as ST, synthetic as S


def test_reference_parser_known_answer(tmp_path):
    header = b'{"test":{"dtype":"F32","shape":[2,2],"data_offsets":[0,16]},"__metadata__":{"foo":"bar"}}'
    assert len(header) == 0x59                                      # the test's hard-coded preamble 5900000000000000
    blob = bytes.fromhex("5900000000000000") + header + struct.pack("<4f", 1.0, 2.0, 3.0, 4.0)
    infos, meta, start = ST.parse_header(blob)
    assert meta == {"foo": "bar"} and start == 8 + 0x59
    assert infos["test"].dtype == "F32" and infos["test"].shape == (2, 2) and infos["test"].data_offsets == (0, 16)
    p = tmp_path / "t.safetensors"
    p.write_bytes(blob)
    t = ST.SafeTensorsFile(str(p)).load("test")
    assert t["dtype"] == N.DT_F32 and t["shape"] == (2, 2)
    np.testing.assert_array_equal(t["data"], np.array([[1, 2], [3, 4]], dtype=np.float32))


def test_header_length_is_validated():
    with pytest.raises(ValueError, match="negative"):
        ST.parse_header(struct.pack("<q", -1) + b"{}")
    with pytest.raises(ValueError, match="exceeds"):
        ST.parse_header(struct.pack("<q", (1 << 30) + 1) + b"{}")
    with pytest.raises(ValueError, match="truncated"):
        ST.parse_header(struct.pack("<q", 100) + b"{}")
    with pytest.raises(ValueError, match="Unsupported Tensor type"):
        h = b'{"x":{"dtype":"Q5","shape":[1,32],"data_offsets":[0,20]}}'
        ST.parse_header(struct.pack("<q", len(h)) + h)


def _write_tiny(tmp_path, shards=1, tied=False):
    cfg = dict(S.TINY)
    if tied:
        cfg["tied"] = True
    w = S.make_weights(cfg, seed=3)
    d = str(tmp_path / f"tiny-{shards}-{int(tied)}")
    ST.write_llama_checkpoint(d, cfg, w, shards=shards)
    return cfg, w, d


@pytest.mark.parametrize("shards", [1, 3])
def test_jq4_checkpoint_round_trip(tmp_path, shards):
    """Q4 tag with LOGICAL shape but rows*cols/2 bytes + '<name>.qb' F32 scales (Weights.java:159-171); 1-D BF16 norm
    weights; multi-file index (SafeTensorIndex.java:87-119)."""
    cfg, w, d = _write_tiny(tmp_path, shards)
    ck = ST.Checkpoint(d)
    ti = ck.info("model.layers.0.self_attn.q_proj.weight")
    A, E = cfg["n_heads"] * cfg["head_size"], cfg["embedding_length"]
    assert ti.dtype == "Q4" and ti.shape == (A, E) and ti.data_offsets[1] - ti.data_offsets[0] == A * E // 2
    assert ck.info("model.layers.0.self_attn.q_proj.weight.qb").shape == (A, E // 32)
    assert ck.info("model.norm.weight").shape == (E,) and ck.info("model.norm.weight").dtype == "BF16"
    assert ck.model_dtype() in (N.DT_Q4, N.DT_F32)   # majority over tensors incl. the .qb siblings
    cfg2, w2 = ST.load_llama_weights(d)
    for k in ("embedding_length", "hidden_length", "n_heads", "n_kv_heads", "head_size", "n_layers", "vocab_size",
              "context_length", "rope_theta", "bos_token"):
        assert cfg2[k] == cfg[k], k
    assert abs(cfg2["rms_eps"] - cfg["rms_eps"]) < 1e-12 and cfg2["weight_dtype"] == N.DT_Q4
    assert set(w2) == set(w)
    for key, t in w.items():
        g = w2[key]
        assert g["dtype"] == t["dtype"] and tuple(g["shape"]) == tuple(t["shape"]), key
        np.testing.assert_array_equal(np.asarray(g["data"]).reshape(-1), np.asarray(t["data"]).reshape(-1))
        if t["scales"] is not None:
            np.testing.assert_array_equal(np.asarray(g["scales"]), t["scales"])
    # the bytes decode to the same weights (Q4ByteBufferTensor.get)
    q = w2[(0, S.W_Q)]
    np.testing.assert_array_equal(jq4.dequantize_q4(np.asarray(q["data"]), np.asarray(q["scales"])),
                                  jq4.dequantize_q4(w[(0, S.W_Q)]["data"], w[(0, S.W_Q)]["scales"]))


def test_row_windows_follow_the_q4_hack(tmp_path):
    """Weights.getLoadOffsets (:101-120): shard rows [r0, r0+n) start at r0*columnLength bytes, columnLength halved for Q4;
    the .qb sibling is windowed with the same rows."""
    cfg, w, d = _write_tiny(tmp_path)
    ck = ST.Checkpoint(d)
    name = "model.layers.1.mlp.gate_proj.weight"
    full = ck.load(name)
    H = cfg["hidden_length"]
    part = ck.load(name, row_window=(H // 4, H // 2))
    assert part["shape"] == (H // 2, cfg["embedding_length"])
    np.testing.assert_array_equal(part["data"], full["data"][H // 4:H // 4 + H // 2])
    np.testing.assert_array_equal(part["scales"], full["scales"][H // 4:H // 4 + H // 2])
    with pytest.raises(ValueError):
        ck.load(name, row_window=(H - 1, 2))
    with pytest.raises(ValueError, match="with offset"):
        ck.load("model.norm.weight", row_window=(0, 1))
    with pytest.raises(KeyError):
        ck.load("model.layers.9.mlp.gate_proj.weight")


def test_layer_shards_and_tied_head(tmp_path):
    """Only the shard's layers are read; embeddings on the first shard, final norm (+ tied table) on the last
    (LlamaModel.java:67-98,152-173)."""
    cfg, w, d = _write_tiny(tmp_path, tied=True)
    c0, w0 = ST.load_llama_weights(d, layer_range=(0, 1))
    c1, w1 = ST.load_llama_weights(d, layer_range=(1, 2))
    assert c0["tied"] and (-1, S.W_LMHEAD) not in w1
    assert {k[0] for k in w0 if k[0] >= 0} == {0} and {k[0] for k in w1 if k[0] >= 0} == {1}
    assert (-1, S.W_EMBED) in w0 and (-1, S.W_FINALNORM) not in w0
    assert (-1, S.W_FINALNORM) in w1 and (-1, S.W_EMBED) in w1       # tied: the last shard needs the table as LM head


def test_corrupt_files_are_rejected(tmp_path):
    cfg, w, d = _write_tiny(tmp_path)
    p = os.path.join(d, "model.safetensors")
    raw = bytearray(open(p, "rb").read())
    (hlen,) = struct.unpack_from("<q", raw, 0)
    hdr = json.loads(raw[8:8 + hlen])
    hdr["model.norm.weight"]["data_offsets"][1] += 2     # size no longer matches dtype x shape
    hb = json.dumps(hdr, separators=(",", ":")).encode()
    bad = os.path.join(d, "bad.safetensors")
    open(bad, "wb").write(struct.pack("<q", len(hb)) + hb + bytes(raw[8 + hlen:]))
    with pytest.raises(ValueError):
        ST.SafeTensorsFile(bad)
    open(bad, "wb").write(bytes(raw[:len(raw) - 100]))   # data section cut short
    with pytest.raises(ValueError, match="outside the file"):
        ST.SafeTensorsFile(bad)
