"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/jlama_hip.h declares, fails loudly (no CPU fallback), and the product never touches oracle/."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "jlama_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(jh_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from jlama_amd import _native as N
    assert os.path.exists(N.LIB_PATH), "run __graft_entry__.build() first"
    L = C.CDLL(N.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 45
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    # the ctypes prototype table covers exactly the header
    assert sorted(N.EXPORTS) == declared


def test_library_is_built_from_the_current_sources():
    """The .so carries the hash of the sources it was compiled from (jh_source_hash): a stale binary -- edited kernels,
    forgotten rebuild -- must fail here, not silently test old code on the GPU box."""
    from jlama_amd import _native as N
    assert N.built_hash() == N.source_hash(), "libjlamahip.so is stale: run __graft_entry__.build()"
    L = C.CDLL(N.LIB_PATH)
    L.jh_source_hash.restype = C.c_char_p
    assert L.jh_source_hash().decode() == N.source_hash()


def test_only_c_abi_symbols_are_exported():
    from jlama_amd import _native as N
    out = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    jh = [s for s in syms if s.startswith("jh_")]
    assert set(jh) == set(_declared())


def test_fails_loudly_without_gpu_or_library(monkeypatch):
    from jlama_amd import _native as N
    L = N.lib()
    assert L.jh_name().decode().startswith("HIP")
    assert L.jh_parallel_split_size() == 1          # NativeGPUTensorOperations.java:98-101
    assert L.jh_preferred_working_qtype() == N.DT_I8
    try:
        N.init(0)
        has_gpu = True
    except N.JhError as e:
        has_gpu = False
        assert e.code == N.JH_ERR_NO_DEVICE
    if not has_gpu:
        import numpy as np
        from jlama_amd.hip_tensor_operations import HipTensorOperations
        with pytest.raises(N.JhError):
            HipTensorOperations()
        x = np.zeros(32, dtype=np.float32)
        assert L.jh_scale_f32(2.0, N.ptr(x), 0, 32) == N.JH_ERR_NO_DEVICE   # compute calls refuse, no CPU path
    # missing library => hard error, not a fallback
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", "/nonexistent/libjlamahip.so")
    with pytest.raises(RuntimeError):
        N.lib()


def test_host_side_geometry_and_rope_do_not_need_a_gpu():
    """jh_kv_page_geometry / jh_rope_table are host-side restatements inside the product; check them against the
    reference facts directly (no oracle involved)."""
    import json
    import numpy as np
    from jlama_amd import _native as N, kv
    assert kv.page_geometry(32, 8192, 1024) == (32, 32)
    assert kv.page_geometry(16, 131072, 512) == (16, 128)
    assert kv.page_geometry(10, 8192, 1024) == (10, 102)
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "rope_kat.json")))
    t = np.empty((kat["context"] * 64, 2), dtype=np.float32)
    N.check(N.lib().jh_rope_table(128, kat["context"], kat["theta"], 1.0, N.ptr(t)))
    np.testing.assert_allclose(t[64:128, 1], kat["sin_pos1"], atol=1e-4)
    np.testing.assert_allclose(t[64 * 64:65 * 64, 1], kat["sin_pos64"], atol=1e-4)


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under jlama_amd/ or include/ may import, link or load it."""
    bad = []
    for base in ("jlama_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                    txt = open(os.path.join(dp, fn), errors="replace").read()
                    for m in re.finditer(r"^\s*(from|import)\s+oracle|libjlama_oracle|jlama_oracle\.c|oracle/_ref|libjlama_ref", txt, re.M):
                        bad.append((fn, m.group(0)))
    assert not bad, bad
    from jlama_amd import _native as N
    ldd = subprocess.run(["ldd", N.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "torch" not in ldd


def test_ctypes_prototypes_have_the_headers_arity():
    """Every prototype in jlama_amd/_native.py must take as many arguments as the declaration in include/jlama_hip.h
    (a drifted binding would pass garbage through the C ABI instead of failing)."""
    from jlama_amd import _native as N
    hdr = open(os.path.join(ROOT, "include", "jlama_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", "", hdr)
    arity = {}
    for m in re.finditer(r"\b(jh_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        arity[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    assert set(arity) == set(N.EXPORTS)
    bad = {name: (len(proto[1]), arity[name]) for name, proto in N._PROTOS.items() if len(proto[1]) != arity[name]}
    assert not bad, bad


def test_process_options_are_explicit_and_named():
    """jh_set_option / jh_clear_options work without a GPU; a name the library never asks for is an error (a misspelt switch must
    not be a silent no-op), the table of known names equals the names at the opt_int() call sites, and nothing but the
    environment snapshot of jh_init calls getenv."""
    from jlama_amd import _native as N
    N.clear_options()
    N.set_option("JH_ATTN_SPLITS", 3)
    N.set_option("JH_TILED_COPY", "transient")
    with pytest.raises(Exception) as e:
        N.set_option("JH_ATTN_SPLITZ", 3)
    assert "no option named" in str(e.value)
    N.clear_options()
    csrc = os.path.join(ROOT, "jlama_amd", "csrc")
    src = "\n".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h")))
    asked = set(re.findall(r'opt_int\("([A-Z0-9_]+)"', src))
    table = src[src.index("JH_KNOWN_OPTIONS[] = {"):]
    table = set(re.findall(r'"(JH_[A-Z0-9_]+)"', table[:table.index("};")]))
    assert asked == table, (sorted(asked - table), sorted(table - asked))
    listed = set(re.findall(r'"(JH_[A-Z0-9_]+)"', src[src.index("JH_ENV_OPTIONS[] = {"):].split("};")[0]))
    assert listed <= table and len(listed) <= 6
    code = "\n".join(open(os.path.join(ROOT, "jlama_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "jlama_amd", "csrc")))
    assert len(re.findall(r"\bgetenv\s*\(", code)) == 1


def test_host_mirror_library_is_a_caller_of_the_c_abi_only():
    """libjlamahost.so (csrc/host_mirror.cpp: the reference's host restated above the boundary) loads without a GPU, exports its
    jhost_* entry points, is built from the current source, depends on libjlamahip.so for every provider call (undefined jh_*
    symbols = exactly entry points include/jlama_hip.h declares) and on no device runtime, torch or oracle itself."""
    from jlama_amd import _native as N
    assert os.path.exists(N.HOST_LIB_PATH), "run __graft_entry__.build() first"
    assert open(N.HOST_LIB_PATH + ".key").read() == N.host_source_hash(), "libjlamahost.so is stale: run __graft_entry__.build()"
    L = N.host_lib()
    for s in N.HOST_EXPORTS:
        assert hasattr(L, s), s
    out = subprocess.run(["nm", "-D", N.HOST_LIB_PATH], capture_output=True, text=True).stdout
    undefined = {l.split()[-1] for l in out.splitlines() if " U " in l}
    used = {s for s in undefined if s.startswith("jh_")}
    assert used and used <= set(_declared()), sorted(used - set(_declared()))
    assert {"jh_gemm_q8_q4", "jh_gemm_f32_q4", "jh_gemm_f32", "jh_gemm_bf16", "jh_gemm_q8_q4_batch", "jh_quantize_q8", "jh_saxpy_batch_f32",
            "jh_scale_f32", "jh_accumulate_f32", "jh_maccumulate_f32", "jh_register_tensor"} <= used
    assert not [s for s in undefined if s.startswith(("hip", "jo_"))]
    defined = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert {s for s in defined if s.startswith("jhost_")} == set(N.HOST_EXPORTS)
    needed = subprocess.run(["readelf", "-d", N.HOST_LIB_PATH], capture_output=True, text=True).stdout
    assert "libjlamahip.so" in needed and "amdhip" not in needed and "torch" not in needed and "oracle" not in needed
    # without a GPU the host's constructor throws like the Java provider's would (TensorOperationsProvider then falls through)
    try:
        N.init(0)
    except N.JhError:
        from jlama_amd import synthetic as S
        from jlama_amd.host_mirror import HostAsIsModel
        with pytest.raises(N.JhError):
            HostAsIsModel(dict(S.TINY), {})
