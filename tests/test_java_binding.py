"""The Java half of the drop-in (java/) cannot be compiled here (no JDK in the image), so what CAN drift silently is
checked mechanically: every Panama FFM descriptor in NativeHip.java against the C declaration in include/jlama_hip.h
(argument count AND kinds: int / long / float / pointer), every NativeHip method the provider calls exists with that
arity, and the provider overrides the whole TensorOperations interface."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JAVA = os.path.join(ROOT, "java", "src", "main")


def _header_signatures():
    hdr = open(os.path.join(ROOT, "include", "jlama_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", "", hdr)
    sigs = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(jh_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()

        def kind(t):
            t = t.strip()
            if "*" in t:
                return "p"
            if re.search(r"\bint64_t\b|\blong\b", t):
                return "l"
            if re.search(r"\bfloat\b", t):
                return "f"
            if re.search(r"\bdouble\b", t):
                return "d"
            return "i"
        kinds = "" if args in ("", "void") else "".join(kind(a) for a in args.split(","))
        sigs[name] = (kind(ret) if ret != "void" else "v", kinds)
    return sigs


def test_ffm_descriptors_match_the_c_header():
    src = open(os.path.join(JAVA, "java22/com/github/tjake/jlama/tensor/operations/cnative/NativeHip.java")).read()
    sigs = _header_signatures()
    handles = re.findall(r'MethodHandle (jh_\w+) = h\("(jh_\w+)", (JAVA_INT|JAVA_LONG|ADDRESS)(?:, sig\("([ilfp]*)"\))?\)', src)
    assert len(handles) >= 24
    res = {"JAVA_INT": "i", "JAVA_LONG": "l", "ADDRESS": "p"}
    for var, name, ret, args in handles:
        assert var == name
        assert name in sigs, name
        assert (res[ret], args) == sigs[name], (name, (res[ret], args), sigs[name])
    # every public static wrapper passes exactly the handle's arguments
    for var, name, ret, args in handles:
        m = re.search(r"public static \w+ " + name + r"\(([^)]*)\)", src)
        assert m, name
        nparams = 0 if not m.group(1).strip() else len(m.group(1).split(","))
        assert nparams == len(args), (name, nparams, len(args))
    assert src.count("{") == src.count("}")


def test_provider_covers_the_interface_and_only_calls_bound_entry_points():
    prov = open(os.path.join(JAVA, "java/com/github/tjake/jlama/tensor/operations/HipTensorOperations.java")).read()
    bind = open(os.path.join(JAVA, "java22/com/github/tjake/jlama/tensor/operations/cnative/NativeHip.java")).read()
    bound = {m.group(1): (0 if not m.group(2).strip() else len(m.group(2).split(",")))
             for m in re.finditer(r"public static \w+ (jh_\w+)\(([^)]*)\)", bind)}
    for m in re.finditer(r"NativeHip\.(jh_\w+)\(", prov):
        assert m.group(1) in bound, m.group(1)
    # argument counts of the GEMM calls (the long ones are where a dropped argument would hide)
    for m in re.finditer(r"NativeHip\.(jh_gemm\w+)\((.*?)\);", prov, flags=re.S):
        depth, n, cur = 0, 0, m.group(2)
        for ch in cur:
            depth += ch in "([" ; depth -= ch in ")]"
            n += (ch == "," and depth == 0)
        assert n + 1 == bound[m.group(1)], (m.group(1), n + 1, bound[m.group(1)])
    for method in ("name", "parallelSplitSize", "preferredWorkingQuantizedType", "registerModelTensor", "batchDotProduct",
                   "dotProductBatchChunk", "accumulate", "maccumulate", "saxpy", "scale", "quantize"):
        assert re.search(r"public \S+ " + method + r"\(", prov), method
    assert prov.count("{") == prov.count("}") and prov.count("(") == prov.count(")")
    patch = open(os.path.join(ROOT, "java", "TensorOperationsProvider.patch")).read()
    assert "HipTensorOperations" in patch and "jlama.force_hip_tensor_operations" in patch


def test_tier2_ffm_descriptors_and_the_resident_model_host():
    """NativeHipModel.java (resident model / session / pipeline / tensor-parallel entry points) against the header, and
    HipResidentLlama.java: only bound entry points, the bound arity, and the jh_config layout field for field."""
    src = open(os.path.join(JAVA, "java22/com/github/tjake/jlama/tensor/operations/cnative/NativeHipModel.java")).read()
    sigs = _header_signatures()
    handles = re.findall(r'MethodHandle (jh_\w+) = h\("(jh_\w+)", (JAVA_INT|JAVA_LONG|ADDRESS)(?:, sig\("([ilfp]*)"\))?\)', src)
    assert len(handles) >= 35
    res = {"JAVA_INT": "i", "JAVA_LONG": "l", "ADDRESS": "p"}
    bound = {}
    for var, name, ret, args in handles:
        assert var == name and name in sigs, name
        assert (res[ret], args) == sigs[name], (name, (res[ret], args), sigs[name])
        m = re.search(r"public static \w+ " + name + r"\(([^)]*)\)", src)
        assert m, name
        bound[name] = 0 if not m.group(1).strip() else len(m.group(1).split(","))
        assert bound[name] == len(args), name
    for must in ("jh_model_create", "jh_model_set_weight", "jh_session_create", "jh_forward", "jh_sample", "jh_decode_n", "jh_session_set_eos",
                 "jh_session_set_strict", "jh_stage_decode_async", "jh_session_stream", "jh_pipeline_create", "jh_tp_group_create"):
        assert must in bound, must
    host = open(os.path.join(JAVA, "java/com/github/tjake/jlama/model/hip/HipResidentLlama.java")).read()
    for m in re.finditer(r"NativeHipModel\.(jh_\w+)\((.*?)\)\)?;", host, flags=re.S):
        name = m.group(1)
        assert name in bound, name
        depth, n = 0, 0
        for ch in m.group(2):
            depth += ch in "([" ; depth -= ch in ")]"
            n += (ch == "," and depth == 0)
        assert n + 1 == bound[name], (name, n + 1, bound[name])
    # jh_config: the Java struct layout lists the C fields in order
    hdr = open(os.path.join(ROOT, "include", "jlama_hip.h")).read()
    body = re.search(r"typedef struct jh_config \{(.*?)\} jh_config;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            c_fields += [f.strip() for f in decl.split(None, 1)[1].split(",")]
    j_fields = re.findall(r'withName\("(\w+)"\)', host)
    assert j_fields == c_fields, (j_fields, c_fields)
    assert host.count("{") == host.count("}") and host.count("(") == host.count(")")
