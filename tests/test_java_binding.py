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
