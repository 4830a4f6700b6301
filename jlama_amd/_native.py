"""Loader / builder of libjlamahip.so and its ctypes prototypes (include/jlama_hip.h)."""
import ctypes as C
import hashlib
import os
import sys
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(HERE, "lib", "libjlamahip.so")
OBJ_DIR = os.path.join(HERE, "lib", "obj")
CSRC = os.path.join(HERE, "csrc")
# translation units of libjlamahip.so (compiled in parallel, one object each)
UNITS = ["core", "tier1", "model", "layers", "prefill", "decode", "tp", "pipeline", "probe", "gemm_launch",
         "gemv_q4_a", "gemv_q4_b", "gemv_q4_c", "gemv_ref", "gemv_bf16"]
SRCS = [os.path.join(CSRC, u + ".hip") for u in UNITS]
HDRS = [os.path.join(CSRC, h) for h in ("jh_kernels.h", "jh_p16.h", "jh_t16.h", "jh_seqsum.h", "jh_bf16r.h", "jh_host.h", "jh_launch.h")] + \
       [os.path.join(ROOT, "include", "jlama_hip.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value"]

JH_OK, JH_ERR_NO_DEVICE, JH_ERR_OOM, JH_ERR_UNSUPPORTED, JH_ERR_INVALID, JH_ERR_HIP = 0, -1, -2, -3, -4, -5
DT_F32, DT_BF16, DT_I8, DT_Q4 = 0, 1, 2, 3
(W_Q, W_K, W_V, W_O, W_GATE, W_UP, W_DOWN, W_NORM1, W_NORM2, W_EMBED, W_LMHEAD, W_FINALNORM) = range(12)
TAP = {"input_emb": 0, "query": 2, "key": 3, "value": 4, "query+rope": 5, "key+rope": 6, "after_attention": 7,
       "attn_res": 8, "ff_h": 10, "post_ff_res": 11}


class JhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"jlama-hip error {code}: {msg}")
        self.code = code


class UnsupportedOperation(JhError):
    """Mirror of Java's UnsupportedOperationException for unsupported dtype pairs."""


class Config(C.Structure):
    _fields_ = [("embedding_length", C.c_int32), ("hidden_length", C.c_int32), ("n_heads", C.c_int32),
                ("n_kv_heads", C.c_int32), ("head_size", C.c_int32), ("n_layers", C.c_int32),
                ("vocab_size", C.c_int32), ("context_length", C.c_int32), ("weight_dtype", C.c_int32),
                ("layer_start", C.c_int32), ("layer_end", C.c_int32), ("rms_eps", C.c_float),
                ("rope_theta", C.c_float), ("rope_scaling", C.c_float)]


def source_hash():
    """sha256 over the library's sources; compiled into the .so (jh_source_hash) so a stale binary is detectable."""
    h = hashlib.sha256()
    for p in SRCS + HDRS:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(CFLAGS).encode())   # a flag change (-ffp-contract, visibility ...) makes the binary stale too
    return h.hexdigest()[:32]


def built_hash():
    """The source hash the existing .so was compiled from, or None (missing / predates the hash).  Read by scanning the
    file: dlopen-ing a stale library here would pin it in this process."""
    try:
        with open(LIB_PATH, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(b"JHSRCHASH:")
    return blob[i + 10:i + 42].decode(errors="replace") if i >= 0 else None


def _unit_key(unit, hdr_digest):
    h = hashlib.sha256(hdr_digest)
    with open(os.path.join(CSRC, unit + ".hip"), "rb") as f:
        h.update(f.read())
    h.update(" ".join(CFLAGS).encode())
    return h.hexdigest()[:32]


def build(force=False, verbose=False, jobs=None):
    """Compile libjlamahip.so for gfx950 (hipcc cross-compiles without a GPU): one object per translation unit, in parallel,
    then one link.  Without `force` the whole step is skipped only when the existing binary carries the hash of the current
    sources, and an object is reused only when ITS source, every header and the flags hash to the key stored beside it
    (never by timestamps).  `force` recompiles every unit."""
    import fcntl
    os.makedirs(OBJ_DIR, exist_ok=True)
    want = source_hash()
    if not force and built_hash() == want:
        return LIB_PATH
    # several ranks of one torchrun may get here at once: one builds, the others wait and find the fresh binary
    with open(os.path.join(os.path.dirname(LIB_PATH), ".build.lock"), "w") as lockf:
        fcntl.flock(lockf, fcntl.LOCK_EX)
        try:
            if not force and built_hash() == want:
                return LIB_PATH
            return _build_locked(want, force, verbose, jobs)
        finally:
            fcntl.flock(lockf, fcntl.LOCK_UN)


def _build_locked(want, force, verbose, jobs):
    from concurrent.futures import ThreadPoolExecutor
    hd = hashlib.sha256()
    for p in HDRS:
        with open(p, "rb") as f:
            hd.update(f.read())
    hdr_digest = hd.digest()

    def compile_unit(unit):
        obj, keyf = os.path.join(OBJ_DIR, unit + ".o"), os.path.join(OBJ_DIR, unit + ".key")
        key = _unit_key(unit, hdr_digest) + (want if unit == "core" else "")   # core.hip carries the source hash of the whole library
        if not force and os.path.exists(obj) and os.path.exists(keyf) and open(keyf).read() == key:
            return unit, False
        if os.path.exists(keyf):
            os.remove(keyf)                       # a failed compile must not leave an old key beside a clobbered object
        tmp = obj + ".tmp"
        cmd = ["hipcc"] + CFLAGS + ["-c", os.path.join(CSRC, unit + ".hip"), "-o", tmp]
        if unit == "core":
            cmd.insert(-3, f'-DJH_SRC_HASH="{want}"')
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(tmp, obj)
        with open(keyf, "w") as f:
            f.write(key)
        return unit, True

    with ThreadPoolExecutor(max_workers=jobs or min(len(UNITS), os.cpu_count() or 4)) as ex:
        done = list(ex.map(compile_unit, UNITS))
    if verbose:
        print("compiled:", [u for u, c in done if c], "reused:", [u for u, c in done if not c], flush=True)
    link = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH + ".tmp"] + [os.path.join(OBJ_DIR, u + ".o") for u in UNITS]
    subprocess.check_call(link)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    build_host(force=True, verbose=verbose)
    return LIB_PATH


# libjlamahost.so: the reference's host restated in C++ ABOVE the C ABI (csrc/host_mirror.cpp) -- the caller, not the backend.
# Plain g++ (no device code); it links only the exported C ABI of libjlamahip.so.
HOST_LIB_PATH = os.path.join(HERE, "lib", "libjlamahost.so")
HOST_SRC = os.path.join(CSRC, "host_mirror.cpp")


def host_source_hash():
    h = hashlib.sha256()
    for p in (HOST_SRC, os.path.join(ROOT, "include", "jlama_hip.h")):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:32]


def build_host(force=False, verbose=False):
    keyf = HOST_LIB_PATH + ".key"
    want = host_source_hash()
    if not force and os.path.exists(HOST_LIB_PATH) and os.path.exists(keyf) and open(keyf).read() == want:
        return HOST_LIB_PATH
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-fopenmp", "-fvisibility=hidden", HOST_SRC, "-o", HOST_LIB_PATH,
           "-L" + os.path.dirname(LIB_PATH), "-ljlamahip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(keyf, "w") as f:
        f.write(want)
    return HOST_LIB_PATH


_lib = None

_i, _l, _f, _d, _p = C.c_int, C.c_int64, C.c_float, C.c_double, C.c_void_p
_PROTOS = {
    "jh_init": (_i, [_i, _p]),
    "jh_name": (C.c_char_p, []),
    "jh_parallel_split_size": (_i, []),
    "jh_preferred_working_qtype": (_i, []),
    "jh_last_error": (C.c_char_p, []),
    "jh_synchronize": (_i, []),
    "jh_tp_group_status": (_i, [_p, _p, _i]),
    "jh_tp_rank_signature": (_i, [_p, _p]),
    "jh_set_option": (_i, [C.c_char_p, _i]),
    "jh_clear_options": (_i, []),
    "jh_register_tensor": (_l, [_p, _l]),
    "jh_unregister_tensor": (_i, [_l]),
    "jh_gemm_q8_q4": (_i, [_l, _l, _p, _p, _i, _p, _p, _i, _p, _i] + [_i] * 9),
    "jh_gemm_f32_q4": (_i, [_l, _l, _p, _i, _p, _p, _i, _p, _i] + [_i] * 8),
    "jh_gemm_f32": (_i, [_l, _p, _i, _p, _i, _p, _i] + [_i] * 7),
    "jh_gemm_bf16": (_i, [_l, _p, _i, _p, _i, _p, _p, _i] + [_i] * 7),
    "jh_gemm_f32_bf16": (_i, [_l, _p, _i, _p, _i, _p, _p, _i] + [_i] * 7),
    "jh_gemm_f32_batch": (_i, [_i, _p, _p, _i, _p, _i, _p, _i] + [_i] * 7),
    "jh_gemm_bf16_batch": (_i, [_i, _p, _p, _i, _p, _i, _p, _p, _i] + [_i] * 7),
    "jh_gemm_f32_bf16_batch": (_i, [_i, _p, _p, _i, _p, _i, _p, _p, _i] + [_i] * 7),
    "jh_gemm_q8_q4_batch": (_i, [_i, _p, _p, _p, _p, _i, _p, _p, _i, _p, _i] + [_i] * 9),
    "jh_gemm_f32_q4_batch": (_i, [_i, _p, _p, _p, _i, _p, _p, _i, _p, _i] + [_i] * 8),
    "jh_accumulate_f32": (_i, [_p, _p, _i, _i]),
    "jh_accumulate_f32_q4": (_i, [_p, _p, _p, _i, _i]),
    "jh_maccumulate_f32": (_i, [_p, _p, _i, _i]),
    "jh_scale_f32": (_i, [_f, _p, _i, _i]),
    "jh_saxpy_f32": (_i, [_f, _p, _p, _i, _i, _i]),
    "jh_saxpy_batch_f32": (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _i]),
    "jh_quantize_q8": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _i]),
    "jh_quantize_bf16": (_i, [_p, _l, _p]),
    "jh_rmsnorm_f32": (_i, [_p, _p, _f, _i, _f, _p]),
    "jh_softmax_f32": (_i, [_p, _i, _i]),
    "jh_silu_mul_f32": (_i, [_p, _p, _i]),
    "jh_layernorm_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "jh_gelu_f32": (_i, [_p, _i]),
    "jh_rope_table": (_i, [_i, _i, _d, _d, _p]),
    "jh_rope_apply_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i]),
    "jh_kv_page_geometry": (_i, [_l, _i, _i, _i, _i, _p]),
    "jh_model_create": (_i, [_p, _p]),
    "jh_model_destroy": (_i, [_p]),
    "jh_model_set_weight": (_i, [_p, _i, _i, _i, _p, _p, _i, _i, _i]),
    "jh_model_weight_bytes": (_l, [_p]),
    "jh_model_tiled_bytes": (_l, [_p]),
    "jh_model_released_bytes": (_l, [_p]),
    "jh_session_create": (_i, [_p, _i, _l, _p]),
    "jh_session_destroy": (_i, [_p]),
    "jh_session_page_info": (_i, [_p, _p]),
    "jh_tp_group_create": (_i, [_p, _i, _p]),
    "jh_tp_group_destroy": (_i, [_p]),
    "jh_tp_group_forward": (_i, [_p, _p, _i, _i]),
    "jh_tp_group_sample": (_i, [_p, _p]),
    "jh_tp_group_decode_n": (_i, [_p, _i, _i, _i, _p]),
    "jh_tp_rank_create": (_i, [_p, _i, _i, _p]),
    "jh_tp_rank_handles": (_i, [_p, _p]),
    "jh_tp_rank_connect": (_i, [_p, _p]),
    "jh_tp_rank_decode_n": (_i, [_p, _i, _i, _i, _p]),
    "jh_pipeline_create": (_i, [_p, _i, _p]),
    "jh_pipeline_destroy": (_i, [_p]),
    "jh_pipeline_peer_access": (_i, [_p, _p, _i]),
    "jh_pipeline_prefill": (_i, [_p, _p, _i, _i, _p]),
    "jh_pipeline_decode_n_async": (_i, [_p, _i, _i, _i]),
    "jh_pipeline_decode_wait": (_i, [_p, _p, _i]),
    "jh_stage_decode_async": (_i, [_p, _p, _p, _i, _p, _p]),
    "jh_forward": (_i, [_p, _p, _p, _i, _i, _p]),
    "jh_forward_device": (_i, [_p, _p, _p, _i, _i, _p]),
    "jh_sample": (_i, [_p, _f, _f, _p, _p]),
    "jh_decode_step": (_i, [_p, _i, _i, _p]),
    "jh_decode_n": (_i, [_p, _i, _i, _i, _p]),
    "jh_decode_n_async": (_i, [_p, _i, _i, _i]),
    "jh_decode_n_sampled": (_i, [_p, _i, _i, _i, _f, _p, _p]),
    "jh_decode_wait": (_i, [_p, _p, _i]),
    "jh_session_set_eos": (_i, [_p, _p, _i]),
    "jh_decode_generated": (_i, [_p, _p]),
    "jh_session_set_strict": (_i, [_p, _i]),
    "jh_source_hash": (C.c_char_p, []),
    "jh_abi_config_layout": (_i, [_p, _i]),
    "jh_get_logits": (_i, [_p, _p]),
    "jh_model_set_kv_head_offset": (_i, [_p, _i]),
    "jh_tp_set_row": (_i, [_p, _i, _p, _i]),
    "jh_tp_attn": (_i, [_p, _i, _p]),
    "jh_tp_ffn": (_i, [_p, _i, _p, _p]),
    "jh_tp_finish_layer": (_i, [_p, _p]),
    "jh_tp_rows_max": (_i, [_p]),
    "jh_tp_set_rows": (_i, [_p, _p, _p, _i, _i]),
    "jh_tp_attn_rows": (_i, [_p, _i, _p]),
    "jh_tp_ffn_rows": (_i, [_p, _i, _p, _p]),
    "jh_tp_finish_layer_rows": (_i, [_p, _p]),
    "jh_tp_finish_rows": (_i, [_p, _p]),
    "jh_session_get_row": (_i, [_p, _p, _i]),
    "jh_set_tap_layer": (_i, [_p, _i]),
    "jh_get_tap": (_i, [_p, _i, _p, _i]),
    "jh_session_stream": (_p, [_p]),
    "jh_decode_stats": (_i, [_p, _p, _p]),
    "jh_session_synchronize": (_i, [_p]),
    "jh_gemm_bench": (_i, [_i, _i, _i, _i, _i, _i, _p]),
    "jh_debug_attn_timeline": (_i, [_p, _i, _p, _i]),
    "jh_debug_gemv_timeline": (_i, [_p, _i, _p, _i]),
    "jh_kernel_bench": (_i, [_p, _i, _i, _p, _p]),
}
EXPORTS = sorted(_PROTOS)


def lib():
    """The loaded library.  Raises (never falls back) if it is missing or cannot be loaded."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        # the struct this binding passes by pointer must be the struct the binary was compiled with
        lay = (C.c_int32 * 32)()
        cnt = L.jh_abi_config_layout(lay, 32)
        mine = [C.sizeof(Config)] + [getattr(Config, f).offset for f, _ in Config._fields_]
        if list(lay[:cnt]) != mine:
            raise RuntimeError(f"jh_config layout mismatch: library {list(lay[:cnt])}, binding {mine}")
        _lib = L
    return _lib


_host_lib = None
_HOST_PROTOS = {
    "jhost_last_error": (C.c_char_p, []),
    "jhost_create": (_i, [_p, _i, _i, _l, _p]),
    "jhost_destroy": (None, [_p]),
    "jhost_set_weight": (_i, [_p, _i, _i, _i, _p, _p, _i, _i]),
    "jhost_forward": (_i, [_p, _p, _p, _i, _i]),
    "jhost_sample": (_i, [_p, _p, _p, _p]),
    "jhost_generate": (_i, [_p, _p, _i, _i, _p, _p, _p]),
    "jhost_page_info": (_i, [_p, _p]),
}
HOST_EXPORTS = sorted(_HOST_PROTOS)


def host_lib():
    """libjlamahost.so (csrc/host_mirror.cpp): the reference's host above the C ABI.  Loads libjlamahip.so first (its only dependency)."""
    global _host_lib
    if _host_lib is None:
        lib()
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(HOST_LIB_PATH)
        for name, (res, args) in _HOST_PROTOS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _host_lib = L
    return _host_lib


def check(rc):
    if rc is not None and rc < 0:
        msg = lib().jh_last_error().decode(errors="replace")
        if rc == JH_ERR_UNSUPPORTED:
            raise UnsupportedOperation(rc, msg)
        raise JhError(rc, msg)
    return rc


def ptr(a):
    """void* of a numpy array (or None)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_option(name, value):
    """Explicit process option (include/jlama_hip.h: jh_set_option); the library does not read tuning knobs from the environment."""
    if isinstance(value, str) and name == "JH_TILED_COPY":
        value = {"resident": 1, "transient": 2}.get(value, 0)
    check(lib().jh_set_option(name.encode(), int(value)))


def clear_options():
    check(lib().jh_clear_options())


def options_from_env(prefix="JH_"):
    """tools/ only: forward every JH_* environment variable with an integer value as an explicit option (the sweeps drive the
    planners this way).  The product never does this."""
    for k, v in os.environ.items():
        if k.startswith(prefix):
            try:
                val = int(v)
            except ValueError:
                if k != "JH_TILED_COPY":
                    continue
                val = {"resident": 1, "transient": 2}.get(v, 0)
            if lib().jh_set_option(k.encode(), val) != 0:
                print(f"[jlama_amd] {k}: the library has no such option (ignored)", file=sys.stderr)


def init(device=0):
    info = (C.c_int64 * 4)()
    check(lib().jh_init(device, info))
    return {"free_bytes": info[0], "cu_count": info[1], "device_count": info[2], "lds_bytes": info[3]}
