"""KvBufferCache page geometry (jlama-core/.../tensor/KvBufferCache.java:58-60,99-112,224-280)."""
import ctypes as C

from . import _native as N

DEFAULT_PAGE_BYTES = 1 << 23  # 8 MiB per page (KvBufferCache.java:59)


def page_geometry(n_layers, context_length, kv_length, max_page_bytes=DEFAULT_PAGE_BYTES, dtype_size=4):
    """(layersPerPage, ctxPerPage) maximising layers*ctx per page; page tensor is
    [layersPerPage, 2(K,V), ctxPerPage, kvLength] in the working dtype (F32)."""
    out = (C.c_int32 * 2)()
    N.check(N.lib().jh_kv_page_geometry(max_page_bytes, n_layers, context_length, kv_length, dtype_size, out))
    return out[0], out[1]
