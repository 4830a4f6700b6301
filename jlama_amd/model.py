"""Host mirror of AbstractModel on the device-resident (Tier-2) C ABI.

Mirrors, at the token-id level, jlama-core/.../model/AbstractModel.java: ``forward`` (:267-279),
``batchForward`` (:295-312), ``sample`` (:443-491) and ``generate`` (:515-646).  Tokenisation is out of scope
(prompts are token ids).  All arithmetic runs in libjlamahip.so on the GPU; this file only marshals.
"""
import ctypes as C
import time

import numpy as np

from . import _native as N

MAX_BATCH_SIZE = 256  # jlama.max_batch_size (AbstractModel.java:57)


class HipLlamaModel:
    def __init__(self, cfg: dict, weights: dict, layer_range=None, device=0, kv_head_offset=0):
        """weights: {(layer|-1, slot): {dtype, data, scales, shape}} with numpy arrays (host) or
        objects exposing ``data_ptr()`` (device tensors); layer_range = this shard's [start, end).
        kv_head_offset: tensor-parallel shard only (cfg then carries the LOCAL head counts / hidden length, see
        distributed.tp_shard_config) -- global index of its first kv head."""
        N.init(device)
        L = cfg["n_layers"]
        ls, le = layer_range if layer_range else (0, L)
        self.cfg = dict(cfg)
        self.layer_range = (ls, le)
        self.c = N.Config(cfg["embedding_length"], cfg["hidden_length"], cfg["n_heads"], cfg["n_kv_heads"],
                          cfg["head_size"], L, cfg["vocab_size"], cfg["context_length"], cfg["weight_dtype"], ls, le,
                          cfg["rms_eps"], cfg["rope_theta"], cfg.get("rope_scaling", 1.0))
        self.h = C.c_void_p()
        N.check(N.lib().jh_model_create(C.byref(self.c), C.byref(self.h)))
        if kv_head_offset:
            N.check(N.lib().jh_model_set_kv_head_offset(self.h, int(kv_head_offset)))
        for (layer, slot), w in weights.items():
            if layer >= 0 and not (ls <= layer < le):
                continue
            self.set_weight(layer, slot, w)

    @staticmethod
    def _addr(a):
        if a is None:
            return None, False
        if hasattr(a, "data_ptr"):  # torch tensor
            return C.c_void_p(a.data_ptr()), bool(getattr(a, "is_cuda", False))
        return N.ptr(np.ascontiguousarray(a)), False

    def set_weight(self, layer, slot, w):
        rows, cols = w["shape"]
        d, dev = self._addr(w["data"])
        s, _ = self._addr(w.get("scales"))
        N.check(N.lib().jh_model_set_weight(self.h, layer, slot, w["dtype"], d, s, rows, cols, 1 if dev else 0))

    def weight_bytes(self):
        return N.lib().jh_model_weight_bytes(self.h)

    def tiled_bytes(self):
        """bytes of the resident MFMA-ordered second copy the batched prefill made (0 with JH_TILED_COPY=transient)"""
        return N.lib().jh_model_tiled_bytes(self.h)

    def released_bytes(self):
        """bytes of row-major projection weights released under JH_STRICT_ONLY=1 (reference-order sessions only from then on)"""
        return N.lib().jh_model_released_bytes(self.h)

    def session(self, max_ctx, max_page_bytes=0):
        return HipSession(self, max_ctx, max_page_bytes)

    def close(self):
        if self.h:
            N.lib().jh_model_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipSession:
    """One KV buffer + activation workspace (KvBufferCache.getKvBuffer, KvBufferCache.java:58-60)."""

    def __init__(self, model, max_ctx, max_page_bytes=0):
        self.model = model
        self.max_ctx = max_ctx
        self.h = C.c_void_p()
        N.check(N.lib().jh_session_create(model.h, max_ctx, max_page_bytes, C.byref(self.h)))

    def page_info(self):
        out = (C.c_int32 * 4)()
        N.check(N.lib().jh_session_page_info(self.h, out))
        return tuple(out)

    # -- AbstractModel.batchForward / forward ---------------------------------------------------------
    def forward(self, tokens=None, start_pos=0, x=None, want_output=True):
        E = self.model.cfg["embedding_length"]
        if tokens is not None:
            tokens = np.ascontiguousarray(tokens, dtype=np.int32).reshape(-1)
            n = tokens.size
        else:
            x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, E)
            n = x.shape[0]
        out = np.empty((n, E), dtype=np.float32) if want_output else None
        N.check(N.lib().jh_forward(self.h, N.ptr(tokens) if tokens is not None else None,
                                   N.ptr(x) if tokens is None else None, n, start_pos, N.ptr(out)))
        return out

    def forward_device(self, tokens, x_in_ptr, n, start_pos, x_out_ptr):
        """Device-pointer variant for layer-sharded pipelines (activations handed to RCCL send/recv)."""
        tk = np.ascontiguousarray(tokens, dtype=np.int32) if tokens is not None else None
        N.check(N.lib().jh_forward_device(self.h, N.ptr(tk) if tk is not None else None,
                                          C.c_void_p(x_in_ptr) if x_in_ptr else None, n, start_pos,
                                          C.c_void_p(x_out_ptr) if x_out_ptr else None))

    def stage_decode_async(self, token_ptr, x_in_ptr, pos, x_out_ptr, token_out_ptr):
        """One decode row of this layer shard, queued on the session's stream (token id and rows in device memory)."""
        v = lambda q: C.c_void_p(q) if q else None
        N.check(N.lib().jh_stage_decode_async(self.h, v(token_ptr), v(x_in_ptr), pos, v(x_out_ptr), v(token_out_ptr)))

    def batch_forward(self, tokens, start_pos=0):
        """Chunks of jlama.max_batch_size rows (AbstractModel.java:304); returns the last chunk's output."""
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        last = None
        for i in range(0, tokens.size, MAX_BATCH_SIZE):
            last = self.forward(tokens[i:i + MAX_BATCH_SIZE], start_pos + i)
        return last

    # -- tensor-parallel halves (device pointers; asynchronous on the session's stream) ---------------------
    def tp_set_row(self, token, pos, x_ptr=None):
        N.check(N.lib().jh_tp_set_row(self.h, int(token) if token is not None else -1, C.c_void_p(x_ptr) if x_ptr else None, pos))

    def tp_attn(self, layer, partial_ptr):
        N.check(N.lib().jh_tp_attn(self.h, layer, C.c_void_p(partial_ptr)))

    def tp_ffn(self, layer, reduced_ptr, partial_ptr):
        N.check(N.lib().jh_tp_ffn(self.h, layer, C.c_void_p(reduced_ptr), C.c_void_p(partial_ptr)))

    def tp_finish_layer(self, reduced_ptr):
        N.check(N.lib().jh_tp_finish_layer(self.h, C.c_void_p(reduced_ptr)))

    # the same halves over a chunk of prompt rows ([rows, E] partial / reduced buffers)
    def tp_rows_max(self):
        return int(N.lib().jh_tp_rows_max(self.h))

    def tp_set_rows(self, tokens, pos):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        N.check(N.lib().jh_tp_set_rows(self.h, N.ptr(tokens), None, int(tokens.size), int(pos)))

    def tp_attn_rows(self, layer, partial_ptr):
        N.check(N.lib().jh_tp_attn_rows(self.h, layer, C.c_void_p(partial_ptr)))

    def tp_ffn_rows(self, layer, reduced_ptr, partial_ptr):
        N.check(N.lib().jh_tp_ffn_rows(self.h, layer, C.c_void_p(reduced_ptr), C.c_void_p(partial_ptr)))

    def tp_finish_layer_rows(self, reduced_ptr):
        N.check(N.lib().jh_tp_finish_layer_rows(self.h, C.c_void_p(reduced_ptr)))

    def tp_finish_rows(self, rows_out_ptr=None):
        N.check(N.lib().jh_tp_finish_rows(self.h, C.c_void_p(rows_out_ptr) if rows_out_ptr else None))

    def current_row(self):
        out = np.empty(self.model.cfg["embedding_length"], dtype=np.float32)
        N.check(N.lib().jh_session_get_row(self.h, N.ptr(out), 0))
        return out

    # -- AbstractModel.sample ---------------------------------------------------------------------------
    def sample(self, temperature=0.0, u=0.5, want_logits=False):
        tok = C.c_int32()
        logits = np.empty(self.model.cfg["vocab_size"], dtype=np.float32) if want_logits else None
        N.check(N.lib().jh_sample(self.h, temperature, u, C.byref(tok), N.ptr(logits)))
        return (tok.value, logits) if want_logits else tok.value

    def logits(self):
        out = np.empty(self.model.cfg["vocab_size"], dtype=np.float32)
        N.check(N.lib().jh_get_logits(self.h, N.ptr(out)))
        return out

    def decode_step(self, token, pos):
        tok = C.c_int32()
        N.check(N.lib().jh_decode_step(self.h, int(token), int(pos), C.byref(tok)))
        return tok.value

    def decode_n(self, first_token, start_pos, n):
        """n greedy steps chained on the device; fewer ids come back when a stop token (set_eos) ended the loop."""
        out = np.empty(n, dtype=np.int32)
        N.check(N.lib().jh_decode_n(self.h, int(first_token), int(start_pos), int(n), N.ptr(out)))
        return out[:self.decode_generated()]

    def decode_n_sampled(self, first_token, start_pos, n, temperature, uniforms):
        """n steps of the device loop with temperature sampling: uniforms[i] decides the i-th sampled token (AbstractModel.java:475-489)."""
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        assert u.size >= n
        out = np.empty(n, dtype=np.int32)
        N.check(N.lib().jh_decode_n_sampled(self.h, int(first_token), int(start_pos), int(n), float(temperature), N.ptr(u), N.ptr(out)))
        return out[:self.decode_generated()]

    def decode_generated(self):
        k = C.c_int32()
        N.check(N.lib().jh_decode_generated(self.h, C.byref(k)))
        return k.value

    def set_eos(self, eos_tokens):
        """Config.eosTokens for the device loop (AbstractModel.java:600-603)."""
        ids = np.ascontiguousarray(list(eos_tokens or ()), dtype=np.int32)
        N.check(N.lib().jh_session_set_eos(self.h, N.ptr(ids) if ids.size else None, int(ids.size)))

    def set_strict(self, on=True):
        """Panama-order verification kernels (include/jlama_hip.h: jh_session_set_strict)."""
        N.check(N.lib().jh_session_set_strict(self.h, 1 if on else 0))

    def decode_n_async(self, first_token, start_pos, n):
        N.check(N.lib().jh_decode_n_async(self.h, int(first_token), int(start_pos), int(n)))

    def decode_wait(self, n):
        out = np.empty(n, dtype=np.int32)
        N.check(N.lib().jh_decode_wait(self.h, N.ptr(out), n))
        return out[:self.decode_generated()]

    def decode_stats(self):
        ms, k = C.c_double(), C.c_int32()
        N.check(N.lib().jh_decode_stats(self.h, C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def synchronize(self):
        N.check(N.lib().jh_session_synchronize(self.h))

    def kernel_bench(self, which, iters=3):
        """(ms per launch, algorithmic bytes per launch) of decode kernel `which` (0 qkv,1 attn,2 o,3 gate/up,4 down)."""
        ms, b = C.c_double(), C.c_int64()
        N.check(N.lib().jh_kernel_bench(self.h, which, iters, C.byref(ms), C.byref(b)))
        return ms.value, b.value

    def stream(self):
        return N.lib().jh_session_stream(self.h)

    # -- taps --------------------------------------------------------------------------------------------
    def set_tap_layer(self, layer):
        N.check(N.lib().jh_set_tap_layer(self.h, layer))

    def tap(self, name, n):
        out = np.empty(n, dtype=np.float32)
        got = N.check(N.lib().jh_get_tap(self.h, N.TAP[name], N.ptr(out), n))
        assert got == n, (name, got, n)
        return out

    # -- AbstractModel.generate at the token-id level ---------------------------------------------------
    def generate(self, prompt_tokens, ntokens, temperature=0.0, rng=None, eos_tokens=(), on_device_loop=True):
        """prompt_tokens already contain BOS.  Returns dict(tokens, prompt_ms, generate_ms, tokens_generated);
        decode clock starts after the first sampled token (AbstractModel.java:589), like the reference.  Stop tokens are
        tested on the ids sampled INSIDE the loop only (:590-603): the token sampled from the prompt is always fed on."""
        prompt_tokens = np.ascontiguousarray(prompt_tokens, dtype=np.int32)
        eos_tokens = tuple(int(t) for t in (eos_tokens or ()))
        t0 = time.perf_counter()
        self.batch_forward(prompt_tokens, 0)
        u = float(rng.random()) if (rng is not None and temperature > 0) else 0.5
        nxt = self.sample(temperature, u)
        t1 = time.perf_counter()
        out = [nxt]
        start = prompt_tokens.size
        n_more = ntokens - start
        if on_device_loop and n_more > 0:
            self.set_eos(eos_tokens)             # the device loop honours the stop tokens itself (finish_token_kernel)
            if temperature == 0.0:
                out.extend(int(t) for t in self.decode_n(nxt, start, n_more))
            else:                                # one uniform per sampled token, drawn in the order the host loop would draw them
                us = [float(rng.random()) if rng is not None else 0.5 for _ in range(n_more)]
                out.extend(int(t) for t in self.decode_n_sampled(nxt, start, n_more, temperature, us))
        else:
            for i in range(start, ntokens):
                if temperature == 0.0:
                    nxt = self.decode_step(nxt, i)
                else:
                    self.forward([nxt], i, want_output=False)
                    nxt = self.sample(temperature, float(rng.random()) if rng is not None else 0.5)
                out.append(nxt)
                if nxt in eos_tokens:
                    break
        t2 = time.perf_counter()
        return {"tokens": np.array(out, dtype=np.int32), "prompt_ms": (t1 - t0) * 1e3, "generate_ms": (t2 - t1) * 1e3,
                "tokens_generated": len(out) - 1}

    def close(self):
        if self.h:
            N.lib().jh_session_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_stage_models(cfg, weights_for_stage, devices):
    """Shard models of a one-process pipeline: stage k holds layers [k*L/N, (k+1)*L/N) (DistributedContext.java:75-77) on
    HIP device devices[k].  weights_for_stage(k, (layer_start, layer_end), device) -> weight dict of stage k (the first
    stage's with the embedding table, the last one's with final norm / LM head)."""
    n = len(devices)
    L = cfg["n_layers"]
    if L % n:
        raise ValueError(f"{L} layers do not split evenly over {n} stages")
    per = L // n
    return [HipLlamaModel(cfg, weights_for_stage(k, (k * per, (k + 1) * per), dev), layer_range=(k * per, (k + 1) * per), device=dev)
            for k, dev in enumerate(devices)]


class HipPipeline:
    """One-process layer-sharded pipeline over the GPUs of one node (include/jlama_hip.h: jh_pipeline_*): one session per
    stage model; activations hop device to device with stream-ordered peer copies, no host round trip per token.  Several
    pipelines over the same stage models = several sessions in flight."""

    def __init__(self, models, max_ctx):
        self.models = list(models)
        self.sessions = [m.session(max_ctx) for m in self.models]
        n = len(self.sessions)
        arr = (C.c_void_p * n)(*[s.h for s in self.sessions])
        self.h = C.c_void_p()
        N.check(N.lib().jh_pipeline_create(arr, n, C.byref(self.h)))

    def prefill(self, tokens, start_pos=0):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        tok = C.c_int32()
        N.check(N.lib().jh_pipeline_prefill(self.h, N.ptr(tokens), tokens.size, start_pos, C.byref(tok)))
        return tok.value

    def peer_access(self):
        """Per stage: how the hop into it travels -- True direct peer access (xGMI), False staged copies, None same device;
        entry 0 is the sampled id's way back to the first stage."""
        n = len(self.sessions)
        out = (C.c_int32 * n)()
        N.check(N.lib().jh_pipeline_peer_access(self.h, out, n))
        return [None if v < 0 else bool(v) for v in out]

    def decode_n_async(self, first_token, start_pos, n):
        N.check(N.lib().jh_pipeline_decode_n_async(self.h, int(first_token), int(start_pos), int(n)))

    def decode_wait(self, n):
        out = np.empty(n, dtype=np.int32)
        N.check(N.lib().jh_pipeline_decode_wait(self.h, N.ptr(out), n))
        return out[:self.sessions[-1].decode_generated()]

    def decode_n(self, first_token, start_pos, n):
        self.decode_n_async(first_token, start_pos, n)
        return self.decode_wait(n)

    def close(self):
        if self.h:
            N.lib().jh_pipeline_destroy(self.h)
            self.h = C.c_void_p()
        for s in self.sessions:
            s.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipTPGroup:
    """One-process tensor-parallel group (include/jlama_hip.h: jh_tp_group_*): head-split shard models
    (distributed.tp_shard_config / tp_shard_weights), one session each, reductions as one-shot peer writes + local sums in
    shard order -- no host synchronisation inside a layer."""

    def __init__(self, shard_models, max_ctx):
        self.models = list(shard_models)
        self.sessions = [m.session(max_ctx) for m in self.models]
        n = len(self.sessions)
        arr = (C.c_void_p * n)(*[s.h for s in self.sessions])
        self.h = C.c_void_p()
        N.check(N.lib().jh_tp_group_create(arr, n, C.byref(self.h)))

    def forward(self, tokens, start_pos=0):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        N.check(N.lib().jh_tp_group_forward(self.h, N.ptr(tokens), tokens.size, start_pos))

    def sample(self):
        tok = C.c_int32()
        N.check(N.lib().jh_tp_group_sample(self.h, C.byref(tok)))
        return tok.value

    def decode_n(self, first_token, start_pos, n):
        out = np.empty(n, dtype=np.int32)
        N.check(N.lib().jh_tp_group_decode_n(self.h, int(first_token), int(start_pos), int(n), N.ptr(out)))
        return out[:self.sessions[0].decode_generated()]   # fewer than n only after a stop token (set_eos on shard 0's session)

    def status(self):
        """What the last decode_n ran on and whether any meeting timed out (include/jlama_hip.h: jh_tp_group_status)."""
        out = (C.c_int32 * 6)()
        N.check(N.lib().jh_tp_group_status(self.h, out, 6))
        return {"mode": {0: "none", 1: "graph replay per shard and token, in-kernel meetings", 2: "event-ordered host loop"}[out[0]],
                "timeouts": out[1], "graph_path_in_use": bool(out[2]), "gemv_push": out[3], "flags_per_launch": out[4]}

    def close(self):
        if self.h:
            N.lib().jh_tp_group_destroy(self.h)
            self.h = C.c_void_p()
        for s in self.sessions:
            s.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipTPRank:
    """One rank of the tensor-parallel group when every shard lives in its own process (include/jlama_hip.h: jh_tp_rank_*): the
    other ranks' slot / flag / mailbox buffers are mapped through hipIpc handles exchanged by the host side
    (distributed.tp_generate_ipc does it with an all_gather); decode_n replays one captured graph per token -- the one-process
    group's graph -- with no collective library on the data path."""

    HANDLE_BYTES = 192

    def __init__(self, session, rank, n_ranks):
        self.session, self.rank, self.n = session, rank, n_ranks
        self.h = C.c_void_p()
        N.check(N.lib().jh_tp_rank_create(session.h, rank, n_ranks, C.byref(self.h)))

    def handles(self):
        buf = (C.c_ubyte * self.HANDLE_BYTES)()
        N.check(N.lib().jh_tp_rank_handles(self.h, buf))
        return bytes(buf)

    def connect(self, all_handles):
        all_handles = bytes(all_handles)
        assert len(all_handles) == self.n * self.HANDLE_BYTES
        buf = (C.c_ubyte * len(all_handles)).from_buffer_copy(all_handles)
        N.check(N.lib().jh_tp_rank_connect(self.h, buf))

    def signature(self):
        """The word every rank must agree on before decode_n (kernel family, CU count, push mode, shard shape)."""
        v = C.c_int64()
        N.check(N.lib().jh_tp_rank_signature(self.h, C.byref(v)))
        return v.value

    def status(self):
        out = (C.c_int32 * 6)()
        N.check(N.lib().jh_tp_group_status(self.h, out, 6))
        return {"timeouts": out[1], "gemv_push": out[3], "flags_per_launch": out[4], "connected": bool(out[5])}

    def decode_n(self, first_token, start_pos, n):
        out = np.empty(n, dtype=np.int32)
        N.check(N.lib().jh_tp_rank_decode_n(self.h, int(first_token), int(start_pos), int(n), N.ptr(out)))
        return out[:self.session.decode_generated()] if self.rank == 0 else None

    def close(self):
        if self.h:
            N.lib().jh_tp_group_destroy(self.h)
            self.h = C.c_void_p()
