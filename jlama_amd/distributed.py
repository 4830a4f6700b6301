"""Layer-sharded inference across the GPUs of one node -- Jlama's cluster layer split re-hosted on RCCL.

Reference behaviour: ``DistributedContext.layerStart/layerEnd``
(jlama-core/.../model/DistributedContext.java:75-77) gives every worker a contiguous layer range;
``AbstractModel.forward`` runs only that range (AbstractModel.java:321-326); a worker hands its ``[batch, E]`` F32
output to the next layer shard (``PassRecord.tensor``, jlama-net/.../Worker.java:193-196,226-248), the last shard's
final row goes to the coordinator which samples (Coordinator.java:184) and feeds the token back to shard 0.

Here: one process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests), rank r owns
layers ``[r*L/N, (r+1)*L/N)`` and their KV pages; the hop between shards is a point-to-point send/recv of the F32
activation (16 KiB per decode token at E=4096), the sampled token id (4 B) returns from the last rank to rank 0.
No collective is needed on this path.  Because a single session is strictly sequential through the shards, N sessions
are kept in flight (session j is on stage (t-j) mod N at tick t) so every GPU streams its weights on every tick.

torch is plumbing only (process group + device buffers handed to RCCL); all arithmetic is in libjlamahip.so.
"""
import os
import time

import numpy as np

from ._native import UnsupportedOperation


def layer_range(rank, world, n_layers):
    """DistributedContext layer split: contiguous, equal ranges (requires n_layers % world == 0)."""
    if n_layers % world:
        raise ValueError(f"{n_layers} layers do not split evenly over {world} shards")
    per = n_layers // world
    return rank * per, (rank + 1) * per


class ShardEngine:
    """What a pipeline stage must provide.  x buffers are torch tensors on the engine's device."""

    def forward_tokens(self, session, tokens, start_pos, x_out):  # first shard: embedding rows -> x_out [n,E]
        raise NotImplementedError

    def forward_x(self, session, x_in, n, start_pos, x_out):      # later shards
        raise NotImplementedError

    def sample(self, session):                                    # last shard: argmax of the last forwarded row
        raise NotImplementedError


class HipShardEngine(ShardEngine):
    """A layer shard resident on one MI355X (Tier-2 C ABI); activations never leave HBM between RCCL and the kernels."""

    def __init__(self, cfg, weights, rank, world, device_index, n_sessions, max_ctx, strict=False):
        import torch
        from .model import HipLlamaModel
        self.torch = torch
        ls, le = layer_range(rank, world, cfg["n_layers"])
        self.model = HipLlamaModel(cfg, weights, layer_range=(ls, le), device=device_index)
        self.sessions = [self.model.session(max_ctx) for _ in range(n_sessions)]
        if strict:   # reference order (the path whose ids are bit-exact: what `value` is at N = 1)
            for s in self.sessions:
                s.set_strict(True)

    def _sync_torch(self):
        self.torch.cuda.current_stream().synchronize()

    def forward_tokens(self, session, tokens, start_pos, x_out):
        s = self.sessions[session]
        s.forward_device(np.ascontiguousarray(tokens, dtype=np.int32), 0, len(tokens), start_pos, x_out.data_ptr())
        s.synchronize()

    def forward_x(self, session, x_in, n, start_pos, x_out):
        s = self.sessions[session]
        self._sync_torch()  # the recv into x_in ran on torch's stream
        s.forward_device(None, x_in.data_ptr(), n, start_pos, x_out.data_ptr())
        s.synchronize()

    def sample(self, session):
        return self.sessions[session].sample(0.0, 0.5)

    # -- stream-ordered stage (pipeline_decode_streamed): nothing below touches the host
    def stream(self, session):
        """torch view of the session's own HIP stream: RCCL send/recv issued under it are ordered with the kernels."""
        if not hasattr(self, "_ext"):
            self._ext = {}
        if session not in self._ext:
            self._ext[session] = self.torch.cuda.ExternalStream(self.sessions[session].stream())
        return self._ext[session]

    def stage_step(self, session, token, x_in, pos, x_out, token_out):
        p = lambda t: t.data_ptr() if t is not None else 0
        self.sessions[session].stage_decode_async(p(token), p(x_in), pos, p(x_out), p(token_out))


def pipeline_prefill(dist, engine, rank, world, session, prompt, E, device, dtype):
    """Prefill one session through all shards (batchForward, AbstractModel.java:295-312); returns the first sampled
    token on every rank (broadcast from the last shard)."""
    import torch
    n = len(prompt)
    x_in = torch.empty((n, E), dtype=dtype, device=device)
    x_out = torch.empty((n, E), dtype=dtype, device=device)
    if rank == 0:
        engine.forward_tokens(session, prompt, 0, x_out)
    else:
        dist.recv(x_in, src=rank - 1)
        engine.forward_x(session, x_in, n, 0, x_out)
    tok = torch.zeros(1, dtype=torch.int32, device=device)
    if rank < world - 1:
        dist.send(x_out, dst=rank + 1)
    else:
        tok[0] = engine.sample(session)
    dist.broadcast(tok, src=world - 1)
    return int(tok.item())


def _grp(tok_group):
    """kwargs of a token-path send/recv: the id travels on its OWN communicator when one is given (bench_pipeline makes one).
    Rows (rank r -> r+1) and ids (last rank -> rank 0) cross between the same pair of ranks when N = 2; on one communicator
    their sends and receives interleave in one issue order, and a row send would sit in front of the receive the peer's id send
    is waiting for -- legal only while the transport buffers small sends eagerly.  Two communicators never order one direction
    behind the other."""
    return {"group": tok_group} if tok_group is not None else {}


def pipeline_decode(dist, engine, rank, world, first_tokens, start_pos, steps_per_session, E, device, dtype, n_sessions=None, tok_group=None):
    """Greedy-decode ``steps_per_session`` tokens for each of ``n_sessions`` (default: ``world``) sessions with the sessions
    staggered across the pipeline.  Work item q = (session q % S, step q // S); every rank handles the items in that
    order, so with S = world every GPU is busy on every tick and with S = 1 the run is the single-stream (batch-1) case.
    Returns the tokens the LAST rank sampled, as an int array [sessions, steps] (zeros on other ranks)."""
    import torch
    N = world
    S = n_sessions if n_sessions else world
    n_items = S * steps_per_session
    out = np.zeros((S, steps_per_session), dtype=np.int32)
    x_in = torch.empty((1, E), dtype=dtype, device=device)
    x_out = [torch.empty((1, E), dtype=dtype, device=device) for _ in range(2)]
    tok_in = torch.zeros(1, dtype=torch.int32, device=device)
    tok_sends = []
    pending = None
    for tick in range(n_items + N - 1):
        q = tick - rank
        if q < 0 or q >= n_items:
            continue
        j, k = q % S, q // S
        buf = tick & 1
        if rank == 0:
            if k == 0:
                token = int(first_tokens[j])
            elif N == 1:
                token = int(out[j, k - 1])
            else:
                dist.recv(tok_in, src=N - 1, **_grp(tok_group))      # sampled by the last shard one tick ago
                token = int(tok_in.item())
            engine.forward_tokens(j, [token], start_pos + k, x_out[buf])
        else:
            dist.recv(x_in, src=rank - 1)
            engine.forward_x(j, x_in, 1, start_pos + k, x_out[buf])
        if rank < N - 1:
            if pending is not None:
                pending.wait()
            pending = dist.isend(x_out[buf], dst=rank + 1)
        else:
            t = engine.sample(j)
            out[j, k] = t
            if N > 1 and k + 1 < steps_per_session:
                # never wait for a token send inside the loop: rank 0 posts its receive S items later, and with more sessions
                # in flight than stages a blocking send here stalls the whole ring behind it
                tk = torch.full((1,), int(t), dtype=torch.int32, device=device)
                tok_sends.append((dist.isend(tk, dst=0, **_grp(tok_group)), tk))
    if pending is not None:
        pending.wait()
    for w_, _ in tok_sends:
        w_.wait()
    return out


def pipeline_decode_streamed(dist, engine, rank, world, first_tokens, start_pos, steps_per_session, E, device, dtype, n_sessions=None, tok_group=None):
    """pipeline_decode without a single host round trip inside the loop.  Same schedule (work item q = (session q % S, step
    q // S) in that order on every rank), but every hop is issued under the session's own HIP stream (``engine.stream``):
    ``dist.recv`` makes that stream wait for the RCCL transfer, ``engine.stage_step`` queues the shard's graph behind it,
    ``dist.isend`` ships the row (or, from the last rank, the sampled id -- a device word, read by rank 0's embedding kernel)
    when the graph is done.  The host only enqueues; sessions overlap on the GPU because each has its own stream.  Buffers are
    per session: a row buffer is rewritten only after the stream has waited for the send that read it.
    Returns the ids sampled by the LAST rank, [sessions, steps] (zeros elsewhere)."""
    import torch
    N = world
    S = n_sessions if n_sessions else world
    steps = steps_per_session
    first, last = rank == 0, rank == N - 1
    x_in = [torch.empty((1, E), dtype=dtype, device=device) for _ in range(S)] if not first else None
    x_out = [torch.empty((1, E), dtype=dtype, device=device) for _ in range(S)] if not last else None
    tok_in = torch.zeros((S, 1), dtype=torch.int32, device=device) if first else None
    out_dev = torch.zeros((S, steps), dtype=torch.int32, device=device) if last else None
    on_gpu = torch.device(device).type == "cuda"
    sync = (lambda: torch.cuda.synchronize(device)) if on_gpu else (lambda: None)   # CPU engines (tests over gloo) are synchronous
    if first:
        tok_in.copy_(torch.as_tensor(np.asarray(first_tokens[:S], dtype=np.int32).reshape(S, 1)))
    sync()
    sent = [None] * S          # the last isend that read x_out[j]
    keep = []                  # token sends in flight (each reads its own element of out_dev)
    for q in range(S * steps):
        j, k = q % S, q // S
        with torch.cuda.stream(engine.stream(j) if on_gpu else None):
            if first:
                if k > 0 and N > 1:
                    dist.recv(tok_in[j], src=N - 1, **_grp(tok_group))            # id sampled for step k-1 of this session
                elif k > 0:
                    tok_in[j].copy_(out_dev[j, k - 1:k])       # one rank: first and last stage are the same session
            else:
                dist.recv(x_in[j], src=rank - 1)
            if sent[j] is not None:
                sent[j].wait()                                 # stream-level: x_out[j] is free again
                sent[j] = None
            engine.stage_step(j, tok_in[j] if first else None, None if first else x_in[j], start_pos + k,
                              None if last else x_out[j], out_dev[j, k:k + 1] if last else None)
            if not last:
                sent[j] = dist.isend(x_out[j], dst=rank + 1)
            elif N > 1 and k + 1 < steps:
                keep.append(dist.isend(out_dev[j, k:k + 1], dst=0, **_grp(tok_group)))
    for w in keep + [w for w in sent if w is not None]:
        w.wait()
    sync()
    return out_dev.cpu().numpy() if last else np.zeros((S, steps), dtype=np.int32)


# ------------------------------------------------------------------------------------------------ tensor parallel (f2)
# Head split, SURVEY.md 8(e): DistributedContext.java:79-98 gives model shard r of N the attention heads
# [r*heads/N, (r+1)*heads/N), the kv heads [r*kvHeads/N, ...) and the hidden rows [r*H/N, ...); q/k/v/gate/up are split by
# OUTPUT rows (Weights.getLoadOffsets row windows), o/down by INPUT columns (attentionSegment / hiddenSegment of
# dotProductChunk, CausalSelfAttention.java:365-376, MLPBlock.java:147-158); the partial [B,E] results are summed over
# shards (tensorReducer :378 / :160) before the residual.  The reference caps N at the number of kv heads
# (JlamaService.java:65-68).  Here the sum is one RCCL all-reduce of 16 KiB (E=4096) per half layer; with 2 per layer it
# is latency-bound, so this path raises capacity (a 70B model over 8 GPUs without pipeline bubbles), not batch-1 speed.

def tp_shard_config(cfg, rank, size):
    """(local cfg, kv_head_offset) of model shard `rank` of `size`."""
    if cfg["n_kv_heads"] % size or cfg["n_heads"] % size or cfg["hidden_length"] % (32 * size):
        raise ValueError(f"cannot split {cfg['n_heads']} heads / {cfg['n_kv_heads']} kv heads / H={cfg['hidden_length']} over {size} shards")
    c = dict(cfg)
    c["n_heads"] = cfg["n_heads"] // size
    c["n_kv_heads"] = cfg["n_kv_heads"] // size
    c["hidden_length"] = cfg["hidden_length"] // size
    return c, rank * c["n_kv_heads"]


def _contig(a):
    """contiguous copy of a window: numpy (host weights) or torch (weights generated on a device)"""
    return a.contiguous() if hasattr(a, "contiguous") else np.ascontiguousarray(a)


def _rows(w, r0, n):
    out = dict(w)
    out["data"] = _contig(w["data"][r0:r0 + n])
    out["scales"] = None if w.get("scales") is None else _contig(w["scales"][r0:r0 + n])
    out["shape"] = (n, w["shape"][1])
    return out


def _cols(w, c0, n):
    """K-column window [c0, c0+n) of a weight; Q4 nibbles are [rows, cols/2] bytes in blocks of 32 columns = 16 bytes."""
    from . import _native as N
    out = dict(w)
    if w["dtype"] == N.DT_Q4:
        assert c0 % 32 == 0 and n % 32 == 0
        out["data"] = _contig(w["data"][:, c0 // 2:(c0 + n) // 2])
        out["scales"] = _contig(w["scales"][:, c0 // 32:(c0 + n) // 32])
    else:
        out["data"] = _contig(w["data"][:, c0:c0 + n])
    out["shape"] = (w["shape"][0], n)
    return out


def tp_shard_weights(cfg, weights, rank, size, device=None):
    """The windows of `weights` (full model; host arrays, or torch tensors resident on a GPU) that model shard `rank` holds.
    Norm weights, the embedding table and the LM head are replicated (every shard computes the same residual stream; rank 0
    samples).  device: torch device the windows are moved to (torch weights only: a shard on another GPU than the generator's)."""
    if device is not None:
        out = tp_shard_weights(cfg, weights, rank, size)
        moved = {}
        for k, w in out.items():
            w = dict(w)
            for f in ("data", "scales"):
                if w.get(f) is not None and hasattr(w[f], "to"):
                    w[f] = w[f].to(device)
            moved[k] = w
        return moved
    from . import synthetic as S
    hs = cfg["head_size"]
    A, KV, H = cfg["n_heads"] * hs // size, cfg["n_kv_heads"] * hs // size, cfg["hidden_length"] // size
    out = {}
    for (layer, slot), w in weights.items():
        if layer < 0 or slot in (S.W_NORM1, S.W_NORM2):
            out[(layer, slot)] = w
        elif slot == S.W_Q:
            out[(layer, slot)] = _rows(w, rank * A, A)
        elif slot in (S.W_K, S.W_V):
            out[(layer, slot)] = _rows(w, rank * KV, KV)
        elif slot in (S.W_GATE, S.W_UP):
            out[(layer, slot)] = _rows(w, rank * H, H)
        elif slot == S.W_O:
            out[(layer, slot)] = _cols(w, rank * A, A)
        elif slot == S.W_DOWN:
            out[(layer, slot)] = _cols(w, rank * H, H)
    return out


class TPEngine:
    """What a tensor-parallel shard provides; buffers are torch tensors on the engine's device ([E] float32)."""

    def set_row(self, token, pos):
        raise NotImplementedError

    def attn(self, layer, partial):                 # partial <- this shard's o-projection partial
        raise NotImplementedError

    def ffn(self, layer, reduced, partial):         # x1 = x + reduced; partial <- this shard's down-projection partial
        raise NotImplementedError

    def finish_layer(self, reduced):                # x = x1 + reduced
        raise NotImplementedError

    def sample(self):                               # rank 0: greedy token of the current row
        raise NotImplementedError

    # the same halves over a chunk of prompt rows ([rows, E] buffers); rows_max() == 0: this engine feeds rows one at a time
    def rows_max(self):
        return 0

    def set_rows(self, tokens, pos):
        raise NotImplementedError

    def attn_rows(self, layer, partial):
        raise NotImplementedError

    def ffn_rows(self, layer, reduced, partial):
        raise NotImplementedError

    def finish_layer_rows(self, reduced):
        raise NotImplementedError

    def finish_rows(self):                          # the chunk's last row becomes the current row
        raise NotImplementedError

    def stream_context(self):                       # the stream the halves AND the all-reduces are ordered on
        import contextlib
        return contextlib.nullcontext()


class HipTPEngine(TPEngine):
    """One head-split shard resident on one MI355X (Tier-2 C ABI); one session (batch 1)."""

    def __init__(self, cfg, weights, rank, size, device_index, max_ctx):
        import torch
        from .model import HipLlamaModel
        self.torch = torch
        lcfg, off = tp_shard_config(cfg, rank, size)
        self.model = HipLlamaModel(lcfg, tp_shard_weights(cfg, weights, rank, size), device=device_index, kv_head_offset=off)
        self.s = self.model.session(max_ctx)

    def stream_context(self):
        """Everything of a row -- the halves (jh_tp_attn / jh_tp_ffn / jh_tp_finish_layer) and the RCCL all-reduces between
        them -- is ordered on the session's own HIP stream: no host synchronisation inside a row."""
        if not hasattr(self, "_ext"):
            self._ext = self.torch.cuda.ExternalStream(self.s.stream())
        return self.torch.cuda.stream(self._ext)

    def set_row(self, token, pos):
        self.s.tp_set_row(token, pos)

    def attn(self, layer, partial):
        self.s.tp_attn(layer, partial.data_ptr())

    def ffn(self, layer, reduced, partial):
        self.s.tp_ffn(layer, reduced.data_ptr(), partial.data_ptr())

    def finish_layer(self, reduced):
        self.s.tp_finish_layer(reduced.data_ptr())

    def sample(self):
        return self.s.sample(0.0, 0.5)   # synchronises the session stream (the row, incl. its last all-reduce, is complete)

    def rows_max(self):
        return self.s.tp_rows_max()

    def set_rows(self, tokens, pos):
        self.s.tp_set_rows(tokens, pos)

    def attn_rows(self, layer, partial):
        self.s.tp_attn_rows(layer, partial.data_ptr())

    def ffn_rows(self, layer, reduced, partial):
        self.s.tp_ffn_rows(layer, reduced.data_ptr(), partial.data_ptr())

    def finish_layer_rows(self, reduced):
        self.s.tp_finish_layer_rows(reduced.data_ptr())

    def finish_rows(self):
        self.s.tp_finish_rows()


def tp_forward_row(dist, engine, token, pos, layers, buf):
    """One row through all layers on every shard: 2 all-reduces (sum) per layer.  buf: [E] float32 on the engine's device."""
    with engine.stream_context():
        engine.set_row(token, pos)
        for li in range(*layers):
            engine.attn(li, buf)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            engine.ffn(li, buf, buf)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            engine.finish_layer(buf)


TP_ROWS_MIN = 4   # shorter pieces of a prompt go row by row (as jh_forward does below prefill_batch_min)


def tp_forward_prompt(dist, engine, prompt, start_pos, layers, cfg, device, dtype, buf):
    """The prompt rows through all layers on every shard.  Engines with a batched path take chunks of up to rows_max() rows: the
    partial [rows, E] results are all-reduced ONCE per half-layer and chunk (what the reference's reducer sees in batchForward,
    CausalSelfAttention.java:378 / MLPBlock.java:160) instead of once per row; the rest goes row by row (tp_forward_row).  Every
    rank must take the same path: the chunk size is the minimum over the ranks."""
    import torch
    n, E = len(prompt), cfg["embedding_length"]
    cap = torch.tensor([int(engine.rows_max())], dtype=torch.int32, device=device)
    dist.all_reduce(cap, op=dist.ReduceOp.MIN)
    cap = int(cap.item())
    done = 0
    if cap >= TP_ROWS_MIN:
        rows_buf = torch.empty((min(cap, n), E), dtype=dtype, device=device)
        while n - done >= TP_ROWS_MIN:
            rows = min(cap, n - done)
            part = rows_buf[:rows]
            with engine.stream_context():
                # a shard can refuse a chunk it has no batched path for at THIS position (jh_tp_set_rows: JH_ERR_UNSUPPORTED, e.g. the
                # per-row attention kernel's score rows outgrow LDS); every rank must then take the row loop, so the verdict is MIN-reduced
                try:
                    engine.set_rows(np.asarray(prompt[done:done + rows], dtype=np.int32), start_pos + done)
                    fits = 1
                except UnsupportedOperation:
                    fits = 0
                ok = torch.tensor([fits], dtype=torch.int32, device=device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    break
                for li in range(*layers):
                    engine.attn_rows(li, part)
                    dist.all_reduce(part, op=dist.ReduceOp.SUM)
                    engine.ffn_rows(li, part, part)
                    dist.all_reduce(part, op=dist.ReduceOp.SUM)
                    engine.finish_layer_rows(part)
                engine.finish_rows()
            done += rows
    for i in range(done, n):
        tp_forward_row(dist, engine, int(prompt[i]), start_pos + i, layers, buf)
    return {"rows_per_chunk": cap if cap >= TP_ROWS_MIN else 1, "rows_batched": done}


def tp_generate(dist, engine, rank, prompt, n_gen, cfg, device, dtype):
    """Greedy generation with every rank holding a head-split shard: rows are fed one position at a time
    (batchForwardSlow order), rank 0 samples (Coordinator.java:184) and broadcasts the token id."""
    import torch
    buf = torch.empty(cfg["embedding_length"], dtype=dtype, device=device)
    tok = torch.zeros(1, dtype=torch.int32, device=device)
    layers = (0, cfg["n_layers"])
    out = []
    tp_forward_prompt(dist, engine, prompt, 0, layers, cfg, device, dtype, buf)
    pos = len(prompt)
    for _ in range(n_gen):
        with engine.stream_context():
            if rank == 0:
                tok[0] = engine.sample()
            dist.broadcast(tok, src=0)
            nxt = int(tok.item())
        out.append(nxt)
        tp_forward_row(dist, engine, nxt, pos, layers, buf)
        pos += 1
    return np.asarray(out, dtype=np.int32)


def tp_generate_ipc(dist, engine, rank, size, prompt, n_gen, cfg, device, dtype):
    """tp_generate with the decode steps on IPC-mapped peer memory instead of all-reduces: the prompt rows go through the
    all-reduce host above (one position at a time), rank 0 samples the first id, then every rank replays the tensor-parallel
    group's token graph (jh_tp_rank_*: partial rows pushed into every rank's slot over xGMI, shard-ordered sums, ids through
    device mailboxes).  The only collectives are the 192-byte handle exchange and the broadcast of the first id."""
    import torch
    from .model import HipTPRank
    buf = torch.empty(cfg["embedding_length"], dtype=dtype, device=device)
    layers = (0, cfg["n_layers"])
    tp_forward_prompt(dist, engine, prompt, 0, layers, cfg, device, dtype, buf)
    tok = torch.zeros(1, dtype=torch.int32, device=device)
    with engine.stream_context():
        if rank == 0:
            tok[0] = engine.sample()
        dist.broadcast(tok, src=0)
        first = int(tok.item())
    tpr = HipTPRank(engine.s, rank, size)
    xdev = device if dist.get_backend() == "nccl" else torch.device("cpu")    # (gloo gathers host tensors only)
    mine = torch.frombuffer(bytearray(tpr.handles()), dtype=torch.uint8).to(xdev)
    gathered = [torch.empty_like(mine) for _ in range(size)]
    dist.all_gather(gathered, mine)
    tpr.connect(b"".join(bytes(g.cpu().numpy().tobytes()) for g in gathered))
    # the meeting protocol assumes one launch plan on every rank (flag words per producer, push vs scatter): compare the signatures
    sig = torch.tensor([tpr.signature()], dtype=torch.int64, device=xdev)
    sigs = [torch.empty_like(sig) for _ in range(size)]
    dist.all_gather(sigs, sig)
    if len({int(x.item()) for x in sigs}) != 1:
        raise RuntimeError(f"tensor-parallel ranks disagree on the launch plan (kernel mode / CU count / push option): {[int(x.item()) for x in sigs]}")
    dist.barrier()                                   # every rank has mapped every buffer before anyone stores into one
    ids = tpr.decode_n(first, len(prompt), n_gen - 1) if n_gen > 1 else np.zeros(0, np.int32)
    dist.barrier()                                   # nobody unmaps while a peer's last graph may still be storing
    tpr.close()
    return np.concatenate([[first], ids]).astype(np.int32) if rank == 0 else None


def tp_rank_bench(dist, engine, rank, size, prompt, steps, cfg, device, dtype, use_ipc=True):
    """Single-stream decode rate of the head-split group with one shard per rank (DistributedContext.java:79-98): the prompt in
    chunks (tp_forward_prompt: one all-reduce per half-layer and chunk), then `steps` greedy tokens -- on the IPC-mapped token
    graphs (jh_tp_rank_*: partials pushed into every rank's slot over xGMI, nothing on the host inside a token) when `use_ipc`,
    else row by row with two all-reduces per layer (tp_forward_row; also what runs when the IPC mapping is refused).  Timed like
    every leg of the bench: barrier + synchronize on both sides, MAX over the ranks.  Returns a dict on every rank."""
    import torch
    buf = torch.empty(cfg["embedding_length"], dtype=dtype, device=device)
    layers = (0, cfg["n_layers"])
    on_gpu = device != "cpu" and getattr(device, "type", "cpu") != "cpu"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    xdev = device if dist.get_backend() == "nccl" else torch.device("cpu")    # (gloo gathers host tensors only)
    t0 = time.perf_counter()
    info = tp_forward_prompt(dist, engine, prompt, 0, layers, cfg, device, dtype, buf)
    tok = torch.zeros(1, dtype=torch.int32, device=device)
    with engine.stream_context():
        if rank == 0:
            tok[0] = engine.sample()
        dist.broadcast(tok, src=0)
        first = int(tok.item())
    prefill_ms = (time.perf_counter() - t0) * 1e3
    mode, err, tpr = "all-reduce rows (2 per layer and token)", None, None
    if use_ipc:
        # Every rank walks the SAME collective sequence whatever happens locally (a rank that raised between two all_gathers
        # left its peers inside one until the watchdog): each non-collective step (constructor, connect, signature) ends in a
        # MIN-reduced verdict, and the ranks fall back together at the first step any of them failed.
        def agree(ok_local):
            flag = torch.tensor([1 if ok_local else 0], dtype=torch.int32, device=xdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        def step(fn):
            nonlocal err
            try:
                return fn(), True
            except Exception as e:   # noqa: BLE001 -- the leg falls back, the line says why
                err = err or repr(e)[:300]
                return None, False

        def make():
            from .model import HipTPRank
            t = HipTPRank(engine.s, rank, size)
            return t, torch.frombuffer(bytearray(t.handles()), dtype=torch.uint8).to(xdev)

        made, ok = step(make)
        ok = agree(ok)
        if ok:
            tpr, mine = made
            gathered = [torch.empty_like(mine) for _ in range(size)]
            dist.all_gather(gathered, mine)
            _, ok = step(lambda: tpr.connect(b"".join(bytes(g.cpu().numpy().tobytes()) for g in gathered)))
            ok = agree(ok)
        if ok:
            sig_v, ok = step(lambda: int(tpr.signature()))
            ok = agree(ok)
        if ok:
            sig = torch.tensor([sig_v], dtype=torch.int64, device=xdev)
            sigs = [torch.empty_like(sig) for _ in range(size)]
            dist.all_gather(sigs, sig)
            if len({int(x.item()) for x in sigs}) != 1:     # every rank sees the same list: they fall back together
                err, ok = err or "ranks disagree on the launch plan", False
        if ok:
            mode = "token graphs on IPC-mapped peer memory (in-kernel meetings over xGMI)"
        else:
            if made is not None:
                try:
                    made[0].close()
                except Exception:   # noqa: BLE001
                    pass
            tpr = None
            err = err or "a peer rank could not set up its IPC shard"

    def run(n):
        if tpr is not None:
            ids = tpr.decode_n(first, len(prompt), n)
            return ids
        out, nxt, pos = [], first, len(prompt)
        for _ in range(n):
            tp_forward_row(dist, engine, nxt, pos, layers, buf)
            with engine.stream_context():
                if rank == 0:
                    tok[0] = engine.sample()
                dist.broadcast(tok, src=0)
                nxt = int(tok.item())
            out.append(nxt)
            pos += 1
        return np.asarray(out, dtype=np.int32)

    run(min(4, steps))                       # graph capture / warm-up, untimed (rewrites the same KV rows)
    dist.barrier()
    sync()
    t0 = time.perf_counter()
    ids = run(steps)
    sync()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=xdev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    st = tpr.status() if tpr is not None else {}
    if tpr is not None:
        dist.barrier()                       # nobody unmaps while a peer's last graph may still be storing
        tpr.close()
    return {"shards": size, "single_stream_tokens_per_s": round(steps / dt, 2), "ms_per_token": round(dt / steps * 1e3, 4), "steps": int(steps),
            "seconds": dt, "decode": mode, "ipc_error": err, "prompt_rows": int(len(prompt)), "prompt_rows_batched": info["rows_batched"],
            "prompt_rows_per_chunk": info["rows_per_chunk"], "prefill_ms": round(prefill_ms, 2),
            "first_ids": [int(first)] + [int(t) for t in (ids[:7] if ids is not None else [])],
            "meeting_timeouts": st.get("timeouts"), "gemv_push": st.get("gemv_push"), "flags_per_launch": st.get("flags_per_launch")}


def _tp_ipc_selftest(rank, world, port, n_gen, strict):
    """One rank of tests/test_gpu_model.py::test_rank_per_process_tensor_parallel_over_ipc: both ranks on HIP device 0, gloo for
    the handle exchange and the prompt's all-reduces; rank 0 prints the generated ids as JSON."""
    import json
    import torch
    import torch.distributed as dist
    from . import synthetic as S
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    cfg = dict(S.SMALL)
    if world == 4:
        cfg["n_kv_heads"] = 4
    w = S.make_weights(cfg, seed=41)
    prompt = S.prompt_tokens(cfg, n=12, seed=7)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    engine = HipTPEngine(cfg, w, rank, world, 0, 96)
    if strict:
        engine.s.set_strict(True)
    if os.environ.get("JH_TP_SELFTEST_BENCH"):   # the bench's rank-per-GPU tensor-parallel leg over the same two processes
        leg = tp_rank_bench(dist, engine, rank, world, prompt, n_gen - 1, cfg, device, torch.float32, use_ipc=True)
        if rank == 0:
            print(json.dumps({"leg": leg}), flush=True)
        dist.destroy_process_group()
        return
    ids = tp_generate_ipc(dist, engine, rank, world, prompt, n_gen, cfg, device, torch.float32)
    if rank == 0:
        print(json.dumps({"ids": [int(t) for t in ids]}), flush=True)
    dist.destroy_process_group()


def one_process_pipeline_bench(config, n_gpus, steps, warmup, prompt_n, devices=None, probe_iters=3, strict=False):
    """The one-process N-device host (jh_pipeline_*, BASELINE north_star): stage k on HIP device k, hops are stream-ordered
    peer copies.  Measures the single-stream (batch-1) decode rate and the aggregate rate of N sessions in flight over the
    same stage models -- EXACTLY `steps` tokens in each timed leg, bracketed by a synchronize of every stage stream.
    Returns a dict (also printed as JSON by `python -m jlama_amd.distributed --one-process ...`)."""
    import torch
    from . import _native as N, synthetic as S, synthetic_torch as ST
    from .model import HipPipeline, build_stage_models
    cfg = dict(getattr(S, config))
    devices = list(devices) if devices is not None else list(range(n_gpus))
    n = len(devices)
    prompt = S.prompt_tokens(cfg, n=prompt_n, seed=1234)
    per_session = max(1, steps // n)
    max_ctx = prompt.size + max(steps, warmup) + 8

    def weights_for_stage(k, rng, dev):
        torch.cuda.set_device(dev)
        w = ST.make_weights(cfg, seed=0, layers=rng, device=f"cuda:{dev}", need_embed=(k == 0 or cfg.get("tied", False)), need_head=(k == n - 1))
        torch.cuda.synchronize()
        return w

    N.init(devices[0])
    N.set_option("JH_STRICT_ORDER", 1 if strict else 0)   # every session created below starts in that mode (reference order = the N = 1 line's `value`)
    models = build_stage_models(cfg, weights_for_stage, devices)
    pipes = [HipPipeline(models, max_ctx) for _ in range(n)]
    tp0 = time.perf_counter()
    firsts = [p.prefill(prompt) for p in pipes]
    prefill_ms = (time.perf_counter() - tp0) * 1e3 / n

    def sync_all():
        for p in pipes:
            for s in p.sessions:
                s.synchronize()

    if warmup > 0:
        pipes[0].decode_n(firsts[0], prompt.size, warmup)
    for p in pipes[1:]:                       # every pipeline's graphs captured before anything is timed
        p.decode_n(firsts[pipes.index(p)], prompt.size, 1)
    # single stream: one session, `steps` tokens
    sync_all()
    t0 = time.perf_counter()
    toks = pipes[0].decode_n(firsts[0], prompt.size, steps)
    sync_all()
    dt_single = time.perf_counter() - t0
    # N sessions in flight: every pipeline queued before any is awaited
    sync_all()
    t0 = time.perf_counter()
    for p, f in zip(pipes, firsts):
        p.decode_n_async(f, prompt.size, per_session)
    outs = [p.decode_wait(per_session) for p in pipes]
    sync_all()
    dt_agg = time.perf_counter() - t0
    same = all(np.array_equal(o, outs[0]) for o in outs) and np.array_equal(outs[0], toks[:per_session])
    probe = None
    if cfg["weight_dtype"] == N.DT_Q4:         # dominant kernel (gate|up GEMV) on stage 0's layers, HIP events on its stream
        ms, b = pipes[0].sessions[0].kernel_bench(3, probe_iters)
        probe = {"us": round(ms * 1e3, 3), "bytes": b, "GBps": round(b / (ms * 1e-3) / 1e9, 1)}
    peer_access = pipes[0].peer_access()
    tp = None
    if n > 1 and cfg["weight_dtype"] == N.DT_Q4 and not os.environ.get("JH_BENCH_NO_TP_LEG"):
        for p in pipes:
            p.close()
        pipes, models = [], None
        try:
            tp = one_process_tp_leg(cfg, devices, prompt, steps)
        except Exception as e:   # noqa: BLE001 -- an optional leg must never cost the line
            tp = {"error": repr(e)[:400]}
    return {"mode": "one process, %d devices, hipMemcpyPeerAsync hops ordered by events" % n, "devices": devices, "tensor_parallel": tp,
            "order": "reference order (bit-exact ids)" if strict else "order-free kernels",
            "single_stream_tokens_per_s": round(steps / dt_single, 2), "single_stream_ms_per_token": round(dt_single / steps * 1e3, 4),
            "aggregate_tokens_per_s": round(per_session * n / dt_agg, 2), "aggregate_s": dt_agg, "sessions": n,
            "steps_per_session": per_session, "sessions_agree": bool(same), "peer_access": peer_access,
            "prefill_ms_per_session": round(prefill_ms, 2), "prompt_rows": int(prompt.size), "gate_up_probe": probe,
            "first_ids": [int(t) for t in toks[:8]]}


def one_process_tp_leg(cfg, devices, prompt, steps):
    """The head-split (tensor-parallel) group over the same devices, one shard per device (jh_tp_group_*: one graph replay per
    shard and token, partial rows pushed into every shard's slot over xGMI, summed in shard order): the single-stream decode
    rate -- the one number that CAN grow with the GPU count (a layer-split stream cannot).  The full model is
    generated on the first device and every shard's windows are copied to its own device."""
    import torch
    from . import synthetic_torch as ST
    from .model import HipLlamaModel, HipTPGroup
    n = len(devices)
    if cfg["n_kv_heads"] % n or cfg["n_heads"] % n:
        return {"skipped": f"{cfg['n_kv_heads']} kv heads do not split over {n} shards (the reference caps shards at the kv head count)"}
    torch.cuda.set_device(devices[0])
    w = ST.make_weights(cfg, seed=0, device=f"cuda:{devices[0]}")
    torch.cuda.synchronize()
    models = []
    for r, dev in enumerate(devices):
        lc, off = tp_shard_config(cfg, r, n)
        sw = tp_shard_weights(cfg, w, r, n, device=torch.device("cuda", dev))
        torch.cuda.synchronize()
        models.append(HipLlamaModel(lc, sw, device=dev, kv_head_offset=off))
        del sw
    del w
    torch.cuda.empty_cache()
    n_prompt = int(prompt.size)                              # the whole prompt, in chunks (jh_tp_group_forward)
    grp = HipTPGroup(models, n_prompt + steps + 16)
    grp.forward(prompt[:n_prompt], 0)
    first = grp.sample()
    grp.decode_n(first, n_prompt, min(8, steps))             # graph capture, untimed
    t0 = time.perf_counter()
    ids = grp.decode_n(first, n_prompt, steps)
    dt = time.perf_counter() - t0
    st = grp.status()                                        # which loop actually ran the timed steps, and whether a meeting ever timed out
    grp.close()
    return {"shards": n, "devices": list(devices), "single_stream_tokens_per_s": round(len(ids) / dt, 2), "steps": int(len(ids)),
            "prompt_rows": n_prompt, "first_ids": [int(t) for t in ids[:8]], "loop": st["mode"], "meeting_timeouts": st["timeouts"],
            "gemv_push": st["gemv_push"], "flags_per_launch": st["flags_per_launch"],
            "note": "one-process tensor-parallel group, one head-split shard per device; 2 meetings per layer over xGMI peer stores "
                    "(`loop` says which host loop the timed steps ran on)"}


def multi_gpu_extras(args, cfg, gate_up_probe, device_index=0):
    """What every N>1 bench line carries besides its throughput: the dominant kernel's roofline (measured on one stage: the
    kernels of a layer shard are the 1-GPU kernels) with the committed counter traffic, and the CPU baseline (same leg as N=1,
    full model on the host cores)."""
    import bench
    from . import synthetic_torch as ST
    traffic, us_rocprof, prof = bench._profiled(args.config)
    roof = None
    if gate_up_probe:
        roof = {"bound": "hbm", "kernel": bench.DOMINANT_KERNEL, "achieved": gate_up_probe["GBps"], "peak": bench.HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gate_up_probe["GBps"] / bench.HBM_PEAK_GBS, 4), "traffic": traffic, "bytes_per_launch": gate_up_probe["bytes"],
                "us_per_launch": gate_up_probe["us"], "us_per_launch_rocprof": us_rocprof, "profile": prof,
                "note": "dominant kernel of one stage (the layer shards run the 1-GPU kernels), HIP events on the stage's stream"}
    cpu = None
    if not args.no_cpu_baseline:
        import torch
        torch.cuda.set_device(device_index)
        w = ST.make_weights(cfg, seed=0, device=f"cuda:{device_index}")
        host_w = ST.to_host(w)
        del w
        torch.cuda.empty_cache()
        cpu = bench.cpu_baseline(cfg, host_w, 8, args.cpu_steps)
    return roof, cpu


hard_exit = {"now": False}   # set when a leg left a collective stuck: bench.py prints rank 0's line and every rank leaves through os._exit


def bench_pipeline(args, cfg, backend="nccl", engine_factory=None, tp_engine_factory=None):
    """bench.py's N>1 leg (one rank per GPU under torch.distributed.run).  Returns the JSON dict on rank 0.
    Three measurements: N sessions in flight through the rank-per-GPU RCCL pipeline (the timed K tokens of the contract),
    the same pipeline with ONE session (single-stream, batch-1: bounded by a single GPU's rate by construction), and --
    from rank 0 in a child process -- the one-process N-device host over the same GPUs.
    Then the head-split group with one shard per rank (tp_rank_bench).  `value` of the line is the best SINGLE-STREAM rate over
    the two modes -- like for like with the N = 1 line, which is one batch-1 stream -- and `config.parallelism` names the mode;
    the throughput of N sessions in flight is reported beside it as `aggregate_tokens_per_s`.
    ``backend="gloo"`` + ``engine_factory(rank, world, n_sessions, max_ctx) -> ShardEngine`` (+ ``tp_engine_factory(rank, world,
    max_ctx) -> TPEngine``) run the same control flow on CPU (tests/test_distributed.py drives every branch of the N>1 line that
    way; the product path is nccl + HipShardEngine / HipTPEngine)."""
    import json
    import subprocess
    import sys
    import torch
    import torch.distributed as dist
    from . import synthetic as S
    on_gpu = backend == "nccl"
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if world == 1:   # JH_BENCH_FORCE_PIPELINE=1 without a launcher: a one-rank group over the same code path
        import socket
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(port))):
            os.environ.setdefault(k, v)
    if on_gpu:
        if not os.environ.get("JH_KEEP_NCCL_DEBUG"):   # level VERSION and up print a banner on stdout; the JSON line is the contract
            os.environ["NCCL_DEBUG"] = "NONE"
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group(backend="nccl", device_id=device)
        dev_sync = torch.cuda.synchronize
    else:
        device = torch.device("cpu")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        dev_sync = lambda: None
    tok_group = dist.new_group() if world > 1 else None   # the sampled id's way back: its own communicator (see _grp)
    # every rank contributes a one: what the communicator actually spans (rccl_ranks_seen in the line) and which device each rank drives
    _seen = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(_seen)
    ranks_seen = int(_seen.item())
    _devs = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(world)]
    dist.all_gather(_devs, torch.tensor([local], dtype=torch.int32, device=device))
    rank_devices = [int(d.item()) for d in _devs]
    L, E = cfg["n_layers"], cfg["embedding_length"]
    ls, le = layer_range(rank, world, L)
    prompt = S.prompt_tokens(cfg, n=args.prompt, seed=1234)
    cpu_base = None
    if on_gpu and rank == 0 and not getattr(args, "no_cpu_baseline", False):
        # the CPU leg of the N=1 line (full model on the host cores), before this rank's shard occupies the GPU; the other
        # ranks meanwhile build their shards and wait in the first collective
        _, cpu_base = multi_gpu_extras(args, cfg, None, local)
    # Sessions in flight for `value`: one per GPU (the fewest that keep every stage busy).  A second timed leg runs two per GPU:
    # with one, a stage idles through every launch edge of its graph; two sessions' graphs on two streams fill each other's
    # edges (one GPU, no sharding: 673 -> 871 tok/s, profiles/r02j_*).  Reported beside `value`, never as it.
    spg = max(1, int(os.environ.get("JH_BENCH_SESSIONS_PER_GPU", "1")))
    n_sess = world * spg
    n_sess2 = 0 if os.environ.get("JH_BENCH_NO_EXTRA_LEG") else 2 * n_sess
    n_all = max(n_sess, n_sess2)
    steps_per_session = max(1, args.steps // n_sess)
    single_steps = args.steps                      # the leg `value` can come from: EXACTLY K steps of one stream
    strict = not getattr(args, "fast_order", False)   # reference order, as the N = 1 line's `value`
    max_ctx = prompt.size + max(steps_per_session, args.warmup, single_steps) + 8
    if engine_factory is not None:
        engine = engine_factory(rank, world, n_all, max_ctx)
    else:
        from . import synthetic_torch as ST
        w = ST.make_weights(cfg, seed=0, layers=(ls, le), device=device, need_embed=(rank == 0 or cfg.get("tied", False)),
                            need_head=(rank == world - 1))
        engine = HipShardEngine(cfg, w, rank, world, local, n_sessions=n_all, max_ctx=max_ctx, strict=strict)
        del w
    firsts_all = [pipeline_prefill(dist, engine, rank, world, j, prompt, E, device, torch.float32) for j in range(n_all)]
    firsts = firsts_all[:n_sess]
    if on_gpu:   # the RCCL banner (printed at communicator creation through buffered C stdio) leaves every rank's buffer NOW,
        try:     # long before rank 0 prints the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:   # noqa: BLE001
            pass
    # The decode loop: stream-ordered hops (pipeline_decode_streamed: RCCL send/recv on the sessions' streams, token id as a
    # device word) unless JH_PIPELINE_HOST_SYNC=1 asks for the host-synchronised reference loop.  The first ids of both are
    # compared before anything is timed; a mismatch falls back to the reference loop and is reported.
    streamed = not os.environ.get("JH_PIPELINE_HOST_SYNC")
    check_steps = max(1, min(4, steps_per_session))
    ref_ids = pipeline_decode(dist, engine, rank, world, firsts, prompt.size, check_steps, E, device, torch.float32, n_sessions=n_sess, tok_group=tok_group)
    streamed_ok = None
    if streamed:
        got = pipeline_decode_streamed(dist, engine, rank, world, firsts, prompt.size, check_steps, E, device, torch.float32, n_sessions=n_sess, tok_group=tok_group)
        flag = torch.tensor([1 if (rank != world - 1 or np.array_equal(got, ref_ids)) else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        streamed_ok = bool(int(flag.item()))
        streamed = streamed_ok
    decode = pipeline_decode_streamed if streamed else pipeline_decode
    if args.warmup > 0:  # untimed decode ticks on the same sessions' KV tail (positions beyond the timed range are rewritten)
        decode(dist, engine, rank, world, firsts, prompt.size, max(1, args.warmup // n_sess), E, device, torch.float32, n_sessions=n_sess, tok_group=tok_group)

    def timed(fn):
        if on_gpu:   # see bench._quiesce: a default-stream synchronize can stay blocked for tens of ms of wall time behind earlier
            for _ in range(50):   # graph replays although the device is idle; the closing synchronize must not measure that
                time.sleep(0.02)
                tq = time.perf_counter()
                dev_sync()
                torch.cuda.current_stream().synchronize()
                if time.perf_counter() - tq < 1e-3:
                    break
        dist.barrier()
        dev_sync()
        t0 = time.perf_counter()
        r = fn()
        dev_sync()
        dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return r, float(dt.item())

    toks, dt = timed(lambda: decode(dist, engine, rank, world, firsts, prompt.size, steps_per_session, E, device, torch.float32, n_sessions=n_sess, tok_group=tok_group))
    _, dt1 = timed(lambda: decode(dist, engine, rank, world, firsts[:1], prompt.size, single_steps, E, device, torch.float32, n_sessions=1, tok_group=tok_group))
    host_sync = None
    if streamed:   # the reference loop beside it (what the hops cost when the host sits in them)
        _, dth = timed(lambda: pipeline_decode(dist, engine, rank, world, firsts, prompt.size, steps_per_session, E, device, torch.float32, n_sessions=n_sess, tok_group=tok_group))
        host_sync = round(steps_per_session * n_sess / dth, 2)
    total = steps_per_session * n_sess
    double_up = None
    if n_sess2:   # the same K tokens with twice the sessions in flight
        sps2 = max(1, args.steps // n_sess2)
        decode(dist, engine, rank, world, firsts_all, prompt.size, 1, E, device, torch.float32, n_sessions=n_sess2, tok_group=tok_group)
        _, dt2 = timed(lambda: decode(dist, engine, rank, world, firsts_all, prompt.size, sps2, E, device, torch.float32, n_sessions=n_sess2, tok_group=tok_group))
        double_up = {"sessions_in_flight": n_sess2, "steps_per_session": sps2, "aggregate_tokens_per_s": round(sps2 * n_sess2 / dt2, 2)}
    one_proc = None
    if world > 1 and not os.environ.get("JH_BENCH_NO_ONE_PROCESS"):
        # The one-process host drives the same GPUs from a child of rank 0 (its own HIP contexts).  The other ranks wait on the
        # rendezvous store -- on the HOST: an RCCL barrier would park a spinning kernel on every GPU the child is measuring.
        from datetime import timedelta
        from torch.distributed.distributed_c10d import _get_default_store
        store = _get_default_store()
        if rank == 0 and not on_gpu:
            one_proc = {"skipped": "no GPU (control-flow run)"}
            store.set("jh_one_process_leg_done", "1")
        elif rank == 0:
            try:
                env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                r = subprocess.run([sys.executable, "-m", "jlama_amd.distributed", "--one-process", "--config", args.config, "--gpus", str(world),
                                    "--steps", str(args.steps), "--warmup", str(args.warmup), "--prompt", str(args.prompt),
                                    "--strict", "1" if strict else "0"],
                                   capture_output=True, text=True, timeout=420, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
                one_proc = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else {"error": (r.stderr or "")[-400:]}
            except Exception as e:   # noqa: BLE001 -- the contract line must be printed whatever this optional leg does
                one_proc = {"error": repr(e)[:400]}
            store.set("jh_one_process_leg_done", "1")
        else:
            try:
                store.wait(["jh_one_process_leg_done"], timedelta(seconds=480))
            except Exception:   # noqa: BLE001 -- fall through to the collective barrier below
                pass
    dist.barrier()
    # ---- the head-split (tensor-parallel) group, one shard per rank: the mode whose single-stream rate CAN grow with N
    tp_leg = None
    can_tp = world > 1 and cfg["n_kv_heads"] % world == 0 and cfg["n_heads"] % world == 0 and cfg["hidden_length"] % (32 * world) == 0
    if world > 1 and not can_tp:
        tp_leg = {"skipped": f"{cfg['n_kv_heads']} kv heads / H = {cfg['hidden_length']} do not split over {world} shards"}
    elif world > 1 and not os.environ.get("JH_BENCH_NO_TP_LEG") and (tp_engine_factory is not None or (on_gpu and cfg["weight_dtype"] == 3)):
        # This leg has never run on more than one GPU (1-GPU boxes): it must not be able to cost the line.  It runs on a helper
        # thread; a rank whose leg raises, or is still inside a collective when the budget is over, says so on the rendezvous STORE
        # (TCP, not RCCL), every rank reads everybody's verdict there, and if one is not "ok" no further collective is issued: rank 0
        # prints the line from what it has measured and all ranks leave through os._exit (bench.py: `hard_exit`).
        import threading
        from datetime import timedelta
        from torch.distributed.distributed_c10d import _get_default_store
        state = {"leg": None, "err": None}

        def _tp_work():
            try:
                if on_gpu:
                    torch.cuda.set_device(local)             # the device is a per-thread setting of the HIP runtime
                tp_ctx = prompt.size + args.steps + 16
                if tp_engine_factory is not None:
                    tp_engine = tp_engine_factory(rank, world, tp_ctx)
                else:
                    from . import synthetic_torch as ST
                    wf = ST.make_weights(cfg, seed=0, device=device)      # the full model on every rank's own device; its windows stay
                    torch.cuda.synchronize()
                    tp_engine = HipTPEngine(cfg, wf, rank, world, local, tp_ctx)
                    del wf
                    torch.cuda.empty_cache()
                    if strict:
                        tp_engine.s.set_strict(True)
                state["leg"] = tp_rank_bench(dist, tp_engine, rank, world, prompt, args.steps, cfg, device, torch.float32, use_ipc=on_gpu)
            except Exception as e:   # noqa: BLE001
                state["err"] = repr(e)[:400]

        budget = float(os.environ.get("JH_BENCH_TP_TIMEOUT", "420"))
        th = threading.Thread(target=_tp_work, daemon=True)
        th.start()
        th.join(timeout=budget)
        mine = "hung" if th.is_alive() else ("error" if state["err"] else "ok")
        store = _get_default_store()
        store.set(f"jh_tp_leg_{rank}", mine)
        verdicts = []
        for r in range(world):
            try:
                store.wait([f"jh_tp_leg_{r}"], timedelta(seconds=budget + 60))
                verdicts.append(store.get(f"jh_tp_leg_{r}").decode())
            except Exception:   # noqa: BLE001 -- that rank never reported
                verdicts.append("silent")
        if all(v == "ok" for v in verdicts):
            tp_leg = state["leg"]
        else:
            tp_leg = {"error": state["err"] or f"tensor-parallel leg did not complete on every rank within {budget:.0f} s", "rank_verdicts": verdicts}
            hard_exit["now"] = True                          # a collective may be stuck: no barrier, no destroy_process_group from here on
    out = None
    if rank == 0:
        gate_up = None
        if on_gpu and cfg["weight_dtype"] == 3:   # DT_Q4: dominant kernel on this rank's layers
            ms_p, b_p = engine.sessions[0].kernel_bench(3, getattr(args, "probe_iters", 3))
            gate_up = {"us": round(ms_p * 1e3, 3), "bytes": b_p, "GBps": round(b_p / (ms_p * 1e-3) / 1e9, 1)}
        roof_k = None
        if on_gpu:
            saved = getattr(args, "no_cpu_baseline", False)
            args.no_cpu_baseline = True
            roof_k, _ = multi_gpu_extras(args, cfg, gate_up, local)
            args.no_cpu_baseline = saved
        tps = total / dt
        wbytes, kvb = S.weight_bytes(cfg), S.kv_bytes_per_position(cfg)
        bytes_per_token = wbytes + kvb * (prompt.size + (steps_per_session - 1) / 2.0 + 2)
        # `value`: ONE batch-1 stream of K tokens, whichever mode serves it faster (the N = 1 line is one stream too)
        pp_single = single_steps / dt1
        tp_single = (tp_leg or {}).get("single_stream_tokens_per_s") or 0.0
        best_tp = world > 1 and tp_single > pp_single
        value, value_dt = (tp_single, tp_leg["seconds"]) if best_tp else (pp_single, dt1)
        parallelism = (f"tensor parallel tp{world}: head-split shards, one per GPU, {tp_leg['decode']}" if best_tp else
                       f"layer-sharded pp{world} ({L // world} layers/GPU), RCCL send/recv of [1,E] F32, one stream through all stages")
        out = {"metric": "decode tokens/sec Llama-3-8B JQ4, 128-tok prompt" if args.config == "LLAMA3_8B" else f"decode tokens/sec {args.config} JQ4",
               "value": round(value, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(value_dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
               "scaling_detail": f"one batch-1 stream of {args.steps} tokens whatever N (total work fixed): value = the best single-stream rate "
                                 f"over the two modes (layer split {round(pp_single, 2)}, tensor parallel {round(tp_single, 2) if tp_single else None} tok/s); "
                                 f"the throughput of {n_sess} independent sessions in flight is aggregate_tokens_per_s",
               "aggregate_tokens_per_s": round(tps, 2), "aggregate_steps": total,
               "single_stream_tokens_per_s": round(pp_single, 2), "tensor_parallel_tokens_per_s": round(tp_single, 2) if tp_single else None,
               "order": "reference order (bit-exact ids)" if strict else "order-free kernels", "vs_baseline": None,
               "rccl_ranks_seen": ranks_seen, "rank_devices": rank_devices,
               "dtype": "i8xq4->f32", "data": "synthetic",
               "config": {"workload": f"{args.config} JQ4, {prompt.size}-row prefill + {args.steps} greedy decode steps, batch 1 (one stream); "
                                      f"aggregate leg: {steps_per_session} steps x {n_sess} sessions in flight", "parallelism": parallelism,
                          "sessions_in_flight": n_sess, "sessions_per_gpu": spg, "aggregate_tokens_per_s": round(tps, 2),
                          "per_session_tokens_per_s": round(tps / n_sess, 2), "two_sessions_per_gpu": double_up,
                          "single_stream_tokens_per_s": round(single_steps / dt1, 2), "single_stream_steps": single_steps,
                          "hops": "stream-ordered (RCCL send/recv on the session streams, token id fed back as a device word)" if streamed
                                  else "host-synchronised", "streamed_ids_equal_host_synchronised": streamed_ok,
                          "host_synchronised_aggregate_tokens_per_s": host_sync,
                          "note": "a layer-split stream passes through all GPUs in sequence and cannot exceed the 1-GPU rate (SURVEY.md 8d); "
                                  "the head-split group can, which is why it is a candidate for value"},
               "roofline": roof_k if roof_k else {"bound": "hbm", "kernel": "whole pipeline (control-flow run: no GPU)", "achieved": None,
                                                   "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None},
               "pipeline_roofline": {"achieved_GBps_per_gpu": round(bytes_per_token * tps / 1e9 / world, 1),
                                     "frac_of_8TBps": round(bytes_per_token * tps / 1e9 / world / 8000.0, 4),
                                     "note": "aggregate leg, per-GPU average: algorithmic bytes of all sessions / time / GPUs"},
               "token_roofline": {"bytes_per_token": int(bytes_per_token), "achieved_GBps": round(bytes_per_token * value / 1e9, 1),
                                  "frac_of_8TBps_x_N": round(bytes_per_token * value / 1e9 / (8000.0 * world), 4),
                                  "note": "the single stream of `value` against N GPUs' HBM (a head-split token reads 1/N of the weights per GPU)"},
               "cpu_baseline": cpu_base, "one_process_pipeline": one_proc,
               "tensor_parallel": {"rank_per_gpu": tp_leg,
                                   "one_process": (one_proc or {}).get("tensor_parallel") if isinstance(one_proc, dict) else None}}
    if not hard_exit["now"]:
        dist.destroy_process_group()
    # RCCL prints its version banner through C stdio, which is block-buffered when stdout is a pipe and would otherwise land AFTER
    # the JSON line at exit: flush it now so that the contract line is the last thing on stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:   # noqa: BLE001
        pass
    return out


if __name__ == "__main__":
    import argparse
    import json
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # only matters when several shards share one device (loopback runs)
    ap = argparse.ArgumentParser()
    ap.add_argument("--one-process", action="store_true")
    ap.add_argument("--tp-ipc-selftest", action="store_true")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--port", type=int, default=29533)
    ap.add_argument("--n-gen", type=int, default=12)
    ap.add_argument("--strict", type=int, default=0)
    ap.add_argument("--config", default="LLAMA3_8B")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--devices", default="")
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--prompt", type=int, default=128)
    a = ap.parse_args()
    if a.tp_ipc_selftest:
        _tp_ipc_selftest(a.rank, a.world, a.port, a.n_gen, a.strict)
        raise SystemExit(0)
    devs = [int(d) for d in a.devices.split(",")] if a.devices else None
    print(json.dumps(one_process_pipeline_bench(a.config, a.gpus, a.steps, a.warmup, a.prompt, devs, strict=bool(a.strict))), flush=True)
