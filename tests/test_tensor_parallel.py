"""Tensor-parallel (head-split) path, SURVEY.md 8(e)/f2 -- CPU tests with the ORACLE as the shard engine.

* shard geometry + weight windows (DistributedContext.java:79-98, Weights.getLoadOffsets :101-120);
* the oracle's lock-step N-shard forward (partials summed in shard order before the residual) stays within the Q8
  noise floor of the un-sharded forward, and is EXACTLY the un-sharded forward for N=1;
* world_size-2 gloo run of jlama_amd.distributed.tp_generate == the lock-step oracle, bit for bit (a two-term sum is
  order-independent)."""
import os
import socket

import numpy as np
import pytest

from jlama_amd import distributed as D, jq4, synthetic as S


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_geometry_and_windows():
    cfg = dict(S.SMALL)   # 8 heads, 2 kv heads, hs 128, E 512, H 1024
    w = S.make_weights(cfg, seed=1)
    with pytest.raises(ValueError):
        D.tp_shard_config(cfg, 0, 4)   # more shards than kv heads (JlamaService.java:65-68)
    full_o = jq4.dequantize_q4(w[(0, S.W_O)]["data"], w[(0, S.W_O)]["scales"])
    full_q = jq4.dequantize_q4(w[(0, S.W_Q)]["data"], w[(0, S.W_Q)]["scales"])
    full_d = jq4.dequantize_q4(w[(1, S.W_DOWN)]["data"], w[(1, S.W_DOWN)]["scales"])
    for r in range(2):
        lc, off = D.tp_shard_config(cfg, r, 2)
        assert (lc["n_heads"], lc["n_kv_heads"], lc["hidden_length"], off) == (4, 1, 512, r)
        sw = D.tp_shard_weights(cfg, w, r, 2)
        A = lc["n_heads"] * lc["head_size"]
        assert sw[(0, S.W_Q)]["shape"] == (A, 512) and sw[(0, S.W_K)]["shape"] == (128, 512)
        assert sw[(0, S.W_O)]["shape"] == (512, A) and sw[(1, S.W_DOWN)]["shape"] == (512, 512)
        np.testing.assert_array_equal(jq4.dequantize_q4(sw[(0, S.W_Q)]["data"], sw[(0, S.W_Q)]["scales"]), full_q[r * A:(r + 1) * A])
        np.testing.assert_array_equal(jq4.dequantize_q4(sw[(0, S.W_O)]["data"], sw[(0, S.W_O)]["scales"]), full_o[:, r * A:(r + 1) * A])
        np.testing.assert_array_equal(jq4.dequantize_q4(sw[(1, S.W_DOWN)]["data"], sw[(1, S.W_DOWN)]["scales"]), full_d[:, r * 512:(r + 1) * 512])
        assert sw[(-1, S.W_EMBED)] is w[(-1, S.W_EMBED)] and sw[(0, S.W_NORM1)] is w[(0, S.W_NORM1)]


def _shard_models(O, cfg, w, size):
    ms = []
    for r in range(size):
        lc, off = D.tp_shard_config(cfg, r, size)
        ms.append(O.OracleModel(lc, D.tp_shard_weights(cfg, w, r, size), kv_head_offset=off))
    return ms


def test_oracle_lockstep_tp_matches_unsharded(oracle):
    cfg = dict(S.SMALL)
    w = S.make_weights(cfg, seed=12)
    prompt = S.prompt_tokens(cfg, n=24, seed=5)
    full = oracle.OracleModel(cfg, w)
    want = full.session().forward(prompt, 0)
    one = _shard_models(oracle, cfg, w, 1)
    np.testing.assert_array_equal(oracle.forward_tp([one[0].session()], prompt, 0), want)   # N=1: the same arithmetic
    two = _shard_models(oracle, cfg, w, 2)
    got = oracle.forward_tp([m.session() for m in two], prompt, 0)
    # the K split changes the float grouping of o/down (partial sums per shard): Q8-noise-level agreement, and the
    # first row (no earlier code flips) to float rounding
    assert np.abs(got - want).max() <= 4e-2 * np.abs(want).max()
    assert np.abs(got[0] - want[0]).max() <= 1e-4 * np.abs(want[0]).max()
    lt, lw = full.sample(got[-1])[1], full.sample(want[-1])[1]
    assert np.abs(lt - lw).max() <= 4e-2


def _tp_worker(rank, world, port, cfg, prompt, n_gen, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    import torch
    import torch.distributed as dist
    from oracle import oracle as O

    class OracleTPEngine(D.TPEngine):
        def __init__(self):
            w = S.make_weights(cfg, seed=12)
            lc, off = D.tp_shard_config(cfg, rank, world)
            self.full = O.OracleModel(cfg, w) if rank == 0 else None     # rank 0 samples with the replicated head
            self.m = O.OracleModel(lc, D.tp_shard_weights(cfg, w, rank, world), kv_head_offset=off)
            self.s = self.m.session()
            self.x = self.x1 = None
            self.pos = 0

        def set_row(self, token, pos):
            self.x, self.pos = self.m.embed_rows([token]), pos

        def attn(self, layer, partial):
            partial.copy_(torch.from_numpy(self.s.tp_attn(layer, self.x, self.pos)[0]))

        def ffn(self, layer, reduced, partial):
            self.x1 = self.x + reduced.numpy()[None, :]
            partial.copy_(torch.from_numpy(self.s.tp_ffn(layer, self.x1)[0]))

        def finish_layer(self, reduced):
            self.x = self.x1 + reduced.numpy()[None, :]

        def sample(self):
            return self.full.sample(self.x[0])[0]

        # prompt chunks: [rows, E] partials, one all-reduce per half-layer and chunk (distributed.tp_forward_prompt)
        def rows_max(self):
            return 4 if rank == 0 else 5          # (the host takes the minimum over the ranks: chunks of 4 rows, then single rows)

        def set_rows(self, tokens, pos):
            self.x, self.pos = self.m.embed_rows(list(tokens)), pos
            self.chunks = getattr(self, "chunks", 0) + 1

        def attn_rows(self, layer, partial):
            partial.copy_(torch.from_numpy(self.s.tp_attn(layer, self.x, self.pos)))

        def ffn_rows(self, layer, reduced, partial):
            self.x1 = self.x + reduced.numpy()
            partial.copy_(torch.from_numpy(self.s.tp_ffn(layer, self.x1)))

        def finish_layer_rows(self, reduced):
            self.x = self.x1 + reduced.numpy()

        def finish_rows(self):
            self.x = self.x[-1:]

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    eng = OracleTPEngine()
    toks = D.tp_generate(dist, eng, rank, prompt, n_gen, cfg, "cpu", torch.float32)
    if rank == 0:
        q.put((toks.tolist(), getattr(eng, "chunks", 0)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_process_tp_equals_lockstep_oracle(oracle):
    import torch.multiprocessing as mp
    cfg = dict(S.TINY)   # 4 heads, 2 kv heads
    prompt = S.prompt_tokens(cfg, n=6, seed=2)
    n_gen = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_tp_worker, args=(r, 2, port, cfg, prompt, n_gen, q)) for r in range(2)]
    for p in ps:
        p.start()
    got, chunks = q.get(timeout=240)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert chunks == 1          # 7 prompt rows = one chunk of 4 (the ranks' minimum) + 3 single rows: both host paths ran
    # lock-step reference: both shards in this process, partials summed r=0 then r=1
    w = S.make_weights(cfg, seed=12)
    full = oracle.OracleModel(cfg, w)
    sess = [m.session() for m in _shard_models(oracle, cfg, w, 2)]
    x = oracle.forward_tp(sess, prompt, 0)
    want, pos = [], prompt.size
    for _ in range(n_gen):
        t = full.sample(x[-1])[0]
        want.append(t)
        x = oracle.forward_tp(sess, [t], pos)
        pos += 1
    assert got == want


def test_four_process_tp_matches_lockstep_oracle_within_noise(oracle):
    """world_size 4 (a model with 4 kv heads): the all-reduce's summation order over four partials is the backend's, so the
    comparison with the shard-ordered lock-step oracle is at the float-regrouping level: greedy ids must agree wherever the
    un-sharded oracle's own top-2 margin is meaningful."""
    import torch.multiprocessing as mp
    cfg = dict(S.TINY)
    cfg.update(n_heads=8, n_kv_heads=4, head_size=32, embedding_length=256, hidden_length=512)
    prompt = S.prompt_tokens(cfg, n=5, seed=4)
    n_gen = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_tp_worker, args=(r, 4, port, cfg, prompt, n_gen, q)) for r in range(4)]
    for p_ in ps:
        p_.start()
    got, _ = q.get(timeout=300)
    for p_ in ps:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    w = S.make_weights(cfg, seed=12)
    full = oracle.OracleModel(cfg, w)
    sess = [m.session() for m in _shard_models(oracle, cfg, w, 4)]
    x = oracle.forward_tp(sess, prompt, 0)
    pos = prompt.size
    for g in got:                       # teacher-forced on the distributed run's tokens
        t, lg = full.sample(x[-1])
        top2 = np.partition(lg, -2)[-2:]
        assert g == t or top2[1] - top2[0] <= 4e-2, (g, t)
        x = oracle.forward_tp(sess, [g], pos)
        pos += 1


def test_prompt_host_picks_chunks_or_rows_consistently():
    """distributed.tp_forward_prompt: the chunk size is the minimum of rows_max() over the ranks; pieces shorter than TP_ROWS_MIN and
    engines without a batched path go row by row; every position is fed exactly once, in order."""
    import torch

    class FakeDist:
        class ReduceOp:
            MIN, SUM = "min", "sum"

        def __init__(self, other_cap):
            self.other_cap = other_cap

        def all_reduce(self, t, op=None):
            if op == "min":
                t.clamp_(max=self.other_cap)

    class Rec(D.TPEngine):
        def __init__(self, cap):
            self.cap, self.log = cap, []

        def rows_max(self):
            return self.cap

        def set_row(self, token, pos):
            self.log.append(("row", pos, 1))

        def attn(self, layer, partial): pass
        def ffn(self, layer, reduced, partial): pass
        def finish_layer(self, reduced): pass

        def set_rows(self, tokens, pos):
            self.log.append(("rows", pos, len(tokens)))

        def attn_rows(self, layer, partial):
            assert partial.shape[0] == self.log[-1][2]

        def ffn_rows(self, layer, reduced, partial): pass
        def finish_layer_rows(self, reduced): pass
        def finish_rows(self): pass

    cfg = dict(S.TINY)
    buf = torch.empty(cfg["embedding_length"])
    for mine, other, n, want in [
        (256, 256, 129, [("rows", 0, 129)]),
        (256, 6, 15, [("rows", 0, 6), ("rows", 6, 6), ("row", 12, 1), ("row", 13, 1), ("row", 14, 1)]),   # 3 left: below TP_ROWS_MIN
        (0, 256, 5, [("row", i, 1) for i in range(5)]),                                                 # one rank has no batched path
        (256, 256, 3, [("row", i, 1) for i in range(3)]),
        (8, 8, 20, [("rows", 0, 8), ("rows", 8, 8), ("rows", 16, 4)]),
    ]:
        eng = Rec(mine)
        info = D.tp_forward_prompt(FakeDist(other), eng, list(range(n)), 0, (0, 2), cfg, "cpu", torch.float32, buf)
        assert eng.log == want, (mine, other, n, eng.log)
        assert sum(c for _, _, c in eng.log) == n and info["rows_batched"] == sum(c for k, _, c in eng.log if k == "rows")
    # a shard that refuses a chunk at some position (jh_tp_set_rows: JH_ERR_UNSUPPORTED) sends every rank to the row loop for the rest

    class Refuses(Rec):
        def set_rows(self, tokens, pos):
            if pos >= 8:
                raise D.UnsupportedOperation(-3, "no batched path at this position")
            super().set_rows(tokens, pos)

    eng = Refuses(4)
    info = D.tp_forward_prompt(FakeDist(4), eng, list(range(14)), 0, (0, 2), cfg, "cpu", torch.float32, buf)
    assert eng.log == [("rows", 0, 4), ("rows", 4, 4)] + [("row", i, 1) for i in range(8, 14)] and info["rows_batched"] == 8

