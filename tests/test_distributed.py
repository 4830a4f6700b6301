"""Layer-sharded pipeline (jlama_amd.distributed) on CPU: world_size 2 and 4 over gloo, with the ORACLE as the shard
engine (tests may use the oracle; the product engine is HipShardEngine).  Checks the DistributedContext semantics the
reference tests in-process (jlama-net/.../JlamaServiceTest.java:50-106, DistributedServiceTest.java:43-118): every
session's tokens equal the un-sharded single-process generation, bit for bit (transfers are copies)."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cfg, n_prompt, steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="2")
    import torch
    import torch.distributed as dist
    from jlama_amd import distributed as D, synthetic as S
    from oracle import oracle as O

    class OracleShardEngine(D.ShardEngine):
        def __init__(self):
            ls, le = D.layer_range(rank, world, cfg["n_layers"])
            self.w = S.make_weights(cfg, seed=3)
            self.m = O.OracleModel(cfg, self.w, layer_range=(ls, le))
            self.s = [self.m.session() for _ in range(world)]
            self.last = [None] * world

        def forward_tokens(self, session, tokens, start_pos, x_out):
            x = self.s[session].forward(np.asarray(tokens, dtype=np.int32), start_pos)
            x_out.copy_(torch.from_numpy(x))
            self.last[session] = x[-1]

        def forward_x(self, session, x_in, n, start_pos, x_out):
            x = self.s[session].forward(None, start_pos, x=x_in.numpy()[:n])
            x_out.copy_(torch.from_numpy(x))
            self.last[session] = x[-1]

        def sample(self, session):
            return self.m.sample(self.last[session])[0]

        # the stage form pipeline_decode_streamed drives (HipShardEngine: jh_stage_decode_async on the session's stream)
        def stream(self, session):
            return None

        def stage_step(self, session, token, x_in, pos, x_out, token_out):
            if token is not None:
                x = self.s[session].forward(np.array([int(token[0])], dtype=np.int32), pos)
            else:
                x = self.s[session].forward(None, pos, x=x_in.numpy()[:1])
            if x_out is not None:
                x_out.copy_(torch.from_numpy(x))
            if token_out is not None:
                token_out[0] = int(self.m.sample(x[-1])[0])

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    eng = OracleShardEngine()
    E = cfg["embedding_length"]
    prompts = [S.prompt_tokens(cfg, n=n_prompt, seed=100 + j) for j in range(world)]
    firsts = [D.pipeline_prefill(dist, eng, rank, world, j, prompts[j], E, "cpu", torch.float32) for j in range(world)]
    toks = D.pipeline_decode(dist, eng, rank, world, firsts, prompts[0].size, steps, E, "cpu", torch.float32)
    # single-stream form (bench.py's batch-1 leg): ONE session through all stages; session 0 decodes the same positions again
    single = D.pipeline_decode(dist, eng, rank, world, firsts[:1], prompts[0].size, steps, E, "cpu", torch.float32, n_sessions=1)
    # the stream-ordered schedule (what bench.py --gpus N runs over RCCL): same ids, sessions in flight and single stream
    streamed = D.pipeline_decode_streamed(dist, eng, rank, world, firsts, prompts[0].size, steps, E, "cpu", torch.float32)
    streamed1 = D.pipeline_decode_streamed(dist, eng, rank, world, firsts[:1], prompts[0].size, steps, E, "cpu", torch.float32, n_sessions=1)
    if rank == world - 1:
        assert streamed.tolist() == toks.tolist() and streamed1.tolist() == single.tolist()
        q.put((firsts, toks.tolist(), single.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_layer_sharded_pipeline_matches_single_process(world):
    import torch.multiprocessing as mp
    from jlama_amd import distributed as D, synthetic as S
    from oracle import oracle as O
    cfg = dict(S.TINY)
    cfg["n_layers"] = 8 if world == 8 else 4     # world 8 = the driver's largest launch (one layer per rank here)
    n_prompt, steps = (4, 3) if world == 8 else (6, 5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg, n_prompt, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    firsts, toks, single = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference: AbstractModel.generate per session
    w = S.make_weights(cfg, seed=3)
    om = O.OracleModel(cfg, w)
    for j in range(world):
        prompt = S.prompt_tokens(cfg, n=n_prompt, seed=100 + j)
        want, _, _ = om.session().generate(prompt, steps + 1)
        assert firsts[j] == want[0]
        np.testing.assert_array_equal(np.array(toks[j]), want[1:])
        if j == 0:
            np.testing.assert_array_equal(np.array(single[0]), want[1:])


def test_layer_range_matches_distributed_context():
    from jlama_amd import distributed as D
    assert D.layer_range(0, 8, 80) == (0, 10) and D.layer_range(7, 8, 80) == (70, 80)   # Llama-3-70B on 8 GPUs
    assert [D.layer_range(r, 4, 32) for r in range(4)] == [(0, 8), (8, 16), (16, 24), (24, 32)]
    with pytest.raises(ValueError):
        D.layer_range(0, 3, 32)


def _bench_worker(rank, world, port, cfg, q, fail_tp_on_rank=-1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      OMP_NUM_THREADS="2")
    import argparse
    import numpy as np
    import torch
    from jlama_amd import distributed as D, synthetic as S
    from oracle import oracle as O

    class OracleShardEngine(D.ShardEngine):
        def __init__(self, rank, world, n_sessions, max_ctx):
            ls, le = D.layer_range(rank, world, cfg["n_layers"])
            self.m = O.OracleModel(cfg, S.make_weights(cfg, seed=0), layer_range=(ls, le))
            self.s = [self.m.session() for _ in range(n_sessions)]
            self.last = [None] * n_sessions

        def forward_tokens(self, session, tokens, start_pos, x_out):
            x = self.s[session].forward(np.asarray(tokens, dtype=np.int32), start_pos)
            x_out.copy_(torch.from_numpy(x))
            self.last[session] = x[-1]

        def forward_x(self, session, x_in, n, start_pos, x_out):
            x = self.s[session].forward(None, start_pos, x=x_in.numpy()[:n])
            x_out.copy_(torch.from_numpy(x))
            self.last[session] = x[-1]

        def sample(self, session):
            return self.m.sample(self.last[session])[0]

        def stream(self, session):
            return None

        def stage_step(self, session, token, x_in, pos, x_out, token_out):
            if token is not None:
                x = self.s[session].forward(np.array([int(token[0])], dtype=np.int32), pos)
            else:
                x = self.s[session].forward(None, pos, x=x_in.numpy()[:1])
            if x_out is not None:
                x_out.copy_(torch.from_numpy(x))
            if token_out is not None:
                token_out[0] = int(self.m.sample(x[-1])[0])

    class OracleTPEngine(D.TPEngine):
        """head-split shard on the oracle: the rank-per-GPU tensor-parallel leg of the line (tp_rank_bench) on CPU"""
        def __init__(self, rank, world, max_ctx):
            if rank == fail_tp_on_rank:
                raise RuntimeError("this rank cannot build its shard")      # (the other rank is then alone in the leg's first collective)
            w = S.make_weights(cfg, seed=0)
            lc, off = D.tp_shard_config(cfg, rank, world)
            self.full = O.OracleModel(cfg, w) if rank == 0 else None
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

        def rows_max(self):
            return 4

        def set_rows(self, tokens, pos):
            self.x, self.pos = self.m.embed_rows(list(tokens)), pos

        def attn_rows(self, layer, partial):
            partial.copy_(torch.from_numpy(self.s.tp_attn(layer, self.x, self.pos)))

        def ffn_rows(self, layer, reduced, partial):
            self.x1 = self.x + reduced.numpy()
            partial.copy_(torch.from_numpy(self.s.tp_ffn(layer, self.x1)))

        def finish_layer_rows(self, reduced):
            self.x = self.x1 + reduced.numpy()

        def finish_rows(self):
            self.x = self.x[-1:]

    args = argparse.Namespace(gpus=world, steps=8, warmup=2, prompt=5, config="TINY")
    out = D.bench_pipeline(args, cfg, backend="gloo", engine_factory=OracleShardEngine, tp_engine_factory=OracleTPEngine)
    if rank == 0:
        q.put(out)
    if D.hard_exit["now"]:        # what bench.py does: a collective is stuck on some rank, nobody waits for a teardown
        import time
        time.sleep(1.0)           # (let the queue's feeder thread flush)
        os._exit(0)


@pytest.mark.parametrize("world", [2, 4])
def test_bench_pipeline_control_flow_on_cpu(world):
    """Every branch of bench.py's N>1 line that the 1-GPU test boxes cannot reach -- the rank-per-GPU pipeline bench with
    world > 1: prefill of N sessions, streamed-vs-host-synchronised id check, warm-up, the three timed legs, the host-side
    wait for the one-process leg on the rendezvous store, the JSON line -- run over gloo with the oracle as the shard engine."""
    import torch.multiprocessing as mp
    from jlama_amd import synthetic as S
    cfg = dict(S.TINY)
    cfg["n_layers"] = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, cfg, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the line is ONE batch-1 stream of exactly K steps (like the N = 1 line): value = the best single-stream rate over the two modes,
    # named in config.parallelism; N sessions in flight are reported beside it
    assert out["n_gpus"] == world and out["steps"] == 8 and out["value"] > 0 and out["scaling"] == "strong"
    tpl = out["tensor_parallel"]["rank_per_gpu"]
    if world == 2:     # TINY has 2 kv heads: two head-split shards exist, four do not
        assert tpl["shards"] == 2 and tpl["steps"] == 8 and tpl["single_stream_tokens_per_s"] > 0 and tpl["prompt_rows_batched"] == 4
        assert tpl["decode"].startswith("all-reduce rows")        # (the IPC token graphs need GPUs)
        assert out["tensor_parallel_tokens_per_s"] == tpl["single_stream_tokens_per_s"]
    else:
        assert "skipped" in tpl and out["tensor_parallel_tokens_per_s"] is None
    assert out["value"] == max(out["single_stream_tokens_per_s"], out["tensor_parallel_tokens_per_s"] or 0.0)
    assert ("tensor parallel" in out["config"]["parallelism"]) == ((out["tensor_parallel_tokens_per_s"] or 0.0) > out["single_stream_tokens_per_s"])
    assert out["aggregate_tokens_per_s"] > 0 and out["aggregate_steps"] == (8 // world) * world
    c = out["config"]
    assert c["sessions_in_flight"] == world and c["sessions_per_gpu"] == 1 and c["streamed_ids_equal_host_synchronised"] is True
    assert c["two_sessions_per_gpu"]["sessions_in_flight"] == 2 * world and c["two_sessions_per_gpu"]["aggregate_tokens_per_s"] > 0
    assert c["single_stream_tokens_per_s"] > 0 and c["host_synchronised_aggregate_tokens_per_s"] > 0
    assert out["one_process_pipeline"] == {"skipped": "no GPU (control-flow run)"}
    for key in ("metric", "unit", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "roofline"):
        assert key in out


def test_bench_line_survives_a_tensor_parallel_leg_that_wedges():
    """The rank-per-GPU tensor-parallel leg has never run on more than one GPU.  If it raises on one rank (the others are then stuck in
    a collective) the line must still come out: verdicts travel over the rendezvous store, no further collective is issued, `value`
    falls back to the layer-split single stream, the leg's key carries the error, and every rank leaves without a teardown."""
    import torch.multiprocessing as mp
    from jlama_amd import synthetic as S
    cfg = dict(S.TINY)
    cfg["n_layers"] = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    os.environ["JH_BENCH_TP_TIMEOUT"] = "6"
    try:
        procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, cfg, q, 1)) for r in range(2)]
        for p in procs:
            p.start()
        out = q.get(timeout=240)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        del os.environ["JH_BENCH_TP_TIMEOUT"]
    tpl = out["tensor_parallel"]["rank_per_gpu"]
    assert "error" in tpl and sorted(tpl["rank_verdicts"]) == ["error", "hung"], tpl
    assert out["tensor_parallel_tokens_per_s"] is None and out["value"] == out["single_stream_tokens_per_s"] > 0
    assert "layer-sharded" in out["config"]["parallelism"] and out["steps"] == 8


def test_bare_multi_gpu_bench_invocation_always_prints_one_json_line():
    """`python bench.py --gpus 2 --steps 8` without a launcher (no RANK): the one-process host is the path -- never a
    rendezvous, never a traceback.  Here (no GPU) it must end with exactly one JSON line carrying the structured error."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "8"], capture_output=True, text=True, cwd=root, env=env,
                       timeout=240)
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-400:] + r.stderr[-400:]
    out = json.loads(lines[0])
    assert "Traceback" not in r.stderr
    try:
        import torch
        visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:   # noqa: BLE001
        visible = 0
    if visible < 2:
        assert out["n_gpus_visible"] == visible and "error" in out and out["value"] is None and r.returncode != 0
    else:
        assert out["n_gpus"] == 2 and out["value"] > 0 and "single_stream_tokens_per_s" in out and out["cpu_baseline"]
