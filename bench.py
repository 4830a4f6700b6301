#!/usr/bin/env python3
"""bench.py -- decode tokens/s of Llama-3-8B JQ4 on MI355X (BASELINE.json metric), with roofline + CPU baseline.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under torch.distributed.run,
one rank per GPU.  A "step" = one decoded token (one pass of the hot path): W untimed warm-up decode steps on a
throw-away session, then a fresh session is prefilled with the 129-row prompt (128 synthetic ids + BOS,
AbstractModel.java:549-555) and EXACTLY K greedy decode steps are timed (clock starts after the first sampled
token, AbstractModel.java:589), bracketed by barrier + device synchronize; rank 0 prints ONE JSON line.

N=1: the whole model on one GPU.  N>1: DistributedContext layer split (DistributedContext.java:75-77) -- rank r owns
layers [r*L/N,(r+1)*L/N) and its KV pages; [1,E] F32 activations hop rank->rank with RCCL send/recv and the sampled
token returns to rank 0 (jlama_amd/distributed.py).

`--config` selects the other BASELINE.json configs (LLAMA32_1B, MISTRAL_7B: parity-test cases and profile lines, not the
driver's bench line).  The line carries `roofline` (dominant kernel, HIP events on the session's stream),
`cpu_baseline` (the reference's C SIMD GEMM driving the restated decode loop on the host cores) and
`parity_full_size` (same weights on GPU and CPU: strict-order ids/logits bit-for-bit, and the pairwise logit distances
GPU <-> Panama-order oracle <-> reference C GEMM).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6290.0  # what a float4 copy reaches on this part (MI355X_MICROARCH.md: 79 % of peak): roofline.frac_of_achievable
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense BF16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF headline includes 2:1 sparsity)
MFMA_I8_PEAK_TOPS = 5000.0       # dense I8 MFMA ~ 2x the BF16 rate (MI355X_MICROARCH.md: >= 3944 TOPS measured)
DOMINANT_KERNEL = ("gate+up GEMV of the order-free loop, fused RMSNorm+Q8 prologue, SiLU*up epilogue: gemv_t16_kernel<PRO_RMS_Q8,EPI_SILU_MUL> since round 4 "
                   "(the reference-order kernel is the faster one; jh_t16.h), gemv_i8q4_kernel<PRO_RMS_Q8,EPI_SILU_MUL> where its T16 copy is not resident")
DOMINANT_KERNEL_REF_ORDER = ("gemv_t16_kernel<PRO_RMS_Q8,EPI_SILU_MUL> (gate+up GEMV in the reference's summation order: block pair sums on "
                             "v_mfma_i32_16x16x32_i8 with a one-hot activation operand, cvt + fma chains on the VALU; jh_t16.h)")


def _profiled(config):
    """Counter / trace figures of the dominant kernel from the committed rocprofv3 passes (profiles/): HBM bytes per
    launch (FETCH_SIZE / WRITE_SIZE, gfx950-corrected) and the kernel-trace average duration.  Counters cannot be
    collected from inside this process, so they are reported ONLY when the profile was taken on a binary built from the
    sources this run uses (source hash recorded by tools/profile_round.sh); otherwise `traffic` is null and the stale
    figure is named as such."""
    from jlama_amd import _native as N
    import glob
    cands = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{config}_dominant_kernel.json"))):
        try:
            cands.append((path, json.load(open(path))))
        except (OSError, ValueError):
            pass
    if not cands:
        return None, None, None
    want = N.source_hash()
    match = [c for c in cands if c[1].get("source_hash") == want]
    path, d = (match or cands)[-1]          # the newest round's record of THIS build, else the newest record (reported as stale)
    fresh = d.get("source_hash") == want
    ro = d.get("reference_order") or {}
    return (d.get("traffic_bytes_per_launch") if fresh else None), (d.get("us_per_launch_rocprof") if fresh else None), \
        {"file": os.path.relpath(path, ROOT), "source_hash": d.get("source_hash"), "matches_this_build": fresh,
         "traffic_bytes_per_launch": d.get("traffic_bytes_per_launch"), "us_per_launch_rocprof": d.get("us_per_launch_rocprof"),
         "reference_order": {"kernel": ro.get("kernel"), "traffic_bytes_per_launch": ro.get("traffic_bytes_per_launch"),
                             "us_per_launch_rocprof": ro.get("us_per_launch_rocprof")} if ro else None}


def _quiesce(torch, kick=None):
    """Opening bracket of a timed region: torch.cuda.synchronize(), repeated until it returns at once.  After the warm-up session's
    graph replays the HIP runtime sometimes keeps a default-stream synchronize blocked for ~40-70 ms of WALL time although the
    device is idle (seen with K = 20: torch.cuda.synchronize() behind a 28 ms decode took 39 ms more; never with K = 256, where
    that time has long passed) -- a closing synchronize must measure the K steps, not that.  `kick` (a few untimed steps of the
    same loop) runs after the pauses, so that the timed region does not start on clocks that dropped while this waited."""
    if kick:       # first: every graph the timed region replays gets its FIRST launch here (round 5: the 4-tokens-per-launch graphs were
        kick()     # first launched by the kick below, and the runtime's ~40 ms of deferred work after a first launch fell into the timed region)
    for _ in range(50):
        time.sleep(0.02)
        t = time.perf_counter()
        torch.cuda.synchronize()
        torch.cuda.current_stream().synchronize()
        if time.perf_counter() - t < 1e-3:
            break
    if kick:
        kick()
    torch.cuda.synchronize()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--config", default="LLAMA3_8B")
    ap.add_argument("--prompt", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-strict", action="store_true")     # skip the reference-order leg (strict_tokens_per_s)
    # `value` / `roofline` = the REFERENCE-ORDER path (ids and logits bit-identical to the Panama-order oracle: the path that meets
    # BASELINE north_star's parity bar); the order-free kernels are reported beside it as fast_tokens_per_s.  --fast-order prints the
    # line the other way round (value = order-free kernels, strict_tokens_per_s = reference order), as rounds 1-3 did.
    ap.add_argument("--fast-order", action="store_true")
    ap.add_argument("--reference-order", action="store_true")   # (default now; kept so that older command lines still parse)
    ap.add_argument("--cpu-steps", type=int, default=256)   # ~12 s of CPU work at the reference path's ~21 tok/s
    ap.add_argument("--parity-steps", type=int, default=256)   # free-running strict-order ids compared with the oracle
    ap.add_argument("--probe-iters", type=int, default=3)
    ap.add_argument("--no-tier1-host", action="store_true")   # skip the host-as-is leg (tier1_host_tokens_per_s)
    ap.add_argument("--tier1-steps", type=int, default=8)     # decode steps of that leg (each ~0.1-0.3 s on the 8B model)
    return ap.parse_args()


TF_STEPS = 32
METRIC_STEPS = 256   # BASELINE.json: "128-tok prefill + 256-tok decode"


def cpu_baseline(cfg, host_w, n_prompt, n_decode):
    """Reference native-SIMD GEMM (oracle/_ref = vector_simd.c compiled as-is) + restated Java ops, threaded like the
    reference: T = max(2, availableProcessors/2) (PhysicalCoreExecutor.java:27), where availableProcessors honours the
    container's CPU quota as the JVM does.  Bounded sample of the same workload."""
    from oracle import oracle as O
    from jlama_amd import synthetic as S
    avail = O.available_cpus()
    T = max(2, avail // 2)
    m = O.OracleModel(cfg, host_w)
    kind = "port"
    if O.ref_lib() is not None and cfg["weight_dtype"] == O.DT_Q4:
        m.use_reference_gemm(T)
        kind = "reference"
    prompt = S.prompt_tokens(cfg, n=n_prompt - 1, seed=1234)
    sess = m.session()
    # prompt rows one at a time (batchForwardSlow, AbstractModel.java:282-290): the reference's C tiler leaves
    # output corners uncomputed for M > 5 (tests/test_oracle.py::test_reference_library_tiler_...), M = 1 is exact
    for i, t in enumerate(prompt):
        x = sess.forward([t], i)
    tok, _ = m.sample(x[-1])
    t0 = time.perf_counter()
    for i in range(n_decode):
        x = sess.forward([tok], prompt.size + i)
        tok, _ = m.sample(x[-1])
    dt = time.perf_counter() - t0
    return {"value": round(n_decode / dt, 3), "unit": "tokens/s", "cores": T, "kind": kind,
            "sample": f"{n_decode} greedy decode steps after a {n_prompt}-row prompt, full {cfg['n_layers']}-layer model; "
                      f"{'reference C SIMD GEMM (vector_simd.c, AVX-512 VNNI kernels) + ' if kind == 'reference' else ''}restated "
                      f"Java ops; T={T} threads = max(2, available/2), {avail} CPUs available to the container of {os.cpu_count()} on the host"}


def _dist(a, b):
    """max and mean-over-steps of max |a - b| for two lists of logit vectors."""
    d = [float(np.abs(x - y).max()) for x, y in zip(a, b)]
    return {"max": round(max(d), 6), "mean_of_max": round(float(np.mean(d)), 6)}


def full_size_parity(cfg, model, host_w, n_prompt=8, n_free=256, n_tf=TF_STEPS):
    """End-to-end parity at FULL size on identical weights (GPU resident copy vs host copy).

    A. Panama-order oracle (the restated reference provider) runs the prompt and n_free greedy steps free.
    B. The GPU in STRICT ORDER does the same: ids must be IDENTICAL and logits equal to ~0 -- this is the measurement
       that the fast kernels' residual is summation order only.
    C. Teacher-forced on A's ids for n_tf steps, three logit streams are compared pairwise: the GPU's fast kernels, the
       Panama-order oracle, and the oracle driven by the reference's own compiled C GEMM (oracle/_ref) -- i.e. the
       reference's two CPU providers.  Their mutual distance is the noise floor any second implementation sits on."""
    from oracle import oracle as O
    from jlama_amd import synthetic as S
    prompt = S.prompt_tokens(cfg, n=n_prompt - 1, seed=1234)
    out = {"prompt_rows": int(prompt.size), "free_steps": n_free, "teacher_forced_steps": n_tf}
    # ---- A: Panama-order oracle, free-running
    om = O.OracleModel(cfg, host_w)
    osess = om.session()
    t0 = time.perf_counter()
    x = osess.forward(prompt, 0)
    tok, lg = om.sample(x[-1])
    ids_o, logits_o = [tok], [lg]
    for i in range(n_free):
        x = osess.forward([tok], prompt.size + i)
        tok, lg = om.sample(x[-1])
        ids_o.append(tok)
        if len(logits_o) <= n_tf:
            logits_o.append(lg)
    last_logits_o = lg
    out["oracle_panama_order_s"] = round(time.perf_counter() - t0, 1)
    ids_o = np.array(ids_o, dtype=np.int32)
    # ---- B: GPU in reference order (JQ4: jh_t16.h / jh_p16.h; BF16: jh_bf16r.h), free-running
    ss = model.session(prompt.size + n_free + 8)
    ss.set_strict(True)
    ss.batch_forward(prompt, 0)
    tok_s, lg_s = ss.sample(0.0, 0.5, want_logits=True)
    logits_s, ids_s = [lg_s], [tok_s]
    tok = tok_s
    for i in range(min(n_tf, n_free)):                    # host loop while logits are being collected ...
        tok = ss.decode_step(tok, prompt.size + i)
        ids_s.append(tok)
        logits_s.append(ss.logits())
    if n_free > n_tf:                                     # ... then the on-device loop
        ids_s.extend(int(t) for t in ss.decode_n(tok, prompt.size + n_tf, n_free - n_tf))
    last_logits_s = ss.logits()
    ids_s = np.array(ids_s, dtype=np.int32)
    ss.close()
    out["strict_order"] = {"ids_equal": int((ids_s == ids_o).cumprod().sum()), "n_ids": int(ids_o.size),
                           "logits_vs_panama_oracle": _dist(logits_s, logits_o[:len(logits_s)]),
                           "last_step_logits_max_abs_diff": float(np.abs(last_logits_s - last_logits_o).max())}
    if cfg["weight_dtype"] != O.DT_Q4:
        return _full_size_parity_dense(cfg, model, prompt, out, ids_o, logits_o, n_free, n_tf), ids_o
    # ---- C: teacher-forced on A's ids: GPU fast kernels, and the oracle on the reference's compiled C GEMM
    fs = model.session(prompt.size + n_tf + 8)
    fs.batch_forward(prompt, 0)
    tok_f, lg_f = fs.sample(0.0, 0.5, want_logits=True)
    logits_f, ids_f = [lg_f], [tok_f]
    for i in range(n_tf):
        ids_f.append(fs.decode_step(int(ids_o[i]), prompt.size + i))
        logits_f.append(fs.logits())
    fs.close()
    logits_o_tf = logits_o[:n_tf + 1]
    pair = {"gpu_fast__panama_oracle": _dist(logits_f, logits_o_tf)}
    if O.ref_lib() is not None and cfg["weight_dtype"] == O.DT_Q4:
        rm = O.OracleModel(cfg, host_w)
        rm.use_reference_gemm(max(2, O.available_cpus() // 2))
        rs = rm.session()
        for i, t in enumerate(prompt):
            x = rs.forward([t], i)
        _, lg = rm.sample(x[-1])
        logits_r = [lg]
        for i in range(n_tf):
            x = rs.forward([int(ids_o[i])], prompt.size + i)
            _, lg = rm.sample(x[-1])
            logits_r.append(lg)
        pair["gpu_fast__reference_c_gemm"] = _dist(logits_f, logits_r)
        pair["panama_oracle__reference_c_gemm"] = _dist(logits_o_tf, logits_r)
    out["teacher_forced_pairwise_logit_distance"] = pair
    out["logit_scale"] = float(np.abs(logits_o[0]).max())
    # argmax agreement of the fast kernels against a FIXED margin: steps whose oracle top-2 margin exceeds 0.25
    # (~3% of the logit scale, above every pairwise distance measured so far)
    decided = agree = 0
    for i, want in enumerate(logits_o_tf):
        top2 = np.partition(want, -2)[-2:]
        if top2[1] - top2[0] > 0.25:
            decided += 1
            agree += int(ids_f[i] == int(np.argmax(want)))
    out["fast_argmax_vs_oracle_at_margin_0.25"] = {"decided_steps": decided, "agree": agree, "of": len(logits_o_tf)}
    return out, ids_o


def _full_size_parity_dense(cfg, model, prompt, out, ids_o, logits_o, n_free, n_tf):
    """BF16 model (config 4): one kernel set, no Q8 step function on the path.  Teacher-forced on the oracle's ids the logits
    must sit within 1e-3 of the logit scale (BASELINE north_star's F32 tolerance; what remains is F32 summation order plus the
    rare BF16 rounding flip of an activation), and the free-running greedy ids are compared token for token."""
    fs = model.session(prompt.size + max(n_tf, n_free) + 8)
    fs.batch_forward(prompt, 0)
    tok_f, lg_f = fs.sample(0.0, 0.5, want_logits=True)
    logits_f, ids_tf = [lg_f], [tok_f]
    for i in range(n_tf):
        ids_tf.append(fs.decode_step(int(ids_o[i]), prompt.size + i))
        logits_f.append(fs.logits())
    fs.close()
    logits_o_tf = logits_o[:n_tf + 1]
    scale = float(np.abs(logits_o[0]).max())
    d = _dist(logits_f, logits_o_tf)
    out["teacher_forced_logits_vs_oracle"] = dict(d, max_rel_to_logit_scale=round(d["max"] / scale, 7))
    out["logit_scale"] = scale
    out["teacher_forced_argmax_equal"] = int(sum(int(a == int(np.argmax(w))) for a, w in zip(ids_tf, logits_o_tf)))
    out["teacher_forced_steps_compared"] = len(logits_o_tf)
    # free-running greedy ids: batched (MFMA) prefill + the on-device loop, exactly what the timed run executes
    gs = model.session(prompt.size + n_free + 8)
    gs.batch_forward(prompt, 0)
    g0 = gs.sample()
    ids_g = np.concatenate([[g0], gs.decode_n(g0, prompt.size, n_free)]).astype(np.int32)
    gs.close()
    out["free_running_ids_equal_prefix"] = int((ids_g == ids_o[:ids_g.size]).cumprod().sum())
    out["free_running_ids_compared"] = int(ids_g.size)
    margins = []
    for w in logits_o_tf:
        top2 = np.partition(w, -2)[-2:]
        margins.append(float(top2[1] - top2[0]))
    out["oracle_min_top2_margin_teacher_forced"] = round(min(margins), 6)
    return out


def run_single(args, cfg):
    import torch
    from jlama_amd import _native as N, synthetic as S, synthetic_torch as ST
    from jlama_amd.model import HipLlamaModel
    torch.cuda.set_device(0)
    N.init(0)
    t0 = time.time()
    w = ST.make_weights(cfg, seed=0, device="cuda")
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    model = HipLlamaModel(cfg, w)
    prompt = S.prompt_tokens(cfg, n=args.prompt, seed=1234)
    want_metric_shape = (not args.no_strict and args.steps != METRIC_STEPS and
                         prompt.size + METRIC_STEPS + 8 + 2 * cfg["n_kv_heads"] <= cfg["context_length"])
    max_ctx = prompt.size + max(args.steps, args.warmup, METRIC_STEPS if want_metric_shape else 0) + 8
    # warm-up on a throw-away session (builds the hipGraph, faults the weights in)
    ws = model.session(max_ctx)
    ws.batch_forward(prompt[:8], 0)
    first = ws.sample()
    if args.warmup > 0:
        ws.decode_n(first, 8, args.warmup)
    ws.close()
    # measured run
    s = model.session(max_ctx)
    tp0 = time.perf_counter()
    s.batch_forward(prompt, 0)
    first = s.sample()
    prompt_cold_ms = (time.perf_counter() - tp0) * 1e3      # includes this session's one-time prefill buffer allocation
    tp0 = time.perf_counter()
    s.batch_forward(prompt, 0)                                # same rows again (rewrites the same KV rows): steady state
    first = s.sample()
    prompt_ms = (time.perf_counter() - tp0) * 1e3
    # untimed: capture every decode-graph variant the timed positions will replay (short / medium / long context attention);
    # the timed run rewrites these rows
    last_pos = prompt.size + args.steps - 1
    for p0 in sorted({prompt.size, last_pos} | {b for b in (512, 513, 2048, 2049) if prompt.size <= b <= last_pos}):
        s.decode_n(first, p0, 1)
    _quiesce(torch, lambda: s.decode_n(first, prompt.size, min(4, args.steps)))
    s.synchronize()
    t0 = time.perf_counter()
    s.decode_n_async(first, prompt.size, args.steps)
    t_a = time.perf_counter()
    toks = s.decode_wait(args.steps)
    t_w = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if os.environ.get("JH_BENCH_DEBUG"):
        print(f"[bench] queue {1e3 * (t_a - t0):.2f} ms, wait {1e3 * (t_w - t_a):.2f} ms, torch sync {1e3 * (t0 + dt - t_w):.2f} ms", file=sys.stderr)
    assert toks.size == args.steps
    ev_ms, kernels = s.decode_stats()
    tps = args.steps / dt
    # roofline of the dominant kernel (gate/up GEMV: 54% of the weight bytes), HIP events on the session's stream -- probed right
    # after the timed decode, before the reference-order leg (a VALU-heavy prompt in front of it costs the probe ~0.6 us of clocks)
    probe = {}
    names = ["qkv", "attention", "o_proj", "gate_up", "down"]
    is_q4 = cfg["weight_dtype"] == N.DT_Q4
    if is_q4:
        for i, nm in enumerate(names):
            ms, b = s.kernel_bench(i, args.probe_iters)
            probe[nm] = {"us": round(ms * 1e3, 3), "bytes": b, "GBps": round(b / (ms * 1e-3) / 1e9, 1)}
        dom = probe["gate_up"]
    # ---- the same K steps in REFERENCE ORDER (jh_p16.h: every float accumulation in the Panama provider's order; ids and logits
    # bit-identical to the oracle, see parity_full_size): timed exactly like `value`, in this same process
    strict = None
    if not args.no_strict:
        ss = model.session(max_ctx)
        ss.set_strict(True)                    # the whole leg in reference order: prompt (M-row p16 GEMMs), sampling, decode
        ss.batch_forward(prompt, 0)
        ss.sample()
        tp0 = time.perf_counter()
        ss.batch_forward(prompt, 0)            # steady state, timed like prefill_ms of the fast leg
        sfirst = ss.sample()
        sprompt_ms = (time.perf_counter() - tp0) * 1e3
        for p0 in sorted({prompt.size, last_pos}):
            ss.decode_n(sfirst, p0, 1)         # graph capture, untimed
        _quiesce(torch, lambda: ss.decode_n(sfirst, prompt.size, min(4, args.steps)))
        ss.synchronize()
        t0 = time.perf_counter()
        ss.decode_n_async(sfirst, prompt.size, args.steps)
        stoks = ss.decode_wait(args.steps)
        torch.cuda.synchronize()
        sdt = time.perf_counter() - t0
        assert stoks.size == args.steps
        sev_ms, skernels = ss.decode_stats()
        # the metric's own shape (BASELINE: 256 decode steps behind the 129-row prompt) whatever K the caller asked for: the same
        # session, the same bracket, positions prompt.size ... prompt.size + 255 -- the driver's K = 20 covers positions 129..148 only
        metric_shape = None
        if want_metric_shape:
            ss.decode_n(sfirst, prompt.size + METRIC_STEPS - 1, 1)     # graph variant of the last position, untimed
            _quiesce(torch, lambda: ss.decode_n(sfirst, prompt.size, 4))
            ss.synchronize()
            t0 = time.perf_counter()
            ss.decode_n_async(sfirst, prompt.size, METRIC_STEPS)
            mtoks = ss.decode_wait(METRIC_STEPS)
            torch.cuda.synchronize()
            mdt = time.perf_counter() - t0
            assert mtoks.size == METRIC_STEPS and np.array_equal(mtoks[:args.steps], stoks[:METRIC_STEPS])
            metric_shape = {"tokens_per_s": round(METRIC_STEPS / mdt, 2), "ms_per_step": round(mdt / METRIC_STEPS * 1e3, 4), "steps": METRIC_STEPS}
        sprobe = {}
        for i, nm in ((0, "qkv"), (1, "attention"), (2, "o_proj"), (3, "gate_up"), (4, "down"), (9, "lm_head")) if is_q4 else ():
            ms, b = ss.kernel_bench(i, args.probe_iters)
            sprobe[nm] = {"us": round(ms * 1e3, 3), "bytes": b, "GBps": round(b / (ms * 1e-3) / 1e9, 1)}
        ss.close()
        strict = {"tokens_per_s": round(args.steps / sdt, 2), "ms_per_step": round(sdt / args.steps * 1e3, 4), "prefill_ms": round(sprompt_ms, 2),
                  "metric_shape": metric_shape,
                  "event_ms_per_token": round(sev_ms, 4), "kernels_per_token": skernels, "kernels": sprobe,
                  "note": "reference-order kernels (" + ("jh_t16.h, jh_p16.h" if is_q4 else "jh_bf16r.h, jh_p16.h") + "): bit-identical ids and logits vs the Panama-order oracle "
                          "(parity_full_size.strict_order), same K steps, same bracket as `value`"}
    wbytes = S.weight_bytes(cfg)
    kvb = S.kv_bytes_per_position(cfg)
    mean_pos = prompt.size + (args.steps - 1) / 2.0
    bytes_per_token = wbytes + kvb * (mean_pos + 1) + kvb
    traffic, us_rocprof, prof = _profiled(args.config)
    # prefill as a GEMM workload: 2 * rows * (projection weights) flops; against the dense MFMA peak of the dtype this is
    # the "MFMA utilisation" of BASELINE configs[3] (BF16); SURVEY 8d: at M = 129 the HBM roofline caps it at ~41 %
    per_layer = sum(r * c for r, c in S.layer_shapes(cfg).values())
    prefill_flops = 2.0 * prompt.size * cfg["n_layers"] * per_layer
    prefill_tflops = prefill_flops / (prompt_ms * 1e-3) / 1e12
    mfma_peak = MFMA_I8_PEAK_TOPS if is_q4 else MFMA_BF16_PEAK_TFLOPS
    bytes_per_weight = 0.625 if is_q4 else 2.0
    out = {
        "metric": "decode tokens/sec Llama-3-8B JQ4, 128-tok prompt" if args.config == "LLAMA3_8B" else f"decode tokens/sec {args.config}",
        "value": round(tps, 2), "unit": "tokens/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "i8xq4->f32" if is_q4 else "bf16xbf16->f32", "data": "synthetic",
        "config": {"workload": f"{args.config} {'JQ4 (Q4 weights, I8 activations' if is_q4 else 'BF16 (BF16 weights and activations'}, F32 paged KV), {prompt.size}-row prefill + "
                               f"{args.steps} greedy decode steps, batch 1", "parallelism": "1 GPU",
                   "kernels_per_token": kernels, "prefill_ms": round(prompt_ms, 2), "prefill_cold_ms": round(prompt_cold_ms, 2),
                   "prefill_tokens_per_s": round(prompt.size / prompt_ms * 1e3, 1)},
        "roofline": ({"bound": "hbm", "kernel": DOMINANT_KERNEL,
                      "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(dom["GBps"] / HBM_PEAK_GBS, 4), "frac_of_achievable": round(dom["GBps"] / HBM_ACHIEVABLE_GBS, 4), "traffic": traffic,
                      "bytes_per_launch": dom["bytes"], "us_per_launch": dom["us"], "us_per_launch_rocprof": us_rocprof,
                      "profile": prof} if is_q4 else
                     {"bound": "hbm", "kernel": "whole decode step (per-kernel probe is JQ4-only)",
                      "achieved": round(bytes_per_token * tps / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(bytes_per_token * tps / 1e9 / HBM_PEAK_GBS, 4),
                      "frac_of_achievable": round(bytes_per_token * tps / 1e9 / HBM_ACHIEVABLE_GBS, 4), "traffic": None}),
        "token_roofline": {"bytes_per_token": int(bytes_per_token), "achieved_GBps": round(bytes_per_token * tps / 1e9, 1),
                           "frac_of_8TBps": round(bytes_per_token * tps / 1e9 / HBM_PEAK_GBS, 4),
                           "frac_of_achievable": round(bytes_per_token * tps / 1e9 / HBM_ACHIEVABLE_GBS, 4),
                           "event_ms_per_token": round(ev_ms, 4)},
        "prefill_mfma": {"rows": int(prompt.size), "flops": prefill_flops, "achieved_TFLOPs": round(prefill_tflops, 1),
                         "peak_TFLOPs": mfma_peak, "frac": round(prefill_tflops / mfma_peak, 4),
                         "hbm_bound_frac": round(min(1.0, 2.0 * prompt.size * HBM_PEAK_GBS * 1e9 / (bytes_per_weight * mfma_peak * 1e12)), 3),
                         "note": "projection GEMMs of the prefill (2*rows*weights flops; attention excluded) over the whole prefill "
                                 "time, against the dense " + ("BF16" if not is_q4 else "I8") + " MFMA peak; hbm_bound_frac = the share "
                                 "of that peak the weight stream allows at this row count (SURVEY.md 8d)"},
        "kernels": probe,
        "weights_gen_s": round(gen_s, 1),
    }
    if strict:
        # the dominant kernel of that path (gemv_i8q4_p16_kernel, gate|up) against the same roofline, counter traffic hash-gated as above
        ro = (prof or {}).get("reference_order") or {}
        fresh = bool(prof and prof.get("matches_this_build"))
        if is_q4:
            gu = strict["kernels"]["gate_up"]
            strict["roofline"] = {"bound": "hbm", "kernel": DOMINANT_KERNEL_REF_ORDER,
                                  "achieved": gu["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gu["GBps"] / HBM_PEAK_GBS, 4),
                                  "frac_of_achievable": round(gu["GBps"] / HBM_ACHIEVABLE_GBS, 4),
                                  "traffic": ro.get("traffic_bytes_per_launch") if fresh else None, "bytes_per_launch": gu["bytes"], "us_per_launch": gu["us"],
                                  "us_per_launch_rocprof": ro.get("us_per_launch_rocprof") if fresh else None}
        else:   # BF16: no per-kernel probe -- the whole reference-order decode step against the token's bytes
            sg = bytes_per_token * strict["tokens_per_s"] / 1e9
            strict["roofline"] = {"bound": "hbm", "kernel": "whole decode step, reference order (gemv_bf16r_kernel x 4 + attention per layer)",
                                  "achieved": round(sg, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(sg / HBM_PEAK_GBS, 4),
                                  "frac_of_achievable": round(sg / HBM_ACHIEVABLE_GBS, 4), "traffic": None}
        out["strict_tokens_per_s"] = strict["tokens_per_s"]
        out["strict_order"] = strict
        # reference-order rate over the metric's own 256 decode positions (timed in this run; = value when --steps 256)
        ms_shape = strict.pop("metric_shape")
        if ms_shape or args.steps == METRIC_STEPS:
            out["value_at_metric_shape"] = dict(ms_shape or {"tokens_per_s": strict["tokens_per_s"], "ms_per_step": strict["ms_per_step"], "steps": METRIC_STEPS},
                                                order="reference", prompt_rows=int(prompt.size))
            bpt = wbytes + kvb * (prompt.size + (METRIC_STEPS - 1) / 2.0 + 1) + kvb
            out["value_at_metric_shape"]["token_roofline_frac_of_8TBps"] = round(bpt * out["value_at_metric_shape"]["tokens_per_s"] / 1e9 / HBM_PEAK_GBS, 4)
        if not args.fast_order:   # the path with bit-exact ids is the headline of this line
            out["fast_tokens_per_s"] = out["value"]
            out["fast_roofline"] = out["roofline"]
            out["value"], out["ms_per_step"] = strict["tokens_per_s"], strict["ms_per_step"]
            out["roofline"] = strict["roofline"]
            out["config"]["kernels_per_token"] = strict["kernels_per_token"]
            out["config"]["prefill_ms_fast_kernels"] = out["config"]["prefill_ms"]
            out["config"]["prefill_tokens_per_s_fast_kernels"] = out["config"]["prefill_tokens_per_s"]
            out["config"]["prefill_ms"] = strict["prefill_ms"]
            out["config"]["prefill_tokens_per_s"] = round(prompt.size / strict["prefill_ms"] * 1e3, 1)
            del out["config"]["prefill_cold_ms"]
            # the reference-order prompt GEMMs run on the F16 MFMA (gemm_t16_kernel): same projection flops over ITS prefill time
            pm = out["prefill_mfma"]
            out["prefill_mfma_fast_kernels"] = dict(pm)
            so_tflops = prefill_flops / (strict["prefill_ms"] * 1e-3) / 1e12
            pm.update({"achieved_TFLOPs": round(so_tflops, 1), "peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS, "frac": round(so_tflops / MFMA_BF16_PEAK_TFLOPS, 4),
                       "hbm_bound_frac": round(min(1.0, 2.0 * prompt.size * HBM_PEAK_GBS * 1e9 / (bytes_per_weight * MFMA_BF16_PEAK_TFLOPS * 1e12)), 3),
                       "note": ("reference-order prompt: projection GEMMs (2*rows*weights flops) over the whole prefill time against the dense F16 MFMA "
                                "peak; every (row, weight row, block) still pays 16 ordered fmas on the VALU (jh_t16.h), which bounds it far below") if is_q4 else
                               ("reference-order prompt of a BF16 model: GemmerBF16's chains are one VALU fma per (prompt row, weight) -- no matrix-core "
                                "formulation keeps the rounding order (jh_bf16r.h); the MFMA figure of this config is prefill_mfma_fast_kernels")})
            out["config"]["workload"] += " -- REFERENCE ORDER (every float accumulation in the Panama provider's order; ids and logits bit-identical)"
            tr = out["token_roofline"]
            tr["achieved_GBps"] = round(tr["bytes_per_token"] * out["value"] / 1e9, 1)
            tr["frac_of_8TBps"] = round(tr["achieved_GBps"] / HBM_PEAK_GBS, 4)
            tr["frac_of_achievable"] = round(tr["achieved_GBps"] / HBM_ACHIEVABLE_GBS, 4)
            tr["event_ms_per_token"] = strict["event_ms_per_token"]
    host_w = None
    if not args.no_cpu_baseline:
        host_w = ST.to_host(w)
        # the metric's own workload: the same prompt.size-row prompt (one row at a time on the CPU, ~6 s for 129 rows of the 8B
        # model), then decode positions prompt.size ... prompt.size + cpu_steps - 1
        out["cpu_baseline"] = cpu_baseline(cfg, host_w, int(prompt.size), args.cpu_steps)
    if not args.no_parity and not is_q4:
        host_w = host_w or ST.to_host(w)
        # on the metric's own prompt (the rows the timed run prefills through the batched reference-order kernels); the oracle's
        # BF16 GEMM takes the whole chunk per weight pass (AVX2 bodies, bit-equal to the scalar text), 16 teacher-forced steps
        par, _ = full_size_parity(cfg, model, host_w, int(prompt.size), args.parity_steps, 16)
        out["parity_full_size"] = par
    if not args.no_parity and is_q4:
        host_w = host_w or ST.to_host(w)
        # on the metric's own prompt: the same prompt.size rows the timed run prefills (batched reference-order prefill)
        par, ids_o = full_size_parity(cfg, model, host_w, int(prompt.size), args.parity_steps, TF_STEPS)
        # free-running ids of the FAST kernels vs the oracle's (diverge at the first near-tie, as any two float summation
        # orders do on random weights; the strict-order run above does not)
        pp = S.prompt_tokens(cfg, n=int(prompt.size) - 1, seed=1234)
        ps = model.session(pp.size + args.parity_steps + 8)
        ps.batch_forward(pp, 0)
        gfirst = ps.sample()
        ids_fast = np.concatenate([[gfirst], ps.decode_n(gfirst, pp.size, args.parity_steps)])
        par["fast_free_running_ids_equal_prefix"] = int((ids_fast == ids_o).cumprod().sum())
        out["parity_full_size"] = par
    if not args.no_tier1_host:
        host_w = host_w or ST.to_host(w)
        ids_ref = ids_o if (not args.no_parity and is_q4) else None
        out["tier1_host"] = tier1_host_leg(cfg, host_w, prompt, args.tier1_steps, ids_ref)
        out["tier1_host_tokens_per_s"] = out["tier1_host"]["tokens_per_s"]
    return out, toks


def tier1_host_leg(cfg, host_w, prompt, n_steps, ids_ref):
    """The cost of "the Java host stays as is" (BASELINE north_star), MEASURED: the reference's host restated above the C ABI
    (libjlamahost.so, jlama_amd/csrc/host_mirror.cpp = AbstractModel / TransformerBlock / CausalSelfAttention / MLPBlock calling only
    the provider entry points on host buffers) generates on the metric's own prompt, in reference order, with the provider's default
    split (GEMMs on the device, element-wise methods on the host delegate: HipTensorOperations.java:45-57) and the reference's pfor
    width T = max(2, available/2).  Every call is a PCIe round trip: this is Tier 1, never `value`."""
    from jlama_amd import _native as N
    from jlama_amd.host_mirror import HostAsIsModel
    from oracle import oracle as O   # (available_cpus only: the JVM-like view of the container's CPU quota)
    T = max(2, O.available_cpus() // 2)
    N.set_option("JH_STRICT_ORDER", 1)
    try:
        hm = HostAsIsModel(cfg, host_w, elementwise_on_device=False, threads=T)
        res = hm.generate(prompt, n_steps + 1)      # n_steps decode steps behind the first sampled token (AbstractModel.java:589)
        hm.close()
    finally:
        N.clear_options()
    out = {"tokens_per_s": round(n_steps / (res["decode_ms"] * 1e-3), 3), "ms_per_step": round(res["decode_ms"] / n_steps, 2),
           "prompt_ms": round(res["prompt_ms"], 1), "prompt_rows": int(prompt.size), "steps": n_steps, "threads": T,
           "provider_calls_per_token": None, "order": "reference (JH_STRICT_ORDER=1: gemm_reford_kernel)",
           "elementwise": "host delegate (provider default)",
           "note": "libjlamahost.so: the reference's host restated in C++ above the Tier-1 C ABI; every provider call ships host buffers over PCIe and "
                   "synchronises (per layer and token: 8 weight GEMVs + heads x KV pages score GEMMs)"}
    if ids_ref is not None:
        k = min(len(ids_ref), res["tokens"].size)
        out["ids_equal_to_oracle"] = int((res["tokens"][:k] == ids_ref[:k]).cumprod().sum())
        out["ids_compared"] = int(k)
    out["provider_calls_total"] = res["provider_calls"]
    del out["provider_calls_per_token"]
    return out


def run_one_process(args, cfg):
    """`--gpus N` WITHOUT a launcher (no RANK in the environment): the one-process N-device host BASELINE's north_star names
    (jh_pipeline_*: stage k on HIP device k, activations hop by stream-ordered peer copies over xGMI).  Always ends with
    exactly one JSON line: the result, or {"error": ..., "n_gpus_visible": k} when the node has fewer devices."""
    import torch
    from jlama_amd import distributed as D, synthetic as S
    n = args.gpus
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    base = {"metric": "decode tokens/sec Llama-3-8B JQ4, 128-tok prompt" if args.config == "LLAMA3_8B" else f"decode tokens/sec {args.config}",
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup}
    if visible < n:
        return dict(base, error=f"--gpus {n} but only {visible} HIP device(s) visible", n_gpus_visible=visible, value=None), 1
    if cfg["n_layers"] % n:
        return dict(base, error=f"{cfg['n_layers']} layers do not split evenly over {n} stages", n_gpus_visible=visible, value=None), 1
    try:
        r = D.one_process_pipeline_bench(args.config, n, args.steps, args.warmup, args.prompt, probe_iters=args.probe_iters, strict=not args.fast_order)
        roof, cpu = D.multi_gpu_extras(args, cfg, r["gate_up_probe"], 0)
    except Exception as e:   # noqa: BLE001 -- the contract is one JSON line, never a traceback
        return dict(base, error=repr(e)[:600], n_gpus_visible=visible, value=None), 1
    is_q4 = cfg["weight_dtype"] == 3
    tokens = r["steps_per_session"] * r["sessions"]
    # `value`: ONE batch-1 stream of K tokens (like the N = 1 line), in whichever mode serves it faster -- layer split (a stream
    # passes through all stages in sequence) or the head-split group (its single-stream rate can grow with N)
    tpl = r.get("tensor_parallel") or {}
    tp_single = tpl.get("single_stream_tokens_per_s") or 0.0
    best_tp = n > 1 and tp_single > r["single_stream_tokens_per_s"]
    value = tp_single if best_tp else r["single_stream_tokens_per_s"]
    ms_per_step = 1e3 / value
    wbytes, kvb = S.weight_bytes(cfg), S.kv_bytes_per_position(cfg)
    bytes_per_token = wbytes + kvb * (r["prompt_rows"] + (args.steps - 1) / 2.0 + 2)
    out = dict(base, value=value, unit="tokens/s", steps=args.steps, ms_per_step=round(ms_per_step, 4),
               higher_is_better=True, scaling="strong",
               scaling_detail=(f"one batch-1 stream of {args.steps} tokens whatever N (total work fixed): value = the best single-stream rate over "
                               f"the two modes (layer split {r['single_stream_tokens_per_s']}, tensor parallel {tp_single or None} tok/s); the throughput "
                               f"of {n} independent sessions in flight is aggregate_tokens_per_s") if n > 1 else "1 GPU, batch 1",
               aggregate_tokens_per_s=r["aggregate_tokens_per_s"], aggregate_steps=tokens,
               single_stream_tokens_per_s=r["single_stream_tokens_per_s"], tensor_parallel_tokens_per_s=tp_single or None,
               order=r.get("order"), vs_baseline=None,
               dtype="i8xq4->f32" if is_q4 else "bf16xbf16->f32", data="synthetic",
               config={"workload": f"{args.config}, {r['prompt_rows']}-row prefill + {r['steps_per_session']} greedy decode steps x "
                                   f"{r['sessions']} sessions in flight" + (" (single stream)" if n == 1 else ""),
                       "parallelism": (f"one process, tensor parallel tp{n}: head-split shards, one per GPU, in-kernel meetings over xGMI peer stores"
                                       if best_tp else
                                       f"one process, layer-sharded pp{n} ({cfg['n_layers'] // n} layers/GPU), stream-ordered "
                                       "hipMemcpyPeerAsync hops of [1,E] F32 (xGMI peer copies, not RCCL send/recv: one host, no communicator)"),
                       "peer_access": r["peer_access"], "sessions_in_flight": r["sessions"], "sessions_agree": r["sessions_agree"],
                       "single_stream_ms_per_token": r["single_stream_ms_per_token"], "prefill_ms": r["prefill_ms_per_session"],
                       "first_ids": r["first_ids"]},
               roofline=roof if roof else {"bound": "hbm", "kernel": "whole decode step", "achieved": round(bytes_per_token * value / 1e9 / n, 1),
                                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bytes_per_token * value / 1e9 / n / HBM_PEAK_GBS, 4),
                                           "traffic": None},
               pipeline_roofline={"achieved_GBps_per_gpu": round(bytes_per_token * value / 1e9 / n, 1),
                                  "frac_of_8TBps": round(bytes_per_token * value / 1e9 / n / HBM_PEAK_GBS, 4)},
               cpu_baseline=cpu, tensor_parallel=r.get("tensor_parallel"))
    return out, 0


def main():
    args = parse()
    from jlama_amd import synthetic as S
    cfg = dict(getattr(S, args.config))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" in os.environ and world > 1:
        # one rank per GPU under torch.distributed.run: the RCCL send/recv pipeline
        from jlama_amd import distributed as D
        out = D.bench_pipeline(args, cfg)
        if out is not None:
            print(json.dumps(out), flush=True)
        if D.hard_exit["now"]:     # a leg left a collective stuck on some rank: the line is out, nobody waits for a teardown that cannot finish
            sys.stdout.flush()
            os._exit(0)
        return
    if args.gpus > 1 or os.environ.get("JH_BENCH_FORCE_PIPELINE"):
        out, rc = run_one_process(args, cfg)   # bare invocation: the one-process host, never a rendezvous
        print(json.dumps(out), flush=True)
        sys.exit(rc)
    out, _ = run_single(args, cfg)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
