"""GPU parity at FULL size: the Llama-3-8B JQ4 model of BASELINE.json's metric (32 layers, E=4096, H=14336, V=128256,
synthetic weights generated on the GPU and copied to the host bit for bit).

What is asserted (bench.py prints the same block as `parity_full_size` on the metric's 256-step run):
  * STRICT ORDER: the metric's 129-row prompt (one batched reference-order prefill, AbstractModel.java:549-555) + 96 free-running
    greedy steps -- ids identical to the Panama-order oracle's, logits equal;
  * the fast kernels' teacher-forced logits sit inside (a small multiple of) the envelope spanned by the reference's OWN
    two CPU providers: the Panama-order restatement vs the reference's compiled C SIMD GEMM (oracle/_ref);
  * every one of the 32 layers in isolation (oracle layer l fed the GPU's input rows of layer l): strict order
    bit-identical, fast kernels within the single-flip bound with per-layer medians at float-ordering level.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_PROMPT, N_FREE, N_TF = 129, 96, 16   # the metric's own prompt: 128 ids + BOS through the batched reference-order prefill


@pytest.fixture(scope="module")
def full8b(gpu):
    import torch
    from jlama_amd import synthetic as S, synthetic_torch as ST
    from jlama_amd.model import HipLlamaModel
    torch.cuda.set_device(0)
    cfg = dict(S.LLAMA3_8B)
    w = ST.make_weights(cfg, seed=0, device="cuda")
    torch.cuda.synchronize()
    model = HipLlamaModel(cfg, w)                 # device-to-device upload
    host_w = ST.to_host(w)                        # the identical bytes for the oracle
    del w
    torch.cuda.empty_cache()
    yield cfg, model, host_w
    model.close()


def test_full_size_strict_ids_and_provider_envelope(full8b, oracle):
    import bench
    cfg, model, host_w = full8b
    par, ids_o = bench.full_size_parity(cfg, model, host_w, N_PROMPT, N_FREE, N_TF)
    print("parity_full_size:", par)
    st = par["strict_order"]
    assert st["n_ids"] == N_FREE + 1
    assert st["ids_equal"] == st["n_ids"], st                      # bit-exact token ids at temperature 0 (BASELINE north_star)
    assert st["logits_vs_panama_oracle"]["max"] <= 1e-5, st        # in practice exactly 0
    assert st["last_step_logits_max_abs_diff"] <= 1e-5, st
    pw = par["teacher_forced_pairwise_logit_distance"]
    if "panama_oracle__reference_c_gemm" in pw:
        env = pw["panama_oracle__reference_c_gemm"]
        for k in ("gpu_fast__panama_oracle", "gpu_fast__reference_c_gemm"):
            # three implementations that differ only in float summation order are three draws from the same noise
            # process (Q8 code flips cascading through 32 layers); a defect would add to it
            # (measured ratios 1.10 / 1.02: the gate leaves room for another seed, not for a regression)
            assert pw[k]["max"] <= 1.3 * env["max"] + 1e-3, (k, pw)
            assert pw[k]["mean_of_max"] <= 1.2 * env["mean_of_max"] + 1e-3, (k, pw)
    d = par["fast_argmax_vs_oracle_at_margin_0.25"]
    assert d["agree"] == d["decided_steps"], d


def test_full_size_every_layer_in_isolation(full8b, oracle):
    from jlama_amd import synthetic as S
    from test_gpu_parity import FLIP_TOL, NOFLIP_TOL, layer_teacher_forced
    cfg, model, host_w = full8b
    prompt = S.prompt_tokens(cfg, n=7, seed=4321)
    layer_teacher_forced(model, oracle, cfg, host_w, prompt, 32, strict=True)          # asserts bit equality per layer
    rel = layer_teacher_forced(model, oracle, cfg, host_w, prompt, 32, strict=False)
    print("per-layer teacher-forced (fast kernels): max", rel.max(), "share <= 1e-5:", float((rel <= NOFLIP_TOL).mean()))
    assert rel.max() <= FLIP_TOL, (rel.max(), np.unravel_index(rel.argmax(), rel.shape))
    assert (rel <= NOFLIP_TOL).mean() >= 0.5
    # every layer past the first: the median row sits at float-ordering level (a flip is the exception; an addressing defect
    # in a layer would lift ALL its rows).  Layer 0 is pinned by itself: its inputs are Q4-lattice embedding rows, whose block
    # maxima sit on quantizer boundaries far more often (4 of its 8 rows caught a flip in round 2) -- there at least 3 of the 8
    # rows must match to float-ordering level and none may exceed the single-flip bound
    med = np.median(rel, axis=1)
    assert (med[1:] <= NOFLIP_TOL).all(), med
    assert (rel[0] <= NOFLIP_TOL).sum() >= 3 and rel[0].max() <= FLIP_TOL, rel[0]
    assert (rel.min(axis=1) <= NOFLIP_TOL).all(), rel.min(axis=1)       # every layer has rows that match to 1e-5
