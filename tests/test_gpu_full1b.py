"""GPU parity at FULL size for BASELINE configs[1]: Llama-3.2-1B-Instruct JQ4 (16 layers, E=2048, H=8192, head size 64, tied
embedding / LM head, V=128256), synthetic weights generated on the GPU and copied to the host bit for bit.

Same assertions as tests/test_gpu_full8b.py: reference order (jh_p16.h) -- prompt + free-running greedy ids identical to the
Panama-order oracle, logits equal; the order-free kernels inside the envelope spanned by the reference's own two CPU providers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_PROMPT, N_FREE, N_TF = 8, 96, 16


@pytest.fixture(scope="module")
def full1b(gpu):
    import torch
    from jlama_amd import synthetic as S, synthetic_torch as ST
    from jlama_amd.model import HipLlamaModel
    torch.cuda.set_device(0)
    cfg = dict(S.LLAMA32_1B)
    w = ST.make_weights(cfg, seed=0, device="cuda")
    torch.cuda.synchronize()
    model = HipLlamaModel(cfg, w)
    host_w = ST.to_host(w)
    del w
    torch.cuda.empty_cache()
    yield cfg, model, host_w
    model.close()


def test_full_size_1b_strict_ids_and_provider_envelope(full1b, oracle):
    import bench
    cfg, model, host_w = full1b
    par, ids_o = bench.full_size_parity(cfg, model, host_w, N_PROMPT, N_FREE, N_TF)
    print("parity_full_size (Llama-3.2-1B):", par)
    st = par["strict_order"]
    assert st["n_ids"] == N_FREE + 1
    assert st["ids_equal"] == st["n_ids"], st
    assert st["logits_vs_panama_oracle"]["max"] <= 1e-5, st        # in practice exactly 0
    assert st["last_step_logits_max_abs_diff"] <= 1e-5, st
    pw = par["teacher_forced_pairwise_logit_distance"]
    if "panama_oracle__reference_c_gemm" in pw:
        env = pw["panama_oracle__reference_c_gemm"]
        for k in ("gpu_fast__panama_oracle", "gpu_fast__reference_c_gemm"):
            assert pw[k]["max"] <= 1.5 * env["max"] + 1e-3, (k, pw)
            assert pw[k]["mean_of_max"] <= 1.3 * env["mean_of_max"] + 1e-3, (k, pw)
    d = par["fast_argmax_vs_oracle_at_margin_0.25"]
    assert d["agree"] == d["decided_steps"], d


def test_full_size_1b_every_layer_in_isolation_reference_order(full1b, oracle):
    from jlama_amd import synthetic as S
    from test_gpu_parity import layer_teacher_forced
    cfg, model, host_w = full1b
    prompt = S.prompt_tokens(cfg, n=7, seed=4321)
    layer_teacher_forced(model, oracle, cfg, host_w, prompt, 32, strict=True)          # asserts bit equality per layer
