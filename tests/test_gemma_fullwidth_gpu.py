"""The north star's "logits within 1e-3 rel of reference" on the geometry -- and therefore the KERNEL SELECTIONS -- the benchmark runs
(the 12-layer conditioned test of tests/test_gemma_conditioned_gpu.py has D = 1280, head sizes 64 / 128, window 8,
T = 20 and so never reaches gemm256*, W4A8, the LDS-DMA flash prefill or the HS 256 / 512 decode attention).

Model: FULL WIDTH -- D 3840, 16 query heads, 8 KV heads x 256 (sliding window 1024) and 1 KV head x 512 (global, partial rotary 128), F 15360 -- four
layers (three sliding-window + one global), vocabulary 2048, parameters conditioned like a trained model's (tests/ref_gemma.py CONDITIONED_PROFILE).
Prefill T = 2048 (the benchmark's chunk: 256 x 256 and 256 x 128 LDS-DMA GEMMs, the fused GeGLU epilogue, W4A8 on the fp8 matrix cores for the fp4 policy,
LDS-DMA flash attention with rows that see a full window) and three decode steps at positions 2048 .. 2050 (fused matvecs, split-K flash decode at HS 256 /
512 over a 2048-row cache, graph replay), three weight policies.

Expectation: the oracle composition with every bf16 rounding on, generated in the build container (float64 BLAS, ~15 minutes -- too slow for the GPU box) and
committed as tests/golden/gemma_fullwidth_logits.npz with its script (make_gemma_fullwidth_golden.py).  Bar: max |gpu - oracle| <= BAR * max |oracle|."""
import os

import numpy as np
import pytest

from mila_amd import host

pytestmark = pytest.mark.gpu

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gemma_fullwidth_logits.npz")
BAR = 1e-3
# measured on MI355X (round 3, profiles/r03_fullwidth_logit_report.txt): bf16 1.7e-4, fp8 1.7e-4 (prefill) / 2.0e-4 (decode), fp4 5.5e-4 (W4A8 prefill) / 3.6e-4 (decode)
# -- the W4A8 leg, whose per-token e4m3 activations turn a 1-ulp bf16 difference into a 6 % step (tests/test_conditioned_cpu.py), is inside the same 1e-3 here
BAR_W4A8_PREFILL = 1e-3


def _rel(got, exp):
    return float(np.abs(got.astype(np.float64) - exp.astype(np.float64)).max() / np.abs(exp).max())


FIXTURE_W8A8 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gemma_fullwidth_logits_w8a8.npz")


@pytest.mark.parametrize("policy", ["bf16", "fp8", "fp4", "fp8-w8a8"])
def test_full_width_model_holds_the_logit_bar_on_the_benchmarked_kernels(policy):
    # "fp8-w8a8": the PerChannelFp8<> policy with the opt-in W8A8 prefill (setFp8ActivationPrefill: the policy's own e4m3 weights + per-channel scales on the fp8 matrix
    # cores, per-token e4m3 activations) against its own fixture -- the same composition with that Linear arithmetic at T > 1 (make_gemma_fullwidth_golden.py --w8a8)
    w8a8 = policy == "fp8-w8a8"
    fx = np.load(FIXTURE_W8A8 if w8a8 else FIXTURE, allow_pickle=False)
    policy = "fp8" if w8a8 else policy
    cfg = {str(k): int(v) for k, v in zip(fx["cfg_keys"], fx["cfg_vals"])}
    profile = {str(k): float(v) for k, v in zip(fx["profile_keys"], fx["profile_vals"])}
    tokens, nxt = fx["tokens"].astype(np.int32), [int(t) for t in fx["next_tokens"]]
    exp = fx["logits_" + policy]
    T = len(tokens)
    assert cfg["embedding_dim"] == 3840 and cfg["hidden_dim"] == 15360 and cfg["window"] == 1024 and T == 2048
    g = host.Gemma(policy, cfg, max_seq=int(fx["max_seq"]), max_prefill=T, seed=int(fx["seed"]), profile=profile)
    try:
        if w8a8:
            g.set_fp8_activation_prefill(True)
            policy = "fp8-w8a8"
        got = g.prefill(tokens)
        errs = {"prefill T=2048": _rel(got, exp[0])}
        bar = BAR_W4A8_PREFILL if policy in ("fp4", "fp8-w8a8") else BAR
        assert np.all(np.isfinite(got))
        lines = ["%s prefill T=%d: %.2e of max|logit| (bar %.0e)" % (policy, T, errs["prefill T=2048"], bar)]
        worst_decode = 0.0
        for i, tok in enumerate(nxt):
            out = g.decode(tok, T + i, "graph")
            e = _rel(out, exp[1 + i])
            worst_decode = max(worst_decode, e)
            lines.append("%s decode @%d: %.2e of max|logit|" % (policy, T + i, e))
        print("\n".join(lines))
        rep = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(rep):
            with open(os.path.join(rep, "fullwidth_logit_report.txt"), "a") as f:
                f.write("\n".join(lines) + "\n")
        assert errs["prefill T=2048"] <= bar, lines
        # the decode steps read the caches the prefill wrote: for the fp4 policy they inherit the W4A8 prefill's distance
        assert worst_decode <= bar, lines
    finally:
        g.close()
