"""Whole-model logits at the north star's 1e-3 (BASELINE.json: "logits within 1e-3 rel of reference").

Under SURVEY.md's unit-scale random weights a 1-ulp bf16 difference between two CORRECT implementations grows ~1.4x per block
(tools/depth_probe.py, tools/condition_probe.py), so tests/test_gemma_host_gpu.py can only hold a 6-layer random model to 1e-1 of the
logit range.  Here the synthetic model is CONDITIONED like a trained one (tests/ref_gemma.py CONDITIONED_PROFILE: residual updates
small against the stream, attention scores O(1), a layer scalar != 1) so that differences stay at the rounding floor, and a
12-layer Gemma-shaped model with two global layers is held, for all three weight policies, decode and prefill, to

    max |logit_gpu - logit_oracle|  <=  1e-3 * max |logit_oracle|

against the oracle composition with every bf16 rounding on (block order: Gemma.Block.ixx:197-356; Gemma.ixx:281-297).  The same
bar is met by the oracle composition run on a different summation order (tests/test_conditioned_cpu.py), i.e. it is the distance
between two correct implementations, and a wrong term in a fused prologue (a dropped layer scalar, a norm weight from the wrong
layer, eps misplaced) lands orders of magnitude outside it (checked below by perturbing the ORACLE)."""
import os

import numpy as np
import pytest

from mila_amd import host
from ref_gemma import CONDITIONED_PROFILE, RefGemma

pytestmark = pytest.mark.gpu

CFG = dict(vocab_size=2048, embedding_dim=1280, num_layers=12, num_heads=4, num_kv_heads=2, head_dim=64, hidden_dim=2560,
           global_head_dim=128, num_global_kv_heads=1, window=8, sliding_window_pattern=6, global_rotary_dim=32)
TOKENS = [(7 * i + 3) % 2048 for i in range(20)]      # 20 positions: past the sliding window (8), both global layers see all of them
MAX_SEQ = 64
BAR = 1e-3
# The fp4 policy's prefill is W4A8 at every M > 1 (the reference's default, CudaLinearOp.ixx:646-715; round 3 -- it used to fall back to W4A16 at this T): the
# activations are re-quantized per token to e4m3 in front of every Linear, so a 1-ulp bf16 difference between two correct implementations becomes a 6 % step of
# that element whenever it sits at an e4m3 rounding boundary.  Two correct W4A8 compositions therefore sit further apart than two bf16 ones: the oracle against
# itself with RMSNorm reduced in fp32 instead of double differs by 1.5e-3 (tests/test_conditioned_cpu.py); the GPU measures 1.8e-3.  The reference's own bar for ONE
# W4A8 Linear against the exact-weight path is 1e-1 * row_absmax (Linear.Cuda.cpp:760-774).  Everything else -- bf16 and fp8 prefill, every decode leg -- stays at 1e-3.
BAR_W4A8_PREFILL = 3e-3


def _report(tag, got, exp):
    d = np.abs(got.astype(np.float64) - exp.astype(np.float64))
    rel = d / np.abs(exp).max()
    hist = np.histogram(np.log10(np.maximum(rel, 1e-12)), bins=[-12, -7, -6, -5, -4, -3.5, -3, -2, 0])[0]
    line = ("%s: max %.2e of max|logit| (%.2e of std), histogram of log10(err/max|logit|) over (-inf,-7,-6,-5,-4,-3.5,-3,-2,0]: %s"
            % (tag, rel.max(), d.max() / exp.std(), hist.tolist()))
    print(line)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):      # kept beside the test log on the GPU box; the summary is copied into profiles/
        with open(os.path.join(out, "conditioned_logit_report.txt"), "a") as f:
            f.write(line + "\n")
    return float(rel.max())


def test_fp8_policy_w8a8_opt_in_prefill_on_the_conditioned_model():
    """PerChannelFp8<> with setFp8ActivationPrefill(true) (row g1): the policy's own e4m3 weights x per-token e4m3 activations on the fp8 matrix cores, against the
    oracle composition with that Linear arithmetic at T > 1 (RefGemma w8a8_prefill).  Per-token e4m3 activations -> the W4A8 leg's bar (see BAR_W4A8_PREFILL)."""
    refp = RefGemma(CFG, "fp8", seed=7, profile=CONDITIONED_PROFILE, w8a8_prefill=True)
    exp = refp.forward(TOKENS, 0, MAX_SEQ)
    p = host.Gemma("fp8", CFG, max_seq=MAX_SEQ, max_prefill=32, seed=7, profile=CONDITIONED_PROFILE)
    p.set_fp8_activation_prefill(True)
    got = p.prefill(TOKENS)
    assert _report("fp8-w8a8 prefill T=%d" % len(TOKENS), got, exp) <= BAR_W4A8_PREFILL
    exp1 = refp.forward([5], len(TOKENS), MAX_SEQ)
    assert _report("fp8-w8a8 decode after prefill", p.decode(5, len(TOKENS), "fused"), exp1) <= BAR_W4A8_PREFILL
    # and it IS another function than the policy's default W8A16 prefill (tests/test_conditioned_cpu.py measures 3.2e-3 between the two compositions)
    w8a16 = RefGemma(CFG, "fp8", seed=7, profile=CONDITIONED_PROFILE, staged_prefill=True).forward(TOKENS, 0, MAX_SEQ)
    _report("fp8-w8a8 prefill vs the W8A16 composition (another function)", got, w8a16)
    assert np.abs(got - w8a16).max() > BAR * np.abs(w8a16).max()
    p.close()


@pytest.mark.parametrize("policy", ["fp4", "fp8-w8a8"])
def test_teacher_forced_fp8_activation_prefill_holds_1e3(policy):
    """besides the 3e-3 end-to-end bar of the per-token-e4m3 prefills (W4A8, the fp4 policy's default; W8A8, the fp8 policy's opt-in), the NORTH STAR's
    1e-3 holds when the oracle is teacher-forced: it multiplies the e4m3 activation bytes and scales the GPU's own Linears consumed (host.activation_tap, all 48 Linear calls
    of the 12 layers, in order) instead of re-quantizing its own -- everything else (norms, RoPE, attention, GeGLU, residuals, the contraction and its epilogue) is the
    oracle's.  What the 3e-3 bar absorbs is therefore e4m3 code flips caused by 1-ulp bf16 differences upstream, and nothing else; the flips are counted and reported."""
    pol = "fp8" if policy == "fp8-w8a8" else policy
    p = host.Gemma(pol, CFG, max_seq=MAX_SEQ, max_prefill=32, seed=7, profile=CONDITIONED_PROFILE)
    if policy == "fp8-w8a8":
        p.set_fp8_activation_prefill(True)
    fused = p.prefill(TOKENS)                        # the product path (fused glue: tails hand over per-token e4m3 rows)
    p.set_fused_prefill(False)                       # ... and one launch per reference op: every Linear quantizes inside RocmLinearOp::forward, where the tap sits
    with host.activation_tap() as tap:
        got = p.prefill(TOKENS)
    p.close()
    assert np.array_equal(got.view(np.uint32), fused.view(np.uint32)), "the tapped per-op prefill is not the product prefill"
    assert len(tap.records) == 4 * CFG["num_layers"]
    ref = RefGemma(CFG, pol, seed=7, profile=CONDITIONED_PROFILE, w4a8_prefill=(policy == "fp4"), w8a8_prefill=(policy == "fp8-w8a8"))
    ref.forced = iter(tap.records)
    exp = ref.forward(TOKENS, 0, MAX_SEQ)
    flips = ref.forced_flips / max(ref.forced_codes, 1)
    err = _report("%s prefill T=%d, teacher-forced (%.3f %% of the e4m3 activation codes differ from the oracle's own)" % (policy, len(TOKENS), 100 * flips), got, exp)
    assert err <= BAR, err
    assert flips < 0.02


@pytest.mark.parametrize("policy", ["bf16", "fp8", "fp4"])
def test_conditioned_12_layer_model_holds_1e3_on_decode_and_prefill(policy):
    ref = RefGemma(CFG, policy, seed=7, profile=CONDITIONED_PROFILE)
    g = {m: host.Gemma(policy, CFG, max_seq=MAX_SEQ, max_prefill=1, seed=7, profile=CONDITIONED_PROFILE) for m in ("reference", "graph")}
    worst = 0.0
    for pos, tok in enumerate(TOKENS):
        exp = ref.forward([tok], pos, MAX_SEQ)
        out = {m: mdl.decode(tok, pos, m) for m, mdl in g.items()}
        assert np.array_equal(out["reference"].view(np.uint32), out["graph"].view(np.uint32)), "graph replay != reference order at %d" % pos
        assert np.all(np.isfinite(out["graph"]))
        if pos in (0, 7, 8, 9, len(TOKENS) - 1):
            worst = max(worst, _report("%s decode @%d" % (policy, pos), out["graph"], exp))
        else:
            worst = max(worst, float(np.abs(out["graph"] - exp).max() / np.abs(exp).max()))
    assert worst <= BAR, worst
    for mdl in g.values():
        mdl.close()
    # prefill (GEMM + flash attention + fused glue) of the same prompt: the fp8 policy multiplies by bf16(dequantized weight) there (W8A16), the fp4 policy
    # runs the reference's default W4A8 (e4m3 weights x per-token e4m3 activations, CudaLinearOp.ixx:646-715) at EVERY M > 1 -- T = 20 included, since round 3
    refp = RefGemma(CFG, policy, seed=7, profile=CONDITIONED_PROFILE, staged_prefill=True, w4a8_prefill=True)
    exp = refp.forward(TOKENS, 0, MAX_SEQ)
    p = host.Gemma(policy, CFG, max_seq=MAX_SEQ, max_prefill=32, seed=7, profile=CONDITIONED_PROFILE)
    got = p.prefill(TOKENS)
    bar = BAR_W4A8_PREFILL if policy == "fp4" else BAR
    assert _report("%s prefill T=%d" % (policy, len(TOKENS)), got, exp) <= bar
    # and one decode step on top of the prefilled caches (which carry the prefill's distance)
    exp1 = refp.forward([5], len(TOKENS), MAX_SEQ)
    assert _report("%s decode after prefill" % policy, p.decode(5, len(TOKENS), "fused"), exp1) <= bar
    p.close()


def test_the_bar_catches_small_wrong_terms():
    """what "a wrong-but-small term in a fused prologue" costs on this model: a dropped layer scalar (0.969 -> 1) moves the logits by
    ~1e-2 of max|logit|, a 10 % error in the post-norm weights by 6e-3, a 20 % error in the q/k-norm weights by ~4e-3 (10 %: 1.9e-3) --
    all outside the 1e-3 bar the GPU meets
    (a 2 % norm-weight error would sit AT the bar: that is the resolution of a whole-model bf16 comparison)"""
    base = RefGemma(CFG, "bf16", seed=7, profile=CONDITIONED_PROFILE).forward(TOKENS[:6], 0, MAX_SEQ)
    g = host.Gemma("bf16", CFG, max_seq=MAX_SEQ, max_prefill=8, seed=7, profile=CONDITIONED_PROFILE)
    got = g.prefill(TOKENS[:6])
    g.close()
    assert np.abs(got - base).max() <= BAR * np.abs(base).max()
    for name, prof in (("layer scalar dropped", dict(CONDITIONED_PROFILE, layer_scalar=1.0)),
                       ("post-norm weights 10 % off", dict(CONDITIONED_PROFILE, post_norm_center=CONDITIONED_PROFILE["post_norm_center"] * 1.10)),
                       ("q/k-norm weights 20 % off", dict(CONDITIONED_PROFILE, qk_norm_center=CONDITIONED_PROFILE["qk_norm_center"] * 1.20))):
        wrong = RefGemma(CFG, "bf16", seed=7, profile=prof).forward(TOKENS[:6], 0, MAX_SEQ)
        err = np.abs(got - wrong).max() / np.abs(wrong).max()
        print("%s: %.2e" % (name, err))
        assert err > 2 * BAR, (name, err)
