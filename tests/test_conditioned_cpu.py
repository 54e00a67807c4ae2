"""CPU companion of tests/test_gemma_conditioned_gpu.py: the 1e-3 whole-model bar is the distance between two CORRECT
implementations of the conditioned model -- the oracle composition (double accumulation) against the same composition on FP32
accumulation in another order -- while the unit-scale random profile sits far outside it.  Small sizes: seconds on one core."""
import numpy as np

from ref_gemma import CONDITIONED_PROFILE, DEFAULT_PROFILE, RefGemma

CFG = dict(vocab_size=1024, embedding_dim=512, num_layers=12, num_heads=4, num_kv_heads=2, head_dim=64, hidden_dim=1024,
           global_head_dim=128, num_global_kv_heads=1, window=8, sliding_window_pattern=6, global_rotary_dim=32)
TOK = [(7 * i + 3) % 1024 for i in range(12)]


def _distance(profile):
    a = RefGemma(CFG, "bf16", 7, profile=profile).forward(TOK, 0, 32)
    b = RefGemma(CFG, "bf16", 7, profile=profile, f32_stand_in=True).forward(TOK, 0, 32)
    return float(np.abs(a - b).max() / np.abs(a).max())


def test_conditioned_profile_keeps_two_correct_implementations_within_1e3():
    assert _distance(CONDITIONED_PROFILE) <= 1e-3


def test_unit_scale_random_weights_do_not():
    assert _distance(DEFAULT_PROFILE) > 1e-2


def test_w4a8_prefill_composition_is_held_by_two_accumulation_orders_too():
    """the fp4 policy's default prefill (W4A8: e4m3 weights x per-token e4m3 activations): products of two e4m3 values are exact in FP32, so a second
    implementation that accumulates in FP32 in another order lands on (nearly) the same bf16 outputs -- and the path is a different arithmetic from the
    exact-weight W4A16 prefill, by more than the bar (which is why the GPU must not switch between the two with the prompt length)"""
    a = RefGemma(CFG, "fp4", 7, profile=CONDITIONED_PROFILE, w4a8_prefill=True).forward(TOK, 0, 32)
    b = RefGemma(CFG, "fp4", 7, profile=CONDITIONED_PROFILE, w4a8_prefill=True, w4a8_f32_order=True).forward(TOK, 0, 32)
    c = RefGemma(CFG, "fp4", 7, profile=CONDITIONED_PROFILE, staged_prefill=True).forward(TOK, 0, 32)
    assert np.abs(a - b).max() <= 1e-3 * np.abs(a).max()
    assert np.abs(a - c).max() > 2e-3 * np.abs(a).max()


class _Fp32Norm(RefGemma):
    """a second correct implementation whose RMSNorm reduces in float32 in reverse order: 1-ulp bf16 differences upstream of every Linear"""
    def rms(self, x, w):
        x = np.asarray(x, np.float32)
        ms = (x[..., ::-1] ** 2).sum(-1, dtype=np.float32, keepdims=True) / np.float32(x.shape[-1])
        return self.r(x * (np.float32(1) / np.sqrt(ms + np.float32(1e-6))) * np.asarray(w, np.float32))


def test_w4a8_turns_upstream_rounding_into_e4m3_steps_hence_its_own_bar():
    """why tests/test_gemma_conditioned_gpu.py holds the fp4 policy's W4A8 prefill to 3e-3 and not 1e-3: with per-token e4m3 activations a 1-ulp bf16 difference
    upstream flips e4m3 codes (6 % steps), so two correct compositions differ by more than under bf16 activations -- but stay far inside 3e-3"""
    a = RefGemma(CFG, "fp4", 7, profile=CONDITIONED_PROFILE, w4a8_prefill=True).forward(TOK, 0, 32)
    b = _Fp32Norm(CFG, "fp4", 7, profile=CONDITIONED_PROFILE, w4a8_prefill=True).forward(TOK, 0, 32)
    d = float(np.abs(a - b).max() / np.abs(a).max())
    assert 0.0 < d <= 3e-3, d


def test_w8a8_prefill_composition_has_the_w4a8_legs_resolution():
    """the fp8 policy's OPT-IN W8A8 prefill (the policy's e4m3 weights x per-token e4m3 activations; RefGemma w8a8_prefill): like W4A8 it re-quantizes the activations in
    front of every Linear, so two correct compositions (RMSNorm reduced in double / in float32 in reverse order) sit within 3e-3, not 1e-3 -- and it is another function
    than the policy's default W8A16 prefill by more than that bar (which is why it is a switch, never a silent dispatch)"""
    a = RefGemma(CFG, "fp8", 7, profile=CONDITIONED_PROFILE, w8a8_prefill=True).forward(TOK, 0, 32)
    b = _Fp32Norm(CFG, "fp8", 7, profile=CONDITIONED_PROFILE, w8a8_prefill=True).forward(TOK, 0, 32)
    c = RefGemma(CFG, "fp8", 7, profile=CONDITIONED_PROFILE, staged_prefill=True).forward(TOK, 0, 32)
    d = float(np.abs(a - b).max() / np.abs(a).max())
    assert d <= 3e-3, d                                        # (0.0 on this small model: no e4m3 code flips; the W4A8 leg above measures 1.5e-3)
    assert np.abs(a - c).max() > 2e-3 * np.abs(a).max()      # measured 3.2e-3
