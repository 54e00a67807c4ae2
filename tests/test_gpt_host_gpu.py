"""GPT-2 (BASELINE.json configs 1-2) through the C++ host mirror vs the restated reference CPU backend
(oracle orc_cpu_gpt2_forward = GptTransformer::forward over CpuLinearOp / CpuLayerNormOp / CpuAttentionOp /
CpuGeluOp / CpuResidualOp / CpuEncoderOp), fed the identical bf16-rounded parameters.

The GPU path keeps bf16 activations between ops, the CPU reference is FP32 end to end, so the bar is the
reference's own BF16 bar (5e-2 + 5e-2|y|, Linear.Cuda.cpp:111-129) on the logits; argmax agreement is
checked where the top-2 margin exceeds the bar."""
import numpy as np
import pytest

import orc
from mila_amd import host

pytestmark = pytest.mark.gpu


def make_params(rng, V, maxT, C_, L):
    def t(*shape, scale=0.02, offset=0.0):
        return orc.to_bf16_bits((rng.standard_normal(shape) * scale + offset).astype(np.float32))
    ps = [t(V, C_, scale=0.05), t(maxT, C_, scale=0.02)]
    for _ in range(L):
        ps += [t(C_, scale=0.1, offset=1.0), t(C_, scale=0.05), t(3 * C_, C_, scale=C_ ** -0.5), t(3 * C_, scale=0.02),
               t(C_, C_, scale=C_ ** -0.5), t(C_, scale=0.02), t(C_, scale=0.1, offset=1.0), t(C_, scale=0.05),
               t(4 * C_, C_, scale=C_ ** -0.5), t(4 * C_, scale=0.02), t(C_, 4 * C_, scale=(4 * C_) ** -0.5), t(C_, scale=0.02)]
    ps += [t(C_, scale=0.1, offset=1.0), t(C_, scale=0.05), t(V, C_, scale=C_ ** -0.5)]
    return ps


@pytest.mark.parametrize("V,maxT,C_,L,NH,B,T", [(512, 64, 128, 2, 2, 2, 24), (1000, 128, 768, 3, 12, 1, 64)])
def test_gpt2_forward_matches_the_restated_cpu_backend(V, maxT, C_, L, NH, B, T):
    rng = np.random.default_rng(C_ + T)
    params = make_params(rng, V, maxT, C_, L)
    tokens = rng.integers(0, V, (B, T)).astype(np.int32)
    g = host.Gpt(V, maxT, C_, L, NH, B, T)
    g.load_parameters(params)
    got = orc.from_bf16_bits(g.forward(tokens))
    exp = orc.cpu_gpt2_forward(tokens, [orc.from_bf16_bits(p) for p in params], C_, L, NH, V, maxT)
    assert np.all(np.isfinite(got))
    assert np.all(np.abs(got - exp) <= 5e-2 + 5e-2 * np.abs(exp)), np.abs(got - exp).max()
    srt = np.sort(exp, axis=-1)
    clear = (srt[..., -1] - srt[..., -2]) > 0.2
    assert np.array_equal(got.argmax(-1)[clear], exp.argmax(-1)[clear])
    with pytest.raises(IndexError):
        bad = tokens.copy()
        bad[0, 3] = V
        g.forward(bad)
    g.close()


@pytest.mark.parametrize("V,maxT,C_,L,NH,B,T,Tp", [(512, 64, 128, 2, 2, 2, 24, 9), (50257, 1024, 768, 12, 12, 1, 64, 40)], ids=["small", "config1_gpt2_124M_B1_T64"])
def test_gpt2_fp32_on_the_device_matches_the_reference_cpu_backend__config_1(V, maxT, C_, L, NH, B, T, Tp):
    """BASELINE config 1's model -- GPT-2 124M, FP32, B = 1, T = 64 -- ON THE DEVICE, through the FP32 rows the reference keeps "for validation and reference"
    (OperationTraits.Cuda.ixx:50-54, :108-126, :274-282; csrc/fp32_rows.hip), against the reference's CPU backend restated (orc_cpu_gpt2_forward) on the SAME float
    parameters: FP32 against FP32, so the bar is the reference's FP32 one -- 1e-3 + 1e-4 |y| per logit (Linear.Cpu.cpp:278 holds one Linear to
    1e-4 absolute on unit-scale data) -- instead of bf16's 5e-2.  Forward, then the prefill / decode session over the FP32 KV caches."""
    rng = np.random.default_rng(C_ + T)
    params = [orc.from_bf16_bits(p) for p in make_params(rng, V, maxT, C_, L)]      # float32 values (bf16-representable: the bf16 model below sees the same numbers)
    tokens = rng.integers(0, V, (B, T)).astype(np.int32)
    exp = orc.cpu_gpt2_forward(tokens, params, C_, L, NH, V, maxT)
    g = host.Gpt(V, maxT, C_, L, NH, B, T, precision="fp32")
    g.load_parameters(params)
    got = g.forward(tokens)
    assert got.dtype == np.float32 and np.all(np.isfinite(got))
    err = np.abs(got - exp)
    assert np.all(err <= 1e-3 + 1e-4 * np.abs(exp)), float(err.max())
    print("fp32 GPT-2 (C=%d, L=%d) on the device vs the CPU backend: max |err| %.2e, max |logit| %.2f, forward %.2f ms" % (C_, L, float(err.max()), float(np.abs(exp).max()), g.last_ms))
    assert np.array_equal(got.argmax(-1), exp.argmax(-1)) or float(err.max()) < 1e-4
    # the KV-cache session: prefill, then decode to the end
    lp = g.prefill(tokens[:, :Tp])
    assert np.all(np.abs(lp - exp[:, Tp - 1]) <= 1e-3 + 1e-4 * np.abs(exp[:, Tp - 1]))
    for pos in range(Tp, min(T, Tp + 6)):
        ld = g.decode(tokens[:, pos], pos)
        assert np.all(np.abs(ld - exp[:, pos]) <= 1e-3 + 1e-4 * np.abs(exp[:, pos])), pos
    st = g.memory_stats()
    assert st["actual"]["device_parameter_bytes"] == sum(p.size for p in params) * 4
    g.close()
    # the bf16 row of the same model on the same values sits two orders of magnitude further from the CPU backend: what the FP32 rows buy as a validation path
    if C_ <= 128:
        b = host.Gpt(V, maxT, C_, L, NH, B, T)
        b.load_parameters([orc.to_bf16_bits(p) for p in params])
        eb = float(np.abs(orc.from_bf16_bits(b.forward(tokens)) - exp).max())
        b.close()
        assert eb > 20 * float(err.max())


def test_gpt2_124m_full_width_single_block_row():
    """config-2 widths (C=768, NH=12, V=50257) on a short sequence: every Linear/LayerNorm/MHA/LPE row at
    its real size against the CPU reference (the full B=8, T=1024 forward is 2 TFLOP on the FP32 CPU path)."""
    V, maxT, C_, L, NH, B, T = 50257, 1024, 768, 1, 12, 1, 16
    rng = np.random.default_rng(1)
    params = make_params(rng, V, maxT, C_, L)
    tokens = rng.integers(0, V, (B, T)).astype(np.int32)
    g = host.Gpt(V, maxT, C_, L, NH, B, T)
    g.load_parameters(params)
    got = orc.from_bf16_bits(g.forward(tokens))
    exp = orc.cpu_gpt2_forward(tokens, [orc.from_bf16_bits(p) for p in params], C_, L, NH, V, maxT)
    assert np.all(np.abs(got - exp) <= 5e-2 + 5e-2 * np.abs(exp)), np.abs(got - exp).max()
    g.close()


def test_gpt2_124m_config2_full_size_properties():
    """BASELINE.json config 2 at its full size (GPT-2 124M, B=8, T=1024: 2.2 TFLOP, beyond the FP32 CPU oracle's reach), checked
    through size-independent properties of a causal transformer, bit for bit:
      * a batch row's logits do not depend on its slot or on the other rows (rows 0 and 5 carry the same tokens);
      * causality: changing tokens at positions >= 512 leaves every logit of positions < 512 untouched;
      * the first 64 positions of row 0 equal a B=1, T=64 forward of the same tokens up to the reference's BF16 bar (a different
        GEMM tile grid), and that short forward matches the restated CPU backend."""
    V, maxT, C_, L, NH, B, T = 50257, 1024, 768, 12, 12, 8, 1024
    rng = np.random.default_rng(124)
    params = make_params(rng, V, maxT, C_, L)
    tokens = rng.integers(0, V, (B, T)).astype(np.int32)
    tokens[5] = tokens[0]
    g = host.Gpt(V, maxT, C_, L, NH, B, T)
    g.load_parameters(params)
    a = g.forward(tokens)
    print("GPT-2 124M B=8 T=1024 forward: %.2f ms" % g.last_ms)
    assert a.shape == (B, T, V)
    assert np.array_equal(a[0], a[5]), "identical rows in different slots differ"
    tokens2 = tokens.copy()
    tokens2[:, 512:] = rng.integers(0, V, (B, T - 512))
    b = g.forward(tokens2)
    assert np.array_equal(a[:, :512], b[:, :512]), "a later token changed an earlier position"
    assert not np.array_equal(a[:, 512:], b[:, 512:])
    f = orc.from_bf16_bits(a[0, :64])
    assert np.all(np.isfinite(f))
    g.close()
    s = host.Gpt(V, maxT, C_, L, NH, 1, 64)
    s.load_parameters(params)
    short = orc.from_bf16_bits(s.forward(tokens[:1, :64]))[0]
    s.close()
    assert np.all(np.abs(f - short) <= 5e-2 + 5e-2 * np.abs(short)), np.abs(f - short).max()
    exp = orc.cpu_gpt2_forward(tokens[:1, :64], [orc.from_bf16_bits(p) for p in params], C_, L, NH, V, maxT)[0]
    assert np.all(np.abs(short - exp) <= 5e-2 + 5e-2 * np.abs(exp)), np.abs(short - exp).max()


@pytest.mark.parametrize("V,maxT,C_,L,NH,B,T,Tp", [(512, 64, 128, 2, 2, 2, 24, 9), (1000, 128, 768, 3, 12, 1, 64, 40), (256, 32, 64, 2, 8, 3, 16, 5)],
                         ids=["HS64_B2", "HS64_width768", "HS8_generic_kernel_B3"])
def test_gpt2_prefill_then_decode_matches_the_full_forward_and_the_cpu_backend(V, maxT, C_, L, NH, B, T, Tp):
    """GptTransformer::prefill / decode (GptTransformer.ixx:330-441) over the per-block KV caches (GptBlock::decode, GptBlock.ixx:253-281; MHA over a KV
    cache, CudaMhaOp.ixx:145-380): prefill T' tokens, then decode the rest one by one -- every step's logits equal that position's row of the full-sequence
    forward (another kernel family: the reference's BF16 bar) and of the restated CPU backend; prefill after decode steps starts a new session"""
    rng = np.random.default_rng(C_ + T + B)
    params = make_params(rng, V, maxT, C_, L)
    tokens = rng.integers(0, V, (B, T)).astype(np.int32)
    g = host.Gpt(V, maxT, C_, L, NH, B, T)
    g.load_parameters(params)
    full = orc.from_bf16_bits(g.forward(tokens))                       # [B, T, V]
    exp = orc.cpu_gpt2_forward(tokens, [orc.from_bf16_bits(p) for p in params], C_, L, NH, V, maxT)

    def near(got, want, what):
        assert np.all(np.isfinite(got)), what
        assert np.all(np.abs(got - want) <= 5e-2 + 5e-2 * np.abs(want)), (what, float(np.abs(got - want).max()))

    for session in range(2):
        lp = orc.from_bf16_bits(g.prefill(tokens[:, :Tp]))             # [B, V]: the last prompt position
        near(lp, full[:, Tp - 1], "prefill logits vs forward (session %d)" % session)
        near(lp, exp[:, Tp - 1], "prefill logits vs CPU backend")
        for pos in range(Tp, T):
            ld = orc.from_bf16_bits(g.decode(tokens[:, pos], pos))
            near(ld, full[:, pos], "decode @%d vs forward" % pos)
            near(ld, exp[:, pos], "decode @%d vs CPU backend" % pos)
            srt = np.sort(exp[:, pos], axis=-1)
            clear = (srt[:, -1] - srt[:, -2]) > 0.2
            assert np.array_equal(ld.argmax(-1)[clear], exp[:, pos].argmax(-1)[clear])
    with pytest.raises(ValueError):
        g.decode(tokens[:, 0], T)                                      # position beyond the built sequence length
    g.close()


def test_the_model_is_the_references_graph():
    """GptTransformer.ixx:828-858 (lenc, tf_layer_<i>, ln_final, lm_head), GptBlock.ixx:512-546 (attn, ln_1, ln_2, fc_qkv_proj, fc_out_proj, res_1, res_2, mlp),
    MLP.ixx:410-412 (fc_1, gelu, fc_2): the same components under the same names, in construction order"""
    m = host.Gpt(128, 16, 64, 2, 4, 1, 8)
    try:
        want = ["gpt.lenc"]
        for i in range(2):
            b = "gpt.tf_layer_%d" % i
            want += [b] + [b + "." + leaf for leaf in ("attn", "ln_1", "ln_2", "fc_qkv_proj", "fc_out_proj", "res_1", "res_2", "mlp", "mlp.fc_1", "mlp.gelu", "mlp.fc_2")]
        want += ["gpt.ln_final", "gpt.lm_head"]
        assert m.component_names() == want
    finally:
        m.close()
