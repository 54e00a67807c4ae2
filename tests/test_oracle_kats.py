"""Pin the CPU oracle against the reference's OWN test scenarios (CPU-only, no GPU).

The reference ships no golden-vector files: its tests pair closed-form input
generators with an in-test host reference and a tolerance.  Each test below restates one of those
scenarios -- same generator, same shapes, same tolerance, an independent float64 numpy reference --
and requires the oracle (oracle/mila_oracle.c) to pass it exactly as the reference's op must.
Citations are relative to /root/reference/Mila/Tests/Dnn/Components.
"""
import numpy as np
import pytest

import orc


# ---- generators restated from the reference tests ------------------------------------------------
def lin_weight(out_f, in_f):
    # Linear/Linear.Cpu.cpp + Linear.Cuda.cpp:70-75: 0.1*((13o+7i)%17-8)/17
    o = np.arange(out_f)[:, None]
    i = np.arange(in_f)[None, :]
    return (np.float32(0.1) * (((o * 13 + i * 7) % 17).astype(np.float32) - np.float32(8.0)) / np.float32(17.0)).astype(np.float32)


def lin_bias(out_f):
    # Linear.Cuda.cpp:77-80: 0.1*((o%5)-2)/5
    o = np.arange(out_f)
    return (np.float32(0.1) * ((o % 5).astype(np.float32) - np.float32(2.0)) / np.float32(5.0)).astype(np.float32)


def spread(shape):
    # Linear.Cuda.cpp:214-224 (spreadHost): i/size*2-1
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.float32)
    return (i / np.float32(n) * np.float32(2.0) - np.float32(1.0)).reshape(shape).astype(np.float32)


def norm_weight(n):
    # Normalization/LayerNorm/LayerNorm.Cpu.cpp:43-46: 0.5 + 0.1*((i%5)-2)
    i = np.arange(n)
    return (np.float32(0.5) + np.float32(0.1) * ((i % 5) - 2).astype(np.float32)).astype(np.float32)


def norm_bias(n):
    # LayerNorm.Cpu.cpp:48-51: 0.05*((i%7)-3)
    i = np.arange(n)
    return (np.float32(0.05) * ((i % 7) - 3).astype(np.float32)).astype(np.float32)


def sin_spread(shape, phase=0.0):
    # Attention/MHA/MultiHeadAttention.Cpu.cpp:104-110: sin(0.2*i + phase)
    n = int(np.prod(shape))
    return np.sin(np.float32(0.2) * np.arange(n, dtype=np.float32) + np.float32(phase)).astype(np.float32).reshape(shape)


# ---- Linear (Linear.Cpu.cpp:247-358, tol 1e-4) ----------------------------------------------------
@pytest.mark.parametrize("shape,bias", [((2, 3, 4), True), ((2, 3, 4), False), ((16, 4), True)])
def test_cpu_linear_matches_reference_scenarios(shape, bias):
    in_f, out_f = 4, 3
    W = lin_weight(out_f, in_f)
    b = lin_bias(out_f) if bias else None
    X = spread(shape)
    Y = orc.cpu_linear(X, W, b)
    exp = X.astype(np.float64).reshape(-1, in_f) @ W.astype(np.float64).T
    if bias:
        exp = exp + b.astype(np.float64)
    np.testing.assert_allclose(Y.reshape(-1, out_f), exp, atol=1e-4, rtol=0)
    # batch 6 -> naive (long double) path, batch 16 -> unrolled (float) path: both must agree
    Yn = orc.cpu_linear(X, W, b, path="naive")
    np.testing.assert_allclose(Y, Yn, atol=1e-6, rtol=0)


def test_cpu_linear_unrolled_is_float_accumulation_seeded_with_bias():
    # CpuLinearOp.ixx:428-446: result = bias; result += x*w in float, in K order
    rng = np.random.default_rng(7)
    X = rng.standard_normal((8, 64)).astype(np.float32)
    W = rng.standard_normal((5, 64)).astype(np.float32)
    b = rng.standard_normal(5).astype(np.float32)
    Y = orc.cpu_linear(X, W, b, path="unrolled")
    exp = np.empty((8, 5), np.float32)
    for m in range(8):
        for n in range(5):
            acc = np.float32(b[n])
            for k in range(64):
                acc = np.float32(acc + np.float32(X[m, k] * W[n, k]))
            exp[m, n] = acc
    assert np.array_equal(Y, exp)


# ---- GELU (Activations/Gelu/Gelu.Cpu.cpp:150-160, tol 1e-4) + reference header pin -----------------
def test_cpu_gelu_reference_scenario_and_known_values():
    x = spread((2, 3, 8)) * np.float32(3.0)
    y = orc.cpu_gelu(x)
    xd = x.astype(np.float64)
    exp = 0.5 * xd * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (xd + 0.044715 * xd ** 3)))
    np.testing.assert_allclose(y, exp, atol=1e-4, rtol=0)
    # values recorded in SURVEY.md section 0 finding 6 from the reference header itself
    assert abs(orc.lib.orc_gelu_tanh(1.0) - 0.841192007) < 1e-7
    assert abs(orc.lib.orc_gelu_tanh(-2.5) - (-0.0150842965)) < 1e-8


def test_activation_restatement_is_bit_identical_to_reference_header():
    """oracle/_ref is the reference's ElementwiseActivation.h compiled where it lies."""
    ref = orc.ref_activation_lib()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference tree absent on this box)")
    xs = np.concatenate([np.linspace(-12, 12, 4001), np.random.default_rng(0).standard_normal(2000) * 4,
                         [0.0, -0.0, 1e-30, -1e-30, 88.0, -88.0]]).astype(np.float32)
    for x in xs:
        a = np.float32(orc.lib.orc_gelu_tanh(float(x)))
        b = np.float32(ref.ref_gelu_tanh(float(x)))
        assert a.tobytes() == b.tobytes(), (x, a, b)
        a = np.float32(orc.lib.orc_silu(float(x)))
        b = np.float32(ref.ref_silu(float(x)))
        assert a.tobytes() == b.tobytes(), (x, a, b)
    y = orc.cpu_gelu(xs)
    exp = np.array([ref.ref_gelu_tanh(float(x)) for x in xs], dtype=np.float32)
    assert np.array_equal(y, exp)


def test_activation_restatement_matches_the_committed_reference_vectors():
    """tests/golden/activation_reference.json holds the reference header's own outputs (made by make_activation_golden.py from
    oracle/_ref): the pin travels to boxes without /root/reference; float32 bit patterns, compared bit for bit"""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "activation_reference.json")))
    xs = np.array(g["x"], dtype=np.uint32).view(np.float32)
    assert xs.size > 600
    for x, ge, si in zip(xs, g["gelu_tanh"], g["silu"]):
        assert int(np.float32(orc.lib.orc_gelu_tanh(float(x))).view(np.uint32)) == ge, x
        assert int(np.float32(orc.lib.orc_silu(float(x))).view(np.uint32)) == si, x
    assert np.array_equal(orc.cpu_gelu(xs).view(np.uint32), np.array(g["gelu_tanh"], dtype=np.uint32))


# ---- LayerNorm (Normalization/LayerNorm/LayerNorm.Cpu.cpp:210-255, tol 1e-4) ------------------------
@pytest.mark.parametrize("bias", [True, False])
def test_cpu_layernorm_reference_scenario(bias):
    Cn = 8
    X = spread((2, 3, Cn)) * np.float32(2.0) + np.float32(0.25)
    w = norm_weight(Cn)
    b = norm_bias(Cn) if bias else None
    Y, mean, rstd = orc.cpu_layernorm(X, w, b, eps=1e-5, return_stats=True)
    xd = X.astype(np.float64).reshape(-1, Cn)
    mu = xd.mean(1, keepdims=True)
    var = ((xd - mu) ** 2).mean(1, keepdims=True)
    exp = (xd - mu) / np.sqrt(var + 1e-5) * w.astype(np.float64) + (b.astype(np.float64) if bias else 0.0)
    np.testing.assert_allclose(Y.reshape(-1, Cn), exp, atol=1e-4, rtol=0)
    np.testing.assert_allclose(mean, mu[:, 0], atol=1e-6)
    np.testing.assert_allclose(rstd, 1.0 / np.sqrt(var[:, 0] + 1e-5), rtol=1e-6)


# ---- Softmax (Normalization/Softmax.Cpu.cpp:150-165: rows sum to 1 within 1e-5) ---------------------
@pytest.mark.parametrize("axis", [-1, 1, 0])
def test_cpu_softmax_reference_scenario(axis):
    X = sin_spread((3, 4, 5), 0.3) * np.float32(4.0)
    Y = orc.cpu_softmax(X, axis=axis)
    e = np.exp(X.astype(np.float64) - X.astype(np.float64).max(axis=axis, keepdims=True))
    exp = e / e.sum(axis=axis, keepdims=True)
    np.testing.assert_allclose(Y, exp, atol=1e-6, rtol=0)
    np.testing.assert_allclose(Y.astype(np.float64).sum(axis=axis), 1.0, atol=1e-5)


# ---- MHA (Attention/MHA/MultiHeadAttention.Cpu.cpp:180-203, tol 1e-4) --------------------------------
def test_cpu_mha_reference_scenario():
    B, T, Cm, NH = 2, 3, 8, 2
    X = sin_spread((B, T, 3 * Cm), 0.0)
    Y = orc.cpu_mha(X, NH)
    HS = Cm // NH
    xd = X.astype(np.float64)
    exp = np.zeros((B, T, Cm))
    for b in range(B):
        for h in range(NH):
            q = xd[b, :, h * HS:(h + 1) * HS]
            k = xd[b, :, Cm + h * HS:Cm + (h + 1) * HS]
            v = xd[b, :, 2 * Cm + h * HS:2 * Cm + (h + 1) * HS]
            for i in range(T):
                s = (k[:i + 1] @ q[i]) / np.sqrt(HS)
                p = np.exp(s - s.max())
                p /= p.sum()
                exp[b, i, h * HS:(h + 1) * HS] = p @ v[:i + 1]
    np.testing.assert_allclose(Y, exp, atol=1e-4, rtol=0)


# ---- Residual / LPE (Connections/Residual.Cpu.cpp, Encodings/Lpe/Lpe.Cpu.cpp) ------------------------
def test_cpu_residual_and_lpe():
    A, Bt = spread((2, 3, 8)), sin_spread((2, 3, 8), 0.7)
    assert np.array_equal(orc.cpu_residual(A, Bt), A + Bt)
    V, Cn, maxT = 11, 8, 6
    wte = sin_spread((V, Cn), 0.1)
    wpe = sin_spread((maxT, Cn), 2.0)
    tok = np.array([[1, 5, 10], [0, 3, 3]], dtype=np.int32)
    Y = orc.cpu_lpe(tok, wte, wpe, out_T=maxT)
    # output rows are strided by the BUILT max length; positions come from the input
    assert np.array_equal(Y[:, :3], wte[tok] + wpe[:3][None])
    assert np.all(Y[:, 3:] == 0)
    with pytest.raises(IndexError):
        orc.cpu_lpe(np.array([[V]], dtype=np.int32), wte, wpe)
    with pytest.raises(IndexError):
        orc.cpu_lpe(np.array([[-1]], dtype=np.int32), wte, wpe)


# ---- RMSNorm (Normalization/RmsNorm/RmsNorm.Cuda.cpp:52-90) -------------------------------------------
def test_rmsnorm_reference_host_formula():
    Cn = 8
    X = spread((2, 3, Cn)) * np.float32(2.0)
    w, b = norm_weight(Cn), norm_bias(Cn)
    Y, rstd = orc.rmsnorm(X, w, b, eps=1e-5, return_rstd=True)
    xd = X.astype(np.float64).reshape(-1, Cn)
    r = 1.0 / np.sqrt((xd ** 2).sum(1, keepdims=True) / Cn + 1e-5)
    exp = xd * r * w + b
    np.testing.assert_allclose(Y.reshape(-1, Cn), exp, atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(rstd, r[:, 0], rtol=1e-6)
    # Gemma usage: offset 0, no bias, eps 1e-6; and the unit-offset variant
    Y2 = orc.rmsnorm(X, w, None, eps=1e-6, w_offset=1.0)
    r2 = 1.0 / np.sqrt((xd ** 2).sum(1, keepdims=True) / Cn + 1e-6)
    np.testing.assert_allclose(Y2.reshape(-1, Cn), xd * r2 * (w.astype(np.float64) + 1.0), atol=1e-6, rtol=1e-6)


def test_rmsnorm_strided_inner_axis():
    X = np.random.default_rng(3).standard_normal((2, 6, 3)).astype(np.float32)
    w = norm_weight(6)
    Y = orc.rmsnorm(X, w, None, eps=1e-5, inner=3)
    xd = X.astype(np.float64)
    r = 1.0 / np.sqrt((xd ** 2).mean(axis=1, keepdims=True) + 1e-5)
    np.testing.assert_allclose(Y, xd * r * w[None, :, None], atol=1e-6, rtol=1e-6)


# ---- RoPE (Encodings/Rope/Rope.Cuda.cpp:51-113; partial-rotary case :290) -----------------------------
@pytest.mark.parametrize("rotary_dim", [0, 4])
def test_rope_reference_host_formula(rotary_dim):
    B, T, H, D, base, off = 2, 5, 2, 8, 10000.0, 3
    X = sin_spread((B, T, H, D), 0.4)
    cos, sin = orc.rope_build_cache(16, D, base, rotary_dim)
    Y = orc.rope_rotate(X, cos, sin, pos_offset=off)
    half = D // 2
    pairs = rotary_dim // 2 if 0 < rotary_dim < D else half
    exp = X.astype(np.float64).copy()
    for t in range(T):
        for i in range(pairs):
            ang = (t + off) * base ** (-2.0 * i / D)
            c, s = np.cos(ang), np.sin(ang)
            x0 = X[:, t, :, i].astype(np.float64)
            x1 = X[:, t, :, i + half].astype(np.float64)
            exp[:, t, :, i] = x0 * c - x1 * s
            exp[:, t, :, i + half] = x0 * s + x1 * c
    np.testing.assert_allclose(Y, exp, atol=1e-3, rtol=1e-3)     # the reference's FP32 bar
    np.testing.assert_allclose(Y, exp, atol=2e-6, rtol=0)        # and far inside it
    if pairs < half:
        assert np.all(cos[:, pairs:] == 1.0) and np.all(sin[:, pairs:] == 0.0)


# ---- GeGLU (FFN/Swiglu/Swiglu.Cuda.cpp:315-345) -------------------------------------------------------
def test_geglu_gate_half_first():
    X = sin_spread((3, 16), 0.2) * np.float32(2.0)
    Y = orc.geglu(X)
    g = X[:, :8].astype(np.float64)
    u = X[:, 8:].astype(np.float64)
    exp = 0.5 * g * (1 + np.tanh(np.sqrt(2 / np.pi) * (g + 0.044715 * g ** 3))) * u
    np.testing.assert_allclose(Y, exp, atol=1e-6, rtol=1e-6)


# ---- FP8 / FP4 formats (Linear.Cuda.cpp:929-951 decoder, :1046 reconstruction bar, :1055-1140) -------
def test_e4m3_codec_exhaustive_roundtrip_and_rounding():
    lut = orc.E4M3_LUT
    assert lut[0x7e] == 448.0 and np.isnan(lut[0x7f]) and lut[0x08] == 2.0 ** -6 and lut[0x01] == 2.0 ** -9
    for code in range(256):
        if (code & 0x7f) == 0x7f:
            continue
        assert orc.lib.orc_f32_to_e4m3(float(lut[code])) == code or lut[code] == 0.0
    # round-to-nearest-even on exact midpoints between neighbours, saturation above 448
    pos = sorted(float(v) for v in lut[:0x7f])
    for lo, hi in zip(pos[:-1], pos[1:]):
        mid = (lo + hi) / 2
        c = orc.lib.orc_f32_to_e4m3(mid)
        assert float(lut[c]) in (lo, hi)
        assert c % 2 == 0, "ties go to the even mantissa"
        assert float(lut[orc.lib.orc_f32_to_e4m3(np.nextafter(np.float32(mid), np.float32(0)))]) == lo
        assert float(lut[orc.lib.orc_f32_to_e4m3(np.nextafter(np.float32(mid), np.float32(1e9)))]) == hi
    for v in (448.0, 449.0, 463.9, 464.0, 1e6, np.inf):
        assert orc.lib.orc_f32_to_e4m3(v) == 0x7e
        assert orc.lib.orc_f32_to_e4m3(-v) == 0xfe
    assert orc.lib.orc_f32_to_e4m3(float("nan")) & 0x7f == 0x7f
    # independent check against torch's float8_e4m3fn cast where it is finite
    torch = pytest.importorskip("torch")
    xs = (np.random.default_rng(1).standard_normal(20000) * 100).astype(np.float32)
    xs = xs[np.abs(xs) < 448]
    tq = torch.from_numpy(xs).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    oq = np.array([orc.lib.orc_f32_to_e4m3(float(v)) for v in xs], dtype=np.uint8)
    assert np.array_equal(tq, oq)


def test_e2m1_thresholds_and_lut():
    assert list(orc.E2M1_LUT[:8]) == [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]
    assert list(orc.E2M1_LUT[8:]) == [-0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0]
    cuts = [0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5.0]
    for i, c in enumerate(cuts):
        assert orc.lib.orc_f32_to_e2m1(c) == i + 1                       # strict '<'
        assert orc.lib.orc_f32_to_e2m1(float(np.nextafter(np.float32(c), np.float32(0)))) == i
        assert orc.lib.orc_f32_to_e2m1(-c) == 8 + i + 1
    assert orc.lib.orc_f32_to_e2m1(-0.0) == 0 and orc.lib.orc_f32_to_e2m1(100.0) == 7


def test_fp8_per_channel_quantize_reference_scenario():
    # Linear.Cuda.cpp:955-1046: quantize the closed-form weight, reconstruct within 0.08|w|+1e-3
    N, K = 32, 64
    Wb = orc.to_bf16_bits(lin_weight(N, K))
    q, s = orc.quantize_fp8_per_channel(Wb)
    W = orc.from_bf16_bits(Wb)
    assert q.shape == (N, K) and s.shape == (N,)
    np.testing.assert_array_equal(s, (np.abs(W).max(1) / np.float32(448.0)).astype(np.float32))
    rec = orc.dequant_fp8(q, s)
    assert np.all(np.abs(rec - W) <= 0.08 * np.abs(W) + 1e-3)
    # an all-zero row gets scale 1 and zero bytes
    Wz = Wb.copy()
    Wz[3] = 0
    qz, sz = orc.quantize_fp8_per_channel(Wz)
    assert sz[3] == 1.0 and not qz[3].any()


def test_fp4_per_group_quantize_reference_scenario():
    # Linear.Cuda.cpp:1055-1140: shapes [N,K/2] and [N,K/128]; low nibble = even column
    N, K, G = 8, 256, 128
    rng = np.random.default_rng(5)
    Wb = orc.to_bf16_bits(rng.standard_normal((N, K)).astype(np.float32) * 0.05)
    q, s = orc.quantize_fp4_per_group(Wb, G)
    assert q.shape == (N, K // 2) and s.shape == (N, K // G)
    W = orc.from_bf16_bits(Wb)
    am = np.abs(W).reshape(N, K // G, G).max(-1)
    np.testing.assert_array_equal(s, (am / np.float32(6.0)).astype(np.float32))
    # nearest-code property: every nibble is a nearest representable of w/scale
    lut = np.array([0, .5, 1, 1.5, 2, 3, 4, 6])
    ratio = W.astype(np.float64) / np.repeat(s, G, axis=1)
    nib = np.empty((N, K), np.uint8)
    nib[:, 0::2] = q & 0xf
    nib[:, 1::2] = q >> 4
    dec = np.where(nib & 8, -1.0, 1.0) * lut[nib & 7]
    best = np.abs(np.abs(ratio)[..., None] - lut).min(-1)
    np.testing.assert_allclose(np.abs(dec - ratio), best, atol=1e-6)
    # the group maximum always lands on +-6
    assert np.all(np.abs(dec).reshape(N, K // G, G).max(-1) == 6.0)
    rec = orc.dequant_fp4(q, s, G)
    np.testing.assert_array_equal(rec, (dec * np.repeat(s, G, axis=1)).astype(np.float32))


def test_quantized_linear_definitions_agree_with_dequantized_matmul():
    rng = np.random.default_rng(11)
    N, K, G = 16, 256, 128
    Wb = orc.to_bf16_bits(rng.standard_normal((N, K)).astype(np.float32) * 0.05)
    x = orc.round_bf16(rng.standard_normal((3, K)).astype(np.float32))
    bias = orc.to_bf16_bits(rng.standard_normal(N).astype(np.float32) * 0.1)
    q8, s8 = orc.quantize_fp8_per_channel(Wb)
    q4, s4 = orc.quantize_fp4_per_group(Wb, G)
    b64 = orc.from_bf16_bits(bias).astype(np.float64)
    y = orc.linear_bf16w(x, Wb, bias)
    np.testing.assert_allclose(y, x.astype(np.float64) @ orc.from_bf16_bits(Wb).astype(np.float64).T + b64, rtol=1e-6, atol=1e-6)
    y8 = orc.linear_fp8w(x, q8, s8, bias)
    np.testing.assert_allclose(y8, x.astype(np.float64) @ orc.dequant_fp8(q8, s8).astype(np.float64).T + b64, rtol=1e-5, atol=1e-5)
    y4 = orc.linear_fp4w(x, q4, s4, G, bias)
    np.testing.assert_allclose(y4, x.astype(np.float64) @ orc.dequant_fp4(q4, s4, G).astype(np.float64).T + b64, rtol=1e-5, atol=1e-5)


# ---- windowed GQA: mask semantics (OPS/Attention/GQA/Kernels/Gqa.Prefill.Bf16.cu:76-81) ---------------
@pytest.mark.parametrize("window", [0, 3])
def test_gqa_attention_mask_and_head_mapping(window):
    rng = np.random.default_rng(2)
    B, T, NH, NKV, HS = 2, 7, 4, 2, 8
    q = rng.standard_normal((B, T, NH, HS)).astype(np.float32)
    k = rng.standard_normal((B, T, NKV, HS)).astype(np.float32)
    v = rng.standard_normal((B, T, NKV, HS)).astype(np.float32)
    out = orc.gqa_attention(q, k, v, 0, window, 0.5).reshape(B, T, NH, HS)
    for b in range(B):
        for t in range(T):
            lo = max(0, t - window + 1) if window > 0 else 0
            for h in range(NH):
                kv = h // (NH // NKV)
                s = (k[b, lo:t + 1, kv].astype(np.float64) @ q[b, t, h].astype(np.float64)) * 0.5
                p = np.exp(s - s.max())
                p /= p.sum()
                np.testing.assert_allclose(out[b, t, h], p @ v[b, lo:t + 1, kv].astype(np.float64), atol=1e-6)
    # decode at position pos == prefill row pos (the two paths define the same key set)
    pos = T - 1
    dec = orc.gqa_attention(q[:, pos:pos + 1], k, v, pos, window, 0.5)
    np.testing.assert_allclose(dec[:, 0], out[:, pos].reshape(B, -1), atol=1e-7)


def test_kv_ring_write_and_readback():
    rng = np.random.default_rng(4)
    B, NKV, HS, cap = 1, 2, 4, 5
    Kc = np.zeros((B, NKV, cap, HS), np.float32)
    Vc = np.zeros_like(Kc)
    hist_k = rng.standard_normal((B, 9, NKV, HS)).astype(np.float32)
    hist_v = rng.standard_normal((B, 9, NKV, HS)).astype(np.float32)
    orc.kv_write(Kc, Vc, hist_k[:, :4], hist_v[:, :4], 0)
    orc.kv_write(Kc, Vc, hist_k[:, 4:9], hist_v[:, 4:9], 4)     # wraps
    lin = orc.kv_ring_to_linear(Kc, 4, 5)                       # the last `cap` positions survive
    assert np.array_equal(lin, hist_k[:, 4:9])
    assert np.array_equal(Kc[0, 1, 8 % cap], hist_k[0, 8, 1])


def test_bf16_rne():
    assert orc.lib.orc_f32_to_bf16(1.0) == 0x3f80
    # 1 + 2^-8 is a tie -> even (1.0); 1 + 3*2^-8 is a tie -> even (1+2^-6)
    assert orc.lib.orc_f32_to_bf16(1.0 + 2.0 ** -8) == 0x3f80
    assert orc.lib.orc_f32_to_bf16(1.0 + 3 * 2.0 ** -8) == 0x3f82
    torch = pytest.importorskip("torch")
    xs = np.random.default_rng(9).standard_normal(50000).astype(np.float32) * 1e3
    ref = torch.from_numpy(xs).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(orc.to_bf16_bits(xs), ref)


def test_stochastic_sampler_reference_scenarios():
    """the restated multinomial sampler reproduces every closed-form expectation of the reference's own tests
    (Tests/Dnn/Samplers/Sampling.Cuda.cpp:152-262, :387-401)"""
    f = lambda v: np.array(v, dtype=np.float32)
    s = lambda lg, **kw: orc.sample_stochastic(lg, kw.get("softcap", 0.0), kw.get("t", 1.0), kw.get("k", 0), kw.get("p", 1.0), kw["r"])[0]
    assert s(f([8, 2, 3, 4, 5, 6, 7, 1]), k=1, r=0.99) == 0                       # TopK1_PicksArgmax
    assert s(f([1, 2, 3, 4, 5, 6, 7, 8]), r=0.0) == 0                             # FullMultinomial_BoundaryR
    assert s(f([1, 2, 3, 4, 5, 6, 7, 8]), r=0.999999) == 7
    assert s(f([1, 5, 2, 8, 3, 6, 4, 7]), t=0.8, r=0.42) == s(f([1, 5, 2, 8, 3, 6, 4, 7]), t=0.8, r=0.42)
    for i in range(20):
        assert s(f([1, 2, 3, 4, 5, 6, 70, 80]), k=2, r=i / 20.0) in (6, 7)        # TopK_RestrictsSupport
        assert s(f([0, 0, 0, 0, 0, 0, 0, 20]), p=0.5, r=i / 20.0) == 7            # TopP_RestrictsSupport
    V = 262144                                                                    # Pipeline_BoundaryR_AtGemmaVocab (uniform logits)
    assert s(np.zeros(V, dtype=np.float32), r=0.0) == 0
    assert s(np.zeros(V, dtype=np.float32), r=0.999999) == V - 1
    # closed form: probabilities .4 .3 .2 .1
    lg = np.log(np.array([0.4, 0.3, 0.2, 0.1])).astype(np.float32)
    assert [s(lg, r=r) for r in (0.1, 0.45, 0.75, 0.95)] == [0, 1, 2, 3]
    assert [s(lg, p=0.65, r=r) for r in (0.1, 0.5, 0.6, 0.99)] == [0, 0, 1, 1]     # nucleus {0, 1}: .7 > .65
    assert [s(lg, k=2, r=r) for r in (0.1, 0.5, 0.6, 0.99)] == [0, 0, 1, 1]
    # a tie across the top-k boundary drops the whole tie (the reference's bisection excludes the (k+1)-th value)
    assert all(s(f([5, 3, 3, 1]), k=2, r=r) == 0 for r in (0.0, 0.5, 0.99))
    # softcap is applied before the temperature: 30 * tanh(1000 / 30) == 30 * tanh(990 / 30) -> a coin flip; without it
    # token 1 has probability e^-10
    assert [s(f([1000, 990]), softcap=30.0, r=r) for r in (0.25, 0.75)] == [0, 1]
    assert [s(f([1000, 990]), r=r) for r in (0.25, 0.75)] == [0, 0]


# ---- net-level pins: the independent implementation the reference validates its checkpoints against ------------------------
def _golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name), allow_pickle=False)


def test_gpt2_forward_matches_the_huggingface_fixture():
    """orc_cpu_gpt2_forward (the restated reference CPU backend, GptTransformer.ixx:221-254) reproduces the FP32 logits of the
    transformers GPT2LMHeadModel on the same parameters (tests/golden/make_gpt2_hf_golden.py): pins the net-level wiring the
    reference's own tests only check for shape and finiteness (GptTransformer.Cpu.cpp:226-255)"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_gpt2_hf_golden as g
    fx = _golden("gpt2_hf_logits.npz")
    V, maxT, C_, L, NH, B, T = [int(v) for v in fx["dims"]]
    assert (V, maxT, C_, L, NH, B, T) == (g.V, g.MAXT, g.C_, g.L, g.NH, g.B, g.T) and int(fx["seed"]) == g.SEED
    assert np.array_equal(fx["tokens"], g.gpt2_tokens())
    got = orc.cpu_gpt2_forward(fx["tokens"], g.gpt2_params(), C_, L, NH, V, maxT)
    exp = fx["logits"]
    assert got.shape == exp.shape == (B, T, V)
    assert np.abs(got - exp).max() <= 1e-4 * np.abs(exp).max(), np.abs(got - exp).max()      # measured 5e-7


def test_gemma4_composition_matches_the_huggingface_fixture():
    """tests/ref_gemma.py with exact=True (no intermediate bf16 rounding) reproduces the FP32 logits of the transformers
    Gemma4ForCausalLM on the same synthetic weights, at prefixes below, at and beyond the sliding window -- as one prefill and
    token by token through the KV history; the bf16-rounding composition the GPU is held to stays within the bf16 bar of it"""
    from ref_gemma import RefGemma
    fx = _golden("gemma4_hf_logits.npz")
    cfg = {str(k): int(v) for k, v in zip(fx["cfg_keys"], fx["cfg_vals"])}
    seed, tokens = int(fx["seed"]), [int(t) for t in fx["tokens"]]
    assert cfg["window"] < max(int(n) for n in fx["prefixes"])            # the window really cuts
    worst = 0.0
    for n, exp in zip(fx["prefixes"], fx["logits"]):
        n = int(n)
        got = RefGemma(cfg, "bf16", seed, exact=True).forward(tokens[:n], 0, 64)
        worst = max(worst, float(np.abs(got - exp).max() / np.abs(exp).max()))
    assert worst <= 1e-4, worst                                           # measured 1e-5
    # decode path: one token at a time, logits after every step
    r = RefGemma(cfg, "bf16", seed, exact=True)
    rb = RefGemma(cfg, "bf16", seed)                                      # every component output rounded to bf16
    by_len = {int(n): e for n, e in zip(fx["prefixes"], fx["logits"])}
    for pos, tok in enumerate(tokens):
        got = r.forward([tok], pos, 64)
        gb = rb.forward([tok], pos, 64)
        if pos + 1 in by_len:
            exp = by_len[pos + 1]
            assert np.abs(got - exp).max() <= 1e-4 * np.abs(exp).max(), (pos, np.abs(got - exp).max())
            # bf16 after every component through 6 random-weight layers: the same 1e-1-of-range bar the GPU path is held to
            # against this composition (tests/test_gemma_host_gpu.py); measured worst 5.7e-2
            assert np.abs(gb - exp).max() <= 1e-1 * np.abs(exp).max(), (pos, np.abs(gb - exp).max())
