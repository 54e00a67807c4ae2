"""GPU parity of the FP32 rows (csrc/fp32_rows.hip; OPS/OperationTraits.Cuda.ixx:50-54, :108-126, :274-282: the reference keeps FP32 paths of Linear / MHA / LPE / RoPE "for
validation and reference") against the reference's OWN CPU ops, restated line by line in oracle/mila_oracle.c (CpuLinearOp.ixx:384-456, CpuAttentionOp.ixx:310-460,
CpuEncoderOp.ixx:255-330) -- at the reference's FP32 tolerances: Linear 1e-4 absolute on its closed-form scenario (Linear.Cpu.cpp:247-358) and 1e-3 + 1e-4 |y| on
model-sized rows, MHA 1e-4 (MultiHeadAttention.Cpu.cpp:180-203), LPE exact, RoPE 1e-3 (Rope.Cuda.cpp:51-113)."""
import ctypes as C

import numpy as np
import pytest
import torch

import orc
from gpu_util import dev_f32, dev_i32, empty_f32, host
from mila_amd import capi
from test_oracle_kats import lin_bias, lin_weight, sin_spread, spread

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,bias", [((2, 3, 4), True), ((2, 3, 4), False), ((16, 4), True)])
def test_linear_fp32_reference_scenario__Linear_Cpu_cpp_247(shape, bias):
    in_f, out_f = 4, 3
    W, b, X = lin_weight(out_f, in_f), (lin_bias(out_f) if bias else None), spread(shape)
    M = X.size // in_f
    Y = empty_f32(M, out_f)
    capi.call("gemm_fp32", Y, dev_f32(X), dev_f32(W), dev_f32(b) if bias else None, M, in_f, out_f, 0)
    exp = orc.cpu_linear(X, W, b).reshape(M, out_f)
    np.testing.assert_allclose(host(Y), exp, atol=1e-4, rtol=0)
    y1 = empty_f32(out_f)
    capi.call("matvec_fp32", y1, dev_f32(X.reshape(M, in_f)[0]), dev_f32(W), dev_f32(b) if bias else None, in_f, out_f)
    np.testing.assert_allclose(host(y1), exp[0], atol=1e-4, rtol=0)


@pytest.mark.parametrize("M,K,N,bias,act", [(1, 768, 2304, True, 0), (64, 768, 3072, True, 1), (64, 3072, 768, True, 0), (37, 100, 50257 // 64, False, 0), (8, 3840, 512, False, 0)])
def test_linear_fp32_model_sized_rows_against_the_reference_cpu_op(M, K, N, bias, act):
    """GPT-2's shapes (and an odd one): the restated CpuLinearOp (long double accumulation on the naive path, float seeded with the bias on the unrolled one) is the
    reference; FP32 Linear bar 1e-3 + 1e-4 |y|, measured far inside it"""
    rng = np.random.default_rng(M + K + N)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, N).astype(np.float32) if bias else None
    exp = orc.cpu_linear(X, W, b)
    if act:
        exp = orc.cpu_gelu(exp)
    Y = empty_f32(M, N)
    if M == 1:
        capi.call("matvec_fp32", Y, dev_f32(X), dev_f32(W), dev_f32(b) if bias else None, K, N)
    else:
        capi.call("gemm_fp32", Y, dev_f32(X), dev_f32(W), dev_f32(b) if bias else None, M, K, N, act)
    got = host(Y)
    assert np.all(np.abs(got - exp) <= 1e-3 + 1e-4 * np.abs(exp))
    assert np.abs(got - exp).max() <= 2e-5 * max(1.0, np.abs(exp).max())
    with pytest.raises(capi.MilaError):
        capi.call("gemm_fp32", Y, dev_f32(X), dev_f32(W), None, M, K, N, 7)


def test_mha_fp32_reference_scenario__MultiHeadAttention_Cpu_cpp_180():
    B, T, Cm, NH = 2, 3, 8, 2
    X = sin_spread((B, T, 3 * Cm), 0.0)
    Y = empty_f32(B, T, Cm)
    capi.call("mha_fp32", Y, dev_f32(X), B, T, Cm, NH)
    np.testing.assert_allclose(host(Y), orc.cpu_mha(X, NH), atol=1e-4, rtol=0)


@pytest.mark.parametrize("B,T,Cm,NH", [(2, 64, 768, 12), (1, 300, 256, 4), (3, 17, 96, 3)])
def test_mha_fp32_and_its_kv_cache_session_against_the_reference_cpu_op(B, T, Cm, NH):
    """forward on the packed rows, then the KV-cache session the reference's op offers (CudaMhaOp.ixx:145-380): prefill T - 1 rows into the caches, decode row T - 1"""
    rng = np.random.default_rng(T + Cm)
    X = rng.standard_normal((B, T, 3 * Cm)).astype(np.float32)
    exp = orc.cpu_mha(X, NH)
    Xd = dev_f32(X)
    Y = empty_f32(B, T, Cm)
    capi.call("mha_fp32", Y, Xd, B, T, Cm, NH)
    np.testing.assert_allclose(host(Y), exp, atol=1e-4, rtol=0)
    HS, cap = Cm // NH, T + 3
    Kc = torch.full((B, NH, cap, HS), float("nan"), device="cuda")
    Vc = torch.full((B, NH, cap, HS), float("nan"), device="cuda")
    capi.call("mha_kv_write_fp32", Kc, Vc, dev_f32(X[:, :T - 1]), B, T - 1, Cm, NH, 0, cap)
    assert np.array_equal(host(Kc)[:, :, :T - 1], X[:, :T - 1, Cm:2 * Cm].reshape(B, T - 1, NH, HS).transpose(0, 2, 1, 3))
    Y1 = empty_f32(B, Cm)
    capi.call("mha_decode_fp32", Y1, dev_f32(X[:, T - 1]), Kc, Vc, B, Cm, NH, cap, T - 1)
    np.testing.assert_allclose(host(Y1), exp[:, T - 1], atol=1e-4, rtol=0)
    with pytest.raises(capi.MilaError):
        capi.call("mha_decode_fp32", Y1, dev_f32(X[:, T - 1]), Kc, Vc, B, Cm, NH, cap, cap)      # position out of range (CudaMhaOp.ixx:262-265)


def test_lpe_fp32_is_exact_and_flags_bad_ids__Lpe_Cpu_cpp():
    V, Cn, maxT = 11, 8, 6
    wte, wpe = sin_spread((V, Cn), 0.1), sin_spread((maxT, Cn), 2.0)
    tok = np.array([[1, 5, 10], [0, 3, 3]], dtype=np.int32)
    Y = torch.zeros((2, maxT, Cn), dtype=torch.float32, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    capi.call("lpe_fp32", Y, dev_i32(tok), dev_f32(wte), dev_f32(wpe), 2, 3, Cn, maxT, V, flag)
    assert np.array_equal(host(Y), orc.cpu_lpe(tok, wte, wpe, out_T=maxT)) and int(host(flag)[0]) == 0
    capi.call("lpe_fp32", Y, dev_i32(np.array([[1, V, 2]], dtype=np.int32)), dev_f32(wte), dev_f32(wpe), 1, 3, Cn, maxT, V, flag)
    assert int(host(flag)[0]) == 2                                            # 1-based flat position of the bad id (CpuEncoderOp.ixx: index out of vocabulary range)


@pytest.mark.parametrize("rotary_dim", [0, 4])
def test_rope_fp32__Rope_Cuda_cpp_51(rotary_dim):
    B, T, H, D, base, off = 2, 5, 2, 8, 10000.0, 3
    X = sin_spread((B, T, H, D), 0.4)
    cos, sin = empty_f32(16, D // 2), empty_f32(16, D // 2)
    capi.call("rope_build_cache", cos, sin, 16, D, float(base), rotary_dim)
    Xd = dev_f32(X)
    capi.call("rope_forward_fp32", Xd, None, Xd, None, cos, sin, B, T, H, 1, D, off, 16)      # in place, as the component rotates (Rope.ixx:107)
    ocos, osin = orc.rope_build_cache(16, D, base, rotary_dim)
    np.testing.assert_allclose(host(Xd), orc.rope_rotate(X, ocos, osin, pos_offset=off), atol=1e-3, rtol=1e-3)
    np.testing.assert_allclose(host(Xd), orc.rope_rotate(X, ocos, osin, pos_offset=off), atol=2e-6, rtol=0)
