"""GPU parity: the W8A8 prefill of the PerChannelFp8<> policy (SURVEY.md section 8 row g1 / BASELINE config 4: "Linear<PerChannelFp8<>> weights, CDNA4 fp8_e4m3 MFMA").

The policy's own e4m3 [N, K] weights + scale[N] are the fp8 matrix cores' operand as they lie in HBM (Quantization/Weight/Policies.ixx:39-40: "FP8 matmul consumes weights and
scales natively -- no dequantization on the forward hot path"); activations are quantized per token exactly as on the reference's W4A8 path (Fp8Prefill/CudaFp8Prefill.cu:108-160).

    y = bf16( (sum_k X8 W8)[m, n] * scale[n] * s_m + bias[n] )        fp32, one rounding

Opt-in: the reference's arithmetic for this policy is W8A16 (CudaLinearOp.ixx:597-644), which stays the default.  Bars:
  * integer steps (weight quantization, activation quantization) bit-exact against the oracle;
  * GEMM + epilogue within 2 bf16 ulp of the float64 composition of the same e4m3 operands (oracle: orc_linear_fp8a_fp8w with per-row weight scales);
  * every kernel form that can serve a row (LDS-DMA tiles, masked tiles, skinny weight stream, split-K) within the same bar, the tile forms bit-identical to one another;
  * within the reference's own activation-quantized bar, 1e-1 x row_absmax (Tests/Dnn/Components/Linear/Linear.Cuda.cpp:760-774), of the W8A16 Linear on the same weights.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import orc
from gpu_util import assert_bf16_close, bits, dev_f32, dev_u16, dev_u8, empty_f32, empty_u16, empty_u8, host
from mila_amd import capi

pytestmark = pytest.mark.gpu


def _operands(rng, M, K, N):
    Wb = orc.to_bf16_bits((rng.standard_normal((N, K)) / np.sqrt(K) * rng.uniform(0.25, 4.0, (N, 1))).astype(np.float32))      # channel scales spread over 4 octaves
    X = orc.round_bf16((rng.standard_normal((M, K)) * rng.uniform(0.2, 3.0, (M, 1))).astype(np.float32))
    if M > 7:
        X[7] = 0.0                                                              # an all-zero token: the 1e-12 guard
    w8, sc = orc.quantize_fp8_per_channel(Wb)
    x8, ts = orc.quantize_act_fp8_per_token(X)
    return Wb, X, w8, sc, x8, ts


def _expected(x8, ts, w8, sc, rows, bb):
    exp = orc.linear_fp8a_fp8w(x8[rows], ts[rows], w8, sc, 1.0, None).astype(np.float64)
    if bb is not None:
        exp = exp + orc.from_bf16_bits(bb).astype(np.float64)
    return exp


def _sample_rows(M):
    main = M - M % 256
    return sorted({0, 1, min(7, M - 1), M // 2, M - 1, max(0, main - 1), min(M - 1, main), min(M - 1, main + 127), min(M - 1, main + 128)})


@pytest.mark.parametrize("M,K,N,bias", [(2048, 256, 8192, False), (512, 384, 30720, True), (2048, 128, 3840, True), (2048, 1280, 3840, False), (2048, 256, 30720, True),
                                        (2, 3840, 8704, False), (16, 512, 256, True), (33, 192, 250, True), (300, 384, 3840, True), (300, 256, 8192, True),
                                        (2000, 256, 8192, False), (2049, 256, 8192, True), (2048 + 77, 128, 3840, False), (1024 + 255, 256, 30720, True)])
def test_w8a8_prefill_matches_the_oracle_at_every_row_count(M, K, N, bias):
    lib = capi.load()
    lib.mila_cdna4_gemm_w8a8_scratch_bytes.restype = C.c_size_t
    assert lib.mila_cdna4_gemm_fp8_applicable(M, K, N) == 1
    rng = np.random.default_rng(M * 11 + N + K)
    Wb, X, w8, sc, x8, ts = _operands(rng, M, K, N)
    bb = orc.to_bf16_bits(rng.uniform(-0.1, 0.1, N).astype(np.float32)) if bias else None
    bd = dev_u16(bb) if bias else None
    # 1. the policy's quantize-on-load and the per-token activation quantization: integer outputs, bit-exact
    W8, SC = empty_u8(N, K), empty_f32(N)
    capi.call("quantize_fp8_per_channel", W8, SC, dev_u16(Wb), N, K)
    assert np.array_equal(host(W8), w8) and np.array_equal(host(SC), sc)
    X8, TS = empty_u8(M, K), empty_f32(M)
    capi.call("quantize_fp8_per_token", X8, TS, dev_u16(orc.to_bf16_bits(X)), M, K)
    assert np.array_equal(host(X8), x8) and np.array_equal(host(TS), ts)
    # 2. GEMM + epilogue on the weights as the policy stores them
    need_ws = lib.mila_cdna4_gemm_fp8_workspace_bytes(M, K, N)
    ws = torch.empty(max(need_ws, 16), dtype=torch.uint8, device="cuda")
    Y = empty_u16(M, N)
    capi.call("gemm_fp8_w8a8_ws", Y, X8, W8, TS, SC, bd, M, K, N, ws, C.c_size_t(need_ws))
    rows = _sample_rows(M)
    exp = _expected(x8, ts, w8, sc, rows, bb)
    # fp32 accumulation order is the only freedom (exact e4m3 products): 2 ulp, and 1e-3 of the output range for outputs that are the difference of large partial sums
    assert_bf16_close(bits(Y)[rows], exp, 2, 1e-3 * float(np.abs(exp).max()), "W8A8 GEMM vs the float64 composition")
    # 3. the one-call form (quantizes the activations itself) gives the same bits
    need = lib.mila_cdna4_gemm_w8a8_scratch_bytes(M, K, N)
    scratch = torch.empty(need, dtype=torch.uint8, device="cuda")
    Y2 = empty_u16(M, N)
    capi.call("gemm_bf16_w8a8", Y2, dev_u16(orc.to_bf16_bits(X)), W8, SC, bd, M, K, N, scratch, C.c_size_t(need))
    assert np.array_equal(bits(Y2), bits(Y))
    with pytest.raises(capi.MilaError):
        capi.call("gemm_bf16_w8a8", Y2, dev_u16(orc.to_bf16_bits(X)), W8, SC, bd, M, K, N, scratch, C.c_size_t(8))
    # 4. every form that can serve a row: 1 = masked 128-row LDS tiles (the LDS-DMA kernels' instruction chain: bit-identical where those serve), 2 = skinny weight stream
    #    (eight interleaved K chains: rounding-level differences)
    forms = {}
    for form in (1, 2):
        Yf = empty_u16(M, N)
        capi.tune("gemm_fp8.tail_form", form)
        try:
            capi.call("gemm_fp8_w8a8_ws", Yf, X8, W8, TS, SC, bd, M, K, N, None, C.c_size_t(0))
        finally:
            capi.tune_reset()
        forms[form] = bits(Yf)
        assert_bf16_close(forms[form][rows], exp, 2, 1e-3 * float(np.abs(exp).max()), "W8A8 GEMM, tail form %d" % form)
    if need_ws == 0:
        tail = M % 256
        tiles = ((M + 255) // 256) * (N // 128)
        if K % 128 or N % 128 or not (M >= 512 or (M > 16 and tiles >= 120)):
            big = 0
        elif M < 512:
            big = M
        else:
            big = M - tail if 0 < tail <= 64 else M
        assert np.array_equal(forms[1][:big], bits(Y)[:big]), "the masked LDS-tile kernel and the LDS-DMA fp8 kernels differ"
    # 5. against the policy's reference arithmetic (W8A16: exact activations on the dequantized weights): the reference's bar for an activation-quantized prefill
    ref16 = orc.linear_fp8w(X[rows], w8, sc).astype(np.float64)
    if bias:
        ref16 = ref16 + orc.from_bf16_bits(bb).astype(np.float64)
    got = orc.from_bf16_bits(bits(Y)[rows]).astype(np.float64)
    assert np.all(np.abs(got - ref16).max(axis=1) <= 1e-1 * np.maximum(np.abs(ref16).max(axis=1), 1e-6))


@pytest.mark.parametrize("M,K,F", [(512, 256, 15360), (2048, 128, 15360), (2, 256, 15360), (100, 384, 1000), (512 + 100, 256, 15360), (2049, 128, 15360), (300, 256, 4096),
                                   (512, 1024, 15360), (2048, 1024, 15360), (512, 1152, 15360)])      # (the last three: the interior K-tile bodies of the fp8 GeGLU mode, even and odd K-tile counts)
def test_w8a8_geglu_form_is_bit_identical_to_linear_then_geglu(M, K, F):
    """fc_gate_up + GeGLU in one kernel (Gemma.Block.ixx:343-348) on the W8A8 path: the gate rows use scale[n], the up rows scale[F + n]; same bits as the Linear over
    [2F, K] followed by geglu_bf16, whichever tile form serves the rows"""
    lib = capi.load()
    lib.mila_cdna4_gemm_w8a8_scratch_bytes.restype = C.c_size_t
    assert lib.mila_cdna4_gemm_geglu_w4a8_applicable(M, K, F) == 1
    rng = np.random.default_rng(M + F)
    Wb, X, w8, sc, x8, ts = _operands(rng, M, K, 2 * F)
    W8, SC, X8, TS = dev_u8(w8), dev_f32(sc), dev_u8(x8), dev_f32(ts)
    need_ws = lib.mila_cdna4_gemm_fp8_workspace_bytes(M, K, 2 * F)
    ws = torch.empty(max(need_ws, 16), dtype=torch.uint8, device="cuda")
    GU, Y0, Y1, Y2 = empty_u16(M, 2 * F), empty_u16(M, F), empty_u16(M, F), empty_u16(M, F)
    capi.call("gemm_fp8_w8a8_ws", GU, X8, W8, TS, SC, None, M, K, 2 * F, ws, C.c_size_t(need_ws))
    capi.call("geglu_bf16", Y0, GU, M, F)
    capi.call("gemm_geglu_fp8_w8a8", Y1, X8, W8, TS, SC, M, K, F)
    assert np.array_equal(bits(Y0), bits(Y1))
    need = lib.mila_cdna4_gemm_w8a8_scratch_bytes(M, K, 2 * F)
    scratch = torch.empty(need, dtype=torch.uint8, device="cuda")
    capi.call("gemm_geglu_bf16_w8a8", Y2, dev_u16(orc.to_bf16_bits(X)), W8, SC, M, K, F, scratch, C.c_size_t(need))
    assert np.array_equal(bits(Y2), bits(Y1))
    # the float64 composition on sampled rows: gate / up rounded to bf16 as the Linear stores them, GeGLU in double
    rows = _sample_rows(M)
    gu = orc.round_bf16(orc.linear_fp8a_fp8w(x8[rows], ts[rows], w8, sc, 1.0, None).astype(np.float32))
    exp = orc.geglu(gu)
    assert_bf16_close(bits(Y1)[rows], exp, 2, 2e-3 * float(np.abs(exp).max()), "W8A8 Linear + GeGLU vs the float64 composition")
    for form in (1, 2):
        capi.tune("gemm_fp8.tail_form", form)
        try:
            capi.call("gemm_geglu_fp8_w8a8", Y2, X8, W8, TS, SC, M, K, F)
            capi.call("gemm_fp8_w8a8_ws", GU, X8, W8, TS, SC, None, M, K, 2 * F, None, C.c_size_t(0))
            capi.call("geglu_bf16", Y0, GU, M, F)
        finally:
            capi.tune_reset()
        assert np.array_equal(bits(Y2), bits(Y0)), "tail form %d: fused GeGLU epilogue != Linear + GeGLU" % form


@pytest.mark.parametrize("M,K,N", [(300, 3840, 3840), (100, 15360, 3840), (2303, 4096, 3840)])
def test_w8a8_split_k_through_the_workspace(M, K, N):
    """short prompts and long-prompt remainders split K through the caller's workspace (the W4A8 forms' rule): fixed-order sum of the fp32 partials, the same epilogue"""
    lib = capi.load()
    need_ws = lib.mila_cdna4_gemm_fp8_workspace_bytes(M, K, N)
    assert need_ws > 0
    rng = np.random.default_rng(M + K)
    Wb, X, w8, sc, x8, ts = _operands(rng, M, K, N)
    bb = orc.to_bf16_bits(rng.uniform(-0.1, 0.1, N).astype(np.float32))
    W8, SC, X8, TS = dev_u8(w8), dev_f32(sc), dev_u8(x8), dev_f32(ts)
    ws = torch.empty(need_ws, dtype=torch.uint8, device="cuda")
    Y, Y2 = empty_u16(M, N), empty_u16(M, N)
    capi.call("gemm_fp8_w8a8_ws", Y, X8, W8, TS, SC, dev_u16(bb), M, K, N, ws, C.c_size_t(need_ws))
    capi.call("gemm_fp8_w8a8_ws", Y2, X8, W8, TS, SC, dev_u16(bb), M, K, N, ws, C.c_size_t(need_ws))
    assert np.array_equal(bits(Y), bits(Y2)), "split-K is not deterministic"
    rows = _sample_rows(M)
    exp = _expected(x8, ts, w8, sc, rows, bb)
    assert_bf16_close(bits(Y)[rows], exp, 2, 1e-3 * float(np.abs(exp).max()), "W8A8 split-K vs the float64 composition")
    with pytest.raises(capi.MilaError):
        capi.call("gemm_fp8_w8a8_ws", Y2, X8, W8, TS, SC, dev_u16(bb), M, K, N, ws, C.c_size_t(need_ws - 16))
