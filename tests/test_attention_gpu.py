"""GPU parity: KV-cache append, split-K flash-decode, prefill attention and GPT-2 MHA against the
float64 oracle, through the C ABI.  Geometries follow the reference's own fused-decode parity tests
(Tests/Dnn/Components/Attention/GQA/CudaGqaOp.Cuda.cpp:1080-1110: Gemma-global NH16/NKV1/HS512,
Gemma-local NH16/NKV8/HS256/window, Llama NH32/NKV8/HS128); the reference's bar is 3e-2 abs on
uniform(-1,1) data (:61-75,:611), this build's is <= 1 bf16 ulp + 2e-3 abs vs float64.
Dead zones of the cache are poisoned with NaN (the reference's out-of-tree harness does the same,
Docs/Discussions/DecodePerformanceCampaign.md:95-106): a kernel that touches a row outside the live
band produces NaN and fails."""
import numpy as np
import pytest
import torch

import orc
from gpu_util import assert_bf16_close, bits, dev_u16, empty_u16, host
from mila_amd import capi

pytestmark = pytest.mark.gpu

NAN_BITS = 0x7fc0


def _bf(x):
    return orc.round_bf16(np.asarray(x, dtype=np.float32))


def _d(x):
    return dev_u16(orc.to_bf16_bits(x))


def _poisoned_cache(B, NKV, cap, HS):
    return torch.full((B, NKV, cap, HS), NAN_BITS, dtype=torch.int16, device="cuda")


def _fill_cache(hist_k, hist_v, cap, chunk):
    """append the whole history through the kv_write entry point in chunks; returns device caches"""
    B, T, NKV, HS = hist_k.shape
    Kc, Vc = _poisoned_cache(B, NKV, cap, HS), _poisoned_cache(B, NKV, cap, HS)
    for s in range(0, T, chunk):
        e = min(T, s + chunk)
        capi.call("kv_write_bf16", Kc, Vc, _d(hist_k[:, s:e]), _d(hist_v[:, s:e]), B, e - s, NKV, HS, s, cap)
    return Kc, Vc


def test_kv_write_ring_layout_bit_exact():
    rng = np.random.default_rng(0)
    B, NKV, HS, cap, T = 2, 2, 64, 10, 23
    hk, hv = _bf(rng.standard_normal((B, T, NKV, HS))), _bf(rng.standard_normal((B, T, NKV, HS)))
    Kc, Vc = _fill_cache(hk, hv, cap, 7)
    eK = np.zeros((B, NKV, cap, HS), np.float32)
    eV = np.zeros_like(eK)
    for s in range(0, T, 7):
        orc.kv_write(eK, eV, hk[:, s:s + 7], hv[:, s:s + 7], s)
    assert np.array_equal(bits(Kc), orc.to_bf16_bits(eK)) and np.array_equal(bits(Vc), orc.to_bf16_bits(eV))
    with pytest.raises(capi.InvalidArgument):
        capi.call("kv_write_bf16", Kc, Vc, _d(hk), _d(hv), B, T, NKV, HS, 0, cap)    # chunk > capacity


GEOMS = [
    # name, NH, NKV, HS, window, scale
    ("gemma_local", 16, 8, 256, 1024, 1.0),
    ("gemma_global", 16, 1, 512, 0, 1.0),
    ("llama", 32, 8, 128, 0, 128 ** -0.5),
    ("gpt2", 12, 12, 64, 0, 0.125),
    ("mqa_small_window", 4, 1, 64, 5, 0.3),
]


@pytest.mark.parametrize("name,NH,NKV,HS,window,scale", GEOMS)
@pytest.mark.parametrize("length", [1, 2, 37, 300, 1500])
def test_decode_attention(name, NH, NKV, HS, window, scale, length):
    rng = np.random.default_rng(length + HS)
    B = 2 if HS <= 128 else 1
    # unbounded cache (capacity == context) with NaN beyond `length` and before the band
    cap = 2048
    hk = _bf(rng.uniform(-1, 1, (B, length, NKV, HS)) * 0.5)
    hv = _bf(rng.uniform(-1, 1, (B, length, NKV, HS)))
    q = _bf(rng.uniform(-1, 1, (B, 1, NH, HS)))
    Kc, Vc = _fill_cache(hk, hv, cap, 512)
    if window > 0 and length > window:      # rows older than the band must never be read
        Kc[:, :, : length - window] = NAN_BITS
        Vc[:, :, : length - window] = NAN_BITS
    Y = empty_u16(B, NH * HS)
    nbytes = capi.load().mila_cdna4_attn_decode_scratch_bytes(B, NH, HS)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    import ctypes as C
    capi.call("attn_decode_bf16", Y, _d(q), Kc, Vc, scratch, C.c_size_t(nbytes), B, NH, NKV, HS, cap, length, window,
              float(scale))
    exp = orc.gqa_attention(q, hk, hv, length - 1, window, scale)[:, 0]
    assert_bf16_close(bits(Y), exp, 1, 2e-3, "decode %s len %d" % (name, length))
    # the XCD-local grid (experiment, attn.xcd_local: splits of a head group and their combine workgroups share blockIdx.x % 8) is the same arithmetic per workgroup
    Y2 = empty_u16(B, NH * HS)
    capi.tune("attn.xcd_local", 1)
    try:
        capi.call("attn_decode_bf16", Y2, _d(q), Kc, Vc, scratch, C.c_size_t(nbytes), B, NH, NKV, HS, cap, length, window, float(scale))
    finally:
        capi.tune_reset()
    assert np.array_equal(bits(Y2), bits(Y)), "xcd-local grid changed bits"


@pytest.mark.parametrize("window,cap,length", [(8, 8, 30), (16, 23, 100), (1024, 1100, 2600)])
def test_decode_attention_bounded_ring_equals_unbounded(window, cap, length):
    """the reference's ring-vs-unbounded oracle test (CudaGqaOp.Cuda.cpp:529-567)"""
    rng = np.random.default_rng(cap)
    B, NH, NKV, HS = 1, 16, 8, 256
    hk = _bf(rng.uniform(-1, 1, (B, length, NKV, HS)) * 0.5)
    hv = _bf(rng.uniform(-1, 1, (B, length, NKV, HS)))
    q = _bf(rng.uniform(-1, 1, (B, 1, NH, HS)))
    Kc, Vc = _fill_cache(hk, hv, cap, min(cap, 7) if cap < 64 else 64)
    Y = empty_u16(B, NH * HS)
    import ctypes as C
    nbytes = capi.load().mila_cdna4_attn_decode_scratch_bytes(B, NH, HS)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    capi.call("attn_decode_bf16", Y, _d(q), Kc, Vc, scratch, C.c_size_t(nbytes), B, NH, NKV, HS, cap, length, window, 1.0)
    exp = orc.gqa_attention(q, hk, hv, length - 1, window, 1.0)[:, 0]
    assert_bf16_close(bits(Y), exp, 1, 2e-3, "ring decode")
    with pytest.raises(capi.InvalidArgument):     # band larger than the ring
        capi.call("attn_decode_bf16", Y, _d(q), Kc, Vc, scratch, C.c_size_t(nbytes), B, NH, NKV, HS, cap, length, 0, 1.0)


def test_decode_online_softmax_rescale_branch_is_exercised():
    """spike one key so the running max jumps late in the band (cdna guide rule 26)"""
    rng = np.random.default_rng(9)
    B, NH, NKV, HS, length = 1, 16, 1, 512, 777
    hk = _bf(rng.uniform(-1, 1, (B, length, NKV, HS)) * 0.1)
    hv = _bf(rng.uniform(-1, 1, (B, length, NKV, HS)))
    q = _bf(rng.uniform(-1, 1, (B, 1, NH, HS)))
    hk[0, 700, 0] = _bf(q[0, 0, 3] * 0.5)           # large positive score for head 3 at position 700
    hk[0, 5, 0] = _bf(q[0, 0, 7] * 0.5)
    Kc, Vc = _fill_cache(hk, hv, 1024, 256)
    Y = empty_u16(B, NH * HS)
    import ctypes as C
    nbytes = capi.load().mila_cdna4_attn_decode_scratch_bytes(B, NH, HS)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    capi.call("attn_decode_bf16", Y, _d(q), Kc, Vc, scratch, C.c_size_t(nbytes), B, NH, NKV, HS, 1024, length, 0, 1.0)
    assert_bf16_close(bits(Y), orc.gqa_attention(q, hk, hv, length - 1, 0, 1.0)[:, 0], 1, 2e-3, "spiked decode")


# every workgroup shape of the LDS-DMA flash kernels: heads per workgroup follow the group size (4 | GS, 2 | GS, odd), at HS = 512 with the d-split
FLASH_FORM_GEOMS = [
    ("hs256_gs4", 8, 2, 256, 40, 1.0),
    ("hs256_gs1", 2, 2, 256, 0, 0.5),
    ("hs512_gs2", 4, 2, 512, 0, 1.0),
    ("hs512_gs1", 2, 2, 512, 16, 1.0),
    ("hs128_gs1", 2, 2, 128, 0, 1.0),
]


@pytest.mark.parametrize("name,NH,NKV,HS,window,scale", GEOMS + FLASH_FORM_GEOMS)
def test_prefill_attention_chunked(name, NH, NKV, HS, window, scale):
    """two chunks with a position offset; local window smaller than the history"""
    rng = np.random.default_rng(HS + NH)
    B, T = 1, 70
    win = min(window, 24) if window > 0 else 0
    cap = 128
    hk = _bf(rng.uniform(-1, 1, (B, T, NKV, HS)) * 0.5)
    hv = _bf(rng.uniform(-1, 1, (B, T, NKV, HS)))
    q = _bf(rng.uniform(-1, 1, (B, T, NH, HS)))
    Kc, Vc = _poisoned_cache(B, NKV, cap, HS), _poisoned_cache(B, NKV, cap, HS)
    Y = empty_u16(B, T, NH * HS)
    for s, e in ((0, 41), (41, 70)):
        capi.call("kv_write_bf16", Kc, Vc, _d(hk[:, s:e]), _d(hv[:, s:e]), B, e - s, NKV, HS, s, cap)
        Yc = empty_u16(B, e - s, NH * HS)
        capi.call("attn_prefill_bf16", Yc, _d(q[:, s:e]), Kc, Vc, B, e - s, NH, NKV, HS, cap, s, win, float(scale))
        Y[:, s:e] = Yc
    exp = orc.gqa_attention(q, hk, hv, 0, win, scale)
    assert_bf16_close(bits(Y), exp, 1, 2e-3, "prefill %s" % name)
    # decode at the last position reproduces the last prefill row (same key set by definition)
    import ctypes as C
    nbytes = capi.load().mila_cdna4_attn_decode_scratch_bytes(B, NH, HS)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    Yd = empty_u16(B, NH * HS)
    capi.call("attn_decode_bf16", Yd, _d(q[:, T - 1]), Kc, Vc, scratch, C.c_size_t(nbytes), B, NH, NKV, HS, cap, T, win,
              float(scale))
    assert_bf16_close(bits(Yd), exp[:, T - 1], 1, 2e-3, "decode == last prefill row")


@pytest.mark.parametrize("name,NH,NKV,HS,window", [("gemma_local", 16, 8, 256, 1024), ("gemma_global", 16, 1, 512, 0)])
def test_flash_prefill_forms_give_the_same_bits(name, NH, NKV, HS, window):
    """the tuning forms of the flash prefill kernel (9: 8-wave workgroups at HS 256 too -- taken where half the chunk sees a full window --, 2: HS 512 as 4-wave
    d-split workgroups, 1: the register-staged kernels) run the same operations per output element as the default form 8: identical outputs"""
    rng = np.random.default_rng(HS)
    B, off, T = 1, 1024, 256                                  # a chunk behind 1024 cached positions: every row of a window-1024 layer sees a full window
    cap = off + T
    K = dev_u16(orc.to_bf16_bits(_bf(rng.uniform(-1, 1, (B, NKV, cap, HS)) * 0.5)))
    V = dev_u16(orc.to_bf16_bits(_bf(rng.uniform(-1, 1, (B, NKV, cap, HS)))))
    q = _d(_bf(rng.uniform(-1, 1, (B, T, NH, HS))))
    lib = capi.load()
    outs = {}
    try:
        for form in (8, 9, 10, 11, 2, 1):
            capi.tune("flash.form", form)
            Y = empty_u16(B, T, NH * HS)
            capi.call("attn_prefill_bf16", Y, q, K, V, B, T, NH, NKV, HS, cap, off, window, 1.0)
            outs[form] = bits(Y).copy()
    finally:
        capi.tune_reset()
    for form in (9, 10, 11, 2, 1):
        assert np.array_equal(outs[8], outs[form]), "form %d differs from the default" % form


@pytest.mark.parametrize("B,T,Cm,NH", [(2, 3, 8 * 8, 1), (2, 33, 768, 12), (1, 130, 256, 4)])
def test_mha_packed_qkv_vs_reference_cpu_op(B, T, Cm, NH):
    """GPT-2 attention vs the restated CpuAttentionOp (MultiHeadAttention.Cpu.cpp scenario: sin spread)"""
    n = B * T * 3 * Cm
    X = _bf(np.sin(np.float32(0.2) * np.arange(n, dtype=np.float32)).reshape(B, T, 3 * Cm))
    Y = empty_u16(B, T, Cm)
    capi.call("mha_bf16", Y, _d(X), B, T, Cm, NH)
    exp = orc.cpu_mha(X, NH)
    assert_bf16_close(bits(Y), exp, 1, 2e-3, "mha")
    got = orc.from_bf16_bits(bits(Y)).reshape(exp.shape)
    assert np.abs(got - exp).max() <= 3e-2          # the reference's BF16 attention bar
