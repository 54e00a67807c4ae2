"""GPU parity: the fused decode-step kernels reproduce the unfused chain of entry points bit for
bit (every intermediate bf16 rounding included) and agree with the oracle.
Chain restated from GemmaBlock::decode (Components/Transformers/Gemma/Gemma.Block.ixx:287-356)."""
import ctypes as C

import numpy as np
import pytest
import torch

import orc
from gpu_util import assert_bf16_close, bits, dev_f32, dev_u16, dev_u8, empty_f32, empty_u16
from mila_amd import capi

pytestmark = pytest.mark.gpu


def _bf(x):
    return orc.round_bf16(np.asarray(x, dtype=np.float32))


def _d(x):
    return dev_u16(orc.to_bf16_bits(x))


def _weights(rng, N, K, fmt, G=128):
    Wb = orc.to_bf16_bits((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    if fmt == 0:
        return dev_u16(Wb), None, Wb
    if fmt == 1:
        q, s = orc.quantize_fp8_per_channel(Wb)
        return dev_u8(q), dev_f32(s), (q, s)
    q, s = orc.quantize_fp4_per_group(Wb, G)
    return dev_u8(q), dev_f32(s), (q, s)


def _matvec(fmt, y, x, W, s, K, N, G=128):
    if fmt == 0:
        capi.call("matvec_bf16", y, x, W, None, K, N)
    elif fmt == 1:
        capi.call("matvec_bf16_qfp8", y, x, W, s, None, K, N)
    else:
        capi.call("matvec_bf16_qfp4", y, x, W, s, None, K, N, G)


def _args(**kw):
    a = capi.fused_matvec_args()
    for k, v in kw.items():
        if hasattr(v, "data_ptr"):
            v = v.data_ptr()
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("fmt", [0, 1, 2])
@pytest.mark.parametrize("K,N", [(3840, 520), (4096, 129)])
def test_norm_matvec_equals_unfused_chain(fmt, K, N):
    rng = np.random.default_rng(fmt + K)
    x = _d(_bf(rng.standard_normal(K) * 2))
    nw = _d(_bf(1 + 0.1 * rng.uniform(-1, 1, K)))
    W, s, _ = _weights(rng, N, K, fmt)
    # unfused: rmsnorm -> matvec
    xn, y0 = empty_u16(K), empty_u16(N)
    capi.call("rmsnorm_bf16", xn, None, x, nw, None, 1, K, 1, 1e-6, 0.0)
    _matvec(fmt, y0, xn, W, s, K, N)
    y1 = empty_u16(N)
    a = _args(y=y1, x=x, W=W, scales=s if s is not None else 0, norm_w=nw, post_w=0, res=0, res_out=0,
              post_scale=1.0, eps=1e-6, fmt=fmt, K=K, N=N, group=128, geglu=0)
    capi.check(capi.load().mila_cdna4_fused_norm_matvec(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert np.array_equal(bits(y0), bits(y1))


@pytest.mark.parametrize("fmt", [0, 1, 2])
@pytest.mark.parametrize("post_scale", [1.0, 0.75])
def test_sandwich_tail_geglu_matvec_equals_unfused_chain(fmt, post_scale):
    """x = o_proj out; a = post_norm(x); r = res + a (* scalar); h = pre_ffn_norm(r);
    gate_up = Linear(h); y = geglu(gate_up)   (Gemma.Block.ixx:343-348)"""
    rng = np.random.default_rng(fmt + 11)
    K, H = 3840, 264                     # N = H outputs, weight rows 2H
    x = _d(_bf(rng.standard_normal(K)))
    res = _d(_bf(rng.standard_normal(K)))
    pw = _d(_bf(1 + 0.1 * rng.uniform(-1, 1, K)))
    nw = _d(_bf(1 + 0.1 * rng.uniform(-1, 1, K)))
    W, s, _ = _weights(rng, 2 * H, K, fmt)
    a_, r_, h_, gu, y0 = empty_u16(K), empty_u16(K), empty_u16(K), empty_u16(2 * H), empty_u16(H)
    capi.call("rmsnorm_bf16", a_, None, x, pw, None, 1, K, 1, 1e-6, 0.0)
    capi.call("residual_bf16", r_, res, a_, C.c_int64(K))
    if post_scale != 1.0:
        capi.call("scale_bf16", r_, r_, C.c_int64(K), post_scale)
    capi.call("rmsnorm_bf16", h_, None, r_, nw, None, 1, K, 1, 1e-6, 0.0)
    _matvec(fmt, gu, h_, W, s, K, 2 * H)
    capi.call("geglu_bf16", y0, gu, 1, H)
    y1, r1 = empty_u16(H), empty_u16(K)
    a = _args(y=y1, x=x, W=W, scales=s if s is not None else 0, norm_w=nw, post_w=pw, res=res, res_out=r1,
              post_scale=post_scale, eps=1e-6, fmt=fmt, K=K, N=H, group=128, geglu=1)
    capi.check(capi.load().mila_cdna4_fused_norm_matvec(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert np.array_equal(bits(r_), bits(r1)), "residual stream differs"
    assert np.array_equal(bits(y0), bits(y1)), "geglu output differs"


@pytest.mark.parametrize("NH,NKV,HS,rot,base,kv_shared", [(16, 8, 256, 0, 1e4, False), (16, 1, 512, 128, 1e6, True)])
def test_qkv_post_equals_unfused_chain_and_oracle(NH, NKV, HS, rot, base, kv_shared):
    rng = np.random.default_rng(HS)
    cap, pos, max_seq = 64, 77, 128
    q = _bf(rng.standard_normal((NH, HS)))
    k = _bf(rng.standard_normal((NKV, HS)))
    v = k if kv_shared else _bf(rng.standard_normal((NKV, HS)))
    qw, kw = _bf(1 + 0.1 * rng.uniform(-1, 1, HS)), _bf(1 + 0.1 * rng.uniform(-1, 1, HS))
    cos, sin = empty_f32(max_seq, HS // 2), empty_f32(max_seq, HS // 2)
    capi.call("rope_build_cache", cos, sin, max_seq, HS, float(base), rot)
    # unfused chain
    qn, kn, vn = empty_u16(NH, HS), empty_u16(NKV, HS), empty_u16(NKV, HS)
    capi.call("rmsnorm_bf16", qn, None, _d(q), _d(qw), None, NH, HS, 1, 1e-6, 0.0)
    capi.call("rmsnorm_bf16", kn, None, _d(k), _d(kw), None, NKV, HS, 1, 1e-6, 0.0)
    capi.call("rmsnorm_bf16", vn, None, _d(v), None, None, NKV, HS, 1, 1e-6, 0.0)
    capi.call("rope_forward_bf16", qn, kn, qn, kn, cos, sin, 1, 1, NH, NKV, HS, pos, max_seq)
    K0 = torch.zeros((1, NKV, cap, HS), dtype=torch.int16, device="cuda")
    V0 = torch.zeros_like(K0)
    capi.call("kv_write_bf16", K0, V0, kn, vn, 1, 1, NKV, HS, pos, cap)
    # fused
    K1, V1, q1 = torch.zeros_like(K0), torch.zeros_like(K0), empty_u16(NH, HS)
    capi.call("fused_qkv_post", q1, K1, V1, _d(q), _d(k), _d(v), _d(qw), _d(kw), None, cos, sin, NH, NKV, HS, pos, cap, 1e-6)
    assert np.array_equal(bits(q1), bits(qn))
    assert np.array_equal(bits(K1), bits(K0)) and np.array_equal(bits(V1), bits(V0))
    # oracle: norm -> (bf16) -> rope with the device cache
    from gpu_util import host
    qe = orc.rope_rotate(_bf(orc.rmsnorm(q, qw, None, eps=1e-6)).reshape(1, 1, NH, HS), host(cos), host(sin), pos)
    assert_bf16_close(bits(q1), qe, 1, 1e-30, "fused q vs oracle")


@pytest.mark.parametrize("NH,NKV,HS,rot,base,window,kv_shared", [(16, 8, 256, 0, 1e4, 1024, False), (16, 1, 512, 128, 1e6, 0, True),
                                                                 (4, 2, 64, 0, 1e4, 8, False)])
@pytest.mark.parametrize("pos", [0, 5, 63, 64, 200, 1500])
def test_fused_attn_decode_equals_qkv_post_plus_attn_decode(NH, NKV, HS, rot, base, window, kv_shared, pos):
    """one launch (prologue: q/k/v norm + RoPE + KV append) vs the two-launch chain, bit for bit, including
    the cache contents; positions at split boundaries and inside/outside the sliding window"""
    rng = np.random.default_rng(HS + pos)
    cap, max_seq = 2048, 2048
    hist = min(pos, 1600)
    Kc0 = torch.from_numpy(orc.to_bf16_bits(rng.uniform(-1, 1, (1, NKV, cap, HS)).astype(np.float32) * 0.5).view(np.int16)).cuda()
    Vc0 = torch.from_numpy(orc.to_bf16_bits(rng.uniform(-1, 1, (1, NKV, cap, HS)).astype(np.float32)).view(np.int16)).cuda()
    q = _bf(rng.standard_normal((NH, HS)))
    k = _bf(rng.standard_normal((NKV, HS)))
    v = k if kv_shared else _bf(rng.standard_normal((NKV, HS)))
    qw, kw = _d(_bf(1 + 0.1 * rng.uniform(-1, 1, HS))), _d(_bf(1 + 0.1 * rng.uniform(-1, 1, HS)))
    cos, sin = empty_f32(max_seq, HS // 2), empty_f32(max_seq, HS // 2)
    capi.call("rope_build_cache", cos, sin, max_seq, HS, float(base), rot)
    nbytes = capi.load().mila_cdna4_attn_decode_scratch_bytes(1, NH, HS)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    # chain
    K0, V0, q0, y0 = Kc0.clone(), Vc0.clone(), empty_u16(NH, HS), empty_u16(NH * HS)
    capi.call("fused_qkv_post", q0, K0, V0, _d(q), _d(k), _d(v), qw, kw, None, cos, sin, NH, NKV, HS, pos, cap, 1e-6)
    capi.call("attn_decode_bf16", y0, q0, K0, V0, scratch, C.c_size_t(nbytes), 1, NH, NKV, HS, cap, pos + 1, window, 1.0)
    # one launch
    K1, V1, y1 = Kc0.clone(), Vc0.clone(), empty_u16(NH * HS)
    capi.call("fused_attn_decode_bf16", y1, K1, V1, _d(q), _d(k), _d(v), qw, kw, None, cos, sin, scratch, C.c_size_t(nbytes), NH, NKV, HS,
              cap, pos, None, window, 1.0, 1e-6)
    assert np.array_equal(bits(K1), bits(K0)) and np.array_equal(bits(V1), bits(V0)), "cache rows differ"
    assert np.array_equal(bits(y1), bits(y0)), "attention output differs"
    # graph-replay form: position from device memory
    K2, V2, y2 = Kc0.clone(), Vc0.clone(), empty_u16(NH * HS)
    pd = torch.tensor([pos], dtype=torch.int32, device="cuda")
    capi.call("fused_attn_decode_bf16", y2, K2, V2, _d(q), _d(k), _d(v), qw, kw, None, cos, sin, scratch, C.c_size_t(nbytes), NH, NKV, HS,
              cap, 0, pd, window, 1.0, 1e-6)
    assert np.array_equal(bits(y2), bits(y0)) and np.array_equal(bits(K2), bits(K0))
    del hist
