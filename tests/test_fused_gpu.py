"""GPU parity: the fused decode-step kernels reproduce the unfused chain of entry points bit for
bit (every intermediate bf16 rounding included) and agree with the oracle.
Chain restated from GemmaBlock::decode (Components/Transformers/Gemma/Gemma.Block.ixx:287-356)."""
import ctypes as C

import numpy as np
import pytest
import torch

import orc
from gpu_util import assert_bf16_close, bits, dev_f32, dev_i32, dev_u16, dev_u8, empty_f32, empty_u16, host
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
    capi.call("rmsnorm_bf16", xn, None, x, nw, None, 1, 1, K, 1e-6, 0.0)
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
    capi.call("rmsnorm_bf16", a_, None, x, pw, None, 1, 1, K, 1e-6, 0.0)
    capi.call("residual_bf16", r_, res, a_, C.c_int64(K))
    if post_scale != 1.0:
        capi.call("scale_bf16", r_, r_, C.c_int64(K), post_scale)
    capi.call("rmsnorm_bf16", h_, None, r_, nw, None, 1, 1, K, 1e-6, 0.0)
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
    capi.call("rmsnorm_bf16", qn, None, _d(q), _d(qw), None, NH, 1, HS, 1e-6, 0.0)
    capi.call("rmsnorm_bf16", kn, None, _d(k), _d(kw), None, NKV, 1, HS, 1e-6, 0.0)
    capi.call("rmsnorm_bf16", vn, None, _d(v), None, None, NKV, 1, HS, 1e-6, 0.0)
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


@pytest.mark.parametrize("B,NH,NKV,HS,rot,base,window,kv_shared", [(3, 16, 8, 256, 0, 1e4, 1024, False), (2, 16, 1, 512, 128, 1e6, 0, True), (5, 4, 2, 64, 0, 1e4, 8, False)])
@pytest.mark.parametrize("pos", [0, 63, 64, 1500])
def test_fused_attn_decode_takes_a_batch_like_the_reference_kernels(B, NH, NKV, HS, rot, base, window, kv_shared, pos):
    """Gqa.Decode.Bf16.cu:379-387 puts the batch in the grid; the one-launch form does too (round 3: it was B == 1): B rows of a packed [B, 1, q | k | v] projection at one
    position, each on its own caches -- bit for bit the per-row fused_qkv_post + the batched attn_decode_bf16, cache contents included"""
    rng = np.random.default_rng(HS + pos + B)
    cap, max_seq = 2048, 2048
    Kc0 = torch.from_numpy(orc.to_bf16_bits(rng.uniform(-1, 1, (B, NKV, cap, HS)).astype(np.float32) * 0.5).view(np.int16)).cuda()
    Vc0 = torch.from_numpy(orc.to_bf16_bits(rng.uniform(-1, 1, (B, NKV, cap, HS)).astype(np.float32)).view(np.int16)).cuda()
    qw_, kw_ = NH * HS, NKV * HS
    packed = qw_ + kw_ * (1 if kv_shared else 2) + 24                       # + 24: the row stride need not be the sum of the parts
    rows = _bf(rng.standard_normal((B, packed)))
    rows_d = _d(rows)
    q_off, k_off = 0, qw_
    v_off = k_off if kv_shared else k_off + kw_
    qw, kw = _d(_bf(1 + 0.1 * rng.uniform(-1, 1, HS))), _d(_bf(1 + 0.1 * rng.uniform(-1, 1, HS)))
    cos, sin = empty_f32(max_seq, HS // 2), empty_f32(max_seq, HS // 2)
    capi.call("rope_build_cache", cos, sin, max_seq, HS, float(base), rot)
    nbytes = capi.load().mila_cdna4_attn_decode_scratch_bytes(B, NH, HS)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    # chain: fused_qkv_post row by row on that row's caches, then the batched decode
    K0, V0, q0, y0 = Kc0.clone(), Vc0.clone(), empty_u16(B, NH * HS), empty_u16(B, NH * HS)
    for b in range(B):
        capi.call("fused_qkv_post", q0[b], K0[b], V0[b], rows_d[b, q_off:], rows_d[b, k_off:], rows_d[b, v_off:], qw, kw, None, cos, sin, NH, NKV, HS, pos, cap, 1e-6)
    capi.call("attn_decode_bf16", y0, q0, K0, V0, scratch, C.c_size_t(nbytes), B, NH, NKV, HS, cap, pos + 1, window, 1.0)
    # one launch
    K1, V1, y1 = Kc0.clone(), Vc0.clone(), empty_u16(B, NH * HS)
    capi.call("fused_attn_decode_batch_bf16", y1, K1, V1, rows_d[0, q_off:], rows_d[0, k_off:], rows_d[0, v_off:], C.c_int64(packed), qw, kw, None, cos, sin,
              scratch, C.c_size_t(nbytes), B, NH, NKV, HS, cap, pos, None, window, 1.0, 1e-6)
    assert np.array_equal(bits(K1), bits(K0)) and np.array_equal(bits(V1), bits(V0)), "cache rows differ"
    assert np.array_equal(bits(y1), bits(y0)), "attention output differs"
    # and against the oracle on one row: norm -> rope -> attention over that row's history
    with pytest.raises(capi.InvalidArgument):
        capi.call("fused_attn_decode_batch_bf16", y1, K1, V1, rows_d[0, q_off:], rows_d[0, k_off:], rows_d[0, v_off:], C.c_int64(8), qw, kw, None, cos, sin,
                  scratch, C.c_size_t(nbytes), B, NH, NKV, HS, cap, pos, None, window, 1.0, 1e-6)


def _chain_args(**kw):
    a = capi.decode_chain_args()
    for k, v in kw.items():
        if hasattr(v, "data_ptr"):
            v = v.data_ptr()
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("fmt,head_fmt", [(0, None), (1, None), (2, None), (0, 0), (1, 1), (2, 1)])
def test_decode_chain_equals_the_four_launch_sequence(fmt, head_fmt):
    """o_proj -> tail -> gate_up + GeGLU -> down -> tail -> next qkv_proj / lm_head in ONE launch
    (Gemma.Block.ixx:287-356) is bit-identical to matvec + fused_norm_matvec(geglu) + matvec + fused_norm_matvec,
    launch after launch on the same scratch (the arrival counter / epoch words carry over)."""
    rng = np.random.default_rng(100 + fmt * 7 + (head_fmt or 0))
    D, F, KA = 1024, 2304, 512
    NN = 1040 if head_fmt is None else 4099
    nfmt = fmt if head_fmt is None else head_fmt
    Wo, so, _ = _weights(rng, D, KA, fmt)
    Wg, sg, _ = _weights(rng, 2 * F, D, fmt)
    Wd, sd, _ = _weights(rng, D, F, fmt)
    Wn, sn, _ = _weights(rng, NN, D, nfmt)
    nws = [_d(_bf(1 + 0.1 * rng.uniform(-1, 1, D))) for _ in range(4)]    # post_attn, pre_ffn, post_ffn, next_norm
    lib, stream = capi.load(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nbytes = lib.mila_cdna4_decode_chain_scratch_bytes(D, F)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    capi.call("decode_chain_init", scratch, C.c_size_t(nbytes))
    z = 0
    for it in range(3):
        attn = _d(_bf(rng.standard_normal(KA)))
        res = _d(_bf(rng.standard_normal(D)))
        # --- four launches ---
        a0, h0, d0 = empty_u16(D), empty_u16(F), empty_u16(D)
        r1, r2 = empty_u16(D), empty_u16(D)
        _matvec(fmt, a0, attn, Wo, so, KA, D)
        fa = _args(y=h0, x=a0, W=Wg, scales=sg if sg is not None else z, norm_w=nws[1], post_w=nws[0], res=res, res_out=r1,
                   post_scale=1.0, eps=1e-6, fmt=fmt, K=D, N=F, group=128, geglu=1)
        capi.check(lib.mila_cdna4_fused_norm_matvec(C.byref(fa), stream))
        _matvec(fmt, d0, h0, Wd, sd, F, D)
        y0 = empty_f32(NN) if head_fmt is not None else empty_u16(NN)
        fb = _args(y=y0, x=d0, W=Wn, scales=sn if sn is not None else z, norm_w=nws[3], post_w=nws[2], res=r1, res_out=r2,
                   post_scale=0.75, eps=1e-6, fmt=nfmt, K=D, N=NN, group=128, geglu=0, f32_out=int(head_fmt is not None))
        capi.check(lib.mila_cdna4_fused_norm_matvec(C.byref(fb), stream))
        # --- one launch ---
        y1 = empty_f32(NN) if head_fmt is not None else empty_u16(NN)
        r2c = empty_u16(D)
        ca = _chain_args(attn=attn, res=res, res_out=r2c, y=y1, W_o=Wo, s_o=so if so is not None else z, W_gate_up=Wg,
                         s_gate_up=sg if sg is not None else z, W_down=Wd, s_down=sd if sd is not None else z, W_next=Wn,
                         s_next=sn if sn is not None else z, post_attn_w=nws[0], pre_ffn_w=nws[1], post_ffn_w=nws[2],
                         next_norm_w=nws[3], layer_scalar=0.75, eps=1e-6, fmt=fmt, group=128, next_fmt=nfmt, next_group=128,
                         f32_out=int(head_fmt is not None), D=D, F=F, K_attn=KA, N_next=NN, scratch=scratch,
                         scratch_bytes=nbytes)
        capi.check(lib.mila_cdna4_decode_chain(C.byref(ca), stream))
        err = C.c_int32(-1)
        capi.check(lib.mila_cdna4_decode_chain_status(C.c_void_p(scratch.data_ptr()), C.byref(err), stream))
        assert err.value == 0, "a hand-off wait gave up (code %d)" % err.value
        assert np.array_equal(bits(r2), bits(r2c)), "residual stream differs (launch %d)" % it
        if head_fmt is not None:
            assert np.array_equal(y0.cpu().numpy().view(np.uint32), y1.cpu().numpy().view(np.uint32)), "logits differ (launch %d)" % it
        else:
            assert np.array_equal(bits(y0), bits(y1)), "qkv differs (launch %d)" % it


def test_decode_chain_rejects_bad_arguments():
    lib, stream = capi.load(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    a = _chain_args(D=1024, F=2304, K_attn=512, N_next=8, fmt=0, next_fmt=0)
    with pytest.raises(capi.InvalidArgument):
        capi.check(lib.mila_cdna4_decode_chain(C.byref(a), stream))
    with pytest.raises(capi.InvalidArgument):
        capi.check(lib.mila_cdna4_decode_chain(None, stream))


@pytest.mark.parametrize("NH,NKV,HS,rot,base,kv_shared", [(16, 8, 256, 0, 1e4, False), (16, 1, 512, 128, 1e6, True), (4, 2, 64, 0, 1e4, False)])
def test_qkv_post_prefill_equals_split_norm_rope_kvwrite(NH, NKV, HS, rot, base, kv_shared):
    """T packed qkv rows -> q (normed, roped) + KV cache rows in one launch == split3 + q_norm + k_norm + v_norm +
    rope.prefill + kv_write (Gemma.Block.ixx:215-262), bit for bit, ring wrap-around included"""
    rng = np.random.default_rng(HS + NKV)
    T, cap, pos0, max_seq = 37, 48, 30, 128          # positions 30..66 wrap the 48-row ring
    qd, kd = NH * HS, NKV * HS
    width = qd + kd + (0 if kv_shared else kd)
    packed = _bf(rng.standard_normal((T, width)))
    qw, kw, vw = (_bf(1 + 0.1 * rng.uniform(-1, 1, HS)) for _ in range(3))
    cos, sin = empty_f32(max_seq, HS // 2), empty_f32(max_seq, HS // 2)
    capi.call("rope_build_cache", cos, sin, max_seq, HS, float(base), rot)
    P = _d(packed)
    # unfused chain
    q0, k0, v0 = empty_u16(T, qd), empty_u16(T, kd), empty_u16(T, kd)
    capi.call("split3_bf16", q0, k0, None if kv_shared else v0, P, T, qd, kd, 0 if kv_shared else kd)
    qn, kn, vn = empty_u16(T, qd), empty_u16(T, kd), empty_u16(T, kd)
    capi.call("rmsnorm_bf16", qn, None, q0, _d(qw), None, T * NH, 1, HS, 1e-6, 0.0)
    capi.call("rmsnorm_bf16", kn, None, k0, _d(kw), None, T * NKV, 1, HS, 1e-6, 0.0)
    capi.call("rmsnorm_bf16", vn, None, k0 if kv_shared else v0, _d(vw), None, T * NKV, 1, HS, 1e-6, 0.0)
    capi.call("rope_forward_bf16", qn, kn, qn, kn, cos, sin, 1, T, NH, NKV, HS, pos0, max_seq)
    K0 = torch.zeros((1, NKV, cap, HS), dtype=torch.int16, device="cuda")
    V0 = torch.zeros_like(K0)
    capi.call("kv_write_bf16", K0, V0, kn, vn, 1, T, NKV, HS, pos0, cap)
    # fused: pointers into the packed rows
    K1, V1, q1 = torch.zeros_like(K0), torch.zeros_like(K0), empty_u16(T, qd)
    base_ptr = P.data_ptr()
    kptr = C.c_void_p(base_ptr + 2 * qd)
    vptr = kptr if kv_shared else C.c_void_p(base_ptr + 2 * (qd + kd))
    capi.call("fused_qkv_post_prefill", q1, K1, V1, P, kptr, vptr, C.c_int64(width), _d(qw), _d(kw), _d(vw), cos, sin, T, NH, NKV, HS, pos0, cap, 1e-6)
    assert np.array_equal(bits(q1), bits(qn)), "q differs"
    assert np.array_equal(bits(K1), bits(K0)), "K cache differs"
    assert np.array_equal(bits(V1), bits(V0)), "V cache differs"
    with pytest.raises(capi.InvalidArgument):
        capi.call("fused_qkv_post_prefill", q1, K1, V1, P, kptr, vptr, C.c_int64(width), _d(qw), _d(kw), _d(vw), cos, sin, cap + 1, NH, NKV, HS, pos0, cap, 1e-6)


@pytest.mark.parametrize("D", [3840, 2056])
@pytest.mark.parametrize("post_scale,with_next", [(1.0, True), (0.75, True), (0.75, False)])
def test_tail_norm_equals_rmsnorm_residual_scale_rmsnorm(D, post_scale, with_next):
    """the prefill sandwich tail over T rows in one launch == RmsNorm + Residual (+ scale) + RmsNorm (Gemma.Block.ixx:339-356)"""
    rng = np.random.default_rng(D)
    T = 19
    a = _d(_bf(rng.standard_normal((T, D)) * 3))
    res = _d(_bf(rng.standard_normal((T, D))))
    pw, nw = _d(_bf(1 + 0.1 * rng.uniform(-1, 1, D))), _d(_bf(1 + 0.1 * rng.uniform(-1, 1, D)))
    an, r0, x0 = empty_u16(T, D), empty_u16(T, D), empty_u16(T, D)
    capi.call("rmsnorm_bf16", an, None, a, pw, None, T, 1, D, 1e-6, 0.0)
    capi.call("residual_bf16", r0, res, an, C.c_int64(T * D))
    if post_scale != 1.0:
        capi.call("scale_bf16", r0, r0, C.c_int64(T * D), post_scale)
    capi.call("rmsnorm_bf16", x0, None, r0, nw, None, T, 1, D, 1e-6, 0.0)
    r1, x1 = empty_u16(T, D), empty_u16(T, D)
    capi.call("fused_tail_norm_bf16", r1, x1 if with_next else None, a, res, pw, nw if with_next else None, T, D, post_scale, 1e-6)
    assert np.array_equal(bits(r0), bits(r1)), "residual stream differs"
    if with_next:
        assert np.array_equal(bits(x0), bits(x1)), "normed output differs"
    with pytest.raises(capi.InvalidArgument):
        capi.call("fused_tail_norm_bf16", r1, None, a, res, pw, None, T, 512, post_scale, 1e-6)


@pytest.mark.parametrize("D", [3840, 2056, 8192])
def test_tail_norm_with_per_token_fp8_output_equals_tail_norm_then_quantize(D):
    """fused_tail_norm_quant_bf16: the normalised rows also as the W4A8 Linear's operand -- bit for bit what quantize_fp8_per_token makes of XN
    (Fp8Prefill/CudaFp8Prefill.cu:108-160 folded into the producer); an all-zero row takes the 1e-12 floor"""
    rng = np.random.default_rng(D + 1)
    T = 21
    A = _bf(rng.standard_normal((T, D)) * 3)
    R = _bf(rng.standard_normal((T, D)))
    A[5] = 0.0
    R[5] = 0.0                                                   # a row whose tail output is exactly zero
    a, res = _d(A), _d(R)
    pw, nw = _d(_bf(1 + 0.1 * rng.uniform(-1, 1, D))), _d(_bf(1 + 0.1 * rng.uniform(-1, 1, D)))
    r0, x0 = empty_u16(T, D), empty_u16(T, D)
    capi.call("fused_tail_norm_bf16", r0, x0, a, res, pw, nw, T, D, 0.75, 1e-6)
    q0 = torch.empty((T, D), dtype=torch.uint8, device="cuda")
    s0 = torch.empty((T,), dtype=torch.float32, device="cuda")
    capi.call("quantize_fp8_per_token", q0, s0, x0, T, D)
    r1, x1 = empty_u16(T, D), empty_u16(T, D)
    q1 = torch.full((T, D), 0x55, dtype=torch.uint8, device="cuda")
    s1 = torch.full((T,), -1.0, dtype=torch.float32, device="cuda")
    capi.call("fused_tail_norm_quant_bf16", r1, x1, q1, s1, a, res, pw, nw, T, D, 0.75, 1e-6)
    assert np.array_equal(bits(r0), bits(r1)) and np.array_equal(bits(x0), bits(x1))
    assert torch.equal(s0, s1), "per-token scales differ"
    assert torch.equal(q0, q1), "e4m3 rows differ"
    assert float(s1[5]) == np.float32(1e-12) / np.float32(448.0)
    with pytest.raises(capi.InvalidArgument):
        capi.call("fused_tail_norm_quant_bf16", r1, x1, q1, None, a, res, pw, nw, T, D, 0.75, 1e-6)


@pytest.mark.parametrize("NH,NKV,HS,rot,base,window,kv_shared,cap", [(16, 8, 256, 0, 1e4, 1024, False, 2048), (16, 1, 512, 128, 1e6, 0, True, 512),
                                                                     (4, 2, 256, 0, 1e4, 128, False, 256)])
@pytest.mark.parametrize("fmt", [0, 1, 2])
@pytest.mark.parametrize("pos", [3, 200, 1500])
def test_attention_partials_plus_combining_o_proj_equals_attention_plus_o_proj(NH, NKV, HS, rot, base, window, kv_shared, cap, fmt, pos):
    """flash-decode without its combine launch + an o_proj whose prologue combines the split partials == the two-launch
    attention + the plain o_proj Linear, bit for bit (cache rows too)"""
    pos = min(pos, cap - 1)
    rng = np.random.default_rng(HS + NKV + pos + fmt)
    max_seq, D = 2048, 520
    lib = capi.load()
    splits = lib.mila_cdna4_attn_decode_split_count(1, NH, NKV, HS, cap, window)
    assert splits > 1
    Kc0 = torch.from_numpy(orc.to_bf16_bits(rng.uniform(-1, 1, (1, NKV, cap, HS)).astype(np.float32) * 0.5).view(np.int16)).cuda()
    Vc0 = torch.from_numpy(orc.to_bf16_bits(rng.uniform(-1, 1, (1, NKV, cap, HS)).astype(np.float32)).view(np.int16)).cuda()
    q = _bf(rng.standard_normal((NH, HS)))
    k = _bf(rng.standard_normal((NKV, HS)))
    v = k if kv_shared else _bf(rng.standard_normal((NKV, HS)))
    qw, kw = _d(_bf(1 + 0.1 * rng.uniform(-1, 1, HS))), _d(_bf(1 + 0.1 * rng.uniform(-1, 1, HS)))
    cos, sin = empty_f32(max_seq, HS // 2), empty_f32(max_seq, HS // 2)
    capi.call("rope_build_cache", cos, sin, max_seq, HS, float(base), rot)
    nbytes = lib.mila_cdna4_attn_decode_scratch_bytes(1, NH, HS)
    scratch = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    W, s, _ = _weights(rng, D, NH * HS, fmt)
    # two launches + plain Linear
    K0, V0, a0, y0 = Kc0.clone(), Vc0.clone(), empty_u16(NH * HS), empty_u16(D)
    capi.call("fused_attn_decode_bf16", a0, K0, V0, _d(q), _d(k), _d(v), qw, kw, None, cos, sin, scratch, C.c_size_t(nbytes), NH, NKV, HS,
              cap, pos, None, window, 1.0, 1e-6)
    _matvec(fmt, y0, a0, W, s, NH * HS, D)
    # partials + combining Linear
    K1, V1, y1 = Kc0.clone(), Vc0.clone(), empty_u16(D)
    scratch.zero_()
    capi.call("fused_attn_decode_partials_bf16", K1, V1, _d(q), _d(k), _d(v), qw, kw, None, cos, sin, scratch, C.c_size_t(nbytes), NH, NKV, HS,
              cap, pos, None, window, 1.0, 1e-6)
    capi.call("matvec_attn_combine", y1, scratch, splits, NH, HS, W, s, fmt, D, 128)
    assert np.array_equal(bits(K1), bits(K0)) and np.array_equal(bits(V1), bits(V0)), "cache rows differ"
    assert np.array_equal(bits(y1), bits(y0)), "o_proj output differs"


@pytest.mark.parametrize("NH,NKV,HS,rot,base,window,kv_shared,cap", [(16, 8, 256, 0, 1e4, 1024, False, 2048), (16, 1, 512, 128, 1e6, 0, True, 2048),
                                                                     (4, 2, 64, 0, 1e4, 128, False, 256), (8, 8, 128, 0, 1e4, 0, False, 512)])
@pytest.mark.parametrize("pos", [3, 200, 1500])
def test_onepass_attention_equals_attention_plus_combine(NH, NKV, HS, rot, base, window, kv_shared, cap, pos):
    """split flash-decode whose last-arriving workgroup merges its head-group's partials in the same launch == attention + combine
    launches, bit for bit, call after call on the SAME scratch with changing queries (a stale partial from the previous call, or
    a ticket left armed, would show), while another stream keeps the memory system busy"""
    pos = min(pos, cap - 1)
    rng = np.random.default_rng(7 * HS + NKV + pos)
    max_seq = 2048
    lib = capi.load()
    splits = lib.mila_cdna4_attn_decode_split_count(1, NH, NKV, HS, cap, window)
    assert splits > 1
    Kc0 = torch.from_numpy(orc.to_bf16_bits(rng.uniform(-1, 1, (1, NKV, cap, HS)).astype(np.float32) * 0.5).view(np.int16)).cuda()
    Vc0 = torch.from_numpy(orc.to_bf16_bits(rng.uniform(-1, 1, (1, NKV, cap, HS)).astype(np.float32)).view(np.int16)).cuda()
    qw, kw = _d(_bf(1 + 0.1 * rng.uniform(-1, 1, HS))), _d(_bf(1 + 0.1 * rng.uniform(-1, 1, HS)))
    cos, sin = empty_f32(max_seq, HS // 2), empty_f32(max_seq, HS // 2)
    capi.call("rope_build_cache", cos, sin, max_seq, HS, float(base), rot)
    nbytes = lib.mila_cdna4_attn_decode_scratch_bytes(1, NH, HS)
    scratch0 = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    scratch1 = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    nt = lib.mila_cdna4_attn_decode_ticket_count(1, NH)
    assert nt >= NH
    tickets = torch.zeros(nt, dtype=torch.int32, device="cuda")
    reps = 24
    qs = [_d(_bf(rng.standard_normal((NH, HS)))) for _ in range(reps)]
    ks = [_d(_bf(rng.standard_normal((NKV, HS)))) for _ in range(reps)]
    vs = ks if kv_shared else [_d(_bf(rng.standard_normal((NKV, HS)))) for _ in range(reps)]
    y0 = [empty_u16(NH * HS) for _ in range(reps)]
    y1 = [empty_u16(NH * HS) for _ in range(reps)]
    K0, V0, K1, V1 = Kc0.clone(), Vc0.clone(), Kc0.clone(), Vc0.clone()
    for i in range(reps):
        capi.call("fused_attn_decode_bf16", y0[i], K0, V0, qs[i], ks[i], vs[i], qw, kw, None, cos, sin, scratch0, C.c_size_t(nbytes), NH, NKV, HS,
                  cap, pos, None, window, 1.0, 1e-6)
    torch.cuda.synchronize()
    # background load on a second stream while the one-pass launches run back to back
    side = torch.cuda.Stream()
    big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    big2 = torch.empty_like(big)
    with torch.cuda.stream(side):
        for _ in range(6):
            big2.copy_(big)
    for i in range(reps):
        capi.call("fused_attn_decode_onepass_bf16", y1[i], K1, V1, qs[i], ks[i], vs[i], qw, kw, None, cos, sin, scratch1, C.c_size_t(nbytes),
                  tickets, C.c_size_t(nt), NH, NKV, HS, cap, pos, None, window, 1.0, 1e-6)
    torch.cuda.synchronize()
    assert np.array_equal(bits(K1), bits(K0)) and np.array_equal(bits(V1), bits(V0)), "cache rows differ"
    for i in range(reps):
        assert np.array_equal(bits(y1[i]), bits(y0[i])), "one-pass output differs at call %d" % i
    assert int(tickets.abs().sum().item()) == 0, "a ticket was left armed"
    # graph-replay form
    pd = torch.tensor([pos], dtype=torch.int32, device="cuda")
    y2 = empty_u16(NH * HS)
    capi.call("fused_attn_decode_onepass_bf16", y2, K1, V1, qs[-1], ks[-1], vs[-1], qw, kw, None, cos, sin, scratch1, C.c_size_t(nbytes),
              tickets, C.c_size_t(nt), NH, NKV, HS, cap, 0, pd, window, 1.0, 1e-6)
    assert np.array_equal(bits(y2), bits(y0[-1]))
    with pytest.raises(capi.InvalidArgument):
        capi.call("fused_attn_decode_onepass_bf16", y2, K1, V1, qs[0], ks[0], vs[0], qw, kw, None, cos, sin, scratch1, C.c_size_t(nbytes),
                  tickets, C.c_size_t(1), NH, NKV, HS, cap, pos, None, window, 1.0, 1e-6)


def test_prefetch_l3_reads_only():
    """the Infinity-Cache prefetch touches every 128-byte line of a range and writes nothing"""
    n = 3 * (1 << 20) + 128 * 5
    src = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
    keep = src.clone()
    sink = torch.zeros(4, dtype=torch.float32, device="cuda")
    for wgs in (1, 64, 300):
        capi.call("prefetch_l3", src, C.c_size_t(n), wgs, sink)
    capi.call("prefetch_l3", src, C.c_size_t(0), 64, sink)
    torch.cuda.synchronize()
    assert torch.equal(src, keep) and float(sink.abs().sum()) == 0.0
    with pytest.raises(capi.InvalidArgument):
        capi.call("prefetch_l3", src, C.c_size_t(n), 0, sink)


@pytest.mark.parametrize("fmt,N,with_res", [(0, 262144, True), (1, 262144, True), (0, 5000, False), (1, 300, False)])
def test_lm_head_epilogue_runs_the_samplers_first_stage(fmt, N, with_res):
    """round 3: the lm_head launch (tail + final norm + matvec, FP32 logits) also leaves one (largest logit, index) partial per workgroup in the sampler's scratch, and
    sample_argmax_final_advance picks the token from them -- the token sample_argmax_fp32 picks from the logits of the same launch (ties to the LOWEST index,
    Sampling.cu:23-75: duplicated weight rows make exact ties across waves and workgroups), with the position bump and the publication of sample_argmax_advance_fp32"""
    lib, stream = capi.load(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(N + fmt)
    K = 3840
    Wb = orc.to_bf16_bits((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    dup = [N - 1, N // 2 + 1, 7] if N > 1000 else [N - 1, 3]
    best = Wb[5].copy()
    if fmt == 0:
        Wd, sd = dev_u16(Wb), None
    else:
        q, sc = orc.quantize_fp8_per_channel(Wb)
        Wd, sd = dev_u8(q), dev_f32(sc)
    x = _bf(rng.standard_normal(K))
    nw = _d(_bf(1 + 0.1 * rng.uniform(-1, 1, K)))
    logits = empty_f32(N)
    nb = lib.mila_cdna4_sample_scratch_bytes()
    scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
    blocks = C.c_int(0)
    kw = dict(y=logits, x=_d(x), W=Wd, scales=sd if sd is not None else 0, norm_w=nw, post_w=0, res=0, res_out=0, post_scale=1.0, eps=1e-6, fmt=fmt, K=K, N=N, group=128,
              geglu=0, f32_out=1)
    if with_res:
        res, r_out, pw = _d(_bf(rng.standard_normal(K))), empty_u16(K), _d(_bf(1 + 0.1 * rng.uniform(-1, 1, K)))
        kw.update(post_w=pw, res=res, res_out=r_out, post_scale=0.75)
    a = _args(**kw)
    a.argmax_scratch = scratch.data_ptr(); a.argmax_scratch_bytes = nb; a.argmax_blocks = C.pointer(blocks)
    capi.check(lib.mila_cdna4_fused_norm_matvec(C.byref(a), stream))
    lg = host(logits)
    # make exact ties at the maximum: copy the winning row over a few others and run again
    win = int(np.argmax(lg))
    rows = sorted(set(dup + [win]))
    Wd[rows] = Wd[win].clone()
    if fmt != 0:
        sd[rows] = sd[win].clone()
    del best
    capi.check(lib.mila_cdna4_fused_norm_matvec(C.byref(a), stream))
    lg = host(logits)
    assert 0 < blocks.value <= nb // 8
    assert np.all(lg[rows] == lg[win]) and int(np.argmax(lg)) == rows[0], "the tie was not constructed"
    tok, tok_ref, pos = dev_i32(np.array([-1])), dev_i32(np.array([-1])), dev_i32(np.array([9]))
    seq = torch.tensor([2], dtype=torch.int64, device="cuda")
    ring = torch.zeros(4, dtype=torch.int64, device="cuda")
    capi.call("sample_argmax_final_advance", tok, scratch, C.c_size_t(nb), blocks.value, pos, seq, ring, 4)
    scratch2 = torch.empty(nb, dtype=torch.uint8, device="cuda")
    capi.call("sample_argmax_fp32", logits, tok_ref, N, scratch2, C.c_size_t(nb))
    assert int(host(tok)[0]) == int(host(tok_ref)[0]) == rows[0]
    assert int(host(pos)[0]) == 10 and int(seq.item()) == 3 and int(ring[3].item()) == (3 << 32) | rows[0]
    # the logits of a launch with the first stage are the logits of a launch without it
    b = _args(**kw)
    logits2 = empty_f32(N)
    b.y = logits2.data_ptr()
    capi.check(lib.mila_cdna4_fused_norm_matvec(C.byref(b), stream))
    assert np.array_equal(host(logits2).view(np.uint32), lg.view(np.uint32))
    with pytest.raises(capi.InvalidArgument):
        capi.call("sample_argmax_final_advance", tok, scratch, C.c_size_t(nb), 0, pos, seq, ring, 4)
    # a workgroup rule beyond the sampler's 512 partial slots (here a tuned 2048) is capped BEFORE the launch -- the scratch's guard bytes stay, the token is the same
    guarded = torch.full((nb + 4096,), 0x5A, dtype=torch.uint8, device="cuda")
    a.argmax_scratch = guarded.data_ptr()
    try:
        capi.tune("matvec.max_workgroups", 2048)
        capi.check(lib.mila_cdna4_fused_norm_matvec(C.byref(a), stream))
    finally:
        capi.tune_reset()
    assert 0 < blocks.value <= nb // 8 and bool((guarded[nb:] == 0x5A).all()), "the argmax epilogue wrote past its partial slots"
    capi.call("sample_argmax_final_advance", tok, guarded[:nb], C.c_size_t(nb), blocks.value, pos, seq, ring, 4)
    assert int(host(tok)[0]) == rows[0]


@pytest.mark.parametrize("B,pos", [(1, 0), (1, 37), (1, 5000), (1, 8191), (2, 4500)])
def test_long_band_global_layer_takes_the_matrix_core_decode_and_the_fused_form_is_its_chain(B, pos):
    """round 3: 16 query heads on one KV head over a band of >= 4096 keys (Gemma's global layers in a long context) decode on the matrix cores (attn_decode_mfma_kernel):
    the choice depends on (window, capacity) only, so attn_decode_bf16, the fused entry, its device-position form and the batch form all take it -- the fused forms ARE
    the chain fused_qkv_post + attn_decode_bf16 there, bit for bit -- and the result is within 1 bf16 ulp of the double-precision oracle like the flash prefill's.
    With the hook off (the wave-per-position kernel) the same call stays within the same bar: two kernels, one function."""
    lib = capi.load()
    NH, NKV, HS, cap, window, base, rot = 16, 1, 512, 8192, 0, 1e6, 128
    # round 4: the launch geometry follows the live-length BUCKET (4096, 8192, ... keys; csrc/attention.hip: band_bucket), and the matrix-core decode starts at the 8192
    # bucket: positions >= 4096 of this cache take it, earlier ones the wave-per-position kernel -- asserted through last_form below
    want_form = "attn_decode_mfma" if pos + 1 > 4096 else "attn_decode"
    rng = np.random.default_rng(pos + B)
    hist_k = _bf(rng.uniform(-1, 1, (B, NKV, cap, HS)) * 0.5)
    hist_v = _bf(rng.uniform(-1, 1, (B, NKV, cap, HS)))
    Kc0, Vc0 = _d(hist_k), _d(hist_v)
    packed = NH * HS + NKV * HS
    rows = _bf(rng.standard_normal((B, packed)))
    rows_d = _d(rows)
    qw, kw = _d(_bf(1 + 0.1 * rng.uniform(-1, 1, HS))), _d(_bf(1 + 0.1 * rng.uniform(-1, 1, HS)))
    cos, sin = empty_f32(cap, HS // 2), empty_f32(cap, HS // 2)
    capi.call("rope_build_cache", cos, sin, cap, HS, float(base), rot)
    nbytes = lib.mila_cdna4_attn_decode_scratch_bytes(B, NH, HS)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    # chain
    K0, V0, q0, y0 = Kc0.clone(), Vc0.clone(), empty_u16(B, NH * HS), empty_u16(B, NH * HS)
    for b in range(B):
        capi.call("fused_qkv_post", q0[b], K0[b], V0[b], rows_d[b], rows_d[b, NH * HS:], rows_d[b, NH * HS:], qw, kw, None, cos, sin, NH, NKV, HS, pos, cap, 1e-6)
    capi.last_form()
    capi.call("attn_decode_bf16", y0, q0, K0, V0, scratch, C.c_size_t(nbytes), B, NH, NKV, HS, cap, pos + 1, window, 1.0)
    assert capi.last_form() == [want_form]
    # fused forms
    K1, V1, y1 = Kc0.clone(), Vc0.clone(), empty_u16(B, NH * HS)
    if B == 1:
        capi.call("fused_attn_decode_bf16", y1, K1, V1, rows_d[0], rows_d[0, NH * HS:], rows_d[0, NH * HS:], qw, kw, None, cos, sin, scratch, C.c_size_t(nbytes), NH, NKV, HS,
                  cap, pos, None, window, 1.0, 1e-6)
    else:
        capi.call("fused_attn_decode_batch_bf16", y1, K1, V1, rows_d[0], rows_d[0, NH * HS:], rows_d[0, NH * HS:], C.c_int64(packed), qw, kw, None, cos, sin, scratch,
                  C.c_size_t(nbytes), B, NH, NKV, HS, cap, pos, None, window, 1.0, 1e-6)
    assert np.array_equal(bits(K1), bits(K0)) and np.array_equal(bits(V1), bits(V0)) and np.array_equal(bits(y1), bits(y0))
    if B == 1:
        K2, V2, y2 = Kc0.clone(), Vc0.clone(), empty_u16(B, NH * HS)
        pd = torch.tensor([pos], dtype=torch.int32, device="cuda")
        # the device-position form: `position` carries the live-length bound the launch was captured for (any value of the same bucket gives the eager form's bits)
        capi.call("fused_attn_decode_bf16", y2, K2, V2, rows_d[0], rows_d[0, NH * HS:], rows_d[0, NH * HS:], qw, kw, None, cos, sin, scratch, C.c_size_t(nbytes), NH, NKV, HS,
                  cap, 4096 if pos + 1 <= 4096 else 8192, pd, window, 1.0, 1e-6)
        assert np.array_equal(bits(y2), bits(y0)) and np.array_equal(bits(K2), bits(K0))
    # oracle: the roped q rows against the linear history 0 .. pos (the appended row included)
    Kh = orc.from_bf16_bits(bits(K0)).reshape(B, NKV, cap, HS)[:, :, :pos + 1].transpose(0, 2, 1, 3)
    Vh = orc.from_bf16_bits(bits(V0)).reshape(B, NKV, cap, HS)[:, :, :pos + 1].transpose(0, 2, 1, 3)
    qn = orc.from_bf16_bits(bits(q0)).reshape(B, 1, NH, HS)
    exp = orc.gqa_attention(qn, np.ascontiguousarray(Kh), np.ascontiguousarray(Vh), pos, window, 1.0)[:, 0]
    assert_bf16_close(bits(y0), exp, 1, 2e-3, "%s vs oracle" % want_form)
    # the OTHER kernel on the same inputs (the matrix-core decode forced from the first bucket on / switched off): two kernels, one function
    y3 = empty_u16(B, NH * HS)
    try:
        if want_form == "attn_decode_mfma":
            capi.tune("attn.mfma_decode", 0)
        else:
            capi.tune("attn.mfma_min_band", 4096)
        capi.last_form()
        capi.call("attn_decode_bf16", y3, q0, K0, V0, scratch, C.c_size_t(nbytes), B, NH, NKV, HS, cap, pos + 1, window, 1.0)
        assert capi.last_form() == ["attn_decode" if want_form == "attn_decode_mfma" else "attn_decode_mfma"]
    finally:
        capi.tune_reset()
    assert_bf16_close(bits(y3), exp, 1, 2e-3, "the other decode kernel vs oracle")
