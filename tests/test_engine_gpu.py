"""The decode ENGINE (csrc/engine.hip): o_proj -> tail -> fc_gate_up + GeGLU -> fc_down -> tail -> next qkv_proj / lm_head as ONE
persistent launch (LDS-DMA loader ring + data-tagged granule hand-offs) is BIT-IDENTICAL to the sequence of four launches
matvec + fused_norm_matvec(geglu) + matvec + fused_norm_matvec (Gemma.Block.ixx:287-356), launch after launch on one scratch, for
ragged shapes (columns not a multiple of 256, rows shorter than one 1-KiB piece, fp4 scale rows not 16-byte aligned) and for the
Gemma-4 12B geometry itself; no hand-off wait gives up."""
import ctypes as C

import numpy as np
import pytest
import torch

from gpu_util import bits, dev_f32, dev_u16, dev_u8, empty_f32, empty_u16
from mila_amd import capi
from test_fused_gpu import _args, _bf, _chain_args, _d, _matvec, _weights

pytestmark = pytest.mark.gpu


def _device_weights(N, K, fmt, seed):
    """full-size matrices: generated and quantized on the device (the CPU oracle would take minutes)"""
    W = empty_u16(N, K)
    capi.call("fill_uniform_bf16", W, C.c_int64(N * K), C.c_uint64(seed), float(K) ** -0.5, 0.0)
    if fmt == 0:
        return W, None
    if fmt == 1:
        q, s = torch.empty((N, K), dtype=torch.uint8, device="cuda"), empty_f32(N)
        capi.call("quantize_fp8_per_channel", q, s, W, N, K)
        return q, s
    q, s = torch.empty((N, K // 2), dtype=torch.uint8, device="cuda"), empty_f32(N, K // 128)
    capi.call("quantize_fp4_per_group", q, s, W, N, K, 128)
    return q, s


def _run_pair(lib, stream, fmt, nfmt, head, D, F, KA, NN, Wo, so, Wg, sg, Wd, sd, Wn, sn, nws, attn, res, scratch, nbytes):
    z = 0
    a0, h0, d0 = empty_u16(D), empty_u16(F), empty_u16(D)
    r1, r2 = empty_u16(D), empty_u16(D)
    _matvec(fmt, a0, attn, Wo, so, KA, D)
    fa = _args(y=h0, x=a0, W=Wg, scales=sg if sg is not None else z, norm_w=nws[1], post_w=nws[0], res=res, res_out=r1,
               post_scale=1.0, eps=1e-6, fmt=fmt, K=D, N=F, group=128, geglu=1)
    capi.check(lib.mila_cdna4_fused_norm_matvec(C.byref(fa), stream))
    _matvec(fmt, d0, h0, Wd, sd, F, D)
    y0 = empty_f32(NN) if head else empty_u16(NN)
    fb = _args(y=y0, x=d0, W=Wn, scales=sn if sn is not None else z, norm_w=nws[3], post_w=nws[2], res=r1, res_out=r2,
               post_scale=0.75, eps=1e-6, fmt=nfmt, K=D, N=NN, group=128, geglu=0, f32_out=int(head))
    capi.check(lib.mila_cdna4_fused_norm_matvec(C.byref(fb), stream))
    y1 = torch.full((NN,), float("nan"), dtype=torch.float32, device="cuda") if head else torch.full((NN,), 0x7fc0, dtype=torch.int16, device="cuda")
    r2c = torch.full((D,), 0x7fc0, dtype=torch.int16, device="cuda")
    ca = _chain_args(attn=attn, res=res, res_out=r2c, y=y1, W_o=Wo, s_o=so if so is not None else z, W_gate_up=Wg,
                     s_gate_up=sg if sg is not None else z, W_down=Wd, s_down=sd if sd is not None else z, W_next=Wn,
                     s_next=sn if sn is not None else z, post_attn_w=nws[0], pre_ffn_w=nws[1], post_ffn_w=nws[2],
                     next_norm_w=nws[3], layer_scalar=0.75, eps=1e-6, fmt=fmt, group=128, next_fmt=nfmt, next_group=128,
                     f32_out=int(head), D=D, F=F, K_attn=KA, N_next=NN, scratch=scratch, scratch_bytes=nbytes)
    capi.check(lib.mila_cdna4_decode_engine(C.byref(ca), stream))
    err = C.c_int32(-1)
    capi.check(lib.mila_cdna4_decode_engine_status(C.c_void_p(scratch.data_ptr()), C.byref(err), stream))
    assert err.value == 0, "a wait inside the engine gave up (code %d)" % err.value
    assert np.array_equal(bits(r2), bits(r2c)), "residual stream differs"
    if head:
        assert np.array_equal(y0.cpu().numpy().view(np.uint32), y1.cpu().numpy().view(np.uint32)), "logits differ"
    else:
        assert np.array_equal(bits(y0), bits(y1)), "qkv differs"


@pytest.mark.parametrize("fmt,head_fmt", [(0, None), (1, None), (2, None), (0, 0), (1, 1), (2, 1)])
def test_engine_equals_the_four_launch_sequence_on_ragged_shapes(fmt, head_fmt):
    rng = np.random.default_rng(300 + fmt * 7 + (head_fmt or 0))
    D, F, KA = 1024, 2304, 512
    NN = 1040 if head_fmt is None else 4099
    nfmt = fmt if head_fmt is None else head_fmt
    lib, stream = capi.load(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if not lib.mila_cdna4_decode_engine_applicable(fmt, 128, D, F, KA, NN, nfmt):
        pytest.skip("the engine does not serve this geometry on this device")
    Wo, so, _ = _weights(rng, D, KA, fmt)
    Wg, sg, _ = _weights(rng, 2 * F, D, fmt)
    Wd, sd, _ = _weights(rng, D, F, fmt)
    Wn, sn, _ = _weights(rng, NN, D, nfmt)
    nws = [_d(_bf(1 + 0.1 * rng.uniform(-1, 1, D))) for _ in range(4)]
    nbytes = lib.mila_cdna4_decode_engine_scratch_bytes(D, F)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    capi.call("decode_engine_init", scratch, C.c_size_t(nbytes))
    for it in range(4):       # launch after launch on the same scratch: the tags must not alias
        attn = _d(_bf(rng.standard_normal(KA)))
        res = _d(_bf(rng.standard_normal(D)))
        _run_pair(lib, stream, fmt, nfmt, head_fmt is not None, D, F, KA, NN, Wo, so, Wg, sg, Wd, sd, Wn, sn, nws, attn, res, scratch, nbytes)


@pytest.mark.parametrize("fmt", [0, 1, 2])
@pytest.mark.parametrize("layer", ["local", "global", "head"])
def test_engine_equals_the_four_launch_sequence_on_the_gemma_12b_geometry(fmt, layer):
    """D 3840, F 15360; attention width 4096 (local) / 8192 (global); next = qkv_proj of a local (8192) or global (8704) layer, or
    the tied lm_head (262144 rows, bf16 / fp8 table, fp32 logits)"""
    rng = np.random.default_rng(17 + fmt)
    D, F = 3840, 15360
    KA = 8192 if layer == "global" else 4096
    head = layer == "head"
    NN = 262144 if head else (8704 if layer == "local" else 8192)      # after a local layer's tail comes (5 times in 6) a local qkv; test both widths
    nfmt = (0 if fmt == 0 else 1) if head else fmt
    lib, stream = capi.load(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.mila_cdna4_decode_engine_applicable(fmt, 128, D, F, KA, NN, nfmt), "the engine must serve the benchmark's geometry"
    Wo, so = _device_weights(D, KA, fmt, 1)
    Wg, sg = _device_weights(2 * F, D, fmt, 2)
    Wd, sd = _device_weights(D, F, fmt, 3)
    Wn, sn = _device_weights(NN, D, nfmt, 4)
    nws = [_d(_bf(1 + 0.1 * rng.uniform(-1, 1, D))) for _ in range(4)]
    nbytes = lib.mila_cdna4_decode_engine_scratch_bytes(D, F)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    capi.call("decode_engine_init", scratch, C.c_size_t(nbytes))
    for it in range(2):
        attn = _d(_bf(rng.standard_normal(KA)))
        res = _d(_bf(rng.standard_normal(D)))
        _run_pair(lib, stream, fmt, nfmt, head, D, F, KA, NN, Wo, so, Wg, sg, Wd, sd, Wn, sn, nws, attn, res, scratch, nbytes)


def test_engine_rejects_what_it_cannot_serve():
    lib, stream = capi.load(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert not lib.mila_cdna4_decode_engine_applicable(0, 128, 16384, 2304, 512, 8, 0)       # D beyond the prologue's share
    assert not lib.mila_cdna4_decode_engine_applicable(2, 128, 1024, 2304, 500, 8, 2)         # K_attn not a multiple of the group
    a = _chain_args(D=1024, F=2304, K_attn=512, N_next=8, fmt=0, next_fmt=0)
    with pytest.raises(capi.InvalidArgument):
        capi.check(lib.mila_cdna4_decode_engine(C.byref(a), stream))
    with pytest.raises(capi.InvalidArgument):
        capi.check(lib.mila_cdna4_decode_engine(None, stream))
