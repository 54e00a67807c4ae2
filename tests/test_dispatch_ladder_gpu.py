"""One oracle sweep over the Linear dispatch ladder: RocmLinearOp::forward -- through the host mirror's Linear component (libmila_host) -- at
M in {1, 2, 16, 17, 32, 33, 64, 65, 255, 256, 300, 511, 512, 576, 1041, 2048, 2049, 2303} x the four Gemma-4 12B Linear shapes x the weight policies (bf16, PerChannelFp8<>
with its default W8A16 prefill and with the opt-in W8A8 one, PerGroupFp4<128> = W4A8), sampled rows against the oracle of the arithmetic that policy runs at that row count,
AND the kernel form that served the call (csrc/internal.h: mila_cdna4_last_form) against a committed table (tests/golden/dispatch_ladder.json): a threshold edit in
csrc/gemm.hip / gemm256.hip cannot silently route a prompt length to another form -- the table changes, and this test says so until the table is regenerated on purpose
(MILA_RECORD_LADDER=1 python -m pytest tests/test_dispatch_ladder_gpu.py  ->  gpurun_out/dispatch_ladder.json).

Reference: CudaLinearOp::forward's branches (OPS/Linear/CudaLinearOp.ixx:543-827): M == 1 decode matvecs; M > 1: bf16 cuBLASLt, fp8 weights 2-phase W8A16, fp4 weights W4A8."""
import json
import os

import numpy as np
import pytest

import orc
import synth
from gpu_util import assert_bf16_close
from mila_amd import capi, host

pytestmark = pytest.mark.gpu

ROWS = [1, 2, 16, 17, 32, 33, 64, 65, 255, 256, 300, 511, 512, 576, 1041, 2048, 2049, 2303]
SHAPES = [("qkv_proj(local)", 3840, 8192), ("o_proj(local)", 4096, 3840), ("fc_gate_up", 3840, 30720), ("fc_down", 15360, 3840)]
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dispatch_ladder.json")
RECORD = os.environ.get("MILA_RECORD_LADDER") == "1"


def _expected(policy, M, x, Wb, q):
    """the float64 composition of the arithmetic this policy runs at this row count"""
    if policy == "bf16":
        return orc.linear_bf16w(x, Wb)
    if policy in ("fp8", "fp8-w8a8"):
        w8, sc = q
        if M == 1:
            return orc.linear_fp8w(x, w8, sc)                                            # decode: scale applied once after the reduction (CudaMatVecBias.Bf16.cu:246-250)
        if policy == "fp8":
            return orc.linear_bf16w(x, orc.to_bf16_bits(orc.dequant_fp8(w8, sc)))        # W8A16: bf16(dequantized weight), bf16 GEMM (CudaLinearOp.ixx:597-644)
        x8, ts = orc.quantize_act_fp8_per_token(x)
        return orc.linear_fp8a_fp8w(x8, ts, w8, sc, 1.0)                                 # W8A8 (opt-in)
    q4, s4, ws, w8 = q
    if M == 1:
        return orc.linear_fp4w(x, q4, s4, 128)
    x8, ts = orc.quantize_act_fp8_per_token(x)                                           # W4A8 (CudaLinearOp.ixx:646-715): two-step epilogue
    raw = orc.linear_fp8a_fp8w(x8, np.ones(len(ts), dtype=np.float32), w8, None, ws)
    return orc.round_bf16(raw.astype(np.float32)).astype(np.float64) * ts.astype(np.float64)[:, None]


@pytest.mark.parametrize("policy", ["bf16", "fp8", "fp8-w8a8", "fp4"])
@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_every_row_count_reaches_a_tested_form_and_the_oracle(policy, shape):
    name, K, N = shape
    pol = "fp8" if policy == "fp8-w8a8" else policy
    Wb = synth.fill_bf16(1000 + K + N, N * K, np.float32(1.0) / np.sqrt(np.float32(K)), 0.0).reshape(N, K)
    lin = host.LinearComponent(pol, K, N, max(ROWS))
    lin.load("weight", Wb)
    if policy == "fp8-w8a8":
        lin.set(fp8_activation_prefill=True)
    if pol == "fp8":
        q = orc.quantize_fp8_per_channel(Wb)
    elif pol == "fp4":
        q4, s4 = orc.quantize_fp4_per_group(Wb, 128)
        ws = orc.fp8_weight_scale_from_groups(s4)
        q = (q4, s4, ws, orc.upcast_fp4_to_fp8(q4, s4, ws, 128))
    else:
        q = None
    golden = json.load(open(GOLDEN)) if os.path.exists(GOLDEN) else {}
    table = {}
    rng = np.random.default_rng(K + N)
    try:
        for M in ROWS:
            X = orc.round_bf16((rng.standard_normal((M, K)) * rng.uniform(0.3, 2.0, (M, 1))).astype(np.float32))
            capi.last_form()
            Y = lin.forward(orc.to_bf16_bits(X))
            forms = capi.last_form()
            assert forms, "no kernel form was noted for M = %d" % M
            table["%s/%s/%d" % (policy, name, M)] = forms
            rows = sorted({0, M - 1})
            exp = _expected(policy, M, X[rows] if M > 1 else X, Wb, q)
            assert_bf16_close(Y[rows], exp, 2, 2e-3 * float(np.abs(exp).max()), "%s %s M=%d via %s" % (policy, name, M, "+".join(forms)))
    finally:
        lin.close()
    if RECORD:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "dispatch_ladder.json")
        have = json.load(open(out)) if os.path.exists(out) else {}
        have.update(table)
        json.dump(have, open(out, "w"), indent=0, sort_keys=True)
        return
    assert golden, "tests/golden/dispatch_ladder.json is missing: record it on a GPU box (MILA_RECORD_LADDER=1)"
    diff = {k: (v, golden.get(k)) for k, v in table.items() if golden.get(k) != v}
    assert not diff, "the dispatch ladder routes these calls to other forms than the committed table says: %s" % diff
