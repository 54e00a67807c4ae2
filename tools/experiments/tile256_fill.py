"""Where between the 256 x 256 grid and the 256 x 128 ring does a shape belong whose 256 x 256 tiles fill 70-80 % of their rounds?  GPT-2's fc_1 (384 tiles = 1.5 rounds) and
neighbours, gemm.tile256_min_fill 80 (default) against 70 / 60 / 50.  Minimum of five interleaved passes."""
import json
import os
import sys

import torch

os.environ.setdefault("MILA_CDNA4_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mila_amd import capi  # noqa: E402

capi.load()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, M, K, N, act in (("gpt2 fc_1 + gelu (384 tiles of 256^2)", 8192, 768, 3072, 1), ("gpt2 qkv (288)", 8192, 768, 2304, 0), ("N = 4608 (576)", 8192, 768, 4608, 0), ("M 2048 N 11264 (352)", 2048, 3840, 11264, 0),
                           ("M 2048 N 9216 (288)", 2048, 3840, 9216, 0), ("M 2048 N 12288 (384)", 2048, 3840, 12288, 0)):
    X = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16).view(torch.int16)
    W = (torch.randn((N, K), device="cuda") / K ** 0.5).to(torch.bfloat16).view(torch.int16)
    b = (torch.randn((N,), device="cuda") * 0.1).to(torch.bfloat16).view(torch.int16)
    fills = (80, 70, 50)
    Y = {f: torch.empty((M, N), dtype=torch.int16, device="cuda") for f in fills}
    best = {f: 1e9 for f in fills}
    forms = {}
    fn = "gemm_gelu_bf16" if act else "gemm_bf16"
    for _ in range(5):
        for f in fills:
            capi.tune_reset()
            capi.tune("gemm.tile256_min_fill", f)
            capi.tune("gemm.colsplit", 0)
            best[f] = min(best[f], timed(lambda: capi.call(fn, Y[f], X, W, b, M, K, N)))
            forms[f] = capi.last_form()[-1]
    capi.tune_reset()
    print(json.dumps({"case": name, "us_by_min_fill": {str(f): round(v, 1) for f, v in best.items()}, "forms": forms, "same_bits": all(bool(torch.equal(Y[80], Y[f])) for f in fills)}), flush=True)
