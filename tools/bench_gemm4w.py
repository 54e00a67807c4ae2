"""The four-wave 256 x 256 GEMM (csrc/experiments/gemm4w.hip, libmila_cdna4_experiments.so) against the eight-wave ping-pong kernel of the product: bit-identity and time(K)."""
import json
import os
import sys

os.environ.setdefault("MILA_CDNA4_TUNING", "1")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

M = 2048
lib = capi.load()


def timed(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, geglu in ((8192, False), (30720, False), (30720, True)):
    for K in (1920, 3840, 7680):
        X = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16).view(torch.int16)
        W = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16).view(torch.int16)
        out = {}
        for sched in (5, 7):
            four = sched == 7
            if geglu:
                Y = torch.full((M, N // 2), 0x7fc0, dtype=torch.int16, device="cuda")
                fn = (lambda: capi.call("exp_gemm4w_geglu_bf16", Y, X, W, M, K, N // 2)) if four else (lambda: capi.call("gemm_geglu_bf16", Y, X, W, M, K, N // 2))
            else:
                Y = torch.full((M, N), 0x7fc0, dtype=torch.int16, device="cuda")
                fn = (lambda: capi.call("exp_gemm4w_bf16", Y, X, W, None, M, K, N)) if four else (lambda: capi.call("gemm_bf16", Y, X, W, None, M, K, N))
            t = timed(fn)
            out[sched] = (t, Y.clone())
        same = bool(torch.equal(out[5][1], out[7][1]))
        print(json.dumps({"N": N, "K": K, "geglu": geglu, "us_8wave": round(out[5][0], 1), "us_4wave": round(out[7][0], 1), "bit_identical": same,
                          "TFLOPs_8wave": round(2.0 * M * K * N / out[5][0] / 1e6, 1), "TFLOPs_4wave": round(2.0 * M * K * N / out[7][0] / 1e6, 1)}), flush=True)
