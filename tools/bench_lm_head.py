"""GPT-2's lm_head (M = 8192, K = 768, N = 50257 -> 823 MB of logits): where does the time go?  N aligned / 8-aligned / odd, K x 1 / 2 / 4 (slope = K loop, intercept = per-tile cost).
    python tools/bench_lm_head.py > gpurun_out/r03_lm_head.txt"""
import json
import os
import sys

import torch

os.environ.setdefault("MILA_CDNA4_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

lib = capi.load()
M = 8192


def timed(fn, n=10):
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


for N in (50176, 50264, 50257):
    row = {"N": N}
    for K in (768, 3072):
        X = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16).view(torch.int16)
        W = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16).view(torch.int16)
        Y = torch.empty((M, N), dtype=torch.int16, device="cuda")
        for tag, sched, persistent, rowwise in (("default", 5, 1, 1), ("direct_stores_persistent", 5, 1, 0), ("direct_stores_one_wg_per_tile", 5, 0, 0), ("ring_256x128", 2, 1, 1)):
            capi.tune("gemm.schedule", sched)
            capi.tune("gemm.persistent", persistent)
            capi.tune("gemm.rowwise_epilogue", rowwise)
            row["K%d_%s_us" % (K, tag)] = round(timed(lambda: capi.call("gemm_bf16", Y, X, W, None, M, K, N)), 1)
        capi.tune_reset()
        del X, W, Y
    print(json.dumps(row), flush=True)
