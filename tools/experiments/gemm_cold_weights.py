"""Is a prefill GEMM slower when its weights come from HBM than when they sit in the Infinity Cache (256 MB)?  The same launch over ONE weight matrix (hot after the
first call when it fits) and rotating over enough matrices that every call streams from HBM, as in the model."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mila_amd import capi  # noqa: E402

M = 2048
capi.load()
for name, K, N in (("qkv_local", 3840, 8192), ("o_local", 4096, 3840), ("gate_up", 3840, 30720), ("down", 15360, 3840)):
    X = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16).view(torch.int16)
    nW = max(2, int(600e6 // (N * K * 2)) + 1)
    Ws = [((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16).view(torch.int16) for _ in range(nW)]
    Y = torch.empty((M, N), dtype=torch.int16, device="cuda")
    for mode in ("one matrix", "rotating"):
        i = [0]

        def go():
            capi.call("gemm_bf16", Y, X, Ws[i[0] % nW if mode == "rotating" else 0], None, M, K, N)
            i[0] += 1
        for _ in range(2 * nW):
            go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 4 * nW
        e0.record()
        for _ in range(n):
            go()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(json.dumps({"shape": name, "weights_MB": round(N * K * 2 / 1e6, 1), "matrices": nW, "mode": mode, "us": round(ms * 1e3, 1), "TFLOPs": round(2.0 * M * K * N / ms / 1e9, 1)}), flush=True)
    del Ws
