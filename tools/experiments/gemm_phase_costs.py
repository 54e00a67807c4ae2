"""Where a K-tile of gemm256_kernel<.,2> spends its cycles: the same launch with the staging, the fragment reads or the MFMAs left out
(MILA_GEMM_DBG bits 1 / 2 / 4; a diagnostic build only -- results are garbage, only the time is read)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mila_amd import capi  # noqa: E402

M = 2048
lib = capi.load()
for name, K, N in (("qkv_local", 3840, 8192), ("gate_up", 3840, 30720)):
    X = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16).view(torch.int16)
    W = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16).view(torch.int16)
    Y = torch.empty((M, N), dtype=torch.int16, device="cuda")
    for _ in range(5):
        capi.call("gemm_bf16", Y, X, W, None, M, K, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        capi.call("gemm_bf16", Y, X, W, None, M, K, N)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(json.dumps({"dbg": os.environ.get("MILA_GEMM_DBG", "0"), "shape": name, "us": round(ms * 1e3, 1), "TFLOPs": round(2.0 * M * K * N / ms / 1e9, 1)}), flush=True)
