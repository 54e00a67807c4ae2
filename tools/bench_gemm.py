"""TFLOP/s of the prefill GEMM on the Gemma-4-12B shapes (M = 2048), 256-tile vs 128-tile kernel."""
import os
os.environ.setdefault("MILA_CDNA4_TUNING", "1")      # enables the mila_cdna4_tune_* hooks in this process (csrc/internal.h)
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

SHAPES = [("qkv_local", 3840, 8192), ("qkv_global", 3840, 8704), ("o_local", 4096, 3840), ("o_global", 8192, 3840),
          ("gate_up", 3840, 30720), ("down", 15360, 3840)]
M = 2048
lib = capi.load()
for name, K, N in SHAPES:
    X = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16).view(torch.int16)
    W = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16).view(torch.int16)
    Y = torch.empty((M, N), dtype=torch.int16, device="cuda")
    for force in (0, 1):
        capi.tune("gemm.force128", force)
        for _ in range(3):
            capi.call("gemm_bf16", Y, X, W, None, M, K, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            capi.call("gemm_bf16", Y, X, W, None, M, K, N)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(json.dumps({"shape": name, "M": M, "K": K, "N": N, "kernel": "128x128 regstage" if force else "auto (256x256 glds if applicable)",
                          "us": round(ms * 1e3, 1), "TFLOPs": round(2.0 * M * K * N / ms / 1e9, 1)}), flush=True)
capi.tune_reset()
