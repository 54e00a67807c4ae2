"""gemm256_kernel schedules side by side on the Gemma prefill shapes: gemm.schedule 4 (two phases per K-tile) against 5 / 6 (6 = 5 with gemm.persistent = 0) (8 / 16 of a
phase's MFMAs issued behind its closing barrier).  Same instruction sequence per accumulator => the outputs must be bit-identical."""
import os
os.environ.setdefault("MILA_CDNA4_TUNING", "1")
import json
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mila_amd import capi  # noqa: E402

M = 2048
lib = capi.load()
scheds = [int(a) for a in sys.argv[1:]] or [4, 5, 6]
for name, K, N in (("qkv_local", 3840, 8192), ("gate_up", 3840, 30720), ("o_local", 4096, 3840), ("down", 15360, 3840)):
    X = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16).view(torch.int16)
    W = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16).view(torch.int16)
    Y = torch.empty((M, N), dtype=torch.int16, device="cuda")
    ref = None
    for rep in range(4):
        for sc in scheds:
            capi.tune("gemm.schedule", 5 if sc == 6 else sc)
            capi.tune("gemm.persistent", 0 if sc == 6 else 1)
            for _ in range(5):
                capi.call("gemm_bf16", Y, X, W, None, M, K, N)
            torch.cuda.synchronize()
            same = None
            if ref is None: ref = Y.clone()
            else: same = bool(torch.equal(ref, Y))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                capi.call("gemm_bf16", Y, X, W, None, M, K, N)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            print(json.dumps({"shape": name, "schedule": sc, "rep": rep, "us": round(ms * 1e3, 1), "TFLOPs": round(2.0 * M * K * N / ms / 1e9, 1), "same_bits": same}), flush=True)
capi.tune_reset()
