"""Where a K-tile of gemm256_kernel (two-phase ping-pong) spends its time: the same launch with the staging, the fragment reads or the MFMAs left out
(diagnostic library built by gemm_skip.sh; MILA_GEMM_SKIP = 1 | 2 | 4 bit mask), bf16 and fp8 x fp8, qkv and fc_gate_up shapes at M = 2048."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mila_amd import capi  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "tools", "experiments", "_build", "libmila_cdna4_gemmskip.so")
capi.load()
M = 2048


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, K, N in (("qkv_local", 3840, 8192), ("gate_up", 3840, 30720)):
    X = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16).view(torch.int16)
    W = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16).view(torch.int16)
    Y = torch.empty((M, N), dtype=torch.int16, device="cuda")
    us = timed(lambda: capi.call("gemm_bf16", Y, X, W, None, M, K, N))
    b = torch.randint(0, 256, (M, K), device="cuda", dtype=torch.uint8)
    X8 = torch.where((b & 0x7F) == 0x7F, b & 0xFE, b)
    b = torch.randint(0, 256, (N, K), device="cuda", dtype=torch.uint8)
    W8 = torch.where((b & 0x7F) == 0x7F, b & 0xFE, b)
    ts = torch.full((M,), 1e-3, device="cuda", dtype=torch.float32)
    ws = torch.full((1,), 1e-3, device="cuda", dtype=torch.float32)
    us8 = timed(lambda: capi.call("gemm_fp8_scaled", Y, X8, W8, ts, ws, None, M, K, N))
    print(json.dumps({"skip": os.environ.get("MILA_GEMM_SKIP", "0"), "shape": name, "bf16_us": round(us, 1), "fp8_us": round(us8, 1)}), flush=True)
