"""Launch the fp8 x fp8 GEMM (+ GeGLU form on fc_gate_up) a few times (for rocprofv3 --pmc passes).  usage: gemm_fp8_only.py <K> <N> [geglu]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

K, N = int(sys.argv[1]), int(sys.argv[2])
geglu = len(sys.argv) > 3
M = 2048
capi.load()
b = torch.randint(0, 256, (M, K), device="cuda", dtype=torch.uint8)
X8 = torch.where((b & 0x7F) == 0x7F, b & 0xFE, b)
Ws = []
for _ in range(3):
    b = torch.randint(0, 256, (N, K), device="cuda", dtype=torch.uint8)
    Ws.append(torch.where((b & 0x7F) == 0x7F, b & 0xFE, b))
ts = torch.full((M,), 1e-3, device="cuda", dtype=torch.float32)
ws = torch.full((1,), 1e-3, device="cuda", dtype=torch.float32)
Y = torch.empty((M, N // 2 if geglu else N), dtype=torch.int16, device="cuda")
for i in range(9):
    if geglu:
        capi.call("gemm_geglu_fp8_scaled", Y, X8, Ws[i % 3], ts, ws, M, K, N // 2)
    else:
        capi.call("gemm_fp8_scaled", Y, X8, Ws[i % 3], ts, ws, None, M, K, N)
torch.cuda.synchronize()
