"""Launch this repo's bf16 GEMM on one Gemma prefill shape a few times (for rocprofv3 --pmc passes).  usage: gemm_only.py <K> <N> [schedule]"""
import os
os.environ.setdefault("MILA_CDNA4_TUNING", "1")      # enables the mila_cdna4_tune_* hooks in this process (csrc/internal.h)
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

K, N = int(sys.argv[1]), int(sys.argv[2])
lib = capi.load()
if len(sys.argv) > 3:
    capi.tune("gemm.schedule", int(sys.argv[3]))
M = 2048
X = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
Ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(3)]
Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for i in range(int(os.environ.get("MILA_GEMM_LAUNCHES", "9"))):
    capi.call("gemm_bf16", Y.view(torch.int16), X.view(torch.int16), Ws[i % 3].view(torch.int16), None, M, K, N)
torch.cuda.synchronize()
