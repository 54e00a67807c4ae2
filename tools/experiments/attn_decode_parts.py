"""Which part of the decode attention pair's time is the fused prologue (q / k norm + RoPE + KV append), which the band itself?  Launches, on the sliding-window
geometry of Gemma-4 12B (16 heads on 8 KV heads x 256, window 1024) at position 2048, (a) the fused entry as the captured graph calls it (device position),
(b) the plain attn_decode_bf16 on already-roped q and an already-appended cache; read the kernel durations from a rocprofv3 --kernel-trace of this script.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_attn_parts -- python3 tools/experiments/attn_decode_parts.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mila_amd import capi  # noqa: E402

NH, NKV, HS, cap, window, pos = 16, 8, 256, 4096, 1024, 2048
lib = capi.load()
bf = lambda t: t.to(torch.bfloat16).view(torch.int16)
K = bf(torch.randn((1, NKV, cap, HS), device="cuda") * 0.3)
V = bf(torch.randn((1, NKV, cap, HS), device="cuda"))
q, k, v = bf(torch.randn(NH * HS, device="cuda")), bf(torch.randn(NKV * HS, device="cuda")), bf(torch.randn(NKV * HS, device="cuda"))
qw, kw = bf(torch.rand(HS, device="cuda") + 0.5), bf(torch.rand(HS, device="cuda") + 0.5)
cos = torch.rand((cap, HS // 2), device="cuda", dtype=torch.float32)
sin = torch.rand((cap, HS // 2), device="cuda", dtype=torch.float32)
y = torch.empty(NH * HS, dtype=torch.int16, device="cuda")
nb = lib.mila_cdna4_attn_decode_scratch_bytes(1, NH, HS)
scratch = torch.empty(nb, dtype=torch.uint8, device="cuda")
pd = torch.tensor([pos], dtype=torch.int32, device="cuda")
filler = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for i in range(60):
    filler.zero_()          # K / V out of the caches, as between two tokens of a 24 GB model
    capi.call("fused_attn_decode_bf16", y, K, V, q, k, v, qw, kw, None, cos, sin, scratch, C.c_size_t(nb), NH, NKV, HS, cap, pos + 1, pd, window, 1.0, 1e-6)
    filler.zero_()
    capi.call("attn_decode_bf16", y, q, K, V, scratch, C.c_size_t(nb), 1, NH, NKV, HS, cap, pos + 1, window, 1.0)
torch.cuda.synchronize()
print("done")
