"""per-token e4m3 activation quantization (the W4A8 prefill's per-forward pass) by row length at M = 2048: us and GB/s (2 B read + 1 B written per element).
The launch-to-launch time of this Python loop is ~10 us whatever the kernel: read the kernel durations from `rocprofv3 --kernel-trace --stats -- python3 tools/bench_quantize_act.py` (round 3: 6.1 / 6.1 / 9.9 / 17.2 us at K = 3840 / 4096 / 8192 / 15360; a wave-per-row
form measured 6.7 / 11.2 us at K = 4096 / 8192 -- slower, not kept).
    python tools/bench_quantize_act.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

capi.load()
M = 2048
for K in (3840, 4096, 8192, 15360):
    # rotate over several buffers so that the rows come from HBM as in the model
    nbuf = 8
    Xs = [(torch.randn((M, K), device="cuda")).to(torch.bfloat16).view(torch.int16) for _ in range(nbuf)]
    X8 = torch.empty((M, K), dtype=torch.uint8, device="cuda")
    ts = torch.empty((M,), dtype=torch.float32, device="cuda")
    for i in range(nbuf):
        capi.call("quantize_fp8_per_token", X8, ts, Xs[i], M, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 40
    for i in range(n):
        capi.call("quantize_fp8_per_token", X8, ts, Xs[i % nbuf], M, K)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(json.dumps({"M": M, "K": K, "us": round(us, 2), "GBps": round(3.0 * M * K / us / 1e3, 1)}), flush=True)
