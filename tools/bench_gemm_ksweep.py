"""Where does a prefill GEMM's time go?  time(K) at fixed M x N is a line: slope = the K loop (per K-tile), intercept = everything a tile costs outside it
(launch, the first K-tiles' latency, epilogue arithmetic, the write burst).  bf16 and fp8 x fp8, plain and GeGLU epilogues, one-round and multi-round grids.
    python tools/bench_gemm_ksweep.py > gpurun_out/ksweep.txt"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

M = 2048
lib = capi.load()


def timed(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def e4m3_bytes(shape):
    b = torch.randint(0, 256, shape, device="cuda", dtype=torch.uint8)
    return torch.where((b & 0x7F) == 0x7F, b & 0xFE, b)


KS = (1920, 3840, 7680, 15360)
rows = []
for name, N, geglu in (("N=8192 (256 tiles, one round)", 8192, False), ("N=30720 (960 tiles)", 30720, False), ("F=15360 + GeGLU (960 tiles)", 30720, True), ("N=3840 (240 tiles of 256x128)", 3840, False)):
    for fmt in ("bf16", "fp8"):
        us = []
        for K in KS:
            if fmt == "bf16":
                X = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16).view(torch.int16)
                W = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16).view(torch.int16)
                if geglu:
                    Y = torch.empty((M, N // 2), dtype=torch.int16, device="cuda")
                    t = timed(lambda: capi.call("gemm_geglu_bf16", Y, X, W, M, K, N // 2))
                else:
                    Y = torch.empty((M, N), dtype=torch.int16, device="cuda")
                    t = timed(lambda: capi.call("gemm_bf16", Y, X, W, None, M, K, N))
            else:
                X8, W8 = e4m3_bytes((M, K)), e4m3_bytes((N, K))
                ts = torch.full((M,), 1e-3, device="cuda", dtype=torch.float32)
                ws = torch.full((1,), 1e-3, device="cuda", dtype=torch.float32)
                if geglu:
                    Y = torch.empty((M, N // 2), dtype=torch.int16, device="cuda")
                    t = timed(lambda: capi.call("gemm_geglu_fp8_scaled", Y, X8, W8, ts, ws, M, K, N // 2))
                else:
                    Y = torch.empty((M, N), dtype=torch.int16, device="cuda")
                    t = timed(lambda: capi.call("gemm_fp8_scaled", Y, X8, W8, ts, ws, None, M, K, N))
            us.append(t)
        kt = 64 if fmt == "bf16" else 128
        nk = [K // kt for K in KS]
        slope = (us[-1] - us[0]) / (nk[-1] - nk[0])
        icpt = us[1] - slope * nk[1]
        row = {"case": name, "fmt": fmt, "us_by_K": dict(zip(KS, [round(u, 1) for u in us])), "us_per_ktile_per_launch": round(slope, 3), "fixed_us_per_launch": round(icpt, 1),
               "TFLOPs_at_K3840": round(2.0 * M * 3840 * N / us[1] / 1e6, 1)}
        rows.append(row)
        print(json.dumps(row), flush=True)
