"""usage: bench_gemm_fp8.py [schedule].  TFLOP/s of the fp8 x fp8 prefill GEMM (the W4A8 path's contraction: e4m3 activations x e4m3 weights, per-token and per-tensor scales in the epilogue) on the
Gemma-4-12B shapes at M = 2048, and of its GeGLU form on fc_gate_up."""
import os
if len(__import__("sys").argv) > 1: os.environ.setdefault("MILA_CDNA4_TUNING", "1")      # a schedule argument: enable the tuning hooks (csrc/internal.h)
import json
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

M = 2048
lib = capi.load()
if len(sys.argv) > 1:
    capi.tune("gemm.schedule", int(sys.argv[1]))      # 3: fp8 shapes prefer the 256 x 128 ring; 4 (default): the two-phase 256 x 256 kernel wherever it applies


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def e4m3_bytes(shape):
    b = torch.randint(0, 256, shape, device="cuda", dtype=torch.uint8)
    return torch.where((b & 0x7F) == 0x7F, b & 0xFE, b)          # no NaN encodings


for name, K, N in (("qkv_local", 3840, 8192), ("o_local", 4096, 3840), ("gate_up", 3840, 30720), ("down", 15360, 3840)):
    X8, W8 = e4m3_bytes((M, K)), e4m3_bytes((N, K))
    ts = torch.full((M,), 1e-3, device="cuda", dtype=torch.float32)
    ws = torch.full((1,), 1e-3, device="cuda", dtype=torch.float32)
    Y = torch.empty((M, N), dtype=torch.int16, device="cuda")
    ms = timed(lambda: capi.call("gemm_fp8_scaled", Y, X8, W8, ts, ws, None, M, K, N))
    print(json.dumps({"schedule": sys.argv[1] if len(sys.argv) > 1 else "default", "shape": name, "M": M, "K": K, "N": N, "us": round(ms * 1e3, 1), "TFLOPs": round(2.0 * M * K * N / ms / 1e9, 1)}), flush=True)
    if name == "gate_up":
        F = N // 2
        Yg = torch.empty((M, F), dtype=torch.int16, device="cuda")
        ms = timed(lambda: capi.call("gemm_geglu_fp8_scaled", Yg, X8, W8, ts, ws, M, K, F))
        print(json.dumps({"shape": "gate_up + GeGLU", "M": M, "K": K, "F": F, "us": round(ms * 1e3, 1), "TFLOPs": round(2.0 * M * K * N / ms / 1e9, 1)}), flush=True)
