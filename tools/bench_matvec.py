"""Micro-benchmark of the decode matvec on the Gemma-4-12B shapes: GB/s per (shape, format, R, U).
Weights are cycled through enough distinct buffers (> 1 GiB) that the 256 MiB Infinity Cache never
serves a re-read.  Prints one JSON line per configuration.  Not part of the product path."""
import os
os.environ.setdefault("MILA_CDNA4_TUNING", "1")      # enables the mila_cdna4_tune_* hooks in this process (csrc/internal.h)
import argparse
import ctypes as C
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

SHAPES = [("qkv_local", 3840, 8192), ("qkv_global", 3840, 8704), ("o_local", 4096, 3840), ("o_global", 8192, 3840),
          ("gate_up", 3840, 30720), ("down", 15360, 3840), ("lm_head", 3840, 262144)]


def bytes_per_call(fmt, K, N):
    if fmt == 0:
        return N * K * 2
    if fmt == 1:
        return N * K + N * 4
    return N * K // 2 + N * (K // 128) * 4


def run(fmt, K, N, R, U, iters, lib, max_blocks=0, hot=False):
    wbytes = bytes_per_call(fmt, K, N)
    nbuf = max(2, min(64, (1 << 30) // wbytes + 1))
    if hot:   # the same matrix every launch: served from the 256 MiB Infinity Cache when it fits
        nbuf = 1
    if fmt == 0:
        Ws = [torch.randint(-30000, 30000, (N, K), dtype=torch.int16, device="cuda") for _ in range(nbuf)]
        Ss = [None] * nbuf
    elif fmt == 1:
        Ws = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
        Ss = [torch.rand(N, device="cuda") for _ in range(nbuf)]
    else:
        Ws = [torch.randint(0, 255, (N, K // 2), dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
        Ss = [torch.rand(N, K // 128, device="cuda") for _ in range(nbuf)]
    x = torch.randn(K, device="cuda").to(torch.bfloat16).view(torch.int16)
    y = torch.empty(N, dtype=torch.int16, device="cuda")
    capi.tune("matvec.rows_per_wave", R)
    capi.tune("matvec.chunks_in_flight", U)
    capi.tune("matvec.max_workgroups", max_blocks)

    def call(i):
        W, s = Ws[i % nbuf], Ss[i % nbuf]
        if fmt == 0:
            capi.call("matvec_bf16", y, x, W, None, K, N)
        elif fmt == 1:
            capi.call("matvec_bf16_qfp8", y, x, W, s, None, K, N)
        else:
            capi.call("matvec_bf16_qfp4", y, x, W, s, None, K, N, 128)

    for i in range(3):
        call(i)
    torch.cuda.synchronize()
    # capture one pass over all buffers in a graph: device-side timing, no host launch gaps
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(max(nbuf, 8)):
            call(i)
    g.replay()
    torch.cuda.synchronize()
    reps = max(1, iters // max(nbuf, 8))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * max(nbuf, 8))
    return us, wbytes / us / 1e3   # GB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--hot", action="store_true", help="also time each shape re-reading ONE matrix (Infinity-Cache resident)")
    ap.add_argument("--no-ceilings", action="store_true")
    a = ap.parse_args()
    lib = capi.load()
    # ceilings
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
    dst = torch.empty_like(src)
    sink = torch.zeros(4, device="cuda")
    for name in (() if a.no_ceilings else ("stream_copy", "stream_read")):
        for _ in range(3):
            if name == "stream_copy":
                capi.call(name, dst, src, C.c_size_t(n))
            else:
                capi.call(name, sink, src, C.c_size_t(n))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            if name == "stream_copy":
                capi.call(name, dst, src, C.c_size_t(n))
            else:
                capi.call(name, sink, src, C.c_size_t(n))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        moved = n * (2 if name == "stream_copy" else 1)
        print(json.dumps({"kernel": name, "GBps": round(moved / ms / 1e6, 1), "ms": round(ms, 4)}), flush=True)
    del src, dst
    combos = [(0, 0, 0)] if a.quick else [(0, 0, 0), (1, 1, 0), (1, 2, 0), (1, 4, 0), (2, 1, 0), (2, 2, 0), (2, 4, 0), (4, 1, 0), (4, 2, 0), (1, 2, 512), (2, 2, 512), (2, 1, 512), (4, 2, 512)]
    for fmt in (0, 1, 2):
        for name, K, N in SHAPES:
            for R, U, MB in combos:
                us, gbps = run(fmt, K, N, R, U, a.iters if N < 100000 else 30, lib, MB)
                print(json.dumps({"kernel": "matvec", "fmt": ["bf16", "fp8", "fp4"][fmt], "shape": name, "K": K, "N": N,
                                  "R": R, "U": U, "max_blocks": MB, "us": round(us, 2), "GBps": round(gbps, 1),
                                  "frac_of_8TBps": round(gbps / 8000, 3)}), flush=True)
                if a.hot:
                    us, gbps = run(fmt, K, N, R, U, a.iters if N < 100000 else 30, lib, MB, hot=True)
                    print(json.dumps({"kernel": "matvec_hot", "fmt": ["bf16", "fp8", "fp4"][fmt], "shape": name, "K": K, "N": N,
                                      "R": R, "U": U, "us": round(us, 2), "GBps": round(gbps, 1)}), flush=True)
    capi.tune_reset()


if __name__ == "__main__":
    main()
