"""Yardstick only (never on the product path): the vendor library's bf16 matrix-vector product (torch.mv / F.linear -> rocBLAS /
hipBLASLt) on the Gemma-4 12B decode shapes beside this repo's matvec through the C ABI; cold weights (cycled buffers), graph replay."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

SHAPES = [("qkv_local", 3840, 8192), ("o_local", 4096, 3840), ("gate_up", 3840, 30720), ("down", 15360, 3840)]


def time_graph(fn, n, reps=10):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


def main():
    capi.load()
    for name, K, N in SHAPES:
        nb = max(4, (1 << 30) // (N * K * 2) + 1)
        Ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(nb)]
        x = torch.randn(K, device="cuda").to(torch.bfloat16)
        x2 = x.view(1, K)
        y = torch.empty(N, dtype=torch.bfloat16, device="cuda")
        y2 = torch.empty(1, N, dtype=torch.bfloat16, device="cuda")
        t_mv = time_graph(lambda i: torch.mv(Ws[i % nb], x, out=y), nb)
        t_lin = time_graph(lambda i: torch.matmul(x2, Ws[i % nb].t(), out=y2), nb)
        t_ours = time_graph(lambda i: capi.call("matvec_bf16", y.view(torch.int16), x.view(torch.int16), Ws[i % nb].view(torch.int16), None, K, N), nb)
        b = N * K * 2
        print(json.dumps({"shape": name, "K": K, "N": N, "torch_mv_us": round(t_mv, 2), "torch_mv_GBps": round(b / t_mv / 1e3), "torch_linear_us": round(t_lin, 2),
                          "torch_linear_GBps": round(b / t_lin / 1e3), "ours_us": round(t_ours, 2), "ours_GBps": round(b / t_ours / 1e3)}), flush=True)


if __name__ == "__main__":
    main()
