"""Where the fixed cost of a decode matvec launch goes: chains of small launches replayed from a graph, microseconds per launch.
Not part of the product path."""
import os
os.environ.setdefault("MILA_CDNA4_TUNING", "1")      # enables the mila_cdna4_tune_* hooks in this process (csrc/internal.h)
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402


def time_chain(fn, n=64, reps=20):
    for _ in range(3):
        fn(0)
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
    lib = capi.load()
    pos = torch.zeros(1, dtype=torch.int32, device="cuda")
    out = {"advance_position(1 thread)": time_chain(lambda i: capi.call("advance_position", pos))}
    nb = 24
    for fmt, fname in ((0, "bf16"), (2, "fp4")):
        for K, N in ((512, 256), (4096, 256), (4096, 3840), (4096, 4096 * 4), (3840, 8192)):
            if fmt == 0:
                Ws = [torch.randint(-30000, 30000, (N, K), dtype=torch.int16, device="cuda") for _ in range(nb)]
                Ss = [None] * nb
            else:
                Ws = [torch.randint(0, 255, (N, K // 2), dtype=torch.uint8, device="cuda") for _ in range(nb)]
                Ss = [torch.rand(N, K // 128, device="cuda") for _ in range(nb)]
            x = torch.randn(K, device="cuda").to(torch.bfloat16).view(torch.int16)
            y = torch.empty(N, dtype=torch.int16, device="cuda")
            for mb in (0, 64, 128):
                capi.tune("matvec.max_workgroups", mb)

                def call(i):
                    if fmt == 0:
                        capi.call("matvec_bf16", y, x, Ws[i % nb], None, K, N)
                    else:
                        capi.call("matvec_bf16_qfp4", y, x, Ws[i % nb], Ss[i % nb], None, K, N, 128)
                out["matvec %s K=%d N=%d max_blocks=%d" % (fname, K, N, mb)] = round(time_chain(call, n=nb * 2), 2)
            del Ws, Ss
    capi.tune_reset()
    for k, v in out.items():
        print(json.dumps({"case": k, "us_per_launch": round(v, 2)}), flush=True)


if __name__ == "__main__":
    main()
