"""Yardstick only (never on the product path): the vendor library's bf16 NT GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the
Gemma-4 12B prefill shapes, beside this repo's hand-written kernels through the C ABI.  TFLOP/s per shape."""
import os
os.environ.setdefault("MILA_CDNA4_TUNING", "1")      # enables the mila_cdna4_tune_* hooks in this process (csrc/internal.h)
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi  # noqa: E402

SHAPES = [("qkv_local", 3840, 8192), ("o_local", 4096, 3840), ("gate_up", 3840, 30720), ("down", 15360, 3840)]
M = 2048


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    capi.load()
    for name, K, N in SHAPES:
        # several distinct weight buffers so that no call re-reads a cached matrix
        nb = 4
        X = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
        Ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(nb)]
        Y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        i = [0]

        def vendor():
            i[0] += 1
            torch.matmul(X, Ws[i[0] % nb].t(), out=Y)

        def ours():
            i[0] += 1
            capi.call("gemm_bf16", Y.view(torch.int16), X.view(torch.int16), Ws[i[0] % nb].view(torch.int16), None, M, K, N)
        fl = 2.0 * M * K * N
        lib = capi.load()
        tv = timeit(vendor)
        capi.tune("gemm.schedule", 0)
        t0 = timeit(ours)
        capi.tune("gemm.schedule", 1)
        t1 = timeit(ours)
        capi.tune("gemm.schedule", 5)      # the default: two phases per K-tile, static priority
        t2 = timeit(ours)
        i[0] = 0; ours()
        y3 = Y.clone()
        # same accumulation order in both schedules: the outputs must be bit-identical
        capi.tune("gemm.schedule", 0); i[0] = 0; ours(); y0 = Y.clone()
        capi.tune("gemm.schedule", 1); i[0] = 0; ours(); same = bool(torch.equal(y0, Y)) and bool(torch.equal(y0, y3))
        ref = (X.float() @ Ws[1].float().t())
        err = float((Y.float() - ref).abs().max() / ref.abs().max())
        print(json.dumps({"shape": name, "M": M, "K": K, "N": N, "vendor_us": round(tv, 1), "vendor_TFLOPs": round(fl / tv / 1e6, 1),
                          "lockstep_us": round(t0, 1), "lockstep_TFLOPs": round(fl / t0 / 1e6, 1),
                          "pingpong_us": round(t1, 1), "pingpong_TFLOPs": round(fl / t1 / 1e6, 1), "pp2_TFLOPs": round(fl / t2 / 1e6, 1), "bit_identical": same, "rel_err_vs_f32": err}), flush=True)


if __name__ == "__main__":
    main()
