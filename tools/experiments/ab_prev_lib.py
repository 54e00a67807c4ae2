"""A/B of two BUILDS of the library in one process on one box (the pool's boxes differ by up to 10 % on clock-bound kernels): tools/experiments/_build/libmila_cdna4_prev.so
(link the current objects with an older gemm256.o) against mila_amd/lib/libmila_cdna4.so, variants interleaved, minimum of four passes.
    python tools/experiments/ab_prev_lib.py > gpurun_out/ab_prev.txt"""
import ctypes as C
import json
import os

import torch

os.environ.setdefault("MILA_CDNA4_TUNING", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = {"current": C.CDLL(os.path.join(ROOT, "mila_amd", "lib", "libmila_cdna4.so")), "previous": C.CDLL(os.path.join(ROOT, "tools", "experiments", "_build", "libmila_cdna4_prev.so"))}


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


CASES = [("N = 4096", "plain", 2048, 3840, 4096), ("N = 16384", "plain", 2048, 3840, 16384), ("gpt2 fc_1 + bias", "bias", 8192, 768, 3072), ("gpt2 qkv + bias", "bias", 8192, 768, 2304), ("gpt2 fc_2 + bias", "bias", 8192, 3072, 768), ("gpt2 lm_head", "plain", 8192, 768, 50257), ("gpt2 proj + bias", "bias", 8192, 768, 768),
         ("gemma qkv", "plain", 2048, 3840, 8192), ("gemma o_proj", "plain", 2048, 4096, 3840), ("gemma fc_gate_up + GeGLU", "geglu", 2048, 3840, 30720), ("gemma fc_down", "plain", 2048, 15360, 3840)]
for name, kind, M, K, N in CASES:
    if N % 128 != 0 and os.environ.get("AB_SKIP_RAGGED"):      # a "previous" build without the ragged-N staging must not see a ragged N
        continue
    X = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16).view(torch.int16)
    W = ((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16).view(torch.int16)
    b = torch.zeros((N,), dtype=torch.int16, device="cuda") if kind == "bias" else None
    Y = torch.empty((M, N // 2 if kind == "geglu" else N), dtype=torch.int16, device="cuda")
    outs, best = {}, {}
    variants = (("previous", "previous", None), ("current", "current", None))
    for rnd in range(4):
        for tag, which, order in (variants if rnd % 2 == 0 else variants[::-1]):
            lib = libs[which]
            if kind == "geglu":
                fn = lambda: lib.mila_cdna4_gemm_geglu_bf16(P(Y), P(X), P(W), M, K, N // 2, None)
            else:
                fn = lambda: lib.mila_cdna4_gemm_bf16(P(Y), P(X), P(W), P(b), M, K, N, None)
            assert fn() == 0
            best[tag] = min(best.get(tag, 1e30), timed(fn))
            outs[tag] = Y.clone()
    row = {"case": name, "M": M, "K": K, "N": N, "same_bits": all(bool(torch.equal(outs["previous"], o)) for o in outs.values())}
    for tag, _, _ in variants:
        row[tag + "_us"] = round(best[tag], 1)
    print(json.dumps(row), flush=True)


# fp8 x fp8 forms (W4A8 / W8A8 prefill): the same A/B
def e4m3_bytes(shape):
    b = torch.randint(0, 256, shape, device="cuda", dtype=torch.uint8)
    return torch.where((b & 0x7F) == 0x7F, b & 0xFE, b)


for name, kind, M, K, N in (("gemma qkv fp8", "fp8", 2048, 3840, 8192), ("gemma o_proj fp8", "fp8", 2048, 4096, 3840), ("gemma fc_gate_up + GeGLU fp8", "fp8geglu", 2048, 3840, 30720),
                            ("gemma fc_down fp8", "fp8", 2048, 15360, 3840)):
    X8, W8 = e4m3_bytes((M, K)), e4m3_bytes((N, K))
    ts = torch.full((M,), 1e-3, device="cuda", dtype=torch.float32)
    ws = torch.full((1,), 1e-3, device="cuda", dtype=torch.float32)
    Y = torch.empty((M, N // 2 if kind == "fp8geglu" else N), dtype=torch.int16, device="cuda")
    outs, best = {}, {}
    variants = (("previous", "previous"), ("current", "current"))
    for rnd in range(4):
        for tag, which in (variants if rnd % 2 == 0 else variants[::-1]):
            lib = libs[which]
            if kind == "fp8geglu":
                fn = lambda: lib.mila_cdna4_gemm_geglu_fp8_scaled(P(Y), P(X8), P(W8), P(ts), P(ws), M, K, N // 2, None)
            else:
                fn = lambda: lib.mila_cdna4_gemm_fp8_scaled(P(Y), P(X8), P(W8), P(ts), P(ws), None, M, K, N, None)
            assert fn() == 0
            best[tag] = min(best.get(tag, 1e30), timed(fn))
            outs[tag] = Y.clone()
    row = {"case": name, "M": M, "K": K, "N": N, "same_bits": bool(torch.equal(outs["previous"], outs["current"]))}
    for tag, _ in variants:
        row[tag + "_us"] = round(best[tag], 1)
    print(json.dumps(row), flush=True)
