"""Did a round regress a prefill GEMM?  The round-2 library (all of round 2's csrc/ built into tools/experiments/_build/libmila_cdna4_r02.so) against the current one in ONE
process on ONE box, on the Gemma prefill shapes with the weights ROTATING through more buffers than the Infinity Cache holds (as in the model: every layer's weights
come from HBM), variants interleaved, minimum of four passes.
    python tools/experiments/ab_r02_lib.py > gpurun_out/ab_r02.txt"""
import ctypes as C
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
# extra library tags on the command line: tools/experiments/_build/libmila_cdna4_<tag>.so (e.g. builds of intermediate commits, to bisect a regression)
TAGS = ["round2"] + sys.argv[1:]
libs = {"current": C.CDLL(os.path.join(ROOT, "mila_amd", "lib", "libmila_cdna4.so"))}
for t in TAGS:
    libs[t] = C.CDLL(os.path.join(ROOT, "tools", "experiments", "_build", "libmila_cdna4_%s.so" % ("r02" if t == "round2" else t)))
ORDER = ["current"] + TAGS


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def timed(fn, n):
    fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


CASES = [("qkv", "plain", 2048, 3840, 8192), ("o_proj", "plain", 2048, 4096, 3840), ("fc_gate_up + GeGLU", "geglu", 2048, 3840, 30720), ("fc_down", "plain", 2048, 15360, 3840),
         ("gpt2 qkv", "plain", 8192, 768, 2304), ("gpt2 fc_1", "plain", 8192, 768, 3072), ("gpt2 fc_2", "plain", 8192, 3072, 768), ("gpt2 proj", "plain", 8192, 768, 768)]
for name, kind, M, K, N in CASES:
    nbuf = max(2, int(600e6 / (N * K * 2)) + 1)            # > 256 MB of weights in rotation
    X = (torch.rand((M, K), device="cuda") * 2 - 1).to(torch.bfloat16).view(torch.int16)
    Ws = [((torch.rand((N, K), device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16).view(torch.int16) for _ in range(nbuf)]
    Y = torch.empty((M, N // 2 if kind == "geglu" else N), dtype=torch.int16, device="cuda")
    best, outs = {}, {}
    for rnd in range(4):
        for tag in (ORDER if rnd % 2 == 0 else ORDER[::-1]):
            lib = libs[tag]
            if kind == "geglu":
                fn = lambda i: lib.mila_cdna4_gemm_geglu_bf16(P(Y), P(X), P(Ws[i % nbuf]), M, K, N // 2, None)
            else:
                fn = lambda i: lib.mila_cdna4_gemm_bf16(P(Y), P(X), P(Ws[i % nbuf]), None, M, K, N, None)
            assert fn(0) == 0
            best[tag] = min(best.get(tag, 1e30), timed(fn, 2 * nbuf))
            fn(0)
            torch.cuda.synchronize()
            outs[tag] = Y.clone()
    same = bool(torch.equal(outs["current"], outs["round2"]))
    close = float((outs["current"].view(torch.bfloat16).float() - outs["round2"].view(torch.bfloat16).float()).abs().max())
    row = {"case": name, "M": M, "K": K, "N": N, "weight_buffers": nbuf, "same_bits_as_round2": same, "max_abs_diff": close}
    for t in ORDER:
        row[t + "_us"] = round(best[t], 1)
    print(json.dumps(row), flush=True)
    del Ws
