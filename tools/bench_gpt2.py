"""BASELINE config 2: GPT-2 124M bf16 forward, B = 8, T = 1024, on 1 x MI355X through the GptTransformer mirror (random-init parameters, synthetic tokens).
    python tools/bench_gpt2.py [reps]          ->  one JSON line: ms, TFLOP/s, fraction of the bf16 MFMA peak"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import host  # noqa: E402

V, maxT, C_, L, NH, B, T = 50257, 1024, 768, 12, 12, 8, 1024
FLOP = 2.0 * B * T * (L * 12 * C_ * C_ + C_ * V) + L * 4.0 * B * NH * (C_ // NH) * (T * (T + 1) / 2)      # Linear + causal attention


def bf16_bits(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def params(rng):
    def t(*shape, scale=0.02, offset=0.0):
        return bf16_bits(rng.standard_normal(shape).astype(np.float32) * np.float32(scale) + np.float32(offset))
    ps = [t(V, C_, scale=0.05), t(maxT, C_, scale=0.02)]
    for _ in range(L):
        ps += [t(C_, scale=0.1, offset=1.0), t(C_, scale=0.05), t(3 * C_, C_, scale=C_ ** -0.5), t(3 * C_, scale=0.02), t(C_, C_, scale=C_ ** -0.5), t(C_, scale=0.02),
               t(C_, scale=0.1, offset=1.0), t(C_, scale=0.05), t(4 * C_, C_, scale=C_ ** -0.5), t(4 * C_, scale=0.02), t(C_, 4 * C_, scale=(4 * C_) ** -0.5), t(C_, scale=0.02)]
    ps += [t(C_, scale=0.1, offset=1.0), t(C_, scale=0.05), t(V, C_, scale=C_ ** -0.5)]
    return ps


def run(reps=5):
    rng = np.random.default_rng(124)
    g = host.Gpt(V, maxT, C_, L, NH, B, T)
    g.load_parameters(params(rng))
    tokens = rng.integers(0, V, (B, T)).astype(np.int32)
    ms = []
    for _ in range(reps + 2):
        g.forward_timed(tokens)
        ms.append(g.last_ms)
    g.close()
    best = float(np.median(ms[2:]))
    return {"workload": "gpt2_124M_bf16_B8_T1024", "ms": round(best, 3), "TFLOPs": round(FLOP / best / 1e9, 1), "mfma_frac": round(FLOP / best / 1e9 / 2500.0, 4),
            "flop": FLOP, "data": "synthetic (random-init GPT-2 124M, random tokens)"}


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 5)))
