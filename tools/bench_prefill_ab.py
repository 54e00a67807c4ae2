"""Same-box, same-process A/B of a named tuning variable on the T = 2048 prefill of the full Gemma-4 12B model: the model is built once per policy and the variants are
timed interleaved (the pool's boxes differ by 3-8 % on clock-bound kernels, more than most effects).
usage: python tools/bench_prefill_ab.py NAME V0 V1 [--policies bf16,fp8,fp8-w8a8,fp4] [--rounds 3]"""
import os
os.environ.setdefault("MILA_CDNA4_TUNING", "1")
import argparse
import json
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import capi, host  # noqa: E402


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("values", nargs="+", type=int)
    ap.add_argument("--policies", default="bf16,fp8-w8a8,fp4")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--T", type=int, default=2048)
    a = ap.parse_args()
    capi.load()
    for pol in a.policies.split(","):
        w8a8 = pol == "fp8-w8a8"
        m = host.Gemma("fp8" if w8a8 else pol, dict(host.GEMMA4_12B), max_seq=a.T + 8, max_prefill=a.T, seed=1234)
        if w8a8:
            m.set_fp8_activation_prefill(True)
        m.time_prefill(a.T, 1)
        res = {v: [] for v in a.values}
        for _ in range(a.rounds):
            for v in a.values:
                capi.tune(a.name, v)
                m.time_prefill(a.T, 1)
                res[v] += [m.time_prefill(a.T, 1) for _ in range(3)]
        capi.tune_reset()
        print(json.dumps({"policy": pol, "variable": a.name, "T": a.T, "median_ms": {str(v): round(median(x), 3) for v, x in res.items()},
                          "min_ms": {str(v): round(min(x), 3) for v, x in res.items()}}), flush=True)
        m.close()


if __name__ == "__main__":
    main()
