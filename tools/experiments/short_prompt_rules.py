"""fp4-policy prefill of SHORT prompts by dispatch rule of the W4A8 GEMMs below 512 rows (gemm_fp8.big_rule 0 / 1 / 2 / 3):
0 = masked 128-row tiles (default), 1 = LDS-DMA kernels from 128 rows on, 2 = LDS-DMA kernels where their grid has >= 120 tiles.
    MILA_CDNA4_TUNING=1 python tools/experiments/short_prompt_rules.py"""
import json
import os
import sys

os.environ.setdefault("MILA_CDNA4_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mila_amd import capi, host  # noqa: E402

lib = capi.load()
m = host.Gemma("fp4", max_seq=2048, max_prefill=1024, seed=1)
out = {}
for rule in (0, 1, 2, 3):
    capi.tune("gemm_fp8.big_rule", rule)
    res = {}
    for T in (65, 100, 200, 300, 320, 400, 511, 700, 1000):
        m.time_prefill(T, 1)
        res[T] = round(m.time_prefill(T, 3), 3)
    out["rule%d" % rule] = res
    print(rule, res, flush=True)
capi.tune_reset()
m.close()
print(json.dumps(out))
