"""bf16-policy prefill (ms) at ragged / short prompt lengths by the tile count from which the 256 x 128 LDS-DMA ring is taken whatever its last round's fill
(gemm.ldsdma_loose_tiles = n; 0 = the fill rule only).    MILA_CDNA4_TUNING=1 python tools/experiments/bf16_ragged_rules.py"""
import json
import os
import sys

os.environ.setdefault("MILA_CDNA4_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mila_amd import capi, host  # noqa: E402

lib = capi.load()
m = host.Gemma("bf16", max_seq=4096, max_prefill=2304, seed=1)
out = {}
for loose in (0, 30, 60, 120):
    capi.tune("gemm.ldsdma_loose_tiles", loose)
    res = {}
    for T in (100, 200, 300, 400, 511, 700, 1000, 2100, 2303):
        m.time_prefill(T, 1)
        res[T] = round(m.time_prefill(T, 2), 3)
    out["loose_%d" % loose] = res
    print(loose, res, flush=True)
capi.tune_reset()
m.close()
print(json.dumps(out))
