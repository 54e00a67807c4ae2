"""fp4 policy (W4A8) prefill at ragged prompt lengths: T = 2049 / 2000 / 100 against T = 2048 (within 1.1x)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import host

policies = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fp4"]
if os.environ.get("RAGGED_TUNE"):      # named tuning variables (csrc/internal.h), e.g. "gemm.splitk=0" = no split-K, "gemm.skinny_ahead_rows=8,gemm.splitk_min_rows=9"
    os.environ["MILA_CDNA4_TUNING"] = "1"
    from mila_amd import capi
    for kv in os.environ["RAGGED_TUNE"].split(","):
        name, value = kv.split("=", 1)
        capi.tune(name, int(value))
out = {}
for pol in policies:
    m = host.Gemma(pol, max_seq=4096, max_prefill=2304, seed=1)
    res = {}
    for T in [int(t) for t in os.environ.get("RAGGED_T", "2048,2049,2000,2303,300,16").split(",")]:
        m.time_prefill(T, 1)
        res[T] = round(m.time_prefill(T, 3), 3)
    m.close()
    out[pol] = res
    print(pol, res, flush=True)
print(json.dumps(out))
