"""fp4 policy (W4A8) prefill at ragged prompt lengths: T = 2049 / 2000 / 100 against T = 2048 (within 1.1x)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import host

policies = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fp4"]
if os.environ.get("RAGGED_TUNE"):      # mila_cdna4_tune_gemm codes (csrc/internal.h), e.g. "205" = no split-K, "208,308" = the skinny kernels up to 8 rows only
    os.environ["MILA_CDNA4_TUNING"] = "1"
    from mila_amd import capi
    for code in os.environ["RAGGED_TUNE"].split(","):
        capi.check(capi.load().mila_cdna4_tune_gemm(int(code)))
if os.environ.get("RAGGED_TUNE_FP8"):      # mila_cdna4_tune_gemm_fp8_tail_only codes
    os.environ["MILA_CDNA4_TUNING"] = "1"
    from mila_amd import capi
    for code in os.environ["RAGGED_TUNE_FP8"].split(","):
        capi.check(capi.load().mila_cdna4_tune_gemm_fp8_tail_only(int(code)))
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
