"""Full-size prefill, T = 2048: one stream vs the two half-chunks on two streams (GemmaTransformer::setPrefillOverlap).  usage: python tools/bench_prefill_overlap.py [bf16 fp8]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import host  # noqa: E402

for policy in (sys.argv[1:] or ["bf16", "fp8"]):
    m = host.Gemma(policy, None, max_seq=2304, max_prefill=2048, seed=1234)
    base = min(m.time_prefill(2048, 3) for _ in range(2))
    m.set_prefill_overlap(True)
    over = min(m.time_prefill(2048, 3) for _ in range(2))
    print("%s: prefill T=2048 one stream %.2f ms, two half-chunks on two streams %.2f ms (%.3fx)" % (policy, base, over, base / over), flush=True)
    m.close()
