"""The reference's published regime beside the headline (Mila/Docs/Discussions/DecodePerformanceCampaign.md:113-117, CHANGELOG.md:235-236:
decode at a 32K-token context, a 22.5K-token chunked prefill): three policies, chunked prefill through the KV caches (bounded ring on the sliding-window layers
optional), then graph-replayed decode at the end of the context.
    python tools/bench_long_context.py [--context 32768] [--prefill-tokens 22528] [--policies bf16,fp8,fp4] [--bounded 0|1]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import host  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--context", type=int, default=32768)
ap.add_argument("--prefill-tokens", type=int, default=22528)
ap.add_argument("--chunk", type=int, default=2048)
ap.add_argument("--policies", default="bf16,fp8,fp4")
ap.add_argument("--bounded", type=int, default=0)
ap.add_argument("--steps", type=int, default=64)
a = ap.parse_args()
HBM = 8000.0
out = {}
for pol in a.policies.split(","):
    cfg = dict(host.GEMMA4_12B, bounded_local_kv=a.bounded)
    m = host.Gemma(pol, cfg, max_seq=a.context + a.steps + 16, max_prefill=a.chunk, seed=1234)
    r = {}
    ms = m.time_prefill_chunked(a.prefill_tokens)
    r["chunked_prefill_%d_ms" % a.prefill_tokens] = round(ms, 2)
    r["chunked_prefill_tok_s"] = round(a.prefill_tokens / ms * 1e3, 1)
    ms = m.time_prefill_chunked(a.context)                      # fills the caches to the context length
    r["chunked_prefill_%d_ms" % a.context] = round(ms, 2)
    info = m.info(a.context)
    t = m.time_decode(a.context, a.steps, 8, "graph")
    r["decode_ms_per_token"] = round(t["wall_ms_per_step"], 4)
    r["decode_tok_s"] = round(1e3 / t["wall_ms_per_step"], 2)
    r["bytes_per_token_GB"] = round(info["decode_bytes_per_token"] / 1e9, 3)
    r["whole_token_frac_of_8TBps"] = round(info["decode_bytes_per_token"] / (t["wall_ms_per_step"] * 1e-3) / 1e9 / HBM, 4)
    out[pol] = r
    print(pol, json.dumps(r), flush=True)
    m.close()
print(json.dumps({"context": a.context, "prefill_tokens": a.prefill_tokens, "chunk": a.chunk, "bounded_local_kv": a.bounded, "policies": out}))
