"""End-to-end L6 path on the full-size model: GemmaModel.generate (chunked prefill of the prompt, then the decode-ahead loop with a 4-byte token
readback per step) against the bare graph-replay rate bench.py reports.  usage: python tools/bench_generate.py [bf16 fp8 fp4]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from mila_amd import host  # noqa: E402

CONTEXT, PROMPT, NEW = 4096, 2048, 256
rng = np.random.default_rng(0)
prompt = rng.integers(2, 262144, PROMPT).tolist()
for policy in (sys.argv[1:] or ["bf16", "fp8", "fp4"]):
    g = host.GemmaModel.synthetic(policy, None, context=CONTEXT, prefill_chunk=2048, seed=1234)
    unused = [262143]
    g.generate(prompt, max_new_tokens=8, stop_tokens=unused)            # warm-up: graph capture, scratch growth
    t0 = time.perf_counter()
    toks, why, reused = g.generate(prompt[:-1] + [7], max_new_tokens=1, stop_tokens=unused)     # a prompt that differs in its last token: reuse 2047, prefill 1
    t_reuse = time.perf_counter() - t0
    t0 = time.perf_counter()
    toks, why, reused0 = g.generate([9] + prompt[1:], max_new_tokens=1, stop_tokens=unused)     # differs in its first token: full prefill
    t_prefill = time.perf_counter() - t0
    def timed(n, **kw):
        t = time.perf_counter()
        out, reason, _ = g.generate([9] + prompt[1:], max_new_tokens=n, stop_tokens=unused, **kw)     # full reuse but the last position, then n - 1 decode steps
        return time.perf_counter() - t, out, reason
    # the decode rate is the DIFFERENCE of two lengths: the prefix-reuse prefill and the first sample cancel
    ta, _, _ = timed(NEW // 2 + 1)
    tb, toks, why = timed(NEW + 1)
    sa, _, _ = timed(NEW // 2 + 1, temperature=0.8, top_k=64, top_p=0.95, seed=1)
    sb, _, _ = timed(NEW + 1, temperature=0.8, top_k=64, top_p=0.95, seed=1)
    print("%s: time to first token, %d-token prompt: %.1f ms (full prefill) / %.1f ms (prefix reuse %d); greedy generate (graph + device sampler): %.1f tok/s (%s); stochastic (captured step + eager sampler): %.1f tok/s" % (
        policy, PROMPT, t_prefill * 1e3, t_reuse * 1e3, reused, (NEW // 2) / (tb - ta), why, (NEW // 2) / (sb - sa)), flush=True)
    g.close()
