import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mila_amd import host
SMALL = dict(vocab_size=1024, embedding_dim=256, num_layers=6, num_heads=4, num_kv_heads=2, head_dim=64, hidden_dim=512,
             global_head_dim=128, num_global_kv_heads=1, window=16, sliding_window_pattern=3, global_rotary_dim=32)
g = host.GemmaModel.synthetic("fp4", SMALL, context=128, prefill_chunk=16, seed=3)
rng = np.random.default_rng(0)
ref = None
t0 = time.time()
for it in range(300):
    prompt = rng.integers(2, 1024, int(rng.integers(1, 60))).tolist()
    kw = dict(top_k=1) if it % 3 else dict(temperature=0.9, top_k=40, top_p=0.9, seed=it)
    out, why, reused = g.generate(prompt, max_new_tokens=int(rng.integers(1, 40)), stop_tokens=[1023], **kw)
    assert all(0 <= t < 1024 for t in out)
    if it % 50 == 0:
        a, _, _ = g.generate([5, 6, 7, 8], max_new_tokens=12, stop_tokens=[1023])
        ref = ref or a
        assert a == ref, (a, ref)
print("300 mixed generate() calls ok in %.1f s" % (time.time() - t0))
g.close()
