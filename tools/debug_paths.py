import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mila_amd import host
SMALL = dict(vocab_size=1024, embedding_dim=256, num_layers=6, num_heads=4, num_kv_heads=2, head_dim=64, hidden_dim=512,
             global_head_dim=128, num_global_kv_heads=1, window=8, sliding_window_pattern=6, global_rotary_dim=32)
for pol in ("bf16", "fp8", "fp4"):
    ms = {m: host.Gemma(pol, SMALL, max_seq=64, max_prefill=1, seed=7) for m in ("reference", "fused", "graph", "fused2")}
    for pos, tok in enumerate([5, 900, 17, 3, 512, 77, 1023, 0, 42, 256, 8, 640, 99, 1, 300]):
        o = {m: g.decode(tok, pos, "fused" if m == "fused2" else m) for m, g in ms.items()}
        print(pol, pos, "ref==fused", np.array_equal(o["reference"], o["fused"]), "fused==fused2", np.array_equal(o["fused"], o["fused2"]),
              "ref==graph", np.array_equal(o["reference"], o["graph"]), "maxdiff ref-fused %.3e" % np.abs(o["reference"] - o["fused"]).max())
    for g in ms.values():
        g.close()
