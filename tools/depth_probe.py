import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd import host
T=2048
V=host.GEMMA4_12B["vocab_size"]
TOKS=[int((7919*i+13)%V) for i in range(T)]
for L in (6, 12, 24, 48):
    cfg=dict(host.GEMMA4_12B, num_layers=L)
    g=host.Gemma("bf16", cfg, max_seq=T+16, max_prefill=T, seed=1234)
    pre=g.prefill(TOKS)
    re=g.decode(TOKS[-1], T-1, "fused")
    # also: prefill of the first T-1 tokens then decode the last (fresh cache rows)
    g2=host.Gemma("bf16", cfg, max_seq=T+16, max_prefill=T, seed=1234)
    g2.prefill(TOKS[:-1]) if False else None
    print(L, 'rel', float(np.abs(re-pre).max()/np.abs(pre).max()), 'cos', float(np.dot(re,pre)/np.linalg.norm(re)/np.linalg.norm(pre)), flush=True)
    g.close(); g2.close()
