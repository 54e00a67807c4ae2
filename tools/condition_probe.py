"""CPU-only: how a bf16 rounding difference propagates through N Gemma blocks under a synthetic-parameter profile.
Compares the oracle composition (double accumulation) with the same composition on FP32 BLAS accumulation in reversed order -- a stand-in
for "another correct kernel" -- and prints max|dlogit| / max|logit| per depth.  Used to choose tests/ref_gemma.py CONDITIONED_PROFILE."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ref_gemma import CONDITIONED_PROFILE, DEFAULT_PROFILE, RefGemma  # noqa: E402

CFG = dict(vocab_size=2048, embedding_dim=512, num_heads=4, num_kv_heads=2, head_dim=64, hidden_dim=1024, global_head_dim=128,
           num_global_kv_heads=1, window=8, sliding_window_pattern=6, global_rotary_dim=32)
TOK = [(7 * i + 3) % 2048 for i in range(20)]
for name, prof in (("default", DEFAULT_PROFILE), ("conditioned", CONDITIONED_PROFILE)):
    for policy in sys.argv[1:] or ["bf16"]:
        for L in (6, 12, 24):
            cfg = dict(CFG, num_layers=L)
            a, b = RefGemma(cfg, policy, 7, profile=prof), RefGemma(cfg, policy, 7, profile=prof, f32_stand_in=True)
            la, lb = a.forward(TOK, 0, 64), b.forward(TOK, 0, 64)
            print("%-12s %-5s L=%2d  rel %.2e   max|logit| %.3f" % (name, policy, L, np.abs(la - lb).max() / np.abs(la).max(), np.abs(la).max()), flush=True)
