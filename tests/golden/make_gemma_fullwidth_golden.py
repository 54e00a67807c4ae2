"""Generates tests/golden/gemma_fullwidth_logits.npz: the oracle composition (tests/ref_gemma.py, every bf16 rounding on) of a FULL-WIDTH conditioned
Gemma -- D 3840, 16 heads, head size 256 / 512, F 15360, window 1024, four layers of which the last is global, vocabulary 2048 -- on a T = 2048 prompt and
three decode steps past it, for the three weight policies.  This is the geometry bench.py runs, so the GPU side of the test
(tests/test_gemma_fullwidth_gpu.py) goes through the kernel selections the benchmark goes through: the 256 x 256 and 256 x 128 LDS-DMA GEMMs, the fused
GeGLU epilogue, W4A8 on the fp8 matrix cores, the LDS-DMA flash prefill at HS 256 / 512 with a full sliding window, the split-K flash decode.

about 3.7 TFLOP of float64 matrix products per policy: minutes in the build container, too slow for the GPU box -- hence a committed fixture.
The Linear and attention contractions run on float64 BLAS instead of the C oracle's scalar double loops (the same arithmetic up to float64 rounding,
1e-16: far below every bf16 rounding boundary that matters); everything else IS tests/ref_gemma.py.

    python tests/golden/make_gemma_fullwidth_golden.py            # ~15 minutes on 8 cores, ~20 GB of host memory
    python tests/golden/make_gemma_fullwidth_golden.py --w8a8     # the fp8 policy with the opt-in W8A8 prefill -> gemma_fullwidth_logits_w8a8.npz (~5 minutes)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import orc  # noqa: E402
from ref_gemma import CONDITIONED_PROFILE, RefGemma, bf  # noqa: E402

CFG = dict(vocab_size=2048, embedding_dim=3840, num_layers=4, num_heads=16, num_kv_heads=8, head_dim=256, hidden_dim=15360,
           global_head_dim=512, num_global_kv_heads=1, window=1024, sliding_window_pattern=4, global_rotary_dim=128)
SEED, T, STEPS, MAX_SEQ = 11, 2048, 3, 2048 + 8
TOKENS = [(37 * i + 11) % 2048 for i in range(T)]
NEXT = [5, 1900, 77]


class BlasRef(RefGemma):
    """RefGemma with the two contractions on float64 BLAS"""

    def _dense(self, W):
        key = id(W[1])
        if key not in self._dense_cache:
            if W[0] == "bf16":
                Wf, post = orc.from_bf16_bits(W[1]).astype(np.float64), None
            elif W[0] == "fp8":
                Wf, post = orc.E4M3_LUT[W[1]].astype(np.float64), W[2].astype(np.float64)          # scale applied once after the reduction (CudaMatVecBias.Bf16.cu:246-250)
            else:
                Wf, post = orc.dequant_fp4(W[1], W[2], 128).astype(np.float64), None
            self._dense_cache[key] = (Wf, post)                                                     # 16 matrices, 7.3 GB of float64 in all
        return self._dense_cache[key]

    def linear(self, x, W, round_out=True):
        x2 = np.asarray(x, np.float32)
        rows = x2.reshape(-1, x2.shape[-1]).shape[0]
        if W[0] == "fp4" and self.w4a8_prefill and round_out and rows > 1:
            return self._linear_w4a8(x2, W)
        if W[0] == "fp8" and self.w8a8_prefill and round_out and rows > 1:
            return self._linear_w8a8(x2, W)
        if W[0] != "bf16" and self.staged_prefill and round_out and rows > 1:
            Wf = orc.dequant_fp8(W[1], W[2]) if W[0] == "fp8" else orc.dequant_fp4(W[1], W[2], 128)
            y = x2.astype(np.float64) @ bf(Wf).astype(np.float64).T
        else:
            Wf, post = self._dense(W)
            y = x2.astype(np.float64) @ Wf.T
            if post is not None:
                y = y * post[None, :]
        y = y.astype(np.float32)
        return self.r(y) if round_out else y

    def _linear_w4a8(self, x, W):
        key = id(W[1])
        if key not in self._w8:
            ws = orc.fp8_weight_scale_from_groups(W[2])
            self._w8 = {key: (orc.E4M3_LUT[orc.upcast_fp4_to_fp8(W[1], W[2], ws, 128)].astype(np.float64), ws)}
        w8f, ws = self._w8[key]
        shp = x.shape
        x8, ts = orc.quantize_act_fp8_per_token(x.reshape(-1, shp[-1]))
        raw = ((orc.E4M3_LUT[x8].astype(np.float64) @ w8f.T) * float(ws)).astype(np.float32)          # sB * acc
        y = bf(raw).astype(np.float32) * ts.astype(np.float32)[:, None]
        return self.r(y).reshape(shp[:-1] + (w8f.shape[0],))


    def _linear_w8a8(self, x, W):
        Wf, post = self._dense(W)                                                                   # e4m3 values, per-channel scales
        shp = x.shape
        x8, ts = orc.quantize_act_fp8_per_token(x.reshape(-1, shp[-1]))
        y = ((orc.E4M3_LUT[x8].astype(np.float64) @ Wf.T) * post[None, :] * ts.astype(np.float64)[:, None]).astype(np.float32)
        return self.r(y).reshape(shp[:-1] + (Wf.shape[0],))


def blas_attention(q, K, V, pos, window, scale):
    """orc.gqa_attention's semantics (double math, scale before max / exp, keys max(0, t - window + 1) .. t) head by head on BLAS"""
    _, Tq, NH, HS = q.shape
    Tk, NKV = K.shape[1], K.shape[2]
    out = np.empty((1, Tq, NH * HS), np.float32)
    tq = pos + np.arange(Tq)[:, None]
    tk = np.arange(Tk)[None, :]
    visible = (tk <= tq) & ((tk > tq - window) if window > 0 else True)
    for h in range(NH):
        kv = h // (NH // NKV)
        s = (q[0, :, h].astype(np.float64) @ K[0, :, kv].astype(np.float64).T) * float(scale)
        s = np.where(visible, s, -np.inf)
        p = np.exp(s - s.max(axis=1, keepdims=True))
        out[0, :, h * HS:(h + 1) * HS] = ((p @ V[0, :, kv].astype(np.float64)) / p.sum(axis=1, keepdims=True)).astype(np.float32)
    return out


def main():
    real_attention = orc.gqa_attention
    orc.gqa_attention = lambda q, K, V, pos, window, scale: blas_attention(np.asarray(q), np.asarray(K), np.asarray(V), pos, window, scale) if q.shape[1] > 8 else real_attention(q, K, V, pos, window, scale)
    out = {"cfg_keys": np.array(list(CFG)), "cfg_vals": np.array([CFG[k] for k in CFG], dtype=np.int64), "seed": np.int64(SEED), "tokens": np.array(TOKENS, dtype=np.int32),
           "next_tokens": np.array(NEXT, dtype=np.int32), "max_seq": np.int64(MAX_SEQ),
           "profile_keys": np.array(list(CONDITIONED_PROFILE)), "profile_vals": np.array([CONDITIONED_PROFILE[k] for k in CONDITIONED_PROFILE], dtype=np.float64)}
    # --w8a8: the fp8 policy with the opt-in W8A8 prefill only, into its own fixture (gemma_fullwidth_logits_w8a8.npz): the decode steps then read the caches THAT prefill wrote
    w8a8 = "--w8a8" in sys.argv
    for policy in (("fp8",) if w8a8 else ("bf16", "fp8", "fp4")):
        t0 = time.time()
        ref = BlasRef(CFG, policy, SEED, profile=CONDITIONED_PROFILE, staged_prefill=True, w4a8_prefill=True, w8a8_prefill=w8a8)
        ref._dense_cache = {}
        rows = [ref.forward(TOKENS, 0, MAX_SEQ)]
        ref._w8 = {}
        print("%s prefill: %.0f s" % (policy, time.time() - t0), flush=True)
        for i, tok in enumerate(NEXT):
            rows.append(ref.forward([tok], T + i, MAX_SEQ))
        out["logits_" + policy] = np.stack(rows).astype(np.float32)
        print("%s done: %.0f s, max|logit| %.3f" % (policy, time.time() - t0, float(np.abs(rows[0]).max())), flush=True)
        del ref
    name = "gemma_fullwidth_logits_w8a8.npz" if w8a8 else "gemma_fullwidth_logits.npz"
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", os.path.join(HERE, name))


if __name__ == "__main__":
    main()
