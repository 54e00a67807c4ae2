"""Oracle composition of the Gemma-4 forward (decode + prefill) from the CPU oracle's ops, with every
bf16 rounding the component chain performs.  Order restated from GemmaBlock::prefill / ::decode
(/root/reference/Mila/Src/Dnn/Components/Transformers/Gemma/Gemma.Block.ixx:197-356) and
GemmaTransformer::decode (Gemma.ixx:281-297).  TEST INFRASTRUCTURE ONLY.

Block-level parity is "unpinned" in the reference tree (its Gemma tests assert shapes/finiteness only,
SURVEY.md section 4); this composition is pinned op-by-op through the oracle's own pins, and as a WHOLE by
tests/golden/gemma4_hf_logits.npz: with exact=True (no intermediate bf16 rounding) it must reproduce the
FP32 logits of the transformers Gemma4ForCausalLM -- the implementation the reference validates its own
checkpoints against (Tools/Converters/Gemma/gemma_4_BF16/hf_gemma_greedy_validation.py) -- on the same
synthetic weights (tests/golden/make_gemma4_hf_golden.py, tests/test_oracle_kats.py)."""
import numpy as np

import orc
import synth


def bf(x):
    return orc.round_bf16(np.asarray(x, dtype=np.float32))


# synthetic-parameter profile (GemmaTransformer::SyntheticProfile, Mila/Gemma.h): multipliers on the generator of SURVEY.md section 8d
DEFAULT_PROFILE = dict(linear_gain=1.0, qk_norm_center=1.0, post_norm_center=1.0, layer_scalar=1.0, table_gain=1.0)
# a CONDITIONED model: like a trained one, every branch adds a small update to the residual stream (post-norm weights 0.1 against an
# embedding of RMS ~2.3: table_gain 4), attention scores stay O(1) (q/k norm weights 0.35: score std sqrt(HD) * 0.35^2), and the
# per-layer output scale is not 1.  A bf16 rounding difference then stays at the rounding floor instead of growing ~1.4x per layer
# as under unit-scale random weights (tools/condition_probe.py: 12 layers 8e-2 -> 7e-4 of max|logit|).
CONDITIONED_PROFILE = dict(linear_gain=1.0, qk_norm_center=0.35, post_norm_center=0.1, layer_scalar=0.96875, table_gain=4.0)


class RefGemma:
    def __init__(self, cfg, policy, seed, exact=False, profile=None, f32_stand_in=False, staged_prefill=False, w4a8_prefill=False, w4a8_f32_order=False, w8a8_prefill=False):
        self.c = dict(cfg)
        self.policy = policy
        self.exact = exact          # True: keep every intermediate in FP32 (the wiring check against the HF FP32 forward)
        self.p = dict(DEFAULT_PROFILE if profile is None else profile)
        # True: a T > 1 forward under a quantized policy multiplies by bf16(dequantized weight), the arithmetic of the reference's 2-phase
        # prefill (CudaLinearOp.ixx:597-644, :716-764: dequantize to a bf16 scratch, then a bf16 GEMM) that the in-register-dequantizing and
        # the staged GEMMs both restate; False: the decode matvec's arithmetic (scales applied in FP32) for every T
        self.staged_prefill = staged_prefill
        # True (fp4 policy only): a T > 1 forward runs the reference's DEFAULT prefill for PerGroupFp4 -- W4A8 (CudaLinearOp.ixx:646-715): the fp4 weights upcast
        # to e4m3 against the per-tensor scale sB, the activations quantized per token to e4m3, an fp8 x fp8 contraction, and the two-step epilogue
        # bf16(float(bf16(acc * sB)) * s_m).  Takes precedence over staged_prefill for fp4 weights.
        self.w4a8_prefill = w4a8_prefill
        # True (fp8 policy only): a T > 1 forward runs the OPT-IN W8A8 prefill (RocmLinearOp::setFp8ActivationPrefill on PerChannelFp8<>; Policies.ixx:39-40): the policy's own e4m3
        # weights, per-token e4m3 activations (CudaFp8Prefill.cu:108-160), an fp8 x fp8 contraction, y = bf16((acc * scale[n]) * s_m).  Takes precedence over staged_prefill.
        self.w8a8_prefill = w8a8_prefill
        # teacher forcing (tests/test_gemma_conditioned_gpu.py): an iterator of (M, K, N, x8, ts) records -- the per-token e4m3 activations the GPU's Linears consumed,
        # in call order (host.activation_tap) -- which the fp8 x fp8 Linears below multiply INSTEAD of quantizing their own input: the comparison then measures
        # everything but upstream e4m3 code flips
        self.forced = None
        self.w4a8_f32_order = w4a8_f32_order      # tests/test_conditioned_cpu.py: the same arithmetic accumulated in FP32 in another order (a second correct implementation)
        self._w8 = {}
        self.f32_stand_in = f32_stand_in   # tools/condition_probe.py only: FP32 BLAS accumulation as a stand-in for another summation order
        c = self.c
        D, H = c["embedding_dim"], c["hidden_dim"]
        self.layers = []
        for i in range(c["num_layers"]):
            g = (i + 1) % c["sliding_window_pattern"] == 0
            HD = c["global_head_dim"] if g else c["head_dim"]
            NKV = c["num_global_kv_heads"] if g else c["num_kv_heads"]
            NH = c["num_heads"]
            qw, kvw = NH * HD, NKV * HD
            packed = qw + (1 if g else 2) * kvw
            b = seed * 1000003 + i * 64
            L = dict(g=g, HD=HD, NKV=NKV, NH=NH,
                     qkv=self._lin(b + 1, packed, D), o=self._lin(b + 2, D, qw), gu=self._lin(b + 3, 2 * H, D),
                     down=self._lin(b + 4, D, H),
                     input_norm=self._norm(b + 5, D), q_norm=self._norm(b + 6, HD, self.p["qk_norm_center"]), k_norm=self._norm(b + 7, HD, self.p["qk_norm_center"]),
                     post_attn=self._norm(b + 8, D, self.p["post_norm_center"]), pre_ffn=self._norm(b + 9, D), post_ffn=self._norm(b + 10, D, self.p["post_norm_center"]),
                     K=np.zeros((1, 0, NKV, HD), np.float32), V=np.zeros((1, 0, NKV, HD), np.float32))
            self.layers.append(L)
        nl = c["num_layers"]
        self.final_norm = self._norm(seed * 1000003 + 64 * nl + 1, D)
        tb = synth.fill_bf16(seed * 1000003 + 64 * nl + 2, c["vocab_size"] * D, np.float32(self.p["table_gain"]) / np.sqrt(np.float32(D)), 0.0).reshape(c["vocab_size"], D)
        if policy == "bf16":
            self.table = ("bf16", tb)
        else:
            self.table = ("fp8",) + orc.quantize_fp8_per_channel(tb)
        self.rope = {}

    def _lin(self, seed, N, K):
        wb = synth.fill_bf16(seed, N * K, np.float32(self.p["linear_gain"]) / np.sqrt(np.float32(K)), 0.0).reshape(N, K)
        if self.policy == "bf16":
            return ("bf16", wb)
        if self.policy == "fp8":
            return ("fp8",) + orc.quantize_fp8_per_channel(wb)
        return ("fp4",) + orc.quantize_fp4_per_group(wb, 128)

    @staticmethod
    def _norm(seed, n, center=1.0):
        return orc.from_bf16_bits(synth.fill_bf16(seed, n, np.float32(0.1) * np.float32(center), np.float32(center)))

    def r(self, x):
        return np.asarray(x, dtype=np.float32) if self.exact else bf(x)

    def linear(self, x, W, round_out=True):
        if self.f32_stand_in:
            Wf = orc.from_bf16_bits(W[1]) if W[0] == "bf16" else orc.dequant_fp8(W[1], W[2]) if W[0] == "fp8" else orc.dequant_fp4(W[1], W[2], 128)
            y = (np.asarray(x, np.float32)[..., ::-1] @ np.ascontiguousarray(Wf[:, ::-1].T)).astype(np.float32)
            return self.r(y) if round_out else y
        rows = np.asarray(x).reshape(-1, np.asarray(x).shape[-1]).shape[0]
        if W[0] == "fp4" and self.w4a8_prefill and round_out and rows > 1:
            return self._linear_w4a8(x, W)
        if W[0] == "fp8" and self.w8a8_prefill and round_out and rows > 1:
            return self._linear_w8a8(x, W)
        if W[0] == "bf16":
            y = orc.linear_bf16w(x, W[1])
        elif self.staged_prefill and round_out and np.asarray(x).reshape(-1, np.asarray(x).shape[-1]).shape[0] > 1:
            Wf = orc.dequant_fp8(W[1], W[2]) if W[0] == "fp8" else orc.dequant_fp4(W[1], W[2], 128)
            y = orc.linear_bf16w(x, orc.to_bf16_bits(Wf))
        elif W[0] == "fp8":
            y = orc.linear_fp8w(x, W[1], W[2])
        else:
            y = orc.linear_fp4w(x, W[1], W[2], 128)
        return self.r(y) if round_out else y

    def _linear_w4a8(self, x, W):
        """CudaW4A16Gemm.cu:244-323 (sB, fp4 -> e4m3), CudaFp8Prefill.cu:108-211 (per-token activations, epilogue); oracle ops of tests/orc.py"""
        key = id(W[1])
        if key not in self._w8:
            ws = orc.fp8_weight_scale_from_groups(W[2])
            self._w8[key] = (orc.upcast_fp4_to_fp8(W[1], W[2], ws, 128), ws)
        w8, ws = self._w8[key]
        x2 = np.asarray(x, dtype=np.float32)
        shp = x2.shape
        x8, ts = self._activations(x2.reshape(-1, shp[-1]), w8.shape[0])
        if self.w4a8_f32_order:
            a = orc.E4M3_LUT[x8].astype(np.float32)[:, ::-1]
            b = np.ascontiguousarray(orc.E4M3_LUT[w8].astype(np.float32)[:, ::-1].T)
            raw = (a @ b).astype(np.float32) * np.float32(ws)
        else:
            raw = orc.linear_fp8a_fp8w(x8, np.ones(len(ts), dtype=np.float32), w8, None, ws)      # sB * acc
        y = bf(raw).astype(np.float32) * ts.astype(np.float32)[:, None]
        return self.r(y).reshape(shp[:-1] + (w8.shape[0],))

    def _linear_w8a8(self, x, W):
        """the fp8 policy's weights as they are stored (e4m3 [N, K] + scale[N]) x per-token e4m3 activations: (acc * scale[n]) * s_m, one rounding (csrc/common.h: w8a8_scale_bias)"""
        x2 = np.asarray(x, dtype=np.float32)
        shp = x2.shape
        x8, ts = self._activations(x2.reshape(-1, shp[-1]), W[1].shape[0])
        y = orc.linear_fp8a_fp8w(x8, ts, W[1], W[2], 1.0)
        return self.r(y).reshape(shp[:-1] + (W[1].shape[0],))

    def _activations(self, x2, N):
        """per-token e4m3 activations of an fp8 x fp8 Linear: the oracle's own quantization, or -- teacher-forced -- the next record of the GPU's"""
        if self.forced is None:
            return orc.quantize_act_fp8_per_token(x2)
        M, K, Nr, x8, ts = next(self.forced)
        assert (M, K, Nr) == (x2.shape[0], x2.shape[1], N), "teacher forcing out of step: GPU record %s, oracle Linear %s" % ((M, K, Nr), (x2.shape[0], x2.shape[1], N))
        own8, own_ts = orc.quantize_act_fp8_per_token(x2)
        self.forced_flips = getattr(self, "forced_flips", 0) + int(np.count_nonzero(own8 != x8))
        self.forced_codes = getattr(self, "forced_codes", 0) + int(x8.size)
        return x8, ts

    def rms(self, x, w):
        return self.r(orc.rmsnorm(x, w, None, eps=1e-6))

    def embed(self, tokens):
        D = self.c["embedding_dim"]
        s = np.sqrt(np.float32(D))
        if self.table[0] == "bf16":
            rows = orc.from_bf16_bits(self.table[1][tokens])
        else:
            q, sc = self.table[1], self.table[2]
            rows = self.r(orc.E4M3_LUT[q[tokens]] * sc[tokens][:, None])
        return self.r(rows * s)

    def _rope_cache(self, L, max_seq):
        key = (L["HD"], L["g"])
        if key not in self.rope:
            c = self.c
            self.rope[key] = orc.rope_build_cache(max_seq, L["HD"], 1e6 if L["g"] else 1e4, c["global_rotary_dim"] if L["g"] else 0)
        return self.rope[key]

    def block(self, x, L, pos, max_seq):
        """x [T, D] (bf16-valued f32); positions pos..pos+T-1; appends to the layer's K/V history"""
        T = x.shape[0]
        NH, NKV, HD = L["NH"], L["NKV"], L["HD"]
        qkv = self.linear(self.rms(x, L["input_norm"]), L["qkv"])
        q = qkv[:, :NH * HD].reshape(T, NH, HD)
        k = qkv[:, NH * HD:NH * HD + NKV * HD].reshape(T, NKV, HD)
        v = k if L["g"] else qkv[:, NH * HD + NKV * HD:].reshape(T, NKV, HD)
        qn = self.rms(q, L["q_norm"])
        kn = self.rms(k, L["k_norm"])
        cos, sin = self._rope_cache(L, max_seq)
        qr = self.r(orc.rope_rotate(qn[None], cos, sin, pos))[0]
        kr = self.r(orc.rope_rotate(kn[None], cos, sin, pos))[0]
        vn = self.rms(v, np.ones(HD, np.float32))
        L["K"] = np.concatenate([L["K"][:, :pos], kr[None]], axis=1)
        L["V"] = np.concatenate([L["V"][:, :pos], vn[None]], axis=1)
        window = 0 if L["g"] else self.c["window"]
        attn = self.r(orc.gqa_attention(qr[None], L["K"], L["V"], pos, window, 1.0))[0]
        o = self.linear(attn, L["o"])
        res1 = self.r(x + self.rms(o, L["post_attn"]))
        gu = self.linear(self.rms(res1, L["pre_ffn"]), L["gu"])
        act = self.r(orc.geglu(gu))
        dn = self.linear(act, L["down"])
        res2 = self.r(res1 + self.rms(dn, L["post_ffn"]))
        return self.r(res2 * np.float32(self.p["layer_scalar"]))

    def forward(self, tokens, pos, max_seq):
        x = self.embed(np.asarray(tokens, dtype=np.int64))
        for L in self.layers:
            x = self.block(x, L, pos, max_seq)
        last = self.rms(x[-1:], self.final_norm)
        return self.linear(last, self.table, round_out=False)[0]
