"""Generates tests/golden/gemma4_hf_logits.npz: FP32 logits of the transformers Gemma4ForCausalLM (the implementation the
reference validates its Gemma-4 checkpoints against: Mila/Tools/Converters/Gemma/gemma_4_BF16/hf_gemma_greedy_validation.py,
hf_gemma_activation_dump.py) on a small Gemma-4-shaped configuration with this repo's synthetic weights.

The reference's own Gemma tests assert shapes and finiteness only and its HF token-parity test needs a 24 GB checkpoint that is
not available offline, so this third-party forward is what pins the Gemma block WIRING of the
oracle composition tests/ref_gemma.py (sandwich norms, K = V and the scale-free V norm on global layers, partial rotary on
the global layers, sliding window, attention scale 1.0, sqrt(D) embedding scale, tied head, layer scalar).

Run in the build container (transformers 5.x + torch CPU, no network: the model is built from a config object):

    python tests/golden/make_gemma4_hf_golden.py

The fixture holds inputs and expected outputs only: config, seed, token ids, prefix lengths and the logits at the last position
of every prefix.  Weights are regenerated from the seed by tests/synth.py (counter-based, identical on every machine)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import orc  # noqa: E402
from ref_gemma import RefGemma  # noqa: E402

CFG = dict(vocab_size=1024, embedding_dim=256, num_layers=6, num_heads=4, num_kv_heads=2, head_dim=64, hidden_dim=512,
           global_head_dim=128, num_global_kv_heads=1, window=8, sliding_window_pattern=6, global_rotary_dim=32)
SEED = 7
TOKENS = [5, 900, 17, 3, 512, 77, 1023, 0, 42, 256, 8, 640, 99, 1, 300, 711, 64, 2, 450, 31]
PREFIXES = [1, 2, 5, 8, 9, 13, 20]          # below, at and beyond the sliding window of 8


def hf_model(ref):
    from transformers.models.gemma4.configuration_gemma4 import Gemma4TextConfig
    from transformers.models.gemma4.modeling_gemma4 import Gemma4ForCausalLM
    c = ref.c
    hc = Gemma4TextConfig(vocab_size=c["vocab_size"], hidden_size=c["embedding_dim"], intermediate_size=c["hidden_dim"],
                          num_hidden_layers=c["num_layers"], num_attention_heads=c["num_heads"], num_key_value_heads=c["num_kv_heads"],
                          head_dim=c["head_dim"], global_head_dim=c["global_head_dim"], num_global_key_value_heads=c["num_global_kv_heads"],
                          sliding_window=c["window"], attention_k_eq_v=True, hidden_size_per_layer_input=0, max_position_embeddings=64,
                          rms_norm_eps=1e-6, final_logit_softcapping=None, tie_word_embeddings=True,
                          rope_parameters={"sliding_attention": {"rope_type": "default", "rope_theta": 10000.0},
                                           "full_attention": {"rope_type": "proportional", "rope_theta": 1000000.0,
                                                              "partial_rotary_factor": c["global_rotary_dim"] / c["global_head_dim"]}})
    hc._attn_implementation = "eager"
    m = Gemma4ForCausalLM(hc).float().eval()

    def t(bits):
        return torch.from_numpy(orc.from_bf16_bits(bits).copy())

    def tn(w):
        return torch.from_numpy(np.asarray(w, dtype=np.float32).copy())

    sd = {}
    for i, L in enumerate(ref.layers):
        NH, NKV, HD = L["NH"], L["NKV"], L["HD"]
        pre = "model.layers.%d." % i
        qkv = t(L["qkv"][1])
        sd[pre + "self_attn.q_proj.weight"] = qkv[:NH * HD]
        sd[pre + "self_attn.k_proj.weight"] = qkv[NH * HD:NH * HD + NKV * HD]
        if not L["g"]:
            sd[pre + "self_attn.v_proj.weight"] = qkv[NH * HD + NKV * HD:]
        sd[pre + "self_attn.o_proj.weight"] = t(L["o"][1])
        gu = t(L["gu"][1])
        H = c["hidden_dim"]
        sd[pre + "mlp.gate_proj.weight"] = gu[:H]
        sd[pre + "mlp.up_proj.weight"] = gu[H:]
        sd[pre + "mlp.down_proj.weight"] = t(L["down"][1])
        sd[pre + "self_attn.q_norm.weight"] = tn(L["q_norm"])
        sd[pre + "self_attn.k_norm.weight"] = tn(L["k_norm"])
        sd[pre + "input_layernorm.weight"] = tn(L["input_norm"])
        sd[pre + "post_attention_layernorm.weight"] = tn(L["post_attn"])
        sd[pre + "pre_feedforward_layernorm.weight"] = tn(L["pre_ffn"])
        sd[pre + "post_feedforward_layernorm.weight"] = tn(L["post_ffn"])
        sd[pre + "layer_scalar"] = torch.ones(1)
    sd["model.norm.weight"] = tn(ref.final_norm)
    sd["model.embed_tokens.weight"] = t(ref.table[1])
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "inv_freq" not in k and "embed_scale" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    return m


def main():
    torch.manual_seed(0)
    ref = RefGemma(CFG, "bf16", SEED, exact=True)
    m = hf_model(ref)
    out = []
    with torch.no_grad():
        full = m(input_ids=torch.tensor([TOKENS]), use_cache=False).logits[0].numpy().astype(np.float32)
        for n in PREFIXES:
            lg = m(input_ids=torch.tensor([TOKENS[:n]]), use_cache=False).logits[0, -1].numpy().astype(np.float32)
            # a causal model: the prefix forward and the full forward agree at position n - 1
            assert np.allclose(lg, full[n - 1], rtol=1e-4, atol=1e-4)
            out.append(lg)
    path = os.path.join(HERE, "gemma4_hf_logits.npz")
    np.savez_compressed(path, logits=np.stack(out), tokens=np.array(TOKENS, dtype=np.int32), prefixes=np.array(PREFIXES, dtype=np.int32),
                        seed=np.int64(SEED), cfg_keys=np.array(sorted(CFG)), cfg_vals=np.array([CFG[k] for k in sorted(CFG)], dtype=np.int64),
                        source=np.array("transformers %s Gemma4ForCausalLM, float32, eager attention" % __import__("transformers").__version__))
    # report how the oracle composition compares right now
    worst = 0.0
    for n, e in zip(PREFIXES, out):
        r = RefGemma(CFG, "bf16", SEED, exact=True)
        got = r.forward(TOKENS[:n], 0, 64)
        worst = max(worst, float(np.abs(got - e).max() / np.abs(e).max()))
    print("wrote", path, "| oracle(exact) vs HF worst relative deviation:", worst)


if __name__ == "__main__":
    main()
