"""Generates tests/golden/gpt2_hf_logits.npz: FP32 logits of the transformers GPT2LMHeadModel on a small GPT-2-shaped
configuration with synthetic weights, to pin the NET-LEVEL wiring of the oracle's restated reference CPU backend
(oracle/mila_oracle.c: orc_cpu_gpt2_forward = GptTransformer::forward, Components/Transformers/Gpt/GptTransformer.ixx:221-254).
The reference's own GPT tests assert shapes and finiteness only (Tests/Dnn/Components/Transformers/Gpt/GptTransformer.Cpu.cpp:226-255),
and its checkpoints are converted from the same HF model family, so this is the independent implementation to agree with.

    python tests/golden/make_gpt2_hf_golden.py       (build container: transformers + torch CPU, model built from a config object)

The fixture holds inputs and expected outputs only; parameters are regenerated from the seeds by gpt2_params() below
(tests/synth.py, counter-based)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import orc  # noqa: E402
import synth  # noqa: E402

V, MAXT, C_, L, NH, B, T = 256, 32, 64, 3, 4, 2, 24
SEED = 11


def gpt2_params(seed=SEED, V=V, maxT=MAXT, C_=C_, L=L):
    """the parameter list of orc.cpu_gpt2_forward (fp32 values that are bf16-representable)"""
    k = [seed * 1000]

    def t(n, amp, offset=0.0):
        k[0] += 1
        return orc.from_bf16_bits(synth.fill_bf16(k[0], n, amp, offset))
    ps = [t(V * C_, 0.1).reshape(V, C_), t(maxT * C_, 0.05).reshape(maxT, C_)]
    for _ in range(L):
        ps += [t(C_, 0.2, 1.0), t(C_, 0.1), t(3 * C_ * C_, 1.5 * C_ ** -0.5).reshape(3 * C_, C_), t(3 * C_, 0.05),
               t(C_ * C_, 1.5 * C_ ** -0.5).reshape(C_, C_), t(C_, 0.05), t(C_, 0.2, 1.0), t(C_, 0.1),
               t(4 * C_ * C_, 1.5 * C_ ** -0.5).reshape(4 * C_, C_), t(4 * C_, 0.05),
               t(4 * C_ * C_, 1.5 * (4 * C_) ** -0.5).reshape(C_, 4 * C_), t(C_, 0.05)]
    ps += [t(C_, 0.2, 1.0), t(C_, 0.1), t(V * C_, 1.5 * C_ ** -0.5).reshape(V, C_)]
    return ps


def gpt2_tokens():
    return (synth.uniform(SEED + 5, B * T) * V).astype(np.int32).reshape(B, T) % V


def main():
    import torch
    from transformers import GPT2Config, GPT2LMHeadModel
    cfg = GPT2Config(vocab_size=V, n_positions=MAXT, n_embd=C_, n_layer=L, n_head=NH, activation_function="gelu_new", layer_norm_epsilon=1e-5,
                     resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    m = GPT2LMHeadModel(cfg).float().eval()
    ps = gpt2_params()
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    sd = {"transformer.wte.weight": tt(ps[0]), "transformer.wpe.weight": tt(ps[1])}
    for l in range(L):
        p = ps[2 + 12 * l: 14 + 12 * l]
        pre = "transformer.h.%d." % l
        sd.update({pre + "ln_1.weight": tt(p[0]), pre + "ln_1.bias": tt(p[1]),
                   pre + "attn.c_attn.weight": tt(p[2].T), pre + "attn.c_attn.bias": tt(p[3]),          # HF Conv1D keeps [in, out]
                   pre + "attn.c_proj.weight": tt(p[4].T), pre + "attn.c_proj.bias": tt(p[5]),
                   pre + "ln_2.weight": tt(p[6]), pre + "ln_2.bias": tt(p[7]),
                   pre + "mlp.c_fc.weight": tt(p[8].T), pre + "mlp.c_fc.bias": tt(p[9]),
                   pre + "mlp.c_proj.weight": tt(p[10].T), pre + "mlp.c_proj.bias": tt(p[11])})
    sd.update({"transformer.ln_f.weight": tt(ps[-3]), "transformer.ln_f.bias": tt(ps[-2]), "lm_head.weight": tt(ps[-1])})
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.endswith(".attn.bias") and not k.endswith("masked_bias")]
    assert not missing and not unexpected, (missing, unexpected)
    tokens = gpt2_tokens()
    with torch.no_grad():
        logits = m(input_ids=torch.from_numpy(tokens.astype(np.int64))).logits.numpy().astype(np.float32)
    path = os.path.join(HERE, "gpt2_hf_logits.npz")
    np.savez_compressed(path, logits=logits, tokens=tokens, dims=np.array([V, MAXT, C_, L, NH, B, T], dtype=np.int64), seed=np.int64(SEED),
                        source=np.array("transformers %s GPT2LMHeadModel, float32, eager attention, untied head" % __import__("transformers").__version__))
    got = orc.cpu_gpt2_forward(tokens, ps, C_, L, NH, V, MAXT)
    print("wrote", path, "| oracle vs HF worst relative deviation:", float(np.abs(got - logits).max() / np.abs(logits).max()))


if __name__ == "__main__":
    main()
