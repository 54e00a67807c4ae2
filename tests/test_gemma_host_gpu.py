"""Model-level parity through the C++ host mirror (GemmaTransformer<TWeightQuant> on DeviceType::Rocm):
  * the reference-order path (one launch per component), the fused 6-launch schedule and the captured
    hipGraph give BIT-IDENTICAL logits;
  * prefill(T) followed by decode agrees with decoding the same tokens one by one (prefill GEMM/flash
    path vs decode matvec path; the reference's cross-path bar is 1e-1 * absmax, Linear.Cuda.cpp:760-774);
  * logits agree with the oracle composition (tests/ref_gemma.py) on a small Gemma-shaped config with
    one global layer, for all three weight policies."""
import numpy as np
import pytest

from mila_amd import host
from ref_gemma import RefGemma

pytestmark = pytest.mark.gpu

SMALL = dict(vocab_size=1024, embedding_dim=256, num_layers=6, num_heads=4, num_kv_heads=2, head_dim=64, hidden_dim=512,
             global_head_dim=128, num_global_kv_heads=1, window=8, sliding_window_pattern=6, global_rotary_dim=32)
TOKENS = [5, 900, 17, 3, 512, 77, 1023, 0, 42, 256, 8, 640, 99, 1, 300]
MAX_SEQ = 64


@pytest.mark.parametrize("policy", ["bf16", "fp8", "fp4"])
def test_three_decode_paths_are_bit_identical_and_match_the_oracle(policy):
    ref = RefGemma(SMALL, policy, seed=7)
    models = {m: host.Gemma(policy, SMALL, max_seq=MAX_SEQ, max_prefill=1, seed=7) for m in ("reference", "fused", "graph")}
    mode_of = {}
    worst, errs = 0.0, []
    for pos, tok in enumerate(TOKENS):
        out = {m: g.decode(tok, pos, mode_of.get(m, m)) for m, g in models.items()}
        for m in ("fused", "graph"):
            assert np.array_equal(out["reference"].view(np.uint32), out[m].view(np.uint32)), "%s != reference-order at %d" % (m, pos)
        exp = ref.forward([tok], pos, MAX_SEQ)
        assert np.all(np.isfinite(out["fused"]))
        err = np.abs(out["fused"] - exp).max() / np.abs(exp).max()
        worst = max(worst, err)
        errs.append(err)
    # bf16 intermediates: an occasional 1-ulp flip of an intermediate propagates through 6 random-weight
    # layers (fp4's coarse weights amplify it most); the reference's own BF16 bar is 5e-2 + 5e-2|y| PER OP.
    # Measured on MI355X: worst 1-3e-2 (bf16/fp8), 5e-2 (fp4) of the logit range; median well below.
    assert worst < 1e-1, worst
    assert float(np.median(errs)) < 3e-2, errs
    for g in models.values():
        g.close()


def test_decode_and_prefill_agree_with_the_huggingface_fixture():
    """the device path against the third-party FP32 forward directly (tests/golden/gemma4_hf_logits.npz: transformers
    Gemma4ForCausalLM on the same synthetic weights): bf16 intermediates through 6 random-weight layers stay within 1e-1 of the
    logit range, and the greedy token agrees wherever the FP32 top-2 margin is clear of that noise"""
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gemma4_hf_logits.npz"), allow_pickle=False)
    cfg = {str(k): int(v) for k, v in zip(fx["cfg_keys"], fx["cfg_vals"])}
    seed, tokens = int(fx["seed"]), [int(t) for t in fx["tokens"]]
    by_len = {int(n): e for n, e in zip(fx["prefixes"], fx["logits"])}
    g = host.Gemma("bf16", cfg, max_seq=MAX_SEQ, max_prefill=32, seed=seed)
    errs = []
    for pos, tok in enumerate(tokens):
        out = g.decode(tok, pos, "graph")
        if pos + 1 in by_len:
            exp = by_len[pos + 1]
            errs.append(float(np.abs(out - exp).max() / np.abs(exp).max()))
            srt = np.sort(exp)
            if srt[-1] - srt[-2] > 0.2 * np.abs(exp).max():
                assert int(out.argmax()) == int(exp.argmax())
    assert max(errs) < 1e-1 and float(np.median(errs)) < 5e-2, errs
    for n, exp in by_len.items():
        if n > 1:
            out = g.prefill(tokens[:n])
            assert np.abs(out - exp).max() < 1e-1 * np.abs(exp).max(), n
    g.close()


@pytest.mark.parametrize("policy", ["bf16", "fp4", "fp4-w4a8"])
def test_prefill_then_decode_matches_token_by_token_decode(policy):
    """prefill kernels vs decode kernels on the same tokens.  The fp4 policy has two prefill arithmetics: "fp4" here is the exact-weight toggle (W4A16:
    setFp8ActivationPrefill(false), the reference's kUseFp8ActivationPrefill = false) -- the same function as the decode matvec up to bf16 weight rounding,
    held to the decode path; "fp4-w4a8" is the DEFAULT (W4A8 at every M > 1, round 3: T = 11 used to fall back to W4A16) -- e4m3 activations are a different
    function from the decode matvec by design (reference bar for ONE Linear: 1e-1 * row_absmax, Linear.Cuda.cpp:760-774; through 6 unit-scale random-weight
    layers the difference is of the order of the logits), so it is held to the oracle's W4A8 composition instead"""
    T = 11
    w4a8 = policy == "fp4-w4a8"
    pol = "fp4" if w4a8 else policy
    a = host.Gemma(pol, SMALL, max_seq=MAX_SEQ, max_prefill=16, seed=3)
    b = host.Gemma(pol, SMALL, max_seq=MAX_SEQ, max_prefill=1, seed=3)
    if policy == "fp4":
        a.set_fp8_activation_prefill(False)
    lp = a.prefill(TOKENS[:T])
    for pos in range(T):
        ld = b.decode(TOKENS[pos], pos, "fused")
    bar = 1e-1     # unit-scale random weights: a 1-ulp bf16 difference grows ~1.4x per block (tools/depth_probe.py)
    if not w4a8:
        # fp4: the exact-weight prefill GEMM rounds dequantized weights to bf16 (reference 2-phase semantics), the decode matvec does not
        assert np.abs(lp - ld).max() <= (1e-1 if policy == "fp4" else 5e-2) * np.abs(ld).max()
        # continue decoding on top of the prefilled cache vs on top of the decoded cache
        l1 = a.decode(TOKENS[T], T, "fused")
        l2 = b.decode(TOKENS[T], T, "fused")
        assert np.abs(l1 - l2).max() <= bar * np.abs(l2).max()
    # and the prefill logits agree with the oracle composition of the arithmetic that ran
    ref = RefGemma(SMALL, pol, seed=3, w4a8_prefill=w4a8, staged_prefill=(policy == "fp4"))
    exp = ref.forward(TOKENS[:T], 0, MAX_SEQ)
    if w4a8:
        # unit-scale random weights amplify a difference ~1.4x per block, and under W4A8 a 1-ulp bf16 difference in an activation is a 6 % e4m3 step: two correct
        # implementations sit at 0.1-0.2 of the logit range here (measured 0.14; the conditioned models of test_gemma_conditioned_gpu.py /
        # test_gemma_fullwidth_gpu.py hold the same path to 3e-3 / 1e-3) -- this leg only guards against a wrong ARITHMETIC (W4A16 instead of W4A8 sits at ~0.5)
        bar = 2.5e-1
        assert np.abs(lp - exp).max() <= bar * np.abs(exp).max()
        assert float(np.dot(lp, exp) / (np.linalg.norm(lp) * np.linalg.norm(exp))) > 0.97
        exp1 = ref.forward([TOKENS[T]], T, MAX_SEQ)     # one decode step on the caches the W4A8 prefill wrote, against the oracle continuing its own W4A8 history
        l1 = a.decode(TOKENS[T], T, "fused")
        assert np.abs(l1 - exp1).max() <= bar * np.abs(exp1).max()
    else:
        assert np.abs(lp - exp).max() <= bar * np.abs(exp).max()
    a.close()
    b.close()


def test_host_mirror_error_behaviour():
    with pytest.raises(ValueError):
        host.Gemma("bf16", dict(SMALL, num_heads=3), max_seq=8)          # heads not a multiple of kv heads
    g = host.Gemma("bf16", SMALL, max_seq=8, max_prefill=4, seed=1)
    with pytest.raises(ValueError):
        g.decode(1, 8, "fused")                                           # position beyond the built length
    with pytest.raises(ValueError):
        g.prefill(list(range(5)))                                         # chunk longer than max_prefill
    g.close()


@pytest.mark.parametrize("policy", ["bf16", "fp4"])
def test_greedy_generation_is_identical_across_paths_and_follows_the_logits(policy):
    """closed autoregressive loop (device sampler feeds the next embedding gather): reference-order, fused and
    graph replay produce the same token ids; each id is the argmax of that step's logits"""
    ids = {}
    for mode in ("reference", "fused", "graph"):
        g = host.Gemma(policy, SMALL, max_seq=MAX_SEQ, max_prefill=1, seed=11)
        ids[mode] = g.generate(5, 0, 20, mode)
        g.close()
    assert np.array_equal(ids["reference"], ids["fused"]) and np.array_equal(ids["reference"], ids["graph"])
    g = host.Gemma(policy, SMALL, max_seq=MAX_SEQ, max_prefill=1, seed=11)
    tok = 5
    for pos in range(6):
        nxt = int(np.argmax(g.decode(tok, pos, "fused")))
        assert nxt == int(ids["fused"][pos])
        tok = nxt
    g.close()


MEDIUM = dict(vocab_size=2048, embedding_dim=1280, num_layers=6, num_heads=4, num_kv_heads=2, head_dim=64, hidden_dim=2560,
              global_head_dim=128, num_global_kv_heads=1, window=8, sliding_window_pattern=6, global_rotary_dim=32)


@pytest.mark.parametrize("policy", ["bf16", "fp4"])
def test_fused_prefill_glue_gives_the_same_logits_and_cache_as_one_launch_per_op(policy):
    """prefill with the fused glue (packed-qkv post-processing straight into the cache, one-launch sandwich tails) vs one
    launch per reference op: identical logits, and identical KV caches as seen by the next decode step"""
    toks = [(7 * i + 3) % 2048 for i in range(24)]
    a = host.Gemma(policy, MEDIUM, max_seq=MAX_SEQ, max_prefill=32, seed=5)
    b = host.Gemma(policy, MEDIUM, max_seq=MAX_SEQ, max_prefill=32, seed=5)
    b.set_fused_prefill(False)
    la, lb = a.prefill(toks), b.prefill(toks)
    assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))
    da, db = a.decode(11, len(toks), "fused"), b.decode(11, len(toks), "fused")
    assert np.array_equal(da.view(np.uint32), db.view(np.uint32))
    a.close()
    b.close()


WIDE_FFN = dict(vocab_size=2048, embedding_dim=1280, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64, hidden_dim=6400,
                global_head_dim=128, num_global_kv_heads=1, window=64, sliding_window_pattern=2, global_rotary_dim=32)


@pytest.mark.parametrize("policy", ["fp8", "fp4"])
def test_resident_prefill_weights_give_the_bits_of_per_forward_staging(policy):
    """the quantized prefill with its staging pass kept resident (fp8 -> bf16 / fp4 -> e4m3 once at load, the default) vs re-staged
    into scratch on every forward (the reference's structure): identical logits and identical KV caches, on a shape whose
    fc_gate_up takes the LDS-DMA kernels (T = 1024, F = 6400) while other Linears of the block do not"""
    T = 1024
    toks = [(13 * i + 5) % 2048 for i in range(T)]
    a = host.Gemma(policy, WIDE_FFN, max_seq=T + 8, max_prefill=T, seed=3)
    b = host.Gemma(policy, WIDE_FFN, max_seq=T + 8, max_prefill=T, seed=3)
    b.set_resident_prefill_weights(False)
    la, lb = a.prefill(toks), b.prefill(toks)
    assert np.all(np.isfinite(la)) and np.array_equal(la.view(np.uint32), lb.view(np.uint32))
    da, db = a.decode(11, T, "fused"), b.decode(11, T, "fused")
    assert np.array_equal(da.view(np.uint32), db.view(np.uint32))
    # switching it back on rebuilds the shadows from the quantized weights
    b.set_resident_prefill_weights(True)
    assert np.array_equal(b.prefill(toks).view(np.uint32), la.view(np.uint32))
    a.close()
    b.close()


@pytest.mark.parametrize("policy", ["bf16", "fp8", "fp4"])
def test_safetensors_round_trip_and_quantize_on_load(policy, tmp_path):
    """weight ingestion: a model saved in its storage form reloads into a differently-initialised model
    with identical logits; a BF16 checkpoint loads into a quantized model by quantize-on-load and gives the logits of the model whose
    weights were quantized from the same bf16 values; the files are the public SafeTensors format (the Python package reads them)"""
    st_torch = pytest.importorskip("safetensors.torch")      # torch carries bfloat16 / float8_e4m3fn, numpy does not
    import torch
    toks = [(7 * i + 3) % 2048 for i in range(24)]
    a = host.Gemma(policy, MEDIUM, max_seq=MAX_SEQ, max_prefill=32, seed=5)
    ref_prefill, ref_decode = a.prefill(toks), a.decode(11, len(toks), "fused")
    f = tmp_path / ("gemma_%s.safetensors" % policy)
    a.save_safetensors(f)
    a.close()
    names, meta = host.safetensors_list(f)
    by = {n: (d, b, sh) for n, d, b, sh in names}
    D, F = MEDIUM["embedding_dim"], MEDIUM["hidden_dim"]
    want = {"bf16": ("BF16", 2 * F * D * 2, (2 * F, D)), "fp8": ("F8_E4M3", 2 * F * D, (2 * F, D)), "fp4": ("U8", 2 * F * D // 2, (2 * F, D // 2))}[policy]
    # the reference's flat vocabulary (root name dropped, Core/LanguageModel.ixx:137-146; Gemma.ixx:588-657; Gemma.Block.ixx:546-560)
    assert by["tf_layer_0.fc_gate_up.weight"] == want
    assert ("tf_layer_0.fc_gate_up.weight_scale" in by) == (policy != "bf16")
    assert by["tf_layer_3.layer_scalar"] == ("F32", 4, (1,)) and "tf_layer_0.v_norm.weight" in by and "rmsn_final.weight" in by
    V = MEDIUM["vocab_size"]
    assert by["temb.wte"] == (("BF16", V * D * 2, (V, D)) if policy == "bf16" else ("F8_E4M3", V * D, (V, D)))
    assert ("temb.wte_scale" in by) == (policy != "bf16")
    assert not any(n.startswith("lm_head") for n in by)                          # tied: the head adopts the table (Gemma.ixx:540-555)
    assert meta["mila_quantization"] == {"bf16": "none", "fp8": "per_channel_fp8_e4m3", "fp4": "per_group_fp4_128"}[policy]     # LanguageModelConfig.ixx:104-114
    import json
    mc = json.loads(meta["mila_config"])
    assert mc["architecture"] == "gemma4" and mc["tie_word_embeddings"] is True and mc["num_layers"] == MEDIUM["num_layers"] and mc["hidden_dim"] == F
    pub = st_torch.load_file(str(f))                                            # the public package reads the file
    assert pub["rmsn_final.weight"].dtype == torch.bfloat16 and pub["rmsn_final.weight"].numel() == D
    assert pub["tf_layer_0.fc_gate_up.weight"].dtype == {"bf16": torch.bfloat16, "fp8": torch.float8_e4m3fn, "fp4": torch.uint8}[policy]
    b = host.Gemma(policy, MEDIUM, max_seq=MAX_SEQ, max_prefill=32, seed=99)     # different synthetic weights
    assert not np.array_equal(b.prefill(toks).view(np.uint32), ref_prefill.view(np.uint32))
    b.load_safetensors(f)
    assert np.array_equal(b.prefill(toks).view(np.uint32), ref_prefill.view(np.uint32))
    assert np.array_equal(b.decode(11, len(toks), "fused").view(np.uint32), ref_decode.view(np.uint32))
    # the same tensors through the MILA .bin container (what fromPretrained streams, PretrainedReader.ixx:222-231): written by the model,
    # and converted host-side from the SafeTensors file -- both reload to the same bits; a name prefixed with the root's name resolves too
    fbin, fconv = tmp_path / ("gemma_%s.bin" % policy), tmp_path / ("gemma_%s_conv.bin" % policy)
    b.save_milabin(fbin)
    b.close()
    host.pretrained_to_milabin(f, fconv)
    listed, bmeta = host.pretrained_list(fbin)
    assert bmeta["container"] == "mila" and {n: (d, nb, sh) for n, d, nb, sh in listed} == by
    for src in (fbin, fconv):
        e = host.Gemma(policy, MEDIUM, max_seq=MAX_SEQ, max_prefill=32, seed=77)
        e.load_pretrained(src)
        assert np.array_equal(e.prefill(toks).view(np.uint32), ref_prefill.view(np.uint32)), src
        e.close()
    pref = tmp_path / "prefixed.safetensors"
    st_torch.save_file({"gemma." + k: v for k, v in pub.items()}, str(pref), metadata={"mila_quantization": meta["mila_quantization"]})
    e = host.Gemma(policy, MEDIUM, max_seq=MAX_SEQ, max_prefill=32, seed=77)
    e.load_pretrained(pref)
    assert np.array_equal(e.prefill(toks).view(np.uint32), ref_prefill.view(np.uint32))
    e.close()
    if policy != "bf16":
        # the bf16 checkpoint of the same seed: the quantized model built from it must equal the one quantized at construction
        # (the tied table keeps its own policy: the bf16 model's table is bf16, so it is taken from the quantized file)
        src = host.Gemma("bf16", MEDIUM, max_seq=MAX_SEQ, max_prefill=32, seed=5)
        fb = tmp_path / "gemma_bf16_src.safetensors"
        src.save_safetensors(fb)
        src.close()
        c = host.Gemma(policy, MEDIUM, max_seq=MAX_SEQ, max_prefill=32, seed=99)
        c.load_safetensors(fb)
        assert np.array_equal(c.prefill(toks).view(np.uint32), ref_prefill.view(np.uint32))
        c.close()
    # errors: an unknown tensor, a missing one
    t = dict(pub)
    bad = tmp_path / "bad.safetensors"
    st_torch.save_file(dict(t, **{"tf_layer_0.bogus.weight": torch.zeros(4)}), str(bad))
    d = host.Gemma(policy, MEDIUM, max_seq=MAX_SEQ, max_prefill=32, seed=1)
    with pytest.raises(ValueError, match="unknown tensor"):
        d.load_safetensors(bad)
    st_torch.save_file(dict(t, **{"lm_head.weight": t["temb.wte"].clone()}), str(bad))      # an untied head is not a Gemma-4 artifact
    with pytest.raises(ValueError, match="untied"):
        d.load_safetensors(bad)
    t.pop("tf_layer_1.o_proj.weight")
    t.pop("tf_layer_1.o_proj.weight_scale", None)
    st_torch.save_file(t, str(bad))
    with pytest.raises(ValueError, match="lacks"):
        d.load_safetensors(bad)
    if policy != "bf16":      # a pre-quantized artifact loads only under the policy it was written with (GemmaModel.ixx:617-632)
        other = host.Gemma({"fp8": "fp4", "fp4": "fp8"}[policy], MEDIUM, max_seq=MAX_SEQ, max_prefill=32, seed=1)
        with pytest.raises(RuntimeError, match="pre-quantized"):
            other.load_safetensors(f)
        other.close()
    d.close()


def test_a_reference_layout_checkpoint_loads(tmp_path):
    """a file laid out as the reference's converter / writer lays it out -- bf16 tensors under the reference's names, written here by
    the public safetensors package from host-generated values, tied (no lm_head.weight), no mila metadata at all -- loads into a
    quantize-on-load model and gives the logits of the oracle composition on the same values"""
    st_torch = pytest.importorskip("safetensors.torch")
    import torch
    import orc
    import synth
    cfg = SMALL
    D, H, V = cfg["embedding_dim"], cfg["hidden_dim"], cfg["vocab_size"]
    seed = 7
    ref = RefGemma(cfg, "bf16", seed=seed)

    def bf16(bits):
        return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16)
    t = {"temb.wte": bf16(ref.table[1]), "rmsn_final.weight": bf16(orc.to_bf16_bits(ref.final_norm))}
    for i, L in enumerate(ref.layers):
        n = "tf_layer_%d." % i
        for child, key in (("qkv_proj", "qkv"), ("o_proj", "o"), ("fc_gate_up", "gu"), ("fc_down", "down")):
            t[n + child + ".weight"] = bf16(L[key][1])
        for child, key in (("input_norm", "input_norm"), ("q_norm", "q_norm"), ("k_norm", "k_norm"), ("post_attn_norm", "post_attn"),
                           ("pre_ffn_norm", "pre_ffn"), ("post_ffn_norm", "post_ffn")):
            t[n + child + ".weight"] = bf16(orc.to_bf16_bits(L[key]))
        t[n + "v_norm.weight"] = torch.ones(L["HD"], dtype=torch.bfloat16)
        t[n + "layer_scalar"] = torch.tensor([1.0], dtype=torch.float32)
    f = tmp_path / "converted.safetensors"
    st_torch.save_file(t, str(f))
    g = host.Gemma("bf16", cfg, max_seq=MAX_SEQ, max_prefill=16, seed=12345)      # other weights until the file is loaded
    g.load_pretrained(f)
    for pos, tok in enumerate(TOKENS[:4]):
        out = g.decode(tok, pos, "fused")
        exp = ref.forward([tok], pos, MAX_SEQ)
        assert np.abs(out - exp).max() < 1e-1 * np.abs(exp).max()
    same = host.Gemma("bf16", cfg, max_seq=MAX_SEQ, max_prefill=16, seed=seed)    # the generator's own weights: bit-identical logits
    for pos, tok in enumerate(TOKENS[:4]):
        a = same.decode(tok, pos, "fused")
    assert np.array_equal(a.view(np.uint32), out.view(np.uint32))
    same.close()
    g.close()


def test_graph_replay_survives_scratch_growth_and_follows_the_sampler_setting():
    """(1) fp4 policy: graph decode, then a prefill whose W4A8 activation staging GROWS the context scratch, then graph decode again:
    the captured nodes hold only model-owned pointers, so the replay still equals the fused path.  (2) decode('graph') captures a
    graph WITHOUT the sampler node; a following generate(..., 'graph') must re-capture with it -- and produce what the fused loop does"""
    T = 1024                                  # fc_gate_up (F = 6400) takes the W4A8 LDS-DMA path: per-token e4m3 activations staged in context scratch
    cfg = WIDE_FFN
    a = host.Gemma("fp4", cfg, max_seq=T + 32, max_prefill=T, seed=4)
    b = host.Gemma("fp4", cfg, max_seq=T + 32, max_prefill=T, seed=4)
    l0a, l0b = a.decode(9, 0, "graph"), b.decode(9, 0, "fused")
    assert np.array_equal(l0a.view(np.uint32), l0b.view(np.uint32))
    toks = [(13 * i + 5) % cfg["vocab_size"] for i in range(T)]
    pa, pb = a.prefill(toks), b.prefill(toks)                                      # grows the context scratch far past the attention partials
    assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
    for pos in range(T, T + 4):
        la, lb = a.decode(3, pos, "graph"), b.decode(3, pos, "fused")
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), pos
    a.close()
    b.close()
    # (2) on a model whose greedy continuation is not constant (a 2-layer tied model just echoes its input token)
    for seed in range(11, 19):
        c = host.Gemma("bf16", SMALL, max_seq=MAX_SEQ, max_prefill=1, seed=seed)
        ids_fused = c.generate(5, 0, 12, "fused")
        c.close()
        if len(set(ids_fused.tolist())) > 2:
            break
    else:
        pytest.skip("no seed with a varied greedy continuation")
    d = host.Gemma("bf16", SMALL, max_seq=MAX_SEQ, max_prefill=1, seed=seed)
    first = d.decode(5, 0, "graph")                                                # captures WITHOUT the sampler node
    ids_graph = d.generate(5, 0, 12, "graph")                                      # must re-capture with it
    assert np.array_equal(ids_graph, ids_fused), (ids_graph, ids_fused)
    assert int(ids_graph[0]) == int(np.argmax(first))
    again = d.decode(5, 0, "graph")                                                # plain decode still works on the re-captured graph
    assert np.array_equal(again.view(np.uint32), first.view(np.uint32))
    d.close()



def test_sampled_generation_degenerates_to_greedy_and_is_reproducible():
    """top_k = 1 and temperature <= 0 both reproduce the greedy continuation; a fixed seed reproduces a sampled one; and a
    sampled continuation only ever picks tokens inside the top-k set of the logits it was drawn from"""
    g = host.Gemma("bf16", SMALL, max_seq=MAX_SEQ, max_prefill=1, seed=11)
    greedy = g.generate(5, 0, 10, "fused")
    g.close()
    for kw in (dict(temperature=1.0, top_k=1), dict(temperature=0.0)):
        g = host.Gemma("bf16", SMALL, max_seq=MAX_SEQ, max_prefill=1, seed=11)
        assert np.array_equal(g.generate_sampled(5, 0, 10, seed=3, **kw), greedy), kw
        g.close()
    outs = []
    for _ in range(2):
        g = host.Gemma("bf16", SMALL, max_seq=MAX_SEQ, max_prefill=1, seed=11)
        outs.append(g.generate_sampled(5, 0, 10, temperature=0.9, top_k=8, top_p=0.95, seed=7))
        g.close()
    assert np.array_equal(outs[0], outs[1])
    g = host.Gemma("bf16", SMALL, max_seq=MAX_SEQ, max_prefill=1, seed=11)
    tok = 5
    for pos, nxt in enumerate(outs[0]):
        logits = g.decode(int(tok), pos, "fused")
        assert int(nxt) in np.argsort(-logits)[:8], "position %d: sampled token outside the top-8" % pos
        tok = nxt
    g.close()


@pytest.mark.parametrize("policy", ["bf16", "fp8"])
def test_bounded_ring_kv_policy_matches_the_unbounded_cache(policy):
    """SlidingWindowKvCache (CudaGqaOp.ixx:552-574: the sliding-window layers keep window + chunk - 1 rows, global layers stay
    unbounded) must not change a single logit bit: prefill a chunk, then decode far past the ring's capacity"""
    toks = [(11 * i + 2) % 1024 for i in range(12)]
    a = host.Gemma(policy, SMALL, max_seq=MAX_SEQ, max_prefill=16, seed=9)
    b = host.Gemma(policy, dict(SMALL, bounded_local_kv=1), max_seq=MAX_SEQ, max_prefill=16, seed=9)     # ring of 8 + 16 - 1 = 23 rows
    assert np.array_equal(a.prefill(toks).view(np.uint32), b.prefill(toks).view(np.uint32))
    tok = 3
    for pos in range(len(toks), 60):
        mode = "fused" if pos % 2 else "reference"
        la, lb = a.decode(tok, pos, mode), b.decode(tok, pos, mode)
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), "position %d (%s)" % (pos, mode)
        tok = int(np.argmax(la))
    a.close()
    b.close()


SPLIT = dict(vocab_size=1024, embedding_dim=1280, num_layers=6, num_heads=4, num_kv_heads=2, head_dim=256, hidden_dim=2560,
             global_head_dim=512, num_global_kv_heads=1, window=128, sliding_window_pattern=6, global_rotary_dim=128)


@pytest.mark.parametrize("policy", ["bf16", "fp8", "fp4"])
def test_split_decode_attention_gives_the_reference_order_bits(policy):
    """a configuration whose decode attention really splits (window 128 -> 2 splits, global 256-row cache -> 4): the fused and the graph-replayed
    step (attention + combine launches) equal the one-launch-per-component path bit for bit, at positions around every split boundary.
    (The in-launch alternatives -- combine folded into o_proj, one-pass attention, warm-ahead blocks, side-stream prefetch, the decode chain and the
    engine -- were measured slower, left the host mirror in round 3 and are held at kernel level only: tests/test_fused_gpu.py, test_engine_gpu.py.)"""
    models = {m: host.Gemma(policy, SPLIT, max_seq=256, max_prefill=1, seed=21) for m in ("reference", "fused", "graph")}
    tok = 9
    for pos in range(0, 140, 1):
        check = pos in (0, 1, 63, 64, 65, 127, 128, 139)
        step = {m: g.decode(tok, pos, m if (check or m != "reference") else "fused") for m, g in models.items()}
        if check:
            for m in ("fused", "graph"):
                assert np.array_equal(step["reference"].view(np.uint32), step[m].view(np.uint32)), "%s != reference-order at %d" % (m, pos)
        tok = (tok * 7 + pos) % 1024
    for g in models.values():
        g.close()


@pytest.mark.parametrize("cfg_name", ["SMALL", "MEDIUM"])
def test_chunked_prefill_equals_single_chunk_prefill(cfg_name):
    """a prompt fed as two chunks (the second at position_offset = 10) leaves the same KV caches and produces the same logits as
    one chunk: every row's arithmetic is independent of the chunk it arrives in (reference-order glue for SMALL, fused glue for MEDIUM).
    Bit-for-bit this holds while every chunk stays with one GEMM form -- here at most one tile-row each (the same split of K whatever the row count); a chunk of
    another tile count sums K in another order, and chunkings that cross forms agree to the last bf16 bit only (tests/test_reference_scenarios_gpu.py holds
    the 22.5K-token chunked prefill against the oracle)"""
    cfg = {"SMALL": SMALL, "MEDIUM": MEDIUM}[cfg_name]
    V = cfg["vocab_size"]
    toks = [(17 * i + 4) % V for i in range(16)]
    a = host.Gemma("bf16", cfg, max_seq=MAX_SEQ, max_prefill=32, seed=13)
    b = host.Gemma("bf16", cfg, max_seq=MAX_SEQ, max_prefill=32, seed=13)
    la = a.prefill(toks)
    b.prefill(toks[:10])
    lb = b.prefill(toks[10:], position_offset=10)
    assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), "last-position logits differ"
    da, db = a.decode(3, 16, "fused"), b.decode(3, 16, "fused")
    assert np.array_equal(da.view(np.uint32), db.view(np.uint32)), "the caches differ"
    a.close()
    b.close()


def test_components_stand_alone_as_in_the_reference():
    """Components/{Connections/Residual.ixx:93-127, FFN/Swiglu/Swiglu.ixx:92, Encodings/Rope/Rope.ixx:99-200, Attention/GQA/GroupedQueryAttention.ixx:234-400,
    Embeddings/TokenEmbedding.ixx:155-190,336-384, Linear/Linear.ixx:614-680}: name + config constructor, setExecutionContext, build, forward; results against
    the launchers, lifecycle errors (runtime_error before build, invalid_argument on bad shapes / configs), the tied head aliasing the table; and GemmaBlock<kGlobal>
    as its own type on the reference test's geometry (Tests/Dnn/Components/Transformers/Gemma/Gemma.Block.Cuda.cpp:132-262, 357: construct, build errors, local /
    global geometry with K = V, 16 children, layer_scalar as a parameter) with a prefill + decode through both kinds"""
    host.component_scenarios(0)


def test_the_block_is_the_references_graph():
    """Gemma.Block.ixx:858-921: the children of every block under the reference's names, in its construction order; Gemma.ixx: temb, rmsn_final, lm_head"""
    m = host.Gemma("bf16", SMALL, max_seq=32, max_prefill=8, seed=1)
    names = m.component_names()
    leaves = ["input_norm", "q_norm", "k_norm", "v_norm", "post_attn_norm", "pre_ffn_norm", "post_ffn_norm", "qkv_proj", "rope", "gqa", "o_proj", "res_1",
              "fc_gate_up", "geglu", "fc_down", "res_2"]
    want = []
    for i in range(SMALL["num_layers"]):
        want.append("gemma.tf_layer_%d" % i)
        want += ["gemma.tf_layer_%d.%s" % (i, leaf) for leaf in leaves]
    want += ["gemma.temb", "gemma.rmsn_final", "gemma.lm_head"]
    assert names == want


@pytest.mark.parametrize("policy", ["bf16", "fp8"])
def test_two_stream_half_chunk_prefill_gives_the_same_bits(policy):
    """setPrefillOverlap: the chunk's halves on two streams, the second half's attention behind an event on the first half's K / V rows: same kernels on
    the same rows, so the logits and everything decoded afterwards are bit-identical.  The form steps aside where a GEMM would split K through the context's
    workspace (one call at a time, and a split that depends on the row count) -- this small model's 1280-wide projections do, so the split-K form is
    switched off for the comparison (with it on, both models run the one-stream path)"""
    cfg = dict(vocab_size=2048, embedding_dim=1280, num_layers=6, num_heads=8, num_kv_heads=4, head_dim=64, hidden_dim=2560,
               global_head_dim=128, num_global_kv_heads=1, window=256, sliding_window_pattern=3, global_rotary_dim=32)
    T = 1024
    toks = (np.arange(T, dtype=np.int64) * 7919 % 2048).astype(np.int32)
    from mila_amd import capi
    lib = capi.load()
    assert lib.mila_cdna4_gemm_workspace_bytes(512, 1280, 1280) > 0
    a = host.Gemma(policy, cfg, max_seq=T + 8, max_prefill=T, seed=3)
    b = host.Gemma(policy, cfg, max_seq=T + 8, max_prefill=T, seed=3)
    capi.tune("gemm.splitk", 0)
    try:
        assert lib.mila_cdna4_gemm_workspace_bytes(512, 1280, 1280) == 0
        b.set_prefill_overlap(True)
        la, lb = a.prefill(toks), b.prefill(toks)
        assert np.array_equal(la, lb)
        ta, tb = int(np.argmax(la)), int(np.argmax(lb))
        for pos in range(T, T + 4):                       # the caches the two forms left behind are the same
            da, db = a.decode(ta, pos, "fused"), b.decode(tb, pos, "fused")
            assert np.array_equal(da, db)
            ta, tb = int(np.argmax(da)), int(np.argmax(db))
    finally:
        capi.tune_reset()
        a.close()
        b.close()


@pytest.mark.parametrize("policy", ["bf16", "fp4"])
def test_prefill_from_after_rewind_matches_full_prefill__Gemma_Cuda_cpp_338(policy):
    """Tests/Dnn/Components/Transformers/Gemma/Gemma.Cuda.cpp:338-371 (kSeq 24, kSplit 16: rewind to the split, prefill only the tail; tolerance 1e-4 + 1e-3 |e|),
    :373-383 (a rewind beyond the fill is refused), :385-391 (prefillFrom rejects an offset outside the prompt).  Here rows are independent of the chunking: exact bits."""
    seq, split = 24, 16
    toks = ((np.arange(seq, dtype=np.int64) * 5 + 3) % SMALL["vocab_size"]).astype(np.int32)
    m = host.Gemma(policy, SMALL, max_seq=32, max_prefill=seq, seed=4)
    try:
        full = m.prefill(toks)
        assert m.rewind(split)
        inc = m.prefill_from(toks, split)
        assert np.all(np.abs(inc - full) <= 1e-4 + 1e-3 * np.abs(full))
        assert np.array_equal(inc, full)
        m.prefill(toks)
        assert m.rewind(seq - 1) and not m.rewind(seq + 10)
        for bad in (seq, -1):
            with pytest.raises(ValueError, match="outside the prompt"):
                m.prefill_from(toks, bad)
    finally:
        m.close()


def test_a_long_capacity_model_takes_the_matrix_core_decode_on_every_path():
    """a model whose global layers put 16 heads of size 512 on one KV head, built for a context of 4200: the global layers decode on the matrix cores
    (attn_decode_mfma_kernel; chosen by (window, capacity), csrc/attention.hip) -- in the reference-order path, the fused schedule and the graph replay alike:
    bit-identical logits across the three, oracle distance as on the other small models; a prefill in front, so that the decode reads a prefilled cache"""
    cfg = dict(vocab_size=1024, embedding_dim=256, num_layers=4, num_heads=16, num_kv_heads=8, head_dim=64, hidden_dim=512,
               global_head_dim=512, num_global_kv_heads=1, window=8, sliding_window_pattern=2, global_rotary_dim=128)
    max_seq = 4200
    from mila_amd import capi
    ref = RefGemma(cfg, "bf16", seed=11, staged_prefill=True)
    ref.forward(TOKENS[:9], 0, max_seq)
    models = {m: host.Gemma("bf16", cfg, max_seq=max_seq, max_prefill=16, seed=11) for m in ("reference", "fused", "graph")}
    for g in models.values():
        g.prefill(TOKENS[:9])
    worst = 0.0
    try:
        capi.tune("attn.mfma_min_band", 4096)          # (round 4: the default starts the matrix-core decode at the 8192-key bucket; here from the first bucket on)
        for i, tok in enumerate(TOKENS[9:]):
            pos = 9 + i
            capi.last_form()
            out = {m: g.decode(tok, pos, m) for m, g in models.items()}
            assert "attn_decode_mfma" in capi.last_form()
            for m in ("fused", "graph"):
                assert np.array_equal(out["reference"].view(np.uint32), out[m].view(np.uint32)), "%s != reference-order at %d" % (m, pos)
            exp = ref.forward([tok], pos, max_seq)
            worst = max(worst, float(np.abs(out["graph"] - exp).max() / np.abs(exp).max()))
    finally:
        capi.tune_reset()
    assert worst < 1e-1, worst
    for g in models.values():
        g.close()


def test_decode_across_a_live_length_bucket_re_captures_the_graph_and_keeps_the_three_paths_identical():
    """an unwindowed layer's decode geometry -- split count, wave-per-position vs matrix-core kernel -- follows the live-length bucket (4096, 8192, ...
    keys), not the cache capacity: a model built for 8300 positions decodes positions 4090 .. 4100 across the first bucket boundary; the captured graph is re-captured
    there (GemmaTransformer::ensureGraph) and reference-order, fused and graph-replay logits stay bit-identical on both sides; the global layers change kernels at 4096"""
    from mila_amd import capi
    cfg = dict(vocab_size=1024, embedding_dim=256, num_layers=2, num_heads=16, num_kv_heads=8, head_dim=64, hidden_dim=512,
               global_head_dim=512, num_global_kv_heads=1, window=8, sliding_window_pattern=2, global_rotary_dim=128)
    max_seq, start = 8300, 4090
    toks = [(7 * i + 3) % 1024 for i in range(start + 12)]
    models = {m: host.Gemma("bf16", cfg, max_seq=max_seq, max_prefill=1024, seed=11) for m in ("reference", "fused", "graph")}
    for g in models.values():
        for o in range(0, start, 1024):
            g.prefill(toks[o:min(o + 1024, start)], o)
    seen = set()
    for pos in range(start, start + 12):
        capi.last_form()
        out = {m: g.decode(toks[pos], pos, m) for m, g in models.items()}
        forms = capi.last_form()
        seen.add(("attn_decode_mfma" in forms, pos + 1 > 4096))
        for m in ("fused", "graph"):
            assert np.array_equal(out["reference"].view(np.uint32), out[m].view(np.uint32)), "%s != reference-order at %d" % (m, pos)
    assert seen == {(False, False), (True, True)}, seen                 # the matrix-core decode exactly from the 8192-key bucket on
    for g in models.values():
        g.close()


@pytest.mark.parametrize("cfg_name,T", [("MEDIUM", 24), ("WIDE_FFN", 1024)])
def test_w8a8_opt_in_prefill_of_the_fp8_policy(cfg_name, T):
    """PerChannelFp8<> with setFp8ActivationPrefill(true) (row g1; Policies.ixx:39-40): every layer Linear of a T > 1 forward runs fp8 x fp8 on the policy's own e4m3
    weights -- the fused-glue prefill (tails hand over per-token e4m3 rows, fused Linear + GeGLU) and the one-launch-per-op prefill give identical bits; the default stays
    W8A16 (another function: the logits differ, and switching back restores the default's bits); decode is untouched by the switch"""
    cfg = {"MEDIUM": MEDIUM, "WIDE_FFN": WIDE_FFN}[cfg_name]
    toks = [(13 * i + 5) % 2048 for i in range(T)]
    a = host.Gemma("fp8", cfg, max_seq=T + 8, max_prefill=T, seed=3)
    b = host.Gemma("fp8", cfg, max_seq=T + 8, max_prefill=T, seed=3)
    default = a.prefill(toks)
    d_default = a.decode(11, T, "fused")
    a.set_fp8_activation_prefill(True)
    b.set_fp8_activation_prefill(True)
    b.set_fused_prefill(False)
    la, lb = a.prefill(toks), b.prefill(toks)
    assert np.all(np.isfinite(la)) and np.array_equal(la.view(np.uint32), lb.view(np.uint32))
    assert not np.array_equal(la.view(np.uint32), default.view(np.uint32)), "the switch changed nothing"
    da, db = a.decode(11, T, "fused"), b.decode(11, T, "fused")
    assert np.array_equal(da.view(np.uint32), db.view(np.uint32))
    # the same function up to the activation quantization: the reference's own bar for an activation-quantized prefill is 1e-1 of the row's range for ONE Linear
    # (Linear.Cuda.cpp:760-774); through these random-weight layers the logits stay well correlated
    assert float(np.dot(la, default) / (np.linalg.norm(la) * np.linalg.norm(default))) > 0.9
    # off again: the resident bf16 staging is rebuilt and the default's bits return
    a.set_fp8_activation_prefill(False)
    assert np.array_equal(a.prefill(toks).view(np.uint32), default.view(np.uint32))
    assert np.array_equal(a.decode(11, T, "fused").view(np.uint32), d_default.view(np.uint32))
    a.close()
    b.close()
