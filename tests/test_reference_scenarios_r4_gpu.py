"""More of the reference's OWN test scenarios, run one-to-one against the HIP path / the host mirror (file:line in the test ids); round 4's additions to
tests/test_reference_scenarios_gpu.py: Linear's tying and pre-quantized-artifact contracts, the block's layer scalar through a checkpoint, the KV-policy routing,
the sampler's pipeline-vs-reference, softcap, boundary, determinism and enqueue / await scenarios."""
import numpy as np
import pytest

import orc
from mila_amd import host

pytestmark = pytest.mark.gpu

K_IN, K_OUT = 256, 128      # Linear.Cuda.cpp: kInFeatures, kOutFeatures


def weight_value(o, i):
    """Linear.Cuda.cpp:70-75 weightValue"""
    return np.float32(0.1) * (np.float32((o * 13 + i * 7) % 17) - np.float32(8.0)) / np.float32(17.0)


def closed_form_weight(N, K):
    o = np.arange(N)[:, None]
    i = np.arange(K)[None, :]
    return orc.to_bf16_bits((np.float32(0.1) * (((o * 13 + i * 7) % 17).astype(np.float32) - np.float32(8.0)) / np.float32(17.0)).astype(np.float32))


# ---- Linear.Cuda.cpp:618-641: the tying contract on quantized instantiations ----
def test_install_shared_weight_per_group_path_throws__Linear_Cuda_cpp_618():
    assert host.linear_install_shared_probe("fp4", 0) == "logic_error"
    assert host.linear_install_shared_probe("fp4", 1) == "logic_error"


def test_install_shared_weight_per_channel_without_scales_throws__Linear_Cuda_cpp_631():
    assert host.linear_install_shared_probe("fp8", 0) == "logic_error"
    # (the accepted overloads reject the null handle itself: std::invalid_argument, before any device work)
    assert host.linear_install_shared_probe("fp8", 1) == "invalid_argument"
    assert host.linear_install_shared_probe("bf16", 0) == "invalid_argument"
    assert host.linear_install_shared_probe("bf16", 1) == "logic_error"


# ---- Linear.Cuda.cpp:1145-1290: a pre-quantized artifact loads back without requantizing ----
@pytest.mark.parametrize("policy", ["fp8", "fp4"])
def test_pre_quantized_artifact_loads_back_without_requantizing__Linear_Cuda_cpp_1145(policy):
    Wb = closed_form_weight(K_OUT, K_IN)
    a = host.LinearComponent(policy, K_IN, K_OUT, 1)
    a.load("weight", Wb)                                     # first leg: quantize on load from BF16, then export
    w1, s1 = a.read()
    exp_w, exp_s = (orc.quantize_fp8_per_channel(Wb) if policy == "fp8" else orc.quantize_fp4_per_group(Wb, 128))
    assert np.array_equal(w1, exp_w.reshape(-1)) and np.array_equal(s1, exp_s.reshape(-1))
    assert np.all(np.isfinite(s1)) and np.all(s1 > 0)
    b = host.LinearComponent(policy, K_IN, K_OUT, 1)       # second leg: the packed bytes and scales into a fresh component -- the blob SIZE says "do not quantize again"
    b.load("weight", w1)
    b.load("weight_scale", s1)
    w2, s2 = b.read()
    assert np.array_equal(w1, w2) and np.array_equal(s1, s2)                     # byte for byte
    x = orc.to_bf16_bits(np.array([np.float32(0.25) * weight_value(i % K_OUT, i) for i in range(K_IN)], dtype=np.float32))
    ya, yb = a.forward(x), b.forward(x)
    assert np.array_equal(ya, yb) and np.all(np.isfinite(orc.from_bf16_bits(ya)))
    a.close()
    b.close()


# ---- Linear.Cuda.cpp:1296-1425: the reloaded fp4 component COMPUTES the same thing at a prefill shape (the per-tensor e4m3 scale is derived on a pre-quantized load too) ----
def test_pre_quantized_fp4_reload_forward_matches_quantize_on_load__Linear_Cuda_cpp_1304():
    rows, k_in, k_out = 16, 512, 256
    Wb = closed_form_weight(k_out, k_in)
    a = host.LinearComponent("fp4", k_in, k_out, rows)
    a.load("weight", Wb)
    w, s = a.read()
    b = host.LinearComponent("fp4", k_in, k_out, rows)
    b.load("weight", w)
    b.load("weight_scale", s)
    x = orc.to_bf16_bits((np.arange(rows * k_in, dtype=np.float32) / np.float32(rows * k_in) * 2 - 1).astype(np.float32))      # spreadHost
    expected, actual = a.forward(x), b.forward(x)
    assert not np.any(np.isnan(orc.from_bf16_bits(actual))), "pre-quantized FP4 reload produced NaN activations"
    assert np.array_equal(expected, actual)
    a.close()
    b.close()


def test_rejects_scales_on_an_unquantized_build__Linear_Cuda_cpp_1427():
    lin = host.LinearComponent("bf16", K_IN, K_OUT, 1)
    with pytest.raises(ValueError):                         # std::invalid_argument: a quantized artifact reaching an unquantized build fails loudly
        lin.load("weight_scale", np.ones(K_OUT, dtype=np.float32))
    lin.close()


def test_unquantized_path_holds_weight_and_bias_and_no_scales__Linear_Cuda_cpp_1458():
    lin = host.LinearComponent("bf16", K_IN, K_OUT, 1, bias=True)
    Wb = closed_form_weight(K_OUT, K_IN)
    lin.load("weight", Wb)
    lin.load("bias", orc.to_bf16_bits(np.linspace(-1, 1, K_OUT, dtype=np.float32)))
    w, s = lin.read()
    assert s is None and np.array_equal(w.view(np.uint16), Wb.reshape(-1))       # the weight as stored, and nothing like a scales tensor
    lin.close()


# ---- Gemma.Block.Cuda.cpp:323-395: the block's own parameter through a checkpoint, children under their own scopes ----
SMALLG = dict(vocab_size=512, embedding_dim=256, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64, hidden_dim=512, global_head_dim=128, num_global_kv_heads=1, window=8,
              sliding_window_pattern=2, global_rotary_dim=32)


def test_save_writes_layer_scalar_and_scopes_every_child__Gemma_Block_Cuda_cpp_323(tmp_path):
    st = pytest.importorskip("safetensors.numpy")
    prof = dict(linear_gain=1.0, qk_norm_center=1.0, post_norm_center=1.0, layer_scalar=2.5, table_gain=1.0)
    g = host.Gemma("bf16", SMALLG, max_seq=32, max_prefill=8, seed=3, profile=prof)
    f = tmp_path / "scalar.safetensors"
    g.save_safetensors(f)
    g.close()
    names, _ = host.safetensors_list(f)
    by = {n: (d, b, sh) for n, d, b, sh in names}
    assert by["tf_layer_0.layer_scalar"] == ("F32", 4, (1,))                     # the block's own parameter
    import struct
    raw = open(f, "rb").read()
    hlen = struct.unpack("<Q", raw[:8])[0]
    import json
    hdr = json.loads(raw[8:8 + hlen])
    o0, o1 = hdr["tf_layer_0.layer_scalar"]["data_offsets"]
    assert struct.unpack("<f", raw[8 + hlen + o0:8 + hlen + o1])[0] == 2.5
    scoped = [n for n in by if n.startswith("tf_layer_0.") and n != "tf_layer_0.layer_scalar"]
    assert len(scoped) > 5 and len(set(scoped)) == len(scoped), "children collapsed onto a single scope"
    assert {"tf_layer_0.input_norm.weight", "tf_layer_0.qkv_proj.weight", "tf_layer_0.fc_down.weight", "tf_layer_0.post_ffn_norm.weight"} <= set(scoped)


def test_save_then_load_restores_layer_scalar__Gemma_Block_Cuda_cpp_357(tmp_path):
    prof = dict(linear_gain=1.0, qk_norm_center=1.0, post_norm_center=1.0, layer_scalar=2.5, table_gain=1.0)
    src = host.Gemma("bf16", SMALLG, max_seq=32, max_prefill=8, seed=3, profile=prof)
    f = tmp_path / "src.safetensors"
    src.save_safetensors(f)
    want = src.prefill([1, 2, 3, 4])
    src.close()
    dst = host.Gemma("bf16", SMALLG, max_seq=32, max_prefill=8, seed=3)          # a fresh model: layer_scalar defaults to 1.0
    before = dst.prefill([1, 2, 3, 4])
    assert not np.array_equal(before.view(np.uint32), want.view(np.uint32)), "the default scalar is indistinguishable from the saved one"
    dst.load_safetensors(f)
    assert np.array_equal(dst.prefill([1, 2, 3, 4]).view(np.uint32), want.view(np.uint32))
    f2 = tmp_path / "dst.safetensors"
    dst.save_safetensors(f2)                                                   # re-save the target and inspect: 2.5 means restored, 1.0 means load did nothing
    dst.close()
    import json
    import struct
    raw = open(f2, "rb").read()
    hlen = struct.unpack("<Q", raw[:8])[0]
    o0, o1 = json.loads(raw[8:8 + hlen])["tf_layer_1.layer_scalar"]["data_offsets"]
    assert struct.unpack("<f", raw[8 + hlen + o0:8 + hlen + o1])[0] == 2.5


# ---- Gemma.Cuda.cpp:456-480: the sliding-window KV policy reaches the LOCAL layers only (the compile-time half is a static_assert in host/src/gemma_runner.cpp) ----
def test_kv_policy_routes_the_bounded_ring_to_local_layers_only__Gemma_Cuda_cpp_456():
    cfg = dict(SMALLG, num_layers=4, bounded_local_kv=1)
    max_seq, P = 200, 16
    b = host.Gemma("bf16", cfg, max_seq=max_seq, max_prefill=P, seed=3)
    u = host.Gemma("bf16", dict(cfg, bounded_local_kv=0), max_seq=max_seq, max_prefill=P, seed=3)
    cap = min(max_seq, cfg["window"] + P - 1)                                  # CudaGqaOp.ixx:552-574
    local_saving = 2 * 2 * cfg["num_kv_heads"] * (max_seq - cap) * cfg["head_dim"] * 2          # two local layers, K and V
    assert u.memory_stats()["actual"]["device_state_bytes"] - b.memory_stats()["actual"]["device_state_bytes"] == local_saving      # nothing saved on the two global layers
    toks = [(7 * i + 3) % cfg["vocab_size"] for i in range(40)]
    outs = []
    for m in (b, u):                                                           # and the bounded model computes what the unbounded one computes (chunked prefill through the ring)
        m.prefill(toks[:16])
        m.prefill(toks[16:32], 16)
        outs.append(m.decode(toks[32], 32, "fused"))
        m.close()
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


# ---- Sampling.Cuda.cpp:264-300, :365-386, :388-424, :427-510 ----
GEMMA_VOCAB = 262144


def _reference_token(logits, softcap, t, k, p, r):
    tok, margins = orc.sample_stochastic(np.asarray(logits, dtype=np.float32), softcap, t, k, p, r)
    return tok, margins


def test_pipeline_matches_reference_truncated_at_gemma_vocab__Sampling_Cuda_cpp_275():
    """the truncated filter matrix x r grid of the reference test, temperature 0.8, continuous N(0, 4) logits at the Gemma vocabulary; the reference side is its retained
    single-block kernel's semantics restated (orc_sample_stochastic).  A draw whose CDF bracket sits within float rounding of the target is the one case two correct
    implementations may differ on (the reference's own comment on summation order): skipped by its margin, and at least 15 of the 21 draws must be decisive."""
    rng = np.random.default_rng(42)
    logits = (rng.standard_normal(GEMMA_VOCAB) * 4.0).astype(np.float32)
    s = host.Sampler(GEMMA_VOCAB, 0.0)
    s.set_logits(logits)
    decisive = 0
    for top_k, top_p in ((64, 1.0), (0, 0.9), (64, 0.9)):
        for r in (0.0, 0.1, 0.37, 0.5, 0.73, 0.9, 0.999):
            tok, m = _reference_token(logits, 0.0, 0.8, top_k, top_p, r)
            got = s.sample(0.8, top_k, top_p, r)
            if m[0] > 1e-6 and m[2] > 1e-3:
                assert got == tok, "top_k=%d top_p=%g r=%g" % (top_k, top_p, r)
                decisive += 1
            assert s.sample_enqueued(0.8, top_k, top_p, r) == got                # :440-462 Enqueued_MatchesForward_Truncated_AtGemmaVocab: same kernels, same r
    assert decisive >= 15
    s.close()


def test_pipeline_matches_reference_with_softcap__Sampling_Cuda_cpp_365():
    rng = np.random.default_rng(7)
    logits = (rng.standard_normal(GEMMA_VOCAB) * 4.0).astype(np.float32)
    s = host.Sampler(GEMMA_VOCAB, 30.0)
    s.set_logits(logits)
    for r in (0.1, 0.5, 0.9):
        tok, m = _reference_token(logits, 30.0, 0.7, 64, 0.95, r)
        if m[0] > 1e-6 and m[2] > 1e-3:
            assert s.sample(0.7, 64, 0.95, r) == tok, "softcap parity at r=%g" % r
    s.close()


def test_pipeline_boundary_r_and_determinism_at_gemma_vocab__Sampling_Cuda_cpp_388_405():
    s = host.Sampler(GEMMA_VOCAB, 0.0)
    s.set_logits(np.zeros(GEMMA_VOCAB, dtype=np.float32))                       # uniform logits: the CDF is exact in FP32
    assert s.sample(1.0, 0, 1.0, 0.0) == 0
    assert s.sample(1.0, 0, 1.0, 0.999999) == GEMMA_VOCAB - 1
    rng = np.random.default_rng(1234)
    s.set_logits((rng.standard_normal(GEMMA_VOCAB) * 4.0).astype(np.float32))
    assert s.sample(0.8, 64, 1.0, 0.42) == s.sample(0.8, 64, 1.0, 0.42)         # integer-count top-k refinement: run-deterministic
    s.close()


def test_enqueued_path_contract__Sampling_Cuda_cpp_427_510():
    s = host.Sampler(8, 0.0)
    with pytest.raises(TypeError):                                              # AwaitToken_WithoutEnqueue_Throws: std::logic_error, not UB
        s.await_token()
    s.set_logits([3, 9, 1, 7, 5, 2, 8, 4])
    assert s.sample_enqueued(0.0) == 1 and s.sample_enqueued(0.0) == s.sample(0.0)            # Enqueued_MatchesForward_Greedy
    low, high = [9, 1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1, 9]
    for lg, want in ((low, 0), (high, 7), (low, 0)):                            # Enqueued_BackToBack_SingleSlotReuse: each cycle returns its own token
        s.set_logits(lg)
        assert s.sample_enqueued(0.0) == want
    s.set_logits([50.0 if i == 5 else 1.0 for i in range(8)])                   # Enqueued_OrderedAfterPriorStreamWork: the copy and the sampler on one stream, no host sync between
    assert s.sample_enqueued(0.0) == 5
    with pytest.raises(TypeError):
        s.await_token()
    s.close()
