"""L6 caller (GemmaModel: fromPretrained + generate) on the GPU: Models/GemmaModel.ixx:137-170, 439-568, 604-665, 750-770."""
import numpy as np
import pytest

from mila_amd import host

pytestmark = pytest.mark.gpu

SMALL = dict(vocab_size=1024, embedding_dim=256, num_layers=6, num_heads=4, num_kv_heads=2, head_dim=64, hidden_dim=512,
             global_head_dim=128, num_global_kv_heads=1, window=16, sliding_window_pattern=3, global_rotary_dim=32)
PROMPT = [5, 17, 900, 3, 44, 260, 7, 7, 31, 512, 99, 2]


def manual_greedy(policy, prompt, n, context=64, seed=21):
    """prefill, then one reference-order decode step per token: what generate() must reproduce"""
    m = host.Gemma(policy, SMALL, max_seq=context, max_prefill=16, seed=seed)
    try:
        logits = m.prefill(np.asarray(prompt, dtype=np.int32))
        out, pos = [], len(prompt)
        tok = int(np.argmax(logits))
        while len(out) < n:
            out.append(tok)
            if pos >= context:
                break
            tok = int(np.argmax(m.decode(tok, pos, "reference")))
            pos += 1
        return out
    finally:
        m.close()


@pytest.mark.parametrize("policy", ["bf16", "fp8", "fp4"])
def test_greedy_generate_is_prefill_plus_decode_and_reports_the_budget(policy):
    """GemmaModel.ixx:439-568 (decode-ahead loop; greedy = the captured graph with the device sampler as its last node)"""
    want = manual_greedy(policy, PROMPT, 12)
    g = host.GemmaModel.synthetic(policy, SMALL, context=64, prefill_chunk=16, seed=21)
    try:
        got, why, reused = g.generate(PROMPT, max_new_tokens=12, stop_tokens=[1023 if 1023 not in want else 1022])
        assert got == want and why == "length" and reused == 0
        # a chunk smaller than the prompt: chunked prefill (Gemma.ixx:234-267) gives the same continuation
        g2 = host.GemmaModel.synthetic(policy, SMALL, context=64, prefill_chunk=5, seed=21)
        got2, _, _ = g2.generate(PROMPT, max_new_tokens=12, stop_tokens=[1023 if 1023 not in want else 1022])
        g2.close()
        assert got2 == want
    finally:
        g.close()


def test_stop_token_context_bound_and_budget_of_one():
    """the four outcomes of GenerateStatus (Core/GenerateStatus.ixx): a stop token is not passed to on_token and ends with 'stop'; a full context with
    'context_limit'; a spent budget with 'length' (budget takes precedence: GemmaModel.ixx:556-562)"""
    want = manual_greedy("bf16", PROMPT, 30, context=32)
    g = host.GemmaModel.synthetic("bf16", SMALL, context=32, prefill_chunk=16, seed=21)
    try:
        stop = want[3]
        k = want.index(stop)
        got, why, _ = g.generate(PROMPT, stop_tokens=[stop])
        assert got == want[:k] and why == "stop"
        unused = next(t for t in range(1024) if t not in want)
        got, why, _ = g.generate(PROMPT, stop_tokens=[unused])            # no budget: runs to the context bound
        assert why == "context_limit" and got == want[:32 - len(PROMPT) + 1]
        got, why, _ = g.generate(PROMPT, max_new_tokens=1, stop_tokens=[unused])
        assert got == want[:1] and why == "length"
        got, why, _ = g.generate(PROMPT, stop_tokens=[unused], cancel_after=3)      # the client's stop request between two steps: nothing runs past the return
        assert why == "cancelled" and got == want[:3]
        got, why, _ = g.generate(PROMPT, max_new_tokens=5, stop_tokens=[unused])     # and the model is usable afterwards (one ahead-decoded token was drained)
        assert got == want[:5] and why == "length"
        full = (PROMPT * 3)[:32]                                                       # a prompt that fills the context: one token, no decode step
        got, why, _ = g.generate(full, stop_tokens=[unused])
        assert len(got) == 1 and why == "context_limit"
        with pytest.raises(ValueError, match="empty prompt"):
            g.generate([], max_new_tokens=1)
        with pytest.raises(ValueError, match="exceeds deployment context length"):
            g.generate(list(range(40)), max_new_tokens=1)
        with pytest.raises(ValueError, match="outside the vocabulary"):
            g.generate([5, 4000], max_new_tokens=1)
    finally:
        g.close()


@pytest.mark.parametrize("bounded", [False, True])
def test_kv_prefix_reuse_never_changes_the_tokens(bounded):
    """GemmaModel.ixx:466-494: a prompt that extends what the caches hold is prefilled from the common prefix on; a bounded ring that has evicted what
    a rewind needs refuses, and the full prefill gives the same tokens"""
    cfg = dict(SMALL, bounded_local_kv=int(bounded))
    stop = [1023]
    a = host.GemmaModel.synthetic("bf16", cfg, context=64, prefill_chunk=8, seed=5)
    b = host.GemmaModel.synthetic("bf16", cfg, context=64, prefill_chunk=8, seed=5)
    try:
        first, _, r0 = a.generate(PROMPT, max_new_tokens=6, stop_tokens=stop)
        assert r0 == 0
        turn2 = PROMPT + first + [77, 78, 79]
        got, _, reused = a.generate(turn2, max_new_tokens=8, stop_tokens=stop)
        fresh, _, rf = b.generate(turn2, max_new_tokens=8, stop_tokens=stop)
        assert rf == 0 and got == fresh
        if not bounded:
            assert reused == len(PROMPT) + len(first) - 1 or reused == len(PROMPT) + len(first)      # everything the caches held (the last sampled token was never decoded in)
        # a diverging prompt reuses only the common part
        other = PROMPT[:4] + [600, 601]
        got3, _, reused3 = a.generate(other, max_new_tokens=4, stop_tokens=stop)
        fresh3, _, _ = b.generate(other, max_new_tokens=4, stop_tokens=stop)
        assert got3 == fresh3 and reused3 <= 4
    finally:
        a.close()
        b.close()


def test_bounded_ring_refuses_a_rewind_whose_stale_tail_wrapped_over_the_window():
    """CudaGqaOp.ixx:182-203: a bounded ring accepts a rewind only while the stale tail is at most capacity - window (= chunk - 1) tokens; a tail of
    chunk .. capacity - 1 tokens has overwritten ring slots the continuation's window still needs (the port used to accept it and
    attended to future K/V).  window 16, chunk 8 => capacity 23; 45 positions written, divergence 15 before the fill: refused, full prefill, and the
    tokens are a fresh model's"""
    cfg = dict(SMALL, bounded_local_kv=1)
    rng = np.random.default_rng(11)
    long_prompt = [int(t) for t in rng.integers(2, 1000, 40)]
    stop = [1023]
    a = host.GemmaModel.synthetic("bf16", cfg, context=64, prefill_chunk=8, seed=5)
    b = host.GemmaModel.synthetic("bf16", cfg, context=64, prefill_chunk=8, seed=5)
    try:
        first, _, _ = a.generate(long_prompt, max_new_tokens=6, stop_tokens=stop)
        written = len(long_prompt) + len(first) - 1
        for keep in (written - 15, written - 8, written - 22):          # stale tails of chunk .. capacity - 1 tokens: every one must be refused
            other = long_prompt[:keep] + [601, 602, 603]
            a.generate(long_prompt, max_new_tokens=6, stop_tokens=stop)                    # refill the caches to `written`
            got, _, reused = a.generate(other, max_new_tokens=6, stop_tokens=stop)
            fresh, _, _ = b.generate(other, max_new_tokens=6, stop_tokens=stop)
            b.generate([5], max_new_tokens=1, stop_tokens=stop)                           # b never reuses: its history is one unrelated token
            assert reused == 0, keep
            assert got == fresh, keep
        # a tail of at most chunk - 1 tokens is still served from the caches, with the same tokens
        a.generate(long_prompt, max_new_tokens=6, stop_tokens=stop)
        other = (long_prompt + first)[:written - 5] + [604]
        got, _, reused = a.generate(other, max_new_tokens=6, stop_tokens=stop)
        fresh, _, _ = b.generate(other, max_new_tokens=6, stop_tokens=stop)
        assert reused == written - 5 and got == fresh
    finally:
        a.close()
        b.close()


def test_stochastic_sampling_is_seeded_and_top_k_1_is_greedy():
    """SamplingParams (Components/Transformers/SamplingParams.ixx): top_k = 1 == greedy; the same seed gives the same draw sequence"""
    g = host.GemmaModel.synthetic("bf16", SMALL, context=64, prefill_chunk=16, seed=21)
    try:
        greedy, _, _ = g.generate(PROMPT, max_new_tokens=10, stop_tokens=[1023], top_k=1)
        also, _, _ = g.generate(PROMPT, max_new_tokens=10, stop_tokens=[1023], temperature=0.7, top_k=1, top_p=0.9)
        assert also == greedy
        s1, _, _ = g.generate(PROMPT, max_new_tokens=10, stop_tokens=[1023], temperature=1.3, top_k=50, top_p=0.95, seed=42)
        s2, _, _ = g.generate(PROMPT, max_new_tokens=10, stop_tokens=[1023], temperature=1.3, top_k=50, top_p=0.95, seed=42)
        s3, _, _ = g.generate(PROMPT, max_new_tokens=10, stop_tokens=[1023], temperature=1.3, top_k=50, top_p=0.95, seed=43)
        assert s1 == s2 and len(s1) == 10 and all(0 <= t < 1024 for t in s1)
        assert s1 != greedy or s3 != greedy
    finally:
        g.close()


@pytest.mark.parametrize("container", ["bin", "safetensors"])
def test_from_pretrained_reads_the_geometry_from_the_artifact(container, tmp_path):
    """GemmaModel.ixx:604-665: geometry from the checkpoint metadata; a bf16 artifact quantizes on load under any policy; a pre-quantized one loads only under
    its own; a context beyond the trained length is refused"""
    src = host.Gemma("bf16", SMALL, max_seq=64, max_prefill=16, seed=9)
    q4 = host.Gemma("fp4", SMALL, max_seq=64, max_prefill=16, seed=9)
    path, qpath = tmp_path / ("m." + container), tmp_path / ("q." + container)
    try:
        (src.save_milabin if container == "bin" else src.save_safetensors)(path)
        (q4.save_milabin if container == "bin" else q4.save_safetensors)(qpath)
    finally:
        src.close()
        q4.close()
    want = manual_greedy("bf16", PROMPT, 8, seed=9)
    g = host.GemmaModel.from_pretrained(path, "bf16", context=64, prefill_chunk=16)
    try:
        assert g.generate(PROMPT, max_new_tokens=8, stop_tokens=[1023])[0] == want
    finally:
        g.close()
    want4 = manual_greedy("fp4", PROMPT, 8, seed=9)
    for p in (path, qpath):                    # quantize-on-load and the packed artifact give the same model
        g = host.GemmaModel.from_pretrained(p, "fp4", context=64, prefill_chunk=16)
        try:
            assert g.generate(PROMPT, max_new_tokens=8, stop_tokens=[1023])[0] == want4
        finally:
            g.close()
    if container == "safetensors":             # the artifact declares its policy (__metadata__["mila_quantization"])
        with pytest.raises(RuntimeError, match="pre-quantized as 'per_group_fp4_128' but this load requested 'none'"):
            host.GemmaModel.from_pretrained(qpath, "bf16", context=64)
    else:                                      # a .bin carries no declaration (PretrainedReader.ixx:330-344): the packed blobs are refused by the Linear they do not fit
        with pytest.raises(ValueError, match="does not fit this Linear's weight policy"):
            host.GemmaModel.from_pretrained(qpath, "bf16", context=64)
    with pytest.raises(ValueError, match="exceeds trained max_seq_len"):
        host.GemmaModel.from_pretrained(path, "bf16", context=1 << 20)
