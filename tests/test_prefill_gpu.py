"""Batched prompt prefill (llmk_prefill): one call must leave behind what the reference's
token-by-token prompt loop (llama2.f90:376-402) leaves -- the KV cache rows of the prompt positions and the logits
of the last one -- within the 1e-4 relative parity bar, and generation must continue with identical token ids."""
import numpy as np
import pytest

from conftest import REL_TOL, load_golden, rel_err
from llm_f90_amd import llmk
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def continue_greedy(m, first_logits, pos_next, n):
    """greedy loop from the logits of the last prefilled position: returns tokens and logits of the next n positions"""
    toks, lgs = [], []
    tok = int(np.argmax(first_logits)) + 1
    for pos in range(pos_next, pos_next + n):
        toks.append(tok)
        lg = m.forward(tok, pos)
        lgs.append(lg)
        tok = int(np.argmax(lg)) + 1
    return np.asarray(toks, np.int32), np.asarray(lgs)


@pytest.mark.parametrize("tag", ["tiny-gqa-prompt", "tk-small-prompt"])
def test_prefill_matches_reference_golden(tag, gguf):
    """The REAL reference's run on a prompt: prefill [BOS, prompt...] in one call, then decode the rest."""
    g = load_golden(tag)
    fw = gguf.synth_fused(gguf.SHAPES[str(g["shape"])], int(g["seed"]))
    pids = g["prompt_ids"].tolist()
    k, n = len(pids), int(g["n"])
    m = llmk.Llmk(fw)
    lg = m.prefill([2] + pids, 1)                       # positions 1..k+1
    assert rel_err(lg[None], g["logits"][k][None]).max() <= REL_TOL
    toks, lgs = continue_greedy(m, lg, k + 2, n - k - 1)
    assert np.array_equal(toks, g["tokens"][k:n - 1])   # tokens[k] is the first sampled one
    assert rel_err(lgs, g["logits"][k + 1:n]).max() <= REL_TOL
    m.close()


@pytest.mark.parametrize("shape,n", [("tiny-gqa", 1), ("tiny-gqa", 17), ("tiny-mha", 33), ("tiny-hs64", 40), ("tiny-hs128", 20),
                                     ("tiny-70bish", 24)])
def test_prefill_matches_oracle(shape, n, gguf):
    """seeded pseudo-prompts of ragged lengths (1, 17, 33 ... not multiples of the 16-token MFMA group)"""
    s = gguf.SHAPES[shape]
    fw = gguf.synth_fused(s, 777)
    rng = np.random.default_rng(5)
    prompt = [2] + (rng.integers(3, s.vocab_size, n - 1) + 1).tolist()
    o = Oracle(fw, "omp")
    ol = None
    for pos, tok in enumerate(prompt, 1):
        ol = o.forward(tok, pos)
    m = llmk.Llmk(fw)
    lg = m.prefill(prompt, 1)
    assert rel_err(lg[None], ol[None]).max() <= REL_TOL
    m.close()


@pytest.mark.parametrize("tag,ks", [("tk-small-long", [255, 256, 257, 300, 513, 704]), ("tiny-hs128-long", [127, 129, 200, 257, 320])])
def test_prefill_long_prompts_match_the_real_reference(tag, ks, gguf):
    """Long-context goldens from the real reference: the first k positions of its transcript go through llmk_prefill in
    one call; the logits of position k must match the reference's -- the batched attention (attn_kernel with a batch
    dimension) crosses its 256- (head size 64) / 128-timestep (head size 128) tile, at batch boundaries and inside."""
    g = load_golden(tag)
    fw = gguf.synth_fused(gguf.SHAPES[str(g["shape"])], int(g["seed"]))
    fed = [2] + g["tokens"].tolist()                     # token fed at position p (1-based) is fed[p-1]
    m = llmk.Llmk(fw)
    for k in ks:
        m.reset()
        lg = m.prefill(fed[:k], 1)
        assert rel_err(lg[None], g["logits"][k - 1][None]).max() <= REL_TOL, k
    # prefill, then decode across the next tile boundary on the same cache
    m.reset()
    k = ks[0] - 20
    lg = m.prefill(fed[:k], 1)
    for pos in range(k + 1, k + 41):
        lg = m.forward(fed[pos - 1], pos)
        assert rel_err(lg[None], g["logits"][pos - 1][None]).max() <= REL_TOL, pos
    m.close()


def test_prefill_in_two_calls_and_after_decode(gguf):
    """prefill may start at any position: decode 5 tokens, prefill 20 more, compare with the all-sequential run"""
    s = gguf.SHAPES["tiny-gqa"]
    fw = gguf.synth_fused(s, 31)
    rng = np.random.default_rng(9)
    seq = [2] + (rng.integers(3, s.vocab_size, 29) + 1).tolist()
    a = llmk.Llmk(fw)
    for pos, tok in enumerate(seq, 1):
        ref = a.forward(tok, pos)
    b = llmk.Llmk(fw)
    for pos in range(1, 6):
        b.forward(seq[pos - 1], pos)
    b.prefill(seq[5:18], 6)
    lg = b.prefill(seq[18:], 19)
    assert rel_err(lg[None], ref[None]).max() <= REL_TOL
    a.close(); b.close()


def test_prefill_tinyllama_long_prompt_vs_sequential(gguf):
    """BASELINE.json's shape, a 150-token prompt (two batches: 128 + 22), then 8 decoded tokens on the token kernel"""
    s = gguf.SHAPES["tinyllama"]
    fw = gguf.synth_fused(s, 20260928)
    rng = np.random.default_rng(1)
    prompt = [2] + (rng.integers(3, s.vocab_size, 149) + 1).tolist()
    a = llmk.Llmk(fw)
    for pos, tok in enumerate(prompt, 1):
        ref = a.forward(tok, pos)
    rt, rl = continue_greedy(a, ref, len(prompt) + 1, 8)
    b = llmk.Llmk(fw)
    lg = b.prefill(prompt, 1)
    assert rel_err(lg[None], ref[None]).max() <= REL_TOL
    t, l = continue_greedy(b, lg, len(prompt) + 1, 8)
    assert np.array_equal(t, rt)
    assert rel_err(l, rl).max() <= REL_TOL
    a.close(); b.close()


def test_prefill_argument_errors(gguf):
    s = gguf.SHAPES["tiny-gqa"]
    m = llmk.Llmk(gguf.synth_fused(s, 1))
    with pytest.raises(llmk.LlmkError):
        m.prefill([2, 0, 5], 1)                          # token id out of range
    with pytest.raises(llmk.LlmkError):
        m.prefill([2] * 4, s.seq_len - 2)                # runs past the context
    with pytest.raises(llmk.LlmkError):
        m.prefill([], 1)
    m.close()


def test_prefill_falls_back_to_the_token_by_token_pass_when_the_shape_does_not_tile(gguf):
    """a hidden size that is not a multiple of the 64-column step (tiny-mha: H = 352): llmk_prefill runs the
    token-by-token pass inside and is bit-identical to llmk_forward"""
    fw = gguf.synth_fused(gguf.SHAPES["tiny-mha"], 4242, 2)
    prompt = [2, 40, 41, 42, 43]
    a = llmk.Llmk(fw)
    for pos, tok in enumerate(prompt, 1):
        ref = a.forward(tok, pos)
    b = llmk.Llmk(fw)
    assert np.array_equal(b.prefill(prompt, 1), ref)
    a.close(); b.close()


@pytest.mark.parametrize("shape,n", [("tiny-gqa", 19), ("tiny-70bish", 33), ("tk-small16", 40), ("tk-small", 7),
                                     ("tk-small-long", 100), ("tk-small-long", 150), ("tk-small-long", 300), ("tiny-hs64", 33)])
def test_prefill_q4_matches_oracle_on_decoded_weights(shape, n, gguf):
    """q4_0 matrices: the A operand of the MFMA is (nibble - 8) * d computed in registers; pinned, like the q4_0 decode
    path, to the f32 reference arithmetic on the host-decoded weights; then decoding continues from the cache"""
    s = gguf.SHAPES[shape]
    fw = gguf.synth_fused(s, 4242, 2)
    rng = np.random.default_rng(4)
    prompt = [2] + (rng.integers(3, s.vocab_size, n - 1) + 1).tolist()
    o = Oracle(fw.as_f32(), "omp")
    for pos, tok in enumerate(prompt, 1):
        ol = o.forward(tok, pos)
    m = llmk.Llmk(fw)
    lg = m.prefill(prompt, 1)
    assert rel_err(lg[None], ol[None]).max() <= REL_TOL
    tok = int(np.argmax(ol)) + 1
    for pos in range(n + 1, min(n + 5, s.seq_len) + 1):
        ol = o.forward(tok, pos)
        lg = m.forward(tok, pos)
        assert rel_err(lg[None], ol[None]).max() <= REL_TOL
        tok = int(np.argmax(ol)) + 1
    m.close()


@pytest.mark.parametrize("shape,n", [("tiny-gqa", 19), ("tiny-hs64", 33), ("tk-small16", 40), ("tiny-hs128", 5),
                                     ("tk-small-long", 100), ("tk-small-long", 150), ("tiny-hs128-long", 300)])
def test_prefill_f16_matches_oracle_on_decoded_weights(shape, n, gguf):
    """f16 matrices go through the same batched MFMA GEMMs (exact half -> float conversion of the A operand): pinned, like
    the f16 decode path, to the f32 reference arithmetic on the host-decoded weights; then decoding continues from the cache"""
    s = gguf.SHAPES[shape]
    fw = gguf.synth_fused(s, 4242, 1)
    rng = np.random.default_rng(3)
    prompt = [2] + (rng.integers(3, s.vocab_size, n - 1) + 1).tolist()
    o = Oracle(fw.as_f32(), "omp")
    for pos, tok in enumerate(prompt, 1):
        ol = o.forward(tok, pos)
    m = llmk.Llmk(fw)
    lg = m.prefill(prompt, 1)
    assert rel_err(lg[None], ol[None]).max() <= REL_TOL
    tok = int(np.argmax(ol)) + 1
    for pos in range(n + 1, min(n + 5, s.seq_len) + 1):
        ol = o.forward(tok, pos)
        lg = m.forward(tok, pos)
        assert rel_err(lg[None], ol[None]).max() <= REL_TOL
        tok = int(np.argmax(ol)) + 1
    m.close()


@pytest.mark.parametrize("wtype", [0, 1, 2], ids=["f32", "f16", "q4_0"])
def test_prefill_range_guard_redoes_the_call_on_the_f32_instruction(wtype, gguf):
    """The GEMMs multiply on v_mfma_f32_16x16x32_f16 with the f32 activation as two f16 pieces (prefill.h
    pf_gemm_h_kernel): an activation of magnitude >= 65504 fits neither piece.  FFN norm gains of 1e5 put the w1|w3 input
    at ~1e5 and the w2 input at ~1e9 (attention inputs stay normalised, so nothing chaotic happens to the softmax): without
    the guard hi would clamp and the logits would be garbage; with it the call is redone on the f32 instruction and matches
    the oracle, and so does a second call (the context stays on the f32 instruction) and the decode that follows."""
    s = gguf.SHAPES["tk-small16"]
    fw = gguf.synth_fused(s, 77, wtype)
    fw.rms_ffn_weight = (fw.rms_ffn_weight * np.float32(1e5)).astype(np.float32)
    rng = np.random.default_rng(11)
    n = 37
    prompt = [2] + (rng.integers(3, s.vocab_size, n - 1) + 1).tolist()
    o = Oracle(fw.as_f32() if wtype else fw, "omp")
    for pos, tok in enumerate(prompt, 1):
        ol = o.forward(tok, pos)
    assert np.all(np.isfinite(ol))
    m = llmk.Llmk(fw)
    lg = m.prefill(prompt, 1)
    assert np.all(np.isfinite(lg))
    assert rel_err(lg[None], ol[None]).max() <= REL_TOL
    m.reset()
    lg2 = m.prefill(prompt, 1)
    assert np.array_equal(lg, lg2)
    tok = int(np.argmax(ol)) + 1
    for pos in range(n + 1, n + 4):
        ol = o.forward(tok, pos)
        lg = m.forward(tok, pos)
        assert rel_err(lg[None], ol[None]).max() <= REL_TOL
        tok = int(np.argmax(ol)) + 1
    m.close()


@pytest.mark.parametrize("wtype", [0, 1], ids=["f32", "f16"])
def test_prefill_small_activations_are_redone_on_the_f32_instruction(wtype, gguf):
    """The LOW end of the two-piece split (advisor, round 3): below 2^-3 the lo piece is an f16 subnormal and a pair's error is
    2^-24 absolute -- 2e-4 of a value of 3e-4.  Attention norm gains of 2^-10 put the QKV input, and with it the attention
    output (the wo GEMM's input), at ~1e-3..1e-4 for every position; wo is scaled by 2^10 (exact in f32 and in f16) so the
    residual stream keeps its size and the error would reach the logits.  The staging sees a position whose largest element
    is below 2^-7, raises the range flag, and the call is redone on the f32 instruction: logits within 1e-4 of the oracle."""
    s = gguf.SHAPES["tk-small16"]
    fw = gguf.synth_fused(s, 78, wtype)
    fw.rms_att_weight = (fw.rms_att_weight * np.float32(2.0 ** -10)).astype(np.float32)
    fw.wo = (fw.wo * fw.wo.dtype.type(1024)).astype(fw.wo.dtype)
    rng = np.random.default_rng(12)
    n = 41
    prompt = [2] + (rng.integers(3, s.vocab_size, n - 1) + 1).tolist()
    o = Oracle(fw.as_f32() if wtype else fw, "omp")
    for pos, tok in enumerate(prompt, 1):
        ol = o.forward(tok, pos)
    assert np.all(np.isfinite(ol))
    m = llmk.Llmk(fw)
    lg = m.prefill(prompt, 1)
    assert rel_err(lg[None], ol[None]).max() <= REL_TOL
    m.close()
