"""The oracle (oracle/llm_oracle.c, a C restatement of /root/reference/llama2.f90:480-640) against
golden vectors produced by the real reference (tests/golden/make_golden.py). CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_CASES, REL_TOL, compact_err, load_golden, rel_err, safe_positions
from oracle.oracle import Oracle


@pytest.mark.parametrize("tag", GOLDEN_CASES)
@pytest.mark.parametrize("flavour", ["strict", "omp", "fast"])
def test_oracle_matches_reference(tag, flavour, gguf):
    g = load_golden(tag)
    shape = gguf.SHAPES[str(g["shape"])]
    fw = gguf.synth_fused(shape, int(g["seed"]))
    toks, logits = Oracle(fw, flavour).generate(int(g["n"]), prompt=g["prompt_ids"].tolist())
    err = rel_err(logits, g["logits"])
    # the restatement follows the reference's operation order: expect ~1e-6, require <= 1e-5
    assert err.max() <= 1e-5, err
    assert err.max() <= REL_TOL
    assert np.array_equal(toks, g["tokens"])            # bit-exact greedy ids (1-based)


def test_reference_tokenizer_restatement_on_a_vocabulary_with_merges(gguf):
    """tests/golden/tiny-gqa-merge.npz: the real reference on a vocabulary WITH merges (tools/gguf.py merge_vocab).  Its
    prompt ids are not observable directly -- merged or not, the prompt prints the same text -- but its LOGITS are: the oracle
    teacher-forced with the ids of the restated merge loop (oracle.bpe_encode_ref, llama2.f90:658-724) reproduces them, and the
    number of prompt positions fixes where the greedy tokens start.  The rules the vocabulary was built to expose hold."""
    from oracle.oracle import bpe_encode_ref
    g = load_golden("tiny-gqa-merge")
    shape = gguf.SHAPES[str(g["shape"])]
    vocab, scores = gguf.merge_vocab(shape.vocab_size)
    pids = bpe_encode_ref(str(g["prompt"]).encode(), vocab, scores)
    assert pids == g["prompt_ids"].tolist() and len(pids) < len(str(g["prompt"]))      # merges happened
    toks, logits = Oracle(gguf.synth_fused(shape, int(g["seed"])), "strict").generate(int(g["n"]), prompt=pids)
    assert rel_err(logits, g["logits"]).max() <= 1e-5
    assert np.array_equal(toks, g["tokens"])
    # one token per character instead (no merging) is a different computation: the golden discriminates
    flat = [3 + c - 32 + 1 for c in str(g["prompt"]).encode()][:int(g["n"])]
    _, l2 = Oracle(gguf.synth_fused(shape, int(g["seed"])), "strict").generate(int(g["n"]), prompt=flat)
    assert rel_err(l2, g["logits"]).max() > 1e-2
    words = lambda t: [vocab[i - 1] for i in bpe_encode_ref(t, vocab, scores)]
    assert words(b"ing") == [b"in", b"g"]                    # equal scores: the FIRST pair wins (llama2.f90:694)
    assert words(b"her") == [b"he", b"r"]                    # duplicated "er": the first entry's score counts (llama2.f90:648-651)
    assert words(b"other") == [b"o", b"the", b"r"]           # ... with the second entry's score it would be ["other"]
    assert words(b" the") == [b" the"] and words(b"and") == [b"and"]


@pytest.mark.parametrize("tag", ["tinyllama", "tinyllama-f16dec"])
def test_oracle_matches_reference_at_full_tinyllama_size(tag, gguf):
    """BASELINE.json configs[0]: the real reference at its compiled-in TinyLlama-1.1B dims, 320 positions on the synthetic
    weights bench.py uses (compact golden).  The oracle replays the first positions teacher-forced with the reference's
    own tokens (a CPU token is ~0.3 s here; the GPU tests replay all 320).  `-f16dec` (configs[2]): the same run on the
    matrices rounded to f16 and decoded back -- what the oracle is handed when it checks the f16 kernels."""
    g = load_golden(tag)
    fw = gguf.synth_fused(gguf.SHAPES["tinyllama"], int(g["seed"]), 1).as_f32() if tag.endswith("f16dec") else \
        gguf.synth_fused(gguf.SHAPES["tinyllama"], int(g["seed"]))
    n = 12
    toks, logits = Oracle(fw, "omp").generate(n, prompt=g["tokens"][:n].tolist())
    err = compact_err(logits, g, n)
    assert err.max() <= 1e-5, err
    ok = safe_positions(g, n)
    assert np.array_equal((np.argmax(logits, axis=1) + 1)[ok], g["tokens"][:n][ok])


def test_the_two_full_size_tinyllama_goldens_are_one_run():
    """tests/golden/tinyllama-long.npz (2,048 positions, round 6) and tinyllama.npz (320, round 1) come from the same reference
    binary on the same weights: their common positions must hold the same ids, top-8 and checksums, bit for bit."""
    a, b = load_golden("tinyllama"), load_golden("tinyllama-long")
    n = int(a["n"])
    assert int(b["n"]) == 2048 and np.array_equal(a["tokens"], b["tokens"][:n])
    for k in ("top8_idx", "top8_val", "probe_idx", "lsum", "l2", "absmax"):
        assert np.array_equal(a[k], b[k][:n] if b[k].shape[0] == 2048 else b[k]), k


@pytest.mark.skipif(not os.environ.get("LLMK_BIG_ORACLE"), reason="~10 minutes of 8 cores: LLMK_BIG_ORACLE=1")
def test_oracle_matches_reference_over_the_whole_tinyllama_context(gguf):
    """The oracle's own pin at long contexts and full size: all 2,048 positions of tests/golden/tinyllama-long.npz, teacher-forced
    (the GPU tests replay the same golden on the device: test_parity_gpu.py)."""
    g = load_golden("tinyllama-long")
    fw = gguf.synth_fused(gguf.SHAPES["tinyllama"], int(g["seed"]))
    n = int(os.environ.get("LLMK_BIG_ORACLE_N", g["n"]))
    toks, logits = Oracle(fw, "omp").generate(n, prompt=g["tokens"][:n].tolist())
    err = compact_err(logits, g, n)
    assert err.max() <= 1e-5, (err.max(), int(np.argmax(err)))
    ok = safe_positions(g, n)
    assert np.array_equal((np.argmax(logits, axis=1) + 1)[ok], g["tokens"][:n][ok])


@pytest.mark.skipif(not os.environ.get("LLMK_BIG_ORACLE"), reason="31 GB of host memory and ~3 minutes: LLMK_BIG_ORACLE=1")
def test_oracle_matches_reference_at_full_llama2_7b_depth(gguf):
    """BASELINE.json configs[3] at FULL depth: tests/golden/llama2-7b.npz is the real reference (dims patched to Llama-2-7B)
    on the q4_0 blocks of synth_fused_q4_direct decoded to f32; the oracle replays the first positions teacher-forced."""
    g = load_golden("llama2-7b-prompt")
    fw = gguf.synth_fused_q4_direct(gguf.SHAPES["llama2-7b"], int(g["seed"])).as_f32()
    n = 6
    toks, logits = Oracle(fw, "omp").generate(n, prompt=g["tokens"][:n].tolist())
    err = compact_err(logits, g, n)
    assert err.max() <= 1e-5, err
    ok = safe_positions(g, n)
    assert np.array_equal((np.argmax(logits, axis=1) + 1)[ok], g["tokens"][:n][ok])


def test_omp_flavour_is_bit_identical_to_strict(gguf):
    shape = gguf.SHAPES["tiny-hs64"]
    fw = gguf.synth_fused(shape, 7)
    _, a = Oracle(fw, "strict").generate(8)
    _, b = Oracle(fw, "omp").generate(8)
    assert np.array_equal(a, b)


def test_rope_exponent_quirk_is_observable(gguf):
    """pair j uses 10000^-((2j+1)/hs). If the oracle used llama2.c's 2j/hs the
    logits would be far off from pos 2 on; pin that the golden really discriminates."""
    g = load_golden("tiny-gqa")
    assert np.abs(g["logits"][1] - g["logits"][0]).max() > 1e-2


def test_oracle_rejects_bad_indices(gguf):
    fw = gguf.synth_fused(gguf.SHAPES["tiny-gqa"], 1)
    o = Oracle(fw)
    with pytest.raises(ValueError):
        o.forward(0, 1)
    with pytest.raises(ValueError):
        o.forward(1, 65)
