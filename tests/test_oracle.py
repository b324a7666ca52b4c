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


def test_oracle_matches_reference_at_full_tinyllama_size(gguf):
    """BASELINE.json configs[0]: the real reference at its compiled-in TinyLlama-1.1B dims, 320 positions on the synthetic
    weights bench.py uses (compact golden).  The oracle replays the first positions teacher-forced with the reference's
    own tokens (a CPU token is ~0.3 s here; the GPU tests replay all 320)."""
    g = load_golden("tinyllama")
    fw = gguf.synth_fused(gguf.SHAPES["tinyllama"], int(g["seed"]))
    n = 12
    toks, logits = Oracle(fw, "omp").generate(n, prompt=g["tokens"][:n].tolist())
    err = compact_err(logits, g, n)
    assert err.max() <= 1e-5, err
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
