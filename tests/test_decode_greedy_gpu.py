"""llmk_decode_greedy: the temperature-0 generation loop (llama2.f90:379-396) with the argmax on the device and the
launches enqueued back to back.  Bar: the ids are the real reference's own greedy ids (goldens), and bit-identical to
what the per-token entry points return."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden, safe_positions
from llm_f90_amd import llmk

pytestmark = pytest.mark.gpu
LLM = os.path.join(ROOT, "llm.f90_amd", "host", "llm")


@pytest.mark.parametrize("tag", ["tk-small", "tk-small-long", "tiny-gqa", "tiny-hs128"])
@pytest.mark.parametrize("flags", [0, llmk.FLAG_MULTI_KERNEL], ids=["default", "multikernel"])
def test_decode_greedy_returns_the_reference_ids(tag, flags, gguf):
    """tk-small* run the pipelined persistent kernel, the tiny shapes the per-token fallback inside the same call."""
    g = load_golden(tag)
    fw = gguf.synth_fused(gguf.SHAPES[str(g["shape"])], int(g["seed"]))
    n = int(g["n"])
    m = llmk.Llmk(fw, flags=flags)
    if tag.startswith("tk-small") and flags == 0:
        assert m.path() == 1
    seen = []
    ids = m.decode_greedy(2, 1, n, on_token=lambda i, t, u: seen.append((i, t)))
    assert np.array_equal(ids, g["tokens"])                       # the real reference's transcript, id for id
    assert seen == list(enumerate(ids.tolist()))                  # streamed in order, each id once
    m.reset()
    toks, _ = m.generate(n, want_logits=False, greedy_on_device=True)
    assert np.array_equal(ids, toks)
    m.close()


def test_decode_greedy_resumes_after_forward_and_prefill(gguf):
    """positions 1..k through llmk_forward / llmk_prefill (a prompt), the rest in one pipelined call; twice in a row
    (the candidate buffers and id slots are reused); the KV rows it leaves are the per-token path's."""
    g = load_golden("tk-small-long")
    fw = gguf.synth_fused(gguf.SHAPES[str(g["shape"])], int(g["seed"]))
    ref = g["tokens"]
    n = 200
    m = llmk.Llmk(fw)
    for k in (1, 7, 130):
        m.reset()
        tok = 2
        for pos in range(1, k + 1):
            tok = int(np.argmax(m.forward(tok, pos))) + 1
        assert tok == ref[k - 1]
        ids = m.decode_greedy(tok, k + 1, n - k)
        assert np.array_equal(ids, ref[k:n]), k
    m.reset()
    k = 129
    lg = m.prefill([2] + ref[:k - 1].tolist(), 1)
    ids = m.decode_greedy(int(np.argmax(lg)) + 1, k + 1, n - k)
    assert np.array_equal(ids, ref[k:n])
    kv = m.peek(4, fw.shape.kv_dim, 1, n)
    m.reset()
    m.generate(n, want_logits=False)
    # (the pipelined variant is its own instantiation of the kernel: hipcc may contract an epilogue's multiply-adds
    # differently, so the rows agree to rounding, like every other KV check in the suite, not bit for bit)
    np.testing.assert_allclose(kv, m.peek(4, fw.shape.kv_dim, 1, n), rtol=0, atol=1e-5)
    with pytest.raises(llmk.LlmkError):
        m.decode_greedy(2, fw.shape.seq_len, 2)                   # runs past the context
    m.close()


def test_decode_greedy_tinyllama_full_size_matches_the_real_reference(gguf):
    """BASELINE.json configs[1] at its real size: 256 positions in one call against the reference's own ids."""
    g = load_golden("tinyllama")
    s = gguf.SHAPES["tinyllama"]
    m = llmk.Llmk(gguf.synth_fused(s, int(g["seed"])))
    assert m.path() == 1
    n = 256
    ids = m.decode_greedy(2, 1, n)
    safe = safe_positions(g, n)
    first_bad = int(np.argmin(safe)) if not safe.all() else n     # teacher forcing is impossible here: compare up to the
    assert first_bad > 64                                         # first position whose top-1 margin is inside the tolerance
    assert np.array_equal(ids[:first_bad], g["tokens"][:first_bad])
    m.reset()
    toks, _ = m.generate(n, want_logits=False)                    # host consumer, same kernel arithmetic: identical ids
    assert np.array_equal(ids, toks)
    m.close()


@pytest.mark.parametrize("wtype,shape", [(1, "tk-small16")])
def test_decode_greedy_f16_kernel(wtype, shape, gguf):
    fw = gguf.synth_fused(gguf.SHAPES[shape], 4242, wtype)
    m = llmk.Llmk(fw)
    assert m.path() == 1
    n = fw.shape.seq_len
    ids = m.decode_greedy(2, 1, n)
    m.reset()
    toks, _ = m.generate(n, want_logits=False)
    assert np.array_equal(ids, toks)
    m.close()


def test_cli_device_argmax_streams_the_reference_transcript(gguf, tmp_path):
    """`llm --device-argmax` = one llmk_decode_greedy call after the prompt; stdout is the reference's, byte for byte."""
    for tag in ("tk-small", "tk-small-prompt", "tk-small-long-prompt"):
        g = load_golden(tag)
        path = str(tmp_path / "m.gguf")
        gguf.write_synth_gguf(path, gguf.SHAPES[str(g["shape"])], int(g["seed"]))
        args = [LLM, "-m", path, "-n", str(int(g["n"])), "-t", "0", "--device-argmax"]
        if str(g["prompt"]):
            args += ["-p", str(g["prompt"])]
        r = subprocess.run(args, capture_output=True, cwd=str(tmp_path), timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        out, ref = r.stdout.split(b"\n"), bytes(g["stdout"]).split(b"\n")
        k = next(i for i, l in enumerate(ref) if l.startswith(b" Inference time:"))
        assert out[:k] == ref[:k], tag


@pytest.mark.parametrize("at", [1, 7, 32], ids=["first-launch", "mid-pipeline", "last-launch"])
def test_decode_greedy_timeout_inside_the_pipeline_still_returns_the_reference_ids(at):
    """Advisor (round 3, medium): a launch of the pipelined greedy decode that times out leaves garbage candidates, and the
    launches behind it used to fold them and publish ids from them -- llmk_decode_greedy returned LLMK_OK with a wrong
    transcript.  Now no id is published while the device error word is set; the host retires the token kernel and redoes the
    positions from the first missing id on the multi-kernel path.  The debug library launches position `at` one workgroup
    short (LLMK_TK_INJECT_TIMEOUT), so the timeout is real: first, middle and last launch of the pipeline."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    dbg = os.path.join(ROOT, "llm.f90_amd", "csrc", "libllmk_debug.so")
    assert os.path.exists(dbg), "libllmk_debug.so not built (make -C llm.f90_amd debug)"
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "import llm_f90_amd\n"
        "from llm_f90_amd import llmk\n"
        "from llm_f90_amd.tools import gguf\n"
        "from conftest import load_golden\n"
        "g = load_golden('tk-small')\n"
        "fw = gguf.synth_fused(gguf.SHAPES['tk-small'], int(g['seed']))\n"
        "m = llmk.Llmk(fw)\n"
        "assert m.path() == 1\n"
        "seen = []\n"
        "ids = m.decode_greedy(2, 1, int(g['n']), on_token=lambda i, t, u: seen.append((i, t)))\n"
        "assert np.array_equal(ids, g['tokens']), (ids.tolist(), g['tokens'].tolist())\n"
        "assert seen == [(i, int(t)) for i, t in enumerate(g['tokens'])], seen\n"     # streamed once each, in order, the right ids
        "assert m.path() == 0\n"                                                      # retired to the multi-kernel path
        "print('PIPELINE-FALLBACK-OK')\n")
    env = dict(os.environ, LLMK_LIB=dbg, LLMK_TK_INJECT_TIMEOUT=str(at))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, env=env, timeout=300)
    assert r.returncode == 0 and b"PIPELINE-FALLBACK-OK" in r.stdout, r.stdout + r.stderr
    assert b"timed out" in r.stderr and b"multi-kernel path" in r.stderr


@pytest.mark.parametrize("shape", ["tiny-hs64", "tk-small"], ids=["multikernel", "token-kernel"])
def test_greedy_without_a_finite_logit_is_an_error_not_token_zero(shape, gguf):
    """Advisor, round 4: the device argmax answers 0 ("no token") when no logit is finite; llmk_forward_greedy returned LLMK_OK with
    that 0 and llmk_decode_greedy handed it to the caller's callback -- the Fortran host indexes its vocabulary with it.  Now both
    return LLMK_E_NONFINITE (11) before any id or callback; a final rmsnorm gain of NaN makes every logit NaN on both paths (the
    persistent kernel's pipeline raises its sticky word, retires, and the multi-kernel redo reports the error)."""
    from llm_f90_amd import llmk
    fw = gguf.synth_fused(gguf.SHAPES[shape], 3)
    fw.rms_final_weight = np.full_like(fw.rms_final_weight, np.nan)
    m = llmk.Llmk(fw)
    with pytest.raises(llmk.LlmkError) as e:
        m.forward_greedy(2, 1)
    assert e.value.code == 11
    m.reset()
    seen = []
    with pytest.raises(llmk.LlmkError) as e:
        m.decode_greedy(2, 1, 4, on_token=lambda i, t, u: seen.append(t))
    assert e.value.code == 11 and seen == []
    m.close()
