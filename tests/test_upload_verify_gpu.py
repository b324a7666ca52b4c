"""Uploads are verified (three times in round 4 a block of weights reached the device with other bytes than
the host held): the shim sums the tensor's rows on the device and compares with what the CPU staged, stages the block again on a
mismatch, and gives up with LLMK_E_VERIFY after three attempts.  The debug library can damage the first uploaded block on purpose
(LLMK_UPLOAD_INJECT = number of damaged attempts)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "llm.f90_amd", "csrc", "libllmk_debug.so")

CODE = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
from oracle.oracle import Oracle
fw = gguf.synth_fused(gguf.SHAPES["tiny-hs64"], 20260930)
try:
    m = llmk.Llmk(fw, device=0)
except llmk.LlmkError as e:
    print("CREATE-FAILED", e)
    raise SystemExit(0)
toks, logits = m.generate(6)
ot, ol = Oracle(fw, "strict").generate(6)
err = np.max(np.abs(logits - ol), axis=1) / np.max(np.abs(ol), axis=1)
assert err.max() <= 1e-4 and np.array_equal(toks, ot), err
print("PARITY-OK")
""" % ROOT


def _run(inject):
    env = dict(os.environ, LLMK_LIB=DBG, LLMK_UPLOAD_INJECT=str(inject))
    return subprocess.run([sys.executable, "-c", CODE], capture_output=True, env=env, timeout=300, cwd=ROOT)


@pytest.mark.gpu
def test_a_damaged_block_is_staged_again_and_the_model_is_right():
    r = _run(1)
    assert r.returncode == 0 and b"PARITY-OK" in r.stdout, r.stdout + r.stderr
    assert r.stderr.count(b"did not arrive intact") == 1 and b"attempt 1 of 3" in r.stderr and b"staging it again" in r.stderr


@pytest.mark.gpu
def test_three_damaged_attempts_fail_loudly():
    r = _run(3)
    assert r.returncode == 0 and b"CREATE-FAILED" in r.stdout and b"PARITY-OK" not in r.stdout, r.stdout + r.stderr
    assert r.stderr.count(b"did not arrive intact") == 3 and b"three attempts" in r.stdout + r.stderr


@pytest.mark.gpu
def test_without_injection_nothing_is_reported():
    r = _run(0)
    assert r.returncode == 0 and b"PARITY-OK" in r.stdout and b"did not arrive intact" not in r.stderr, r.stdout + r.stderr
