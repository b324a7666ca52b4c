"""The experimental q4_0 instantiation of the persistent kernel with its dots on the matrix core (csrc/q4_mfma.h,
`make -C llm.f90_amd q4m`; measured slower than the product path on MI355X): when the variant library has
been built, the Llama-2-7B parity tests -- column geometry against the oracle, full depth against the real reference's golden,
the full-shape properties -- must hold on it too.  Skipped otherwise (the product build does not contain it)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "llm.f90_amd", "csrc", "variants", "libllmk_q4m.so")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(VARIANT), reason="variant library not built (make -C llm.f90_amd q4m)")
def test_llama2_7b_parity_on_the_matrix_core_instantiation():
    env = dict(os.environ, LLMK_LIB=VARIANT)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_parity_gpu.py"), "-m", "gpu", "-x", "-q",
                        "-k", "llama2_7b"], capture_output=True, env=env, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-2000:]).decode(errors="replace")
