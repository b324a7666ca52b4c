import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import llm_f90_amd  # noqa: E402,F401  (alias for the llm.f90_amd/ directory)

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["tiny-gqa", "tiny-gqa-prompt", "tiny-mha", "tiny-hs64", "tiny-hs128", "tiny-70bish", "tk-small",
                "tk-small-prompt"]

# Parity bar (BASELINE.json north_star): logits within 1e-4 relative of the reference CPU path,
# bit-exact argmax at temperature 0.  "Relative" is measured against the logit scale
# max|reference logits| of that position (an element-wise ratio is meaningless for logits that
# cross zero).
REL_TOL = 1e-4


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    return {k: z[k] for k in z.files}


def rel_err(got, ref):
    """max |got-ref| / max |ref| per position (rows)."""
    got = np.asarray(got, np.float64).reshape(-1, ref.shape[-1])
    ref = np.asarray(ref, np.float64).reshape(-1, ref.shape[-1])
    return np.max(np.abs(got - ref), axis=1) / np.max(np.abs(ref), axis=1)


@pytest.fixture(scope="session")
def gguf():
    from llm_f90_amd.tools import gguf as g
    return g
