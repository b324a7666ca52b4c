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
                "tk-small-prompt",
                # long contexts from the real reference: KV lengths cross the attention kernels' timestep tiles
                "tk-small-long", "tk-small-long-prompt", "tiny-hs128-long"]
# full-size TinyLlama-1.1B from the real reference, reduced to ids + top-8 + 64 probe columns + checksums per position
GOLDEN_COMPACT = ["tinyllama", "tinyllama-f16dec", "tinyllama-long"]

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


def compact_err(logits, g, n=None):
    """Per-position error of full logits [n][V] against a COMPACT golden (make_golden.py): max |diff| over the
    reference's top-8 and the 64 probe columns, and the two checksums (mean and rms of the row), all relative to
    the reference's max |logit| of that position."""
    n = len(logits) if n is None else n
    lg = np.asarray(logits[:n], np.float64)
    scale = g["absmax"][:n].astype(np.float64)
    V = lg.shape[1]
    e_top = np.max(np.abs(np.take_along_axis(lg, g["top8_idx"][:n].astype(np.int64), axis=1) - g["top8_val"][:n]), axis=1)
    e_probe = np.max(np.abs(lg[:, g["probe_idx"]] - g["probe_val"][:n]), axis=1)
    e_mean = np.abs(lg.sum(axis=1) - g["lsum"][:n]) / V
    e_rms = np.abs(np.sqrt((lg * lg).sum(axis=1)) - g["l2"][:n]) / np.sqrt(V)
    return np.maximum.reduce([e_top, e_probe, e_mean, e_rms]) / scale


def top8_elementwise(logits, ref=None, g=None, n=None):
    """Element-wise relative error on the reference's EIGHT LARGEST logits of every position: max |got - ref| / |ref| over them
    (round-5 verdict, weak 1c: the max-norm of rel_err / compact_err would not notice 1e-4 * max|logit| on one small logit; the
    top-8 are what a sampler or an argmax reads, and an element-wise ratio is meaningful there -- none of them is near zero).
    `ref`: the reference's full logits [n][V]; or `g`: a compact golden (top8_idx / top8_val)."""
    lg = np.asarray(logits, np.float64)
    lg = lg.reshape(-1, lg.shape[-1])
    n = len(lg) if n is None else n
    if g is not None:
        idx, val = g["top8_idx"][:n].astype(np.int64), g["top8_val"][:n].astype(np.float64)
    else:
        rf = np.asarray(ref, np.float64).reshape(-1, lg.shape[-1])[:n]
        idx = np.argsort(-rf, axis=1, kind="stable")[:, :8]
        val = np.take_along_axis(rf, idx, axis=1)
    got = np.take_along_axis(lg[:n], idx, axis=1)
    return np.max(np.abs(got - val) / np.maximum(np.abs(val), 1e-30), axis=1)


def safe_positions(g, n=None):
    """positions whose reference top-1 margin is far above the parity tolerance: argmax must agree there"""
    n = len(g["tokens"]) if n is None else n
    scale = g["absmax"][:n] if "absmax" in g else np.abs(g["logits"][:n]).max(axis=1)
    return g["top1_margin"][:n] > 4 * REL_TOL * scale


@pytest.fixture(scope="session")
def gguf():
    from llm_f90_amd.tools import gguf as g
    return g
