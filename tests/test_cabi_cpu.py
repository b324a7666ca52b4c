"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/llmk.h declares, and refuses to run without a device (no CPU fallback). No compute."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from llm_f90_amd import llmk


@pytest.fixture(scope="module")
def lib():
    llmk.build_lib()
    return llmk.lib()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "llmk.h")).read()
    declared = sorted(set(re.findall(r"\b(llmk_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(llmk.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_header_constants_match_binding():
    hdr = open(os.path.join(ROOT, "include", "llmk.h")).read()
    ids = dict(re.findall(r"#define LLMK_(TOKEN_EMBEDDING_TABLE|RMS_ATT_WEIGHT|RMS_FFN_WEIGHT|WQKV|WO|W13|W2|"
                          r"RMS_FINAL_WEIGHT|WCLS)\s+(\d+)", hdr))
    assert {k.lower(): int(v) for k, v in ids.items()} == llmk.TENSOR_IDS


def test_version_and_strerror(lib):
    assert lib.llmk_version() >= 100
    assert b"device" in lib.llmk_strerror(6)
    assert lib.llmk_strerror(0) == b"ok"


def test_bad_arguments_are_rejected_without_touching_a_device(lib):
    h = C.c_void_p()
    assert lib.llmk_create(None, C.byref(h)) == 1
    bad = llmk.Config(100, 256, 2, 8, 2, 300, 64, 0, 0, 0)      # emb_dim not divisible by heads
    assert lib.llmk_create(C.byref(bad), C.byref(h)) == 2
    bad = llmk.Config(128, 256, 2, 8, 2, 300, 64, 7, 0, 0)      # unknown weight type
    assert lib.llmk_create(C.byref(bad), C.byref(h)) == 4
    assert lib.llmk_forward(None, 1, 1, None) == 1
    assert lib.llmk_destroy(None) == 1


def test_no_cpu_fallback(lib):
    """On a box without a GPU create must fail loudly (LLMK_E_NODEVICE or a HIP error)."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    cfg = llmk.Config(128, 256, 2, 8, 2, 300, 64, 0, 0, 0)
    h = C.c_void_p()
    rc = lib.llmk_create(C.byref(cfg), C.byref(h))
    assert rc != 0 and not h.value


def test_tk_shapes_lists_the_instantiated_shapes(lib):
    """llmk_tk_shapes: which model shapes the persistent kernel exists for in this build (no device needed): BASELINE.json's three
    single-GPU configurations, the stock-file q6_K variants and Llama-2-7B f16 (round 6) must be there; a short buffer is an error."""
    shapes = llmk.tk_shapes()
    for want in [(2048, 5632, 32, 4, 32000, "f32"), (2048, 5632, 32, 4, 32000, "f16"), (4096, 11008, 32, 32, 32000, "q4_0"),
                 (4096, 11008, 32, 32, 32000, "q4_0+q6_K"), (2048, 5632, 32, 4, 32000, "q4_0+q6_K"), (4096, 11008, 32, 32, 32000, "f16"),
                 (4096, 14336, 32, 8, 32000, "q4_0+q6_K")]:
        assert want in shapes, (want, shapes)
    small = C.create_string_buffer(8)
    assert lib.llmk_tk_shapes(small, 8) != 0
