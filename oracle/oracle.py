"""TEST INFRASTRUCTURE -- ctypes loader for oracle/llm_oracle.c (the CPU restatement of the
reference forward pass, /root/reference/llama2.f90:480-640).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product path (libllmk.so, the Fortran host) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = C.POINTER(C.c_float)


class _Model(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("emb_dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "seq_len")] + \
               [(n, _F) for n in ("token_embedding_table", "rms_att_weight", "rms_ffn_weight", "wqkv", "wo", "w13",
                                  "w2", "rms_final_weight", "wcls", "key_cache", "value_cache")]


def build(force: bool = False) -> None:
    """Compile the C restatement (and oracle/_ref when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", _HERE] + (["-B"] if force else []), check=True)


def _lib(flavour: str):
    name = {"strict": "liboracle.so", "omp": "liboracle_omp.so", "fast": "liboracle_fast.so"}[flavour]
    path = os.path.join(_HERE, "_build", name)
    if flavour == "omp":  # many-core hosts: do not oversubscribe small shapes with 256 threads
        os.environ.setdefault("OMP_NUM_THREADS", str(min(32, os.cpu_count() or 1)))
    if not os.path.exists(path):
        subprocess.run(["make", "-s", "-C", _HERE, os.path.join("_build", name)], check=True)
    lib = C.CDLL(path)
    lib.oracle_forward.argtypes = [C.POINTER(_Model), C.c_int, C.c_int, _F, _F]
    lib.oracle_forward.restype = C.c_int
    lib.oracle_generate.argtypes = [C.POINTER(_Model), C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_int), _F]
    lib.oracle_generate.restype = C.c_int
    lib.oracle_set_eps.argtypes = [C.c_float]
    lib.oracle_set_eps.restype = None
    return lib


class Oracle:
    """Holds f32 fused weights (tools.gguf.FusedWeights, matrices already decoded to f32) and a
    KV cache; `forward(token, pos)` uses the reference's 1-based conventions."""

    def __init__(self, fw, flavour: str = "strict"):
        assert fw.ggml_type == 0, "oracle consumes f32 weights: pass fw.as_f32()"
        self.lib = _lib(flavour)
        self.shape = s = fw.shape
        self._keep = []
        m = _Model(s.emb_dim, s.hidden_dim, s.n_layers, s.n_heads, s.n_kv_heads, s.vocab_size, s.seq_len)
        for name in ("token_embedding_table", "rms_att_weight", "rms_ffn_weight", "wqkv", "wo", "w13", "w2",
                     "rms_final_weight", "wcls"):
            a = np.ascontiguousarray(getattr(fw, name), dtype=np.float32)
            self._keep.append(a)
            setattr(m, name, a.ctypes.data_as(_F))
        self.key_cache = np.zeros((s.n_layers, s.seq_len, s.kv_dim), np.float32)
        self.value_cache = np.zeros_like(self.key_cache)
        m.key_cache = self.key_cache.ctypes.data_as(_F)
        m.value_cache = self.value_cache.ctypes.data_as(_F)
        self.m = m

    def set_eps(self, eps: float):
        """rmsnorm epsilon other than the reference's 1e-5 (process-wide; only for the opt-in extension's test)"""
        self.lib.oracle_set_eps(eps)

    def reset(self):
        self.key_cache[:] = 0
        self.value_cache[:] = 0

    def forward(self, token: int, pos: int, trace: bool = False):
        s = self.shape
        logits = np.empty(s.vocab_size, np.float32)
        tr = np.empty((s.n_layers + 1, s.emb_dim), np.float32) if trace else None
        rc = self.lib.oracle_forward(C.byref(self.m), token, pos, logits.ctypes.data_as(_F),
                                     tr.ctypes.data_as(_F) if trace else None)
        if rc:
            raise ValueError(f"oracle_forward rc={rc}")
        return (logits, tr) if trace else logits

    def generate(self, n: int, prompt=(), want_logits: bool = True):
        """Reference generation loop at temperature 0 (llama2.f90:376-402). Returns (tokens 1-based, logits[n,V])."""
        s = self.shape
        self.reset()
        toks = np.zeros(n, np.int32)
        logits = np.empty((n, s.vocab_size), np.float32) if want_logits else None
        p = np.asarray(list(prompt), np.int32)
        rc = self.lib.oracle_generate(C.byref(self.m), p.ctypes.data_as(C.POINTER(C.c_int)), len(p), n,
                                      toks.ctypes.data_as(C.POINTER(C.c_int)),
                                      logits.ctypes.data_as(_F) if want_logits else None)
        if rc:
            raise ValueError(f"oracle_generate rc={rc}")
        return toks, logits


def bpe_encode_ref(text: bytes, vocab, scores):
    """TEST INFRASTRUCTURE -- the reference's tokenizer restated (llama2.f90:643-724), 1-based ids.

    lookup (:643-655) is a linear scan that returns the FIRST entry with the same bytes and length;
    bpe_encode starts from one token per byte (:666-668), then repeatedly replaces the adjacent pair whose
    concatenation is the best-scoring vocabulary entry -- `score > best_score` from -1e10, so among equal
    scores the FIRST pair wins (:689-700) -- until no pair is in the vocabulary (:703-705).
    Pinned by tests/golden/make_golden.py: the real reference fed these ids must print this text AND the
    oracle teacher-forced with them must reproduce the reference's logits (tests/test_oracle.py)."""
    first = {}
    for i, t in enumerate(vocab):
        first.setdefault(bytes(t), i + 1)
    toks = [first[text[i:i + 1]] for i in range(len(text))]     # (a byte without a token: the reference indexes vocab_len(-1))
    while True:
        best_score, best_i, best_tok = -1e10, -1, -1
        for i in range(len(toks) - 1):
            cand = first.get(bytes(vocab[toks[i] - 1]) + bytes(vocab[toks[i + 1] - 1]))
            if cand is not None and float(scores[cand - 1]) > best_score:
                best_score, best_i, best_tok = float(scores[cand - 1]), i, cand
        if best_i < 0:
            return toks
        toks[best_i:best_i + 2] = [best_tok]
