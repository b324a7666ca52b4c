"""ctypes binding of the llmk C-ABI (include/llmk.h) -- the same symbols the Fortran host binds
with ISO_C_BINDING (host/llmk_binding.f90).  Used by tests/ and bench.py.

There is no fallback: if libllmk.so is missing or no HIP device is usable this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LLMK_LIB selects another build of the same ABI (csrc/libllmk_debug.so: the persistent kernel's timing aids)
LIB_PATH = os.environ.get("LLMK_LIB") or os.path.join(_HERE, "csrc", "libllmk.so")

TENSOR_IDS = {
    "token_embedding_table": 0, "rms_att_weight": 1, "rms_ffn_weight": 2, "wqkv": 3, "wo": 4, "w13": 5, "w2": 6,
    "rms_final_weight": 7, "wcls": 8,
}
FLAG_NO_GRAPH, FLAG_TIMINGS, FLAG_MULTI_KERNEL = 1, 2, 4
# every symbol include/llmk.h declares
SYMBOLS = ["llmk_create", "llmk_create_tp", "llmk_tp_unique_id", "llmk_tp_init_comm", "llmk_tp_p2p_handle", "llmk_tp_p2p_connect",
           "llmk_tp_p2p_connect_local", "llmk_tp_p2p_selftest", "llmk_tp_p2p_stress", "llmk_tp_p2p_disable", "llmk_tp_begin", "llmk_tp_segment",
           "llmk_tp_read_partial", "llmk_tp_write_partial", "llmk_tp_read_logits", "llmk_upload", "llmk_upload_rows",
           "llmk_set_rope_freqs", "llmk_set_tensor_type", "llmk_set_rms_eps", "llmk_forward", "llmk_prefill", "llmk_forward_greedy", "llmk_decode_greedy", "llmk_reset", "llmk_timings",
           "llmk_time_kernel", "llmk_peek", "llmk_tensor_checksum", "llmk_path", "llmk_tk_shapes", "llmk_tp_ranks_seen", "llmk_destroy", "llmk_strerror", "llmk_version"]
PATH_NAMES = {0: "multi-kernel (5 launches per layer)", 1: "persistent whole-token kernel",
              2: "tensor-parallel rank: 6 launches per layer + one-shot peer-memory exchanges",
              3: "tensor-parallel rank: eager launches + RCCL collectives", 4: "tensor-parallel rank, collectives not connected"}


TOKEN_FN = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_void_p)


class LlmkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"llmk error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("emb_dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size",
                                         "seq_len", "weight_type", "device", "flags")]


def build_lib(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> csrc/libllmk.so (cross-compiles without a GPU)."""
    subprocess.run(["make", "-s", "-C", _HERE, "lib"] + (["-B"] if force else []), check=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built (run `make -C llm.f90_amd lib`); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, ci, cf = C.c_void_p, C.c_int, C.POINTER(C.c_float)
        L.llmk_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        L.llmk_create_tp.argtypes = [C.POINTER(Config), ci, ci, C.POINTER(vp)]
        L.llmk_tp_unique_id.argtypes = [C.c_char_p]
        L.llmk_tp_init_comm.argtypes = [vp, C.c_char_p]
        L.llmk_tp_p2p_handle.argtypes = [vp, C.c_char_p]
        L.llmk_tp_p2p_connect.argtypes = [vp, C.c_char_p]
        L.llmk_tp_p2p_connect_local.argtypes = [vp, C.POINTER(vp)]
        L.llmk_tp_p2p_selftest.argtypes = [vp, ci]
        L.llmk_tp_p2p_stress.argtypes = [vp, ci, C.c_uint]
        L.llmk_tp_p2p_disable.argtypes = [vp]
        L.llmk_tensor_checksum.argtypes = [vp, ci, C.POINTER(C.c_ulonglong)]
        L.llmk_tp_begin.argtypes = [vp, ci, ci]
        L.llmk_tp_segment.argtypes = [vp, ci, ci]
        L.llmk_tp_read_partial.argtypes = [vp, cf]
        L.llmk_tp_write_partial.argtypes = [vp, cf]
        L.llmk_tp_read_logits.argtypes = [vp, cf]
        L.llmk_upload.argtypes = [vp, ci, vp, C.c_size_t, ci]
        L.llmk_upload_rows.argtypes = [vp, ci, ci, ci, ci, vp, C.c_size_t, ci]
        L.llmk_set_rope_freqs.argtypes = [vp, cf, ci]
        L.llmk_set_tensor_type.argtypes = [vp, ci, ci]
        L.llmk_set_rms_eps.argtypes = [vp, C.c_float]
        L.llmk_forward.argtypes = [vp, ci, ci, cf]
        L.llmk_prefill.argtypes = [vp, C.POINTER(ci), ci, ci, cf]
        L.llmk_forward_greedy.argtypes = [vp, ci, ci, C.POINTER(ci)]
        L.llmk_decode_greedy.argtypes = [vp, ci, ci, ci, C.POINTER(ci), vp, vp]
        L.llmk_reset.argtypes = [vp]
        L.llmk_timings.argtypes = [vp, cf]
        L.llmk_time_kernel.argtypes = [vp, ci, ci, cf, C.POINTER(C.c_double)]
        L.llmk_peek.argtypes = [vp, ci, ci, ci, cf, ci]
        L.llmk_destroy.argtypes = [vp]
        L.llmk_path.argtypes = [vp]
        if hasattr(L, "llmk_tk_shapes"):          # (absent from an older build selected with LLMK_LIB for an A/B)
            L.llmk_tk_shapes.argtypes = [C.c_char_p, C.c_size_t]
        L.llmk_tp_ranks_seen.argtypes = [vp]
        L.llmk_strerror.argtypes = [ci]
        L.llmk_strerror.restype = C.c_char_p
        L.llmk_version.argtypes = []
        for s in SYMBOLS:
            if s != "llmk_strerror" and (hasattr(L, s) or not os.environ.get("LLMK_LIB")):
                getattr(L, s).restype = ci
        _lib = L
    return _lib


def _ck(rc):
    if rc != 0:
        raise LlmkError(rc, lib().llmk_strerror(rc).decode())


def tk_shapes():
    """[(E, H, NH, NKV, V, "f32" | "f16" | "q4_0" | "q4_0+q6_K"), ...]: the shapes the persistent kernel is built for (llmk_tk_shapes)"""
    buf = C.create_string_buffer(8192)
    _ck(lib().llmk_tk_shapes(buf, len(buf)))
    out = []
    for item in buf.value.decode().split(";"):
        f = item.split(",")
        out.append(tuple(int(x) for x in f[:5]) + (f[5],))
    return out


class Llmk:
    """One sequence on one GPU. `fw` is tools.gguf.FusedWeights (weight_module layout)."""

    def __init__(self, fw, device: int = 0, flags: int = 0, seq_len: int | None = None, tp_rank: int = 0,
                 tp_size: int = 1):
        s = fw.shape
        self.shape = s
        self.V = s.vocab_size
        self.tp_rank, self.tp_size = tp_rank, tp_size
        cfg = Config(s.emb_dim, s.hidden_dim, s.n_layers, s.n_heads, s.n_kv_heads, s.vocab_size,
                     seq_len or s.seq_len, fw.ggml_type, device, flags)
        self._h = C.c_void_p()
        _ck(lib().llmk_create_tp(C.byref(cfg), tp_rank, tp_size, C.byref(self._h)))
        cls_type = getattr(fw, "cls_type", fw.ggml_type)
        if cls_type != fw.ggml_type:      # e.g. a q6_K output.weight dequantised by the loader
            _ck(lib().llmk_set_tensor_type(self._h, TENSOR_IDS["wcls"], cls_type))
        for name, tid in TENSOR_IDS.items():
            a = np.ascontiguousarray(getattr(fw, name))
            is_mat = name in ("wqkv", "wo", "w13", "w2", "wcls")
            _ck(lib().llmk_upload(self._h, tid, a.ctypes.data, a.nbytes,
                                  (cls_type if name == "wcls" else fw.ggml_type) if is_mat else 0))
        self._finish_init()

    def _finish_init(self):
        # the reference's own f32 expression for the RoPE frequencies (llama2.f90:544-545)
        hs = self.shape.head_size
        fr = np.float32(1.0) / np.power(np.float32(10000.0), (np.arange(1, hs, 2, dtype=np.float32) / np.float32(hs)),
                                        dtype=np.float32)
        self.set_rope_freqs(fr)
        self._logits = np.empty(self.V, np.float32)

    @classmethod
    def create_empty(cls, shape, ggml_type: int, device: int = 0, flags: int = 0, tp_rank: int = 0, tp_size: int = 1,
                     seq_len: int | None = None) -> "Llmk":
        """A context WITHOUT weights: the caller streams them in with upload_rows (bench.build_streamed, the loader tests)."""
        m = cls.__new__(cls)
        m.shape, m.V, m.tp_rank, m.tp_size = shape, shape.vocab_size, tp_rank, tp_size
        cfg = Config(shape.emb_dim, shape.hidden_dim, shape.n_layers, shape.n_heads, shape.n_kv_heads, shape.vocab_size,
                     seq_len or shape.seq_len, ggml_type, device, flags)
        m._h = C.c_void_p()
        _ck(lib().llmk_create_tp(C.byref(cfg), tp_rank, tp_size, C.byref(m._h)))
        m._finish_init()
        return m

    def upload_rows(self, name: str, layer: int, row0: int, arr, ggml_type: int):
        """rows row0.. of layer `layer` of the FULL tensor `name` (a tensor-parallel ctx keeps what its shard holds)"""
        arr = np.ascontiguousarray(arr)
        _ck(lib().llmk_upload_rows(self._h, TENSOR_IDS[name], layer, row0, arr.shape[0] if arr.ndim > 1 else 1, arr.ctypes.data,
                                   arr.nbytes, ggml_type))

    def set_rope_freqs(self, fr):
        fr = np.ascontiguousarray(fr, np.float32)
        _ck(lib().llmk_set_rope_freqs(self._h, fr.ctypes.data_as(C.POINTER(C.c_float)), len(fr)))

    def set_tensor_type(self, name: str, ggml_type: int):
        """give a tensor (the classifier) a ggml type of its own BEFORE it is uploaded (llmk_set_tensor_type)"""
        _ck(lib().llmk_set_tensor_type(self._h, TENSOR_IDS[name], ggml_type))

    def set_rms_eps(self, eps: float):
        _ck(lib().llmk_set_rms_eps(self._h, eps))

    def forward(self, token: int, pos: int) -> np.ndarray:
        """1-based token and pos, as `transformer(token,pos,s,weights)` (llama2.f90:380)."""
        _ck(lib().llmk_forward(self._h, token, pos, self._logits.ctypes.data_as(C.POINTER(C.c_float))))
        return self._logits.copy()

    def forward_raw(self, token: int, pos: int) -> int:
        """Hot loop for bench.py: no copy of the result, returns the status code."""
        return lib().llmk_forward(self._h, token, pos, self._logits.ctypes.data_as(C.POINTER(C.c_float)))

    def prefill(self, tokens, pos0: int = 1) -> np.ndarray:
        """tokens (1-based ids) at positions pos0.. in one call; logits of the last position (llama2.f90:376-402)."""
        t = np.ascontiguousarray(tokens, np.int32)
        _ck(lib().llmk_prefill(self._h, t.ctypes.data_as(C.POINTER(C.c_int)), len(t), pos0,
                               self._logits.ctypes.data_as(C.POINTER(C.c_float))))
        return self._logits.copy()

    def forward_greedy(self, token: int, pos: int) -> int:
        nxt = C.c_int(0)
        _ck(lib().llmk_forward_greedy(self._h, token, pos, C.byref(nxt)))
        return nxt.value

    def decode_greedy(self, token: int, pos0: int, n: int, on_token=None) -> np.ndarray:
        """n positions from pos0 at temperature 0 with the argmax on the device (llmk_decode_greedy); returns the n ids."""
        ids = np.zeros(n, np.int32)
        cb = TOKEN_FN(on_token) if on_token else None
        _ck(lib().llmk_decode_greedy(self._h, token, pos0, n, ids.ctypes.data_as(C.POINTER(C.c_int)),
                                     C.cast(cb, C.c_void_p) if cb else None, None))
        return ids

    def generate(self, n: int, prompt=(), want_logits: bool = True, greedy_on_device: bool = False):
        """The reference generation loop at temperature 0 (llama2.f90:376-402)."""
        self.reset()
        toks = np.zeros(n, np.int32)
        logits = np.empty((n, self.V), np.float32) if want_logits else None
        token = 2
        p = list(prompt)
        for pos in range(1, n + 1):
            if greedy_on_device:
                nxt = self.forward_greedy(token, pos)
            else:
                lg = self.forward(token, pos)
                if want_logits:
                    logits[pos - 1] = lg
                nxt = int(np.argmax(lg)) + 1
            token = p[pos - 1] if pos <= len(p) else nxt
            toks[pos - 1] = token
        return toks, logits

    # ---- tensor parallel ------------------------------------------------------------------------
    @staticmethod
    def tp_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _ck(lib().llmk_tp_unique_id(buf))
        return buf.raw

    def tp_init_comm(self, uid: bytes):
        _ck(lib().llmk_tp_init_comm(self._h, C.create_string_buffer(uid, 128)))

    def tp_p2p_handle(self) -> bytes:
        buf = C.create_string_buffer(64)
        _ck(lib().llmk_tp_p2p_handle(self._h, buf))
        return buf.raw

    def tp_p2p_connect(self, handles):
        """handles: the tp_size 64-byte inbox handles in rank order (other PROCESSES' ranks)"""
        _ck(lib().llmk_tp_p2p_connect(self._h, C.create_string_buffer(b"".join(handles), 64 * len(handles))))

    @staticmethod
    def tp_p2p_connect_local(ranks):
        """ranks: the tp_size Llmk objects of ONE process, in rank order"""
        arr = (C.c_void_p * len(ranks))(*[m._h for m in ranks])
        for m in ranks:
            _ck(lib().llmk_tp_p2p_connect_local(m._h, arr))

    def tp_p2p_selftest(self, iters: int = 64) -> int:
        """0 when this rank's peer-memory exchanges all gave exact sums (every rank must call it); else the LLMK_E_* code"""
        return lib().llmk_tp_p2p_selftest(self._h, iters)

    def tp_p2p_stress(self, iters: int, seed: int) -> int:
        """the self-test's rounds with random delays around every send and read (llmk_tp_p2p_stress); 0 = all sums exact"""
        return lib().llmk_tp_p2p_stress(self._h, iters, seed)

    def tensor_checksum(self, name: str) -> int:
        """64-bit word sum of the tensor's device image (llmk_tensor_checksum)"""
        out = C.c_ulonglong(0)
        _ck(lib().llmk_tensor_checksum(self._h, TENSOR_IDS[name], C.byref(out)))
        return out.value

    def tp_p2p_disable(self):
        _ck(lib().llmk_tp_p2p_disable(self._h))

    def tp_begin(self, token: int, pos: int):
        _ck(lib().llmk_tp_begin(self._h, token, pos))

    def tp_segment(self, seg: int, layer: int = 0):
        _ck(lib().llmk_tp_segment(self._h, seg, layer))

    def tp_read_partial(self) -> np.ndarray:
        out = np.empty(self.shape.emb_dim, np.float32)
        _ck(lib().llmk_tp_read_partial(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def tp_write_partial(self, v: np.ndarray):
        v = np.ascontiguousarray(v, np.float32)
        _ck(lib().llmk_tp_write_partial(self._h, v.ctypes.data_as(C.POINTER(C.c_float))))

    def tp_read_logits(self) -> np.ndarray:
        out = np.empty(self.V // self.tp_size, np.float32)
        _ck(lib().llmk_tp_read_logits(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def reset(self):
        _ck(lib().llmk_reset(self._h))

    def timings(self):
        t = (C.c_float * 5)()
        _ck(lib().llmk_timings(self._h, t))
        return list(t)

    def time_kernel(self, kernel: int, iters: int):
        ms, b = C.c_float(0), C.c_double(0)
        _ck(lib().llmk_time_kernel(self._h, kernel, iters, C.byref(ms), C.byref(b)))
        return ms.value, b.value

    def peek(self, which: int, n: int, layer: int = 0, pos: int = 1):
        out = np.empty(n, np.float32)
        _ck(lib().llmk_peek(self._h, which, layer, pos, out.ctypes.data_as(C.POINTER(C.c_float)), n))
        return out

    def path(self) -> int:
        return lib().llmk_path(self._h)

    def tp_ranks_seen(self) -> int:
        return lib().llmk_tp_ranks_seen(self._h)

    def path_name(self) -> str:
        return PATH_NAMES.get(self.path(), "?")

    def close(self):
        if self._h:
            lib().llmk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
