"""The Fortran host (llm.f90_amd/host) on CPU: its GGUF loader against an independent python reading
of the same file (f32, f16, q4_0; all GGUF KV value types), its CLI error behaviour, and that the
`llm` binary fails loudly without a GPU instead of computing anything on the CPU."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

PKG = os.path.join(ROOT, "llm.f90_amd")
FC = "/opt/rocm/bin/amdflang"
pytestmark = pytest.mark.skipif(not os.path.exists(FC), reason="amdflang not installed")


@pytest.fixture(scope="module")
def tools(tmp_path_factory):
    subprocess.run(["make", "-s", "-C", PKG], check=True)
    d = tmp_path_factory.mktemp("hosttools")
    exe = str(d / "loader_dump")
    src = [os.path.join(PKG, "host", f) for f in ("llm_types.f90", "gguf_loader.f90")]
    subprocess.run([FC, "-O1", *src, os.path.join(ROOT, "tests", "host_tools", "loader_dump.f90"), "-o", exe],
                   check=True, cwd=str(d), capture_output=True)
    return {"dump": exe, "llm": os.path.join(PKG, "host", "llm"), "dir": d}


def _read_dump(path, gguf):
    raw = open(path, "rb").read()
    o = 0

    def take(dt, n):
        nonlocal o
        a = np.frombuffer(raw, dtype=dt, count=n, offset=o)
        o += a.nbytes
        return a
    cfg = take("<i4", 8)
    wtype, width = take("<i4", 2)
    E, H, L, nh, nkv, V, S, KV = [int(x) for x in cfg]
    out = {"cfg": cfg, "wtype": int(wtype), "tl": take("<i4", V), "scores": take("<f4", V)}
    out["vocab"] = [bytes(take("u1", width)) for _ in range(V)]
    out["token_embedding_table"] = take("<f4", V * E).reshape(V, E)
    out["rms_att_weight"] = take("<f4", L * E).reshape(L, E)
    out["rms_ffn_weight"] = take("<f4", L * E).reshape(L, E)
    out["rms_final_weight"] = take("<f4", E)
    rb = lambda k: {0: 4 * k, 1: 2 * k, 2: k // 32 * 18}[int(wtype)]
    for name, rows, k in (("wqkv", L * (E + 2 * KV), E), ("wo", L * E, E), ("w13", L * 2 * H, E), ("w2", L * E, H),
                          ("wcls", V, E)):
        out[name] = take("u1", rows * rb(k)).reshape(rows, rb(k))
    assert o == len(raw)
    return out


@pytest.mark.parametrize("wtype", [0, 1, 2], ids=["f32", "f16", "q4_0"])
@pytest.mark.parametrize("shape", ["tiny-gqa", "tiny-hs64"])
def test_fortran_loader_matches_python_reader(tools, gguf, shape, wtype):
    d = tools["dir"]
    path = str(d / f"{shape}-{wtype}.gguf")
    gguf.write_synth_gguf(path, gguf.SHAPES[shape], 31337, wtype)
    out = str(d / "dump.bin")
    r = subprocess.run([tools["dump"], path, out], capture_output=True, check=True)
    assert b"data offset" in r.stdout
    got = _read_dump(out, gguf)
    fw = gguf.load_fused(path)
    s = fw.shape
    assert list(got["cfg"]) == [s.emb_dim, s.hidden_dim, s.n_layers, s.n_heads, s.n_kv_heads, s.vocab_size, s.seq_len,
                                s.kv_dim]
    assert got["wtype"] == wtype
    for name in ("token_embedding_table", "rms_att_weight", "rms_ffn_weight", "rms_final_weight"):
        assert np.array_equal(got[name], getattr(fw, name)), name
    for name in ("wqkv", "wo", "w13", "w2", "wcls"):
        ref = np.ascontiguousarray(getattr(fw, name))
        assert np.array_equal(got[name].reshape(-1), ref.view(np.uint8).reshape(-1)), name
    vocab = gguf.vocab_strings(s.vocab_size)
    for i, tok in enumerate(vocab):
        assert got["tl"][i] == len(tok)
        assert got["vocab"][i][:len(tok)] == tok
    assert np.array_equal(got["scores"], -np.arange(s.vocab_size, dtype=np.float32))


def test_loader_accepts_every_gguf_kv_type_and_spm_space(tools, gguf):
    """The reference stops on bool/u8/u64/f64... KVs (read_ggml.f90:682-684); ours must not. Also the
    leading U+2581 -> ' ' rewrite (read_ggml.f90:479-497) and a non-default alignment."""
    d = tools["dir"]
    base = str(d / "base.gguf")
    s = gguf.SHAPES["tiny-gqa"]
    gguf.write_synth_gguf(base, s, 7, alignment=64)
    raw = bytearray(open(base, "rb").read())
    g = gguf.read_gguf(base)
    # splice extra KVs of every scalar type + nested/typed arrays right after the header
    def kv(key, t, payload):
        return struct.pack("<Q", len(key)) + key + struct.pack("<I", t) + payload
    extra = b"".join([
        kv(b"x.u8", 0, struct.pack("<B", 7)), kv(b"x.i8", 1, struct.pack("<b", -3)),
        kv(b"x.u16", 2, struct.pack("<H", 9)), kv(b"x.i16", 3, struct.pack("<h", -9)),
        kv(b"x.bool", 7, b"\x01"), kv(b"x.u64", 10, struct.pack("<Q", 1 << 40)),
        kv(b"x.i64", 11, struct.pack("<q", -5)), kv(b"x.f64", 12, struct.pack("<d", 2.5)),
        kv(b"x.arr_i32", 9, struct.pack("<IQ", 5, 3) + struct.pack("<3i", 1, 2, 3)),
        kv(b"x.arr_str", 9, struct.pack("<IQ", 8, 2) + struct.pack("<Q", 2) + b"ab" + struct.pack("<Q", 0)),
        kv(b"x.arr_arr", 9, struct.pack("<IQ", 9, 1) + struct.pack("<IQ", 0, 4) + b"\x01\x02\x03\x04"),
    ])
    n_kv = struct.unpack_from("<q", raw, 16)[0]
    struct.pack_into("<q", raw, 16, n_kv + 11)
    head_end = 24
    # token id 5 gets a sentencepiece space prefix: find its bytes (" " + ...) -- token 5 is chr(34)='"', 1 byte
    new = bytes(raw[:head_end]) + extra + bytes(raw[head_end:g.data_start])
    spm = new.replace(struct.pack("<Q", 7) + b"<00150>", struct.pack("<Q", 8) + "▁".encode() + b"hello", 1)
    assert spm != new
    pad = (-len(spm)) % 64
    path = str(d / "allkv.gguf")
    open(path, "wb").write(spm + b"\0" * pad + bytes(raw[g.data_start:]))
    out = str(d / "dump2.bin")
    subprocess.run([tools["dump"], path, out], capture_output=True, check=True)
    got = _read_dump(out, gguf)
    fw = gguf.synth_fused(s, 7)
    assert np.array_equal(got["wo"].reshape(-1), fw.wo.view(np.uint8).reshape(-1))
    assert got["tl"][150] == 6 and got["vocab"][150][:6] == b" hello"


def test_cli_flags_and_errors(tools, gguf):
    llm = tools["llm"]
    r = subprocess.run([llm, "--nope"], capture_output=True)
    assert b"Unrecognized option:--nope" in r.stdout        # llama2.f90:74
    r = subprocess.run([llm, "-m", "/does/not/exist.gguf"], capture_output=True)
    assert r.returncode != 0
    bad = str(tools["dir"] / "bad.gguf")
    open(bad, "wb").write(b"NOPE" + b"\0" * 64)
    r = subprocess.run([llm, "-m", bad], capture_output=True)
    assert b"Magic numbers do not match" in r.stdout         # read_ggml.f90:123


def test_llm_binary_has_no_cpu_fallback(tools, gguf):
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    path = str(tools["dir"] / "tiny.gguf")
    gguf.write_synth_gguf(path, gguf.SHAPES["tiny-gqa"], 20260928)
    r = subprocess.run([tools["llm"], "-m", path, "-n", "4"], capture_output=True)
    g = load_golden("tiny-gqa")
    assert r.stdout.split(b"\n")[0] == bytes(g["stdout"]).split(b"\n")[0]     # same " data offset" line as the reference
    assert r.returncode != 0 and b"no usable HIP device" in r.stdout


def test_ak_requires_tokenizer_and_rejects_garbage(tools, gguf):
    s = gguf.SHAPES["tiny-gqa"]
    ak = str(tools["dir"] / "m.ak")
    gguf.write_ak(ak, gguf.synth_fused(s, 5))
    r = subprocess.run([tools["llm"], "--ak", "-m", ak, "-n", "4"], capture_output=True)
    assert r.returncode != 0 and b"--ak needs a tokenizer file" in r.stdout
    bad = str(tools["dir"] / "bad.ak")
    open(bad, "wb").write(struct.pack("<7i", 0, 0, 0, 0, 0, 0, 0))
    r = subprocess.run([tools["llm"], "--ak", "-m", bad, "-s", ak], capture_output=True)
    assert r.returncode != 0 and b"not an ak checkpoint" in r.stdout
