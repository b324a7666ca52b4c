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
    wtype, width, wcls_type = take("<i4", 3)
    eps, rope_base = take("<f4", 2)
    E, H, L, nh, nkv, V, S, KV = [int(x) for x in cfg]
    out = {"cfg": cfg, "wtype": int(wtype), "wcls_type": int(wcls_type), "rms_eps": float(eps), "rope_freq_base": float(rope_base),
           "tl": take("<i4", V), "scores": take("<f4", V)}
    out["vocab"] = [bytes(take("u1", width)) for _ in range(V)]
    out["token_embedding_table"] = take("<f4", V * E).reshape(V, E)
    out["rms_att_weight"] = take("<f4", L * E).reshape(L, E)
    out["rms_ffn_weight"] = take("<f4", L * E).reshape(L, E)
    out["rms_final_weight"] = take("<f4", E)
    rbt = lambda t, k: {0: 4 * k, 1: 2 * k, 2: k // 32 * 18, 14: k // 256 * 210}[int(t)]
    for name, rows, k, t in (("wqkv", L * (E + 2 * KV), E, wtype), ("wo", L * E, E, wtype), ("w13", L * 2 * H, E, wtype),
                             ("w2", L * E, H, wtype), ("wcls", V, E, wcls_type)):
        out[name] = take("u1", rows * rbt(t, k)).reshape(rows, rbt(t, k))
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


def test_fortran_loader_hands_a_q6k_output_weight_over_raw_or_dequantised_and_keeps_the_file_s_eps_and_rope_base(tools, gguf):
    """A stock llama.cpp q4_0 file keeps output.weight in q6_K (the reference stops on it, read_ggml.f90:633-635).  Round 6: the
    loader hands the super-blocks over AS THEY LIE IN THE FILE (wcls_type 14: the device dots them, csrc/q6k.h); under
    LLM_DEQUANT_CLS=1 -- the round-2 behaviour -- as f32, decoded exactly as ggml's dequantize_row_q6_K does (the python mirror in
    tools/gguf.py is checked against a scalar transcription of that loop).  It also records the file's rmsnorm epsilon and RoPE
    base for hosts that opt in (the reference ignores both)."""
    d = tools["dir"]
    s = gguf.LlamaShape(256, 512, 2, 4, 2, 320, 32)      # E a multiple of 256: q6_K super-blocks
    fw = gguf.synth_fused(s, 99, 2)
    path = str(d / "q6k.gguf")
    gguf.write_gguf(path, fw, rms_eps=1e-6, rope_freq_base=500000.0, output_q6k=True)
    out = str(d / "dump_q6k.bin")
    for dequant in (False, True):
        env = dict(os.environ, LLM_DEQUANT_CLS="1") if dequant else {k: v for k, v in os.environ.items() if k != "LLM_DEQUANT_CLS"}
        subprocess.run([tools["dump"], path, out], capture_output=True, check=True, env=env)
        got = _read_dump(out, gguf)
        ref = gguf.load_fused(path, dequant_cls=dequant)
        want = 0 if dequant else 14
        assert got["wtype"] == 2 and got["wcls_type"] == want and ref.cls_type == want
        assert np.array_equal(got["wcls"].reshape(-1), np.ascontiguousarray(ref.wcls).view(np.uint8).reshape(-1))
        assert np.array_equal(got["wqkv"].reshape(-1), ref.wqkv.view(np.uint8).reshape(-1))
        assert got["rms_eps"] == np.float32(1e-6) and got["rope_freq_base"] == np.float32(500000.0)
    # the decoded classifier is the q6_K rounding of the q4_0 one: close, not equal; raw and dequantised forms decode alike
    full = gguf.decode(fw.wcls, 2, s.emb_dim)
    assert 0 < np.abs(ref.wcls - full).max() < 0.05 * np.abs(full).max()
    raw = gguf.load_fused(path)
    assert np.array_equal(gguf.decode(raw.wcls, 14, s.emb_dim).reshape(ref.wcls.shape), ref.wcls)


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


def test_converter_cli_replaces_load_f90(tools, gguf, tmp_path):
    """llm.f90_amd/tools/convert.py: GGUF -> "ak" + tokenizer.bin (what the reference's broken load.f90 was for) and
    GGUF -> q4_0 / f16 GGUF; byte-exact against the library writers, and the q4_0 file loads through the Fortran loader."""
    import sys
    s = gguf.SHAPES["tiny-hs64"]
    src = str(tmp_path / "src.gguf")
    gguf.write_synth_gguf(src, s, 42)
    conv = os.path.join(PKG, "tools", "convert.py")
    ak, tok, q4 = str(tmp_path / "m.ak"), str(tmp_path / "tok.bin"), str(tmp_path / "q4.gguf")
    r = subprocess.run([sys.executable, conv, src, "--ak", ak, "--tokenizer", tok, "--gguf", q4, "--type", "q4_0"], capture_output=True)
    assert r.returncode == 0, r.stdout + r.stderr
    ref_ak, ref_tok = str(tmp_path / "ref.ak"), str(tmp_path / "ref_tok.bin")
    fw = gguf.synth_fused(s, 42)
    gguf.write_ak(ref_ak, fw)
    gguf.write_tokenizer_bin(ref_tok, gguf.vocab_strings(s.vocab_size), -np.arange(s.vocab_size, dtype=np.float32))   # the file's scores
    assert open(ak, "rb").read() == open(ref_ak, "rb").read()
    assert open(tok, "rb").read() == open(ref_tok, "rb").read()
    got = gguf.load_fused(q4)
    assert got.ggml_type == 2 and np.array_equal(got.w13, gguf.synth_fused(s, 42, 2).w13)
    out = str(tmp_path / "dump_q4.bin")
    subprocess.run([tools["dump"], q4, out], capture_output=True, check=True)
    d = _read_dump(out, gguf)
    assert d["wtype"] == 2 and np.array_equal(d["wo"].reshape(-1), got.wo.view(np.uint8).reshape(-1))
    r = subprocess.run([sys.executable, conv, src], capture_output=True)
    assert r.returncode != 0 and b"nothing to do" in r.stderr


@pytest.mark.parametrize("ak", [False, True], ids=["gguf", "ak"])
def test_verbose_loader_lines_are_the_reference_s(tools, gguf, ak):
    """`-v` (round-3 verdict, "missing" 4): everything the reference prints while it loads -- header, every key / value
    pair at the reference's 64-character width, position / deficit / data offset, the dims, the twelve "loaded ..." lines in
    the reference's order with the reference's counts, the tokenizer lines, "Loaded weights" -- byte for byte
    (tests/golden/tiny-gqa[-ak]-verbose.npz = the real reference's stdout).  No GPU needed up to that line."""
    g = load_golden("tiny-gqa-ak-verbose" if ak else "tiny-gqa-verbose")
    s = gguf.SHAPES["tiny-gqa"]
    d = tools["dir"]
    if ak:
        path, tok = str(d / "v.ak"), str(d / "v.tok")
        gguf.write_ak(path, gguf.synth_fused(s, int(g["seed"])))
        gguf.write_tokenizer_bin(tok, gguf.vocab_strings(s.vocab_size))
        args = ["-m", path, "--ak", "-s", tok]
    else:
        path = str(d / "v.gguf")
        gguf.write_synth_gguf(path, s, int(g["seed"]))
        args = ["-m", path]
    r = subprocess.run([tools["llm"]] + args + ["-n", str(int(g["n"])), "-t", "0", "-v"], capture_output=True)
    ref = bytes(g["stdout"]).split(b"\n")
    k = ref.index(b" Loaded weights") + 1
    assert r.stdout.split(b"\n")[:k] == ref[:k]


def test_fortran_bpe_encode_matches_the_reference_on_a_vocabulary_with_merges(tools, gguf):
    """`llm --encode` prints what the host's bpe_encode returns (no device needed).  tests/golden/tiny-gqa-merge.npz holds the
    ids the REAL reference used on the same vocabulary (pinned through its logits, tests/test_oracle.py): greedy best-score
    merging over several levels, the first pair on equal scores (llama2.f90:694), and the first index of a duplicated entry
    (llama2.f90:648-651) -- which the host's hashed lookup has to reproduce.  Also through `-s tokenizer.bin` (llama2.f90:321-356)."""
    g = load_golden("tiny-gqa-merge")
    s = gguf.SHAPES["tiny-gqa"]
    vocab, scores = gguf.merge_vocab(s.vocab_size)
    path = str(tools["dir"] / "merge.gguf")
    gguf.write_synth_gguf(path, s, int(g["seed"]), vocab=vocab, scores=scores)

    def ids(prompt, extra=()):
        r = subprocess.run([tools["llm"], "-m", path, "-p", prompt, "--encode", *extra], capture_output=True)
        lines = r.stdout.split(b"\n")
        assert lines[0].startswith(b" data offset"), r.stdout + r.stderr
        return [int(x) for x in lines[1].split()]
    assert ids(str(g["prompt"])) == g["prompt_ids"].tolist()
    words = lambda p: [vocab[i - 1] for i in ids(p)]
    assert words("ing") == [b"in", b"g"]
    assert words("her") == [b"he", b"r"]
    assert words("other") == [b"o", b"the", b"r"]
    assert words(" the") == [b" the"] and words("and") == [b"and"]
    assert ids("") == []
    # a plain vocabulary through the same loop: one token per character
    plain = str(tools["dir"] / "plain.gguf")
    gguf.write_synth_gguf(plain, s, int(g["seed"]))
    r = subprocess.run([tools["llm"], "-m", plain, "-p", "hi there", "--encode"], capture_output=True)
    assert [int(x) for x in r.stdout.split(b"\n")[1].split()] == load_golden("tiny-gqa-prompt")["prompt_ids"].tolist()
    # the same merges with the tokenizer read from tokenizer.bin over a plain-vocabulary file
    tok = str(tools["dir"] / "merge.tok")
    gguf.write_tokenizer_bin(tok, vocab, scores)
    r = subprocess.run([tools["llm"], "-m", plain, "-s", tok, "-p", str(g["prompt"]), "--encode"], capture_output=True)
    assert [int(x) for x in r.stdout.split(b"\n")[1].split()] == g["prompt_ids"].tolist()
