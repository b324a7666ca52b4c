#!/usr/bin/env python3
"""Model-file converter: the working replacement of the reference's `load.f90` (GGUF -> "ak" flat checkpoint +
tokenizer.bin, /root/reference/load.f90:313-421,477-501 -- which does not compile in master: it `use`s a module whose
definition is commented out, load.f90:117-161), extended to re-encode a GGUF's matrices as f16 or q4_0.

    python llm.f90_amd/tools/convert.py model.gguf --ak model.bin --tokenizer tokenizer.bin     # what ./llm --ak -s reads
    python llm.f90_amd/tools/convert.py model.gguf --gguf model-q4_0.gguf --type q4_0           # f32 -> f16 / q4_0 GGUF

Reads f32, f16, q4_0 matrices and a q6_K output.weight (stock llama.cpp q4_0 files); "ak" is f32 by definition
(llama2.f90:160-292 reads float32 only).  Host-side file work only: no GPU, no arithmetic beyond (de)quantisation.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import llm_f90_amd  # noqa: E402,F401
from llm_f90_amd.tools import gguf  # noqa: E402


def tokenizer_of(path: str):
    """vocabulary strings (sentencepiece U+2581 -> ' ', as read_ggml.f90:479-497 does) and scores of a GGUF"""
    g = gguf.read_gguf(path)
    toks = [bytes(t).replace("▁".encode(), b" ") for t in g.kv["tokenizer.ggml.tokens"]]
    scores = np.asarray(g.kv.get("tokenizer.ggml.scores", -np.arange(len(toks))), np.float32)
    return toks, scores, g


def reencode(fw: gguf.FusedWeights, ggml_type: int) -> gguf.FusedWeights:
    f = fw.as_f32()
    s = f.shape
    enc = lambda a: gguf.encode(np.asarray(a, np.float32), ggml_type)
    return gguf.FusedWeights(s, ggml_type, f.token_embedding_table, f.rms_att_weight, f.rms_ffn_weight, f.rms_final_weight,
                             enc(f.wqkv), enc(f.wo), enc(f.w13), enc(f.w2), enc(f.wcls))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("model", help="input GGUF (llama architecture)")
    ap.add_argument("--ak", metavar="OUT", help='write the llama2.c-style flat f32 checkpoint ("ak" format)')
    ap.add_argument("--tokenizer", metavar="OUT", help="write tokenizer.bin (max_len, then score/len/bytes per token)")
    ap.add_argument("--gguf", metavar="OUT", help="write a GGUF with the matrices re-encoded as --type")
    ap.add_argument("--type", default="f32", choices=["f32", "f16", "q4_0"])
    a = ap.parse_args(argv)
    if not (a.ak or a.tokenizer or a.gguf):
        ap.error("nothing to do: give --ak, --tokenizer and/or --gguf")
    fw = gguf.load_fused(a.model)
    s = fw.shape
    print(f"{a.model}: emb {s.emb_dim} hidden {s.hidden_dim} layers {s.n_layers} heads {s.n_heads}/{s.n_kv_heads} vocab {s.vocab_size} "
          f"ctx {s.seq_len}, matrices ggml type {fw.ggml_type}" + (f", classifier type {fw.cls_type}" if fw.cls_type != fw.ggml_type else ""))
    if a.ak:
        gguf.write_ak(a.ak, fw.as_f32())
        print("wrote", a.ak)
    if a.tokenizer:
        toks, scores, _ = tokenizer_of(a.model)
        gguf.write_tokenizer_bin(a.tokenizer, toks, scores)
        print("wrote", a.tokenizer)
    if a.gguf:
        t = {"f32": 0, "f16": 1, "q4_0": 2}[a.type]
        _, _, g = tokenizer_of(a.model)
        gguf.write_gguf(a.gguf, reencode(fw, t) if (t != fw.ggml_type or fw.cls_type != fw.ggml_type) else fw,
                        rms_eps=float(g.kv.get("llama.attention.layer_norm_rms_epsilon", 1e-5)),
                        rope_freq_base=(float(g.kv["llama.rope.freq_base"]) if "llama.rope.freq_base" in g.kv else None),
                        vocab=[bytes(t_) for t_ in g.kv["tokenizer.ggml.tokens"]],
                        scores=np.asarray(g.kv.get("tokenizer.ggml.scores", -np.arange(s.vocab_size)), np.float32))
        print("wrote", a.gguf)
    return 0


if __name__ == "__main__":
    sys.exit(main())
