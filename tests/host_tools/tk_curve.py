"""token kernel duration vs KV length: python tests/host_tools/tk_curve.py [--type f32|f16|q4_0] [--shape NAME] [positions...]
(long positions are reached through llmk_prefill where the shape has it: the curve is about the kernel, not about getting there)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
import numpy as np
argv = sys.argv[1:]
def opt(name, default):
    if name in argv:
        i = argv.index(name); v = argv[i + 1]; del argv[i:i + 2]; return v
    return default
wtype = {"f32": 0, "f16": 1, "q4_0": 2}[opt("--type", "f32")]
shape = opt("--shape", "tinyllama")
s = gguf.SHAPES[shape]
if shape == "llama2-7b":
    fw = gguf.synth_fused_q4_direct(s, 20260928)
else:
    fw = gguf.synth_fused(s, 20260928, wtype)
m = llmk.Llmk(fw)
pts = [int(a) for a in argv] or [1, 32, 64, 127, 129, 192, 256, 257, 384, 512, 1024]
tok, pos = 2, 0
out = []
for p in pts:
    if p - pos > 64:       # a random prompt through the batched prefill: K/V rows of the right shape, quickly
        rng = np.random.default_rng(p)
        toks = rng.integers(1, s.vocab_size, p - 1 - pos).astype(np.int32)
        m.prefill(toks, pos + 1)
        pos = p - 1
    while pos < p:
        pos += 1
        lg = m.forward(tok, pos)
        tok = int(lg.argmax()) + 1
    ms, b = m.time_kernel(6, 50)
    out.append(f"{p}:{ms*1000:.1f}")
print("token kernel us by KV length:", " ".join(out))
