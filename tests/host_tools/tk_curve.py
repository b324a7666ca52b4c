"""token kernel duration vs KV length: python tests/host_tools/tk_curve.py [positions...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
s = gguf.SHAPES["tinyllama"]
fw = gguf.synth_fused(s, 20260928)
m = llmk.Llmk(fw)
pts = [int(a) for a in sys.argv[1:]] or [1, 32, 64, 127, 129, 192, 256, 257, 384, 512, 1024]
tok, pos = 2, 0
out = []
for p in pts:
    while pos < p:
        pos += 1
        lg = m.forward(tok, pos)
        tok = int(lg.argmax()) + 1
    ms, b = m.time_kernel(6, 50)
    out.append(f"{p}:{ms*1000:.1f}")
print("token kernel us by KV length:", " ".join(out))
