"""Block timeline of the prefill w1|w3 GEMM (pf_gemm_kernel): wall-clock stamps per workgroup from the DEBUG library.
   LLMK_LIB=llm.f90_amd/csrc/libllmk_debug.so python tests/host_tools/pf_trace.py [positions]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
s = gguf.SHAPES["tinyllama"]
m = llmk.Llmk(gguf.synth_fused(s, 1, 0))
m.prefill([2] + list(range(5, 5 + 127)), 1)
ms, b = m.time_kernel(7, 66)
print("w1|w3 GEMM, 128 positions: %.1f us per launch (66 launches), %.1f TFLOP/s" % (ms * 1e3, 2 * 128 * b / 4 / (ms * 1e-3) / 1e12))
ms, b = m.time_kernel(7, 1)
total = (2 * s.hidden_dim // 64) * (s.emb_dim // 64)
nb = 512 if total % 512 == 0 or total / (512 * -(-total // 512)) + 0.08 >= total / (256 * -(-total // 256)) else 256
U = -(-total // nb)
nb = -(-total // U)
raw = m.peek(7, nb * 40)
t = raw.view(np.uint64).reshape(nb, 20).astype(np.float64) / 100.0   # us (100 MHz)
t -= t[:, 0].min()
pc = lambda v: "min %6.2f  med %6.2f  p90 %6.2f  max %6.2f" % tuple(np.percentile(v, [0, 50, 90, 100]))
print("blocks", nb, "units per block", U)
print("entry     :", pc(t[:, 0]))
print("prologue  :", pc(t[:, 1] - t[:, 0]))
for i in range(min(U, 16)):
    print("step %2d   :" % i, pc(t[:, 2 + i] - t[:, 1 + i]))
print("exit      :", pc(t[:, 1 + min(U, 16)]))
