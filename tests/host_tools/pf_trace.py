"""Block timeline of the prefill GEMMs (pf_gemm_kernel): wall-clock stamps per workgroup from the DEBUG library.
   LLMK_LIB=llm.f90_amd/csrc/libllmk_debug.so LLMK_PF_PLAN=1|2 python tests/host_tools/pf_trace.py [--type f32|f16|q4_0] [w13|wqkv|wo|w2]
   LLMK_PF_PLAN forces the 16-row groups per wave of pf_plan (csrc/llmk.hip); the tool needs it to know the grid."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
s = gguf.SHAPES["tinyllama"]
wt = 0
if "--type" in sys.argv:
    i = sys.argv.index("--type"); wt = {"f32": 0, "f16": 1, "q4_0": 2}[sys.argv[i + 1]]; del sys.argv[i:i + 2]
which = sys.argv[1] if len(sys.argv) > 1 else "w13"
kern = {"w13": 7, "wqkv": 8, "wo": 9, "w2": 10}[which]
rows = {"w13": 2 * s.hidden_dim, "wqkv": s.emb_dim + 2 * s.kv_dim, "wo": s.emb_dim, "w2": s.emb_dim}[which]
K = s.hidden_dim if which == "w2" else s.emb_dim
m = llmk.Llmk(gguf.synth_fused(s, 1, wt))
m.prefill([2] + list(range(5, 5 + 127)), 1)
ms, b = m.time_kernel(kern, 66)
print("%s GEMM, 128 positions: %.1f us per launch (66 launches), %.1f TFLOP/s" % (which, ms * 1e3, 2 * 128 * rows * K / (ms * 1e-3) / 1e12))
if which == "w2":
    sys.exit(0)            # its activations live where the stamps go
ms, b = m.time_kernel(kern, 1)
nr = int(os.environ.get("LLMK_PF_PLAN", "1"))
total = (rows // (64 * nr)) * (K // 64)
U = -(-total // 256)
nb = -(-total // U)
u = m.peek(7, nb * 40).view(np.uint64).reshape(nb, 20)
ok = u[:, 17] > u[:, 0]
print("shader clock while the kernel runs: %.0f MHz (median over blocks)" % np.median((u[ok, 19] - u[ok, 18]).astype(np.float64) / ((u[ok, 17] - u[ok, 0]).astype(np.float64) / 100.0)))
t = u.astype(np.float64) / 100.0   # us (100 MHz)
t -= t[:, 0].min()
pc = lambda v: "min %6.2f  med %6.2f  p90 %6.2f  max %6.2f" % tuple(np.percentile(v, [0, 50, 90, 100]))
print("blocks", nb, "units per block", U, "(assuming the forced plan applied)")
print("entry     :", pc(t[ok, 0]))
print("prologue  :", pc(t[ok, 1] - t[ok, 0]))
for i in range(min(U, 15)):
    print("step %2d   :" % i, pc(t[ok, 2 + i] - t[ok, 1 + i]))
print("exit      :", pc(t[ok, 17]))
