"""prompt tokens/s of llmk_prefill against the prompt length: python tests/host_tools/pf_curve.py [--type T] [n ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
args = sys.argv[1:]
wt = 0
if args and args[0] == "--type":
    wt = {"f32": 0, "f16": 1, "q4_0": 2}[args[1]]; args = args[2:]
s = gguf.SHAPES["tinyllama"]
m = llmk.Llmk(gguf.synth_fused(s, 1, wt))
rng = np.random.default_rng(1)
for n in [int(a) for a in args] or [8, 16, 32, 64, 100, 128, 129, 192, 256, 384, 512, 1024]:
    prompt = [2] + (rng.integers(3, s.vocab_size, n - 1) + 1).tolist()
    m.reset(); m.prefill(prompt, 1)
    reps = max(3, 1500 // n)
    t0 = time.perf_counter()
    for _ in range(reps):
        m.prefill(prompt, 1)
    dt = (time.perf_counter() - t0) / reps
    print(f"n={n:5d}: {dt*1e3:7.3f} ms  {n/dt:8.0f} prompt tok/s")
