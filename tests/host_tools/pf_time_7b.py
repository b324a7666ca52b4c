"""prefill timing on the Llama-2-7B q4_0 shapes: python tests/host_tools/pf_time_7b.py [n ...]"""
import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
import bench
s = gguf.SHAPES["llama2-7b"]
m = bench.build_streamed(s, 2, None, 0, 0, 0, 1, None)
rng = np.random.default_rng(1)
for n in [int(a) for a in sys.argv[1:]] or [64, 256]:
    prompt = [2] + (rng.integers(3, s.vocab_size, n - 1) + 1).tolist()
    m.reset(); lg = m.prefill(prompt, 1)
    t0 = time.perf_counter()
    for _ in range(3):
        lg = m.prefill(prompt, 1)
    dt = (time.perf_counter() - t0) / 3
    m.reset()
    t0 = time.perf_counter()
    for pos in range(1, 33):
        ls = m.forward(prompt[pos - 1], pos)
    ds = (time.perf_counter() - t0) / 32
    print(f"7B q4_0 prefill n={n}: {dt*1e3:.2f} ms  {n/dt:.0f} tok/s   (token by token: {1/ds:.0f} tok/s, {n/dt*ds:.1f}x)  "
          f"{2*6.607e9*n/dt/1e12:.1f} TFLOP/s")
# parity spot check: prefill of 40 tokens vs sequential
p = [2] + (rng.integers(3, s.vocab_size, 39) + 1).tolist()
m.reset(); a = m.prefill(p, 1)
m.reset()
for pos, tok in enumerate(p, 1):
    b = m.forward(tok, pos)
print("max rel diff prefill vs sequential:", float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
