"""prefill timing: [LLMK_WT=1 for f16] python tests/host_tools/pf_time.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
s = gguf.SHAPES["tinyllama"]
fw = gguf.synth_fused(s, 20260928, int(os.environ.get('LLMK_WT', '0')))
m = llmk.Llmk(fw)
rng = np.random.default_rng(1)
for n in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128, 512]:
    prompt = [2] + (rng.integers(3, s.vocab_size, n - 1) + 1).tolist()
    m.reset(); m.prefill(prompt, 1)
    t0 = time.perf_counter()
    for _ in range(5):
        m.prefill(prompt, 1)
    dt = (time.perf_counter() - t0) / 5
    print(f"prefill n={n}: {dt*1e3:.3f} ms  {n/dt:.0f} tok/s  weights pass rate {4.14e9*((n+63)//64)/dt/1e12:.2f} TB/s")
