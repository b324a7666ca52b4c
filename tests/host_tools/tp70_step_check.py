"""70B geometry, 8 virtual ranks stepped in ONE process (the test plays the collectives): isolates shard upload + rank kernels
from the peer-memory exchange.  LLMK_Q4_KS=0 forbids the column-sliced GEMV."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
from oracle.oracle import Oracle
from test_tp_gpu import tp_generate
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
s = gguf.LlamaShape(8192, 28672, L, 64, 8, 32000, 32)
fw = gguf.synth_fused_q4_direct(s, 70)
n = 3
ot, ol = Oracle(fw.as_f32(), "omp").generate(n)
ranks = [llmk.Llmk(fw, tp_rank=r, tp_size=8) for r in range(8)]
toks, logits = tp_generate(ranks, n, s)
print("KS", os.environ.get("LLMK_Q4_KS"), "nan", int(np.isnan(logits).sum()), "rel", float(np.nanmax(np.abs(logits - ol)) / np.abs(ol).max()), toks, ot)
for which, name in ((0, "x"), (1, "q"), (2, "xb"), (3, "hb")):
    v = ranks[0].peek(which, 8)
    print(" rank0", name, v[:4])
