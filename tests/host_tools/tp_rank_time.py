"""Per-rank kernel time of the 70B q4_0 tensor-parallel configuration, measured on ONE GPU (no collectives):
rank 0 of 8 of a 70B-shaped model with fewer layers (per-layer times are what matter).
    python tests/host_tools/tp_rank_time.py [layers=4] [tp=8]
"""
import os
import sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
import bench

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
s70 = gguf.SHAPES["llama2-70b"]
s = gguf.LlamaShape(s70.emb_dim, s70.hidden_dim, L, s70.n_heads, s70.n_kv_heads, s70.vocab_size, 512)
llmk.Llmk.tp_init_comm = lambda self, uid: None          # no communicator: kernels only
t0 = time.time()
m = bench.build_streamed(s, 2, None, 0, llmk.FLAG_NO_GRAPH, 0, P, None, "none")
print(f"built rank 0/{P} of a 70B-shaped {L}-layer model in {time.time()-t0:.0f} s")
names = ["qkv", "attention", "wo (partial)", "w1|w3", "w2 (partial)", "classifier shard"]
tot = 0.0
for k in range(6):
    ms, b = m.time_kernel(k, 40)
    print(f"  {names[k]:18s} {ms*1000:8.1f} us   {b/1e6:8.2f} MB   {b/ms/1e6:7.0f} GB/s")
    if k < 5:
        tot += ms
ms, b = m.time_kernel(11, 20)
print(f"  whole layer in a hipGraph (as the token pass runs it, no exchanges) {ms*1000:8.1f} us   {b/1e6:8.2f} MB   {b/ms/1e6:7.0f} GB/s")
print(f"per layer {tot*1000:.1f} us -> 80 layers {tot*80:.2f} ms + classifier; weights per rank per layer "
      f"{sum(m.time_kernel(k,1)[1] for k in (0,2,3,4))/1e6:.1f} MB")
