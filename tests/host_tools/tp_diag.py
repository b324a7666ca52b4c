"""diagnostic: two virtual ranks on one GPU through the peer-memory collectives, eager and graph mode"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
s = gguf.SHAPES["tiny-gqa"]
fw = gguf.synth_fused(s, 20260928)
for flags, name in ((llmk.FLAG_NO_GRAPH, "eager"), (0, "graph")):
    ranks = [llmk.Llmk(fw, tp_rank=r, tp_size=2, flags=flags) for r in range(2)]
    llmk.Llmk.tp_p2p_connect_local(ranks)
    out = [None, None]
    def run(i):
        t0 = time.time()
        try:
            out[i] = ranks[i].generate(4)[0]
        except Exception as e:
            out[i] = repr(e)
        print(name, "rank", i, "done in %.2fs" % (time.time() - t0), out[i], flush=True)
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for m in ranks: m.close()
