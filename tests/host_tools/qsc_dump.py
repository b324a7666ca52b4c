import sys, numpy as np
sys.path.insert(0, '/root/repo')
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
import bench
s = gguf.SHAPES["llama2-7b"]
m = bench.build_streamed(s, 2, None, 0, 0, 0, 1, None)
tok = 2
for pos in range(1, 8):
    lg = m.forward(tok, pos)
    tok = int(np.argmax(lg)) + 1
    r = m.peek(8, 8 * 128).reshape(2, 128, 4)
    b = r[pos & 1]
    tags = b[:, 2:].view(np.int32)
    print("pos", pos, "path", m.path(), "buf", pos & 1, "layers 0..5 |xb|", np.round(b[:6, 0], 3), "|hb|", np.round(b[:6, 1], 3), "tags", tags[:3].tolist(), "layer 31", b[31, :2], tags[31].tolist())
