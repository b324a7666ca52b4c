"""70B geometry, 8 rank PROCESSES over hipIpc on one GPU: where do non-finite logits come from?
    python tests/host_tools/tp70_proc_diag.py [L=2] [selftest_iters=16]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def rank_main(rank, dirpath, n, prompt, iters, conn):
    import llm_f90_amd  # noqa: F401
    from llm_f90_amd import llmk as lk
    from llm_f90_amd.tools import gguf as gg
    import test_tp70_gpu as T
    s = gg.LlamaShape(*np.load(os.path.join(dirpath, "shape.npy")).tolist())
    fw = gg.FusedWeights(s, 2)
    for f in T.FIELDS:
        setattr(fw, f, np.load(os.path.join(dirpath, f + ".npy"), mmap_mode="r"))
    m = lk.Llmk(fw, device=0, tp_rank=rank, tp_size=T.P)
    side = T._Side(conn)
    handles, verdicts = [None] * T.P, [None] * T.P
    side.all_gather_object(handles, m.tp_p2p_handle())
    m.tp_p2p_connect(handles)
    side.all_gather_object(verdicts, m.tp_p2p_selftest(iters) if iters else 0)
    out = []
    tok = 2
    for pos in range(1, n + 1):
        lg = m.forward(tok, pos)
        out.append((int(np.isnan(lg).sum()), float(np.nanmax(np.abs(lg))), [float(v) for v in m.peek(0, 4)]))
        tok = prompt[pos - 1]
    conn.send(("result", (out, verdicts, lg)))
    m.close()


if __name__ == "__main__":
    import llm_f90_amd  # noqa: F401
    from llm_f90_amd.tools import gguf
    from oracle.oracle import Oracle
    import test_tp70_gpu as T
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    s = gguf.LlamaShape(8192, 28672, L, 64, 8, 32000, 96)
    fw = gguf.synth_fused_q4_direct(s, 70)
    n = 3
    ot, ol = Oracle(fw.as_f32(), "omp").generate(n)
    with tempfile.TemporaryDirectory(dir="/dev/shm", prefix="llmk_diag_") as td:
        np.save(os.path.join(td, "shape.npy"), np.array([s.emb_dim, s.hidden_dim, s.n_layers, s.n_heads, s.n_kv_heads, s.vocab_size, s.seq_len]))
        for f in T.FIELDS:
            np.save(os.path.join(td, f + ".npy"), np.ascontiguousarray(getattr(fw, f)))
        res = T._run_ranks(rank_main, lambda r: (r, td, n, ot.tolist(), iters), 120)
    for r, (out, verdicts, lg) in enumerate(res):
        print(f"L={L} selftest={iters} rank {r}: per position (nan count, max|logit|, x[:4]) {out} verdicts {verdicts} "
              f"rel(last) {float(np.nanmax(np.abs(lg - ol[-1])) / np.abs(ol[-1]).max()):.2e}")
