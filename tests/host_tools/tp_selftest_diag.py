"""llmk_tp_p2p_selftest across rank PROCESSES on one GPU (hipIpc-mapped inboxes): return code per rank and iteration count.
    python tests/host_tools/tp_selftest_diag.py [P=2] [iters=64] [stagger_s=0]"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def rank_main(rank, P, iters, stagger, conn):
    import llm_f90_amd  # noqa: F401
    from llm_f90_amd import llmk
    from llm_f90_amd.tools import gguf
    fw = gguf.synth_fused(gguf.SHAPES["tiny-mha" if P > 2 else "tiny-gqa"], 5)
    m = llmk.Llmk(fw, device=0, tp_rank=rank, tp_size=P)
    conn.send(m.tp_p2p_handle())
    m.tp_p2p_connect(conn.recv())
    time.sleep(stagger * rank)
    t0 = time.time()
    rc = m.tp_p2p_selftest(iters)
    conn.send((rc, time.time() - t0))
    m.close()


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    stagger = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(P)]
    procs = [ctx.Process(target=rank_main, args=(r, P, iters, stagger, pipes[r][1])) for r in range(P)]
    [p.start() for p in procs]
    handles = [pipes[r][0].recv() for r in range(P)]
    [pipes[r][0].send(handles) for r in range(P)]
    print(f"P={P} iters={iters} stagger={stagger}:", [pipes[r][0].recv() if pipes[r][0].poll(120) else "silent" for r in range(P)])
    [p.join(30) for p in procs]
