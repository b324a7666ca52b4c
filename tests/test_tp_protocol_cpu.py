"""The tensor-parallel exchange protocol across REAL process boundaries on CPU (world size 2, gloo, 127.0.0.1): each
process plays one rank of tests/tp_cpu_model.py -- the same ownership rules llmk_create_tp / llmk_upload apply and the
same three exchange points the GPU token pass has -- with torch.distributed collectives in place of the peer-memory
kernels, and must reproduce the REAL reference's logits and greedy ids (tests/golden).  The GPU tests cover the HIP side
of the same split (tests/test_tp_gpu.py: virtual ranks, two processes over hipIpc); this covers the partition and the
protocol when only CPUs are available."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import REL_TOL, ROOT, load_golden, rel_err

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import llm_f90_amd
    from llm_f90_amd.tools import gguf
    from tp_cpu_model import TpRank
    from conftest import load_golden
    dist.init_process_group("gloo")
    rank, P = dist.get_rank(), dist.get_world_size()
    g = load_golden(%(tag)r)
    fw = gguf.synth_fused(gguf.SHAPES[str(g["shape"])], int(g["seed"]))
    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v)); dist.all_reduce(t); return t.numpy()
    def allgather(v):
        parts = [torch.empty(v.size, dtype=torch.float32) for _ in range(P)]
        dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(v)))
        return np.concatenate([p.numpy() for p in parts])
    m = TpRank(fw, rank, P, allreduce, allgather)
    n = %(n)d
    tok, toks, logits = 2, [], []
    for pos in range(1, n + 1):
        lg = m.forward(tok, pos)
        tok = int(np.argmax(lg)) + 1
        toks.append(tok); logits.append(lg)
    np.savez(os.path.join(%(out)r, "rank%%d.npz" %% rank), toks=np.asarray(toks), logits=np.asarray(logits))
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok")
""")


@pytest.mark.parametrize("tag,n", [("tiny-gqa", 10), ("tiny-mha", 8)])
def test_two_rank_tensor_parallel_protocol_over_gloo(tag, n, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "tag": tag, "n": n, "out": str(tmp_path)})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()
    g = load_golden(tag)
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    for r in res:
        assert rel_err(r["logits"], g["logits"][:n]).max() <= 1e-5      # same arithmetic, partial sums regrouped per rank
        assert np.array_equal(r["toks"], g["tokens"][:n])
    assert np.array_equal(res[0]["logits"], res[1]["logits"])           # both ranks hold the same replicated stream


@pytest.mark.parametrize("P", [2, 3])
def test_two_half_epoch_rule_holds_over_every_interleaving(P):
    """csrc/tp_p2p.h's claim -- a half of the inbox is never overwritten while a peer still reads it -- checked exhaustively:
    every interleaving of the ranks' sends and blocking reads over three token passes (two all-reduces + the all-gather
    each), no ordering assumed between ranks beyond what the granules themselves enforce.  The same search with ONE half
    must find the overwrite (the checker has teeth)."""
    from tp_cpu_model import check_interleavings
    assert check_interleavings(P, tokens=3, ncalls=2, halves=2) is None
    assert check_interleavings(P, tokens=4, ncalls=4, halves=2) is None
    bad = check_interleavings(P, tokens=3, ncalls=2, halves=1)
    assert bad is not None and bad.startswith("deadlock"), bad
