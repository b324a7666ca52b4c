"""N>1 path on CPU: two processes, gloo, 127.0.0.1 -- the barrier / max-over-ranks / whole-job
aggregation bench.py uses for its replicas-only scaling mode (no data-path collective exists)."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    import llm_f90_amd
    from llm_f90_amd.replicas import Replicas
    r = Replicas(backend="gloo")
    assert r.world == 2
    r.barrier()
    elapsed = 0.5 if r.rank == 0 else 2.0          # rank 1 is the slow replica
    assert abs(r.max_over_ranks(elapsed) - 2.0) < 1e-9
    rate = r.aggregate_rate(100, elapsed)            # 2 replicas x 100 tokens / 2.0 s
    assert abs(rate - 100.0) < 1e-9, rate
    r.barrier()
    r.close()
    print("rank", r.rank, "ok")
""") % ROOT


def test_two_replicas_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for rank, p in enumerate(procs):
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()
        assert f"rank {rank} ok" in out.decode()


def test_single_process_is_identity():
    from llm_f90_amd.replicas import Replicas
    env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
    try:
        r = Replicas()
        assert r.world == 1 and r.max_over_ranks(1.5) == 1.5 and r.aggregate_rate(10, 2.0) == 5.0
        r.barrier()
        r.close()
    finally:
        os.environ.update(env)
