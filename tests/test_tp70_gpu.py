"""BASELINE.json configs[4] -- Llama-2-70B q4_0, row-parallel GEMVs over 8 ranks -- at its REAL geometry on the one GPU
this box has: E 8192, H 28672, 64 query heads, 8 kv heads (head size 128), V 32000, eight rank PROCESSES with their inboxes
mapped over hipIpc (the link is HBM instead of xGMI).  Per rank: 8 query heads + 1 kv head, 3,584 hidden rows, 4,000
vocabulary rows, 1,024-column slices of wo and 3,584-column slices of w2 cut on q4_0 block boundaries.

(a) 2 layers against the oracle (the f32 reference path, llama2.f90:480-640, on the host-decoded weights), 64 positions;
(b) all 80 layers (38.6 GB of blocks generated directly in block format): size-independent properties.
The reference loops these shards wrap: llama2.f90:603-605 (wo), :618-620 (w2), :634-636 (classifier)."""
import os

import numpy as np
import pytest

from conftest import REL_TOL, rel_err
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu
P = 8
FIELDS = ("token_embedding_table", "rms_att_weight", "rms_ffn_weight", "rms_final_weight", "wqkv", "wo", "w13", "w2", "wcls")


class _Side:
    """the ranks' side channel over the test's pipes: all_gather_object as torch.distributed spells it (what bench.py's
    build_streamed expects of `rep.dist`); the parent relays (see _run_ranks)"""
    device = None

    def __init__(self, conn):
        self.conn = conn
        self.dist = self

    def all_gather_object(self, out, obj):
        self.conn.send(("gather", obj))
        out[:] = self.conn.recv()


def expected_checksum(fw, name, r):
    """What llmk_tensor_checksum must return for rank r's shard of tensor `name`: the 64-bit sum of the 32-bit words of the
    DEVICE image (csrc/llmk.hip upload_rows_sharded + q4_repack_kernel: per row the K/2 nibble bytes, then the K/32 f16
    scales, zero padding), computed here from the host arrays -- so a shard that arrived damaged, or was cut at the wrong
    rows / columns, is told apart from one that was damaged later."""
    s = fw.shape
    E, H, KV, V, hs = s.emb_dim, s.hidden_dim, s.kv_dim, s.vocab_size, s.head_size
    Eq, KVl, Hl, Vl = s.n_heads // P * hs, s.n_kv_heads // P * hs, H // P, V // P
    a = getattr(fw, name)
    if name in ("token_embedding_table", "rms_att_weight", "rms_ffn_weight", "rms_final_weight"):
        return int(np.ascontiguousarray(a).view("<u4").astype(np.uint64).sum() & np.uint64(0xFFFFFFFFFFFFFFFF))
    K = {"wqkv": E, "wo": E, "w13": E, "w2": H, "wcls": E}[name]
    b = np.asarray(a).reshape(-1, a.shape[-2], K // 32, 18)          # [L or 1][rows][blocks][18 bytes]
    if name == "wqkv":
        b = np.concatenate([b[:, r * Eq:(r + 1) * Eq], b[:, E + r * KVl:E + (r + 1) * KVl], b[:, E + KV + r * KVl:E + KV + (r + 1) * KVl]], axis=1)
    elif name == "wo":
        b = b[:, :, r * Eq // 32:(r + 1) * Eq // 32]
    elif name == "w13":
        b = np.concatenate([b[:, r * Hl:(r + 1) * Hl], b[:, H + r * Hl:H + (r + 1) * Hl]], axis=1)
    elif name == "w2":
        b = b[:, :, r * Hl // 32:(r + 1) * Hl // 32]
    else:
        b = b[:, r * Vl:(r + 1) * Vl]
    b = np.ascontiguousarray(b)
    assert b.shape[2] % 2 == 0                                          # scales pair up into whole words
    nib = np.ascontiguousarray(b[..., 2:18]).view("<u4").astype(np.uint64).sum()
    sc = np.ascontiguousarray(b[..., 0:2]).view("<u2")[..., 0].astype(np.uint64)
    tot = nib + sc[..., 0::2].sum() + np.uint64(65536) * sc[..., 1::2].sum()
    return int(tot & np.uint64(0xFFFFFFFFFFFFFFFF))


def _rank_oracle_case(rank, dirpath, n, prompt, conn):
    """one rank process: its shard of the weights the parent left in dirpath, handles over the pipe, n positions"""
    import llm_f90_amd     # noqa: F401
    from llm_f90_amd import llmk as lk
    from llm_f90_amd.tools import gguf as gg
    s = gg.LlamaShape(*np.load(os.path.join(dirpath, "shape.npy")).tolist())
    fw = gg.FusedWeights(s, 2)
    for f in FIELDS:
        setattr(fw, f, np.load(os.path.join(dirpath, f + ".npy"), mmap_mode="r"))
    m = lk.Llmk(fw, device=0, tp_rank=rank, tp_size=P)
    diag = {"sum_after_upload": {f: m.tensor_checksum(f) for f in FIELDS}}
    side = _Side(conn)
    handles, verdicts = [None] * P, [None] * P
    side.all_gather_object(handles, m.tp_p2p_handle())
    m.tp_p2p_connect(handles)
    side.all_gather_object(verdicts, m.tp_p2p_selftest(16))      # every collective on known integers, all ranks together
    try:
        _, logits = m.generate(n, prompt=prompt)
        greedy, _ = m.generate(8, want_logits=False, greedy_on_device=True)
        diag["error"] = None
    except lk.LlmkError as e:                                    # a timed-out exchange: deliver what there is, the parent reports it
        logits, greedy = np.full((n, s.vocab_size), np.nan, np.float32), np.zeros(8, np.int32)
        diag["error"] = str(e)
    diag["sum_after_run"] = {f: m.tensor_checksum(f) for f in FIELDS}
    diag["x"] = m.peek(0, s.emb_dim)
    diag["kv_row1"] = [m.peek(w, s.kv_dim // P, layer=l, pos=1) for w in (4, 5) for l in range(s.n_layers)]
    conn.send(("result", (logits, greedy, m.path(), verdicts, diag)))
    m.close()


def _rank_stress(rank, rounds, seed, conn):
    """one rank process of the jitter run: a weightless 70B-width ctx (the exchanges need E, V and the inboxes only)"""
    import llm_f90_amd     # noqa: F401
    from llm_f90_amd import llmk as lk
    from llm_f90_amd.tools import gguf as gg
    m = lk.Llmk.create_empty(gg.LlamaShape(8192, 28672, 1, 64, 8, 32000, 32), 2, tp_rank=rank, tp_size=P)
    side = _Side(conn)
    handles, first = [None] * P, [None] * P
    side.all_gather_object(handles, m.tp_p2p_handle())
    m.tp_p2p_connect(handles)
    side.all_gather_object(first, m.tp_p2p_selftest(8))
    import time
    rc, t0 = 0, time.time()
    with open(os.path.join(_fail_dir(), f"stress_rank{rank}.log"), "w") as hb:      # heartbeat: where each rank was, and when
        for part in range(0, rounds, 100):
            rc = m.tp_p2p_stress(min(100, rounds - part), seed + part)
            hb.write(f"{part + 100} rounds rc {rc} at {time.time() - t0:.2f} s\n")
            hb.flush()
            if rc:
                break
    after = m.tp_p2p_selftest(8) if rc == 0 else -1               # and the plain exchanges still work behind it
    conn.send(("result", (first, rc, after)))
    m.close()


def _rank_full_size(rank, n, conn):
    import bench
    import llm_f90_amd     # noqa: F401
    from llm_f90_amd.tools import gguf as gg
    m = bench.build_streamed(gg.SHAPES["llama2-70b"], 2, None, 0, 0, rank, P, _Side(conn), "p2p")
    t1, l1 = m.generate(n)
    t2, l2 = m.generate(n)
    t3, _ = m.generate(n, want_logits=False, greedy_on_device=True)
    conn.send(("result", (t1, l1, bool(np.array_equal(l1, l2) and np.array_equal(t1, t2)), t3, m.path(), m.p2p_selftest)))
    m.close()


def _fail_dir():
    from conftest import ROOT
    out = os.path.join(ROOT, "gpurun_out", "tp70_fail")
    os.makedirs(out, exist_ok=True)
    return out


def _rank_main(target, rank, args, conn, faildir):
    """entry point of a rank process: SIGUSR1 dumps every thread's stack to <faildir>/stack_rank<r>.txt (the parent asks
    for it when a rank goes silent), an exception leaves its traceback in <faildir>/error_rank<r>.txt"""
    import faulthandler
    import signal
    import threading
    import traceback
    if threading.current_thread() is threading.main_thread():
        faulthandler.register(signal.SIGUSR1, file=open(os.path.join(faildir, f"stack_rank{rank}.txt"), "w"), all_threads=True)
    try:
        target(*args, conn)
    except BaseException:
        with open(os.path.join(faildir, f"error_rank{rank}.txt"), "w") as f:
            traceback.print_exc(file=f)
        raise


def _run_ranks(target, args_of, timeout):
    """P ranks on this box's one GPU; the caller relays their all-gathers until every rank has delivered its result.

    EIGHT PROCESSES, NOT NINE.  The driver runs at most 8 processes on a GPU at once (amdgpu hws_max_conc_proc: one VMID
    each); a ninth makes the hardware scheduler time-slice whole processes, and a rank whose kernel spins on a peer that is
    not scheduled waits for the scheduler, not for the peer.  That is what round 3's "one failure in fifteen whole-suite runs"
    was (round 4: the jitter run reproduced it on every whole-suite run -- 35 rounds in 10 s, then `rank 0's partial never
    arrived` after the 20 s wall-clock bound -- and never in a dedicated run): inside the suite THIS pytest process already
    holds a GPU context from the tests before, so eight spawned ranks made nine.  Rank 0 therefore runs here, in a thread of
    this process, and ranks 1..7 are spawned: eight GPU processes whether or not the suite ran first.  A rank that dies is
    named with its exit code and traceback; if a rank goes silent every rank's Python stack is dumped before they are killed."""
    import multiprocessing as mp
    import signal
    import threading
    import time
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(P)]
    faildir = _fail_dir()
    procs = [None] + [ctx.Process(target=_rank_main, args=(target, r, args_of(r), pipes[r][1], faildir)) for r in range(1, P)]
    if os.environ.get("LLMK_TP_TEST_SPAWN_ALL"):      # the control of tests/host_tools/gpu_job.sh nine: rank 0 spawned too (round 3's harness)
        procs[0] = ctx.Process(target=_rank_main, args=(target, 0, args_of(0), pipes[0][1], faildir))
    local_err = []

    def local_rank():
        try:
            _rank_main(target, 0, args_of(0), pipes[0][1], faildir)
        except BaseException as e:      # noqa: BLE001
            local_err.append(e)
    th = threading.Thread(target=local_rank, daemon=True)
    for p in procs[1:]:
        p.start()
    if procs[0] is not None:
        procs[0].start()
    else:
        th.start()

    def read_file(name):
        try:
            return open(os.path.join(faildir, name)).read()[-1500:]
        except OSError:
            return ""

    def get(r):
        t_end = time.time() + timeout
        while not pipes[r][0].poll(1.0):
            dead = [(k, p.exitcode) for k, p in enumerate(procs) if p is not None and p.exitcode not in (None, 0)]
            if local_err:
                dead.append((0, repr(local_err[0])))
            if dead or time.time() > t_end:
                if not dead:
                    for p in procs:
                        if p is not None and p.is_alive():
                            os.kill(p.pid, signal.SIGUSR1)
                    time.sleep(2)
                for p in procs:
                    if p is not None:
                        p.terminate()
                if dead:
                    pytest.fail(f"rank(s) died (rank, exit code): {dead}\n" + "\n".join(read_file(f"error_rank{k}.txt") for k, _ in dead))
                pytest.fail(f"rank {r} went silent for {timeout} s; stacks:\n" + "\n".join(f"--- rank {k}\n" + read_file(f"stack_rank{k}.txt") for k in range(1, P)))
        return pipes[r][0].recv()
    while True:
        msgs = [get(r) for r in range(P)]
        kinds = {k for k, _ in msgs}
        assert len(kinds) == 1, kinds                 # the ranks move in lock step
        if kinds == {"result"}:
            break
        for r in range(P):
            pipes[r][0].send([obj for _, obj in msgs])
    if procs[0] is None:
        th.join(120)
        assert not th.is_alive() and not local_err, local_err
    for p in procs:
        if p is not None:
            p.join(120)
            assert p.exitcode == 0
    return [obj for _, obj in msgs]


def test_llama2_70b_geometry_8_rank_processes_match_oracle(gguf, tmp_path_factory):
    s = gguf.LlamaShape(8192, 28672, 2, 64, 8, 32000, 96)
    fw = gguf.synth_fused_q4_direct(s, 70)           # blocks generated directly; every block has its own scale (some negative)
    d = "/dev/shm" if os.path.isdir("/dev/shm") else None
    import tempfile
    with tempfile.TemporaryDirectory(dir=d, prefix="llmk_tp70_") as td:
        np.save(os.path.join(td, "shape.npy"), np.array([s.emb_dim, s.hidden_dim, s.n_layers, s.n_heads, s.n_kv_heads, s.vocab_size, s.seq_len]))
        for f in FIELDS:
            np.save(os.path.join(td, f + ".npy"), np.ascontiguousarray(getattr(fw, f)))
        n = 64
        ot, ol = Oracle(fw.as_f32(), "omp").generate(n)
        assert np.all(np.isfinite(ol))
        margin = np.sort(ol, axis=1)
        safe = (margin[:, -1] - margin[:, -2]) > 4 * REL_TOL * np.abs(ol).max(axis=1)

        res = _run_ranks(_rank_oracle_case, lambda r: (r, td, n, ot.tolist()), 600)
        want = {f: [expected_checksum(fw, f, r) for r in range(P)] for f in FIELDS}
        del fw
        try:
            for r, (logits, greedy, path, verdicts, diag) in enumerate(res):
                assert diag["error"] is None, (r, diag["error"])
                for f in FIELDS:                                           # the shard arrived whole, and the token pass left it alone
                    assert diag["sum_after_upload"][f] == want[f][r], (r, f, "upload")
                    assert diag["sum_after_run"][f] == want[f][r], (r, f, "after the run")
                assert np.all(np.isfinite(logits)), (r, np.argwhere(~np.isfinite(logits))[:4].tolist())
                assert path == 2                                           # tensor-parallel rank over the peer-memory collectives
                assert verdicts == [0] * P                                 # llmk_tp_p2p_selftest: exact sums on every rank
                err = rel_err(logits, ol)
                assert err.max() <= REL_TOL, (r, err.max(), int(np.argmax(err)))
                assert np.array_equal((np.argmax(logits, axis=1) + 1)[safe], ot[safe])
                assert np.array_equal(logits, res[0][0]), r                # rank-order sums: bit-identical on all eight ranks
                assert np.array_equal(greedy[safe[:8]], ot[:8][safe[:8]])
        except AssertionError:
            _dump_failure("geometry", res, ol, want)                       # what failed, on which rank, from which position on
            raise


def _dump_failure(tag, res, ol, want):
    """Everything a post-mortem needs, where the lease pulls it from (gpurun_out/tp70_fail/): per rank the first non-finite
    and the first out-of-tolerance position with the index of the worst logit, the error text, the self-test verdicts,
    checksum mismatches, x and the first K/V rows; the logits of the first failing rank in full."""
    from conftest import ROOT
    out = os.path.join(ROOT, "gpurun_out", "tp70_fail")
    os.makedirs(out, exist_ok=True)
    lines, saved = [], False
    for r, (logits, greedy, path, verdicts, diag) in enumerate(res):
        fin = np.isfinite(logits).all(axis=1)
        err = rel_err(np.nan_to_num(logits), ol)
        badpos = np.flatnonzero(~fin | (err > REL_TOL))
        sums = [f"{f}:{k}" for f in FIELDS for k in ("sum_after_upload", "sum_after_run") if diag[k][f] != want[f][r]]
        lines.append(f"rank {r}: path {path} verdicts {verdicts} error {diag['error']!r} first non-finite position "
                     f"{int(np.argmin(fin)) if not fin.all() else None} bad positions {badpos[:8].tolist()} (of {len(badpos)}) "
                     f"non-finite logits {int((~np.isfinite(logits)).sum())} checksum mismatches {sums} "
                     f"x finite {bool(np.isfinite(diag['x']).all())} "
                     f"kv rows finite {[bool(np.isfinite(v).all()) for v in diag['kv_row1']]} same as rank 0 {bool(np.array_equal(logits, res[0][0]))}")
        if len(badpos) and not saved:
            p0 = int(badpos[0])
            lines.append(f"  rank {r} position {p0}: worst logit index {int(np.argmax(np.abs(np.nan_to_num(logits[p0]) - ol[p0])))}, "
                         f"non-finite columns {np.flatnonzero(~np.isfinite(logits[p0]))[:8].tolist()}")
            np.savez_compressed(os.path.join(out, f"{tag}_rank{r}.npz"), logits=logits, oracle=ol, x=diag["x"], kv=np.asarray(diag["kv_row1"]))
            saved = True
    with open(os.path.join(out, f"{tag}.txt"), "a") as f:
        f.write("\n".join(lines) + "\n\n")
    print("\n".join(lines))


def test_peer_memory_collectives_under_jitter_8_rank_processes():
    """Round-3 verdict item 1: the REAL exchange kernels (csrc/tp_p2p.h) at the 70B widths (E 8192, V 32000), eight rank
    processes on this GPU, 2,500 rounds of two all-reduces + the all-gather with pseudo-random delays of up to ~15 us
    before and between the sends and the reads of every wavefront -- ranks and waves drift apart by up to a whole exchange.
    Every round's sums are checked exactly on every rank; the plain self-test must still pass behind it."""
    res = _run_ranks(_rank_stress, lambda r: (r, 2500, 0xC0FFEE + 17 * r), 600)
    for r, (first, rc, after) in enumerate(res):
        assert first == [0] * P, (r, first)
        assert rc == 0, (r, rc)
        assert after == 0, (r, after)


def test_llama2_70b_full_size_8_rank_processes_properties():
    """All 80 layers, 4.83 GB of shards per rank: (1) finite, non-trivial logits; (2) the eight ranks hold bit-identical
    logits; (3) two runs are bit-identical; (4) the device argmax picks the host argmax."""
    n = 6
    res = _run_ranks(_rank_full_size, lambda r: (r, n), 900)
    t1, l1 = res[0][0], res[0][1]
    assert np.all(np.isfinite(l1)) and np.abs(l1).max() > 1e-3
    for r, (t, l, rerun_same, tg, path, verdicts) in enumerate(res):
        assert path == 2 and verdicts == [0] * P
        assert np.array_equal(l, l1) and np.array_equal(t, t1), r
        assert rerun_same, r
        assert np.array_equal(tg, t1), r
