"""Tensor-parallel sharding verified on ONE GPU.

* virtual ranks: P contexts created with llmk_create_tp(rank r of P) on the same device, each holding only
  its shard (cut by llmk_upload from the full tensors); the test plays the collective through the
  stepping hooks (sum of the partial E-vectors in rank order, concatenation of the logits slices).  Same
  kernels, same sharded layouts and the same segment order as the RCCL path.
* RCCL: a 1-rank communicator through llmk_tp_init_comm -> llmk_forward (ncclAllReduce / ncclAllGather
  on the ctx stream).  The multi-rank RCCL run itself needs a multi-GPU node (driver side).
"""
import numpy as np
import pytest

from conftest import REL_TOL, load_golden, rel_err
from llm_f90_amd import llmk
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def tp_generate(ranks, n, shape):
    """The reference generation loop at temperature 0 with the collectives done by the test."""
    P = len(ranks)
    for m in ranks:
        m.reset()
    toks = np.zeros(n, np.int32)
    logits = np.empty((n, shape.vocab_size), np.float32)
    token = 2

    def exchange():
        total = ranks[0].tp_read_partial()
        for m in ranks[1:]:
            total = total + m.tp_read_partial()       # fixed rank order, f32
        for m in ranks:
            m.tp_write_partial(total)

    for pos in range(1, n + 1):
        for m in ranks:
            m.tp_begin(token, pos)
        for layer in range(shape.n_layers):
            for m in ranks:
                m.tp_segment(0, layer)
            exchange()
            for m in ranks:
                m.tp_segment(1, layer)
            exchange()
        for m in ranks:
            m.tp_segment(2)
        logits[pos - 1] = np.concatenate([m.tp_read_logits() for m in ranks])
        token = int(np.argmax(logits[pos - 1])) + 1
        toks[pos - 1] = token
    return toks, logits


@pytest.mark.parametrize("shape,P", [("tiny-gqa", 2), ("tiny-mha", 2), ("tiny-mha", 4), ("tk-small", 2)])
def test_virtual_ranks_f32_match_reference_golden(shape, P, gguf):
    g = load_golden(shape)
    s = gguf.SHAPES[shape]
    fw = gguf.synth_fused(s, int(g["seed"]))
    ranks = [llmk.Llmk(fw, tp_rank=r, tp_size=P) for r in range(P)]
    n = min(int(g["n"]), 12)
    toks, logits = tp_generate(ranks, n, s)
    assert rel_err(logits, g["logits"][:n]).max() <= REL_TOL
    assert np.array_equal(toks, g["tokens"][:n])
    for m in ranks:
        m.close()


@pytest.mark.parametrize("wtype", [1, 2], ids=["f16", "q4_0"])
def test_virtual_ranks_quantised_match_oracle(wtype, gguf):
    """Column slices of q4_0 rows are cut on 32-weight block boundaries and re-packed per shard."""
    s = gguf.SHAPES["tiny-mha"]                   # E 128, nh 4, nkv 4, H 352 ... H/P must be a multiple of 32
    s = gguf.LlamaShape(128, 384, 2, 4, 4, 512, 32)
    fw = gguf.synth_fused(s, 99, wtype)
    P = 2
    ranks = [llmk.Llmk(fw, tp_rank=r, tp_size=P) for r in range(P)]
    n = 8
    toks, logits = tp_generate(ranks, n, s)
    ot, ol = Oracle(fw.as_f32(), "omp").generate(n)
    assert rel_err(logits, ol).max() <= REL_TOL
    margin = np.sort(ol, axis=1)
    safe = (margin[:, -1] - margin[:, -2]) > 4 * REL_TOL * np.abs(ol).max()
    assert np.array_equal(toks[safe], ot[safe])
    for m in ranks:
        m.close()


def test_virtual_ranks_with_a_q6k_classifier_match_oracle(gguf):
    """A stock llama.cpp q4_0 file's layout on a tensor-parallel context (round 6): the classifier's raw q6_K super-blocks are split
    by vocabulary rows like every other classifier (llmk_upload on a rank keeps its V / P rows), each rank's gemv_q6k_kernel writes
    its slice of the logits.  E = 512: two super-blocks per row."""
    s = gguf.LlamaShape(512, 1024, 2, 4, 4, 640, 32)
    fw = gguf.with_q6k_classifier(gguf.synth_fused(s, 99, 2))
    assert fw.cls_type == 14
    P = 2
    ranks = [llmk.Llmk(fw, tp_rank=r, tp_size=P) for r in range(P)]
    n = 8
    toks, logits = tp_generate(ranks, n, s)
    ot, ol = Oracle(fw.as_f32(), "omp").generate(n)
    assert rel_err(logits, ol).max() <= REL_TOL
    margin = np.sort(ol, axis=1)
    safe = (margin[:, -1] - margin[:, -2]) > 4 * REL_TOL * np.abs(ol).max()
    assert np.array_equal(toks[safe], ot[safe])
    for m in ranks:
        m.close()


def test_ranks_fed_only_their_own_rows_hold_the_same_shards(gguf):
    """llmk_upload_rows on a tensor-parallel ctx takes row numbers of the FULL tensor and keeps the part its shard holds:
    a rank handed only its own rows (what host/gguf_loader.f90 stream_ggml_weights reads from the file), in pieces, ends
    with the same shard as a rank handed whole layers -- same logits bit for bit."""
    s = gguf.LlamaShape(128, 384, 2, 4, 4, 512, 32)
    fw = gguf.synth_fused(s, 99, 2)
    P = 2
    whole = [llmk.Llmk(fw, tp_rank=r, tp_size=P) for r in range(P)]
    E, H, KV, V, L = s.emb_dim, s.hidden_dim, s.kv_dim, s.vocab_size, s.n_layers
    T = {k: k for k in llmk.TENSOR_IDS}
    own = []
    for r in range(P):
        m = llmk.Llmk.create_empty(s, 2, tp_rank=r, tp_size=P)

        def up(name, layer, row0, arr, typ, m=m):
            m.upload_rows(name, layer, row0, arr, typ)
        for v0 in range(0, V, 100):                                   # replicated, in ragged chunks
            up(T["token_embedding_table"], 0, v0, fw.token_embedding_table[v0:v0 + 100], 0)
        up(T["rms_final_weight"], 0, 0, fw.rms_final_weight, 0)
        Vl, Eq, KVl, Hl = V // P, E // P, KV // P, H // P
        up(T["wcls"], 0, r * Vl, fw.wcls[r * Vl:(r + 1) * Vl], 2)
        up(T["wcls"], 0, ((r + 1) % P) * Vl, fw.wcls[((r + 1) % P) * Vl:((r + 1) % P) * Vl + 7], 2)   # foreign rows: ignored
        for l in range(L):
            up(T["rms_att_weight"], l, 0, fw.rms_att_weight[l], 0)
            up(T["rms_ffn_weight"], l, 0, fw.rms_ffn_weight[l], 0)
            for base, n in ((r * Eq, Eq), (E + r * KVl, KVl), (E + KV + r * KVl, KVl)):
                half = n // 2
                up(T["wqkv"], l, base, fw.wqkv[l, base:base + half], 2)
                up(T["wqkv"], l, base + half, fw.wqkv[l, base + half:base + n], 2)
            up(T["wo"], l, 0, fw.wo[l, :E // 2], 2)
            up(T["wo"], l, E // 2, fw.wo[l, E // 2:], 2)
            for base in (r * Hl, H + r * Hl):
                up(T["w13"], l, base, fw.w13[l, base:base + Hl], 2)
            up(T["w2"], l, 0, fw.w2[l], 2)
        own.append(m)
    n = 6
    a_t, a_l = tp_generate(whole, n, s)
    b_t, b_l = tp_generate(own, n, s)
    assert np.array_equal(a_l, b_l) and np.array_equal(a_t, b_t)
    for m in whole + own:
        m.close()


def test_shards_hold_only_their_share_and_bad_splits_are_rejected(gguf):
    s = gguf.SHAPES["tiny-gqa"]                   # nkv = 2
    fw = gguf.synth_fused(s, 1)
    with pytest.raises(llmk.LlmkError):
        llmk.Llmk(fw, tp_rank=0, tp_size=4)       # 2 kv heads cannot be split 4 ways
    with pytest.raises(llmk.LlmkError):
        llmk.Llmk(fw, tp_rank=2, tp_size=2)       # rank out of range
    m = llmk.Llmk(fw, tp_rank=1, tp_size=2)
    with pytest.raises(llmk.LlmkError):
        m.forward(2, 1)                           # no communicator: LLMK_E_COMM, never a silent single-rank answer
    m.close()


def test_rccl_path_with_one_rank(gguf):
    """ncclCommInitRank + ncclAllReduce/ncclAllGather on the ctx stream (1 rank: the sums are identities)."""
    g = load_golden("tiny-hs64")
    s = gguf.SHAPES["tiny-hs64"]
    fw = gguf.synth_fused(s, int(g["seed"]))
    m = llmk.Llmk(fw, tp_rank=0, tp_size=1)
    m.tp_init_comm(llmk.Llmk.tp_unique_id())
    toks, logits = m.generate(12)
    assert rel_err(logits, g["logits"][:12]).max() <= REL_TOL
    assert np.array_equal(toks, g["tokens"][:12])
    t2, _ = m.generate(12, want_logits=False, greedy_on_device=True)
    assert np.array_equal(t2, g["tokens"][:12])
    m.close()


def _in_threads(ranks, fn):
    """one host thread per rank (ctypes releases the GIL inside llmk_forward): the ranks' kernels run concurrently"""
    import threading
    out, errs = [None] * len(ranks), []

    def run(i):
        try:
            out[i] = fn(ranks[i])
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(i,)) for i in range(len(ranks))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    return out


@pytest.mark.parametrize("shape,P", [("tiny-gqa", 2), ("tiny-mha", 2), ("tk-small", 2)])
def test_one_shot_peer_memory_collectives_virtual_ranks(shape, P, gguf):
    """csrc/tp_p2p.h on hardware: P contexts on this GPU, each with its own stream and inbox, connected with
    llmk_tp_p2p_connect_local; every rank runs llmk_forward in its own host thread and the all-reduce / all-gather
    kernels exchange granules for real (through HBM instead of xGMI).  Reference goldens; all ranks bit-identical.
    (Two ranks per process: the ranks' kernels wait for each other, so each needs its own hardware queue, and one process
    gets four.  More ranks on one GPU run as separate processes, below -- the deployment shape anyway.)"""
    g = load_golden(shape)
    s = gguf.SHAPES[shape]
    fw = gguf.synth_fused(s, int(g["seed"]))
    ranks = [llmk.Llmk(fw, tp_rank=r, tp_size=P) for r in range(P)]
    llmk.Llmk.tp_p2p_connect_local(ranks)
    n = int(g["n"])
    assert _in_threads(ranks, lambda m: m.tp_p2p_selftest(8)) == [0] * P      # every collective on known integers first
    res = _in_threads(ranks, lambda m: m.generate(n))
    for toks, logits in res:
        assert rel_err(logits, g["logits"]).max() <= REL_TOL
        assert np.array_equal(toks, g["tokens"])
        assert np.array_equal(logits, res[0][1])          # rank-order sums: the replicated stream is bit-identical
    res2 = _in_threads(ranks, lambda m: m.generate(n, want_logits=False, greedy_on_device=True))
    for toks, _ in res2:
        assert np.array_equal(toks, g["tokens"])
    for m in ranks:
        m.close()


def test_failed_self_test_falls_back_to_rccl(gguf):
    """What a host does when the peer-memory self-test fails on its node: llmk_tp_p2p_disable + llmk_tp_init_comm, and the
    same ctx decodes over RCCL (1 rank here: a multi-rank communicator needs one GPU per rank)."""
    g = load_golden("tiny-hs64")
    fw = gguf.synth_fused(gguf.SHAPES["tiny-hs64"], int(g["seed"]))
    m = llmk.Llmk(fw, tp_rank=0, tp_size=1)
    assert m.tp_p2p_selftest(1) != 0                  # not connected: an error code, never a pass
    with pytest.raises(llmk.LlmkError):
        llmk.Llmk.tp_p2p_connect_local([m])           # one rank has nobody to exchange with
    m.tp_p2p_disable()
    m.tp_init_comm(llmk.Llmk.tp_unique_id())
    assert m.path() == 3
    toks, logits = m.generate(12)
    assert rel_err(logits, g["logits"][:12]).max() <= REL_TOL and np.array_equal(toks, g["tokens"][:12])
    m.close()


def test_one_shot_collectives_q4_0_virtual_ranks(gguf):
    s = gguf.LlamaShape(128, 384, 2, 4, 4, 512, 32)
    fw = gguf.synth_fused(s, 99, 2)
    ranks = [llmk.Llmk(fw, tp_rank=r, tp_size=2) for r in range(2)]
    llmk.Llmk.tp_p2p_connect_local(ranks)
    n = 8
    ot, ol = Oracle(fw.as_f32(), "omp").generate(n)
    res = _in_threads(ranks, lambda m: m.generate(n, prompt=ot.tolist()))
    for _, logits in res:
        assert rel_err(logits, ol).max() <= REL_TOL
    for m in ranks:
        m.close()


def _ipc_rank(rank, P, shape_name, seed, n, conn, device):
    """child process of test_two_processes_*: one rank, handles exchanged through pipes"""
    import numpy as np     # noqa: F811
    import llm_f90_amd     # noqa: F401
    from llm_f90_amd import llmk as lk
    from llm_f90_amd.tools import gguf as gg
    fw = gg.synth_fused(gg.SHAPES[shape_name], seed)
    m = lk.Llmk(fw, device=device, tp_rank=rank, tp_size=P)
    conn.send(m.tp_p2p_handle())
    handles = conn.recv()
    m.tp_p2p_connect(handles)
    toks, logits = m.generate(n)
    conn.send((toks, logits))
    m.close()


@pytest.mark.parametrize("P,same_device,tag", [(2, True, "tiny-gqa"), (4, True, "tiny-mha"), (2, False, "tiny-gqa"), (4, False, "tiny-mha")],
                         ids=["2-ranks-one-gpu", "4-ranks-one-gpu", "2-gpus", "4-gpus"])
def test_rank_processes_exchange_over_ipc_mapped_inboxes(P, same_device, tag, gguf):
    """The deployment shape: one PROCESS per rank, inboxes exported with hipIpcGetMemHandle and mapped by the peers.
    one-gpu: all ranks on device 0 (what this box has); N-gpus: devices 0..N-1 over xGMI (skipped without them)."""
    import multiprocessing as mp
    import torch
    if not same_device and torch.cuda.device_count() < P:
        pytest.skip(f"needs {P} GPUs")
    g = load_golden(tag)
    n = int(g["n"])
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(P)]
    procs = [ctx.Process(target=_ipc_rank, args=(r, P, tag, int(g["seed"]), n, pipes[r][1], 0 if same_device else r))
             for r in range(P)]
    for p in procs:
        p.start()

    def get(r, what):
        if not pipes[r][0].poll(120):                 # a hung or dead rank must fail the test, not stall it
            for p in procs:
                p.terminate()
            pytest.fail(f"rank {r} never delivered its {what}")
        return pipes[r][0].recv()
    handles = [get(r, "inbox handle") for r in range(P)]
    for r in range(P):
        pipes[r][0].send(handles)
    res = [get(r, "result") for r in range(P)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for toks, logits in res:
        assert rel_err(logits, g["logits"]).max() <= REL_TOL
        assert np.array_equal(toks, g["tokens"])
    for r in range(1, P):
        assert np.array_equal(res[0][1], res[r][1])
