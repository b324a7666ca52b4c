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
