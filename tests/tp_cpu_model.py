"""TEST INFRASTRUCTURE -- a numpy restatement of ONE tensor-parallel rank of the token pass, for the world-size-2 gloo test
(tests/test_tp_protocol_cpu.py).  Not product code: the product's ranks are llmk contexts on GPUs (csrc/llmk.hip,
llmk_create_tp); this file only re-plays, on CPU, WHAT each rank owns and WHERE the three exchanges sit, so that the
partitioning and the exchange protocol are covered when no multi-GPU node is available:

  rank r of P owns   kv heads [r*nkv/P, (r+1)*nkv/P) with their query heads and KV-cache rows,
                     hidden units [r*H/P, (r+1)*H/P) (gate and up rows of w1|w3, the same COLUMNS of w2),
                     the columns of wo that belong to its heads, vocabulary rows [r*V/P, (r+1)*V/P);
  exchanges          all-reduce(sum) of the partial E-vector after wo and after w2 (x += sum), all-gather of the logits.

Arithmetic follows the reference (/root/reference/llama2.f90:480-640: rmsnorm :450-457, RoPE :543-559 with the 2j+1
exponent and 1-based pos, GQA attention :572-598, SwiGLU :615-616) in float32.
"""
import numpy as np

f32 = np.float32


def rmsnorm(x, w):
    return (x * w / np.sqrt(np.dot(x, x) / f32(x.size) + f32(1e-5))).astype(f32)


class TpRank:
    def __init__(self, fw, rank, P, allreduce, allgather):
        """fw: tools.gguf.FusedWeights (f32); allreduce(v) -> sum over ranks; allgather(v) -> concatenation in rank order"""
        s = fw.shape
        self.s, self.r, self.P = s, rank, P
        self.allreduce, self.allgather = allreduce, allgather
        E, H, hs, KV = s.emb_dim, s.hidden_dim, s.head_size, s.kv_dim
        nhl, nkvl, Hl, Vl = s.n_heads // P, s.n_kv_heads // P, H // P, s.vocab_size // P
        self.nhl, self.nkvl, self.Hl, self.Vl = nhl, nkvl, Hl, Vl
        q0, k0 = rank * nhl * hs, rank * nkvl * hs
        self.emb = fw.token_embedding_table
        self.rms_att, self.rms_ffn, self.rms_final = fw.rms_att_weight, fw.rms_ffn_weight, fw.rms_final_weight
        # the cuts llmk_upload makes for rank r (csrc/llmk.hip upload_layer_sharded)
        self.wq = fw.wqkv[:, q0:q0 + nhl * hs]
        self.wk = fw.wqkv[:, E + k0:E + k0 + nkvl * hs]
        self.wv = fw.wqkv[:, E + KV + k0:E + KV + k0 + nkvl * hs]
        self.wo = fw.wo[:, :, q0:q0 + nhl * hs]                         # all E rows, this rank's input columns
        self.w1 = fw.w13[:, rank * Hl:(rank + 1) * Hl]
        self.w3 = fw.w13[:, H + rank * Hl:H + (rank + 1) * Hl]
        self.w2 = fw.w2[:, :, rank * Hl:(rank + 1) * Hl]
        self.wcls = fw.wcls[rank * Vl:(rank + 1) * Vl]
        self.kc = np.zeros((s.n_layers, s.seq_len, nkvl * hs), f32)
        self.vc = np.zeros_like(self.kc)
        j = np.arange(hs // 2, dtype=f32)
        self.freq = (f32(1.0) / np.power(f32(10000.0), (2 * j + 1) / f32(hs), dtype=f32)).astype(f32)

    def rope(self, v, pos):
        hs = self.s.head_size
        v = v.reshape(-1, hs // 2, 2).copy()
        ang = (f32(pos) * self.freq).astype(f32)
        c, sn = np.cos(ang).astype(f32), np.sin(ang).astype(f32)
        a, b = v[:, :, 0].copy(), v[:, :, 1].copy()
        v[:, :, 0] = a * c - b * sn
        v[:, :, 1] = a * sn + b * c
        return v.reshape(-1)

    def forward(self, token, pos):
        """token, pos 1-based as at llama2.f90:380; returns the FULL logits (after the all-gather)"""
        s, hs = self.s, self.s.head_size
        kv_mul = s.n_heads // s.n_kv_heads
        x = self.emb[token - 1].astype(f32).copy()
        for l in range(s.n_layers):
            xb = rmsnorm(x, self.rms_att[l])
            q = self.rope(self.wq[l] @ xb, pos)
            k = self.rope(self.wk[l] @ xb, pos)
            self.kc[l, pos - 1] = k
            self.vc[l, pos - 1] = self.wv[l] @ xb
            att = np.empty(self.nhl * hs, f32)
            for h in range(self.nhl):
                g = h // kv_mul                                          # local kv head
                K = self.kc[l, :pos, g * hs:(g + 1) * hs]
                V = self.vc[l, :pos, g * hs:(g + 1) * hs]
                sc = (K @ q[h * hs:(h + 1) * hs]) / np.sqrt(f32(hs))
                p = np.exp(sc - sc.max()).astype(f32)
                p /= p.sum()
                att[h * hs:(h + 1) * hs] = p @ V
            x = x + self.allreduce((self.wo[l] @ att).astype(f32))      # exchange 1   (llama2.f90:603-605)
            xb = rmsnorm(x, self.rms_ffn[l])
            gte, up = self.w1[l] @ xb, self.w3[l] @ xb
            hb = (gte * (f32(1.0) / (f32(1.0) + np.exp(-gte))) * up).astype(f32)
            x = x + self.allreduce((self.w2[l] @ hb).astype(f32))       # exchange 2   (:618-620)
        x = rmsnorm(x, self.rms_final)
        return self.allgather((self.wcls @ x).astype(f32))              # exchange 3   (:634-636)


# ---- the peer-memory exchange protocol of csrc/tp_p2p.h as a state machine, checked over EVERY interleaving ----------------
def tp_program(P, rank, tokens, ncalls):
    """The memory operations ONE element's threads of rank `rank` perform over `tokens` token passes, in program order
    (csrc/tp_p2p.h): per all-reduce call its P-1 sends then its P-1 blocking reads in rank order, then the all-gather --
    for one logit column per owner: the owner's P-1 sends, everybody else's one blocking read.
    ops: ("send", dst, cell, epoch, value) | ("recv", cell, epoch, expected value); a cell is the address of one granule in
    a rank's inbox: ("ar", half, src) or ("ag", half, column)."""
    ops = []
    per = ncalls + 1
    for serial in range(1, tokens + 1):
        for call in range(ncalls):
            epoch, half = serial * per + call + 1, call & 1
            val = ("part", rank, serial, call)
            for r in range(1, P):
                ops.append(("send", (rank + r) % P, ("ar", half, rank), epoch, val))
            for r in range(P):
                if r != rank:
                    ops.append(("recv", ("ar", half, r), epoch, ("part", r, serial, call)))
        epoch, half = serial * per + ncalls + 1, serial & 1
        for col in range(P):                       # one logit column per owner
            if col == rank:
                for r in range(1, P):
                    ops.append(("send", (rank + r) % P, ("ag", half, col), epoch, ("logit", col, serial)))
        for col in range(P):
            if col != rank:
                ops.append(("recv", ("ag", half, col), epoch, ("logit", col, serial)))
    return ops


def check_interleavings(P, tokens=3, ncalls=2, halves=2):
    """Depth-first search over every interleaving of the P ranks' operations (a blocked read is simply not enabled).
    Returns None when every reachable state is sound, else a description of the first violation:
      * a read whose tag matches but whose value is not the one its exchange must deliver (a granule was overwritten by a
        LATER exchange that reuses the cell AND the epoch -- cannot happen with unique epochs, checked anyway);
      * a deadlock: some rank blocked forever, i.e. the granule it waits for was overwritten before it was read (the
        failure the two alternating halves exist to exclude; halves=1 must produce it)."""
    progs = [tp_program(P, r, tokens, ncalls) for r in range(P)]
    if halves == 1:                                # the broken variant: every exchange in half 0
        progs = [[(o[0], o[1], (o[2][0], 0, o[2][2]), *o[3:]) if o[0] == "send" else (o[0], (o[1][0], 0, o[1][2]), *o[2:])
                  for o in p] for p in progs]
    cells = sorted({(o[1], o[2]) for p in progs for o in p if o[0] == "send"})
    index = {c: i for i, c in enumerate(cells)}
    start = (tuple([0] * P), tuple([None] * len(cells)))
    seen, stack = {start}, [start]
    while stack:
        pcs, mem = stack.pop()
        progressed = False
        for r in range(P):
            if pcs[r] == len(progs[r]):
                continue
            op = progs[r][pcs[r]]
            if op[0] == "send":
                m = list(mem)
                m[index[(op[1], op[2])]] = (op[3], op[4])
                nxt = (pcs[:r] + (pcs[r] + 1,) + pcs[r + 1:], tuple(m))
            else:
                g = mem[index[(r, op[1])]] if (r, op[1]) in index else None
                if g is None or g[0] != op[2]:
                    continue                       # tag is not this exchange's epoch: the thread keeps polling
                if g[1] != op[3]:
                    return f"rank {r} read {g} where {op} was due"
                nxt = (pcs[:r] + (pcs[r] + 1,) + pcs[r + 1:], mem)
            progressed = True
            if nxt not in seen:
                seen.add(nxt)
                stack.append(nxt)
        if not progressed and any(pcs[r] < len(progs[r]) for r in range(P)):
            blocked = {r: progs[r][pcs[r]] for r in range(P) if pcs[r] < len(progs[r])}
            return f"deadlock: {blocked}"
    return None
