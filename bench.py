#!/usr/bin/env python3
"""bench.py -- tokens/sec of the llm.f90 decode hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" = one token through the whole forward pass (`transformer`, /root/reference/llama2.f90:480-640)
via the C-ABI entry llmk_forward, weights already resident in HBM; logits (V floats) are handed back
to the host every step exactly as the reference loop consumes them (llama2.f90:380-393), and the next
token is the greedy argmax.  Workload at N=1: BASELINE.json configs[1], TinyLlama-1.1B f32 decode,
synthetic weights (no model file offline), positions 1..W+K like `./llm -n 256 -t 0`.

N>1: the path does not shard for a 1.1B model ("replicas only"): N independent
replicas, one process per GPU (launched by torch.distributed.run), barrier on both sides of the
timed region, max over ranks, value = total tokens of all replicas / that time ("weak").

Prints ONE JSON line on rank 0, with `roofline` for the dominant kernel -- the persistent whole-token kernel
(token_kernel.h: its launch IS the hot path) wherever it is instantiated, the fused w1|w3 GEMV on the multi-kernel path
(--multi-kernel, other shapes) -- and `cpu_baseline` (the real reference binary from oracle/_ref when present, else the
C port).

Short runs (--steps < 64, e.g. the driver's --steps 20 --warmup 5: a 14 ms timed region) repeat the whole
warm-up + K-step loop REPEATS times on a reset context and report the MEDIAN repetition (every repetition's rate is in
`value_all`): one scheduling hiccup on the host is 3-5 % of a 14 ms region.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import llm_f90_amd  # noqa: E402,F401
from llm_f90_amd import llmk  # noqa: E402
from llm_f90_amd.replicas import Replicas  # noqa: E402
from llm_f90_amd.tools import gguf  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
SEED = 20260928
KERNEL_NAMES = {0: "qkv", 1: "attn", 2: "wo", 3: "w13", 4: "w2", 5: "cls"}


def bytes_per_token(s: gguf.LlamaShape, wtype: int, mean_pos: float) -> float:
    """Algorithmic HBM bytes of one token (SURVEY.md 8d / BASELINE.md section 3)."""
    per_w = {0: 4.0, 1: 2.0, 2: 18.0 / 32.0}[wtype]
    E, L, KV, V = s.emb_dim, s.n_layers, s.kv_dim, s.vocab_size
    return (s.matmul_params() * per_w            # every matmul weight once
            + (2 * L + 1) * E * 4                # rmsnorm gains
            + E * 4                              # embedding row
            + 2 * L * KV * 4 * mean_pos          # KV-cache read
            + 2 * L * KV * 4                     # KV-cache write
            + V * 4)                             # logits write


def pmc_traffic(kernel_substr: str, shape_name: str, type_name: str):
    """HBM read bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC pass OF THIS
    workload (profiles/rNN*_<shape>_<type>_pmc_fetch_size.csv: FETCH_SIZE x 2 x 1024, the gfx950 correction of
    MI355X_MICROARCH.md); None when no profile of this shape and weight type is committed.  (Round-1 files carry no
    workload in their name: they are all TinyLlama f32.)"""
    import csv
    import glob
    pat = re.compile(rf"r\d+[a-z]?_{re.escape(shape_name)}_{re.escape(type_name)}_pmc_fetch_size\.csv$")   # not r04_prefill512_<shape>_...
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_size.csv")) if pat.match(os.path.basename(f)))
    if not files and (shape_name, type_name) == ("tinyllama", "f32"):
        files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r01*_pmc_fetch_size.csv")))
    if not files:
        return None
    for row in csv.DictReader(open(files[-1])):
        if kernel_substr in row["kernel"]:
            return float(row["avg_FETCH_SIZE"]) * 2 * 1024
    return None


def transcript_ids(text: bytes, V: int):
    """1-based token ids of a CLI transcript over tools.gguf.vocab_strings (unique printable strings)."""
    ids, i = [], 0
    while i < len(text):
        m = re.match(rb"<(\d{5})>", text[i:])
        if m:
            ids.append(int(m.group(1)) + 1); i += 7
        elif text.startswith(b"<unk>", i):
            ids.append(1); i += 5
        elif text.startswith(b"</s>", i):
            ids.append(3); i += 4
        elif text.startswith(b"<s>", i):
            ids.append(2); i += 3
        else:
            ids.append(3 + text[i] - 32 + 1); i += 1
    return ids


def cpu_baseline(fw, shape_name: str, wtype: int, n_ref: int = 128, gguf_path: str = None):
    """Reference timed on this box's host cores, 1 thread, bounded sample (10-30 s of CPU work)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "llm_ref")
    if shape_name == "tinyllama" and wtype == 0 and os.path.exists(ref) and gguf_path:
        # the REAL reference (unmodified llama2.f90, dims hard-coded to TinyLlama) on the same weights
        try:
            cmd = [ref, "-m", gguf_path, "-n", str(n_ref), "-t", "0"]
            if subprocess.run(["which", "taskset"], capture_output=True).returncode == 0:
                cmd = ["taskset", "-c", "0"] + cmd
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, timeout=600, cwd=os.path.dirname(gguf_path))
            wall = time.perf_counter() - t0
            m = re.search(rb"([0-9.Ee+-]+)\s*tokens/second", r.stdout)
            if r.returncode == 0 and m:
                lines = r.stdout.split(b"\n")          # line 0: " data offset ...", line 1: the generated text
                return {"_ids": transcript_ids(lines[1].rstrip(b" "), fw.shape.vocab_size) if len(lines) > 1 else [],
                        "value": float(m.group(1)), "unit": "tokens/s", "cores": 1, "kind": "reference",
                        # the reference's own clock is real(int32 ms) (llama2.f90:417-423): a float of a ~1e9 count has a
                        # 64-128 ms quantum, i.e. +-1 % of this ~11 s sample; the subprocess's wall time (with the 4.4 GB load
                        # and the first token in it) brackets it from below
                        "clock": "the reference's own print: real(4) of an int32 ms count, 64-128 ms quantum = +-1 % of the sample",
                        "lower_bound_from_subprocess_wall": round((n_ref - 1) / wall, 3),
                        "sample": f"oracle/_ref/llm_ref (real reference, amdflang -O3 -march=native -ffast-math "
                                  f"-funroll-loops) -n {n_ref} -t 0 on the same synthetic GGUF, 1 thread pinned; "
                                  f"host has {os.cpu_count()} logical cores"}
        except Exception as e:  # fall through to the port
            sys.stderr.write(f"[bench] reference baseline failed: {e}\n")
    from oracle.oracle import Oracle
    o = Oracle(fw.as_f32(), "fast")
    n = 6 if fw.shape.matmul_params() > 5e8 else 64
    o.forward(2, 1)
    t0 = time.perf_counter()
    tok = 2
    for pos in range(2, n + 2):
        tok = int(np.argmax(o.forward(tok, pos))) + 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "tokens/s", "cores": 1, "kind": "port",
            "sample": f"oracle/llm_oracle.c (gcc -O3 -march=native -ffast-math -funroll-loops), {n} tokens, 1 thread; "
                      f"host has {os.cpu_count()} logical cores"}


def fortran_host(gguf_path: str, V: int, n: int, gpu_ids, device: int = 0, runs: int = 3):
    """The drop-in ITSELF (round-4 verdict, item 3): `llm.f90_amd/host/llm` -- the Fortran program that replaces
    llama2.f90's main, calling libllmk.so through ISO_C_BINDING -- on the same synthetic GGUF, `-n n -t 0`, its own
    `tokens/second` line (the reference's convention, llama2.f90:406: (n - 1) / (t_end - t_after_first_token)), with the
    host consumer (maxloc on the logits, like the reference) and with `--device-argmax`.  Outside the timed region; the
    median of `runs` processes each (a process = load 4.4 GB + upload + n tokens)."""
    llm = os.path.join(ROOT, "llm.f90_amd", "host", "llm")
    if not os.path.exists(llm):
        return {"error": "llm.f90_amd/host/llm not built"}
    out = {"n": n, "command": f"llm -m <synthetic gguf> -n {n} -t 0 [-d {device}]", "runs": runs,
           "clock": "8-byte system_clock ticks subtracted before the conversion to real(4) (host/llm.f90 elapsed_ms)"}
    for key, extra in (("tok_s", []), ("tok_s_device_argmax", ["--device-argmax"])):
        vals, ids, wall = [], None, []
        for _ in range(runs):
            t0 = time.perf_counter()
            r = subprocess.run([llm, "-m", gguf_path, "-n", str(n), "-t", "0", "-d", str(device)] + extra, capture_output=True,
                               timeout=600, cwd=os.path.dirname(gguf_path))
            wall.append(time.perf_counter() - t0)
            m = re.search(rb"([0-9.Ee+-]+)\s*tokens/second", r.stdout)
            if r.returncode != 0 or not m:
                return {"error": (r.stdout[-300:] + r.stderr[-300:]).decode(errors="replace")}
            vals.append(float(m.group(1)))
            ids = transcript_ids(r.stdout.split(b"\n")[1].rstrip(b" "), V)
        out[key] = sorted(vals)[len(vals) // 2]
        out[key + "_all"] = [round(v, 1) for v in vals]
        k = min(len(ids), len(gpu_ids))
        out[("ids_match" if not extra else "ids_match_device_argmax")] = f"{sum(x == y for x, y in zip(ids[:k], gpu_ids[:k]))}/{k}"
        out["process_wall_s"] = round(sorted(wall)[len(wall) // 2], 2)
    return out


class _ShapeOnly:
    """what llmk.Llmk needs to create a ctx without weights (they are streamed in layer by layer)"""
    def __init__(self, shape, wtype):
        self.shape, self.ggml_type = shape, wtype


def build_streamed(shape, wtype, fw, device, flags, tp_rank, tp_size, rep, collective="p2p", cls_q6k=False):
    """Create the ctx first, then upload layer by layer (llmk_upload_rows hands over FULL layers; a
    tensor-parallel ctx keeps only its shard).  q4_0 weights of the big shapes are generated directly in
    block format.  With tp_size > 1 the ranks meet over torch.distributed (a side channel only): the 64-byte inbox
    handles of the one-shot peer-memory collectives are all-gathered (collective "p2p"), or rank 0's RCCL unique id is
    broadcast (collective "rccl", the library baseline)."""
    s = shape
    E, H, L, KV, V = s.emb_dim, s.hidden_dim, s.n_layers, s.kv_dim, s.vocab_size
    m = llmk.Llmk.create_empty(s, wtype, device, flags, tp_rank, tp_size)
    if cls_q6k:       # the classifier as raw q6_K super-blocks: what a stock llama.cpp q4_0 file holds (csrc/q6k.h)
        m.set_tensor_type("wcls", gguf.GGML_Q6_K)
    if (tp_size > 1 or rep is not None) and collective == "p2p" and tp_size > 1:
        handles = [None] * tp_size
        rep.dist.all_gather_object(handles, m.tp_p2p_handle())
        m.tp_p2p_connect(handles)
        # the peer-memory path proves itself on THIS node before the first token; unless every rank passes, all ranks
        # drop it together and run over RCCL (the line's "parallelism" says which collective actually ran)
        verdicts = [None] * tp_size
        rep.dist.all_gather_object(verdicts, m.tp_p2p_selftest(64))
        m.p2p_selftest = verdicts
        if any(verdicts):
            m.tp_p2p_disable()
            uids = [None] * tp_size
            rep.dist.all_gather_object(uids, llmk.Llmk.tp_unique_id() if tp_rank == 0 else None)
            m.tp_init_comm(uids[0])
    elif collective == "none":
        pass            # an unconnected rank: kernel timing only (tests/host_tools/tp_rank_time.py)
    elif tp_size > 1 or rep is not None:
        uid = llmk.Llmk.tp_unique_id() if tp_rank == 0 else bytes(128)
        if rep is not None and rep.dist is not None:
            import torch
            t = torch.tensor(list(uid), dtype=torch.uint8, device=rep.device)
            rep.dist.broadcast(t, src=0)
            uid = bytes(t.cpu().tolist())
        m.tp_init_comm(uid)

    T = {k: k for k in llmk.TENSOR_IDS}

    def up(name, layer, arr, typ):
        m.upload_rows(name, layer, 0, arr, typ)
    names = gguf.tensor_names(s)
    idx = {n: i for i, (n, _, _) in enumerate(names)}
    if fw is not None:
        up(T["token_embedding_table"], 0, fw.token_embedding_table, 0)
        up(T["rms_final_weight"], 0, fw.rms_final_weight, 0)
        if cls_q6k and fw.cls_type != gguf.GGML_Q6_K:
            fw = gguf.with_q6k_classifier(fw)
        up(T["wcls"], 0, fw.wcls, fw.cls_type if cls_q6k else wtype)
        for l in range(L):
            up(T["rms_att_weight"], l, fw.rms_att_weight[l], 0)
            up(T["rms_ffn_weight"], l, fw.rms_ffn_weight[l], 0)
            for k in ("wqkv", "wo", "w13", "w2"):
                up(T[k], l, getattr(fw, k)[l], wtype)
        return m
    if wtype == 1:
        # f16 matrices of a big shape (Llama-2-7B f16: 13.2 GB), tensor by tensor: synth_fused's values, never more than one on the host
        f16 = lambda name, rows, K: gguf.synth_tensor(s, SEED, idx[name], (rows, K), "mat").astype(np.float16)
        up(T["token_embedding_table"], 0, gguf.synth_tensor(s, SEED, idx["token_embd.weight"], (V, E), "emb"), 0)
        up(T["rms_final_weight"], 0, gguf.synth_tensor(s, SEED, idx["output_norm.weight"], (E,), "norm"), 0)
        up(T["wcls"], 0, f16("output.weight", V, E), 1)
        for l in range(L):
            pre = f"blk.{l}."
            up(T["rms_att_weight"], l, gguf.synth_tensor(s, SEED, idx[pre + "attn_norm.weight"], (E,), "norm"), 0)
            up(T["rms_ffn_weight"], l, gguf.synth_tensor(s, SEED, idx[pre + "ffn_norm.weight"], (E,), "norm"), 0)
            up(T["wqkv"], l, np.concatenate([f16(pre + "attn_q.weight", E, E), f16(pre + "attn_k.weight", KV, E),
                                             f16(pre + "attn_v.weight", KV, E)]), 1)
            up(T["wo"], l, f16(pre + "attn_output.weight", E, E), 1)
            up(T["w13"], l, np.concatenate([f16(pre + "ffn_gate.weight", H, E), f16(pre + "ffn_up.weight", H, E)]), 1)
            up(T["w2"], l, f16(pre + "ffn_down.weight", E, H), 1)
        return m
    q4 = lambda name, rows, K: gguf.synth_q4_rows(SEED, idx[name], rows, K)
    up(T["token_embedding_table"], 0, gguf.synth_tensor(s, SEED, idx["token_embd.weight"], (V, E), "emb"), 0)
    up(T["rms_final_weight"], 0, gguf.synth_tensor(s, SEED, idx["output_norm.weight"], (E,), "norm"), 0)
    if cls_q6k:
        up(T["wcls"], 0, gguf.quantize_q6_K(gguf.dequantize_q4_0(q4("output.weight", V, E), E)).reshape(V, -1), gguf.GGML_Q6_K)
    else:
        up(T["wcls"], 0, q4("output.weight", V, E), 2)
    for l in range(L):
        pre = f"blk.{l}."
        up(T["rms_att_weight"], l, gguf.synth_tensor(s, SEED, idx[pre + "attn_norm.weight"], (E,), "norm"), 0)
        up(T["rms_ffn_weight"], l, gguf.synth_tensor(s, SEED, idx[pre + "ffn_norm.weight"], (E,), "norm"), 0)
        up(T["wqkv"], l, np.concatenate([q4(pre + "attn_q.weight", E, E), q4(pre + "attn_k.weight", KV, E),
                                         q4(pre + "attn_v.weight", KV, E)]), 2)
        up(T["wo"], l, q4(pre + "attn_output.weight", E, E), 2)
        up(T["w13"], l, np.concatenate([q4(pre + "ffn_gate.weight", H, E), q4(pre + "ffn_up.weight", H, E)]), 2)
        up(T["w2"], l, q4(pre + "ffn_down.weight", E, H), 2)
    return m


def prefill_line(a):
    """`--prefill N`: prompt tokens/s of llmk_prefill (batched MFMA GEMMs) next to the token-by-token loop it replaces.
    Roofline of its dominant kernel, the w1|w3 GEMM at 128 positions: f32 matrix-core peak (157.3 TFLOP/s,
    MI355X_MICROARCH.md; f16 and q4_0 weights are converted exactly to f32 A operands, same instruction) -- at 128 positions
    per pass the GEMM is MFMA-bound (f32: 2*128 flop per 4-byte weight = 64 flop/B, the matrix rate needs 2.5 TB/s of weights)."""
    shape = gguf.SHAPES[a.shape]
    n = a.prefill
    if n < 1 or n + 1 > shape.seq_len:
        raise SystemExit("--prefill N: N < seq_len")
    wt = {"f32": 0, "f16": 1, "q4_0": 2}[a.type]
    m = llmk.Llmk(gguf.synth_fused(shape, SEED, wt))
    rng = np.random.default_rng(SEED)
    prompt = [2] + (rng.integers(3, shape.vocab_size, n - 1) + 1).tolist()
    for _ in range(max(1, a.warmup // 4)):
        m.reset(); m.prefill(prompt, 1)
    reps = max(3, min(20, 2000 // n))
    t0 = time.perf_counter()
    for _ in range(reps):
        lg = m.prefill(prompt, 1)
    dt = (time.perf_counter() - t0) / reps
    m.reset()
    t0 = time.perf_counter()
    nseq = min(n, 64)
    for pos in range(1, nseq + 1):
        m.forward(prompt[pos - 1], pos)
    dt_seq = (time.perf_counter() - t0) / nseq
    ms, wbytes = m.time_kernel(7, 2 * shape.n_layers)
    flop = 2.0 * 128 * 2 * shape.hidden_dim * shape.emb_dim
    kname = f"pf_gemm_kernel<8, {a.type}>"
    # all three weight types run on v_mfma_f32_16x16x32_f16 (prefill.h pf_gemm_h_kernel, activations -- and f32 / q4_0 weights -- as two f16 pieces) unless
    # LLMK_PF_F32_MFMA=1: 256 flop per 2-byte weight at 128 positions is below that instruction's ridge (2,500 TFLOP/s / 8 TB/s
    # = 312 flop/B; q4_0: 455 flop/B, above it, but the kernel's own limit is its VALU work), so the kernel is priced against
    # HBM: weight bytes / launch
    hm = os.environ.get("LLMK_PF_F32_MFMA", "0")[:1] != "1"
    out = {"metric": f"prompt tokens/sec {a.shape} prefill", "value": n / dt, "unit": "tokens/s", "n_gpus": 1,
           "steps": reps, "warmup": max(1, a.warmup // 4), "ms_per_step": 1000.0 * dt, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": a.type, "data": "synthetic",
           "config": {"workload": f"{a.shape} {a.type} prefill of a {n}-token prompt (llmk_prefill, 128 positions per pass)",
                      "token_by_token_tok_s": 1.0 / dt_seq, "speedup": (n / dt) * dt_seq, "seed": SEED,
                      "arithmetic": f"{a.type} weights converted exactly to f32 MFMA operands, f32 accumulate"},
           "roofline": {"bound": "mfma", "kernel": f"{kname} (w1|w3, 128 positions, v_mfma_f32_16x16x4_f32)",
                        "achieved": flop / (ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                        "frac": flop / (ms * 1e-3) / 1e12 / 157.3, "traffic": pmc_traffic("pf_gemm_kernel", "prefill_w13", a.type),
                        "us_per_launch": ms * 1000.0,
                        "flop_per_launch": flop, "weight_bytes_per_launch": wbytes,
                        "hbm_GBps": wbytes / (ms * 1e-3) / 1e9}}
    if hm:
        wdesc = {"f16": "f16 weights as they are", "q4_0": "q4_0 weights (n - 8) d formed exactly in registers, as two exact f16 pieces",
                 "f32": "f32 weights as two f16 pieces (|error| <= 2^-20 |w|, 2^-24 absolute below 2^-3)"}[a.type]
        out["config"]["arithmetic"] = (wdesc + " x f32 activations as two f16 pieces (hi + lo: |error| <= 2^-20 |x| for |x| >= 2^-3, "
                                       "2^-24 absolute below) on v_mfma_f32_16x16x32_f16, exact products, f32 accumulate")
        # Priced against BOTH rooflines (round-3 verdict): HBM with the weight bytes of a launch, and the f16 matrix
        # instruction's dense peak counting the instructions ISSUED -- two per 32-column chunk for f16 weights (x_hi, x_lo),
        # three for f32 / q4_0 weights (wh.x_hi + wl.x_hi + wh.x_lo).  `bound` names the nearer one; neither is near: the step
        # is bound by moving the activation tile (L2 -> registers -> LDS) and by per-launch latency.
        gbps = wbytes / (ms * 1e-3) / 1e9
        pieces = 2 if a.type == "f16" else 3
        issued = flop * pieces / (ms * 1e-3) / 1e12
        MFMA_F16_PEAK = 2500.0                    # dense TFLOP/s of v_mfma_f32_16x16x32_f16 (MI355X_MICROARCH.md)
        f_hbm, f_mfma = gbps / 8000.0, issued / MFMA_F16_PEAK
        # (q4_0 weights: the 128-row strip on eight waves of one row group each, llmk.hip pf_gemm_launch)
        kname = "pf_gemm_h_kernel<8, 1, 2, 8>" if a.type == "q4_0" else "pf_gemm_h_kernel<8, 2"
        traffic = pmc_traffic(kname, f"prefill512_{a.shape}", a.type)
        both = {"hbm": {"achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": f_hbm},
                "mfma": {"achieved": issued, "peak": MFMA_F16_PEAK, "unit": "TFLOP/s issued", "frac": f_mfma, "instructions_per_chunk": pieces,
                         "algorithmic_tflops": flop / (ms * 1e-3) / 1e12}}
        out["roofline"].update({"kernel": (f"pf_gemm_h_kernel<8, 1, q4_0, 8 waves>" if a.type == "q4_0" else f"pf_gemm_h_kernel<8, 2, {a.type}>")
                                          + " (w1|w3, 128 positions, v_mfma_f32_16x16x32_f16)",
                                "traffic": traffic, "tflops": flop / (ms * 1e-3) / 1e12, "both": both,
                                "note": "bound by neither roofline: see `both` and DESIGN.md section 3c"})
        if f_hbm >= f_mfma:
            out["roofline"].update({"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": f_hbm})
        else:
            out["roofline"].update({"bound": "mfma", "achieved": issued, "peak": MFMA_F16_PEAK, "unit": "TFLOP/s", "frac": f_mfma})
    print(json.dumps(out))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=248)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--shape", default="tinyllama", choices=sorted(gguf.SHAPES))
    ap.add_argument("--type", default="f32", choices=["f32", "f16", "q4_0"])
    ap.add_argument("--cls-q6k", action="store_true", help="the classifier as q6_K rows beside f16 / q4_0 matrices: the layout of a stock llama.cpp q4_0 file (output.weight stays q6_K)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fortran-host", action="store_true", help="skip the `fortran_host` leg (the ./llm CLI timed on the same GGUF)")
    ap.add_argument("--fortran-host", action="store_true", help="run the `fortran_host` leg even with --no-cpu-baseline")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (profiling aid)")
    ap.add_argument("--multi-kernel", action="store_true", help="5 launches per layer instead of the persistent token kernel")
    ap.add_argument("--tp", action="store_true",
                    help="tensor-parallel: ONE model sharded over the N ranks (RCCL all-reduce), the 70B configuration; "
                         "default for N>1 is N independent replicas")
    ap.add_argument("--tp-collective", default="p2p", choices=["p2p", "rccl"],
                    help="--tp: one-shot all-reduce over peer memory (csrc/tp_p2p.h, default) or ncclAllReduce (library baseline)")
    ap.add_argument("--greedy-on-device", action="store_true", help="auxiliary line: time llmk_decode_greedy (argmax on the device, launches enqueued back to back) instead of the host consumer")
    ap.add_argument("--repeats", type=int, default=0,
                    help="repetitions of the whole warm-up + K-step run (median reported); default 5 when --steps < 64, else 1")
    ap.add_argument("--prefill", type=int, default=0, metavar="N",
                    help="auxiliary line (not the headline metric): time llmk_prefill on an N-token prompt (SURVEY.md 8f rank 1)")
    a = ap.parse_args()
    if a.prefill:
        return prefill_line(a)

    # The contract is ONE JSON line on stdout.  RCCL (torch's, and ours in --tp mode) prints a version banner
    # through C stdio to fd 1 whenever a communicator is created, so everything this process or its
    # libraries print is sent to stderr and the result line alone goes to the real stdout.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rep = Replicas()
    rank, world, local = rep.rank, rep.world, rep.local
    if os.environ.get("LLMK_SHARE_GPU"):      # test aid: several replicas on one GPU (use with --multi-kernel)
        local = 0

    shape = gguf.SHAPES[a.shape]
    wtype = {"f32": 0, "f16": 1, "q4_0": 2}[a.type]
    K, W = a.steps, a.warmup
    if W + K > shape.seq_len:
        raise SystemExit(f"warmup+steps must be <= seq_len {shape.seq_len}")

    t0 = time.perf_counter()
    flags = (llmk.FLAG_NO_GRAPH if a.no_graph else 0) | (llmk.FLAG_MULTI_KERNEL if a.multi_kernel else 0)
    big = shape.matmul_params() > 3e9            # 7B / 70B: never materialise f32 weights on the host
    if big and (wtype == 0 or (wtype == 1 and (a.tp or shape.matmul_params() > 1e10))):
        raise SystemExit("the 7B/70B shapes are benchmarked as q4_0 (BASELINE.json configs[3], configs[4]); Llama-2-7B also as f16 (13.2 GB)")
    fw = None if big else gguf.synth_fused(shape, SEED, wtype)
    if a.cls_q6k and (wtype == 0 or shape.emb_dim % 256):
        raise SystemExit("--cls-q6k: a q6_K classifier beside f16 / q4_0 matrices, emb_dim a multiple of 256")
    if a.cls_q6k and fw is not None:
        fw = gguf.with_q6k_classifier(fw)
    t_gen = time.perf_counter() - t0
    if a.tp or big:
        m = build_streamed(shape, wtype, fw, local, flags, rank if a.tp else 0, world if a.tp else 1, rep if a.tp else None,
                           a.tp_collective, cls_q6k=a.cls_q6k)
    else:
        m = llmk.Llmk(fw, device=local, flags=flags)
    t_up = time.perf_counter() - t0 - t_gen

    barrier = rep.barrier

    step = m.forward_greedy if a.greedy_on_device else None
    lib, h, lg = llmk.lib(), m._h, m._logits
    import ctypes as C
    lgp = lg.ctypes.data_as(C.POINTER(C.c_float))
    nxt = C.c_int(0)

    def one_run():
        """W untimed warm-up positions, then EXACTLY K timed positions between barrier + sync on both sides
        (every llmk_forward returns after its own stream sync); max over ranks."""
        token = 2
        ids = []                     # the greedy transcript, compared with the reference's own at the end
        for pos in range(1, W + 1):  # untimed warm-up (first call also captures the hipGraph)
            if step:
                token = step(token, pos)
            else:
                token = int(np.argmax(m.forward(token, pos))) + 1
            ids.append(token)
        barrier()
        t_start = time.perf_counter()
        if step:   # the K positions as ONE call: launches enqueued back to back, the argmax never leaves the device
            ids += m.decode_greedy(token, W + 1, K).tolist()
            barrier()
            return rep.max_over_ranks(time.perf_counter() - t_start), ids
        for pos in range(W + 1, W + K + 1):
            if step:
                rc = lib.llmk_forward_greedy(h, token, pos, C.byref(nxt))
                token = nxt.value
            else:
                rc = lib.llmk_forward(h, token, pos, lgp)
                token = int(lg.argmax()) + 1
            ids.append(token)
            if rc:
                raise SystemExit(f"llmk_forward failed: {rc}")
        barrier()
        return rep.max_over_ranks(time.perf_counter() - t_start), ids

    # a 20-step timed region is 14 ms: repeat the whole run on a reset context and report the median repetition
    repeats = a.repeats if a.repeats > 0 else (5 if K < 64 else 1)
    runs = []
    for r in range(repeats):
        if r:
            m.reset()
        runs.append(one_run())
    order = sorted(range(repeats), key=lambda i: runs[i][0])
    elapsed, gpu_ids = runs[order[repeats // 2]]
    if any(ids != gpu_ids for _, ids in runs):
        raise SystemExit("greedy transcripts differ between repetitions")
    if not step and not np.all(np.isfinite(lg)):     # (--greedy-on-device never brings logits to the host)
        raise SystemExit("non-finite logits")

    tok_s = (1 if a.tp else world) * K / elapsed   # --tp: ONE model over all ranks; else N replicas
    mean_pos = W + (K + 1) / 2.0
    bpt = bytes_per_token(shape, wtype, mean_pos)

    out = {
        "metric": "tokens/sec TinyLlama-1.1B decode" if a.shape == "tinyllama" else f"tokens/sec {a.shape} decode",
        "value": tok_s, "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1000.0 * elapsed / K, "higher_is_better": True, "scaling": "strong" if a.tp else "weak", "vs_baseline": None,
        "dtype": a.type, "data": "synthetic",
        "timing": (f"median of {repeats} repetitions of the whole run ({W} warm-up + {K} timed positions each, context reset between)"
                   if repeats > 1 else "one run"),
        "value_all": [round((1 if a.tp else world) * K / t, 1) for t, _ in runs],
        "config": {"workload": f"{a.shape} {a.type}{' + q6_K classifier' if a.cls_q6k else ''} decode, greedy, positions {W + 1}..{W + K} (./llm -n {W + K} -t 0)",
                   "consumer": "device argmax, K positions pipelined in one call (llmk_decode_greedy): an AUXILIARY line, the headline keeps the reference's host consumer" if step else "logits to host + host argmax (llmk_forward)",
                   "parallelism": ((f"tp{world} (row-parallel GEMVs, " + ("one-shot peer-memory all-reduce" if m.path() == 2 else "RCCL all-reduce")
                                    + (f", peer-memory self-test verdicts per rank {getattr(m, 'p2p_selftest', None)}" if a.tp_collective == "p2p" else "")
                                    + (", ALL RANKS SHARING ONE GPU: protocol check, not a scaling number" if os.environ.get("LLMK_SHARE_GPU") else "") + ")")
                                   if a.tp else "replicas" if world > 1 else "single GPU"), "seed": SEED,
                   "path": m.path_name()},
        "ranks_seen": m.tp_ranks_seen(),   # from the collective itself (ncclCommCount / mapped peer inboxes): 1 for replicas
    }
    if rank == 0:
        # Dominant kernel.  Default path for this shape: ONE persistent kernel per token (token_kernel.h):
        # its launch IS the hot path, so achieved = whole-token algorithmic bytes / its HIP-event duration.
        # Multi-kernel path (other shapes / types, --multi-kernel): the fused w1|w3 GEMV (92.3 MB of the
        # 176.2 MB a layer streams).
        try:
            ms, b = m.time_kernel(6, 100)
            out["roofline"] = {"bound": "hbm", "kernel": f"token_kernel<{a.shape}, {a.type}> (persistent whole-token pass: "
                                                         f"{shape.n_layers} layers + classifier)",
                               "achieved": b / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic("token_kernel", a.shape, a.type + ("_q6k" if a.cls_q6k else "")),
                               "bytes_per_launch": b, "us_per_launch": ms * 1000,
                               "note": f"bytes_per_launch = algorithmic bytes of one token at KV length {W + K}"}
        except llmk.LlmkError:
            kern = 3
            m.reset()
            ms, b = m.time_kernel(kern, 10 * shape.n_layers)
            per_k = {}
            for k in KERNEL_NAMES:
                kms, kb = m.time_kernel(k, 5 * shape.n_layers)
                per_k[KERNEL_NAMES[k]] = {"us": round(kms * 1000, 3), "GBps": round(kb / kms / 1e6, 1)}
            ach = b / (ms * 1e-3) / 1e9
            kname = "gemv_q4_kernel<SWIGLU,NORM>" if wtype == 2 else f"gemv_kernel<{a.type},SWIGLU,NORM>"
            out["roofline"] = {"bound": "hbm", "kernel": kname + " (rmsnorm+w1|w3 GEMV+SwiGLU)",
                               "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": pmc_traffic("gemv_q4_kernel<3, true" if wtype == 2 else f"gemv_kernel<{wtype}, 3, true", a.shape, a.type),
                               "bytes_per_launch": b,
                               "us_per_launch": ms * 1000, "kernels": per_k}
        tok_gbs = bpt * (tok_s / (1 if a.tp else world)) / 1e9 / (world if a.tp else 1)   # per-GPU HBM rate
        out["token_roofline"] = {"bytes_per_token": bpt, "achieved": tok_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": tok_gbs / HBM_PEAK_GBS, "roofline_tok_s": HBM_PEAK_GBS * 1e9 / bpt}
        out["setup_s"] = {"weights_gen": round(t_gen, 1), "upload": round(t_up, 1)}
        want_cb = world == 1 and not a.no_cpu_baseline and fw is not None
        want_fh = (world == 1 and not a.no_fortran_host and (a.fortran_host or not a.no_cpu_baseline) and fw is not None
                   and not a.multi_kernel and not a.no_graph and not a.greedy_on_device)
        td = tempfile.TemporaryDirectory(dir=os.environ.get("LLMK_TMP", "/tmp")) if (want_cb or want_fh) else None
        try:            # (the 4.4 GB file must not outlive an exception in the CPU / Fortran legs: advisor, round 5)
            gpath = None
            if td is not None:
                gpath = os.path.join(td.name, f"synthetic-{a.shape}-{a.type}.gguf")
                gguf.write_gguf(gpath, fw)              # the same weights the timed loop ran on, as the file both CLIs read
            if want_fh:
                # the drop-in itself, on this GPU, while this process's context idles
                out["fortran_host"] = fortran_host(gpath, shape.vocab_size, min(shape.seq_len, max(W + K, 256)), gpu_ids, local)
                if "tok_s" in out["fortran_host"]:
                    out["fortran_host"]["vs_ctypes_value"] = round(out["fortran_host"]["tok_s"] / tok_s, 4)
            if want_cb:
                cb = cpu_baseline(fw, a.shape, wtype, gguf_path=gpath)
                if cb is not None and "_ids" in cb:
                    # free parity check at the bench's own size: the REAL reference's greedy transcript (same weights, same
                    # box, positions 1..n) against the ids the GPU just produced in the timed loop
                    ref_ids = cb.pop("_ids")
                    n = min(len(ref_ids), len(gpu_ids))
                    same = [x == y for x, y in zip(ref_ids[:n], gpu_ids[:n])]
                    first = same.index(False) if False in same else None
                    cb["ids_match"] = f"{sum(same)}/{n}"
                    cb["first_mismatch_pos"] = None if first is None else first + 1
                    out["ids_match"] = cb["ids_match"]
                out["cpu_baseline"] = cb
        finally:
            if td is not None:
                td.cleanup()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    m.close()
    rep.close()


if __name__ == "__main__":
    main()
