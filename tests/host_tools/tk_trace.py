#!/usr/bin/env python3
"""Where a token goes inside the persistent kernel: per-phase wall-clock stamps of every CU's service wave.

Needs the debug library (make -C llm.f90_amd debug); run as
    LLMK_LIB=llm.f90_amd/csrc/libllmk_debug.so LLMK_TK_TRACE=1 python tests/host_tools/tk_trace.py --shape llama2-7b --type q4_0 --pos 200
(stamps cost ~10 %: read the segments as proportions).  LLMK_TK_NOSYNC=1 adds the pure streaming time (wrong results).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import llm_f90_amd  # noqa: E402,F401
from llm_f90_amd import llmk  # noqa: E402
from llm_f90_amd.tools import gguf  # noqa: E402
import bench  # noqa: E402

NAMES = ["start", "gathX", "barA_Q", "barB_Q", "pubQKV", "qPoll", "attDone", "gathXB", "barB_O", "gathXA", "barA_A", "barB_A",
         "gathHB", "barA_D", "barB_D", "pubX"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="tinyllama")
    ap.add_argument("--type", default="f32")
    ap.add_argument("--pos", type=int, default=200)
    a = ap.parse_args()
    s = gguf.SHAPES[a.shape]
    wt = {"f32": 0, "f16": 1, "q4_0": 2}[a.type]
    big = s.matmul_params() > 3e9
    fw = None if big else gguf.synth_fused(s, bench.SEED, wt)
    m = bench.build_streamed(s, wt, fw, 0, 0, 0, 1, None) if big else llmk.Llmk(fw)
    tok = 2
    t0 = time.perf_counter()
    for pos in range(1, a.pos + 1):
        tok = int(np.argmax(m.forward(tok, pos))) + 1
    dt = time.perf_counter() - t0
    print(f"{a.pos} tokens, {a.pos / dt:.1f} tok/s incl. python")
    ms, b = m.time_kernel(6, 50)
    print(f"token kernel alone at KV length {a.pos}: {ms * 1000:.1f} us  {b / ms / 1e6:.1f} GB/s  ({b / ms / 1e6 / 80:.1f} % of 8 TB/s)")
    if not os.environ.get("LLMK_TK_TRACE"):
        return
    L = s.n_layers
    raw = m.peek(6, 256 * 16 * 64 * 2).view(np.uint64).reshape(256, 64, 16).astype(np.float64)
    us = (raw - raw[:, 0, 0].min()) / 100.0          # 100 MHz wall clock
    hpc = 256 // s.n_heads
    kvmul = s.n_heads // s.n_kv_heads
    cu = np.arange(256)
    att = (cu % hpc) == ((cu // hpc // kvmul) % hpc)
    nl = min(L, 32)
    seg = np.diff(us[:, 1:nl, :], axis=2)
    print("mean segment us, layers 1..%d" % (nl - 1))
    print("  non-attention CUs: " + " ".join(f"{NAMES[i + 1]}:{seg[~att][:, :, i].mean():.2f}" for i in range(15)))
    print("  attention CUs:     " + " ".join(f"{NAMES[i + 1]}:{seg[att][:, :, i].mean():.2f}" for i in range(15)))
    print("  layer time: %.2f us (x %d layers = %.0f us); classifier tail: %.1f us" % (
        (us[:, 2:nl, 0] - us[:, 1:nl - 1, 0]).mean(), L, (us[:, 2:nl, 0] - us[:, 1:nl - 1, 0]).mean() * L,
        0.0))
    for l in (1, nl // 2, nl - 1):
        def rep(name, prod, cons):
            last = prod.max()
            print(f"  L{l:02d} {name}: producers spread {prod.max() - prod.min():.1f} us; consumers ready after last producer: "
                  f"min {cons.min() - last:.2f} mean {cons.mean() - last:.2f} max {cons.max() - last:.2f}")
        ai = np.where(att)[0]
        rep("x   (w2->qkv) ", us[:, l - 1, 15], us[~att][:, l, 1])
        rep("qkv (->attn)  ", us[~att][:, l, 4], us[ai, l, 5])
        rep("xb  (attn->wo)", us[ai, l, 6], us[~att][:, l, 7])
        rep("xa  (wo->w13) ", us[~att][:, l, 8], us[:, l, 9])
        rep("hb  (w13->w2) ", us[:, l, 11], us[:, l, 12])
    if nl <= 32:
        g = raw[:, 32:32 + min(nl, 22), :10]
        for j, nm in enumerate(["x", "xb", "xa", "hb(1st piece)", "hb(2nd piece)"]):
            print(f"  gather {nm}: passes mean {g[:, 1:, 2 * j].mean():.1f} max {g[:, 1:, 2 * j].max():.0f}; last pass us mean "
                  f"{g[:, 1:, 2 * j + 1].mean() / 100:.2f} max {g[:, 1:, 2 * j + 1].max() / 100:.2f}")
        ai = np.where(att)[0]
        nla = min(nl, 22)
        d = raw[ai][:, 32 + 1:32 + nla, 10:15] / 100.0
        t5 = raw[ai][:, 1:nla, 5] / 100.0
        t6 = raw[ai][:, 1:nla, 6] / 100.0
        print("  attention (service wave): enter->scores %.2f | barrier %.2f | max %.2f | exp+sum %.2f | PV+write %.2f | tail barrier %.2f" % (
            (d[:, :, 0] - t5).mean(), (d[:, :, 1] - d[:, :, 0]).mean(), (d[:, :, 3] - d[:, :, 1]).mean(),
            (d[:, :, 4] - d[:, :, 3]).mean(), (d[:, :, 2] - d[:, :, 4]).mean(), (t6 - d[:, :, 2]).mean()))


if __name__ == "__main__":
    main()
