import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
s = gguf.SHAPES["tinyllama"]
fw = gguf.synth_fused(s, 20260928)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ref = llmk.Llmk(fw, flags=llmk.FLAG_MULTI_KERNEL)
rt, rl = ref.generate(n)
ref.close()
m = llmk.Llmk(fw)
t0 = time.time()
try:
    toks, lg = m.generate(n)
except Exception as e:
    print("TK FAILED:", e); sys.exit(1)
print("tk generate", time.time() - t0, "s")
err = np.max(np.abs(lg - rl), axis=1) / np.max(np.abs(rl), axis=1)
print("rel err per pos:", err)
print("tokens equal:", np.array_equal(toks, rt), toks[:8], rt[:8])
# timing
import ctypes as C
lib, h = llmk.lib(), m._h
lgp = m._logits.ctypes.data_as(C.POINTER(C.c_float))
tok = int(toks[-1]); t0 = time.perf_counter(); K = 200
for pos in range(n + 1, n + K + 1):
    rc = lib.llmk_forward(h, tok, pos, lgp); assert rc == 0, rc
    tok = int(m._logits.argmax()) + 1
dt = time.perf_counter() - t0
print(f"tk: {K/dt:.1f} tok/s  {dt/K*1e6:.1f} us/token")
ms, b = m.time_kernel(6, 100)
print(f"token kernel alone: {ms*1000:.1f} us  {b/ms/1e6:.1f} GB/s")
import os
if os.environ.get("LLMK_TK_TRACE"):
    raw = m.peek(6, 256 * 16 * 64 * 2).view(np.uint64).reshape(256, 64, 16).astype(np.float64)
    t0 = raw[:, 0, 0].min()
    us = (raw - t0) / 100.0   # 100 MHz wall clock
    names = ["start","gathX","barA0","barB0","pubQKV","attGath","attDone","gathXB","barB_O","gathXA","barA_A","barB_A","gathHB","barA_D","barB_D","pubX"]
    for cu in (0, 8, 100, 255):
        print("CU", cu)
        for l in (0, 1, 5, 21):
            d = us[cu, l]
            print("  L%02d " % l + " ".join(f"{n}={d[i]-d[0]:.1f}" for i, n in enumerate(names)) + f"  | layer start @{d[0]:.1f}us")
    seg = np.diff(us[:, 1:22, :], axis=2)
    att = (np.arange(256) % 8) == ((np.arange(256) // 8 // 8) % 8)
    print("mean segment us (non-attention CUs):", " ".join(f"{names[i+1]}:{seg[~att][:, :, i].mean():.2f}" for i in range(15)))
    print("mean segment us (attention CUs):    ", " ".join(f"{names[i+1]}:{seg[att][:, :, i].mean():.2f}" for i in range(15)))
    print("layer time:", (us[:, 2:22, 0] - us[:, 1:21, 0]).mean())
    print("--- exchange latency: (consumer gather done) - (last producer publish), us; and producer skew")
    att_idx = np.where(att)[0]
    for l in (1, 5, 10, 20):
        def rep(name, prod, cons):
            last = prod.max(); print(f"  L{l:02d} {name}: producers finish spread {prod.max()-prod.min():.1f}us; consumers done after last producer: min {cons.min()-last:.2f} mean {cons.mean()-last:.2f} max {cons.max()-last:.2f}")
        rep("x   (w2->qkv) ", us[:, l-1, 15], us[:, l, 1])
        rep("qkv (->attn)  ", us[:, l, 4], us[att_idx, l, 5])
        rep("xb  (attn->wo)", us[att_idx, l, 6], us[:, l, 7])
        rep("xa  (wo->w13) ", us[:, l, 8], us[:, l, 9])
        rep("hb  (w13->w2) ", us[:, l, 11], us[:, l, 12])
    g = raw[:, 32:32+22, :10]
    for j, nm in enumerate(["x", "xb", "xa", "hb(1st half)", "hb(2nd half)"]):
        print(f"gather {nm}: passes mean {g[:, 1:, 2*j].mean():.1f} max {g[:, 1:, 2*j].max():.0f}; last pass us mean {g[:, 1:, 2*j+1].mean()/100:.2f} max {g[:, 1:, 2*j+1].max()/100:.2f}")
    ai = np.where(att)[0]
    d = raw[ai][:, 32+1:32+22, 10:15] / 100.0     # scores done, after barrier, end, after max, after sum
    t5 = raw[ai][:, 1:22, 5] / 100.0; t6 = raw[ai][:, 1:22, 6] / 100.0
    print("attention (service wave): enter->scores %.2f | barrier %.2f | max %.2f | exp+sum %.2f | PV+write %.2f | tail barrier %.2f" % (
        (d[:, :, 0] - t5).mean(), (d[:, :, 1] - d[:, :, 0]).mean(), (d[:, :, 3] - d[:, :, 1]).mean(),
        (d[:, :, 4] - d[:, :, 3]).mean(), (d[:, :, 2] - d[:, :, 4]).mean(), (t6 - d[:, :, 2]).mean()))
