"""q4_0 GEMV with column slices across the block's waves (kernels.h gemv_q4_kernel KS = 4) against the plain form and the
oracle on a K = 8192 shape (LLMK_Q4_KS=1 forces it, =0 forbids it)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    import llm_f90_amd
    from llm_f90_amd import llmk
    from llm_f90_amd.tools import gguf
    s = gguf.LlamaShape(8192, 8192, 1, 64, 8, 512, 16)
    fw = gguf.synth_fused_q4_direct(s, 3)
    m = llmk.Llmk(fw, flags=llmk.FLAG_MULTI_KERNEL)
    toks, logits = m.generate(6)
    np.save(sys.argv[1], logits)
    if sys.argv[1].endswith("0.npy"):
        from oracle.oracle import Oracle
        ot, ol = Oracle(fw.as_f32(), "omp").generate(6)
        np.save(sys.argv[1].replace("0.npy", "oracle.npy"), ol)
else:
    import numpy as np
    for ks in ("0", "1"):
        subprocess.run([sys.executable, __file__, f"/tmp/ks_{ks}.npy"], env=dict(os.environ, LLMK_Q4_KS=ks), check=True)
    a, b, o = np.load("/tmp/ks_0.npy"), np.load("/tmp/ks_1.npy"), np.load("/tmp/ks_oracle.npy")
    rel = lambda x, y: float(np.max(np.abs(x - y)) / np.max(np.abs(y)))
    print("finite", np.isfinite(a).all(), np.isfinite(b).all(), "plain vs oracle", rel(a, o), "KS vs oracle", rel(b, o), "KS vs plain", rel(b, a))
