"""Does freshly allocated VRAM carry what an earlier owner left in it?  (Why an uninitialised read can pass in a dedicated run
and fail inside a long suite: tests/test_tp70_gpu.py)
  (a) the same process frees 4 GB of 0xFF words and allocates again;
  (b) a NEW process allocates while the first one still lives, and after it has gone.
torch is used as a plain allocator here (tool only, nothing of the product)."""
import subprocess
import sys

import torch

N = 1 << 30          # 4 GB of int32 words


def nonzero(x):
    return int((x != 0).sum())


if len(sys.argv) > 1 and sys.argv[1] == "child":
    x = torch.empty(N, dtype=torch.int32, device="cuda")
    print(f"  new process ({sys.argv[2]}): {nonzero(x)} of {N} words of a fresh allocation are not zero")
    sys.exit(0)

for rnd in range(2):
    x = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    del x
    torch.cuda.empty_cache()
    y = torch.empty(N, dtype=torch.int32, device="cuda")
    print(f"round {rnd}: same process, after free + hipMalloc: {nonzero(y)} of {N} words are not zero")
    del y
    torch.cuda.empty_cache()
    subprocess.run([sys.executable, __file__, "child", "first process still alive"])
subprocess.Popen([sys.executable, "-c", "import torch; x = torch.full((1 << 30,), -1, dtype=torch.int32, device='cuda'); torch.cuda.synchronize()"]).wait()
subprocess.run([sys.executable, __file__, "child", "after another process filled 4 GB and exited"])
