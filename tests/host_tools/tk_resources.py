#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch of every persistent-kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage), and -- when
a kernel has scratch at all -- how many of its scratch instructions sit INSIDE A LOOP (from the ISA: a spill in the layer
loop drains the prefetch ring once per ring slot; one in the straight-line classifier tail costs one drain per token).
The persistent kernels sit at the register ceiling, so every edit is checked here before it goes to a GPU.
With --lds64 also the number of 8-byte LDS reads / writes per persistent kernel: the service wave's rmsnorm staging (TkNorm::apply)
moves 16 bytes per access; in round 5 an unrelated edit made hipcc split them into interleaved 8-byte pairs around a packed
multiply -- +7 % on the TinyLlama f16 token, nothing in any resource number.
    python tests/host_tools/tk_resources.py [-DNAME ...] [--all] [--lds64]"""
import os
import re
import subprocess
import sys

pkg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm.f90_amd")
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
# ONE device-only compilation gives both: the resource remarks on stderr, the ISA on stdout (two compilations of llmk.hip made this the
# slowest test of the CPU suite by far)
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-S", "--cuda-device-only", "csrc/llmk.hip",
                    "-o", "-", "-Rpass-analysis=kernel-resource-usage"] + defs, cwd=pkg, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
ISA = r.stdout
def scratch_in_loops():
    """kernel symbol -> (scratch instructions inside loops, scratch instructions in all)"""
    out, cur, inloop = {}, None, False
    for line in ISA.split("\n"):
        t = line.strip()
        if line.startswith("_ZN") and t.endswith(":") is False and ":" in line and not line.startswith(" "):
            cur = line.split(":")[0]
            out[cur] = [0, 0]
            inloop = False
        elif re.match(r"^\.LBB\d+_\d+:", t):
            inloop = "Loop" in line
        elif cur and "scratch_" in t and not t.startswith(";"):
            out[cur][1] += 1
            out[cur][0] += 1 if inloop else 0
    return out


def lds64():
    """kernel symbol -> (ds_read_b64, ds_write_b64) instruction counts"""
    out, cur = {}, None
    for line in ISA.split("\n"):
        if line.startswith("_ZN") and ":" in line and not line.startswith(" "):
            cur = line.split(":")[0]
            out[cur] = [0, 0]
        elif cur and not line.strip().startswith(";"):
            t = line.strip()
            if t.startswith("ds_read_b64"):
                out[cur][0] += 1
            elif t.startswith("ds_write_b64"):
                out[cur][1] += 1
    return out


l64 = lds64() if "--lds64" in sys.argv else None
loops = None
for blk in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
    name = blk.split("\n")[0]
    if "--all" not in sys.argv and "token_kernel" not in name and "tk2" not in name:
        continue

    def g(k):
        m = re.search(k + r": (\d+)", blk)
        return m.group(1) if m else "?"
    demangled = subprocess.run(["c++filt", name.split()[0]], capture_output=True, text=True).stdout.strip()
    short = demangled.replace("llmk::", "").replace("void ", "")[:88]
    scratch, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    extra = ""
    if scratch not in ("0", "?"):
        loops = loops if loops is not None else scratch_in_loops()
        il, al = loops.get(name.split()[0], [-1, -1])
        extra = f" scratch_ops_in_loops {il} of {al}"
    if l64 is not None:
        extra += " lds64 %d/%d" % tuple(l64.get(name.split()[0], [-1, -1]))
    print(f"{short:88s} VGPR {g(' VGPRs'):>3} AGPR {g('AGPRs'):>3} SGPR {g('SGPRs'):>3} scratch {scratch:>4} occ {occ}{extra}")
