#!/usr/bin/env python3
"""VGPRs / SGPRs / scratch of every persistent-kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage).
The persistent kernels sit at the register ceiling, so every edit is checked for spills here before it goes to a GPU.
    python tests/host_tools/tk_resources.py [-DNAME ...] [--all]"""
import os
import re
import subprocess
import sys

pkg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "llm.f90_amd")
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c", "csrc/llmk.hip",
                    "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + defs, cwd=pkg, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
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
    print(f"{short:88s} VGPR {g(' VGPRs'):>3} AGPR {g('AGPRs'):>3} SGPR {g('SGPRs'):>3} scratch {scratch:>4} occ {occ}")
