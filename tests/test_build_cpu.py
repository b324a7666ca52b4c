"""Build-time guards that need no GPU: hipcc cross-compiles for gfx950 here."""
import os
import re
import subprocess
import sys

from conftest import ROOT


def test_persistent_kernels_do_not_spill():
    """The persistent token kernels sit at the 256-VGPR ceiling on purpose (register tile ring).  A spill in a streaming
    wave costs one `s_waitcnt vmcnt(0)` + scratch store per ring slot -- the whole prefetch ring drains -- and hipcc's
    allocation at the ceiling flips on unrelated edits (round 3: `if (a.gflags & 16)` instead of `if (a.herr)` in the
    service wave put 36 bytes of scratch into the f32 kernel's streaming loop).  So, checked on every build: no scratch
    instruction inside a loop, and at most 32 bytes of scratch at all (the f32 kernel currently keeps one 16-byte value
    across its straight-line classifier tail: one store, one reload per token, measured faster than the spill-free
    neighbours of the same schedule -- DESIGN.md section 3b)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_tools", "tk_resources.py"), "--all", "--lds64"], capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    # the prefill GEMMs on the f16 matrix instruction hold a ring of weight stages, two activation stages and 64 accumulators
    # per wave in (unified) registers: they must not spill at all (a first q4_0 version did: prefill.h)
    gemm = [l for l in r.stdout.splitlines() if "pf_gemm_h_kernel" in l]
    assert len(gemm) >= 48, len(gemm)
    for l in gemm:
        m = re.search(r"scratch\s+(\d+)", l)
        assert m and int(m.group(1)) == 0, l
    rows = [l for l in r.stdout.splitlines() if "token_kernel" in l or "tk2" in l]
    assert len(rows) >= 5, r.stdout
    # the f32 / f16 kernels' rmsnorm staging in 16-byte LDS accesses (round 5: hipcc once split them in pairs of 8: +7 % on the f16 token)
    for l in rows:
        if ", 0, -1>" in l or ", 1, -1>" in l:   # weight types f32 / f16 (TkShape<..., WT, CLS>)
            k = re.search(r"lds64 (-?\d+)/(-?\d+)", l)
            assert k and 0 <= int(k.group(1)) <= 4 and 0 <= int(k.group(2)) <= 4, l
    for l in rows:
        m = re.search(r"VGPR\s+(\d+).*scratch\s+(\d+)", l)
        assert m, l
        assert int(m.group(1)) <= 256, l
        if int(m.group(2)):
            assert int(m.group(2)) <= 32, l
            k = re.search(r"scratch_ops_in_loops (-?\d+) of (\d+)", l)
            assert k and int(k.group(1)) == 0 and int(k.group(2)) <= 4, l


def _quoted_includes(path, seen):
    """every file reachable from `path` through #include "..." lines (relative to the including file)"""
    path = os.path.normpath(path)
    if path in seen:
        return
    seen.add(path)
    for line in open(path):
        m = re.match(r'\s*#\s*include\s+"([^"]+)"', line)
        if m:
            _quoted_includes(os.path.join(os.path.dirname(path), m.group(1)), seen)


def test_every_included_header_is_a_prerequisite_of_the_libraries():
    """Round-5 verdict, build hygiene: csrc/q4_units.h (included by token_kernel.h) was not among libllmk.so's prerequisites, so
    an edit of the unit layout alone left a stale library.  Walk the #include "..." graph of llmk.hip and hold the Makefile's
    list (`make print-lib-deps`) to it; both library rules must use that list."""
    pkg = os.path.join(ROOT, "llm.f90_amd")
    r = subprocess.run(["make", "-s", "-C", pkg, "print-lib-deps"], capture_output=True, text=True, check=True)
    deps = {os.path.normpath(os.path.join(pkg, d)) for d in r.stdout.split()}
    seen = set()
    _quoted_includes(os.path.join(pkg, "csrc", "llmk.hip"), seen)
    assert len(seen) >= 7, seen
    missing = sorted(s for s in seen if s not in deps)
    assert not missing, f"not prerequisites of libllmk.so: {missing}"
    mk = open(os.path.join(pkg, "Makefile")).read()
    assert re.search(r"^csrc/libllmk\.so: \$\(LIB_DEPS\)$", mk, re.M) and re.search(r"^csrc/libllmk_debug\.so: \$\(LIB_DEPS\)$", mk, re.M)
