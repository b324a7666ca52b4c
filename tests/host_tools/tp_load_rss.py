#!/usr/bin/env python3
"""Peak resident memory of `llm` while it loads a big q4_0 GGUF (shard-only loading).
Writes a synthetic file of the chosen shape tensor by tensor, then runs the CLI and reads the children's rusage:
    python tests/host_tools/tp_load_rss.py llama2-7b                 whole-file load vs --stream-load, one GPU
    python tests/host_tools/tp_load_rss.py llama2-70b --ngpu 8       eight rank processes sharing this box's GPU
ru_maxrss of RUSAGE_CHILDREN = the largest member of the waited-for process tree (with --ngpu the workers are started
through the shell in the background and are NOT waited for: their own peak is read from /proc/<pid>/status VmHWM while
they run).  One JSON line per run."""
import json
import os
import re
import resource
import shutil
import threading
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import llm_f90_amd  # noqa: E402,F401
from llm_f90_amd.tools import gguf  # noqa: E402

shape_name = sys.argv[1] if len(sys.argv) > 1 else "llama2-7b"
ngpu = int(sys.argv[sys.argv.index("--ngpu") + 1]) if "--ngpu" in sys.argv else 1
s = gguf.SHAPES[shape_name]
need = s.matmul_params() * 18 / 32 + 2 * s.vocab_size * s.emb_dim * 4
base = os.environ.get("LLMK_TMP") or ("/dev/shm" if shutil.disk_usage("/dev/shm").free > 1.3 * need else "/tmp")
if shutil.disk_usage(base).free < 1.2 * need:
    sys.exit(f"not enough space under {base} for a {need / 1e9:.1f} GB file")
LLM = os.path.join(ROOT, "llm.f90_amd", "host", "llm")
with tempfile.TemporaryDirectory(dir=base, prefix="llmk_rss_") as td:
    path = os.path.join(td, f"{shape_name}-q4_0.gguf")
    t0 = time.time()
    size = gguf.write_synth_q4_gguf_streamed(path, s, 20260928)
    t_write = time.time() - t0
    runs = [["--ngpu", str(ngpu)]] if ngpu > 1 else [[], ["--stream-load"]]
    for extra in runs:
        env = dict(os.environ)
        if ngpu > 1:
            env["LLMK_TP_SAME_DEVICE"] = "1"
        t0 = time.time()
        before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
        p = subprocess.Popen([LLM, "-m", path, "-n", "8", "-t", "0"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=td, env=env)
        hwm = {}          # pid -> peak resident kB of every `llm` process seen (rank 0 and its workers)

        def watch():
            while p.poll() is None:
                for pid in os.listdir("/proc"):
                    if not pid.isdigit():
                        continue
                    try:
                        if os.readlink(f"/proc/{pid}/exe") != LLM:
                            continue
                        m = re.search(r"VmHWM:\s+(\d+) kB", open(f"/proc/{pid}/status").read())
                        if m:
                            hwm[pid] = max(hwm.get(pid, 0), int(m.group(1)))
                    except OSError:
                        pass
                time.sleep(0.05)
        th = threading.Thread(target=watch)
        th.start()
        out, err = p.communicate(timeout=3000)
        th.join()
        after = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
        print(json.dumps({"shape": shape_name, "file_GB": round(size / 1e9, 2), "write_s": round(t_write, 1), "args": extra,
                          "rc": p.returncode, "wall_s": round(time.time() - t0, 1),
                          "max_rss_GB_rusage_children": round(max(after, before) / 1e6, 3),
                          "peak_rss_GB_per_process": sorted(round(v / 1e6, 3) for v in hwm.values()),
                          "tokens_line": out.split(b"\n")[1][:60].decode(errors="replace") if out.count(b"\n") > 1 else ""}))
        if p.returncode:
            sys.stderr.write(err.decode(errors="replace")[-2000:])
