"""bench.py's one-line contract, run the way the driver runs it (`--steps 20 --warmup 5`, N = 1): ONE JSON line on stdout with the
fields the driver reads, the `roofline` and `cpu_baseline` objects of the tier framing, and a greedy transcript that matches the
reference's (`ids_match`).  The numbers themselves are not asserted beyond sanity: this guards the plumbing, not the performance."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_driver_invocation_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"],
                       capture_output=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:].decode(errors="replace")
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["unit"] == "tokens/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 100 and abs(d["value"] * d["ms_per_step"] / 1000.0 - 1.0) < 0.02      # tokens/s x s/token = 1 at N = 1
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.05 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 0.5 * rf["bytes_per_launch"]
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "tokens/s" and cb["sample"]
    if cb["kind"] == "reference":
        same, n = d["ids_match"].split("/")                    # the timed loop's ids against the real reference's transcript
        assert same == n and int(n) >= 20, d["ids_match"]
    # the drop-in itself (round-4 verdict, item 3): the Fortran `llm` CLI on the same GGUF, -n 256 -t 0, its own tokens/second line
    fh = d["fortran_host"]
    assert "error" not in fh, fh
    same, n = fh["ids_match"].split("/")
    assert same == n and int(n) >= 25, fh
    same, n = fh["ids_match_device_argmax"].split("/")
    assert same == n, fh
    # The CLI's rate covers positions 2..256 (mean KV length ~129), the 20-step line positions 6..25: the kernel is ~3 % slower at
    # the longer context (663 -> 683 us from KV length 1 to 256), so the bar against THIS line is 6 %; against a
    # 248-step line (bench.py's default, same positions) it is the verdict's 3 %: tests/test_host_gpu.py.
    assert fh["tok_s"] > 0.94 * d["value"], (fh, d["value"])
    assert fh["tok_s_device_argmax"] >= 0.97 * fh["tok_s"], fh
