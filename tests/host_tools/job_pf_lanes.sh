#!/usr/bin/env bash
# prefill lanes (LLMK_PF_LANES = 2, 3, 4): parity first, then interleaved bench lines of the 512- and 2,000-token prompt
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/pf_lanes; : > gpurun_out/pf_lanes/lines.jsonl
for n in 3 4; do
  LLMK_PF_LANES=$n timeout 900 python -m pytest tests/test_prefill_gpu.py tests/test_parity_gpu.py -m gpu -x -q -k "prefill" > gpurun_out/pf_lanes/parity_$n.log 2>&1
  echo "lanes $n parity: $(grep -a 'passed\|failed' gpurun_out/pf_lanes/parity_$n.log | tail -1)"
done
for r in 1 2 3; do
  for cfg in "--prefill 512" "--prefill 512 --type f16" "--prefill 512 --shape llama2-7b --type q4_0" "--prefill 2000 --shape llama2-7b --type q4_0"; do
    for n in 2 3 4; do
      line=$(LLMK_PF_LANES=$n timeout 400 python bench.py --no-cpu-baseline $cfg 2>/dev/null | tail -1)
      echo "{\"lanes\": $n, \"cfg\": \"$cfg\", \"round\": $r, \"line\": ${line:-null}}" >> gpurun_out/pf_lanes/lines.jsonl
    done
  done
done
python - <<'PY'
import json, collections
d = collections.defaultdict(list)
for l in open("gpurun_out/pf_lanes/lines.jsonl"):
    r = json.loads(l)
    if r["line"]: d[(r["cfg"], r["lanes"])].append(r["line"]["value"])
for k in sorted(d): print(f"{k[0]:48s} lanes {k[1]}: " + " ".join(f"{v:9.0f}" for v in d[k]))
PY
