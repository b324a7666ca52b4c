#!/usr/bin/env bash
# Interleaved A/B of library variants (tests/host_tools/build_variant.sh) on ONE box:
#   ab_libs.sh TAG ROUNDS "<bench.py args>" base NAME1 NAME2 ...     ("base" = the product library)
# each variant first passes the parity test named by AB_PARITY (default: the 7B column-geometry test), then ROUNDS interleaved
# bench lines per variant -> gpurun_out/TAG/ab.jsonl + a table
cd "$(dirname "$0")/../.." || exit 1
TAG=$1; N=$2; ARGS=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT; : > $OUT/ab.jsonl
lib() { if [ "$1" = base ]; then echo "$PWD/llm.f90_amd/csrc/libllmk.so"; else echo "$PWD/llm.f90_amd/csrc/variants/libllmk_$1.so"; fi; }
for v in "$@"; do
  if [ -n "${AB_PARITY-tests/test_parity_gpu.py -k column_geometry}" ]; then
    if LLMK_LIB=$(lib $v) timeout 900 python -m pytest ${AB_PARITY-tests/test_parity_gpu.py -k column_geometry} -m gpu -x -q > $OUT/parity_$v.log 2>&1; then echo "parity $v: ok"; else echo "parity $v: FAILED"; tail -5 $OUT/parity_$v.log; fi
  fi
done
for i in $(seq 1 $N); do
  for v in "$@"; do
    line=$(LLMK_LIB=$(lib $v) timeout 600 python bench.py --no-cpu-baseline $ARGS 2>/dev/null | tail -1)
    echo "{\"variant\": \"$v\", \"round\": $i, \"line\": ${line:-null}}" >> $OUT/ab.jsonl
  done
done
python - $OUT/ab.jsonl <<'PY'
import json, sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    r = json.loads(l)
    if r["line"]: d[r["variant"]].append((r["line"]["value"], r["line"].get("roofline", {}).get("us_per_launch", 0)))
for v, xs in d.items():
    print(f"{v:12s} tok/s " + " ".join(f"{a:8.1f}" for a, _ in xs) + "   kernel us " + " ".join(f"{b:7.1f}" for _, b in xs))
PY
