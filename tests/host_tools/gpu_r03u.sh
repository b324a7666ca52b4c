L=$PWD/llm.f90_amd/csrc
python -m pytest tests/test_parity_gpu.py -x -q -k "f16 or tk-small16" 2>&1 | tail -2
mkdir -p gpurun_out/ab; out=gpurun_out/ab; : > $out/gc.jsonl
one() { local label=$1 lib=$2; shift 2; local line; line=$(LLMK_LIB=$lib python bench.py --no-cpu-baseline "$@" 2>>$out/err.log | tail -1); echo "{\"build\": \"$label\", \"args\": \"$*\", \"line\": $line}" >> $out/gc.jsonl; }
for i in 1 2; do for v in _gc0 ""; do one "head$v" $L/libllmk$v.so --type f16; done; done
python - <<'PY'
import json
for r in map(json.loads, open("gpurun_out/ab/gc.jsonl")):
    l = r["line"]; print(f'{r["build"]:16s} {r["args"]:24s} {l["value"]:8.1f} tok/s  kernel {l["roofline"]["us_per_launch"]:7.1f} us')
PY
