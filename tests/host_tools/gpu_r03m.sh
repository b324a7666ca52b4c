python -m pytest tests/test_parity_gpu.py tests/test_decode_greedy_gpu.py -x -q -k "not 7b and not q4" 2>&1 | tail -4
mkdir -p gpurun_out/ab; out=gpurun_out/ab; : > $out/hb3.jsonl
L=$PWD/llm.f90_amd/csrc
one() { local label=$1 lib=$2; shift 2; local line; line=$(LLMK_LIB=$lib python bench.py --no-cpu-baseline "$@" 2>>$out/err.log | tail -1); echo "{\"build\": \"$label\", \"args\": \"$*\", \"line\": $line}" >> $out/hb3.jsonl; }
for i in 1 2; do for v in _hb3off "" _hb3nl32 _hb3nl11; do one "hb3$v" $L/libllmk$v.so --type f16; done; done
for i in 1 2; do for v in _hb3off "" _hb3nl32 _hb3nl11; do one "hb3$v" $L/libllmk$v.so; done; done
python - <<'PY'
import json
for r in map(json.loads, open("gpurun_out/ab/hb3.jsonl")):
    l = r["line"]; print(f'{r["build"]:12s} {r["args"]:12s} {l["value"]:8.1f} tok/s  kernel {l["roofline"]["us_per_launch"]:7.1f} us')
PY
