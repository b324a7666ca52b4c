L=$PWD/llm.f90_amd/csrc
mkdir -p gpurun_out/ab; out=gpurun_out/ab; : > $out/hs.jsonl
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_decode_greedy_gpu.py -x -q 2>&1 | tail -5
one() { local label=$1 lib=$2; shift 2; local line; line=$(LLMK_LIB=$lib python bench.py --no-cpu-baseline "$@" 2>>$out/err.log | tail -1); echo "{\"build\": \"$label\", \"args\": \"$*\", \"line\": $line}" >> $out/hs.jsonl; }
for v in "" _hs0 _pre _pre16 _now1 _now2; do one "head$v" $L/libllmk$v.so --type f16; one "head$v" $L/libllmk$v.so; done
python - <<'PY'
import json
for r in map(json.loads, open("gpurun_out/ab/hs.jsonl")):
    l = r["line"]; print(f'{r["build"]:16s} {r["args"]:36s} {l["value"]:8.1f} tok/s  kernel {l["roofline"]["us_per_launch"]:7.1f} us')
PY
