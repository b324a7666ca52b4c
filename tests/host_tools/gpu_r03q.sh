L=$PWD/llm.f90_amd/csrc
mkdir -p gpurun_out/ab; out=gpurun_out/ab; : > $out/gf4.jsonl
one() { local label=$1 lib=$2; shift 2; local line; line=$(LLMK_LIB=$lib python bench.py --no-cpu-baseline "$@" 2>>$out/err.log | tail -1); echo "{\"build\": \"$label\", \"args\": \"$*\", \"line\": $line}" >> $out/gf4.jsonl; }
for i in 1 2; do for v in _gf0 "" _gf_n1d40 _gf_n1d56 _gf_n4d56 _gf_n4d40; do one "head$v" $L/libllmk$v.so; done; done
for v in _gf0 ""; do one "head$v" $L/libllmk$v.so --steps 20 --warmup 5; one "head$v" $L/libllmk$v.so --type f16; done
python - <<'PY'
import json
for r in map(json.loads, open("gpurun_out/ab/gf4.jsonl")):
    l = r["line"]; print(f'{r["build"]:16s} {r["args"]:24s} {l["value"]:8.1f} tok/s  kernel {l["roofline"]["us_per_launch"]:7.1f} us')
PY
