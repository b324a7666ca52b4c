L=$PWD/llm.f90_amd/csrc
mkdir -p gpurun_out/ab; out=gpurun_out/ab; : > $out/ps.jsonl
one() { local label=$1 lib=$2; shift 2; local line; line=$(LLMK_LIB=$lib python bench.py --no-cpu-baseline "$@" 2>>$out/err.log | tail -1); echo "{\"build\": \"$label\", \"args\": \"$*\", \"line\": $line}" >> $out/ps.jsonl; }
for v in "" _ps0 _ps4 _ps8; do one "head$v" $L/libllmk$v.so --shape llama2-7b --type q4_0; one "head$v" $L/libllmk$v.so --type f16; one "head$v" $L/libllmk$v.so; done
python - <<'PY'
import json
for r in map(json.loads, open("gpurun_out/ab/ps.jsonl")):
    l = r["line"]; print(f'{r["build"]:16s} {r["args"]:36s} {l["value"]:8.1f} tok/s  kernel {l["roofline"]["us_per_launch"]:7.1f} us')
PY
