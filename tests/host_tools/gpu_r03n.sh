mkdir -p gpurun_out/ab; out=gpurun_out/ab; : > $out/xbd.jsonl
L=$PWD/llm.f90_amd/csrc
one() { local label=$1 lib=$2; shift 2; local line; line=$(LLMK_LIB=$lib python bench.py --no-cpu-baseline "$@" 2>>$out/err.log | tail -1); echo "{\"build\": \"$label\", \"args\": \"$*\", \"line\": $line}" >> $out/xbd.jsonl; }
for v in "" _xbd32 _xbd64 _xbd100 _xbd150; do one "xbd$v" $L/libllmk$v.so --type f16; one "xbd$v" $L/libllmk$v.so; one "xbd$v" $L/libllmk$v.so --shape llama2-7b --type q4_0; done
for v in "" _xbd64; do one "xbd$v" $L/libllmk$v.so --type f16; one "xbd$v" $L/libllmk$v.so; done
python - <<'PY'
import json
for r in map(json.loads, open("gpurun_out/ab/xbd.jsonl")):
    l = r["line"]; print(f'{r["build"]:12s} {r["args"]:36s} {l["value"]:8.1f} tok/s  kernel {l["roofline"]["us_per_launch"]:7.1f} us')
PY
