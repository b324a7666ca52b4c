mkdir -p gpurun_out/ab; out=gpurun_out/ab; : > $out/mix_7b.jsonl
L=$PWD/llm.f90_amd/csrc
one() { local label=$1 lib=$2; shift 2; local line; line=$(LLMK_LIB=$lib python bench.py --no-cpu-baseline "$@" 2>>$out/err.log | tail -1); echo "{\"build\": \"$label\", \"args\": \"$*\", \"line\": $line}" >> $out/mix_7b.jsonl; }
for i in 1 2; do one mix $L/libllmk.so --shape llama2-7b --type q4_0; one nomix $L/libllmk_nomix.so --shape llama2-7b --type q4_0; done
one mix $L/libllmk.so --shape tinyllama --type q4_0; one nomix $L/libllmk_nomix.so --shape tinyllama --type q4_0
python - <<'PY'
import json
for r in map(json.loads, open("gpurun_out/ab/mix_7b.jsonl")):
    l = r["line"]; print(f'{r["build"]:8s} {r["args"]:36s} {l["value"]:8.1f} tok/s  kernel {l["roofline"]["us_per_launch"]:7.1f} us {l["config"]["path"]}')
PY
python tests/host_tools/tp_rank_time.py 4 8 2>&1 | tail -8
LLMK_LIB=$L/libllmk_nomix.so python tests/host_tools/tp_rank_time.py 4 8 2>&1 | tail -8
python -m pytest tests/test_parity_gpu.py -x -q -k "q4 or 7b or oracle" 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python -m pytest tests/test_tp70_gpu.py -x -q -k geometry 2>&1 | tail -2; done
