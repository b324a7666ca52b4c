set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03d_pytest.log; cat gpurun_out/r03d_pytest.log
mkdir -p gpurun_out/ab
out=gpurun_out/ab; : > $out/stamp_f32.jsonl
one() { local label=$1 lib=$2; shift 2; local line; line=$(LLMK_LIB=$lib python bench.py --no-cpu-baseline "$@" 2>>$out/err.log | tail -1); echo "{\"build\": \"$label\", \"args\": \"$*\", \"line\": $line}" >> $out/stamp_f32.jsonl; }
L=$PWD/llm.f90_amd/csrc
for i in 1 2 3; do
  (cd ab_r01 && python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | sed 's/^/{"build": "r01", "args": "--steps 20 --warmup 5", "line": /; s/$/}/') >> $out/stamp_f32.jsonl
  one head $L/libllmk.so --steps 20 --warmup 5 --repeats 1
  one stamp1 $L/libllmk_stamp1.so --steps 20 --warmup 5 --repeats 1
  one stamp2 $L/libllmk_stamp2.so --steps 20 --warmup 5 --repeats 1
done
(cd ab_r01 && python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | sed 's/^/{"build": "r01", "args": "", "line": /; s/$/}/') >> $out/stamp_f32.jsonl
one head $L/libllmk.so
one stamp1 $L/libllmk_stamp1.so
one stamp2 $L/libllmk_stamp2.so
python - <<'PY'
import json
for r in map(json.loads, open("gpurun_out/ab/stamp_f32.jsonl")):
    l = r["line"]; print(f'{r["build"]:12s} {r["args"]:32s} {l["value"]:8.1f} tok/s  kernel {l["roofline"]["us_per_launch"]:7.1f} us')
PY
python tests/host_tools/tp_load_rss.py llama2-7b 2>&1 | tail -4 | tee gpurun_out/ab/rss_7b.jsonl
LLMK_TMP=/dev/shm timeout 1500 python tests/host_tools/tp_load_rss.py llama2-70b --ngpu 8 2>&1 | tail -6 | tee gpurun_out/ab/rss_70b.jsonl
