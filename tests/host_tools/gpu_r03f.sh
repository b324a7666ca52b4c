python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03f_pytest.log; cat gpurun_out/r03f_pytest.log
python tests/host_tools/tp_selftest_diag.py 2 64
python tests/host_tools/tp_selftest_diag.py 4 200
mkdir -p gpurun_out/ab; out=gpurun_out/ab; : > $out/hb44.jsonl
one() { local label=$1 lib=$2; shift 2; local line; line=$(LLMK_LIB=$lib python bench.py --no-cpu-baseline "$@" 2>>$out/err.log | tail -1); echo "{\"build\": \"$label\", \"args\": \"$*\", \"line\": $line}" >> $out/hb44.jsonl; }
L=$PWD/llm.f90_amd/csrc
for i in 1 2; do
  one head $L/libllmk.so --type f16; one hb44 $L/libllmk_hb44.so --type f16
  one head $L/libllmk.so; one hb44 $L/libllmk_hb44.so
done
python - <<'PY'
import json
for r in map(json.loads, open("gpurun_out/ab/hb44.jsonl")):
    l = r["line"]; print(f'{r["build"]:8s} {r["args"]:12s} {l["value"]:8.1f} tok/s  kernel {l["roofline"]["us_per_launch"]:7.1f} us')
PY
LLMK_LIB=$L/libllmk_debug.so LLMK_TK_TRACE=1 python tests/host_tools/tk_trace.py --shape tinyllama --type f16 --pos 130 2>&1 | tail -22 | tee gpurun_out/ab/trace_f16.txt
LLMK_LIB=$L/libllmk_debug.so LLMK_TK_TRACE=1 python tests/host_tools/tk_trace.py --shape llama2-7b --type q4_0 --pos 130 2>&1 | tail -22 | tee gpurun_out/ab/trace_7b.txt
LLMK_TMP=/dev/shm timeout 1500 python tests/host_tools/tp_load_rss.py llama2-70b --ngpu 8 2>&1 | tail -6 | tee gpurun_out/ab/rss_70b.jsonl
