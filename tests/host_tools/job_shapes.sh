export LLMK_LIB=$PWD/llm.f90_amd/csrc/variants/libllmk_shapes.so
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -s -k "shapes_added_at_build_time" 2>&1 | grep "build-time\|passed\|failed\|Error" | cut -c1-300
for cfg in "--shape mistral-7b --type q4_0 --cls-q6k" "--shape llama3-8b --type q4_0 --cls-q6k" "--shape mistral-7b --type f16"; do
  timeout 900 python bench.py --no-cpu-baseline $cfg 2>/dev/null | tail -1 | tee -a gpurun_out/r06_shapes_bench.jsonl | cut -c1-600
done
unset LLMK_LIB
timeout 900 python bench.py --no-cpu-baseline --shape mistral-7b --type q4_0 --cls-q6k 2>/dev/null | tail -1 | tee -a gpurun_out/r06_shapes_bench.jsonl | cut -c1-400
