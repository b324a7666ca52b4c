L=$PWD/llm.f90_amd/csrc
timeout 900 python -m pytest tests/test_prefill_gpu.py -x -q -k "q4" 2>&1 | tail -3
for plan in 4 2; do
LLMK_LIB=$L/libllmk_debug.so LLMK_PF_PLAN=$plan python tests/host_tools/pf_trace.py --type q4_0 w13 2>&1 | grep -E "GEMM|prologue|step  [1-3]|exit"
LLMK_LIB=$L/libllmk_debug.so LLMK_PF_PLAN=$plan python bench.py --prefill 512 --shape llama2-7b --type q4_0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100
LLMK_LIB=$L/libllmk_debug.so LLMK_PF_PLAN=$plan python bench.py --prefill 512 --type q4_0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100
done
python bench.py --prefill 512 --shape llama2-7b --type q4_0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100
python bench.py --prefill 512 --type q4_0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100
