L=$PWD/llm.f90_amd/csrc
timeout 600 python -m pytest tests/test_prefill_gpu.py -x -q 2>&1 | tail -3
for plan in 2 1; do
LLMK_LIB=$L/libllmk_debug.so LLMK_PF_PLAN=$plan python tests/host_tools/pf_trace.py --type f16 w13 2>&1 | grep -E "GEMM|prologue|step  [0-4]|step 10|exit"
done
python bench.py --prefill 512 --type f16 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120
