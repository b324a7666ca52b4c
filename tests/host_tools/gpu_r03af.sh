L=$PWD/llm.f90_amd/csrc
timeout 900 python -m pytest tests/test_prefill_gpu.py tests/test_host_gpu.py -x -q 2>&1 | tail -3
python bench.py --prefill 512 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100
LLMK_PF_F32_MFMA=1 python bench.py --prefill 512 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100
for plan in 2 1; do
LLMK_LIB=$L/libllmk_debug.so LLMK_PF_PLAN=$plan python tests/host_tools/pf_trace.py --type f32 w13 2>&1 | grep -E "GEMM|prologue|step  [1-3]|exit"
done
