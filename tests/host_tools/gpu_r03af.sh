L=$PWD/llm.f90_amd/csrc
timeout 600 python -m pytest tests/test_prefill_gpu.py -x -q 2>&1 | tail -5
for env in "" "LLMK_PF_F32_MFMA=1"; do
  echo "== $env"
  env $env python bench.py --prefill 512 --type q4_0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-130
  env $env python bench.py --prefill 512 --shape llama2-7b --type q4_0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-130
done
LLMK_LIB=$L/libllmk_debug.so LLMK_PF_PLAN=1 python tests/host_tools/pf_trace.py --type q4_0 w13 2>&1 | grep -E "GEMM|prologue|step  [0-4]|exit"
