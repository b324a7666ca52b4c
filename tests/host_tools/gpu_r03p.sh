python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r03p_pytest.log; cat gpurun_out/r03p_pytest.log
L=$PWD/llm.f90_amd/csrc
for t in f16 f32; do
LLMK_LIB=$L/libllmk_debug.so LLMK_TK_TRACE=1 LLMK_TK_NOSYNC=1 python tests/host_tools/tk_trace.py --shape tinyllama --type $t --pos 130 2>&1 | grep -E "segment|attention CUs|non-attention|layer time|token kernel alone" | cut -c1-420 | tee gpurun_out/ab/trace_${t}_nosync.txt
LLMK_LIB=$L/libllmk_debug.so LLMK_TK_TRACE=1 python tests/host_tools/tk_trace.py --shape tinyllama --type $t --pos 130 2>&1 | grep -E "segment|attention CUs|non-attention|layer time|token kernel alone" | cut -c1-420 | tee gpurun_out/ab/trace_${t}_sync.txt
done
