L=$PWD/llm.f90_amd/csrc
LLMK_LIB=$L/libllmk_debug.so LLMK_TK_TRACE=1 python tests/host_tools/tk_trace.py --shape llama2-7b --type q4_0 --pos 130 2>&1 | grep -E "segment|attention CUs|non-attention|layer time|token kernel alone|L16|attention \(" | cut -c1-420 | tee gpurun_out/ab/trace_7b_final.txt
python -m pytest tests/test_parity_gpu.py -x -q -k "7b or q4" 2>&1 | tail -2
