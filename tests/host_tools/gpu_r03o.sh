L=$PWD/llm.f90_amd/csrc
LLMK_LIB=$L/libllmk_debug.so LLMK_TK_TRACE=1 python tests/host_tools/tk_trace.py --shape llama2-7b --type q4_0 --pos 130 2>&1 | tail -24 | cut -c1-400 | tee gpurun_out/ab/trace_7b_mix.txt
LLMK_LIB=$L/libllmk_debug.so LLMK_TK_TRACE=1 LLMK_TK_NOSYNC=1 python tests/host_tools/tk_trace.py --shape llama2-7b --type q4_0 --pos 130 2>&1 | tail -24 | cut -c1-400 | tee gpurun_out/ab/trace_7b_mix_nosync.txt
