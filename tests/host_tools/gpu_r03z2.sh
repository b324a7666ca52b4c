L=$PWD/llm.f90_amd/csrc
for t in f16 f32; do
LLMK_LIB=$L/libllmk_debug.so LLMK_TK_TRACE=1 python tests/host_tools/tk_trace.py --type $t --pos 130 2>&1 | cut -c1-600 | tee gpurun_out/ab/trace_hs_$t.txt | head -60
done
