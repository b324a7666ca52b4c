L=$PWD/llm.f90_amd/csrc
mkdir -p gpurun_out/as; : > gpurun_out/as/step.txt
for lib in libllmk.so libllmk_st128.so libllmk_st64.so libllmk_st32.so; do
  echo "== $lib"
  LLMK_LIB=$L/$lib python tests/host_tools/tk_curve.py --shape llama2-7b 1 33 65 96 129 192 256 2>&1 | tail -1
  LLMK_LIB=$L/$lib python tests/host_tools/tk_curve.py 1 65 129 192 256 2>&1 | tail -1
  LLMK_LIB=$L/$lib python tests/host_tools/tk_curve.py --type f16 1 65 129 192 256 2>&1 | tail -1
  for a in "--shape llama2-7b --type q4_0" "" "--type f16"; do LLMK_LIB=$L/$lib python bench.py --no-cpu-baseline $a | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('bench $a', round(l['value'],1), round(l['roofline']['us_per_launch'],1))"; done
done 2>&1 | tee gpurun_out/as/step.txt
