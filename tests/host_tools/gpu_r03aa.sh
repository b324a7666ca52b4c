L=$PWD/llm.f90_amd/csrc
mkdir -p gpurun_out/as
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_decode_greedy_gpu.py -x -q 2>&1 | tail -5
for lib in libllmk.so libllmk_as0.so; do
  echo "== $lib"
  LLMK_LIB=$L/$lib python tests/host_tools/tk_curve.py 1 128 256 257 384 512 768 1024 1536 2048 2>&1 | tail -1
  LLMK_LIB=$L/$lib python tests/host_tools/tk_curve.py --type f16 1 256 257 512 1024 2048 2>&1 | tail -1
  LLMK_LIB=$L/$lib python tests/host_tools/tk_curve.py --shape llama2-7b 1 256 257 512 1024 1536 2048 2>&1 | tail -1
done 2>&1 | tee gpurun_out/as/curve.txt
for a in "" "--type f16" "--shape llama2-7b --type q4_0"; do python bench.py --no-cpu-baseline $a | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('$a', round(l['value'],1), l['roofline']['us_per_launch'])"; done
