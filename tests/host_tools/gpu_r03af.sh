timeout 900 python -m pytest tests/test_prefill_gpu.py tests/test_host_gpu.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
for a in "" "--type f16" "--type q4_0" "--shape llama2-7b --type q4_0"; do
python bench.py --prefill 512 $a --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100
done
