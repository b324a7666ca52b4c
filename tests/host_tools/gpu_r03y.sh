python tests/host_tools/tp_rank_time.py 4 8 2>&1 | tail -8
python -m pytest tests/test_parity_gpu.py tests/test_tp_gpu.py -x -q -k "q4 or quantised or 7b" 2>&1 | tail -2
python bench.py --no-cpu-baseline --shape tinyllama --type q4_0 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('tinyllama q4_0', round(l['value'],1), l['config']['path'])"
