timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -k "in_parts or 7b_full" 2>&1 | tail -8
