timeout 900 python -m pytest tests/test_tp70_gpu.py -x -q 2>&1 | grep -E "passed|failed|warn" | tail -3
