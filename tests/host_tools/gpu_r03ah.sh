timeout 600 python -m pytest tests/test_prefill_gpu.py -x -q -k "range_guard" 2>&1 | tail -8
LLMK_PF_F32_MFMA=1 timeout 600 python -m pytest tests/test_prefill_gpu.py -x -q 2>&1 | tail -3
