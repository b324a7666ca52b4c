mkdir -p gpurun_out/final
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/final/pytest_gpu.txt
