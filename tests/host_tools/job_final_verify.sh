#!/usr/bin/env bash
# round-end verification on ONE fresh box: what the driver runs (pytest -m gpu, smoke(), bench.py), then the same suite with poisoned allocations
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/final/suite.log 2>&1; grep -a "passed\|failed" gpurun_out/final/suite.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
timeout 600 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; cut -c1-700 gpurun_out/final/bench.json
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_driver_workload.json 2>> gpurun_out/final/bench.err; cut -c1-300 gpurun_out/final/bench_driver_workload.json
LLMK_POISON=1 timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/final/poison.log 2>&1; grep -a "passed\|failed" gpurun_out/final/poison.log | tail -1
