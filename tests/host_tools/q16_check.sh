#!/usr/bin/env bash
# the Llama-2-7B q4_0 parity tests (column geometry against the oracle, full depth against the real reference's golden, full-shape
# properties), then the bench line of that configuration: the check every change to the q4_0 unit kernels goes through
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/${LLMK_JOB_TAG:-q16}
OUT=gpurun_out/${LLMK_JOB_TAG:-q16}
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "llama2_7b" 2>&1 | tail -25 | tee $OUT/parity.log
timeout 600 python bench.py --no-cpu-baseline --shape llama2-7b --type q4_0 > $OUT/bench_7b.json 2> $OUT/bench_7b.err; cut -c1-900 $OUT/bench_7b.json
