#!/usr/bin/env bash
# the q4_0 persistent-kernel parity tests (TinyLlama q4_0 against the oracle; Llama-2-7B: column geometry against the oracle, full
# depth against the real reference's golden, full-shape properties, the f16-range fallback), then the bench lines of both shapes:
# the check every change to the q4_0 unit kernels goes through
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/${LLMK_JOB_TAG:-q16}
OUT=gpurun_out/${LLMK_JOB_TAG:-q16}
timeout 1800 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "llama2_7b or tinyllama_q4_0" 2>&1 | tail -25 | tee $OUT/parity.log
timeout 600 python bench.py --no-cpu-baseline --shape llama2-7b --type q4_0 > $OUT/bench_7b.json 2> $OUT/bench_7b.err; cut -c1-600 $OUT/bench_7b.json
timeout 600 python bench.py --no-cpu-baseline --type q4_0 > $OUT/bench_tinyllama_q4_0.json 2> $OUT/bench_tl.err; cut -c1-900 $OUT/bench_tinyllama_q4_0.json
