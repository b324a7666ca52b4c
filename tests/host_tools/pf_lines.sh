#!/usr/bin/env bash
# the four prefill bench lines (512-token prompt) + the prefill tests: what a change to prefill.h goes through
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/${LLMK_JOB_TAG:-pf}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_prefill_gpu.py -m gpu -q 2>&1 | tail -3 | tee $OUT/prefill_tests.log
for cfg in "tinyllama_f32" "tinyllama_f16 --type f16" "tinyllama_q4_0 --type q4_0" "llama2-7b_q4_0 --shape llama2-7b --type q4_0"; do
  set -- $cfg; name=prefill512_$1; shift
  timeout 300 python bench.py --prefill 512 "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_bench.err
  python -c "
import json; d=json.load(open('$OUT/${name}_bench.json')); print('$name', round(d['value']), round(d['roofline']['us_per_launch'],1))"
done
