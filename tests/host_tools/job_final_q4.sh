#!/usr/bin/env bash
# round 6, final tree: the q4_0 configurations' evidence again (their kernel changed after the r06 evidence run: scale records), the 7B
# decomposition with the structure line, one bench line per other configuration as a cross-check on the same box
cd "$(dirname "$0")/../.." || exit 1
export LLMK_JOB_TAG=r06f
bash tests/host_tools/gpu_job.sh profcfg r06f llama2-7b_q4_0 --shape llama2-7b --type q4_0
bash tests/host_tools/gpu_job.sh profcfg r06f llama2-7b_q4_0_q6k --shape llama2-7b --type q4_0 --cls-q6k
bash tests/host_tools/gpu_job.sh profcfg r06f mistral-7b_q4_0_q6k --shape mistral-7b --type q4_0 --cls-q6k
timeout 300 python bench.py --no-cpu-baseline --type q4_0 > gpurun_out/prof_r06f/tinyllama_q4_0_bench.json 2>/dev/null; cut -c1-300 gpurun_out/prof_r06f/tinyllama_q4_0_bench.json
bash tests/host_tools/decomp.sh r06f_decomp7 "--shape llama2-7b --type q4_0" base nohbm debug debug:nosync d_nohbm:nosync
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/prof_r06f/tinyllama_f32_bench.json 2>/dev/null; cut -c1-200 gpurun_out/prof_r06f/tinyllama_f32_bench.json
timeout 300 python bench.py --no-cpu-baseline --type f16 > gpurun_out/prof_r06f/tinyllama_f16_bench.json 2>/dev/null; cut -c1-200 gpurun_out/prof_r06f/tinyllama_f16_bench.json
