#!/usr/bin/env bash
# Round-3 evidence run (GPU box): per BASELINE.json single-GPU configuration an un-profiled bench line, the pipelined-greedy
# line, a rocprofv3 --kernel-trace --stats summary and a FETCH_SIZE pass (separate runs: never --pmc together with
# --stats), reduced into gpurun_out/prof_r03/; then the 70B rank's kernels and the prefill line.
#   usage: bash tests/host_tools/prof_r03.sh [tag]
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
reduce_stats() {
  python - "$1" "$2" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("Total_Duration_Ns", 0)) or 0))
with open(sys.argv[2], "w") as o:
    if rows:
        w = csv.DictWriter(o, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows[:8]:
            w.writerow(r)
print(open(sys.argv[2]).read()[:1200])
PY
}
run_cfg() {  # name, cpu-baseline flag, bench args...
  local name=$1 cb=$2; shift 2
  echo "=== $name: $*"
  timeout 400 python $ROOT/bench.py "$@" $cb > $OUT/${name}_bench.json 2> $OUT/${name}_bench.err
  cut -c1-600 $OUT/${name}_bench.json
  timeout 300 python $ROOT/bench.py "$@" --no-cpu-baseline --greedy-on-device > $OUT/${name}_bench_greedy_pipeline.json 2>> $OUT/${name}_bench.err
  cut -c1-300 $OUT/${name}_bench_greedy_pipeline.json
  rm -rf /tmp/st_$name /tmp/pm_$name
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-graph > $OUT/${name}_bench_under_rocprof.json 2> /tmp/st_$name.err || tail -3 /tmp/st_$name.err
  reduce_stats /tmp/st_$name $OUT/${name}_kernel_stats.csv
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pm_$name -- python $ROOT/bench.py "$@" --no-cpu-baseline --steps 20 --warmup 4 --repeats 1 > /dev/null 2> /tmp/pm_$name.err || tail -3 /tmp/pm_$name.err
  local cc=$(find /tmp/pm_$name -name "*counter_collection.csv" | head -1)
  [ -n "$cc" ] && python $ROOT/profiles/summarize_pmc.py "$cc" FETCH_SIZE | head -6 | tee $OUT/${name}_pmc_fetch_size.csv
}
run_cfg tinyllama_f32 ""
timeout 300 python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/tinyllama_f32_bench_driver_workload.json 2>/dev/null; cut -c1-400 $OUT/tinyllama_f32_bench_driver_workload.json
run_cfg tinyllama_f16 --no-cpu-baseline --type f16
run_cfg llama2-7b_q4_0 --no-cpu-baseline --shape llama2-7b --type q4_0
echo "=== 70B rank kernels"
timeout 300 python $ROOT/tests/host_tools/tp_rank_time.py 4 8 2>&1 | tail -8 | tee $OUT/tp70_rank_kernels.txt
echo "=== prefill 512"
timeout 300 python $ROOT/bench.py --prefill 512 > $OUT/prefill512_bench.json 2> $OUT/prefill512_bench.err
cut -c1-900 $OUT/prefill512_bench.json
rm -rf /tmp/st_pf
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_pf -- python $ROOT/bench.py --prefill 512 > /dev/null 2> /tmp/st_pf.err || tail -3 /tmp/st_pf.err
reduce_stats /tmp/st_pf $OUT/prefill512_kernel_stats.csv
timeout 300 python $ROOT/bench.py --prefill 512 --shape llama2-7b --type q4_0 > $OUT/prefill512_llama2-7b_q4_0_bench.json 2>/dev/null; cut -c1-300 $OUT/prefill512_llama2-7b_q4_0_bench.json
# f16 / q4_0 weights: the f16 matrix instruction (default) beside the f32 instruction (LLMK_PF_F32_MFMA=1), same box
for t in f32 f16 q4_0; do
  timeout 300 python $ROOT/bench.py --prefill 512 --type $t > $OUT/prefill512_tinyllama_${t}_bench.json 2>/dev/null; cut -c1-200 $OUT/prefill512_tinyllama_${t}_bench.json
  LLMK_PF_F32_MFMA=1 timeout 300 python $ROOT/bench.py --prefill 512 --type $t > $OUT/prefill512_tinyllama_${t}_f32_instruction_bench.json 2>/dev/null; cut -c1-200 $OUT/prefill512_tinyllama_${t}_f32_instruction_bench.json
done
LLMK_PF_F32_MFMA=1 timeout 300 python $ROOT/bench.py --prefill 512 --shape llama2-7b --type q4_0 > $OUT/prefill512_llama2-7b_q4_0_f32_instruction_bench.json 2>/dev/null; cut -c1-200 $OUT/prefill512_llama2-7b_q4_0_f32_instruction_bench.json
rm -rf /tmp/st_pfh
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_pfh -- python $ROOT/bench.py --prefill 512 --type f16 > /dev/null 2> /tmp/st_pfh.err || tail -3 /tmp/st_pfh.err
reduce_stats /tmp/st_pfh $OUT/prefill512_tinyllama_f16_kernel_stats.csv
echo "=== KV-length curve"
for a in "" "--type f16" "--shape llama2-7b"; do timeout 300 python $ROOT/tests/host_tools/tk_curve.py $a 1 256 512 1024 2048 2>&1 | tail -1; done | tee $OUT/kv_length_curve.txt
ls -la $OUT
