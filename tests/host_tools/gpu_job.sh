#!/usr/bin/env bash
# ONE parametrised lease script for the GPU box (replaces round 3's 37 one-off gpu_r03*.sh; round-3 verdict, hygiene):
#
#   gpurun --timeout 1500 -- 'bash tests/host_tools/gpu_job.sh <job> [args] [-- <job> [args]]...'
#
# jobs (each writes under gpurun_out/<tag>/, tag = LLMK_JOB_TAG or "job"):
#   suite [pytest args]      python -m pytest tests -m gpu -q [args]            -> suite.log
#   poison [pytest args]     the same under LLMK_POISON=1 (every device allocation pre-filled with NaN bytes: an
#                            uninitialised read fails every time instead of once in fifteen runs)  -> poison.log
#   repeat N <pytest args>   N runs of the given tests, pass/fail per run        -> repeat.log (+ tp70_fail/ on a failure)
#   nine N                   the control behind tests/test_tp70_gpu.py::_run_ranks: N runs of the jitter test with this pytest
#                            process holding a GPU context, rank 0 in-process (8 GPU processes) and spawned (9)   -> nine.log
#   dirty                    does a process see another (or its own earlier) process's freed VRAM?   -> dirty.log
#   bench NAME [bench args]  python bench.py [args]                              -> NAME.json
#   ab NAME N <env>... : [bench args]   N interleaved pairs of bench.py with / without the env setting   -> NAME.jsonl
#   prof TAG                 the evidence run: per single-GPU configuration of BASELINE.json a bench line, the pipelined-greedy
#                            line, rocprofv3 --kernel-trace --stats and a FETCH_SIZE pass (separate runs); the 70B rank's
#                            kernels; the prefill lines with their own stats + FETCH_SIZE; the KV-length curve
#   pfprof                   the prefill lines of `prof` alone (bench lines of the four configurations, stats + FETCH_SIZE of two)
#   profcfg TAG NAME [args]  the evidence of ONE configuration (bench line, greedy line, stats, FETCH_SIZE, its KV-length curve)
#   pmc KIND                 SQ counter passes (q4 | prefill), own runs, no --stats
#   trace [args]             tests/host_tools/tk_trace.py on the debug library (LLMK_TK_TRACE=1)
#   rank                     tests/host_tools/tp_rank_time.py 4 8: the 70B rank's kernels one by one
#   run <command...>         anything else, verbatim
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=${LLMK_JOB_TAG:-job}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT" || exit 1
exec > >(tee -a "$OUT/job.log") 2>&1        # gpurun returns only the tail of stdout: the whole log comes back with gpurun_out/
export TMPDIR=/tmp
DBG=$ROOT/llm.f90_amd/csrc/libllmk_debug.so

reduce_stats() {  # rocprofv3 output dir -> top-8 kernel rows
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
print(open(sys.argv[2]).read()[:1500])
PY
}
stats_of() {  # name, bench args...: rocprofv3 --kernel-trace --stats of that bench command (eager: the profiler dies on hipGraphLaunch)
  local name=$1; shift
  rm -rf /tmp/st_$name
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -- python $ROOT/bench.py "$@" > $OUT/${name}_bench_under_rocprof.json 2> /tmp/st_$name.err) || tail -3 /tmp/st_$name.err
  reduce_stats /tmp/st_$name $OUT/${name}_kernel_stats.csv
}
fetch_of() {  # name, kernel-name filter, bench args...: FETCH_SIZE per launch (its own pass: never --pmc with --stats)
  local name=$1 filt=$2; shift 2
  rm -rf /tmp/pm_$name
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pm_$name -- python $ROOT/bench.py "$@" > /dev/null 2> /tmp/pm_$name.err) || tail -3 /tmp/pm_$name.err
  local cc; cc=$(find /tmp/pm_$name -name "*counter_collection.csv" | head -1)
  [ -n "$cc" ] && python $ROOT/profiles/summarize_pmc.py "$cc" FETCH_SIZE $filt | head -8 | tee $OUT/${name}_pmc_fetch_size.csv
}
run_cfg() {  # name, cpu-baseline flag, bench args...: bench line, pipelined-greedy line, rocprofv3 stats, FETCH_SIZE pass
  local name=$1 cb=$2; shift 2
  echo "=== $name: $*"
  timeout 400 python bench.py "$@" $cb > $OUT/${name}_bench.json 2> $OUT/${name}_bench.err; cut -c1-600 $OUT/${name}_bench.json
  timeout 300 python bench.py "$@" --no-cpu-baseline --greedy-on-device > $OUT/${name}_bench_greedy_pipeline.json 2>> $OUT/${name}_bench.err
  cut -c1-300 $OUT/${name}_bench_greedy_pipeline.json
  stats_of $name "$@" --no-cpu-baseline --no-graph
  fetch_of $name "" "$@" --no-cpu-baseline --steps 20 --warmup 4 --repeats 1
}
job_prof() {
  local cfg name
  run_cfg tinyllama_f32 ""
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/tinyllama_f32_bench_driver_workload.json 2>/dev/null; cut -c1-400 $OUT/tinyllama_f32_bench_driver_workload.json
  run_cfg tinyllama_f16 --no-cpu-baseline --type f16
  run_cfg llama2-7b_q4_0 --no-cpu-baseline --shape llama2-7b --type q4_0
  # round 6: a stock llama.cpp Q4_0 file's layout (q6_K classifier rows on the device), and Llama-2-7B f16 on the persistent kernel
  run_cfg llama2-7b_q4_0_q6k --no-cpu-baseline --shape llama2-7b --type q4_0 --cls-q6k
  run_cfg llama2-7b_f16 --no-cpu-baseline --shape llama2-7b --type f16
  echo "=== 70B rank kernels"; job_rank
  job_pfprof
  echo "=== KV-length curve"
  for a in "" "--type f16" "--shape llama2-7b"; do timeout 300 python tests/host_tools/tk_curve.py $a 1 256 512 1024 2048 2>&1 | tail -1; done | tee $OUT/kv_length_curve.txt
}
job_pfprof() {   # the prefill lines with their own stats + FETCH_SIZE passes
  local cfg name
  echo "=== prefill 512"
  for cfg in "tinyllama_f32" "tinyllama_f16 --type f16" "tinyllama_q4_0 --type q4_0" "llama2-7b_q4_0 --shape llama2-7b --type q4_0"; do
    set -- $cfg; name=prefill512_$1; shift
    timeout 300 python bench.py --prefill 512 "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_bench.err; cut -c1-700 $OUT/${name}_bench.json
  done
  stats_of prefill512_tinyllama_f16 --prefill 512 --type f16
  fetch_of prefill512_tinyllama_f16 pf_gemm --prefill 512 --type f16
  stats_of prefill512_llama2-7b_q4_0 --prefill 512 --shape llama2-7b --type q4_0
  fetch_of prefill512_llama2-7b_q4_0 pf_gemm --prefill 512 --shape llama2-7b --type q4_0
}
job_rank() { timeout 300 python tests/host_tools/tp_rank_time.py 4 8 2>&1 | tail -12 | tee $OUT/tp70_rank_kernels.txt; }
job_pmc() {
  local kind=$1 args filt set d
  if [ "$kind" = q4 ]; then args="--no-cpu-baseline --shape llama2-7b --type q4_0 --steps 20 --warmup 4 --repeats 1"; filt=token_kernel
  else args="--prefill 256"; filt=pf_; fi
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"; do
    d=/tmp/pmc_$(echo $set | md5sum | cut -c1-6); rm -rf $d
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python $ROOT/bench.py $args > /dev/null 2>$d.err) || tail -2 $d.err
    python - "$d" "$filt" <<'PY' | tee -a $OUT/pmc_$kind.txt
import csv, glob, collections, statistics, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            d[r["Kernel_Name"][:70]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in d.items():
    first = next(iter(v.values()))
    print(k, "launches", len(first), "avg_ns", int(statistics.mean(x[1] for x in first)), {c: int(statistics.mean(y[0] for y in x)) for c, x in v.items()})
PY
  done
}
job_nine() {   # the same two tests, the first gives the pytest process a GPU context; 8 vs 9 processes on the GPU
  local n=${1:-3} i v
  : > $OUT/nine.log
  for i in $(seq 1 $n); do
    for v in eight nine; do
      local t0=$(date +%s)
      if [ $v = nine ]; then export LLMK_TP_TEST_SPAWN_ALL=1; else unset LLMK_TP_TEST_SPAWN_ALL; fi
      if timeout 900 python -m pytest tests/test_decode_greedy_gpu.py tests/test_tp70_gpu.py -m gpu -q -k "returns_the_reference_ids or jitter" > /tmp/nine_$i$v.log 2>&1; then r=pass; else r=FAIL; fi
      echo "run $i $v processes: $r in $(( $(date +%s) - t0 )) s; $(grep -h "never arrived" /tmp/nine_$i$v.log | head -1 | cut -c1-200)" | tee -a $OUT/nine.log
    done
  done
  unset LLMK_TP_TEST_SPAWN_ALL
}
job_dirty() { timeout 300 python tests/host_tools/dirty_probe.py 2>&1 | tail -20 | tee $OUT/dirty.log; }
job_repeat() {
  local n=$1 i; shift
  : > $OUT/repeat.log
  for i in $(seq 1 $n); do
    if timeout 900 python -m pytest "$@" -m gpu -x -q > /tmp/rep_$i.log 2>&1; then echo "run $i: pass $(tail -1 /tmp/rep_$i.log)"; else echo "run $i: FAIL"; tail -40 /tmp/rep_$i.log; cp /tmp/rep_$i.log $OUT/repeat_fail_$i.log; fi | tee -a $OUT/repeat.log
  done
}
job_ab() {  # NAME N ENV... : bench args     (":" -- the job list itself is split at "--")
  local name=$1 n=$2; shift 2
  local envs=()
  while [ $# -gt 0 ] && [ "$1" != ":" ]; do envs+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  : > $OUT/$name.jsonl
  local i line
  for i in $(seq 1 $n); do
    line=$(timeout 400 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1); echo "{\"variant\": \"base\", \"line\": $line}" >> $OUT/$name.jsonl
    line=$(env "${envs[@]}" timeout 400 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1); echo "{\"variant\": \"${envs[*]}\", \"line\": $line}" >> $OUT/$name.jsonl
  done
  python - $OUT/$name.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l); b = r["line"]
    print(f'{r["variant"]:40s} {b["value"]:9.1f} {b["unit"]}  kernel {b.get("roofline", {}).get("us_per_launch", 0):8.1f} us')
PY
}

while [ $# -gt 0 ]; do
  job=$1; shift
  args=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  echo "##### $job ${args[*]:-}"
  case $job in
    suite)  timeout 1500 python -m pytest tests -m gpu -q "${args[@]}" > $OUT/suite.log 2>&1; tail -25 $OUT/suite.log ;;
    poison) LLMK_POISON=1 timeout 1500 python -m pytest tests -m gpu -q "${args[@]}" > $OUT/poison.log 2>&1; tail -25 $OUT/poison.log ;;
    repeat) job_repeat "${args[@]}" ;;
    dirty)  job_dirty ;;
    nine)   job_nine "${args[@]}" ;;
    bench)  name=${args[0]}; timeout 600 python bench.py "${args[@]:1}" > $OUT/$name.json 2> $OUT/$name.err; cut -c1-1200 $OUT/$name.json ;;
    ab)     job_ab "${args[@]}" ;;
    prof)   OUT=$ROOT/gpurun_out/prof_${args[0]:-r04}; mkdir -p $OUT; job_prof ;;
    profcfg) OUT=$ROOT/gpurun_out/prof_${args[0]}; mkdir -p $OUT; run_cfg "${args[1]}" --no-cpu-baseline "${args[@]:2}"; timeout 300 python tests/host_tools/tk_curve.py $(case "${args[1]}" in *f16*) echo --type f16;; *7b*) echo --shape llama2-7b;; esac) 1 256 512 1024 2048 2>&1 | tail -1 | tee $OUT/kv_length_curve_${args[1]}.txt ;;
    pfprof) OUT=$ROOT/gpurun_out/prof_${args[0]:-r05}; mkdir -p $OUT; job_pfprof ;;
    pmc)    job_pmc "${args[0]:-q4}" ;;
    trace)  LLMK_LIB=$DBG LLMK_TK_TRACE=1 timeout 300 python tests/host_tools/tk_trace.py "${args[@]}" 2>&1 | cut -c1-400 | tee $OUT/trace_$(echo "${args[*]}" | tr -c 'a-zA-Z0-9\n' _).txt | tail -70 ;;
    rank)   job_rank ;;
    run)    "${args[@]}" ;;
    *)      echo "gpu_job.sh: unknown job $job"; exit 2 ;;
  esac
done
