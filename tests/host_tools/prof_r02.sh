#!/usr/bin/env bash
# Round-2 evidence run (GPU box): for each BASELINE.json single-GPU configuration an un-profiled bench line, a
# rocprofv3 --kernel-trace --stats summary and a FETCH_SIZE pass (separate runs: never --pmc together with --stats),
# reduced into gpurun_out/prof_r02/; profiles/README.md says which files are copied into profiles/.
#   usage: bash tests/host_tools/prof_r02.sh [tag]
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
reduce_stats() {  # dir -> first rows of kernel_stats.csv
  python - "$1" "$2" <<'PY'
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = []
for f in files:
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
run_cfg() {  # name, bench args...
  local name=$1; shift
  echo "=== $name: $*"
  timeout 300 python $ROOT/bench.py "$@" --no-cpu-baseline > $OUT/${name}_bench.json 2> $OUT/${name}_bench.err
  cut -c1-700 $OUT/${name}_bench.json
  rm -rf /tmp/st_$name /tmp/pm_$name
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$name -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-graph > $OUT/${name}_bench_under_rocprof.json 2> /tmp/st_$name.err || tail -3 /tmp/st_$name.err
  reduce_stats /tmp/st_$name $OUT/${name}_kernel_stats.csv
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pm_$name -- python $ROOT/bench.py "$@" --no-cpu-baseline --steps 20 --warmup 4 > /dev/null 2> /tmp/pm_$name.err || tail -3 /tmp/pm_$name.err
  local cc=$(find /tmp/pm_$name -name "*counter_collection.csv" | head -1)
  [ -n "$cc" ] && python $ROOT/profiles/summarize_pmc.py "$cc" FETCH_SIZE | head -6 | tee $OUT/${name}_pmc_fetch_size.csv
}
run_cfg tinyllama_f32
run_cfg tinyllama_f16 --type f16
run_cfg llama2-7b_q4_0 --shape llama2-7b --type q4_0
# where the 7B q4_0 token kernel's cycles go (SQ counters, own passes)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  d=/tmp/sq_$(echo $set | md5sum | cut -c1-6); rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python $ROOT/bench.py --shape llama2-7b --type q4_0 --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2>$d.err || tail -2 $d.err
  python - "$d" <<'PY' | tee -a $OUT/llama2-7b_q4_0_pmc_sq.txt
import csv, glob, collections, statistics, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "token_kernel" in r["Kernel_Name"]:
            d["token_kernel<llama2-7b,q4_0>"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(k, {c: int(statistics.mean(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
done
# batched prefill: bench line, kernel stats of the same command, GEMM block timelines, MFMA/LDS probe
echo "=== prefill 512"
timeout 300 python $ROOT/bench.py --prefill 512 > $OUT/prefill512_bench.json 2> $OUT/prefill512_bench.err
cut -c1-900 $OUT/prefill512_bench.json
rm -rf /tmp/st_pf
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_pf -- python $ROOT/bench.py --prefill 512 > /dev/null 2> /tmp/st_pf.err || tail -3 /tmp/st_pf.err
reduce_stats /tmp/st_pf $OUT/prefill512_kernel_stats.csv
# the other weight types through the same prefill, profiled the same way
for cfg in "tinyllama f16" "tinyllama q4_0" "llama2-7b q4_0"; do
  set -- $cfg; n=prefill512_$1_$2
  timeout 600 python $ROOT/bench.py --prefill 512 --shape $1 --type $2 > $OUT/${n}_bench.json 2> $OUT/${n}_bench.err
  cut -c1-300 $OUT/${n}_bench.json
  rm -rf /tmp/st_$n
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$n -- python $ROOT/bench.py --prefill 512 --shape $1 --type $2 > /dev/null 2> /tmp/st_$n.err || tail -3 /tmp/st_$n.err
  reduce_stats /tmp/st_$n $OUT/${n}_kernel_stats.csv
done
# HBM traffic of the w1|w3 GEMM alone (own PMC pass; only that GEMM is launched: 44 launches walking the layers)
cat > /tmp/pf_w13_only.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["LLMK_ROOT"])
import llm_f90_amd
from llm_f90_amd import llmk
from llm_f90_amd.tools import gguf
m = llmk.Llmk(gguf.synth_fused(gguf.SHAPES["tinyllama"], 1, 0))
print(m.time_kernel(7, 44))
PY
rm -rf /tmp/pm_pf
LLMK_ROOT=$ROOT timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pm_pf -- python /tmp/pf_w13_only.py > /dev/null 2> /tmp/pm_pf.err || tail -3 /tmp/pm_pf.err
cc=$(find /tmp/pm_pf -name "*counter_collection.csv" | head -1)
[ -n "$cc" ] && python $ROOT/profiles/summarize_pmc.py "$cc" FETCH_SIZE | head -4 | tee $OUT/prefill_w13_f32_pmc_fetch_size.csv
for g in w13 wqkv wo w2; do for nr in 1 2; do
  echo "--- $g, $nr row group(s) per wave"; LLMK_PF_PLAN=$nr LLMK_LIB=$ROOT/llm.f90_amd/csrc/libllmk_debug.so timeout 120 python $ROOT/tests/host_tools/pf_trace.py $g
done; done > $OUT/prefill_gemm_timeline.txt 2>&1
tail -30 $OUT/prefill_gemm_timeline.txt
[ -x $ROOT/llm.f90_amd/csrc/probes/pf_mfma_probe ] && timeout 120 $ROOT/llm.f90_amd/csrc/probes/pf_mfma_probe > $OUT/pf_mfma_probe.txt 2>&1
ls -la $OUT
