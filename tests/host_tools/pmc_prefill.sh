#!/usr/bin/env bash
# SQ counters of the prefill GEMM (pf_gemm_kernel) under `bench.py --prefill 256`; own PMC passes, no --stats.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"; do
  d=/tmp/pf_$(echo $set | md5sum | cut -c1-6); rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python $ROOT/bench.py --prefill 256 > /dev/null 2>$d.err || tail -2 $d.err
  python - "$d" <<'PY'
import csv, glob, collections, statistics, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pf_gemm_kernel" in k or "attn_kernel" in k or "pf_epi" in k or "pf_norm" in k:
            d[k[:60]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in d.items():
    n = len(next(iter(v.values())))
    dur = statistics.mean(x[1] for x in next(iter(v.values())))
    print(k, "launches", n, "avg_ns", int(dur), {c: int(statistics.mean(y[0] for y in x)) for c, x in v.items()})
PY
done
