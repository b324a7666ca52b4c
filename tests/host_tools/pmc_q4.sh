cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TA_TA_BUSY GRBM_GUI_ACTIVE"; do
  d=/tmp/q4c_$(echo $set | md5sum | cut -c1-6)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py --shape llama2-7b --type q4_0 --no-cpu-baseline --no-graph --steps 4 --warmup 2 > /dev/null 2>/tmp/err.txt || tail -2 /tmp/err.txt
  python - "$d" <<PY
import csv,glob,collections,statistics,sys
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemv_q4_kernel<3" in r["Kernel_Name"] or "gemv_q4_kernel<1" in r["Kernel_Name"]:
            d[r["Kernel_Name"][11:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items():
    print(k, {c: int(statistics.mean(x)) for c,x in v.items()})
PY
done
