#!/usr/bin/env bash
# The persistent kernel taken apart with experiment builds (token_kernel.h LLMK_EXP_*, tests/host_tools/build_variant.sh):
#   decomp.sh TAG "<bench.py args>" LIB[:nosync] ...      LIB = base | debug | a variant name (csrc/variants/libllmk_NAME.so)
# ":nosync" runs the library with LLMK_TK_NOSYNC=1 (debug builds only: nothing waits for an exchange tag).  One bench line per
# entry (--steps 64): tok/s and the kernel's microseconds at KV length 72 (roofline.us_per_launch) -> gpurun_out/TAG/decomp.txt
cd "$(dirname "$0")/../.." || exit 1
TAG=$1; ARGS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
lib() { case $1 in base) echo "$PWD/llm.f90_amd/csrc/libllmk.so";; debug) echo "$PWD/llm.f90_amd/csrc/libllmk_debug.so";; *) echo "$PWD/llm.f90_amd/csrc/variants/libllmk_$1.so";; esac; }
echo "# bench.py --no-cpu-baseline --steps 64 --warmup 8 $ARGS" | tee -a $OUT/decomp.txt
for e in "$@"; do
  v=${e%%:*}; ns=""; [ "$e" != "$v" ] && ns=1
  line=$(env LLMK_LIB=$(lib $v) ${ns:+LLMK_TK_NOSYNC=1} timeout 400 python bench.py --no-cpu-baseline --steps 64 --warmup 8 $ARGS 2>/dev/null | tail -1)
  python - "$e" "$line" <<'PY' | tee -a $OUT/decomp.txt
import json, sys
try:
    b = json.loads(sys.argv[2]); print(f'{sys.argv[1]:24s} {b["value"]:9.1f} tok/s   kernel {b.get("roofline", {}).get("us_per_launch", 0):8.1f} us')
except Exception as ex:
    print(f'{sys.argv[1]:24s} no line ({ex})')
PY
done
