#!/bin/bash
# Round-3 A/B of the f32 headline on ONE lease: the tree at 94ab266 (round-1 end, extracted to ab_r01/ and built there) against
# HEAD, interleaved, three times each, with the default workload (-n 256) and with the driver's (--steps 20 --warmup 5).
# Lines land in gpurun_out/ab/ (copy to profiles/r03_ab_f32.jsonl).
set -u
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/ab; mkdir -p $out; : > $out/ab_f32.jsonl
one() {  # label dir args...
    local label=$1 dir=$2; shift 2
    local line; line=$(cd "$dir" && python bench.py --no-cpu-baseline "$@" 2>>"$OLDPWD/$out/err.log" | tail -1)
    echo "{\"build\": \"$label\", \"args\": \"$*\", \"line\": $line}" >> $out/ab_f32.jsonl
}
for i in 1 2 3; do
    one r01_94ab266 ab_r01
    one head .
done
for i in 1 2 3; do
    one r01_94ab266 ab_r01 --steps 20 --warmup 5
    one head . --steps 20 --warmup 5 --repeats 1
done
one head . --steps 20 --warmup 5
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/ab/ab_f32.jsonl")]
for r in rows:
    l = r["line"]
    print(f'{r["build"]:12s} {r["args"]:32s} {l["value"]:8.1f} tok/s  kernel {l["roofline"]["us_per_launch"]:7.1f} us  {l.get("value_all", "")}')
PY
