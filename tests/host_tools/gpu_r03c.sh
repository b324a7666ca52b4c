set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03c_pytest.log; cat gpurun_out/r03c_pytest.log
mkdir -p gpurun_out/ab
out=gpurun_out/ab; : > $out/bisect_f32.jsonl
one() { local label=$1 dir=$2; shift 2; local line; line=$(cd "$dir" && python bench.py --no-cpu-baseline "$@" 2>>"$OLDPWD/$out/err.log" | tail -1); echo "{\"build\": \"$label\", \"args\": \"$*\", \"line\": $line}" >> $out/bisect_f32.jsonl; }
for i in 1 2; do
  for c in r01 544076a e7edadb fb04774 eecc2fa 874bd18; do one $c ab_$c --steps 20 --warmup 5; done
  one head . --steps 20 --warmup 5 --repeats 1
done
for c in r01 544076a e7edadb fb04774 eecc2fa 874bd18; do one $c ab_$c; done
one head .
python - <<'PY'
import json
for r in map(json.loads, open("gpurun_out/ab/bisect_f32.jsonl")):
    l = r["line"]; print(f'{r["build"]:12s} {r["args"]:32s} {l["value"]:8.1f} tok/s  kernel {l["roofline"]["us_per_launch"]:7.1f} us')
PY
python bench.py --no-cpu-baseline --greedy-on-device > gpurun_out/ab/greedy_f32.json 2>gpurun_out/ab/greedy_err.log
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --greedy-on-device > gpurun_out/ab/greedy_f32_short.json 2>>gpurun_out/ab/greedy_err.log
python bench.py --no-cpu-baseline --type f16 --greedy-on-device > gpurun_out/ab/f16_greedy.json
python bench.py --no-cpu-baseline --shape llama2-7b --type q4_0 --greedy-on-device > gpurun_out/ab/7b_greedy.json
python - <<'PY'
import json
for f in ("greedy_f32", "greedy_f32_short", "f16_greedy", "7b_greedy"):
    try:
        l = json.load(open(f"gpurun_out/ab/{f}.json")); print(f, round(l["value"], 1), round(l["roofline"]["us_per_launch"], 1), l["config"]["path"], l.get("value_all"))
    except Exception as e: print(f, "ERR", e)
PY
tail -5 gpurun_out/ab/greedy_err.log
df -h /dev/shm /tmp | tail -2; free -g | head -2
python tests/host_tools/tp_load_rss.py llama2-7b 2>&1 | tail -4 | tee gpurun_out/ab/rss_7b.jsonl
