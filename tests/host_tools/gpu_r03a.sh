set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03a_pytest.log; cat gpurun_out/r03a_pytest.log
bash tests/host_tools/ab_r03.sh 2>&1 | tail -20
python bench.py --no-cpu-baseline --greedy-on-device > gpurun_out/ab/greedy_f32.json 2>gpurun_out/ab/greedy_err.log; cat gpurun_out/ab/greedy_f32.json
python bench.py --no-cpu-baseline --type f16 > gpurun_out/ab/f16.json; python bench.py --no-cpu-baseline --type f16 --greedy-on-device > gpurun_out/ab/f16_greedy.json
python bench.py --no-cpu-baseline --shape llama2-7b --type q4_0 > gpurun_out/ab/7b.json
python - <<'PY'
import json
for f in ("greedy_f32", "f16", "f16_greedy", "7b"):
    try:
        l = json.load(open(f"gpurun_out/ab/{f}.json")); print(f, round(l["value"], 1), round(l["roofline"]["us_per_launch"], 1), l["config"]["path"])
    except Exception as e: print(f, "ERR", e)
PY
