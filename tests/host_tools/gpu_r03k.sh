for i in 1 2 3; do
for sw in 1 0; do
LLMK_SPIN_WAIT=$sw python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('spin', $sw, 'short', round(l['value'],1), l['value_all'], round(l['roofline']['us_per_launch'],1))"
done; done
for sw in 1 0; do
LLMK_SPIN_WAIT=$sw python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('spin', $sw, 'default', round(l['value'],1), round(l['roofline']['us_per_launch'],1))"
done
