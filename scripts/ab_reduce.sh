# A/B: the end-of-step reduction riding in the evaluation launched ahead of the decision (default) against its own launch (BSGPU_REDUCE_LAUNCH=1)
B="python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0"
ex() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], d['config'].get('final_cost'), d['config'].get('lm_iterations_per_solve'))" "$1"; }
for r in 1 2; do
$B 2>/dev/null | ex rides
BSGPU_REDUCE_LAUNCH=1 $B 2>/dev/null | ex launch
done
python scripts/small_window.py | tail -4
BSGPU_REDUCE_LAUNCH=1 python scripts/small_window.py | tail -4
