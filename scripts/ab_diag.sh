# A/B: the LM diagonal and the gradient norms as tasks of the factorisation's launch (default) against their own launch (BSGPU_POSE_DIAG_LAUNCH=1)
B="python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0"
ex() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('phases_us_per_lm_step') or {}
print(sys.argv[1], d['value'], d['config'].get('final_cost'), {k: p.get(k) for k in ('assemble_other','factor','backsolve')})" "$1"; }
for r in 1 2; do
$B 2>/dev/null | ex in_chol
BSGPU_POSE_DIAG_LAUNCH=1 $B 2>/dev/null | ex launch
done
$B --workload c3 2>/dev/null | ex c3_in_chol
BSGPU_POSE_DIAG_LAUNCH=1 $B --workload c3 2>/dev/null | ex c3_launch
python scripts/small_window.py | tail -4
BSGPU_POSE_DIAG_LAUNCH=1 python scripts/small_window.py | tail -4
