# A/B: appendix tiles carried by the panel before them (BSGPU_CHOL_EXT, default on) against a panel of their own (=0)
B="python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0"
ex() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('phases_us_per_lm_step') or {}
print(sys.argv[1], d['value'], d['config'].get('final_cost'), {k: p.get(k) for k in ('factor','backsolve')})" "$1"; }
for r in 1 2; do
$B 2>/dev/null | ex ext
BSGPU_CHOL_EXT=0 $B 2>/dev/null | ex no_ext
done
$B --workload c3 2>/dev/null | ex c3_ext
BSGPU_CHOL_EXT=0 $B --workload c3 2>/dev/null | ex c3_no_ext
