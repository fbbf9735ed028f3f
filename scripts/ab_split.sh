#!/bin/bash
# A/B of the factorisation's kFusedSplit depth (dense_plan.h) on C2 / C3 / the reference-sized window: bash scripts/ab_split.sh
B="timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0"
ex() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('phases_us_per_lm_step') or {}
print(sys.argv[1], d['value'], 'factor', p.get('factor'), 'backsolve', p.get('backsolve'), 'cost', d['config']['final_cost'])" "$1"; }
for sp in 2 2; do
  BSGPU_CHOL_SPLIT=$sp $B 2>/dev/null | ex c2-split=$sp
done
for sp in 2 2; do
  BSGPU_CHOL_SPLIT=$sp $B --workload c3 2>/dev/null | ex c3-split=$sp
done
for sp in 2 2; do
  echo "small split=$sp"; BSGPU_CHOL_SPLIT=$sp timeout 120 python scripts/small_window.py 2>&1 | head -2
done
