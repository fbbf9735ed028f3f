#!/bin/bash
# the factorisation's updates with turns (BSGPU_CHOL_NOTURN=0) / as unordered atomic adds (default): bench lines of C2, C3 and the small window, one box
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"
B="timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0 --steps 30"
ex() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('phases_us_per_lm_step') or {}
print(sys.argv[1], d['value'], 'factor', p.get('factor'), 'backsolve', p.get('backsolve'), 'pairs', p.get('pairs'), 'cost %.12e' % d['config']['final_cost'])" "$1"; }
for i in 1 2; do
  for v in 0 1; do
    BSGPU_CHOL_NOTURN=$v $B 2>/dev/null | ex "noturn=$v c2"
    BSGPU_CHOL_NOTURN=$v $B --workload c3 2>/dev/null | ex "noturn=$v c3"
    BSGPU_CHOL_NOTURN=$v timeout 120 python scripts/small_window.py 2>&1 | head -1
  done
done
