#!/bin/bash
# one bench line per workload (C2, C3, the reference-sized window), twice:  bash scripts/ab_env.sh [label]
B="timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0"
ex() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('phases_us_per_lm_step') or {}
print(sys.argv[1], d['value'], 'factor', p.get('factor'), 'backsolve', p.get('backsolve'), 'pairs', p.get('pairs'), 'cost', d['config']['final_cost'])" "$1"; }
for i in 1 2; do
  $B 2>/dev/null | ex "${1:-x} c2"
  $B --workload c3 2>/dev/null | ex "${1:-x} c3"
  timeout 120 python scripts/small_window.py 2>&1 | head -1
done
