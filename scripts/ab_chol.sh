#!/bin/bash
# A/B of the factorisation's plan knobs on C2 / C3 (one line per setting): bash scripts/ab_chol.sh
B="python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0"
ex() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('phases_us_per_lm_step') or {}
print(sys.argv[1], d['value'], 'factor', p.get('factor'), 'backsolve', p.get('backsolve'))" "$1"; }
for beta in 0 0.3 0.5 0.75 1; do BSGPU_TICKET_BETA=$beta $B 2>/dev/null | ex beta=$beta; done
for d in 3 5; do BSGPU_DIM_ORDER_DEPTH=$d $B 2>/dev/null | ex depth=$d; done
BSGPU_DIM_ORDER=0 $B 2>/dev/null | ex tile-order
$B --workload c3 2>/dev/null | ex c3
BSGPU_DIM_ORDER=0 $B --workload c3 2>/dev/null | ex c3-tile-order
