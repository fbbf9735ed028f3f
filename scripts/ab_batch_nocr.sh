#!/bin/bash
# eight C2 windows through one bsgpu_solve_batch per step: C rows kept (BSGPU_NO_CR=0) / not, twice each
for i in 1 2; do
  for v in 0 1; do
    BSGPU_NO_CR=$v timeout 300 python bench.py --windows-per-gpu 8 --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('BSGPU_NO_CR=$v', d['value'], 'ms/step', d['ms_per_step'], 'cost', d['config'].get('final_cost'))"
  done
done
