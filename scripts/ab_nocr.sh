#!/bin/bash
# C2 with (BSGPU_NO_CR=0) / without the C rows: kernel-trace averages + bench value, one box
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for v in 0 1 0 1; do
  echo "== BSGPU_NO_CR=$v"
  BSGPU_NO_CR=$v bash "$ROOT/scripts/kstats.sh" c2 10 2>&1 | grep -i "landmark\|backsub\|pairs_band\|value"
  cd "$ROOT"; BSGPU_NO_CR=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('phases_us_per_lm_step') or {}
print('value', d['value'], 'landmark', p.get('landmark'), 'pairs', p.get('pairs'), 'backsub', p.get('backsub'), 'cost %.12e' % d['config']['final_cost'])"
done
