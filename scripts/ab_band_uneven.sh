#!/bin/bash
# pairs_band_kernel with one unit per first camera pose (0) / a long and a short unit (share of the long one in percent): kernel-trace average + bench value, one box
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for v in 0 60 67 75 0 50 67 80; do
  echo "== BSGPU_BAND_UNEVEN=$v"
  BSGPU_BAND_UNEVEN=$v bash "$ROOT/scripts/kstats.sh" c2 8 2>&1 | grep -i "pairs_band\|value"
  cd "$ROOT"; BSGPU_BAND_UNEVEN=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('phases_us_per_lm_step') or {}
print('value', d['value'], 'landmark', p.get('landmark'), 'pairs', p.get('pairs'), 'factor', p.get('factor'), 'cost %.12e' % d['config']['final_cost'])"
done
