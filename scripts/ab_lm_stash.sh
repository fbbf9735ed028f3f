#!/bin/bash
# landmark_kernel with / without the first row kept in LDS between its two passes: kernel-trace average + bench value, one box.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
for v in 0 1 2 0 1 2; do
  echo "== BSGPU_LM_STASH=$v"
  BSGPU_LM_STASH=$v bash "$ROOT/scripts/kstats.sh" c2 8 2>&1 | grep -i "landmark\|value\|pairs_band\|chol_fused"
  cd "$ROOT"; BSGPU_LM_STASH=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('phases_us_per_lm_step') or {}
print('value', d['value'], 'landmark', p.get('landmark'), 'pairs', p.get('pairs'), 'factor', p.get('factor'))"
done
