#!/bin/bash
# the dissection's cost model (dim_order.h: t_chain0, t_hop, t_hop_tile) swept: C2, C3, the small windows, a lidar-inertial window, a pose graph per setting
#   bash scripts/ab_dim_model.sh "5 10 4" "3 9 8" ...
B="timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0"
ex() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d.get('phases_us_per_lm_step') or {}
print('   ', sys.argv[1], d['value'], 'factor', p.get('factor'), 'backsolve', p.get('backsolve'))" "$1"; }
for cfg in "$@"; do
  set -- $cfg
  export BSGPU_DIM_T_CHAIN0=$1 BSGPU_DIM_T_HOP=$2 BSGPU_DIM_T_HOP_TILE=$3
  echo "t_chain0 $1 t_hop $2 t_hop_tile $3"
  $B 2>/dev/null | ex c2
  $B --workload c3 2>/dev/null | ex c3
  python scripts/small_window.py 2>&1
  python scripts/lio_pg_rate.py
done
