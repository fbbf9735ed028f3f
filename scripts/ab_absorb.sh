# A/B of the dissection's separator absorption and hand-over constant on one box: bash scripts/ab_absorb.sh
run() { python bench.py --workload $1 --steps 15 --warmup 3 --no-cpu-baseline --no-past-l3 --no-other-configs --sustained-seconds 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d.get('phases_us_per_lm_step') or {}; print('$1 $2', d['value'], 'factor', p.get('factor'), 'backsolve', p.get('backsolve'))
"; }
for rep in 1 2; do
for w in c2 c3; do
  for hop in ${HOPS:-10 7 6 5 4}; do BSGPU_DIM_ABSORB=1 BSGPU_DIM_T_HOP=$hop run $w absorb=1,hop=$hop; done
done
done
for hop in ${HOPS:-10 7 6 5 4}; do echo hop=$hop; BSGPU_DIM_ABSORB=1 BSGPU_DIM_T_HOP=$hop python scripts/small_window.py | tail -4; done
