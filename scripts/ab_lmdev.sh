#!/bin/bash
# the decision on the device (BSGPU_LM_DEVICE) off / on: C2 twice each, then the iteration records side by side
B="timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-past-l3 --sustained-seconds 0"
ex() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], d['value'], 'ms', d['ms_per_step'], 'its', d['config']['lm_iterations_per_solve'], 'cost', d['config']['final_cost'])" "$1"; }
for i in 1 2; do
  BSGPU_LM_DEVICE=0 $B 2>/dev/null | ex "dev0 c2"
  BSGPU_LM_DEVICE=1 $B 2>/dev/null | ex "dev1 c2"
done
cat > /tmp/its.py <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
from beam_slam_amd import synthetic
from beam_slam_amd.gpu import GpuSolver
pr = synthetic.c2() if sys.argv[2] == "c2" else synthetic.vio_window(n_kf=20, n_lm=500, seed=20250620)
g = GpuSolver(0); pr.load(g)
o = g.options_vio(); o.max_solver_time_in_seconds = 0.0
s = g.solve(o)
for it in g.iterations(): print("%2d ok %d cost %.12e radius %.17g" % (it.iteration, it.step_is_successful, it.cost, it.trust_region_radius))
PY
for w in c2 small; do for d in 0 1; do echo "== $w BSGPU_LM_DEVICE=$d"; BSGPU_TIMING=1 BSGPU_LM_DEVICE=$d python /tmp/its.py $GRAFT_REPO_ROOT $w 2>&1 | grep -v "^\[bsgpu\] [a-z ]*:.*ms" | tail -16; done; done
